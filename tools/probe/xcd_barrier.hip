// xcd_barrier.hip — how long does a slot exchange between G workgroups take when they all sit on ONE XCD (L2-coherent,
// workgroup-scope = sc0 accesses) compared with workgroups spread over the chip (agent scope)?  Probe for a restart-round
// scan confined to one XCD.  Grid = 8 * G workgroups; with SAMEXCD only those with blockIdx % 8 == 0 take part
// (round-robin dispatch puts them on the same XCD: checked with XCC_ID).
//   hipcc -O3 --offload-arch=gfx950 -o xcd_barrier xcd_barrier.hip && ./xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SCOPE, int PAD, int FENCE>   // SCOPE 0: workgroup scope (sc0), 1: agent scope; PAD: slot stride in 8-byte words; FENCE: __threadfence around the exchange
__global__ __launch_bounds__(256) void k_exchange(unsigned long long* slots, int G, int stride, int rounds, long long* out, int* xcc) {
  if (blockIdx.x % stride) return;
  const int me = blockIdx.x / stride;
  if (me >= G) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[me] = (int)(id & 0xf);
  const int tid = threadIdx.x;
  long long t0 = 0;
  int fail = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (r == 2 && tid == 0) t0 = wall_clock64();
    unsigned long long* s = slots + (size_t)(r & 1) * 256 * PAD;
    __syncthreads();
    if (tid == 0) {
      if (FENCE) __threadfence();
      if (SCOPE) __hip_atomic_store(&s[(size_t)me * PAD], (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(&s[(size_t)me * PAD], (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    int done = 0;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      int ok = 1;
      if (tid < G) {
        unsigned long long w;
        if (SCOPE) w = __hip_atomic_load(&s[(size_t)tid * PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else w = __hip_atomic_load(&s[(size_t)tid * PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ok = w == (unsigned long long)r;
      }
      if (__syncthreads_and(ok)) { done = 1; break; }
    }
    if (!done) { fail = 1; break; }
    if (FENCE) { __threadfence(); __syncthreads(); }
  }
  if (tid == 0) { out[me * 2] = wall_clock64() - t0; out[me * 2 + 1] = fail; }
}

int main() {
  unsigned long long* slots; long long* out; int* xcc;
  CHECK(hipMalloc(&slots, 2 * 256 * 16 * 8)); CHECK(hipMalloc(&out, 512 * 8)); CHECK(hipMalloc(&xcc, 256 * 4));
  const int rounds = 2001;
  const char* names[] = {"same XCD, workgroup scope", "same XCD, agent scope", "spread, agent scope", "spread, agent, padded slots", "spread, agent, fences", "spread, agent, padded, fences"};
  for (int G : {16, 32, 64, 128, 256}) {
    for (int mode = 0; mode < 6; ++mode) {
      if (mode < 2 && G > 32) continue;          // one XCD holds 32 CUs
      if (mode == 0 && G > 16) continue;         // (times out: stale L1 lines)
      CHECK(hipMemset(slots, 0, 2 * 256 * 16 * 8));
      const int stride = mode >= 2 ? 1 : 8;
      switch (mode) {
        case 0: k_exchange<0, 1, 0><<<G * stride, 256>>>(slots, G, stride, rounds, out, xcc); break;
        case 1: case 2: k_exchange<1, 1, 0><<<G * stride, 256>>>(slots, G, stride, rounds, out, xcc); break;
        case 3: k_exchange<1, 16, 0><<<G * stride, 256>>>(slots, G, stride, rounds, out, xcc); break;
        case 4: k_exchange<1, 1, 1><<<G * stride, 256>>>(slots, G, stride, rounds, out, xcc); break;
        default: k_exchange<1, 16, 1><<<G * stride, 256>>>(slots, G, stride, rounds, out, xcc); break;
      }
      CHECK(hipDeviceSynchronize());
      long long h[512]; int hx[256];
      CHECK(hipMemcpy(h, out, 512 * 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hx, xcc, 256 * 4, hipMemcpyDeviceToHost));
      int same = 1, fail = 0;
      for (int i = 0; i < G; ++i) { if (hx[i] != hx[0]) same = 0; fail |= (int)h[i * 2 + 1]; }
      printf("G=%3d %-32s: %6.2f us per exchange   one XCC: %-3s %s\n", G, names[mode], h[0] * 0.01 / (rounds - 1), same ? "yes" : "no",
             fail ? "TIMEOUT" : "ok");
    }
  }
  return 0;
}
