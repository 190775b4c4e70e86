// Probe: what an LDS access costs a wave on gfx950, in shader clocks, for the access shapes of k_eig_chain's scan
// (broadcast reads of a published vector, per-group gathers, strided table reads), alone and with other waves of the
// workgroup doing the same.   hipcc --offload-arch=gfx950 -O3 lds_latency.hip -o lds_latency && ./lds_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k_probe(long long* out, int iters) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 16384; e += blockDim.x) lds[e] = (double)e;
  __syncthreads();
  double acc = 0.0;
  int off = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");
    if (MODE == 0) {          // broadcast: all lanes read the same 64 bytes as 4 x b128
      const d2* p = reinterpret_cast<const d2*>(lds + 8 * (it & 63) + off);
      d2 a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    } else if (MODE == 1) {   // broadcast: 8 x b64 (kept apart by a volatile asm)
      const double* p = lds + 8 * (it & 63) + off;
#pragma unroll
      for (int m = 0; m < 8; ++m) { double v = p[m]; asm volatile("" : "+v"(v)); acc += v; }
    } else if (MODE == 2) {   // one dword, broadcast
      const int* p = reinterpret_cast<const int*>(lds + 8 * (it & 63) + off);
      acc += (double)p[1];
    } else if (MODE == 3) {   // per group of 4 lanes: 64 bytes at stride 64 bytes (the q gather)
      const d2* p = reinterpret_cast<const d2*>(lds + 8 * (lane >> 2) + 512 * wave + off);
      d2 a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    } else if (MODE == 4) {   // one b64 per lane, stride 65 doubles (table column)
      acc += lds[(lane * 65 + (it & 63)) + off];
    } else if (MODE == 5) {   // b64 write then read back of the group's values (gather round trip)
      lds[8192 + tid] = acc;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const d2* p = reinterpret_cast<const d2*>(lds + 8192 + (tid & ~7));
      d2 a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    off = ((int)acc) & 0;    // dependency: the next address waits for this value
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  if (acc == 1.2345) out[63] = 1;
}

template <int MODE>
static void run(const char* what, int waves) {
  long long* d; hipMalloc(&d, 64 * 8);
  const int iters = 2000;
  k_probe<MODE><<<1, 64 * waves, 160 * 1024 - 2048>>>(d, iters);   // LDS request as large as the chain's: alone on its CU
  k_probe<MODE><<<1, 64 * waves, 160 * 1024 - 2048>>>(d, iters);
  long long h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s %d waves: %7.1f clocks per dependent access (wave 0), %7.1f (last wave)\n", what, waves, (double)h[0] / iters, (double)h[waves - 1] / iters);
  hipFree(d);
}

int main() {
  hipFuncSetAttribute((const void*)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipFuncSetAttribute((const void*)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipFuncSetAttribute((const void*)k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipFuncSetAttribute((const void*)k_probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipFuncSetAttribute((const void*)k_probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipFuncSetAttribute((const void*)k_probe<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  for (int w : {1, 4, 8}) {
    run<0>("broadcast 64 B as 4 x ds_read_b128", w);
    run<1>("broadcast 64 B as 8 x ds_read_b64", w);
    run<2>("broadcast one dword", w);
    run<3>("64 B per group of 4 lanes, groups 64 B apart (4 x b128)", w);
    run<4>("one b64 per lane, stride 65 doubles", w);
    run<5>("write b64 + wait + read the group's 64 B", w);
  }
  return 0;
}
