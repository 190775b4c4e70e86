// cu_mask.hip — what does hipExtStreamCreateWithCUMask's bit i mean on MI355X (256 CUs in 8 XCDs)?  A kernel of many short
// workgroups records (XCC_ID, SE_ID, CU_ID) per workgroup; the host counts the distinct compute units a masked stream used.
//   hipcc -O3 --offload-arch=gfx950 -o cu_mask.bin cu_mask.hip && ./cu_mask.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_where(unsigned* out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}

static void run(const char* name, const std::vector<uint32_t>& mask, unsigned* d_out, int nwg) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%-44s: create failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); return; }
  CHECK(hipMemsetAsync(d_out, 0xff, nwg * 4, s));
  k_where<<<nwg, 256, 0, s>>>(d_out, 2000);      // ~20 us at 100 MHz wall clock
  CHECK(hipStreamSynchronize(s));
  std::vector<unsigned> h(nwg);
  CHECK(hipMemcpy(h.data(), d_out, nwg * 4, hipMemcpyDeviceToHost));
  std::set<unsigned> cus;
  int per_xcc[16] = {0};
  std::set<unsigned> seen_x[16];
  for (unsigned v : h) {
    const unsigned xcc = v >> 16, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cus.insert(key);
    seen_x[xcc & 15].insert(key);
  }
  int bits = 0;
  for (uint32_t w : mask) bits += __builtin_popcount(w);
  printf("%-44s: mask bits set %3d -> distinct CUs used %3zu ; per XCC:", name, bits, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %zu", seen_x[x].size());
  printf("\n");
  CHECK(hipStreamDestroy(s));
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount, nw = (ncu + 31) / 32, nwg = 16384;
  printf("multiProcessorCount %d\n", ncu);
  unsigned* d_out;
  CHECK(hipMalloc(&d_out, nwg * 4));
  std::vector<uint32_t> all(nw, 0xffffffffu);
  run("all CUs", all, d_out, nwg);
  { auto m = all; for (int c = 0; c < ncu; c += 32) m[c / 32] &= ~(1u << (c % 32)); run("every 32nd bit cleared", m, d_out, nwg); }
  { auto m = all; for (int c = 0; c < ncu; c += 2) m[c / 32] &= ~(1u << (c % 32)); run("every 2nd bit cleared", m, d_out, nwg); }
  { auto m = all; for (int w = 0; w < nw / 2; ++w) m[w] = 0; run("first half of the words cleared", m, d_out, nwg); }
  { std::vector<uint32_t> m(nw, 0); m[0] = 0xffffffffu; run("only word 0 set", m, d_out, nwg); }
  { std::vector<uint32_t> m(nw, 0); m[0] = 0xffu; run("only bits 0..7 set", m, d_out, nwg); }
  { std::vector<uint32_t> m(nw, 0); for (int w = 0; w < nw; ++w) m[w] = 0x1u; run("bit 0 of every word set", m, d_out, nwg); }
  { std::vector<uint32_t> m(nw, 0xfffffff0u); run("bits 0..3 of every word cleared", m, d_out, nwg); }
  { std::vector<uint32_t> m(1, 0xffff0000u); run("ONE word, upper half set", m, d_out, nwg); }
  { std::vector<uint32_t> m(2, 0xffffffffu); m[0] = 0xfffffff0u; run("TWO words, bits 0..3 cleared", m, d_out, nwg); }
  return 0;
}
