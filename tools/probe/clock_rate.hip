// Probe: how fast clock64() (s_memtime) ticks on this device, against hipEvent time.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(long long ticks, long long* out) {
  const long long t0 = clock64();
  long long t = t0;
  while (t - t0 < ticks) t = clock64();
  const long long w0 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t - t0; out[1] = w0; }
}
int main() {
  long long* d; hipMalloc(&d, 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    k_spin<<<1, 64>>>(200000000LL, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("clock64 ticks %lld in %.3f ms -> %.1f MHz\n", h[0], ms, h[0] / (ms * 1e3));
  }
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("hipDeviceAttributeClockRate %d kHz\n", clk);
  return 0;
}
