// Probe: issue rate of v_mfma_f64_16x16x4_f64 - clocks per instruction with NCH independent accumulators per wave, one wave per
// SIMD (a 256-thread workgroup per CU), register operands only.  What gg_lp_mfma's 96 instructions per evaluation cost at best.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ __launch_bounds__(256) void rate(double* out, long long* clk, int reps) {
  d4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double sum = 0;
  for (int c = 0; c < NCH; ++c) sum += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  const long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* clk;
  hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&clk, 1024 * 8);
  const int reps = 200;
  for (int wgs = 256; wgs <= 512; wgs *= 2) {
    long long h[1024];
    rate<1><<<wgs, 256>>>(out, clk, reps); hipMemcpy(h, clk, wgs * 8, hipMemcpyDeviceToHost);
    printf("%d workgroups, 1 accumulator : %.1f clocks per MFMA (wall clock64 ticks x 24: clock64 at 100 MHz? raw %lld)\n", wgs, (double)h[0] / (reps * 16.0), h[0]);
    rate<2><<<wgs, 256>>>(out, clk, reps); hipMemcpy(h, clk, wgs * 8, hipMemcpyDeviceToHost);
    printf("%d workgroups, 2 accumulators: %.1f per MFMA\n", wgs, (double)h[0] / (reps * 32.0));
    rate<4><<<wgs, 256>>>(out, clk, reps); hipMemcpy(h, clk, wgs * 8, hipMemcpyDeviceToHost);
    printf("%d workgroups, 4 accumulators: %.1f per MFMA\n", wgs, (double)h[0] / (reps * 64.0));
    rate<6><<<wgs, 256>>>(out, clk, reps); hipMemcpy(h, clk, wgs * 8, hipMemcpyDeviceToHost);
    printf("%d workgroups, 6 accumulators: %.1f per MFMA\n", wgs, (double)h[0] / (reps * 96.0));
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); rate<6><<<256, 256>>>(out, clk, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("256 workgroups x 4 waves x 192000 MFMA: %.3f ms -> %.1f ns per MFMA per wave, %.1f TFLOP/s f64\n", ms, ms * 1e6 / 192000.0,
         256.0 * 4 * 192000 * 2048 / (ms * 1e-3) / 1e12);
  return 0;
}
