// Probe: cost in shader clocks of the dependent arithmetic of one Sherman-Morrison fold of k_eig_chain (q = P u as two
// chains of four fma + add, two group sums over 8 lanes by DPP, reciprocal + two Newton steps, the r update) for one wave,
// alone and with a second wave on the same SIMD; and of its pieces.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k_fold(long long* out, int iters, double seedv) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double P[8], u[8], r = 0.3 + 1e-3 * lane, acc = seedv;
#pragma unroll
  for (int m = 0; m < 8; ++m) { P[m] = 0.01 * (m + 1) + 1e-4 * lane; u[m] = 0.1 * (m + 1); }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    double x = acc;
    if (MODE == 0 || MODE == 1) {          // q: two chains of four fma + add
      double qa = 0.0, qb = 0.0;
#pragma unroll
      for (int m = 0; m < 4; ++m) { qa = __builtin_fma(P[m], u[m] + x, qa); qb = __builtin_fma(P[4 + m], u[4 + m] + x, qb); }
      x = qa + qb;
    }
    if (MODE == 0 || MODE == 2) {          // two group sums over 8 lanes
      double tq = x * 0.5, d = x * r;
      tq += dpp_f64<0xB1>(tq); d += dpp_f64<0xB1>(d);
      tq += dpp_f64<0x4E>(tq); d += dpp_f64<0x4E>(d);
      tq += dpp_f64<0x141>(tq); d += dpp_f64<0x141>(d);
      x = tq + d;
    }
    if (MODE == 0 || MODE == 3) {          // reciprocal + two Newton steps + the r update
      const double den = __builtin_fma(0.7, x, 1.0);
      double rc = __builtin_amdgcn_rcp(den);
      rc = rc * __builtin_fma(-den, rc, 2.0);
      rc = rc * __builtin_fma(-den, rc, 2.0);
      x = __builtin_fma(__builtin_fma(-0.7, x, 0.2) * rc, x, r);
    }
    if (MODE == 4) {                        // eight dependent fma
#pragma unroll
      for (int m = 0; m < 8; ++m) x = __builtin_fma(P[m], x, u[m]);
    }
    acc = x * 1e-3;
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = t1 - t0;
  if (acc == 1.2345) out[63] = 1;
}

template <int MODE>
static void run(const char* what, int waves) {
  long long* d; (void)hipMalloc(&d, 64 * 8);
  const int iters = 4000;
  k_fold<MODE><<<1, 64 * waves>>>(d, iters, 0.5);
  k_fold<MODE><<<1, 64 * waves>>>(d, iters, 0.5);
  long long h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-52s %d waves: %7.1f clocks per iteration\n", what, waves, (double)h[0] / iters);
  (void)hipFree(d);
}

int main() {
  for (int w : {1, 4, 8}) {
    run<0>("whole fold (q, sums, reciprocal, r)", w);
    run<1>("q = P u: two chains of four fma + add", w);
    run<2>("two group sums over 8 lanes (DPP)", w);
    run<3>("reciprocal + two Newton steps + r update", w);
    run<4>("eight dependent fma", w);
  }
  return 0;
}
