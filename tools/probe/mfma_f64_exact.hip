// Probe: is v_mfma_f64_16x16x4_f64 bit-for-bit a chain of four fused multiply-adds in k order?
//   D[i][j] = fma(A[i][3], B[3][j], fma(A[i][2], B[2][j], fma(A[i][1], B[1][j], fma(A[i][0], B[0][j], C[i][j]))))
// If it is, a block conditional's chains (spec/gg_lp.h: one fma chain per (direction, statistic), j ascending) can run on the
// matrix cores with the same bits.  Candidates compared on the host: fused ascending, fused descending, unfused ascending,
// pairwise.  Two MFMAs in a row (k = 0..7) are checked as a chain of eight.
// Layout (tools/probe/mfma_f64_layout.hip): lane l holds A[l % 16][l / 16], B[l / 16][l % 16]; result r of lane l = D[4 r + l / 16][l % 16].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void run(const double* A, const double* B, const double* C, double* D, int nk4) {
  // A: [trial][16][4 nk4], B: [trial][4 nk4][16], C, D: [trial][16][16]; one wave per trial
  const int l = threadIdx.x, tr = blockIdx.x, K4 = 4 * nk4;
  d4 acc;
  for (int r = 0; r < 4; ++r) acc[r] = C[(size_t)tr * 256 + (4 * r + l / 16) * 16 + l % 16];
  for (int s = 0; s < nk4; ++s) {
    const double a = A[(size_t)tr * 16 * K4 + (l % 16) * K4 + 4 * s + l / 16];
    const double b = B[(size_t)tr * K4 * 16 + (4 * s + l / 16) * 16 + l % 16];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(size_t)tr * 256 + (4 * r + l / 16) * 16 + l % 16] = acc[r];
}

static double rnd(int spread) {
  const double m = (double)rand() / RAND_MAX * 2.0 - 1.0;
  const int e = rand() % (2 * spread + 1) - spread;
  return ldexp(m, e);
}

int main() {
  const int T = 2048;
  for (int nk4 = 1; nk4 <= 16; nk4 *= 4)
    for (int spread = 0; spread <= 40; spread += 20) {
      const int K4 = 4 * nk4;
      const size_t na = (size_t)T * 16 * K4, nc = (size_t)T * 256;
      double *hA = (double*)malloc(na * 8), *hB = (double*)malloc(na * 8), *hC = (double*)malloc(nc * 8), *hD = (double*)malloc(nc * 8);
      srand(1234 + nk4 + spread);
      for (size_t i = 0; i < na; ++i) { hA[i] = rnd(spread); hB[i] = rnd(spread); }
      for (size_t i = 0; i < nc; ++i) hC[i] = (rand() % 4 == 0) ? 0.0 : rnd(spread);
      double *dA, *dB, *dC, *dD;
      hipMalloc(&dA, na * 8); hipMalloc(&dB, na * 8); hipMalloc(&dC, nc * 8); hipMalloc(&dD, nc * 8);
      hipMemcpy(dA, hA, na * 8, hipMemcpyHostToDevice); hipMemcpy(dB, hB, na * 8, hipMemcpyHostToDevice);
      hipMemcpy(dC, hC, nc * 8, hipMemcpyHostToDevice);
      run<<<T, 64>>>(dA, dB, dC, dD, nk4);
      hipMemcpy(hD, dD, nc * 8, hipMemcpyDeviceToHost);
      long long same[4] = {0, 0, 0, 0}, total = 0;
      for (int tr = 0; tr < T; ++tr)
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) {
            const double* a = hA + (size_t)tr * 16 * K4 + i * K4;
            const double* b = hB + (size_t)tr * K4 * 16 + j;
            const double c = hC[(size_t)tr * 256 + i * 16 + j];
            double v0 = c, v1 = c, v2 = c, v3 = c;
            for (int k = 0; k < K4; ++k) v0 = fma(a[k], b[k * 16], v0);
            for (int s = 0; s < nk4; ++s)
              for (int k = 3; k >= 0; --k) v1 = fma(a[4 * s + k], b[(4 * s + k) * 16], v1);
            for (int k = 0; k < K4; ++k) { volatile double p = a[k] * b[k * 16]; v2 = v2 + p; }
            for (int s = 0; s < nk4; ++s) {
              volatile double p0 = a[4 * s] * b[(4 * s) * 16], p1 = a[4 * s + 1] * b[(4 * s + 1) * 16];
              volatile double p2 = a[4 * s + 2] * b[(4 * s + 2) * 16], p3 = a[4 * s + 3] * b[(4 * s + 3) * 16];
              v3 = v3 + ((p0 + p1) + (p2 + p3));
            }
            const double d = hD[(size_t)tr * 256 + i * 16 + j];
            total += 1;
            same[0] += memcmp(&d, &v0, 8) == 0; same[1] += memcmp(&d, &v1, 8) == 0;
            same[2] += memcmp(&d, &v2, 8) == 0; same[3] += memcmp(&d, &v3, 8) == 0;
          }
      printf("k = %2d, exponent spread +-%2d: %lld outputs; identical to fused ascending %lld, fused descending %lld, unfused ascending %lld, pairwise %lld\n",
             K4, spread, total, same[0], same[1], same[2], same[3]);
      hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
      free(hA); free(hB); free(hC); free(hD);
    }
  return 0;
}
