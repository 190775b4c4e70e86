// Probe of the v_mfma_f64_16x16x4_f64 fragment layout: D = A(16x4) * B(4x16).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int mode) {
  int l = threadIdx.x;
  // hypothesis for inputs: A[i = l%16][k = l/16], B[k = l/16][j = l%16]
  int i = l % 16, k = l / 16, j = l % 16;
  double a, b;
  if (mode == 0) { a = (k == 0) ? (double)i : 0.0; b = (k == 0) ? 1.0 : 0.0; }          // D[i][j] = i
  else if (mode == 1) { a = (k == 0) ? 1.0 : 0.0; b = (k == 0) ? (double)j : 0.0; }     // D[i][j] = j
  else { a = (double)(k + 1); b = (k == 2) ? 1.0 : 0.0; }                               // D = 3 if k pairing is right
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(mode * 64 + l) * 4 + r] = acc[r];
}
int main() {
  double* d; hipMalloc(&d, 3 * 64 * 4 * 8);
  for (int m = 0; m < 3; ++m) probe<<<1, 64>>>(d, m);
  double h[3 * 64 * 4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int m = 0; m < 3; ++m) {
    printf("mode %d\n", m);
    for (int l = 0; l < 64; l += 1) if (l < 4 || l % 16 == 0 || l == 17) printf(" lane %2d: %g %g %g %g\n", l, h[(m*64+l)*4], h[(m*64+l)*4+1], h[(m*64+l)*4+2], h[(m*64+l)*4+3]);
  }
  return 0;
}
