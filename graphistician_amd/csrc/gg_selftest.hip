// gg_selftest.hip — device-side evaluation of the shared arithmetic spec so that
// parity tests can compare gfx950 results with the host build bit for bit.
#include "gg_common.h"

__global__ void k_math(int op, const double* x, const double* y, double* out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i], b = y ? y[i] : 0.0;
  double r;
  switch (op) {
    case 0: r = gg_exp(a); break;
    case 1: r = gg_log(a); break;
    case 2: r = gg_sqrt(a); break;
    case 3: r = a / b; break;
    case 4: r = gg_fma(a, b, a); break;
    case 5: r = (double)(long long)a; break;            // double -> int64 -> double
    case 6: r = (double)((long long)a * (long long)b); break;
    default: r = 0.0;
  }
  out[i] = r;
}

// what: 0 normal, 1 gamma(a), 2 beta(a,b), 3 dirichlet(alpha[K]), 4 NIW posterior draw
__global__ void k_draws(int what, uint64_t seed, const double* prm, int nprm, double* out, long long n, int stride) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  gg_stream s;
  switch (what) {
    case 0: {
      // stream of normals is sequential in the host helper; here element i restarts the stream at index i
      gg_stream_init(&s, seed, GG_SITE_INIT, 0u, (uint32_t)i);
      out[i] = gg_normal(&s);
    } break;
    case 1:
      gg_stream_init(&s, seed, GG_SITE_INIT, 1u, (uint32_t)i);
      out[i] = gg_gamma(&s, prm[0]);
      break;
    case 2:
      gg_stream_init(&s, seed, GG_SITE_INIT, 2u, (uint32_t)i);
      out[i] = gg_beta(&s, prm[0], prm[1]);
      break;
    case 3: {
      double tmp[64];
      gg_dirichlet_at(seed, GG_SITE_INIT, (uint32_t)(16 + i), prm, nprm, tmp);
      for (int k = 0; k < nprm; ++k) out[i * stride + k] = tmp[k];
    } break;
    case 4: {
      // prm = [B, niw packed (B + B*B + 2), n_obs, s1 (B), s2 (B*B)]
      int B = (int)prm[0];
      gg_niw h;
      const double* p = prm + 1;
      for (int j = 0; j < GG_MAXB; ++j) h.mu_0[j] = 0.0;
      for (int j = 0; j < GG_MAXB * GG_MAXB; ++j) h.sigma_0[j] = 0.0;
      for (int j = 0; j < B; ++j) h.mu_0[j] = p[j];
      for (int j = 0; j < B * B; ++j) h.sigma_0[j] = p[B + j];
      h.nu_0 = p[B + B * B];
      h.kappa_0 = p[B + B * B + 1];
      const double* q = p + B + B * B + 2;
      double mu[GG_MAXB], sig[GG_MAXB * GG_MAXB];
      gg_stream_init(&s, seed, GG_SITE_INIT, 4u, (uint32_t)i);
      gg_niw_posterior_draw(&s, &h, B, q[0], q + 1, q + 1 + B, mu, sig);
      for (int j = 0; j < B; ++j) out[i * stride + j] = mu[j];
      for (int j = 0; j < B * B; ++j) out[i * stride + B + j] = sig[j];
    } break;
  }
}

extern "C" {

int gg_selftest_math(int32_t device, int32_t op, const double* x, const double* y, double* out, int64_t n) {
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  double *dx = nullptr, *dy = nullptr, *dout = nullptr;
  GG_HIP(hipMalloc(&dx, n * 8));
  GG_HIP(hipMalloc(&dout, n * 8));
  GG_HIP(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice));
  if (y) {
    GG_HIP(hipMalloc(&dy, n * 8));
    GG_HIP(hipMemcpy(dy, y, n * 8, hipMemcpyHostToDevice));
  }
  k_math<<<(unsigned)((n + 255) / 256), 256>>>(op, dx, dy, dout, n);
  GG_HIP(hipGetLastError());
  GG_HIP(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
  hipFree(dx); hipFree(dout); if (dy) hipFree(dy);
  return GG_OK;
}

int gg_selftest_draws(int32_t device, int32_t what, uint64_t seed, const double* params, int32_t nparams, double* out,
                      int64_t n, int32_t stride) {
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  double *dp = nullptr, *dout = nullptr;
  size_t no = (size_t)n * (stride > 0 ? stride : 1);
  GG_HIP(hipMalloc(&dp, (nparams > 0 ? nparams : 1) * 8));
  GG_HIP(hipMalloc(&dout, no * 8));
  if (nparams > 0) GG_HIP(hipMemcpy(dp, params, nparams * 8, hipMemcpyHostToDevice));
  int np = nparams;
  if (what == 4) np = (int)params[0];
  k_draws<<<(unsigned)((n + 63) / 64), 64>>>(what, seed, dp, what == 3 ? nparams : np, dout, n, stride > 0 ? stride : 1);
  GG_HIP(hipGetLastError());
  GG_HIP(hipMemcpy(out, dout, no * 8, hipMemcpyDeviceToHost));
  hipFree(dp); hipFree(dout);
  return GG_OK;
}

}  // extern "C"
