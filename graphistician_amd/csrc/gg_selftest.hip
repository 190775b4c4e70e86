// gg_selftest.hip — device-side evaluation of the shared arithmetic spec so that
// parity tests can compare gfx950 results with the host build bit for bit.
#include "gg_common.h"
#include "spec/gg_pg.h"
#include "spec/gg_probit.h"

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

// what: 0 normal, 1 gamma(a), 2 beta(a,b), 3 dirichlet(alpha[K]), 4 NIW posterior draw,
//       5 truncated normal N(mu, 1) on z > 0 (a != 0) / z < 0 (a == 0): prm = {mu, a} (a < 0: the sign is drawn first,
//         A ~ Bern(1 - Phi(-mu)), the two-stage construction of test/test_distributions.py:39-56),
//       6 PG(1, z): prm = {z},  7 categorical draw from prm = lp[K] (one uniform per draw)
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
    case 5: {
      const double mu = prm[0];
      int a = (int)prm[1];
      if (a < 0) a = gg_uniform_at(seed, GG_SITE_INIT, 6u, (uint32_t)i) < 1.0 - gg_norm_cdf(-mu) ? 1 : 0;
      out[i] = gg_truncnorm(gg_uniform_at(seed, GG_SITE_INIT, 5u, (uint32_t)i), mu, a);
    } break;
    case 6:
      gg_stream_init(&s, seed, GG_SITE_INIT, 7u, (uint32_t)i);
      out[i] = gg_pg1(&s, prm[0]);
      break;
    case 7: {
      double scratch[128];
      out[i] = (double)gg_categorical(prm, nprm, gg_uniform_at(seed, GG_SITE_INIT, 8u, (uint32_t)i), scratch);
    } break;
  }
}

typedef unsigned int gg_probe_u4 __attribute__((ext_vector_type(4)));
template <bool NTL>
__global__ __launch_bounds__(256) void k_stream_probe(const gg_probe_u4* __restrict__ src, gg_probe_u4* __restrict__ dst, size_t n16,
                                                      unsigned long long* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int acc = 0;
  auto ld = [&](size_t k) { return NTL ? __builtin_nontemporal_load(&src[k]) : src[k]; };
  for (; i + 3 * stride < n16; i += 4 * stride) {   // four independent 16-byte loads in flight per lane
    gg_probe_u4 a = ld(i), b = ld(i + stride), c = ld(i + 2 * stride), d = ld(i + 3 * stride);
    if (dst) { dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d; }
    else acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) {
    gg_probe_u4 a = ld(i);
    if (dst) dst[i] = a; else acc += a.x ^ a.y ^ a.z ^ a.w;
  }
  if (!dst && acc == 0x12345u) atomicAdd(sink, 1ull);   // (never true for the fill pattern; keeps the loads)
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
  k_draws<<<(unsigned)((n + 63) / 64), 64>>>(what, seed, dp, (what == 3 || what == 7) ? nparams : np, dout, n, stride > 0 ? stride : 1);
  GG_HIP(hipGetLastError());
  GG_HIP(hipMemcpy(out, dout, no * 8, hipMemcpyDeviceToHost));
  hipFree(dp); hipFree(dout);
  return GG_OK;
}

// Memory-system probe for the roofline: the rate at which this device streams `bytes` of HBM-resident data
// with the access pattern of the bulk kernels (16-byte loads, every lane its own column of a row-major
// matrix) and nothing else.  mode 0: read only (a checksum keeps the loads alive); mode 1: copy (read + write).
int gg_bench_stream(int32_t device, int64_t bytes, int32_t reps, int32_t mode, double* ms_per_pass) {
  if (bytes < (1 << 20) || reps < 1 || !ms_per_pass || (mode != 0 && mode != 1))
    return gg_fail(GG_ERR_INVALID, "gg_bench_stream: bad arguments");
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  const size_t n16 = (size_t)bytes / 16;
  gg_probe_u4 *src = nullptr, *dst = nullptr;
  unsigned long long* sink = nullptr;
  GG_HIP(hipMalloc((void**)&src, n16 * 16));
  GG_HIP(hipMemset(src, 1, n16 * 16));
  if (mode == 1) GG_HIP(hipMalloc((void**)&dst, n16 * 16));
  GG_HIP(hipMalloc((void**)&sink, 8));
  GG_HIP(hipMemset(sink, 0, 8));
  hipEvent_t e0, e1;
  GG_HIP(hipEventCreate(&e0));
  GG_HIP(hipEventCreate(&e1));
  const int nb = 256 * 16;   // 16 workgroups per CU, grid-strided
  // the same streaming hint as the bulk kernels: non-temporal loads beyond the Infinity Cache (gg_stream_load)
  const bool ntl = (double)bytes > gg_nt_threshold();
  if (ntl) k_stream_probe<true><<<nb, 256>>>(src, dst, n16, sink); else k_stream_probe<false><<<nb, 256>>>(src, dst, n16, sink);   // warm-up
  GG_HIP(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) {
    if (ntl) k_stream_probe<true><<<nb, 256>>>(src, dst, n16, sink); else k_stream_probe<false><<<nb, 256>>>(src, dst, n16, sink);
  }
  GG_HIP(hipEventRecord(e1, 0));
  GG_HIP(hipEventSynchronize(e1));
  GG_HIP(hipGetLastError());
  float ms = 0;
  GG_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_per_pass = ms / reps;
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(src); if (dst) hipFree(dst); hipFree(sink);
  return GG_OK;
}

}  // extern "C"
