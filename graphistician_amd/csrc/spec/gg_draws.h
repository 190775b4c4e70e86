// gg_draws.h — the O(K^2) / O(B^3) conjugate draws of the Gibbs sweep, written
// once for host and device so that both consume the same Philox stream and
// return the same bits.
//
//   Beta       adjacency.py:638,648 (np.random.beta)
//   Dirichlet  adjacency.py:702, weights.py:498 (np.random.dirichlet)
//   NIW        weights.py:161-162,408,413 -> pybasicbayes Gaussian.resample
//              (not vendored): posterior
//                kappa_n = kappa_0 + n, nu_n = nu_0 + n,
//                mu_n = (kappa_0 mu_0 + sum x) / kappa_n,
//                Sigma_n = Sigma_0 + S + kappa_0 n / kappa_n (xbar-mu_0)(xbar-mu_0)^T,
//              S the scatter about xbar; then Sigma ~ IW(Sigma_n, nu_n),
//              mu ~ N(mu_n, Sigma / kappa_n).
// Stream parity with numpy's Mersenne Twister is impossible;
// distributional correctness is asserted by tests/test_draws.py.
#pragma once
#include "gg_math.h"

#define GG_MAXB 4

// Standard normal by Marsaglia's polar method (needs only log, sqrt, divide).
GG_HD double gg_normal(gg_stream* s) {
  for (;;) {
    double u = 2.0 * gg_stream_uniform(s) - 1.0;
    double v = 2.0 * gg_stream_uniform(s) - 1.0;
    double r = u * u + v * v;
    if (r < 1.0 && r > 0.0) {
      return u * gg_sqrt(-2.0 * gg_log(r) / r);
    }
  }
}

// Gamma(shape a, scale 1), Marsaglia & Tsang (2000).
GG_HD double gg_gamma(gg_stream* s, double a) {
  if (!(a > 0.0)) return __builtin_nan("");   // (a shape that is not positive - or NaN - would never leave the loop below)
  double boost = 1.0;
  if (a < 1.0) {
    double u = gg_stream_uniform_pos(s);
    boost = gg_pow_pos(u, 1.0 / a);
    a += 1.0;
  }
  double d = a - 1.0 / 3.0;
  double c = 1.0 / gg_sqrt(9.0 * d);
  for (;;) {
    double x = gg_normal(s);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    double u = gg_stream_uniform_pos(s);
    double x2 = x * x;
    if (u < 1.0 - 0.0331 * x2 * x2) return boost * d * v;
    if (gg_log(u) < 0.5 * x2 + d * (1.0 - v + gg_log(v))) return boost * d * v;
  }
}

GG_HD double gg_beta(gg_stream* s, double a, double b) {
  double x = gg_gamma(s, a);
  double y = gg_gamma(s, b);
  return x / (x + y);
}

// Dirichlet(alpha): component k is Gamma(alpha_k) from its own stream
// (seed, site, sweep, index = k), so the K gammas can be drawn by K device
// threads; the normalising sum is taken in ascending k.
GG_HD double gg_dirichlet_gamma_at(uint64_t seed, uint32_t site, uint32_t sweep, int k, double alpha_k) {
  gg_stream s;
  gg_stream_init(&s, seed, site, sweep, (uint32_t)k);
  return gg_gamma(&s, alpha_k);
}

GG_HD void gg_dirichlet_normalise(const double* g, int K, double* out) {
  double tot = 0.0;
  for (int k = 0; k < K; ++k) tot += g[k];
  for (int k = 0; k < K; ++k) out[k] = g[k] / tot;
}

GG_HD void gg_dirichlet_at(uint64_t seed, uint32_t site, uint32_t sweep, const double* alpha, int K, double* out) {
  for (int k = 0; k < K; ++k) out[k] = gg_dirichlet_gamma_at(seed, site, sweep, k, alpha[k]);
  double tot = 0.0;
  for (int k = 0; k < K; ++k) tot += out[k];
  for (int k = 0; k < K; ++k) out[k] = out[k] / tot;
}

// ---- small dense linear algebra (B <= GG_MAXB), row-major B x B ------------

// Lower Cholesky factor; returns 0 on success.
GG_HD int gg_chol(const double* S, int B, double* L) {
  for (int i = 0; i < B * B; ++i) L[i] = 0.0;
  for (int j = 0; j < B; ++j) {
    double d = S[j * B + j];
    for (int k = 0; k < j; ++k) d -= L[j * B + k] * L[j * B + k];
    if (!(d > 0.0)) return 1;
    double dj = gg_sqrt(d);
    L[j * B + j] = dj;
    for (int i = j + 1; i < B; ++i) {
      double v = S[i * B + j];
      for (int k = 0; k < j; ++k) v -= L[i * B + k] * L[j * B + k];
      L[i * B + j] = v / dj;
    }
  }
  return 0;
}

// Inverse of a lower-triangular matrix.
GG_HD void gg_tri_inv(const double* L, int B, double* Li) {
  for (int i = 0; i < B * B; ++i) Li[i] = 0.0;
  for (int j = 0; j < B; ++j) {
    Li[j * B + j] = 1.0 / L[j * B + j];
    for (int i = j + 1; i < B; ++i) {
      double v = 0.0;
      for (int k = j; k < i; ++k) v -= L[i * B + k] * Li[k * B + j];
      Li[i * B + j] = v / L[i * B + i];
    }
  }
}

// Precision P = Sigma^{-1} and log det Sigma from Sigma (SPD); returns 0 on success.
GG_HD int gg_spd_inv_logdet(const double* S, int B, double* P, double* logdet) {
  double L[GG_MAXB * GG_MAXB], Li[GG_MAXB * GG_MAXB];
  if (gg_chol(S, B, L)) return 1;
  gg_tri_inv(L, B, Li);
  double ld = 0.0;
  for (int i = 0; i < B; ++i) ld += gg_log(L[i * B + i]);
  *logdet = 2.0 * ld;
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) {
      double v = 0.0;
      int k0 = i > j ? i : j;
      for (int k = k0; k < B; ++k) v += Li[k * B + i] * Li[k * B + j];
      P[i * B + j] = v;
    }
  return 0;
}

// NIW hyper-parameters.
struct gg_niw {
  double mu_0[GG_MAXB];
  double sigma_0[GG_MAXB * GG_MAXB];
  double nu_0;
  double kappa_0;
};

// Sigma ~ IW(S, nu) via the Bartlett decomposition; mu ~ N(mu_n, Sigma / kappa).
GG_HD void gg_niw_sample(gg_stream* s, const double* mu_n, const double* S, double kappa, double nu,
                         int B, double* mu_out, double* sigma_out) {
  double L[GG_MAXB * GG_MAXB], A[GG_MAXB * GG_MAXB], Ai[GG_MAXB * GG_MAXB], C[GG_MAXB * GG_MAXB];
  gg_chol(S, B, L);
  for (int i = 0; i < B * B; ++i) A[i] = 0.0;
  for (int i = 0; i < B; ++i) {
    // chi-square(nu - i) = 2 Gamma((nu - i)/2)
    A[i * B + i] = gg_sqrt(2.0 * gg_gamma(s, 0.5 * (nu - (double)i)));
    for (int j = 0; j < i; ++j) A[i * B + j] = gg_normal(s);
  }
  gg_tri_inv(A, B, Ai);
  // C = L * A^{-T};  Sigma = C C^T
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) {
      double v = 0.0;
      for (int k = 0; k < B; ++k) v += L[i * B + k] * Ai[j * B + k];
      C[i * B + j] = v;
    }
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) {
      double v = 0.0;
      for (int k = 0; k < B; ++k) v += C[i * B + k] * C[j * B + k];
      sigma_out[i * B + j] = v;
    }
  // symmetrise exactly
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < i; ++j) sigma_out[j * B + i] = sigma_out[i * B + j];
  double Ls[GG_MAXB * GG_MAXB];
  gg_chol(sigma_out, B, Ls);
  double z[GG_MAXB];
  for (int i = 0; i < B; ++i) z[i] = gg_normal(s);
  double sk = 1.0 / gg_sqrt(kappa);
  for (int i = 0; i < B; ++i) {
    double v = 0.0;
    for (int k = 0; k <= i; ++k) v += Ls[i * B + k] * z[k];
    mu_out[i] = mu_n[i] + sk * v;
  }
}

// Posterior NIW draw from sufficient statistics n, s1 = sum x, s2 = sum x x^T
// (s2 row-major B x B, symmetric).  n == 0 gives a prior draw
// (weights.py:118-119 `_gaussian.resample()`).
GG_HD void gg_niw_posterior_draw(gg_stream* s, const gg_niw* h, int B, double n, const double* s1,
                                 const double* s2, double* mu_out, double* sigma_out) {
  double mu_n[GG_MAXB], S_n[GG_MAXB * GG_MAXB];
  double kappa_n = h->kappa_0 + n;
  double nu_n = h->nu_0 + n;
  if (n > 0.0) {
    double xbar[GG_MAXB];
    for (int i = 0; i < B; ++i) xbar[i] = s1[i] / n;
    for (int i = 0; i < B; ++i) mu_n[i] = (h->kappa_0 * h->mu_0[i] + s1[i]) / kappa_n;
    double w = h->kappa_0 * n / kappa_n;
    for (int i = 0; i < B; ++i)
      for (int j = 0; j < B; ++j) {
        double scatter = s2[i * B + j] - n * xbar[i] * xbar[j];
        double dev = (xbar[i] - h->mu_0[i]) * (xbar[j] - h->mu_0[j]);
        S_n[i * B + j] = h->sigma_0[i * B + j] + scatter + w * dev;
      }
  } else {
    for (int i = 0; i < B; ++i) mu_n[i] = h->mu_0[i];
    for (int i = 0; i < B * B; ++i) S_n[i] = h->sigma_0[i];
  }
  gg_niw_sample(s, mu_n, S_n, kappa_n, nu_n, B, mu_out, sigma_out);
}
