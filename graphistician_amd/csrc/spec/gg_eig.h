// gg_eig.h — the latent-feature draw of the eigenmodels, shared by the HIP engine and the CPU oracle.
//
// Reference: eigenmodel.py:876-908 (_GibbsLogisticEigenmodel._resample_F) and :295-329 (probit).  For node n the
// reference forms the posterior precision and mean term
//     Lambda_n = I / sigma_F + sum_{nn != n} s_{n,nn} g_nn g_nn^T,      h_n = sum_{nn != n} v_{n,nn} g_nn,
// g_nn = F[nn] * lambda (current values), s = omega_{n,nn} + omega_{nn,n} (probit: 2), v = kappa - omega mu_0 summed
// over both directions (probit: Z - mu_0), and draws F[n] ~ N(Lambda_n^{-1} h_n, Lambda_n^{-1}) with
// np.random.multivariate_normal (:908) - whose stream no parallel sampler can reproduce.
//
// The build draws from the same law BY PERTURBATION (Papandreou & Yuille 2010, "Gaussian sampling by local
// perturbations"): because Lambda_n is a sum of rank-one terms, a vector xi ~ N(0, Lambda_n) needs no factorisation,
//     xi_n = z_n / sqrt(sigma_F) + sum_{nn != n} sqrt(s_{n,nn}) eps_{n,nn} g_nn,        z, eps iid N(0, 1),
// and  x = Lambda_n^{-1} (h_n + xi_n)  has mean Lambda_n^{-1} h_n and covariance Lambda_n^{-1} Lambda_n Lambda_n^{-1}
// = Lambda_n^{-1}: exactly the reference's conditional.  The noise rides in the mean coefficient of every pair,
//     b_n = z_n / sqrt(sigma_F) + sum_{nn != n} (v_{n,nn} + sqrt(s_{n,nn}) eps_{n,nn}) g_nn,    x = Lambda_n^{-1} b_n,
// so the sequential scan needs ONE linear solve per node and no square root of a matrix - which is what lets the HIP
// engine keep Lambda_n^{-1} up to date by rank-one (Sherman-Morrison) steps instead of factorising per node
// (csrc/gg_eigen.hip).  Costs N^2 extra normals per sweep, drawn in parallel.
//
// eps_{n,nn} is addressed by (seed, GG_SITE_EIG_FN, sweep, n, nn): one Philox block per pair of neighbours
// (2 pp, 2 pp + 1), each normal by inversion (gg_ndtri, AS 241) of a uniform strictly inside (0, 1).
#pragma once
#include "gg_probit.h"

// Uniform strictly inside (0, 1) from two 32-bit words: (k + 1/2) 2^-52 with a 52-bit k - symmetric about 1/2, every
// value exactly representable (a 53-bit k plus a half would round up to 1.0 at the top).
GG_HD double gg_u52_open(uint32_t a, uint32_t b) {
  const uint64_t k = (((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) >> 1;
  return ((double)k + 0.5) * (1.0 / 4503599627370496.0);
}

GG_HD void gg_eig_noise_pair(uint64_t seed, uint32_t sweep, uint32_t N, uint32_t n, uint32_t pp, double* e0, double* e1) {
  const uint32_t Nh = (N + 1u) >> 1;
  const uint64_t idx = (uint64_t)n * Nh + pp;      // (passes 2^32 at N = 92 682: the high word takes the counter word that was 0)
  const gg_u32x4 r = gg_philox4x32_10((uint32_t)(idx >> 32), (uint32_t)idx, sweep, (uint32_t)GG_SITE_EIG_FN, (uint32_t)seed, (uint32_t)(seed >> 32));
  *e0 = gg_ndtri(gg_u52_open(r.v[0], r.v[1]));
  *e1 = gg_ndtri(gg_u52_open(r.v[2], r.v[3]));
}

GG_HD double gg_eig_noise_at(uint64_t seed, uint32_t sweep, uint32_t N, uint32_t n, uint32_t nn) {
  double e0, e1;
  gg_eig_noise_pair(seed, sweep, N, n, nn >> 1, &e0, &e1);
  return (nn & 1u) ? e1 : e0;
}

// Mean coefficient of neighbour nn in node n's conditional, with the perturbation noise:
// (a - mu_0 s) + sqrt(s) eps, a = the pair's summed kappa (logistic) / summed Z (probit).
GG_HD double gg_eig_coef(double s, double a, double mu0, double eps) {
  return gg_fma(gg_sqrt(s), eps, gg_fma(-mu0, s, a));
}

// x = prec^{-1} b for a symmetric positive-definite prec (row-major D x D); returns 0 on success.
GG_HD int gg_solve_spd(const double* prec, const double* b, int D, double* x) {
  double L[GG_MAXD * GG_MAXD], y[GG_MAXD];
  for (int i = 0; i < D * D; ++i) L[i] = prec[i];
  if (gg_chol_d(L, D)) return 1;
  for (int i = 0; i < D; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= L[i * D + k] * y[k];
    y[i] = v / L[i * D + i];
  }
  for (int i = D - 1; i >= 0; --i) {
    double v = y[i];
    for (int k = i + 1; k < D; ++k) v -= L[k * D + i] * x[k];
    x[i] = v / L[i * D + i];
  }
  return 0;
}
