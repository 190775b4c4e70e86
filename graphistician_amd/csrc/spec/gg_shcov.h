// gg_shcov.h — conjugate draws of SBMGaussianWeightSharedCov (weights.py:501-552): block-pair means with one
// covariance shared by all pairs.  The reference delegates to pybasicbayes GaussianFixedCov / GaussianFixedMean
// (source not in the tree), restated here from their conjugate algebra:
//   GaussianFixedCov(mu_0, sigma_0 = I, sigma).resample(x_1..n):  Lambda_n = I + n Sigma^-1,
//       mu ~ N(Lambda_n^-1 (mu_0 + Sigma^-1 sum x), Lambda_n^-1)
//   GaussianFixedMean(mu = 0, nu_0, lmbda_0).resample(y_1..n):    Sigma ~ IW(lmbda_0 + sum y y^T, nu_0 + n)
// with y = W - Mu over the off-diagonal edges (weights.py:545-548).
#pragma once
#include "gg_draws.h"
#include "gg_pg.h"

// mean of one block pair given the shared covariance; statistics n, s1 = sum x.  Returns nonzero if Sigma is not SPD.
GG_HD int gg_fixed_cov_mean_draw(gg_stream* s, const double* mu_0, const double* sigma, int B, double n, const double* s1,
                                 double* mu_out) {
  double P[GG_MAXB * GG_MAXB], logdet, prec[GG_MAXB * GG_MAXB], h[GG_MAXB], z[GG_MAXB];
  if (gg_spd_inv_logdet(sigma, B, P, &logdet)) return 1;
  for (int i = 0; i < B; ++i) {
    double v = mu_0[i];
    for (int j = 0; j < B; ++j) {
      prec[i * B + j] = (i == j ? 1.0 : 0.0) + n * P[i * B + j];
      v += P[i * B + j] * s1[j];
    }
    h[i] = v;
  }
  for (int i = 0; i < B; ++i) z[i] = gg_normal(s);
  return gg_gauss_from_precision(prec, h, z, B, mu_out);
}

// sum (x - mu)(x - mu)^T from n, s1 = sum x, s2 = sum x x^T
GG_HD void gg_centered_scatter(double n, const double* s1, const double* s2, const double* mu, int B, double* C) {
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) C[i * B + j] = s2[i * B + j] - s1[i] * mu[j] - mu[i] * s1[j] + n * mu[i] * mu[j];
}

// Sigma ~ IW(S, nu) (the Bartlett draw of gg_niw_sample; its mean part is drawn and discarded)
GG_HD void gg_iw_draw(gg_stream* s, const double* S, double nu, int B, double* sigma_out) {
  double zero[GG_MAXB], dummy[GG_MAXB];
  for (int i = 0; i < B; ++i) zero[i] = 0.0;
  gg_niw_sample(s, zero, S, 1.0, nu, B, dummy, sigma_out);
}
