// gg_ldw.h — conjugate draws of the latent-distance Gaussian weight model (graphistician/weights.py:613-903, B = 1),
// shared by the HIP engine and the CPU oracle:  W_ij ~ N(-|l_i - l_j|^2 + b, sigma) on off-diagonal edges,
// l_i ~ N(0, eta I), b ~ N(0, 1), eta ~ IG(1, 1), sigma ~ IW(lmbda_0, nu_0), self edges ~ NIW Gaussian.
#pragma once
#include "gg_draws.h"
#include "gg_shcov.h"

// b | X, sigma with X = W + D over the n off-diagonal edges, sx = sum X (weights.py:833-856)
GG_HD double gg_ldw_b_draw(gg_stream* s, double n, double sx, double sigma) {
  const double sigma_post = 1.0 / (1.0 + n / sigma);
  const double mu_post = sigma_post * (sx / sigma);
  return mu_post + gg_sqrt(sigma_post) * gg_normal(s);
}

// sigma | residuals: IW(lmbda_0 + sum r^2, nu_0 + n) (weights.py:859-863 -> pybasicbayes GaussianFixedMean.resample)
GG_HD double gg_ldw_sigma_draw(gg_stream* s, double lmbda_0, double nu_0, double n, double ss) {
  double S = lmbda_0 + ss, out;
  gg_iw_draw(s, &S, nu_0 + n, 1, &out);
  return out;
}

// eta | L: IG(1 + n_entries / 2, 1 + sum L^2 / 2), drawn as scale / Gamma(shape) (weights.py:870-886)
GG_HD double gg_ldw_eta_draw(gg_stream* s, double n_entries, double sq) {
  return (1.0 + 0.5 * sq) / gg_gamma(s, 1.0 + 0.5 * n_entries);
}
