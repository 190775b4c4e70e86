// gg_lp.h — the arithmetic *order* of the per-node block conditional, shared by
// the HIP engine and the CPU oracle so that both produce the same 64 bits for
// every log-probability and hence the same label from the same uniform.
//
// Reference semantics:
//   adjacency.py:661-695   lp[k] = log m[k] + sum_{n2 != n1} [ A[n1,n2] log p[k,c2]
//                          + (1-A[n1,n2]) log(1-p[k,c2]) + A[n2,n1] log p[c2,k]
//                          + (1-A[n2,n1]) log(1-p[c2,k]) ]
//   weights.py:451-488     lp[k] = log m[k] + sum_{c2} [ sum_{out edges to c2} log N(W; mu[k,c2], Sig[k,c2])
//                          + sum_{in edges from c2} log N(W; mu[c2,k], Sig[c2,k]) ]
// Both are linear in per-(node, block) sufficient statistics: with
//   so[j][t], si[j][t]  statistics of the node's out / in edges to block j
//   T[a][b][t]          per-block-pair coefficient table
// lp[k] = log m[k] + sum_t sum_j T[k][j][t] so[j][t] + sum_t sum_j T[j][k][t] si[j][t].
//   adjacency: t in {ones, zeros},          T = {log p, log(1-p)}
//   Gaussian : t in {n, s1_r, s2_rs (r<=s)}, T = {g, h_r, Q_rs} with
//              log N-sum = n g + h.s1 + sum_{r<=s} Q_rs s2_rs,
//              g = -1/2 mu'P mu - B/2 log 2pi - 1/2 logdet Sigma, h = P mu,
//              Q_rr = -1/2 P_rr, Q_rs = -P_rs  (P = Sigma^{-1}).
// The statistics are exact integers (edge counts; fixed-point weight sums), so
// the only rounding is in the fma chains below, whose order is fixed here:
// one chain per (direction, t), j ascending, then the chains are added to
// log m[k] in the order out-t0, out-t1, ..., in-t0, in-t1, ...
#pragma once
#include "gg_math.h"
#include "gg_draws.h"

#define GG_LOG_2PI 1.8378770664093453
#define GG_MAXNT (2 + GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)   /* + 1: the joint model's partner statistic */

GG_HD int gg_nt_gauss(int B) { return 1 + B + B * (B + 1) / 2; }

// T: [K][K][NT] row-major; so, si: [K][NT]; self: [NT] or null.
// beta is the AIS inverse temperature (graph_models.py:50-64,278-293: the likelihood terms are
// weighted by beta, the prior log m is not); fma(1.0, acc, lp) == lp + acc bit for bit, so
// beta = 1 is the plain conditional.
GG_HD double gg_block_lp(int k, int K, int NT, const double* logm, const double* T,
                         const double* so, const double* si, const double* self, double beta = 1.0) {
  double lp = logm[k];
  for (int t = 0; t < NT; ++t) {
    double acc = 0.0;
    for (int j = 0; j < K; ++j) acc = gg_fma(T[(k * K + j) * NT + t], so[j * NT + t], acc);
    lp = gg_fma(beta, acc, lp);
  }
  for (int t = 0; t < NT; ++t) {
    double acc = 0.0;
    for (int j = 0; j < K; ++j) acc = gg_fma(T[(j * K + k) * NT + t], si[j * NT + t], acc);
    lp = gg_fma(beta, acc, lp);
  }
  if (self) {
    double acc = 0.0;
    for (int t = 0; t < NT; ++t) acc = gg_fma(T[(k * K + k) * NT + t], self[t], acc);
    lp = gg_fma(beta, acc, lp);
  }
  return lp;
}

// Coefficient table entry of one Gaussian (mu, Sigma): out[NT] = {g, h, Q}.
GG_HD int gg_gauss_table(const double* mu, const double* sigma, int B, double* out) {
  double P[GG_MAXB * GG_MAXB];
  double logdet;
  if (gg_spd_inv_logdet(sigma, B, P, &logdet)) return 1;
  double h[GG_MAXB];
  double q = 0.0;
  for (int i = 0; i < B; ++i) {
    double v = 0.0;
    for (int j = 0; j < B; ++j) v += P[i * B + j] * mu[j];
    h[i] = v;
    q += mu[i] * v;
  }
  out[0] = -0.5 * q - 0.5 * (double)B * GG_LOG_2PI - 0.5 * logdet;
  for (int i = 0; i < B; ++i) out[1 + i] = h[i];
  int t = 1 + B;
  for (int r = 0; r < B; ++r)
    for (int s = r; s < B; ++s) out[t++] = (r == s) ? -0.5 * P[r * B + r] : -P[r * B + s];
  return 0;
}

// ---- reference-precision weights (opt-in "wide" storage, B = 1): two 24-bit limbs per weight -----------------------------
// The reference computes on float64 W (abstractions.py:257-274, weights.py:316-331).  The default storage keeps 24 bits
// absolute under ONE scale (q = rint(w 2^s)); the wide mode keeps 48: w = (q0 + q1 2^-24) 2^-s with |q0|, |q1| <= 2^23, stored
// as the two components of a B = 2 integer weight.  Every exact integer statistic the engine forms for B = 2 -
// n, sum q0, sum q1, sum q0^2, sum q0 q1, sum q1^2 - is then exactly what the one-dimensional Gaussian needs:
//     sum w   = 2^-s  (sum q0 + 2^-24 sum q1),
//     sum w^2 = 2^-2s (sum q0^2 + 2^-23 sum q0 q1 + 2^-48 sum q1^2),
// so the bulk passes, the scans and their integer corrections run UNCHANGED on the (q0, q1) data (the B = 2 instantiations),
// and only two places know about the limbs: the parameter step collapses the pair tables to (n, sum w, sum w^2) before its
// one-dimensional NIW draw, and the coefficient table of a pair spreads the one-dimensional (g, h, Q) over the five
// statistics - multiplications by powers of two, exact.
#define GG_WIDE_LIMB 24
#define GG_2M24 (1.0 / 16777216.0)                          /* 2^-24, 2^-23, 2^-48: exact */
#define GG_2M23 (1.0 / 8388608.0)
#define GG_2M48 ((1.0 / 16777216.0) * (1.0 / 16777216.0))
// s1[2], s2[2 x 2] (statistics of the limbs, real units of the FIRST limb) -> sum w, sum w^2
GG_HD void gg_wide_collapse(const double* s1, const double* s2, double* w1, double* w2) {
  *w1 = gg_fma(s1[1], GG_2M24, s1[0]);
  *w2 = gg_fma(s2[3], GG_2M48, gg_fma(s2[1], GG_2M23, s2[0]));
}
// Coefficient table entry of the one-dimensional Gaussian (mu, sigma) against the limbs' statistics: out[6] = {g, h, h 2^-24,
// Q, 2 Q 2^-24, Q 2^-48} (the order of gg_gauss_table for B = 2: n, s1_0, s1_1, s2_00, s2_01, s2_11).
GG_HD int gg_gauss_table_wide(const double* mu, const double* sigma, double* out) {
  double e[3];
  if (gg_gauss_table(mu, sigma, 1, e)) return 1;
  out[0] = e[0];
  out[1] = e[1];
  out[2] = e[1] * GG_2M24;
  out[3] = e[2];
  out[4] = e[2] * GG_2M23;
  out[5] = e[2] * GG_2M48;
  return 0;
}

// Joint SBM (one label vector for adjacency and weights, stochastic_block_model.py:252-302): per ordered pair of nodes
// the conditional adds Bernoulli(p).log_probability(A) = A log p + (1 - A) log(1 - p) to A * log N(W) (:280-296).  Summed
// over a node's potential partners in block j: (g + log p - log(1 - p)) * #edges + h.s1 + Q.s2 + log(1 - p) * #partners,
// i.e. the Gaussian table entry with its first coefficient shifted, followed by one more coefficient for the statistic
// "potential partners in the block" (block size, minus one for the node's own block).  e: NTg Gaussian entries in, NTg + 1 out.
GG_HD void gg_joint_table(double* e, int NTg, double p) {
  const double l1 = gg_log(p), l0 = gg_log(1.0 - p);
  e[0] = (e[0] + l1) - l0;
  e[NTg] = l0;
}

// Categorical draw from unnormalised log-probabilities with one uniform:
// pybasicbayes sample_discrete_from_log (call sites adjacency.py:695,
// weights.py:488; source not vendored): subtract the max,
// exponentiate, inclusive cumulative sum, count entries below u * total.
// The cumulative sum is a Kogge-Stone scan over K' = next power of two
// (zero padded) - the order a wavefront scan produces.
GG_HD int gg_next_pow2(int K) { int p = 1; while (p < K) p <<= 1; return p; }

GG_HD int gg_categorical(const double* lp, int K, double u, double* scratch /* 2*K' */) {
  int Kp = gg_next_pow2(K);
  double mx = lp[0];
  for (int k = 1; k < K; ++k) if (lp[k] > mx) mx = lp[k];
  double* x = scratch;
  double* y = scratch + Kp;
  for (int k = 0; k < Kp; ++k) x[k] = (k < K) ? gg_exp(lp[k] - mx) : 0.0;
  for (int s = 1; s < Kp; s <<= 1) {
    for (int k = 0; k < Kp; ++k) y[k] = (k >= s) ? x[k] + x[k - s] : x[k];
    double* tmp = x; x = y; y = tmp;
  }
  double thr = u * x[K - 1];
  int idx = 0;
  for (int k = 0; k < K; ++k) idx += (thr > x[k]) ? 1 : 0;
  if (idx > K - 1) idx = K - 1;
  return idx;
}
