// gg_pg.h — Polya-gamma PG(1, z) sampler, host + device.
//
// The reference draws the N^2 auxiliary variables of the logistic eigenmodel with
// pypolyagamma's `pgdrawvpar(ppgs, ones(N^2), Mu.ravel(), out)` (eigenmodel.py:
// 867-871); pypolyagamma is a third-party C++/Cython package absent from the
// reference tree, so the sampler is restated from its published
// algorithm: Polson, Scott & Windle (2013), "Bayesian inference for logistic
// models using Polya-gamma latent variables", §4 / Devroye's alternating-series
// method for J*(1, z/2), truncation point t = 0.64:
//   X ~ J*(1, z/2) by rejection from a mixture of a truncated inverse-Gaussian
//   (left of t) and a truncated exponential (right of t), accepted with the
//   alternating series S_n = sum (-1)^i a_i(X);  PG(1, z) = X / 4.
// Round 4: the same sampler written as a sequence of independent, bounded ATTEMPTS (gg_pg1_attempt below) instead of three
// nested rejection loops - the form a 64-wide wavefront can run with most of its lanes active; the law is the same.
// Only anchor inside the reference: E[omega] = tanh(z/2)/(2z) (eigenmodel.py:1062-1063);
// tests/ add the variance and a KS test against a series representation.
//
// exp/log are the reproducible gg_exp/gg_log; the normal cdf uses the platform
// erfc (libm on the host, ocml on the device: <= 1 ulp apart), so host and device
// agree to rounding, not bit for bit — eigenmodel parity is a float tolerance
// (north_star), unlike the block labels.
#pragma once
#include <math.h>
#include "gg_draws.h"

#define GG_PG_TRUNC 0.64
#define GG_PI 3.141592653589793238462643383279502884197

// exp / log of the PG(1, z) sampler: the platform's (libm on the host, ocml on the device, <= 1 ulp apart).  The
// eigenmodels' parity is a float tolerance, an accept / reject comparison flips only when the two sides agree to
// ~1e-16, and the branch-free ocml routines are markedly faster on the device than the bit-reproducible gg_exp / gg_log
// (which the block models need and keep).
#define GG_PG_EXP(x) exp(x)
#define GG_PG_LOG(x) log(x)

GG_HD double gg_norm_cdf(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

GG_HD double gg_expon(gg_stream* s) { return -GG_PG_LOG(gg_stream_uniform_pos(s)); }

// a_n(x): coefficients of the alternating series (Devroye 2009; PSW 2013 eq. 12-13)
GG_HD double gg_pg_a(int n, double x) {
  const double K = ((double)n + 0.5) * GG_PI;
  if (x > GG_PG_TRUNC) return K * GG_PG_EXP(-0.5 * K * K * x);
  if (x > 0.0) {
    double expnt = -1.5 * (GG_PG_LOG(0.5 * GG_PI) + GG_PG_LOG(x)) + GG_PG_LOG(K) - 2.0 * ((double)n + 0.5) * ((double)n + 0.5) / x;
    return GG_PG_EXP(expnt);
  }
  return 0.0;
}
// log((n + 1/2) pi) of the first terms as constants (round 4: nearly every series ends at n = 1, so the left branch paid two
// logarithms per proposal for two numbers known in advance; the constants are the correctly rounded values, libm / ocml are
// within an ulp of them - the same float-tolerance statement as for exp / log above).
GG_HD double gg_pg_logK(int n, double K) {
  return n == 0 ? 0.4515827052894548 : n == 1 ? 1.5501949939575645 : n == 2 ? 2.061020617723555 : n == 3 ? 2.397492854344768
                                                                                                         : GG_PG_LOG(K);
}
// probability mass of the truncated-exponential component
GG_HD double gg_pg_mass_texpon(double Z) {
  const double t = GG_PG_TRUNC;
  const double fz = 0.125 * GG_PI * GG_PI + 0.5 * Z * Z;
  const double b = gg_sqrt(1.0 / t) * (t * Z - 1.0);
  const double a = -gg_sqrt(1.0 / t) * (t * Z + 1.0);
  const double x0 = GG_PG_LOG(fz) + fz * t;
  const double xb = x0 - Z + GG_PG_LOG(gg_norm_cdf(b));
  const double xa = x0 + Z + GG_PG_LOG(gg_norm_cdf(a));
  const double qdivp = 4.0 / GG_PI * (GG_PG_EXP(xb) + GG_PG_EXP(xa));
  return 1.0 / (1.0 + qdivp);
}

// One Philox block of a stream as two uniforms (the stream's pending half, if any, is dropped: the attempts below consume
// whole blocks, so every lane of a wavefront calls the generator the same number of times per attempt).
GG_HD void gg_stream_pair(gg_stream* s, double* u, double* v) {
  gg_u32x4 r = gg_philox4x32_10(s->c0, s->c1, s->c2, s->c3, s->k0, s->k1);
  s->c0 += 1u;
  s->have = 0;
  *u = gg_u53(r.v[0], r.v[1]);
  *v = gg_u53(r.v[2], r.v[3]);
}

// a_n(x) with ONE exponential for both sides of the truncation point (the two sides differ in the argument and the factor in
// front only): the same values as gg_pg_a_pre.
GG_HD double gg_pg_a_sel(int n, double x, double lx) {
  const double K = ((double)n + 0.5) * GG_PI, h = (double)n + 0.5;
  const bool right = x > GG_PG_TRUNC;
  const double arg = right ? -0.5 * K * K * x : -1.5 * (0.4515827052894548 + lx) + gg_pg_logK(n, K) - 2.0 * h * h / x;
  const double v = (right ? K : 1.0) * GG_PG_EXP(arg);
  return x > 0.0 ? v : 0.0;
}

// ONE ATTEMPT of the sampler (round 4, second part).  The textbook form nests three rejection loops (the series around the
// truncated inverse Gaussian's acceptance around its proposal's); on a 64-wide wavefront every level waits for its slowest lane,
// and the counters showed a quarter of the lanes active over the draws' kernel.  The same sampler as a sequence of independent
// attempts, each a bounded, branch-poor piece of work:
//   choose the component (unless a truncated inverse-Gaussian trial is pending from the last attempt: *tig);
//   truncated exponential: X = t + E / fz - a proposal is at hand;
//   truncated inverse Gaussian, Z < 1/t: ONE trial of Devroye's scheme - E1, E2 exponentials, fail unless E1^2 <= 2 E2 / t,
//     X = t / (1 + t E1)^2, fail unless U <= exp(-Z^2 X / 2) (two acceptance tests in a row accept with the product of their
//     probabilities: the law of the nested loops) - Z >= 1/t: one trial of the transformed-normal scheme (one polar candidate
//     for the normal), fail unless X <= t;
//     a failed trial leaves *tig set: the next attempt tries this component again (choosing anew would favour the other one);
//   with a proposal at hand the series decides: accept (omega = X / 4) or reject - either way the component is chosen anew.
// Variates per attempt: block 1 = (component, E1 | root flip), block 2 = (E2, U | polar candidate), block 3 = (series' Y,
// unused) - whole Philox blocks.
// A caller that carries several draws per lane moves a lane on to its next draw as soon as one is accepted (k_eig_omega);
// gg_pg1_pre is the plain loop over attempts, and the oracle's form.
GG_HD bool gg_pg1_attempt(gg_stream* s, double Z, double ptex, int* tig, double* omega) {
  const double t = GG_PG_TRUNC;
  const double fz = 0.125 * GG_PI * GG_PI + 0.5 * Z * Z;
  double u0, u1;
  gg_stream_pair(s, &u0, &u1);
  const double E1 = -GG_PG_LOG(u1 + (0.5 / 9007199254740992.0));
  double X;
  if (!*tig && u0 < ptex) {
    X = t + E1 / fz;
  } else {
    // one trial of the truncated inverse Gaussian; both schemes take block 2 and ONE logarithm (of their own argument)
    *tig = 1;
    const bool small = 1.0 / t > Z;
    double u2, u3;
    gg_stream_pair(s, &u2, &u3);
    const double pu = 2.0 * u2 - 1.0, pv = 2.0 * u3 - 1.0, r = pu * pu + pv * pv;     // polar method's candidate (Z >= 1/t)
    const double L = GG_PG_LOG(small ? u2 + (0.5 / 9007199254740992.0) : r);
    if (small) {
      // Devroye's scheme: E1, E2 exponentials, E1^2 <= 2 E2 / t, then U <= exp(-Z^2 X / 2)
      const double E2 = -L;
      if (E1 * E1 > 2.0 * E2 / t) return false;
      X = 1.0 + E1 * t;
      X = t / (X * X);
      if (u3 > GG_PG_EXP(-0.5 * Z * Z * X)) return false;
    } else {
      // transformed normal (Michael, Schucany & Haas): Y = N(0,1)^2 from one polar trial - a candidate outside the unit disc
      // fails the attempt like any other rejection -, the root flipped by block 1's second uniform (E1 is not used here)
      if (!(r < 1.0 && r > 0.0)) return false;
      const double mu = 1.0 / Z;
      const double Y = pu * pu * (-2.0 * L / r);
      const double half_mu = 0.5 * mu;
      const double mu_Y = mu * Y;
      X = mu + half_mu * mu_Y - half_mu * gg_sqrt(4.0 * mu_Y + mu_Y * mu_Y);
      if (u1 > mu / (mu + X)) X = mu * mu / X;
      if (X > t) return false;
    }
  }
  *tig = 0;
  const double lx = GG_PG_LOG(X > GG_PG_TRUNC ? 1.0 : X);
  double S = gg_pg_a_sel(0, X, lx);
  double uy, un;
  gg_stream_pair(s, &uy, &un);
  const double Y = uy * S;
  S -= gg_pg_a_sel(1, X, lx);
  if (Y <= S) { *omega = 0.25 * X; return true; }
  for (int n = 2;; ++n) {
    if (n & 1) {
      S -= gg_pg_a_sel(n, X, lx);
      if (Y <= S) { *omega = 0.25 * X; return true; }
    } else {
      S += gg_pg_a_sel(n, X, lx);
      if (Y > S) return false;
    }
    if (n > 200) { *omega = 0.25 * X; return true; }   // numerically converged series
  }
}

// omega ~ PG(1, z) given Z = |z| / 2 and the mass of the truncated-exponential component (shared by the two
// draws omega_ij, omega_ji of a pair: same psi)
GG_HD double gg_pg1_pre(gg_stream* s, double Z, double ptex) {
  int tig = 0;
  double omega = 0.0;
  s->have = 0;
  while (!gg_pg1_attempt(s, Z, ptex, &tig, &omega)) {}
  return omega;
}

// omega ~ PG(b, z) for a real shape 0 < b < 1: the tempered logistic likelihood sigma(psi)^(b a) (1 - sigma(psi))^(b (1 - a))
// is augmented with PG(b, psi) and kappa = b (a - 1/2) (the reference leaves the eigenmodel's
// tempering a TODO, graph_models.py:478: BUILD-DEFINED).  No exact sampler is published for b < 1 (Devroye's series is the
// b = 1 case, sums of it the integer ones), so this is the infinite-gamma-sum representation (Polson, Scott & Windle 2013,
// eq. 5)  PG(b, z) = 1 / (2 pi^2) sum_k g_k / ((k - 1/2)^2 + z^2 / (4 pi^2)),  g_k ~ Gamma(b, 1),  truncated after
// GG_PG_TRUNC_TERMS terms with the remainder replaced by its mean (midpoint rule on 1 / (x^2 + c)): exact in the mean,
// short of the true variance by < b / (12 pi^4 K^3) ~ 2e-7 b at K = 16 (5e-6 of the variance b / 24 at z = 0: far below what any
// test of the law can see; rounds 2-3 summed 64 terms - 3e-9 b - at four times the gamma draws, 67 M x 64 per tempered sweep at
// N = 8192).  APPROXIMATE by construction and labelled so.
#define GG_PG_TRUNC_TERMS 16
// One trial of g ~ Gamma(b, 1), 0 < b < 1: Marsaglia & Tsang's scheme for the shape b + 1 with ONE polar candidate for its
// normal, times U^(1/b) - in whole Philox blocks like gg_pg1_attempt (block 1 = the polar candidate, block 2 = the test's
// uniform and the boost's), a rejection anywhere fails the trial (independent trials: the law of the nested loops of
// gg_gamma).  The scheme's squeeze is left out: it only spares a logarithm that a wavefront computes anyway.
GG_HD bool gg_gamma_frac_trial(gg_stream* s, double b, double* g) {
  const double d = b + 1.0 - 1.0 / 3.0, c = 1.0 / gg_sqrt(9.0 * d);
  double p, q;
  gg_stream_pair(s, &p, &q);
  const double pu = 2.0 * p - 1.0, pv = 2.0 * q - 1.0, r = pu * pu + pv * pv;
  if (!(r < 1.0 && r > 0.0)) return false;
  const double x = pu * gg_sqrt(-2.0 * GG_PG_LOG(r) / r);
  double v = 1.0 + c * x;
  if (v <= 0.0) return false;
  v = v * v * v;
  double ut, ub;
  gg_stream_pair(s, &ut, &ub);
  ut += 0.5 / 9007199254740992.0;
  ub += 0.5 / 9007199254740992.0;
  if (!(GG_PG_LOG(ut) < 0.5 * x * x + d * (1.0 - v + GG_PG_LOG(v)))) return false;
  *g = GG_PG_EXP(GG_PG_LOG(ub) / b) * d * v;
  return true;
}
// One attempt of omega ~ PG(b, z): a trial of the term *k (1 .. GG_PG_TRUNC_TERMS; start: *k = 1, *acc = 0) of the gamma sum;
// true with the draw once the last term is in.  c = z^2 / (4 pi^2).
GG_HD bool gg_pg_frac_attempt(gg_stream* s, double b, double c, int* k, double* acc, double* omega) {
  double g;
  if (!gg_gamma_frac_trial(s, b, &g)) return false;
  const double h = (double)*k - 0.5;
  *acc += g / (h * h + c);
  if (++*k <= GG_PG_TRUNC_TERMS) return false;
  const double K = (double)GG_PG_TRUNC_TERMS;
  const double sc = gg_sqrt(c);
  const double tail = c > 1e-12 ? atan2(sc, K) / sc : 1.0 / K;     // int_K^inf dx / (x^2 + c)
  *omega = (*acc + b * tail) / (2.0 * GG_PI * GG_PI);
  return true;
}
GG_HD double gg_pg_frac(gg_stream* s, double b, double z) {
  const double c = z * z / (4.0 * GG_PI * GG_PI);
  int k = 1;
  double acc = 0.0, omega = 0.0;
  while (!gg_pg_frac_attempt(s, b, c, &k, &acc, &omega)) {}
  return omega;
}

// omega ~ PG(1, z)
GG_HD double gg_pg1(gg_stream* s, double z) {
  const double Z = 0.5 * gg_fabs(z);
  return gg_pg1_pre(s, Z, gg_pg_mass_texpon(Z));
}

// Stream index of omega_ij: one stream per ordered pair.
GG_HD double gg_pg1_at(uint64_t seed, uint32_t sweep, uint32_t N, uint32_t i, uint32_t j, double psi) {
  gg_stream s;
  gg_stream_init(&s, seed, GG_SITE_EIG_OMEGA, sweep, i * N + j);
  return gg_pg1(&s, psi);
}

// One standard normal per (site, sweep, index): pre-drawable for a whole sweep.
GG_HD double gg_normal_at(uint64_t seed, uint32_t site, uint32_t sweep, uint32_t index) {
  gg_stream s;
  gg_stream_init(&s, seed, site, sweep, index);
  return gg_normal(&s);
}

// ---- dense D x D helpers for the Gaussian conditionals (D <= GG_MAXD) -------
#define GG_MAXD 16

// In-place lower Cholesky of a symmetric positive-definite matrix (row-major); returns 0 on success.
GG_HD int gg_chol_d(double* Aq, int D) {
  for (int j = 0; j < D; ++j) {
    double d = Aq[j * D + j];
    for (int k = 0; k < j; ++k) d -= Aq[j * D + k] * Aq[j * D + k];
    if (!(d > 0.0)) return 1;
    const double dj = gg_sqrt(d);
    Aq[j * D + j] = dj;
    for (int i = j + 1; i < D; ++i) {
      double v = Aq[i * D + j];
      for (int k = 0; k < j; ++k) v -= Aq[i * D + k] * Aq[j * D + k];
      Aq[i * D + j] = v / dj;
    }
    for (int i = 0; i < j; ++i) Aq[i * D + j] = 0.0;
  }
  return 0;
}

// x ~ N(prec^{-1} h, prec^{-1}) given standard normals z: with prec = L L^T,
// mean = L^{-T} L^{-1} h and x = mean + L^{-T} z = L^{-T} (L^{-1} h + z).
GG_HD int gg_gauss_from_precision(const double* prec, const double* h, const double* z, int D, double* x) {
  double L[GG_MAXD * GG_MAXD], y[GG_MAXD];
  for (int i = 0; i < D * D; ++i) L[i] = prec[i];
  if (gg_chol_d(L, D)) return 1;
  for (int i = 0; i < D; ++i) {          // y = L^{-1} h + z
    double v = h[i];
    for (int k = 0; k < i; ++k) v -= L[i * D + k] * (y[k] - z[k]);
    y[i] = v / L[i * D + i] + z[i];
  }
  for (int i = D - 1; i >= 0; --i) {     // x = L^{-T} y
    double v = y[i];
    for (int k = i + 1; k < D; ++k) v -= L[k * D + i] * x[k];
    x[i] = v / L[i * D + i];
  }
  return 0;
}
