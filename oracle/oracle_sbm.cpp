// oracle_sbm.cpp — CPU restatement of graphistician's SBM / Beta-Bernoulli /
// NIW Gibbs hot path.  TEST INFRASTRUCTURE ONLY: nothing in the product
// (graphistician_amd/) may import, link or call this file; it exists so that
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check and
// time the HIP engine against an independent, obviously-sequential CPU program.
//
// PARITY STATUS: "parity unpinned" in the task's sense — the reference holds no
// golden vector, known-answer test or fixture for this path
// and cannot be imported here (Python 2 + four un-vendored dependencies:
// ordinary import errors, nothing refused).  Fidelity therefore rests on
//   (1) line-by-line citations below,
//   (2) agreement of the two forms restated here (literal loops vs. count form),
//   (3) independent closed forms (scipy) and a pure-Python transliteration of
//       the reference loops (oracle/literal.py) checked in tests/.
//
// Every function is a plain loop nest over the dense N x N data.  Statistics are
// exact integers (edge counts, fixed-point weight sums in __int128), so the only
// floating-point order that matters is the per-node conditional, which is taken
// from the shared arithmetic spec (graphistician_amd/csrc/spec/gg_lp.h) — the
// oracle shares the *arithmetic primitives* (Philox, exp/log, conjugate draws,
// fma-chain order) with the engine and restates the *algorithm* on its own.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "../graphistician_amd/csrc/spec/gg_philox.h"
#include "../graphistician_amd/csrc/spec/gg_math.h"
#include "../graphistician_amd/csrc/spec/gg_draws.h"
#include "../graphistician_amd/csrc/spec/gg_lp.h"
#include "../graphistician_amd/csrc/spec/gg_shcov.h"

#define ORC_NO_EDGE INT32_MIN
#define ORC_PART_PARAMS 1
#define ORC_PART_C 2
#define ORC_PART_M 4

typedef __int128 i128;

static int ceil_log2_int(long long n) { int b = 0; while ((1LL << b) < n) ++b; return b; }

extern "C" {

// ---------------------------------------------------------------------------
// Weight storage: 24-bit fixed point.  W enters the sampler only through
// sums over edges of w and w w^T (pybasicbayes Gaussian.resample /
// log_likelihood, called at weights.py:161-162,325,329,408,413,462,468), so the
// build stores q = rint(w * 2^shift) and accumulates exactly in integers.
// shift is chosen from the largest |w| on an edge so that |q| <= 2^qbits and a
// per-node sum of q^2 over N edges cannot overflow 63 bits.
// ---------------------------------------------------------------------------
int orc_qbits(int N) {
  int b = (62 - ceil_log2_int(N)) / 2;
  return b > 23 ? 23 : b;
}

// Wide weights (spec/gg_lp.h): a ONE-dimensional weight kept as two 24-bit limbs, w = (q0 + q1 2^-24) 2^-shift,
// in the layout of a B = 2 integer weight.  orc_set_wide(1): the Gaussian-weight functions below then take q with two
// components and B = 2 for every statistic, while the parameters (mu, sigma, the NIW priors) are those of one dimension - the
// statistics collapse to (n, sum w, sum w^2) for the draws, the coefficient table spreads (g, h, Q) over the limbs' statistics.
static int g_wide = 0;
void orc_set_wide(int on) { g_wide = on ? 1 : 0; }

int orc_quantise_wide(int N, const uint8_t* A, const double* W, int32_t* q, int* shift_out) {
  double wmax = 0.0;
  for (long long ij = 0; ij < (long long)N * N; ++ij) {
    if (!A[ij]) continue;
    const double a = fabs(W[ij]);
    if (!(a <= 1.7e308)) return -1;
    if (a > wmax) wmax = a;
  }
  if (wmax == 0.0) wmax = 1.0;
  int e;
  frexp(wmax, &e);
  const int shift = orc_qbits(N) - e;
  for (long long ij = 0; ij < (long long)N * N; ++ij) {
    if (!A[ij]) { q[ij * 2] = ORC_NO_EDGE; q[ij * 2 + 1] = 0; continue; }
    const double w = W[ij];
    const long long q0 = llrint(ldexp(w, shift));
    const double rem = w - ldexp((double)q0, -shift);            // exact
    q[ij * 2] = (int32_t)q0;
    q[ij * 2 + 1] = (int32_t)llrint(ldexp(rem, shift + GG_WIDE_LIMB));
  }
  *shift_out = shift;
  return 0;
}

int orc_quantise(int N, int B, const uint8_t* A, const double* W, int32_t* q, int* shift_out) {
  double wmax = 0.0;
  for (long long ij = 0; ij < (long long)N * N; ++ij) {
    if (!A[ij]) continue;
    for (int b = 0; b < B; ++b) {
      double a = fabs(W[ij * B + b]);
      if (!(a <= 1.7e308)) return -1;  // NaN / inf on an edge
      if (a > wmax) wmax = a;
    }
  }
  if (wmax == 0.0) wmax = 1.0;
  int e;
  frexp(wmax, &e);  // wmax = f * 2^e, f in [0.5, 1)
  int shift = orc_qbits(N) - e;
  for (long long ij = 0; ij < (long long)N * N; ++ij) {
    for (int b = 0; b < B; ++b) {
      if (!A[ij]) { q[ij * B + b] = (b == 0) ? ORC_NO_EDGE : 0; continue; }
      q[ij * B + b] = (int32_t)llrint(ldexp(W[ij * B + b], shift));
    }
  }
  *shift_out = shift;
  return 0;
}

// ---------------------------------------------------------------------------
// SBM adjacency  (adjacency.py:440-702)
// ---------------------------------------------------------------------------

static void resample_m(int N, int K, const int32_t* c, const double* pi, double* m, uint64_t seed,
                       uint32_t site, uint32_t sweep);

// adjacency.py:620-648 resample_p: for each (c1,c2) in row-major order, masked
// edge / non-edge counts (diagonal excluded when special_case_self_conns),
// p[c1,c2] ~ Beta(tau1 + n1, tau0 + n0); then p_self from the diagonal.
static void sbm_adj_resample_p(int N, int K, const uint8_t* A, const int32_t* c, double* p,
                               double* p_self, double tau1, double tau0, int special,
                               uint64_t seed, uint32_t sweep, double beta) {
  std::vector<long long> n1((size_t)K * K, 0), n0((size_t)K * K, 0);
  long long d1 = 0, d0 = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      int a = A[(size_t)i * N + j] ? 1 : 0;
      if (i == j) {
        d1 += a; d0 += 1 - a;
        if (special) continue;
      }
      n1[(size_t)c[i] * K + c[j]] += a;
      n0[(size_t)c[i] * K + c[j]] += 1 - a;
    }
  for (int c1 = 0; c1 < K; ++c1)
    for (int c2 = 0; c2 < K; ++c2) {
      gg_stream s;
      gg_stream_init(&s, seed, GG_SITE_ADJ_P, sweep, (uint32_t)(c1 * K + c2));
      // tempered posterior: counts weighted by beta (graph_models.py:366-367)
      p[c1 * K + c2] = gg_beta(&s, tau1 + beta * (double)n1[c1 * K + c2], tau0 + beta * (double)n0[c1 * K + c2]);
    }
  if (special) {
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_ADJ_PSELF, sweep, 0u);
    *p_self = gg_beta(&s, tau1 + beta * (double)d1, tau0 + beta * (double)d0);
  }
}

// Count form of the conditional of node n1 under the current labels
// (SURVEY.md §8a row a6): block histograms of row n1 and column n1, self excluded.
static void sbm_adj_node_lp(int N, int K, const uint8_t* A, const int32_t* c, const double* logm,
                            const double* T /* [K][K][2] = log p, log(1-p) */, int special, int n1,
                            double* lp, double beta = 1.0) {
  std::vector<double> so((size_t)K * 2, 0.0), si((size_t)K * 2, 0.0);
  for (int n2 = 0; n2 < N; ++n2) {
    if (n2 == n1) continue;
    int j = c[n2];
    if (A[(size_t)n1 * N + n2]) so[j * 2 + 0] += 1.0; else so[j * 2 + 1] += 1.0;
    if (A[(size_t)n2 * N + n1]) si[j * 2 + 0] += 1.0; else si[j * 2 + 1] += 1.0;
  }
  double self[2];
  const double* selfp = 0;
  if (!special) {  // adjacency.py:682-683
    double a = A[(size_t)n1 * N + n1] ? 1.0 : 0.0;
    self[0] = a; self[1] = 1.0 - a;
    selfp = self;
  }
  for (int k = 0; k < K; ++k) lp[k] = gg_block_lp(k, K, 2, logm, T, so.data(), si.data(), selfp, beta);
}

static void sbm_adj_tables(int K, const double* m, const double* p, double* logm, double* T) {
  for (int k = 0; k < K; ++k) logm[k] = gg_log(m[k]);
  for (int ab = 0; ab < K * K; ++ab) {
    T[ab * 2 + 0] = gg_log(p[ab]);
    T[ab * 2 + 1] = gg_log(1.0 - p[ab]);
  }
}

// adjacency.py:650-695 resample_c: sequential scan, labels updated in place.
static void sbm_adj_resample_c(int N, int K, const uint8_t* A, int32_t* c, const double* m,
                               const double* p, int special, uint64_t seed, uint32_t sweep, double beta) {
  if (K == 1) return;  // adjacency.py:657-658
  std::vector<double> logm(K), T((size_t)K * K * 2), lp(K), scratch(2 * gg_next_pow2(K));
  sbm_adj_tables(K, m, p, logm.data(), T.data());
  for (int n1 = 0; n1 < N; ++n1) {
    sbm_adj_node_lp(N, K, A, c, logm.data(), T.data(), special, n1, lp.data(), beta);
    double u = gg_uniform_at(seed, GG_SITE_ADJ_C, sweep, (uint32_t)n1);
    c[n1] = gg_categorical(lp.data(), K, u, scratch.data());
  }
}

// adjacency.py:697-702 resample_m.
static void resample_m(int N, int K, const int32_t* c, const double* pi, double* m, uint64_t seed,
                       uint32_t site, uint32_t sweep) {
  std::vector<double> alpha(K);
  for (int k = 0; k < K; ++k) alpha[k] = pi[k];
  for (int n = 0; n < N; ++n) alpha[c[n]] += 1.0;
  gg_dirichlet_at(seed, site, sweep, alpha.data(), K, m);
}

// adjacency.py:620-648 resample_p(As) for a LIST of graphs: `n_conns = sum([A[mask].sum() for A in As])`, the same for the
// non-connections and for the diagonal; As is [nG][N][N].
int orc_sbm_adj_resample_p_multi(int nG, int N, int K, const uint8_t* As, const int32_t* c, double* p, double* p_self,
                                 double tau1, double tau0, int special, uint64_t seed, uint32_t sweep, double beta) {
  std::vector<long long> n1((size_t)K * K, 0), n0((size_t)K * K, 0);
  long long d1 = 0, d0 = 0;
  for (int g = 0; g < nG; ++g) {
    const uint8_t* A = As + (size_t)g * N * N;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        int a = A[(size_t)i * N + j] ? 1 : 0;
        if (i == j) {
          d1 += a; d0 += 1 - a;
          if (special) continue;
        }
        n1[(size_t)c[i] * K + c[j]] += a;
        n0[(size_t)c[i] * K + c[j]] += 1 - a;
      }
  }
  for (int c1 = 0; c1 < K; ++c1)
    for (int c2 = 0; c2 < K; ++c2) {
      gg_stream s;
      gg_stream_init(&s, seed, GG_SITE_ADJ_P, sweep, (uint32_t)(c1 * K + c2));
      p[c1 * K + c2] = gg_beta(&s, tau1 + beta * (double)n1[c1 * K + c2], tau0 + beta * (double)n0[c1 * K + c2]);
    }
  if (special) {
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_ADJ_PSELF, sweep, 0u);
    *p_self = gg_beta(&s, tau1 + beta * (double)d1, tau0 + beta * (double)d0);
  }
  return 0;
}

// adjacency.py:615-618 resample = resample_p -> resample_c -> resample_m.
int orc_sbm_adj_sweep_beta(int N, int K, const uint8_t* A, int32_t* c, double* m, double* p,
                           double* p_self, const double* pi, double tau1, double tau0, int special,
                           uint64_t seed, uint32_t sweep, int parts, double beta) {
  if (parts & ORC_PART_PARAMS) sbm_adj_resample_p(N, K, A, c, p, p_self, tau1, tau0, special, seed, sweep, beta);
  if (parts & ORC_PART_C) sbm_adj_resample_c(N, K, A, c, m, p, special, seed, sweep, beta);
  if (parts & ORC_PART_M) resample_m(N, K, c, pi, m, seed, GG_SITE_ADJ_M, sweep);
  return 0;
}

int orc_sbm_adj_sweep(int N, int K, const uint8_t* A, int32_t* c, double* m, double* p,
                      double* p_self, const double* pi, double tau1, double tau0, int special,
                      uint64_t seed, uint32_t sweep, int parts) {
  if (parts & ORC_PART_PARAMS) sbm_adj_resample_p(N, K, A, c, p, p_self, tau1, tau0, special, seed, sweep, 1.0);
  if (parts & ORC_PART_C) sbm_adj_resample_c(N, K, A, c, m, p, special, seed, sweep, 1.0);
  if (parts & ORC_PART_M) resample_m(N, K, c, pi, m, seed, GG_SITE_ADJ_M, sweep);
  return 0;
}

// lp vector of one node in the count form (spec order) — for two-form tests.
int orc_sbm_adj_node_lp(int N, int K, const uint8_t* A, const int32_t* c, const double* m,
                        const double* p, int special, int n1, double* lp) {
  std::vector<double> logm(K), T((size_t)K * K * 2);
  sbm_adj_tables(K, m, p, logm.data(), T.data());
  sbm_adj_node_lp(N, K, A, c, logm.data(), T.data(), special, n1, lp);
  return 0;
}

// The literal loop nest of adjacency.py:661-693 (cn1 outer, n2 inner, four
// terms, libm log evaluated per term) — slow, obviously faithful.
int orc_sbm_adj_node_lp_literal(int N, int K, const uint8_t* A, const int32_t* c, const double* m,
                                const double* p, int special, int n1, double* lp) {
  for (int cn1 = 0; cn1 < K; ++cn1) {
    double acc = 0.0;
    acc += log(m[cn1]);
    for (int n2 = 0; n2 < N; ++n2) {
      int cn2 = c[n2];
      double a12 = A[(size_t)n1 * N + n2] ? 1.0 : 0.0;
      double a21 = A[(size_t)n2 * N + n1] ? 1.0 : 0.0;
      if (n2 == n1) {
        if (special) continue;
        acc += a12 * log(p[cn1 * K + cn1]) + (1 - a12) * log(1 - p[cn1 * K + cn1]);
      } else {
        acc += a12 * log(p[cn1 * K + cn2]) + (1 - a12) * log(1 - p[cn1 * K + cn2]);
        acc += a21 * log(p[cn2 * K + cn1]) + (1 - a21) * log(1 - p[cn2 * K + cn1]);
      }
    }
    lp[cn1] = acc;
  }
  return 0;
}

// Categorical draw exposed for tests (pybasicbayes sample_discrete_from_log semantics).
int orc_categorical(const double* lp, int K, double u) {
  std::vector<double> scratch(2 * gg_next_pow2(K));
  return gg_categorical(lp, K, u, scratch.data());
}

// abstractions.py:111-115 with adjacency.py:516-528: literal N x N sweep.
// P_ij = p[c_i,c_j]; diagonal p_self (special) ; 0 if !allow_self.
// Term convention: A log P is taken as 0 when A == 0 (the reference's
// unclipped 0 * log 0 = NaN quirk, SURVEY.md §7.3 item 6, is not reproduced).
double orc_sbm_adj_loglik(int N, int K, const uint8_t* A, const int32_t* c, const double* p,
                          double p_self, int allow_self, int special) {
  double ll = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double P = p[c[i] * K + c[j]];
      if (i == j) {
        if (special) P = p_self;
        if (!allow_self) P = 0.0;
      }
      double Pc = P < 1e-64 ? 1e-64 : (P > 1 - 1e-64 ? 1 - 1e-64 : P);  // safe_P, abstractions.py:92-94
      if (A[(size_t)i * N + j]) ll += log(P);
      else ll += log(1.0 - Pc);
    }
  return ll;
}

// ---------------------------------------------------------------------------
// Beta-Bernoulli adjacency (adjacency.py:110-188)
// ---------------------------------------------------------------------------
// adjacency.py:153-181.  NOTE the reference updates p_self with the OFF-diagonal
// priors tau1,tau0 (adjacency.py:175) — reproduced, not fixed.
int orc_bb_adj_sweep_beta(int N, const uint8_t* A, double* p, double* p_self, double tau1, double tau0,
                          int self_connection, uint64_t seed, uint32_t sweep, double beta);
int orc_bb_adj_sweep(int N, const uint8_t* A, double* p, double* p_self, double tau1, double tau0,
                     int self_connection, uint64_t seed, uint32_t sweep) {
  return orc_bb_adj_sweep_beta(N, A, p, p_self, tau1, tau0, self_connection, seed, sweep, 1.0);
}
// tempered form: graph_models.py:178-182 (a + beta nnz, b + beta (N^2 - nnz))
int orc_bb_adj_sweep_beta(int N, const uint8_t* A, double* p, double* p_self, double tau1, double tau0,
                          int self_connection, uint64_t seed, uint32_t sweep, double beta) {
  long long n1 = 0, n0 = 0, d1 = 0, d0 = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      int a = A[(size_t)i * N + j] ? 1 : 0;
      if (i == j) { d1 += a; d0 += 1 - a; } else { n1 += a; n0 += 1 - a; }
    }
  gg_stream s;
  gg_stream_init(&s, seed, GG_SITE_BB_P, sweep, 0u);
  if (self_connection) {
    *p = gg_beta(&s, tau1 + beta * (double)n1, tau0 + beta * (double)n0);
    gg_stream s2;
    gg_stream_init(&s2, seed, GG_SITE_BB_PSELF, sweep, 0u);
    *p_self = gg_beta(&s2, tau1 + beta * (double)d1, tau0 + beta * (double)d0);
  } else {
    *p = gg_beta(&s, tau1 + beta * (double)(n1 + d1), tau0 + beta * (double)(n0 + d0));
  }
  return 0;
}

// Generic Bernoulli log-likelihood with constant p off-diagonal, p_self on it
// (abstractions.py:111-115 with adjacency.py:128-133 / :65-83).
double orc_const_adj_loglik(int N, const uint8_t* A, double p, double p_self) {
  double ll = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      double P = (i == j) ? p_self : p;
      double Pc = P < 1e-64 ? 1e-64 : (P > 1 - 1e-64 ? 1 - 1e-64 : P);
      if (A[(size_t)i * N + j]) ll += log(P);
      else ll += log(1.0 - Pc);
    }
  return ll;
}

// ---------------------------------------------------------------------------
// Gaussian weights
// ---------------------------------------------------------------------------
struct WStat {  // exact statistics of a set of edges
  long long n;
  i128 s1[GG_MAXB];
  i128 s2[GG_MAXB * GG_MAXB];
};
static void wstat_zero(WStat* s) { memset(s, 0, sizeof(WStat)); }
static void wstat_add(WStat* s, const int32_t* q, int B) {
  s->n += 1;
  for (int r = 0; r < B; ++r) {
    s->s1[r] += q[r];
    for (int t = 0; t < B; ++t) s->s2[r * B + t] += (i128)q[r] * (i128)q[t];
  }
}
// exact integers -> real units: correctly rounded conversion, exact power-of-two scaling.
static void wstat_to_double(const WStat* s, int B, int shift, double* n, double* s1, double* s2) {
  *n = (double)s->n;
  for (int r = 0; r < B; ++r) s1[r] = ldexp((double)s->s1[r], -shift);
  for (int r = 0; r < B * B; ++r) s2[r] = ldexp((double)s->s2[r], -2 * shift);
}

static void unpack_niw(const double* packed, int B, gg_niw* h) {
  // packed = [mu_0 (B), sigma_0 (B*B), nu_0, kappa_0]
  memset(h, 0, sizeof(gg_niw));
  for (int i = 0; i < B; ++i) h->mu_0[i] = packed[i];
  for (int i = 0; i < B * B; ++i) h->sigma_0[i] = packed[B + i];
  h->nu_0 = packed[B + B * B];
  h->kappa_0 = packed[B + B * B + 1];
}

// weights.py:399-413 resample_mu_and_Sig: per (c1,c2) the edges of the block
// pair (A==1, off-diagonal when special) feed a NIW posterior draw; then the
// self Gaussian from the diagonal edges.
static void sbm_w_resample_mu_sig(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift,
                                  const int32_t* c, double* mu, double* sigma, double* mu_self,
                                  double* sigma_self, const gg_niw* h, const gg_niw* hs, int special,
                                  uint64_t seed, uint32_t sweep, double beta = 1.0) {
  std::vector<WStat> st((size_t)K * K);
  for (auto& s : st) wstat_zero(&s);
  WStat sd;
  wstat_zero(&sd);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (!A[(size_t)i * N + j]) continue;
      const int32_t* qq = q + ((size_t)i * N + j) * B;
      if (i == j) {
        wstat_add(&sd, qq, B);
        if (special) continue;
      }
      wstat_add(&st[(size_t)c[i] * K + c[j]], qq, B);
    }
  double n, s1[GG_MAXB], s2[GG_MAXB * GG_MAXB];
  const int Bd = g_wide ? 1 : B;      // dimension of the parameters (wide: the limbs' statistics collapse to one weight's)
  for (int c1 = 0; c1 < K; ++c1)
    for (int c2 = 0; c2 < K; ++c2) {
      wstat_to_double(&st[(size_t)c1 * K + c2], B, shift, &n, s1, s2);
      if (g_wide) { double w1, w2; gg_wide_collapse(s1, s2, &w1, &w2); s1[0] = w1; s2[0] = w2; }
      // tempered posterior (build-defined, mirrors graph_models.py:366-367): statistics weighted by beta
      n *= beta;
      for (int r = 0; r < Bd; ++r) s1[r] *= beta;
      for (int r = 0; r < Bd * Bd; ++r) s2[r] *= beta;
      gg_stream s;
      gg_stream_init(&s, seed, GG_SITE_W_NIW, sweep, (uint32_t)(c1 * K + c2));
      gg_niw_posterior_draw(&s, h, Bd, n, s1, s2, mu + (size_t)(c1 * K + c2) * Bd,
                            sigma + (size_t)(c1 * K + c2) * Bd * Bd);
    }
  if (special) {
    wstat_to_double(&sd, B, shift, &n, s1, s2);
    if (g_wide) { double w1, w2; gg_wide_collapse(s1, s2, &w1, &w2); s1[0] = w1; s2[0] = w2; }
    n *= beta;
    for (int r = 0; r < Bd; ++r) s1[r] *= beta;
    for (int r = 0; r < Bd * Bd; ++r) s2[r] *= beta;
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_W_NIW_SELF, sweep, 0u);
    gg_niw_posterior_draw(&s, hs, Bd, n, s1, s2, mu_self, sigma_self);
  }
}

static void pack_stat(const WStat* s, int B, int shift, double* out /* NT */) {
  double n, s1[GG_MAXB], s2[GG_MAXB * GG_MAXB];
  wstat_to_double(s, B, shift, &n, s1, s2);
  out[0] = n;
  for (int r = 0; r < B; ++r) out[1 + r] = s1[r];
  int t = 1 + B;
  for (int r = 0; r < B; ++r)
    for (int u = r; u < B; ++u) out[t++] = s2[r * B + u];
}

static int sbm_w_tables(int K, int B, const double* m, const double* mu, const double* sigma,
                        double* logm, double* T) {
  int NT = gg_nt_gauss(B);
  for (int k = 0; k < K; ++k) logm[k] = gg_log(m[k]);
  for (int ab = 0; ab < K * K; ++ab)
    if (g_wide ? gg_gauss_table_wide(mu + ab, sigma + ab, T + (size_t)ab * NT)
               : gg_gauss_table(mu + (size_t)ab * B, sigma + (size_t)ab * B * B, B, T + (size_t)ab * NT)) return 1;
  return 0;
}

// Sufficient-statistic form of weights.py:451-470 (_evaluate_lkhd) for node n1.
static void sbm_w_node_lp(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift,
                          const int32_t* c, const double* logm, const double* T, int special, int n1,
                          double* lp, double beta = 1.0) {
  int NT = gg_nt_gauss(B);
  std::vector<WStat> so(K), si(K);
  for (int k = 0; k < K; ++k) { wstat_zero(&so[k]); wstat_zero(&si[k]); }
  for (int n2 = 0; n2 < N; ++n2) {
    if (n2 == n1) continue;
    if (A[(size_t)n1 * N + n2]) wstat_add(&so[c[n2]], q + ((size_t)n1 * N + n2) * B, B);
    if (A[(size_t)n2 * N + n1]) wstat_add(&si[c[n2]], q + ((size_t)n2 * N + n1) * B, B);
  }
  std::vector<double> dso((size_t)K * NT), dsi((size_t)K * NT);
  for (int k = 0; k < K; ++k) {
    pack_stat(&so[k], B, shift, dso.data() + (size_t)k * NT);
    pack_stat(&si[k], B, shift, dsi.data() + (size_t)k * NT);
  }
  double self[GG_MAXNT];
  const double* selfp = 0;
  if (!special) {  // weights.py:456-459: out_mask keeps n1 itself -> gaussians[k][k]
    WStat ss;
    wstat_zero(&ss);
    if (A[(size_t)n1 * N + n1]) wstat_add(&ss, q + ((size_t)n1 * N + n1) * B, B);
    pack_stat(&ss, B, shift, self);
    selfp = self;
  }
  for (int k = 0; k < K; ++k) lp[k] = gg_block_lp(k, K, NT, logm, T, dso.data(), dsi.data(), selfp, beta);
}

// weights.py:415-488 resample_c.
static int sbm_w_resample_c(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift,
                            int32_t* c, const double* m, const double* mu, const double* sigma,
                            int special, uint64_t seed, uint32_t sweep, double beta = 1.0) {
  if (K == 1) return 0;  // weights.py:421-422
  int NT = gg_nt_gauss(B);
  std::vector<double> logm(K), T((size_t)K * K * NT), lp(K), scratch(2 * gg_next_pow2(K));
  if (sbm_w_tables(K, B, m, mu, sigma, logm.data(), T.data())) return -2;
  for (int n1 = 0; n1 < N; ++n1) {
    sbm_w_node_lp(N, K, B, A, q, shift, c, logm.data(), T.data(), special, n1, lp.data(), beta);
    double u = gg_uniform_at(seed, GG_SITE_W_C, sweep, (uint32_t)n1);
    c[n1] = gg_categorical(lp.data(), K, u, scratch.data());
  }
  return 0;
}

// weights.py:394-397 resample = resample_mu_and_Sig -> resample_c -> resample_m.
// niw / niw_self: packed [mu_0, sigma_0, nu_0, kappa_0].
int orc_sbm_w_sweep_beta(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, int32_t* c,
                         double* m, double* mu, double* sigma, double* mu_self, double* sigma_self,
                         const double* pi, const double* niw, const double* niw_self, int special,
                         uint64_t seed, uint32_t sweep, int parts, double beta) {
  if (B > GG_MAXB || (g_wide && B != 2)) return -1;
  gg_niw h, hs;
  unpack_niw(niw, g_wide ? 1 : B, &h);
  unpack_niw(niw_self, g_wide ? 1 : B, &hs);
  if (parts & ORC_PART_PARAMS)
    sbm_w_resample_mu_sig(N, K, B, A, q, shift, c, mu, sigma, mu_self, sigma_self, &h, &hs, special, seed, sweep, beta);
  if (parts & ORC_PART_C) {
    int rc = sbm_w_resample_c(N, K, B, A, q, shift, c, m, mu, sigma, special, seed, sweep, beta);
    if (rc) return rc;
  }
  if (parts & ORC_PART_M) resample_m(N, K, c, pi, m, seed, GG_SITE_W_M, sweep);
  return 0;
}

int orc_sbm_w_sweep(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, int32_t* c,
                    double* m, double* mu, double* sigma, double* mu_self, double* sigma_self,
                    const double* pi, const double* niw, const double* niw_self, int special,
                    uint64_t seed, uint32_t sweep, int parts) {
  return orc_sbm_w_sweep_beta(N, K, B, A, q, shift, c, m, mu, sigma, mu_self, sigma_self, pi, niw, niw_self, special,
                              seed, sweep, parts, 1.0);
}

int orc_sbm_w_node_lp(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift,
                      const int32_t* c, const double* m, const double* mu, const double* sigma,
                      int special, int n1, double* lp) {
  int NT = gg_nt_gauss(B);
  std::vector<double> logm(K), T((size_t)K * K * NT);
  if (sbm_w_tables(K, B, m, mu, sigma, logm.data(), T.data())) return -2;
  sbm_w_node_lp(N, K, B, A, q, shift, c, logm.data(), T.data(), special, n1, lp);
  return 0;
}

// log N(x; mu, Sigma) written out as abstractions.py:344-346 does:
// -1/2 |L^{-1}(x-mu)|^2 - B/2 log 2pi - sum log diag L, libm log.
static double mvn_logpdf(const double* x, const double* mu, const double* sigma, int B) {
  double L[GG_MAXB * GG_MAXB];
  gg_chol(sigma, B, L);
  double y[GG_MAXB];
  double quad = 0.0, ld = 0.0;
  for (int i = 0; i < B; ++i) {
    double v = x[i] - mu[i];
    for (int k = 0; k < i; ++k) v -= L[i * B + k] * y[k];
    y[i] = v / L[i * B + i];
    quad += y[i] * y[i];
    ld += log(L[i * B + i]);
  }
  return -0.5 * quad - 0.5 * B * log(2.0 * M_PI) - ld;
}

// The literal form (weights.py:427-449 _evaluate_lkhd_slow): per-edge Gaussian
// log-densities of the de-quantised weights, n2 loop, no sufficient statistics.
int orc_sbm_w_node_lp_literal(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift,
                              const int32_t* c, const double* m, const double* mu,
                              const double* sigma, int special, int n1, double* lp) {
  double x[GG_MAXB];
  for (int cn1 = 0; cn1 < K; ++cn1) {
    double ll = log(m[cn1]);
    for (int n2 = 0; n2 < N; ++n2) {
      int cn2 = c[n2];
      if (n2 == n1) {
        if (special) continue;
        if (A[(size_t)n1 * N + n1]) {
          for (int b = 0; b < B; ++b) x[b] = ldexp((double)q[((size_t)n1 * N + n1) * B + b], -shift);
          ll += mvn_logpdf(x, mu + (size_t)(cn1 * K + cn1) * B, sigma + (size_t)(cn1 * K + cn1) * B * B, B);
        }
      } else {
        if (A[(size_t)n1 * N + n2]) {
          for (int b = 0; b < B; ++b) x[b] = ldexp((double)q[((size_t)n1 * N + n2) * B + b], -shift);
          ll += mvn_logpdf(x, mu + (size_t)(cn1 * K + cn2) * B, sigma + (size_t)(cn1 * K + cn2) * B * B, B);
        }
        if (A[(size_t)n2 * N + n1]) {
          for (int b = 0; b < B; ++b) x[b] = ldexp((double)q[((size_t)n2 * N + n1) * B + b], -shift);
          ll += mvn_logpdf(x, mu + (size_t)(cn2 * K + cn1) * B, sigma + (size_t)(cn2 * K + cn1) * B * B, B);
        }
      }
    }
    lp[cn1] = ll;
  }
  return 0;
}

// weights.py:316-331 log_likelihood: literal per-edge sum.  With K == 1 and
// c == 0 this is also NIWGaussianWeightDistribution's likelihood
// (abstractions.py:257-274 with weights.py:97-115).
double orc_sbm_w_loglik(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift,
                        const int32_t* c, const double* mu, const double* sigma,
                        const double* mu_self, const double* sigma_self, int special) {
  double ll = 0.0;
  double x[GG_MAXB];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (!A[(size_t)i * N + j]) continue;
      if (g_wide) {      // the weight itself, put together from its limbs (exact in double), under the one-dimensional Gaussian
        const int32_t* qq = q + ((size_t)i * N + j) * 2;
        x[0] = ldexp((double)qq[0], -shift) + ldexp((double)qq[1], -shift - GG_WIDE_LIMB);
        if (i == j && special) ll += mvn_logpdf(x, mu_self, sigma_self, 1);
        else ll += mvn_logpdf(x, mu + (size_t)(c[i] * K + c[j]), sigma + (size_t)(c[i] * K + c[j]), 1);
        continue;
      }
      for (int b = 0; b < B; ++b) x[b] = ldexp((double)q[((size_t)i * N + j) * B + b], -shift);
      if (i == j && special) ll += mvn_logpdf(x, mu_self, sigma_self, B);
      else ll += mvn_logpdf(x, mu + (size_t)(c[i] * K + c[j]) * B, sigma + (size_t)(c[i] * K + c[j]) * B * B, B);
    }
  return ll;
}

// weights.py:154-162 NIWGaussianWeightDistribution.resample: one Gaussian for
// the off-diagonal edges, one for the self edges.
int orc_niw_w_sweep_beta(int N, int B, const uint8_t* A, const int32_t* q, int shift, double* mu,
                         double* sigma, double* mu_self, double* sigma_self, const double* niw,
                         const double* niw_self, uint64_t seed, uint32_t sweep, double beta) {
  if (B > GG_MAXB) return -1;
  gg_niw h, hs;
  unpack_niw(niw, B, &h);
  unpack_niw(niw_self, B, &hs);
  std::vector<int32_t> c(N, 0);
  sbm_w_resample_mu_sig(N, 1, B, A, q, shift, c.data(), mu, sigma, mu_self, sigma_self, &h, &hs, 1, seed, sweep, beta);
  return 0;
}
int orc_niw_w_sweep(int N, int B, const uint8_t* A, const int32_t* q, int shift, double* mu,
                    double* sigma, double* mu_self, double* sigma_self, const double* niw,
                    const double* niw_self, uint64_t seed, uint32_t sweep) {
  return orc_niw_w_sweep_beta(N, B, A, q, shift, mu, sigma, mu_self, sigma_self, niw, niw_self, seed, sweep, 1.0);
}

// ---------------------------------------------------------------------------
// SBMGaussianWeightSharedCov (weights.py:501-552): one covariance for all block pairs.
// resample_mu_and_Sig: for (c1, c2) row-major: mean | Sigma from the pair's edges (GaussianFixedCov, prior
// N(mu_0, I)); self Gaussian (NIW, as the parent); then Sigma ~ IW(lmbda_0 + sum over off-diagonal edges of
// (W - Mu)(W - Mu)^T, nu_0 + #edges) with the NEW means; every pair's sigma is set to it.
// prior: packed [mu_0 (B), lmbda_0 (B*B), nu_0, (unused)] like the NIW layout.  special_case_self_conns only.
// ---------------------------------------------------------------------------
static int sbm_wsh_resample_mu_sig(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, const int32_t* c,
                                   double* mu, double* sigma, double* mu_self, double* sigma_self, const gg_niw* h,
                                   const gg_niw* hs, int special, uint64_t seed, uint32_t sweep, double beta) {
  // st: the edges each pair's mean is drawn from (_get_mask, weights.py:308-314: the diagonal belongs to pair
  // (c_i, c_i) unless self connections are special-cased); so: the off-diagonal edges only, which is what the
  // covariance is centred on (:545-548)
  std::vector<WStat> st((size_t)K * K), so((size_t)K * K);
  for (auto& s : st) wstat_zero(&s);
  for (auto& s : so) wstat_zero(&s);
  WStat sd;
  wstat_zero(&sd);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (!A[(size_t)i * N + j]) continue;
      const int32_t* qq = q + ((size_t)i * N + j) * B;
      if (i == j) {
        if (special) wstat_add(&sd, qq, B);
        else wstat_add(&st[(size_t)c[i] * K + c[i]], qq, B);
        continue;
      }
      wstat_add(&st[(size_t)c[i] * K + c[j]], qq, B);
      wstat_add(&so[(size_t)c[i] * K + c[j]], qq, B);
    }
  double n, s1[GG_MAXB], s2[GG_MAXB * GG_MAXB], C[GG_MAXB * GG_MAXB], S[GG_MAXB * GG_MAXB];
  double ntot = 0.0;
  for (int i = 0; i < B * B; ++i) S[i] = h->sigma_0[i];
  for (int ab = 0; ab < K * K; ++ab) {
    wstat_to_double(&st[ab], B, shift, &n, s1, s2);
    n *= beta;
    for (int r = 0; r < B; ++r) s1[r] *= beta;
    for (int r = 0; r < B * B; ++r) s2[r] *= beta;
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_W_NIW, sweep, (uint32_t)ab);
    if (gg_fixed_cov_mean_draw(&s, h->mu_0, sigma + (size_t)ab * B * B, B, n, s1, mu + (size_t)ab * B)) return -2;
    wstat_to_double(&so[ab], B, shift, &n, s1, s2);
    n *= beta;
    for (int r = 0; r < B; ++r) s1[r] *= beta;
    for (int r = 0; r < B * B; ++r) s2[r] *= beta;
    gg_centered_scatter(n, s1, s2, mu + (size_t)ab * B, B, C);
    for (int i = 0; i < B * B; ++i) S[i] += C[i];
    ntot += n;
  }
  if (special) {
    wstat_to_double(&sd, B, shift, &n, s1, s2);
    n *= beta;
    for (int r = 0; r < B; ++r) s1[r] *= beta;
    for (int r = 0; r < B * B; ++r) s2[r] *= beta;
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_W_NIW_SELF, sweep, 0u);
    gg_niw_posterior_draw(&s, hs, B, n, s1, s2, mu_self, sigma_self);
  }
  double Sig[GG_MAXB * GG_MAXB];
  gg_stream s;
  gg_stream_init(&s, seed, GG_SITE_W_SHCOV, sweep, 0u);
  gg_iw_draw(&s, S, h->nu_0 + ntot, B, Sig);
  for (int ab = 0; ab < K * K; ++ab)
    for (int i = 0; i < B * B; ++i) sigma[(size_t)ab * B * B + i] = Sig[i];
  return 0;
}

int orc_sbm_wsh_sweep(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, int32_t* c, double* m, double* mu,
                      double* sigma, double* mu_self, double* sigma_self, const double* pi, const double* prior,
                      const double* niw_self, int special, uint64_t seed, uint32_t sweep, int parts, double beta) {
  if (B > GG_MAXB) return -1;
  gg_niw h, hs;
  unpack_niw(prior, B, &h);
  unpack_niw(niw_self, B, &hs);
  if (parts & ORC_PART_PARAMS) {
    int rc = sbm_wsh_resample_mu_sig(N, K, B, A, q, shift, c, mu, sigma, mu_self, sigma_self, &h, &hs, special, seed, sweep, beta);
    if (rc) return rc;
  }
  if (parts & ORC_PART_C) {
    int rc = sbm_w_resample_c(N, K, B, A, q, shift, c, m, mu, sigma, special, seed, sweep, beta);
    if (rc) return rc;
  }
  if (parts & ORC_PART_M) resample_m(N, K, c, pi, m, seed, GG_SITE_W_M, sweep);
  return 0;
}

// ---------------------------------------------------------------------------
// Spec primitives exposed for tests
// ---------------------------------------------------------------------------
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  gg_u32x4 r = gg_philox4x32_10(c0, c1, c2, c3, k0, k1);
  for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
double orc_normal_at(uint64_t seed, uint32_t site, uint32_t sweep, uint32_t index) {
  gg_stream s;
  gg_stream_init(&s, seed, site, sweep, index);
  return gg_normal(&s);
}
// 64-bit index (a pair of nodes beyond 2^32): the uniform, and the first uniform of the stream, addressed that way
double orc_uniform_at64(uint64_t seed, uint32_t site, uint32_t sweep, uint64_t index) { return gg_uniform_at64(seed, site, sweep, index); }
double orc_stream_first64(uint64_t seed, uint32_t site, uint32_t sweep, uint64_t index) {
  gg_stream s;
  gg_stream_init64(&s, seed, site, sweep, index);
  return gg_stream_uniform(&s);
}

double orc_uniform_at(uint64_t seed, uint32_t site, uint32_t sweep, uint32_t index) {
  return gg_uniform_at(seed, site, sweep, index);
}
void orc_exp_vec(const double* x, double* y, long long n) { for (long long i = 0; i < n; ++i) y[i] = gg_exp(x[i]); }
void orc_log_vec(const double* x, double* y, long long n) { for (long long i = 0; i < n; ++i) y[i] = gg_log(x[i]); }
void orc_normal_vec(uint64_t seed, double* y, long long n) {
  gg_stream s; gg_stream_init(&s, seed, GG_SITE_INIT, 0u, 0u);
  for (long long i = 0; i < n; ++i) y[i] = gg_normal(&s);
}
void orc_gamma_vec(uint64_t seed, double a, double* y, long long n) {
  for (long long i = 0; i < n; ++i) { gg_stream s; gg_stream_init(&s, seed, GG_SITE_INIT, 1u, (uint32_t)i); y[i] = gg_gamma(&s, a); }
}
void orc_beta_vec(uint64_t seed, double a, double b, double* y, long long n) {
  for (long long i = 0; i < n; ++i) { gg_stream s; gg_stream_init(&s, seed, GG_SITE_INIT, 2u, (uint32_t)i); y[i] = gg_beta(&s, a, b); }
}
void orc_dirichlet_vec(uint64_t seed, const double* alpha, int K, double* y, long long n) {
  for (long long i = 0; i < n; ++i) gg_dirichlet_at(seed, GG_SITE_INIT, (uint32_t)(16 + i), alpha, K, y + i * K);
}
// n_draws NIW posterior draws from given statistics; mu_out [n][B], sigma_out [n][B*B]
void orc_niw_vec(uint64_t seed, const double* niw, int B, double n_obs, const double* s1, const double* s2,
                 double* mu_out, double* sigma_out, long long n) {
  gg_niw h; unpack_niw(niw, B, &h);
  for (long long i = 0; i < n; ++i) {
    gg_stream s; gg_stream_init(&s, seed, GG_SITE_INIT, 4u, (uint32_t)i);
    gg_niw_posterior_draw(&s, &h, B, n_obs, s1, s2, mu_out + i * B, sigma_out + i * B * B);
  }
}
int orc_gauss_table(const double* mu, const double* sigma, int B, double* out) { return gg_gauss_table(mu, sigma, B, out); }


// ---------------------------------------------------------------------------
// Joint SBM: ONE label vector for adjacency and weights (legacy GaussianStochasticBlockModel,
// stochastic_block_model.py:202-309, 589-678).
//   resample (:206-230): weight model of every block pair from the edges of the pair (A * c1_and_c2: the diagonal
//     belongs to pair (c_n, c_n), nothing is special-cased) -> resample_p (:232-250: edges / non-edges of the pair,
//     diagonal included) -> resample_c (:252-302) -> resample_m (:304-309).
//   resample_c: lp[cn1] = log m[cn1] + sum_{n2 != n1} [ Bern(p[cn1,cn2]).lp(A[n1,n2]) + Bern(p[cn2,cn1]).lp(A[n2,n1])
//                         + A[n1,n2] log N(W[n1,n2]; cn1,cn2) + A[n2,n1] log N(W[n2,n1]; cn2,cn1) ]
//                         + Bern(p[cn1,cn1]).lp(A[n1,n1]) + A[n1,n1] log N(W[n1,n1]; cn1,cn1).
// Count form: the Gaussian statistics of SBMGaussianWeightDistribution plus, per block, the number of potential
// partners (block size, minus one in the node's own block), with the coefficients of gg_joint_table.
// ---------------------------------------------------------------------------
static void joint_tables(int K, int B, const double* m, const double* p, const double* mu, const double* sigma,
                         double* logm, double* T, int* bad) {
  const int NTg = gg_nt_gauss(B), NT = NTg + 1;
  for (int k = 0; k < K; ++k) logm[k] = gg_log(m[k]);
  for (int ab = 0; ab < K * K; ++ab) {
    double* e = T + (size_t)ab * NT;
    if (gg_gauss_table(mu + (size_t)ab * B, sigma + (size_t)ab * B * B, B, e)) { *bad = 1; return; }
    gg_joint_table(e, NTg, p[ab]);
  }
}

static void joint_node_lp(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, const int32_t* c,
                          const double* logm, const double* T, int n1, double* lp, double beta) {
  const int NTg = gg_nt_gauss(B), NT = NTg + 1;
  std::vector<WStat> so(K), si(K);
  std::vector<int> cnt(K, 0);
  for (int k = 0; k < K; ++k) { wstat_zero(&so[k]); wstat_zero(&si[k]); }
  for (int n2 = 0; n2 < N; ++n2) {
    cnt[c[n2]] += 1;
    if (n2 == n1) continue;
    if (A[(size_t)n1 * N + n2]) wstat_add(&so[c[n2]], q + ((size_t)n1 * N + n2) * B, B);
    if (A[(size_t)n2 * N + n1]) wstat_add(&si[c[n2]], q + ((size_t)n2 * N + n1) * B, B);
  }
  std::vector<double> dso((size_t)K * NT), dsi((size_t)K * NT);
  for (int k = 0; k < K; ++k) {
    pack_stat(&so[k], B, shift, dso.data() + (size_t)k * NT);
    pack_stat(&si[k], B, shift, dsi.data() + (size_t)k * NT);
    const double partners = (double)(cnt[k] - (c[n1] == k ? 1 : 0));
    dso[(size_t)k * NT + NTg] = partners;
    dsi[(size_t)k * NT + NTg] = partners;
  }
  double self[GG_MAXNT];
  WStat ss;
  wstat_zero(&ss);
  if (A[(size_t)n1 * N + n1]) wstat_add(&ss, q + ((size_t)n1 * N + n1) * B, B);
  pack_stat(&ss, B, shift, self);
  self[NTg] = 1.0;
  for (int k = 0; k < K; ++k) lp[k] = gg_block_lp(k, K, NT, logm, T, dso.data(), dsi.data(), self, beta);
}

// parts: ORC_PART_PARAMS = weight models + p, ORC_PART_C, ORC_PART_M.
int orc_sbm_joint_sweep(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, int32_t* c, double* m,
                        double* p, double* mu, double* sigma, const double* pi, const double* niw, double tau1, double tau0,
                        uint64_t seed, uint32_t sweep, int parts, double beta) {
  if (B > GG_MAXB) return -1;
  gg_niw h;
  unpack_niw(niw, B, &h);
  if (parts & ORC_PART_PARAMS) {
    double mus[GG_MAXB], sigs[GG_MAXB * GG_MAXB], ps = 0.0;
    sbm_w_resample_mu_sig(N, K, B, A, q, shift, c, mu, sigma, mus, sigs, &h, &h, 0, seed, sweep, beta);
    sbm_adj_resample_p(N, K, A, c, p, &ps, tau1, tau0, 0, seed, sweep, beta);
  }
  if ((parts & ORC_PART_C) && K > 1) {
    const int NT = gg_nt_gauss(B) + 1;
    std::vector<double> logm(K), T((size_t)K * K * NT), lp(K), scratch(2 * gg_next_pow2(K));
    int bad = 0;
    joint_tables(K, B, m, p, mu, sigma, logm.data(), T.data(), &bad);
    if (bad) return -2;
    for (int n1 = 0; n1 < N; ++n1) {
      joint_node_lp(N, K, B, A, q, shift, c, logm.data(), T.data(), n1, lp.data(), beta);
      const double u = gg_uniform_at(seed, GG_SITE_W_C, sweep, (uint32_t)n1);
      c[n1] = gg_categorical(lp.data(), K, u, scratch.data());
    }
  }
  if (parts & ORC_PART_M) resample_m(N, K, c, pi, m, seed, GG_SITE_W_M, sweep);
  return 0;
}

int orc_sbm_joint_node_lp(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, const int32_t* c,
                          const double* m, const double* p, const double* mu, const double* sigma, int n1, double* lp) {
  const int NT = gg_nt_gauss(B) + 1;
  std::vector<double> logm(K), T((size_t)K * K * NT);
  int bad = 0;
  joint_tables(K, B, m, p, mu, sigma, logm.data(), T.data(), &bad);
  if (bad) return -2;
  joint_node_lp(N, K, B, A, q, shift, c, logm.data(), T.data(), n1, lp, 1.0);
  return 0;
}

// The literal loop nest of stochastic_block_model.py:266-299 (cn1 outer, n2 inner, libm log per term, per-edge Gaussian
// log-densities of the de-quantised weights) - slow, obviously faithful.
int orc_sbm_joint_node_lp_literal(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, const int32_t* c,
                                  const double* m, const double* p, const double* mu, const double* sigma, int n1, double* lp) {
  double x[GG_MAXB];
  auto bern = [](double pp, int a) { return a ? log(pp) : log(1.0 - pp); };
  auto wll = [&](int i, int j, int ci, int cj) {
    for (int b = 0; b < B; ++b) x[b] = ldexp((double)q[((size_t)i * N + j) * B + b], -shift);
    return mvn_logpdf(x, mu + (size_t)(ci * K + cj) * B, sigma + (size_t)(ci * K + cj) * B * B, B);
  };
  for (int cn1 = 0; cn1 < K; ++cn1) {
    double ll = log(m[cn1]);
    for (int n2 = 0; n2 < N; ++n2) {
      const int cn2 = c[n2];
      if (n2 != n1) {
        const int a12 = A[(size_t)n1 * N + n2] ? 1 : 0, a21 = A[(size_t)n2 * N + n1] ? 1 : 0;
        ll += bern(p[cn1 * K + cn2], a12);
        ll += bern(p[cn2 * K + cn1], a21);
        if (a12) ll += wll(n1, n2, cn1, cn2);
        if (a21) ll += wll(n2, n1, cn2, cn1);
      } else {
        const int a11 = A[(size_t)n1 * N + n1] ? 1 : 0;
        ll += bern(p[cn1 * K + cn1], a11);
        if (a11) ll += wll(n1, n1, cn1, cn1);
      }
    }
    lp[cn1] = ll;
  }
  return 0;
}

// log p(A, W | c, p, mu, Sigma): every ordered pair, diagonal included (the sum the conditional above is a slice of).
double orc_sbm_joint_loglik(int N, int K, int B, const uint8_t* A, const int32_t* q, int shift, const int32_t* c,
                            const double* p, const double* mu, const double* sigma) {
  double ll = 0.0, x[GG_MAXB];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const int ab = c[i] * K + c[j];
      const double P = p[ab], Pc = P < 1e-64 ? 1e-64 : (P > 1 - 1e-64 ? 1 - 1e-64 : P);
      if (A[(size_t)i * N + j]) {
        ll += log(P);
        for (int b = 0; b < B; ++b) x[b] = ldexp((double)q[((size_t)i * N + j) * B + b], -shift);
        ll += mvn_logpdf(x, mu + (size_t)ab * B, sigma + (size_t)ab * B * B, B);
      } else {
        ll += log(1.0 - Pc);
      }
    }
  return ll;
}

}  // extern "C"
