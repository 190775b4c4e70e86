// oracle_fast.cpp — the "optimised CPU" flavour of the baseline: the same Gibbs sweep as
// oracle_sbm.cpp (same arithmetic spec, same Philox addressing, hence the same chain bit for bit — checked in
// tests/test_spec_and_oracle.py), organised the way a careful CPU implementation would be:
//   * per-node block statistics built once per half-sweep in one OpenMP pass over the rows of X and of its
//     stored transpose (the reference recounts them per node: adjacency.py:661-693, weights.py:451-470),
//   * the sequential scan then costs O(K^2) per node plus an O(N) column update per label change.
// TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and the tests): never linked into the product.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <omp.h>

#include "../graphistician_amd/csrc/spec/gg_philox.h"
#include "../graphistician_amd/csrc/spec/gg_math.h"
#include "../graphistician_amd/csrc/spec/gg_draws.h"
#include "../graphistician_amd/csrc/spec/gg_lp.h"

namespace {

struct Fast {
  int N, B, shift;
  const uint8_t* A;
  const int32_t* q;
  std::vector<uint8_t> AT;
  std::vector<int32_t> qT;
};

inline int nsi(int B) { return B + B * (B + 1) / 2; }

void unpack_niw(const double* packed, int B, gg_niw* h) {
  memset(h, 0, sizeof(gg_niw));
  for (int i = 0; i < B; ++i) h->mu_0[i] = packed[i];
  for (int i = 0; i < B * B; ++i) h->sigma_0[i] = packed[B + i];
  h->nu_0 = packed[B + B * B];
  h->kappa_0 = packed[B + B * B + 1];
}

void draw_m(int N, int K, const int32_t* c, const double* pi, double* m, uint64_t seed, uint32_t site, uint32_t sweep) {
  std::vector<double> alpha(K);
  for (int k = 0; k < K; ++k) alpha[k] = pi[k];
  for (int n = 0; n < N; ++n) alpha[c[n]] += 1.0;
  gg_dirichlet_at(seed, site, sweep, alpha.data(), K, m);
}

// integer statistics of one (node, block): [0] edge count, [1..B] sum q, then sum q_r q_s (r <= s)
inline void stat_add(long long* s, const int32_t* q, int B, long long sign) {
  s[0] += sign;
  int t = 1;
  for (int r = 0; r < B; ++r) s[t++] += sign * (long long)q[r];
  for (int r = 0; r < B; ++r)
    for (int u = r; u < B; ++u) s[t++] += sign * (long long)q[r] * (long long)q[u];
}

// -> spec-order doubles {n, s1_r, s2_rs (r <= s)} (oracle_sbm.cpp pack_stat: exact conversion, power-of-two scaling)
inline void stat_pack(const long long* s, int B, int shift, double* out) {
  out[0] = (double)s[0];
  int t = 1;
  for (int r = 0; r < B; ++r, ++t) out[t] = ldexp((double)s[t], -shift);
  for (; t < 1 + nsi(B); ++t) out[t] = ldexp((double)s[t], -2 * shift);
}

}  // namespace

extern "C" {

void* orc_fast_create(int N, int B, const uint8_t* A, const int32_t* q, int shift, int nthreads) {
  // never more threads than the caller's CPU share: oversubscribed OpenMP teams spin against each other
  omp_set_dynamic(0);
  omp_set_num_threads(nthreads > 0 ? nthreads : 1);
  Fast* f = new Fast;
  f->N = N; f->B = B; f->shift = shift; f->A = A; f->q = q;
  f->AT.resize((size_t)N * N);
  if (q) f->qT.resize((size_t)N * N * B);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      f->AT[(size_t)i * N + j] = A[(size_t)j * N + i];
      if (q)
        for (int r = 0; r < B; ++r) f->qT[((size_t)i * N + j) * B + r] = q[((size_t)j * N + i) * B + r];
    }
  return f;
}

void orc_fast_destroy(void* h) { delete static_cast<Fast*>(h); }

int orc_fast_threads(void) { return omp_get_max_threads(); }

// adjacency.py:615-618: resample_p -> resample_c -> resample_m (special_case_self_conns aware).
int orc_fast_adj_sweep(void* h, int K, int32_t* c, double* m, double* p, double* p_self, const double* pi, double tau1,
                       double tau0, int special, uint64_t seed, uint32_t sweep) {
  Fast* f = static_cast<Fast*>(h);
  const int N = f->N;
  const uint8_t *A = f->A, *AT = f->AT.data();
  std::vector<int32_t> Go((size_t)N * K, 0), Gi((size_t)N * K, 0);   // ones towards / from block k, self excluded
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    int32_t* go = &Go[(size_t)n * K];
    int32_t* gi = &Gi[(size_t)n * K];
    const uint8_t* ra = A + (size_t)n * N;
    const uint8_t* rt = AT + (size_t)n * N;
    for (int j = 0; j < N; ++j) {
      if (j == n) continue;
      go[c[j]] += ra[j] ? 1 : 0;
      gi[c[j]] += rt[j] ? 1 : 0;
    }
  }
  std::vector<long long> cnt(K, 0);
  for (int n = 0; n < N; ++n) cnt[c[n]] += 1;
  // resample_p (adjacency.py:620-648)
  std::vector<long long> n1((size_t)K * K, 0);
  long long d1 = 0;
  for (int n = 0; n < N; ++n) {
    for (int b = 0; b < K; ++b) n1[(size_t)c[n] * K + b] += Go[(size_t)n * K + b];
    const int a = A[(size_t)n * N + n] ? 1 : 0;
    d1 += a;
    if (!special) n1[(size_t)c[n] * K + c[n]] += a;
  }
  for (int a = 0; a < K; ++a)
    for (int b = 0; b < K; ++b) {
      long long pairs = cnt[a] * cnt[b];
      if (a == b && special) pairs -= cnt[a];
      gg_stream s;
      gg_stream_init(&s, seed, GG_SITE_ADJ_P, sweep, (uint32_t)(a * K + b));
      p[a * K + b] = gg_beta(&s, tau1 + 1.0 * (double)n1[(size_t)a * K + b], tau0 + 1.0 * (double)(pairs - n1[(size_t)a * K + b]));
    }
  if (special) {
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_ADJ_PSELF, sweep, 0u);
    *p_self = gg_beta(&s, tau1 + 1.0 * (double)d1, tau0 + 1.0 * (double)(N - d1));
  }
  // resample_c (adjacency.py:650-695)
  if (K > 1) {
    std::vector<double> logm(K), T((size_t)K * K * 2), lp(K), scratch(2 * gg_next_pow2(K)), so((size_t)K * 2), si((size_t)K * 2);
    for (int k = 0; k < K; ++k) logm[k] = gg_log(m[k]);
    for (int ab = 0; ab < K * K; ++ab) { T[ab * 2] = gg_log(p[ab]); T[ab * 2 + 1] = gg_log(1.0 - p[ab]); }
    for (int n = 0; n < N; ++n) {
      const int cn = c[n];
      for (int j = 0; j < K; ++j) {
        const long long others = cnt[j] - (cn == j ? 1 : 0);
        so[j * 2] = (double)Go[(size_t)n * K + j]; so[j * 2 + 1] = (double)(others - Go[(size_t)n * K + j]);
        si[j * 2] = (double)Gi[(size_t)n * K + j]; si[j * 2 + 1] = (double)(others - Gi[(size_t)n * K + j]);
      }
      double self[2];
      const double* selfp = 0;
      if (!special) { const double a = A[(size_t)n * N + n] ? 1.0 : 0.0; self[0] = a; self[1] = 1.0 - a; selfp = self; }
      for (int k = 0; k < K; ++k) lp[k] = gg_block_lp(k, K, 2, logm.data(), T.data(), so.data(), si.data(), selfp, 1.0);
      const int nw = gg_categorical(lp.data(), K, gg_uniform_at(seed, GG_SITE_ADJ_C, sweep, (uint32_t)n), scratch.data());
      if (nw != cn) {
        c[n] = nw;
        cnt[cn] -= 1; cnt[nw] += 1;
        const uint8_t* col = AT + (size_t)n * N;   // A[i][n]: i -> n, an out-edge of i
        const uint8_t* row = A + (size_t)n * N;    // A[n][i]: n -> i, an in-edge of i
#pragma omp parallel for schedule(static) if (N >= 32768)
        for (int i = 0; i < N; ++i) {
          if (i == n) continue;
          if (col[i]) { Go[(size_t)i * K + cn] -= 1; Go[(size_t)i * K + nw] += 1; }
          if (row[i]) { Gi[(size_t)i * K + cn] -= 1; Gi[(size_t)i * K + nw] += 1; }
        }
      }
    }
  }
  draw_m(N, K, c, pi, m, seed, GG_SITE_ADJ_M, sweep);
  return 0;
}

// weights.py:394-397: resample_mu_and_Sig -> resample_c -> resample_m.
int orc_fast_w_sweep(void* h, int K, int32_t* c, double* m, double* mu, double* sigma, double* mu_self, double* sigma_self,
                     const double* pi, const double* niw, const double* niw_self, int special, uint64_t seed, uint32_t sweep) {
  Fast* f = static_cast<Fast*>(h);
  const int N = f->N, B = f->B, shift = f->shift, NS = 1 + nsi(B), NT = gg_nt_gauss(B);
  if (B > GG_MAXB || !f->q) return -1;
  const uint8_t* A = f->A;
  const int32_t *q = f->q, *qT = f->qT.data();
  gg_niw hh, hs;
  unpack_niw(niw, B, &hh);
  unpack_niw(niw_self, B, &hs);
  std::vector<long long> Go((size_t)N * K * NS, 0), Gi((size_t)N * K * NS, 0);   // self excluded
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    long long* go = &Go[(size_t)n * K * NS];
    long long* gi = &Gi[(size_t)n * K * NS];
    for (int j = 0; j < N; ++j) {
      if (j == n) continue;
      const int32_t* qo = q + ((size_t)n * N + j) * B;
      const int32_t* qi = qT + ((size_t)n * N + j) * B;
      if (qo[0] != INT32_MIN) stat_add(go + (size_t)c[j] * NS, qo, B, 1);
      if (qi[0] != INT32_MIN) stat_add(gi + (size_t)c[j] * NS, qi, B, 1);
    }
  }
  // resample_mu_and_Sig (weights.py:399-413): block-pair statistics = sums of node statistics over the source block
  std::vector<long long> tab((size_t)K * K * NS, 0), sd(NS, 0);
  for (int n = 0; n < N; ++n) {
    for (int b = 0; b < K; ++b)
      for (int s = 0; s < NS; ++s) tab[((size_t)c[n] * K + b) * NS + s] += Go[((size_t)n * K + b) * NS + s];
    if (A[(size_t)n * N + n]) {
      const int32_t* qd = q + ((size_t)n * N + n) * B;
      stat_add(sd.data(), qd, B, 1);
      if (!special) stat_add(&tab[((size_t)c[n] * K + c[n]) * NS], qd, B, 1);
    }
  }
  auto draw = [&](const long long* st, const gg_niw* hy, uint32_t site, uint32_t idx, double* mo, double* so) {
    double pk[GG_MAXNT], s1[GG_MAXB], s2[GG_MAXB * GG_MAXB];
    stat_pack(st, B, shift, pk);
    int t = 1 + B;
    for (int r = 0; r < B; ++r) s1[r] = pk[1 + r];
    for (int r = 0; r < B; ++r)
      for (int u = r; u < B; ++u) { s2[r * B + u] = pk[t]; s2[u * B + r] = pk[t]; ++t; }
    gg_stream s;
    gg_stream_init(&s, seed, site, sweep, idx);
    gg_niw_posterior_draw(&s, hy, B, pk[0], s1, s2, mo, so);
  };
  for (int ab = 0; ab < K * K; ++ab) draw(&tab[(size_t)ab * NS], &hh, GG_SITE_W_NIW, (uint32_t)ab, mu + (size_t)ab * B, sigma + (size_t)ab * B * B);
  if (special) draw(sd.data(), &hs, GG_SITE_W_NIW_SELF, 0u, mu_self, sigma_self);
  // resample_c (weights.py:415-488)
  if (K > 1) {
    std::vector<double> logm(K), T((size_t)K * K * NT), lp(K), scratch(2 * gg_next_pow2(K)), so((size_t)K * NT), si((size_t)K * NT);
    for (int k = 0; k < K; ++k) logm[k] = gg_log(m[k]);
    for (int ab = 0; ab < K * K; ++ab)
      if (gg_gauss_table(mu + (size_t)ab * B, sigma + (size_t)ab * B * B, B, &T[(size_t)ab * NT])) return -2;
    for (int n = 0; n < N; ++n) {
      const int cn = c[n];
      for (int j = 0; j < K; ++j) {
        stat_pack(&Go[((size_t)n * K + j) * NS], B, shift, &so[(size_t)j * NT]);
        stat_pack(&Gi[((size_t)n * K + j) * NS], B, shift, &si[(size_t)j * NT]);
      }
      double self[GG_MAXNT];
      const double* selfp = 0;
      if (!special) {
        long long ss[1 + GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2] = {0};
        if (A[(size_t)n * N + n]) stat_add(ss, q + ((size_t)n * N + n) * B, B, 1);
        stat_pack(ss, B, shift, self);
        selfp = self;
      }
      for (int k = 0; k < K; ++k) lp[k] = gg_block_lp(k, K, NT, logm.data(), T.data(), so.data(), si.data(), selfp);
      const int nw = gg_categorical(lp.data(), K, gg_uniform_at(seed, GG_SITE_W_C, sweep, (uint32_t)n), scratch.data());
      if (nw != cn) {
        c[n] = nw;
#pragma omp parallel for schedule(static) if (N >= 32768)
        for (int i = 0; i < N; ++i) {
          if (i == n) continue;
          const int32_t* qo = qT + ((size_t)n * N + i) * B;   // W[i][n]: an out-edge of i
          const int32_t* qi = q + ((size_t)n * N + i) * B;    // W[n][i]: an in-edge of i
          if (qo[0] != INT32_MIN) { stat_add(&Go[((size_t)i * K + cn) * NS], qo, B, -1); stat_add(&Go[((size_t)i * K + nw) * NS], qo, B, 1); }
          if (qi[0] != INT32_MIN) { stat_add(&Gi[((size_t)i * K + cn) * NS], qi, B, -1); stat_add(&Gi[((size_t)i * K + nw) * NS], qi, B, 1); }
        }
      }
    }
  }
  draw_m(N, K, c, pi, m, seed, GG_SITE_W_M, sweep);
  return 0;
}

}  // extern "C"
