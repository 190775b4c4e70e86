// oracle_eigen.cpp — CPU restatement of the logistic eigenmodel Gibbs sweep
// (graphistician/eigenmodel.py:822-965, base class :14-133, logistic link :795-820).
// TEST INFRASTRUCTURE ONLY (see oracle_sbm.cpp).  PARITY UNPINNED: the reference
// class cannot even be imported under Python 2 (undefined base classes at
// eigenmodel.py:1319-1320) and pypolyagamma is absent; fidelity rests on the line
// citations below, on oracle/literal.py (numpy transliteration of the same
// formulas) and on distributional tests (PG moments, Geweke).
//
// Loops are written the way the reference writes them (dense N x N Omega, double
// loop over nn for every node); no symmetrised storage, no tiling.
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../graphistician_amd/csrc/spec/gg_philox.h"
#include "../graphistician_amd/csrc/spec/gg_math.h"
#include "../graphistician_amd/csrc/spec/gg_draws.h"
#include "../graphistician_amd/csrc/spec/gg_pg.h"
#include "../graphistician_amd/csrc/spec/gg_eig.h"

#define ORC_PART_OMEGA 8
#define ORC_PART_F 16
#define ORC_PART_MU0 32
#define ORC_PART_LMBDA 64

extern "C" {

// eigenmodel.py:75-83: Mu = mu_0 + (F * lambda) F^T, diagonal forced to mu_0.
static double eig_psi(int N, int D, const double* F, const double* lmbda, double mu_0, int i, int j) {
  if (i == j) return mu_0;
  double v = 0.0;
  for (int d = 0; d < D; ++d) v += F[(size_t)i * D + d] * lmbda[d] * F[(size_t)j * D + d];
  return mu_0 + v;
}

// Several graphs per model and missing data.  The reference's Gibbs steps take lists, `for A, O in zip(As, Omegas)`
// (eigenmodel.py:836-857, :889, :918, :946); an entry of A given as ORC_A_MISSING (2) is an unobserved one (NaN in the
// reference, whose log-likelihood zeroes it, :121-127, and whose Gibbs updates would turn NaN on it): it has no auxiliary
// variable and no term in any conditional.  Graph g draws from Philox site (site + (g << 8)); As is [nG][N][N].
#define ORC_A_MISSING 2

static double eig_kappa(uint8_t a, double beta) { return a == ORC_A_MISSING ? 0.0 : beta * ((a ? 1.0 : 0.0) - 0.5); }

// eigenmodel.py:859-874 _resample_Omega: omega_ij ~ PG(1, Mu_ij) for all N^2 entries (of every graph).
// beta != 1 (AIS; build-defined, the reference leaves it a TODO at graph_models.py:478): the likelihood to the power beta
// is augmented with omega_ij ~ PG(beta, Mu_ij) and kappa = beta (A - 1/2) (spec/gg_pg.h: gg_pg_frac).
void orc_eig_resample_omega_multi(int nG, int N, int D, const uint8_t* As, const double* F, const double* lmbda, double mu_0,
                                  uint64_t seed, uint32_t sweep, double* Omegas, double beta) {
  for (int g = 0; g < nG; ++g) {
    const uint8_t* A = As + (size_t)g * N * N;
    double* Omega = Omegas + (size_t)g * N * N;
    const uint32_t site = (uint32_t)GG_SITE_EIG_OMEGA + ((uint32_t)g << 8);
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const double psi = eig_psi(N, D, F, lmbda, mu_0, i, j);
        double o = 0.0;
        if (A[(size_t)i * N + j] != ORC_A_MISSING && beta != 0.0) {
          gg_stream s;
          gg_stream_init64(&s, seed, site, sweep, (uint64_t)i * (uint64_t)N + (uint64_t)j);
          o = beta == 1.0 ? gg_pg1(&s, psi) : gg_pg_frac(&s, beta, psi);
        }
        Omega[(size_t)i * N + j] = o;
      }
  }
}

// eigenmodel.py:876-908 _resample_F: sequential over nodes, F[n] updated in place.  Precision and mean term are the
// reference's loops (over the graphs, then over nn); the draw F[n] ~ N(prec^{-1} h, prec^{-1}) (:908,
// np.random.multivariate_normal) is made by perturbation (spec/gg_eig.h): b = h + z / sqrt(sigma_F) +
// sum_nn sqrt(sum of the pair's omegas) eps[n,nn] f_nn * lambda, x = prec^{-1} b - the same law, no matrix square root.
// The scan over the nodes [n_begin, n_end) only (the whole scan: [0, N)).  F must hold what the scan has left behind when it
// reaches n_begin - the NEW features of the nodes before it, the old ones from there on -, so that a window of a long scan can be
// checked on its own (tests at N = 8192, where 8192 dependent steps of O(N D^2) take minutes on one core).
//
// ref_draw: the draw is made THE REFERENCE'S WAY instead (eigenmodel.py:903-908: `post_cov = inv(post_prec)`, `post_mu =
// post_cov.dot(h)`, `np.random.multivariate_normal(post_mu, post_cov)`): Sigma_post by inversion, x = Sigma_post h + L z with
// L the Cholesky factor of Sigma_post and z the node's D standard normals from their own Philox site (numpy factorises the
// covariance by SVD; any square root gives the same law).  The engine never draws this way; the form exists so that the
// perturbation draw is pinned to the reference's statement and not only to itself (tests/test_spec_and_oracle.py compares
// the two chains in law).
static int eig_resample_F_window(int nG, int N, int D, const uint8_t* As, const double* Omegas, double* F, const double* lmbda,
                                 double mu_0, double sigma_F, uint64_t seed, uint32_t sweep, double beta, int n_begin, int n_end,
                                 int ref_draw = 0) {
  std::vector<double> prec((size_t)D * D), h(D), b(D), x(D), wsum(N);
  for (int n = n_begin; n < n_end; ++n) {
    for (int a = 0; a < D * D; ++a) prec[a] = 0.0;
    for (int a = 0; a < D; ++a) {
      prec[a * D + a] = 1.0 / sigma_F; h[a] = 0.0;
      b[a] = gg_normal_at(seed, GG_SITE_EIG_F, sweep, (uint32_t)(n * D + a)) / gg_sqrt(sigma_F);
    }
    for (int nn = 0; nn < N; ++nn) wsum[nn] = 0.0;
    for (int g = 0; g < nG; ++g) {
      const uint8_t* A = As + (size_t)g * N * N;
      const double* Omega = Omegas + (size_t)g * N * N;
      for (int nn = 0; nn < N; ++nn) {
        if (nn == n) continue;
        const double o1 = Omega[(size_t)n * N + nn], o2 = Omega[(size_t)nn * N + n];
        // O * zcent = kappa - O mu_0 with kappa = A - 1/2 (eigenmodel.py:813-820,892)
        const double k1 = eig_kappa(A[(size_t)n * N + nn], beta) - o1 * mu_0;
        const double k2 = eig_kappa(A[(size_t)nn * N + n], beta) - o2 * mu_0;
        wsum[nn] += o1 + o2;
        const double* f = F + (size_t)nn * D;
        for (int a = 0; a < D; ++a) {
          for (int c = 0; c < D; ++c) prec[a * D + c] += f[a] * f[c] * lmbda[a] * lmbda[c] * (o1 + o2);
          h[a] += (k1 + k2) * f[a] * lmbda[a];
        }
      }
    }
    if (ref_draw) {
      // post_cov = inv(post_prec): column by column through the Cholesky factor of the precision
      double cov[GG_MAXD * GG_MAXD], col[GG_MAXD], e[GG_MAXD], mean[GG_MAXD];
      for (int cidx = 0; cidx < D; ++cidx) {
        for (int a = 0; a < D; ++a) e[a] = a == cidx ? 1.0 : 0.0;
        if (gg_solve_spd(prec.data(), e, D, col)) return -5;
        for (int a = 0; a < D; ++a) cov[a * D + cidx] = col[a];
      }
      for (int a = 0; a < D; ++a)
        for (int c = a + 1; c < D; ++c) { const double s = 0.5 * (cov[a * D + c] + cov[c * D + a]); cov[a * D + c] = cov[c * D + a] = s; }
      for (int a = 0; a < D; ++a) { double v = 0.0; for (int c = 0; c < D; ++c) v += cov[a * D + c] * h[c]; mean[a] = v; }   // post_mu
      if (gg_chol_d(cov, D)) return -5;                                                                                      // cov = L L^T
      for (int a = 0; a < D; ++a) {
        double v = mean[a];
        for (int c = 0; c <= a; ++c) v += cov[a * D + c] * gg_normal_at(seed, GG_SITE_EIG_F_REF, sweep, (uint32_t)(n * D + c));
        F[(size_t)n * D + a] = v;
      }
      continue;
    }
    for (int nn = 0; nn < N; ++nn) {
      if (nn == n) continue;
      const double w = gg_sqrt(wsum[nn]) * gg_eig_noise_at(seed, sweep, (uint32_t)N, (uint32_t)n, (uint32_t)nn);
      const double* f = F + (size_t)nn * D;
      for (int a = 0; a < D; ++a) b[a] += w * f[a] * lmbda[a];
    }
    for (int a = 0; a < D; ++a) b[a] += h[a];
    if (gg_solve_spd(prec.data(), b.data(), D, x.data())) return -5;
    for (int a = 0; a < D; ++a) F[(size_t)n * D + a] = x[a];
  }
  return 0;
}

int orc_eig_resample_F_multi(int nG, int N, int D, const uint8_t* As, const double* Omegas, double* F, const double* lmbda,
                             double mu_0, double sigma_F, uint64_t seed, uint32_t sweep, double beta) {
  return eig_resample_F_window(nG, N, D, As, Omegas, F, lmbda, mu_0, sigma_F, seed, sweep, beta, 0, N);
}

// One graph, the nodes [n_begin, n_end) of the scan; ref_draw != 0: the reference's own form of the draw (see above).
int orc_eig_resample_F_window(int N, int D, const uint8_t* A, const double* Omega, double* F, const double* lmbda, double mu_0,
                              double sigma_F, uint64_t seed, uint32_t sweep, double beta, int n_begin, int n_end, int ref_draw) {
  if (D > GG_MAXD || n_begin < 0 || n_end > N || n_begin > n_end) return -1;
  return eig_resample_F_window(1, N, D, A, Omega, F, lmbda, mu_0, sigma_F, seed, sweep, beta, n_begin, n_end, ref_draw);
}

// The auxiliaries of _resample_Omega (eigenmodel.py:859-874) for the node subset idx only: out[a][b] = omega_{idx[a], idx[b]},
// each from the very stream the full draw gives that entry (one graph, plain site).
void orc_eig_omega_block(int N, int D, const uint8_t* A, const double* F, const double* lmbda, double mu_0, uint64_t seed,
                         uint32_t sweep, double beta, const int32_t* idx, int n_idx, double* out) {
  for (int a = 0; a < n_idx; ++a)
    for (int b = 0; b < n_idx; ++b) {
      const int i = idx[a], j = idx[b];
      const double psi = eig_psi(N, D, F, lmbda, mu_0, i, j);
      double o = 0.0;
      if (A[(size_t)i * N + j] != ORC_A_MISSING && beta != 0.0) {
        gg_stream s;
        gg_stream_init64(&s, seed, (uint32_t)GG_SITE_EIG_OMEGA, sweep, (uint64_t)i * (uint64_t)N + (uint64_t)j);
        o = beta == 1.0 ? gg_pg1(&s, psi) : gg_pg_frac(&s, beta, psi);
      }
      out[(size_t)a * n_idx + b] = o;
    }
}

// eigenmodel.py:910-932 _resample_mu_0.
double orc_eig_resample_mu0_multi(int nG, int N, int D, const uint8_t* As, const double* Omegas, const double* F,
                                  const double* lmbda, double mu_mu_0, double sigma_mu0, uint64_t seed, uint32_t sweep, double beta) {
  double prec = 1.0 / sigma_mu0, h = mu_mu_0 / sigma_mu0;
  for (int g = 0; g < nG; ++g) {
    const uint8_t* A = As + (size_t)g * N * N;
    const double* Omega = Omegas + (size_t)g * N * N;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const double o = Omega[(size_t)i * N + j];
        prec += o;
        const double fdf = (i == j) ? 0.0 : eig_psi(N, D, F, lmbda, 0.0, i, j);
        h += eig_kappa(A[(size_t)i * N + j], beta) - o * fdf;
      }
  }
  const double var = 1.0 / prec;
  return var * h + gg_sqrt(var) * gg_normal_at(seed, GG_SITE_EIG_MU0, sweep, 0u);
}

// eigenmodel.py:934-965 _resample_lmbda.
int orc_eig_resample_lmbda_multi(int nG, int N, int D, const uint8_t* As, const double* Omegas, const double* F, double* lmbda,
                                 double mu_0, const double* mu_lmbda, double sigma_lmbda, uint64_t seed, uint32_t sweep, double beta) {
  std::vector<double> prec((size_t)D * D, 0.0), h(D), z(D), x(D);
  for (int a = 0; a < D; ++a) { prec[a * D + a] = 1.0 / sigma_lmbda; h[a] = mu_lmbda[a] / sigma_lmbda; }
  for (int g = 0; g < nG; ++g) {
    const uint8_t* A = As + (size_t)g * N * N;
    const double* Omega = Omegas + (size_t)g * N * N;
    for (int n1 = 0; n1 < N; ++n1)
      for (int n2 = n1 + 1; n2 < N; ++n2) {
        const double o12 = Omega[(size_t)n1 * N + n2], o21 = Omega[(size_t)n2 * N + n1];
        const double k12 = eig_kappa(A[(size_t)n1 * N + n2], beta) - o12 * mu_0;
        const double k21 = eig_kappa(A[(size_t)n2 * N + n1], beta) - o21 * mu_0;
        const double* f1 = F + (size_t)n1 * D;
        const double* f2 = F + (size_t)n2 * D;
        for (int a = 0; a < D; ++a) {
          for (int b = 0; b < D; ++b) prec[a * D + b] += (o21 + o12) * f1[a] * f1[b] * f2[a] * f2[b];
          h[a] += (k12 + k21) * f1[a] * f2[a];
        }
      }
  }
  for (int a = 0; a < D; ++a) z[a] = gg_normal_at(seed, GG_SITE_EIG_LMBDA, sweep, (uint32_t)a);
  if (gg_gauss_from_precision(prec.data(), h.data(), z.data(), D, x.data())) return -5;
  for (int a = 0; a < D; ++a) lmbda[a] = x[a];
  return 0;
}

// eigenmodel.py:836-857 resample(networks): Omega -> F -> mu_0 -> lambda.  Omegas is an output ([nG][N][N]) so that tests
// can inspect it.
int orc_eig_sweep_multi(int nG, int N, int D, const uint8_t* As, double* F, double* lmbda, double* mu_0, double* Omegas,
                        const double* mu_lmbda, double mu_mu_0, double sigma_mu0, double sigma_F, double sigma_lmbda,
                        int lmbda_given, uint64_t seed, uint32_t sweep, int parts, double beta) {
  if (D > GG_MAXD) return -1;
  if (parts & ORC_PART_OMEGA) orc_eig_resample_omega_multi(nG, N, D, As, F, lmbda, *mu_0, seed, sweep, Omegas, beta);
  if (parts & ORC_PART_F) {
    int rc = orc_eig_resample_F_multi(nG, N, D, As, Omegas, F, lmbda, *mu_0, sigma_F, seed, sweep, beta);
    if (rc) return rc;
  }
  if (parts & ORC_PART_MU0) *mu_0 = orc_eig_resample_mu0_multi(nG, N, D, As, Omegas, F, lmbda, mu_mu_0, sigma_mu0, seed, sweep, beta);
  if ((parts & ORC_PART_LMBDA) && !lmbda_given) {
    int rc = orc_eig_resample_lmbda_multi(nG, N, D, As, Omegas, F, lmbda, *mu_0, mu_lmbda, sigma_lmbda, seed, sweep, beta);
    if (rc) return rc;
  }
  return 0;
}

// One graph: the same sweep (graph 0 draws from the plain sites).
int orc_eig_sweep(int N, int D, const uint8_t* A, double* F, double* lmbda, double* mu_0, double* Omega,
                  const double* mu_lmbda, double mu_mu_0, double sigma_mu0, double sigma_F, double sigma_lmbda,
                  int lmbda_given, uint64_t seed, uint32_t sweep, int parts, double beta) {
  return orc_eig_sweep_multi(1, N, D, A, F, lmbda, mu_0, Omega, mu_lmbda, mu_mu_0, sigma_mu0, sigma_F, sigma_lmbda, lmbda_given,
                             seed, sweep, parts, beta);
}

// eigenmodel.py:110-129 log_likelihood with P = clip(logistic(Mu), 1e-16, 1 - 1e-16) (:803-811).
double orc_eig_loglik(int N, int D, const uint8_t* A, const double* F, const double* lmbda, double mu_0) {
  double ll = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (A[(size_t)i * N + j] == ORC_A_MISSING) continue;      // eigenmodel.py:125-127: nan_to_num(ll) = 0
      double P = 1.0 / (1.0 + exp(-eig_psi(N, D, F, lmbda, mu_0, i, j)));
      P = P < 1e-16 ? 1e-16 : (P > 1 - 1e-16 ? 1 - 1e-16 : P);
      ll += A[(size_t)i * N + j] ? log(P) : log(1.0 - P);
    }
  return ll;
}

// PG(b, z) draws with a real shape 0 < b < 1 (gg_pg_frac) for distributional tests.
void orc_pg_frac_vec(uint64_t seed, double b, double z, double* out, long long n) {
  for (long long i = 0; i < n; ++i) {
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_EIG_OMEGA, 1u, (uint32_t)i);
    out[i] = gg_pg_frac(&s, b, z);
  }
}

// The per-pair normals of the perturbation draw (spec/gg_eig.h), row n: out[nn] for nn < N.
void orc_eig_noise_row(uint64_t seed, uint32_t sweep, int N, int n, double* out) {
  for (int nn = 0; nn < N; ++nn) out[nn] = gg_eig_noise_at(seed, sweep, (uint32_t)N, (uint32_t)n, (uint32_t)nn);
}

// PG(1, z) draws for distributional tests: n draws, stream index = i.
void orc_pg1_vec(uint64_t seed, double z, double* out, long long n) {
  for (long long i = 0; i < n; ++i) {
    gg_stream s;
    gg_stream_init(&s, seed, GG_SITE_EIG_OMEGA, 0u, (uint32_t)i);
    out[i] = gg_pg1(&s, z);
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Probit eigenmodel (eigenmodel.py:223-379).  Written the way the reference writes
// it: dense N x N Z, double loops; the precision uses the constant 2 per unordered pair (:319) and the mean
// the centred auxiliaries zcent = Z - mu_0 (:312-321).
// ---------------------------------------------------------------------------
#include "../graphistician_amd/csrc/spec/gg_probit.h"

extern "C" {

// Lists of graphs and unobserved entries as for the logistic model above: `for A, Z in zip(As, Zs)`; an entry given as
// ORC_A_MISSING has no auxiliary variable (Z = 0 there and it is counted nowhere).
// eigenmodel.py:274-293 _resample_Z: truncated normals around Mu (diagonal Mu = mu_0), lower bound 0 where
// A == 1, upper bound 0 where A == 0.
void orc_probit_resample_Z_multi(int nG, int N, int D, const uint8_t* As, const double* F, const double* lmbda, double mu_0,
                                 uint64_t seed, uint32_t sweep, double* Zs) {
  for (int g = 0; g < nG; ++g) {
    const uint8_t* A = As + (size_t)g * N * N;
    double* Z = Zs + (size_t)g * N * N;
    const uint32_t site = (uint32_t)GG_SITE_EIG_Z + ((uint32_t)g << 8);
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const uint8_t a = A[(size_t)i * N + j];
        Z[(size_t)i * N + j] = a == ORC_A_MISSING ? 0.0
            : gg_truncnorm(gg_uniform_at64(seed, site, sweep, (uint64_t)i * (uint64_t)N + (uint64_t)j), eig_psi(N, D, F, lmbda, mu_0, i, j), a ? 1 : 0);
      }
  }
}

// eigenmodel.py:295-329 _resample_F (the draw by perturbation, as in orc_eig_resample_F_multi: pair weight = the number of
// observed entries of the pair, 2 per fully observed graph).
int orc_probit_resample_F_multi(int nG, int N, int D, const uint8_t* As, const double* Zs, double* F, const double* lmbda,
                                double mu_0, double sigma_F, uint64_t seed, uint32_t sweep) {
  std::vector<double> prec((size_t)D * D), h(D), b(D), x(D), wsum(N);
  for (int n = 0; n < N; ++n) {
    for (int a = 0; a < D * D; ++a) prec[a] = 0.0;
    for (int a = 0; a < D; ++a) {
      prec[a * D + a] = 1.0 / sigma_F; h[a] = 0.0;
      b[a] = gg_normal_at(seed, GG_SITE_EIG_F, sweep, (uint32_t)(n * D + a)) / gg_sqrt(sigma_F);
    }
    for (int nn = 0; nn < N; ++nn) wsum[nn] = 0.0;
    for (int g = 0; g < nG; ++g) {
      const uint8_t* A = As + (size_t)g * N * N;
      const double* Z = Zs + (size_t)g * N * N;
      for (int nn = 0; nn < N; ++nn) {
        if (nn == n) continue;
        const double* f = F + (size_t)nn * D;
        const bool ob1 = A[(size_t)nn * N + n] != ORC_A_MISSING, ob2 = A[(size_t)n * N + nn] != ORC_A_MISSING;
        const double z1 = ob1 ? Z[(size_t)nn * N + n] - mu_0 : 0.0, z2 = ob2 ? Z[(size_t)n * N + nn] - mu_0 : 0.0;
        const double cnt = (ob1 ? 1.0 : 0.0) + (ob2 ? 1.0 : 0.0);
        wsum[nn] += cnt;
        for (int a = 0; a < D; ++a) {
          for (int c = 0; c < D; ++c) prec[a * D + c] += cnt * f[a] * f[c] * lmbda[a] * lmbda[c];
          h[a] += z1 * f[a] * lmbda[a];
          h[a] += z2 * f[a] * lmbda[a];
        }
      }
    }
    for (int nn = 0; nn < N; ++nn) {
      if (nn == n) continue;
      const double w = gg_sqrt(wsum[nn]) * gg_eig_noise_at(seed, sweep, (uint32_t)N, (uint32_t)n, (uint32_t)nn);
      const double* f = F + (size_t)nn * D;
      for (int a = 0; a < D; ++a) b[a] += w * f[a] * lmbda[a];
    }
    for (int a = 0; a < D; ++a) b[a] += h[a];
    if (gg_solve_spd(prec.data(), b.data(), D, x.data())) return -5;
    for (int a = 0; a < D; ++a) F[(size_t)n * D + a] = x[a];
  }
  return 0;
}

// eigenmodel.py:331-352 _resample_mu_0: one observation per observed off-diagonal entry, residual Z - F Lambda F^T.
double orc_probit_resample_mu0_multi(int nG, int N, int D, const uint8_t* As, const double* Zs, const double* F, const double* lmbda,
                                     double mu_mu_0, double sigma_mu0, uint64_t seed, uint32_t sweep) {
  double prec = 1.0 / sigma_mu0, h = mu_mu_0 / sigma_mu0;
  for (int g = 0; g < nG; ++g) {
    const uint8_t* A = As + (size_t)g * N * N;
    const double* Z = Zs + (size_t)g * N * N;
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j)
        if (i != j && A[(size_t)i * N + j] != ORC_A_MISSING) {
          prec += 1.0;
          h += Z[(size_t)i * N + j] - eig_psi(N, D, F, lmbda, 0.0, i, j);
        }
  }
  const double var = 1.0 / prec;
  return var * h + gg_sqrt(var) * gg_normal_at(seed, GG_SITE_EIG_MU0, sweep, 0u);
}

// eigenmodel.py:354-379 _resample_lmbda.
int orc_probit_resample_lmbda_multi(int nG, int N, int D, const uint8_t* As, const double* Zs, const double* F, double* lmbda,
                                    double mu_0, const double* mu_lmbda, double sigma_lmbda, uint64_t seed, uint32_t sweep) {
  std::vector<double> prec((size_t)D * D, 0.0), h(D), z(D), x(D);
  for (int a = 0; a < D; ++a) { prec[a * D + a] = 1.0 / sigma_lmbda; h[a] = mu_lmbda[a] / sigma_lmbda; }
  for (int g = 0; g < nG; ++g) {
    const uint8_t* A = As + (size_t)g * N * N;
    const double* Z = Zs + (size_t)g * N * N;
    for (int n1 = 0; n1 < N; ++n1)
      for (int n2 = n1 + 1; n2 < N; ++n2) {
        const double* f1 = F + (size_t)n1 * D;
        const double* f2 = F + (size_t)n2 * D;
        const bool ob12 = A[(size_t)n1 * N + n2] != ORC_A_MISSING, ob21 = A[(size_t)n2 * N + n1] != ORC_A_MISSING;
        const double z12 = ob12 ? Z[(size_t)n1 * N + n2] - mu_0 : 0.0, z21 = ob21 ? Z[(size_t)n2 * N + n1] - mu_0 : 0.0;
        const double cnt = (ob12 ? 1.0 : 0.0) + (ob21 ? 1.0 : 0.0);
        for (int a = 0; a < D; ++a) {
          for (int b = 0; b < D; ++b) prec[a * D + b] += cnt * f1[a] * f1[b] * f2[a] * f2[b];
          h[a] += z12 * f1[a] * f2[a];
          h[a] += z21 * f1[a] * f2[a];
        }
      }
  }
  for (int a = 0; a < D; ++a) z[a] = gg_normal_at(seed, GG_SITE_EIG_LMBDA, sweep, (uint32_t)a);
  if (gg_gauss_from_precision(prec.data(), h.data(), z.data(), D, x.data())) return -5;
  for (int a = 0; a < D; ++a) lmbda[a] = x[a];
  return 0;
}

// eigenmodel.py:246-272 resample(networks): Z -> F -> mu_0 -> lambda.  Zs is an output ([nG][N][N]).
int orc_probit_sweep_multi(int nG, int N, int D, const uint8_t* As, double* F, double* lmbda, double* mu_0, double* Zs,
                           const double* mu_lmbda, double mu_mu_0, double sigma_mu0, double sigma_F, double sigma_lmbda,
                           int lmbda_given, uint64_t seed, uint32_t sweep, int parts) {
  if (D > GG_MAXD) return -1;
  if (parts & ORC_PART_OMEGA) orc_probit_resample_Z_multi(nG, N, D, As, F, lmbda, *mu_0, seed, sweep, Zs);
  if (parts & ORC_PART_F) {
    int rc = orc_probit_resample_F_multi(nG, N, D, As, Zs, F, lmbda, *mu_0, sigma_F, seed, sweep);
    if (rc) return rc;
  }
  if (parts & ORC_PART_MU0) *mu_0 = orc_probit_resample_mu0_multi(nG, N, D, As, Zs, F, lmbda, mu_mu_0, sigma_mu0, seed, sweep);
  if ((parts & ORC_PART_LMBDA) && !lmbda_given) {
    int rc = orc_probit_resample_lmbda_multi(nG, N, D, As, Zs, F, lmbda, *mu_0, mu_lmbda, sigma_lmbda, seed, sweep);
    if (rc) return rc;
  }
  return 0;
}

int orc_probit_sweep(int N, int D, const uint8_t* A, double* F, double* lmbda, double* mu_0, double* Z,
                     const double* mu_lmbda, double mu_mu_0, double sigma_mu0, double sigma_F, double sigma_lmbda,
                     int lmbda_given, uint64_t seed, uint32_t sweep, int parts) {
  return orc_probit_sweep_multi(1, N, D, A, F, lmbda, mu_0, Z, mu_lmbda, mu_mu_0, sigma_mu0, sigma_F, sigma_lmbda, lmbda_given,
                                seed, sweep, parts);
}

// eigenmodel.py:110-129 log_likelihood with P = clip(Phi(Mu), 1e-16, 1 - 1e-16) (:228-237).
double orc_probit_loglik(int N, int D, const uint8_t* A, const double* F, const double* lmbda, double mu_0) {
  double ll = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (A[(size_t)i * N + j] == ORC_A_MISSING) continue;
      double P = gg_norm_cdf(eig_psi(N, D, F, lmbda, mu_0, i, j));
      P = P < 1e-16 ? 1e-16 : (P > 1 - 1e-16 ? 1 - 1e-16 : P);
      ll += A[(size_t)i * N + j] ? log(P) : log(1.0 - P);
    }
  return ll;
}

void orc_ndtri_vec(const double* p, double* out, long long n) { for (long long i = 0; i < n; ++i) out[i] = gg_ndtri(p[i]); }
void orc_truncnorm_vec(const double* u, const double* mu, const int32_t* a, double* out, long long n) {
  for (long long i = 0; i < n; ++i) out[i] = gg_truncnorm(u[i], mu[i], a[i]);
}

}  // extern "C"
