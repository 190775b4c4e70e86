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

// eigenmodel.py:859-874 _resample_Omega: omega_ij ~ PG(1, Mu_ij) for all N^2 entries.
// beta != 1 (AIS; build-defined, the reference leaves it a TODO at graph_models.py:478): the likelihood to the power beta
// is augmented with omega_ij ~ PG(beta, Mu_ij) and kappa = beta (A - 1/2) (spec/gg_pg.h: gg_pg_frac).
void orc_eig_resample_omega(int N, int D, const double* F, const double* lmbda, double mu_0, uint64_t seed,
                            uint32_t sweep, double* Omega, double beta) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const double psi = eig_psi(N, D, F, lmbda, mu_0, i, j);
      if (beta == 1.0) Omega[(size_t)i * N + j] = gg_pg1_at(seed, sweep, (uint32_t)N, (uint32_t)i, (uint32_t)j, psi);
      else if (beta == 0.0) Omega[(size_t)i * N + j] = 0.0;
      else {
        gg_stream s;
        gg_stream_init(&s, seed, GG_SITE_EIG_OMEGA, sweep, (uint32_t)i * (uint32_t)N + (uint32_t)j);
        Omega[(size_t)i * N + j] = gg_pg_frac(&s, beta, psi);
      }
    }
}

// eigenmodel.py:876-908 _resample_F: sequential over nodes, F[n] updated in place.  Precision and mean term are the
// reference's double loop; the draw F[n] ~ N(prec^{-1} h, prec^{-1}) (:908, np.random.multivariate_normal) is made by
// perturbation (spec/gg_eig.h): b = h + z / sqrt(sigma_F) + sum_nn sqrt(O[n,nn] + O[nn,n]) eps[n,nn] f_nn * lambda,
// x = prec^{-1} b - the same law, no matrix square root.
int orc_eig_resample_F(int N, int D, const uint8_t* A, const double* Omega, double* F, const double* lmbda,
                       double mu_0, double sigma_F, uint64_t seed, uint32_t sweep, double beta) {
  std::vector<double> prec((size_t)D * D), h(D), b(D), x(D);
  for (int n = 0; n < N; ++n) {
    for (int a = 0; a < D * D; ++a) prec[a] = 0.0;
    for (int a = 0; a < D; ++a) {
      prec[a * D + a] = 1.0 / sigma_F; h[a] = 0.0;
      b[a] = gg_normal_at(seed, GG_SITE_EIG_F, sweep, (uint32_t)(n * D + a)) / gg_sqrt(sigma_F);
    }
    for (int nn = 0; nn < N; ++nn) {
      if (nn == n) continue;
      const double o1 = Omega[(size_t)n * N + nn], o2 = Omega[(size_t)nn * N + n];
      // O * zcent = kappa - O mu_0 with kappa = A - 1/2 (eigenmodel.py:813-820,892)
      const double k1 = beta * ((A[(size_t)n * N + nn] ? 1.0 : 0.0) - 0.5) - o1 * mu_0;
      const double k2 = beta * ((A[(size_t)nn * N + n] ? 1.0 : 0.0) - 0.5) - o2 * mu_0;
      const double w = gg_sqrt(o1 + o2) * gg_eig_noise_at(seed, sweep, (uint32_t)N, (uint32_t)n, (uint32_t)nn);
      const double* f = F + (size_t)nn * D;
      for (int a = 0; a < D; ++a) {
        for (int c = 0; c < D; ++c) prec[a * D + c] += f[a] * f[c] * lmbda[a] * lmbda[c] * (o1 + o2);
        h[a] += (k1 + k2) * f[a] * lmbda[a];
        b[a] += w * f[a] * lmbda[a];
      }
    }
    for (int a = 0; a < D; ++a) b[a] += h[a];
    if (gg_solve_spd(prec.data(), b.data(), D, x.data())) return -5;
    for (int a = 0; a < D; ++a) F[(size_t)n * D + a] = x[a];
  }
  return 0;
}

// eigenmodel.py:910-932 _resample_mu_0.
double orc_eig_resample_mu0(int N, int D, const uint8_t* A, const double* Omega, const double* F,
                            const double* lmbda, double mu_mu_0, double sigma_mu0, uint64_t seed, uint32_t sweep, double beta) {
  double prec = 1.0 / sigma_mu0, h = mu_mu_0 / sigma_mu0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const double o = Omega[(size_t)i * N + j];
      prec += o;
      const double fdf = (i == j) ? 0.0 : eig_psi(N, D, F, lmbda, 0.0, i, j);
      h += beta * ((A[(size_t)i * N + j] ? 1.0 : 0.0) - 0.5) - o * fdf;
    }
  const double var = 1.0 / prec;
  return var * h + gg_sqrt(var) * gg_normal_at(seed, GG_SITE_EIG_MU0, sweep, 0u);
}

// eigenmodel.py:934-965 _resample_lmbda.
int orc_eig_resample_lmbda(int N, int D, const uint8_t* A, const double* Omega, const double* F, double* lmbda,
                           double mu_0, const double* mu_lmbda, double sigma_lmbda, uint64_t seed, uint32_t sweep, double beta) {
  std::vector<double> prec((size_t)D * D, 0.0), h(D), z(D), x(D);
  for (int a = 0; a < D; ++a) { prec[a * D + a] = 1.0 / sigma_lmbda; h[a] = mu_lmbda[a] / sigma_lmbda; }
  for (int n1 = 0; n1 < N; ++n1)
    for (int n2 = n1 + 1; n2 < N; ++n2) {
      const double o12 = Omega[(size_t)n1 * N + n2], o21 = Omega[(size_t)n2 * N + n1];
      const double k12 = beta * ((A[(size_t)n1 * N + n2] ? 1.0 : 0.0) - 0.5) - o12 * mu_0;
      const double k21 = beta * ((A[(size_t)n2 * N + n1] ? 1.0 : 0.0) - 0.5) - o21 * mu_0;
      const double* f1 = F + (size_t)n1 * D;
      const double* f2 = F + (size_t)n2 * D;
      for (int a = 0; a < D; ++a) {
        for (int b = 0; b < D; ++b) prec[a * D + b] += (o21 + o12) * f1[a] * f1[b] * f2[a] * f2[b];
        h[a] += (k12 + k21) * f1[a] * f2[a];
      }
    }
  for (int a = 0; a < D; ++a) z[a] = gg_normal_at(seed, GG_SITE_EIG_LMBDA, sweep, (uint32_t)a);
  if (gg_gauss_from_precision(prec.data(), h.data(), z.data(), D, x.data())) return -5;
  for (int a = 0; a < D; ++a) lmbda[a] = x[a];
  return 0;
}

// eigenmodel.py:836-857 resample: Omega -> F -> mu_0 -> lambda.  Omega is an output
// (N x N) so that tests can inspect it.
int orc_eig_sweep(int N, int D, const uint8_t* A, double* F, double* lmbda, double* mu_0, double* Omega,
                  const double* mu_lmbda, double mu_mu_0, double sigma_mu0, double sigma_F, double sigma_lmbda,
                  int lmbda_given, uint64_t seed, uint32_t sweep, int parts, double beta) {
  if (D > GG_MAXD) return -1;
  if (parts & ORC_PART_OMEGA) orc_eig_resample_omega(N, D, F, lmbda, *mu_0, seed, sweep, Omega, beta);
  if (parts & ORC_PART_F) {
    int rc = orc_eig_resample_F(N, D, A, Omega, F, lmbda, *mu_0, sigma_F, seed, sweep, beta);
    if (rc) return rc;
  }
  if (parts & ORC_PART_MU0) *mu_0 = orc_eig_resample_mu0(N, D, A, Omega, F, lmbda, mu_mu_0, sigma_mu0, seed, sweep, beta);
  if ((parts & ORC_PART_LMBDA) && !lmbda_given) {
    int rc = orc_eig_resample_lmbda(N, D, A, Omega, F, lmbda, *mu_0, mu_lmbda, sigma_lmbda, seed, sweep, beta);
    if (rc) return rc;
  }
  return 0;
}

// eigenmodel.py:110-129 log_likelihood with P = clip(logistic(Mu), 1e-16, 1 - 1e-16) (:803-811).
double orc_eig_loglik(int N, int D, const uint8_t* A, const double* F, const double* lmbda, double mu_0) {
  double ll = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
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

// eigenmodel.py:274-293 _resample_Z: truncated normals around Mu (diagonal Mu = mu_0), lower bound 0 where
// A == 1, upper bound 0 where A == 0.
void orc_probit_resample_Z(int N, int D, const uint8_t* A, const double* F, const double* lmbda, double mu_0,
                           uint64_t seed, uint32_t sweep, double* Z) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j)
      Z[(size_t)i * N + j] = gg_truncnorm_at(seed, sweep, (uint32_t)N, (uint32_t)i, (uint32_t)j,
                                             eig_psi(N, D, F, lmbda, mu_0, i, j), A[(size_t)i * N + j] ? 1 : 0);
}

// eigenmodel.py:295-329 _resample_F (the draw by perturbation, as in orc_eig_resample_F: pair weight 2).
int orc_probit_resample_F(int N, int D, const double* Z, double* F, const double* lmbda, double mu_0, double sigma_F,
                          uint64_t seed, uint32_t sweep) {
  std::vector<double> prec((size_t)D * D), h(D), b(D), x(D);
  for (int n = 0; n < N; ++n) {
    for (int a = 0; a < D * D; ++a) prec[a] = 0.0;
    for (int a = 0; a < D; ++a) {
      prec[a * D + a] = 1.0 / sigma_F; h[a] = 0.0;
      b[a] = gg_normal_at(seed, GG_SITE_EIG_F, sweep, (uint32_t)(n * D + a)) / gg_sqrt(sigma_F);
    }
    for (int nn = 0; nn < N; ++nn) {
      if (nn == n) continue;
      const double* f = F + (size_t)nn * D;
      const double z1 = Z[(size_t)nn * N + n] - mu_0, z2 = Z[(size_t)n * N + nn] - mu_0;
      const double w = gg_sqrt(2.0) * gg_eig_noise_at(seed, sweep, (uint32_t)N, (uint32_t)n, (uint32_t)nn);
      for (int a = 0; a < D; ++a) {
        for (int c = 0; c < D; ++c) prec[a * D + c] += 2.0 * f[a] * f[c] * lmbda[a] * lmbda[c];
        h[a] += z1 * f[a] * lmbda[a];
        h[a] += z2 * f[a] * lmbda[a];
        b[a] += w * f[a] * lmbda[a];
      }
    }
    for (int a = 0; a < D; ++a) b[a] += h[a];
    if (gg_solve_spd(prec.data(), b.data(), D, x.data())) return -5;
    for (int a = 0; a < D; ++a) F[(size_t)n * D + a] = x[a];
  }
  return 0;
}

// eigenmodel.py:331-352 _resample_mu_0: N^2 - N observations, residual Z - F Lambda F^T off the diagonal.
double orc_probit_resample_mu0(int N, int D, const double* Z, const double* F, const double* lmbda, double mu_mu_0,
                               double sigma_mu0, uint64_t seed, uint32_t sweep) {
  double prec = 1.0 / sigma_mu0 + ((double)N * N - N), h = mu_mu_0 / sigma_mu0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j)
      if (i != j) h += Z[(size_t)i * N + j] - eig_psi(N, D, F, lmbda, 0.0, i, j);
  const double var = 1.0 / prec;
  return var * h + gg_sqrt(var) * gg_normal_at(seed, GG_SITE_EIG_MU0, sweep, 0u);
}

// eigenmodel.py:354-379 _resample_lmbda.
int orc_probit_resample_lmbda(int N, int D, const double* Z, const double* F, double* lmbda, double mu_0,
                              const double* mu_lmbda, double sigma_lmbda, uint64_t seed, uint32_t sweep) {
  std::vector<double> prec((size_t)D * D, 0.0), h(D), z(D), x(D);
  for (int a = 0; a < D; ++a) { prec[a * D + a] = 1.0 / sigma_lmbda; h[a] = mu_lmbda[a] / sigma_lmbda; }
  for (int n1 = 0; n1 < N; ++n1)
    for (int n2 = n1 + 1; n2 < N; ++n2) {
      const double* f1 = F + (size_t)n1 * D;
      const double* f2 = F + (size_t)n2 * D;
      const double z12 = Z[(size_t)n1 * N + n2] - mu_0, z21 = Z[(size_t)n2 * N + n1] - mu_0;
      for (int a = 0; a < D; ++a) {
        for (int b = 0; b < D; ++b) prec[a * D + b] += 2.0 * f1[a] * f1[b] * f2[a] * f2[b];
        h[a] += z12 * f1[a] * f2[a];
        h[a] += z21 * f1[a] * f2[a];
      }
    }
  for (int a = 0; a < D; ++a) z[a] = gg_normal_at(seed, GG_SITE_EIG_LMBDA, sweep, (uint32_t)a);
  if (gg_gauss_from_precision(prec.data(), h.data(), z.data(), D, x.data())) return -5;
  for (int a = 0; a < D; ++a) lmbda[a] = x[a];
  return 0;
}

// eigenmodel.py:246-272 resample: Z -> F -> mu_0 -> lambda.  Z is an output (N x N).
int orc_probit_sweep(int N, int D, const uint8_t* A, double* F, double* lmbda, double* mu_0, double* Z,
                     const double* mu_lmbda, double mu_mu_0, double sigma_mu0, double sigma_F, double sigma_lmbda,
                     int lmbda_given, uint64_t seed, uint32_t sweep, int parts) {
  if (D > GG_MAXD) return -1;
  if (parts & ORC_PART_OMEGA) orc_probit_resample_Z(N, D, A, F, lmbda, *mu_0, seed, sweep, Z);
  if (parts & ORC_PART_F) {
    int rc = orc_probit_resample_F(N, D, Z, F, lmbda, *mu_0, sigma_F, seed, sweep);
    if (rc) return rc;
  }
  if (parts & ORC_PART_MU0) *mu_0 = orc_probit_resample_mu0(N, D, Z, F, lmbda, mu_mu_0, sigma_mu0, seed, sweep);
  if ((parts & ORC_PART_LMBDA) && !lmbda_given) {
    int rc = orc_probit_resample_lmbda(N, D, Z, F, lmbda, *mu_0, mu_lmbda, sigma_lmbda, seed, sweep);
    if (rc) return rc;
  }
  return 0;
}

// eigenmodel.py:110-129 log_likelihood with P = clip(Phi(Mu), 1e-16, 1 - 1e-16) (:228-237).
double orc_probit_loglik(int N, int D, const uint8_t* A, const double* F, const double* lmbda, double mu_0) {
  double ll = 0.0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
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
