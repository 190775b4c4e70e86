// gg_latent.hip — latent-distance adjacency model on the MI355X (graphistician/adjacency.py:191-437):
//   logit P_ij = -|l_i - l_j|^2 + mu_0 + mu_self [i == j];  resample = HMC on L, HMC on mu_0, HMC on mu_self.
// The reference differentiates with autograd and integrates with hips.inference.hmc (neither is in the image);
// the gradient is derived by hand and the HMC transition restated (spec/gg_latent.h; the test oracle restates the same definition:
// parity unpinned, float tolerance against the oracle's same definition).
//
// Every leapfrog step of the L update needs d log p / d L, i.e. one pass over the N x N adjacency and its stored
// transpose (row i of A gives the out-terms of node i, row i of A^T the in-terms): k_ld_pass, HBM-bound on
// 2 N^2 bytes, one workgroup per node.  The whole transition (momenta, 10 leapfrog steps, Metropolis test, step
// adaptation) runs from device-resident scalars: no host synchronisation inside a resample.
#include "gg_common.h"
#include "spec/gg_pg.h"
#include "spec/gg_latent.h"

#define LD_MAXDIM 4
#define LD_NSTEPS 10

// device scalars
enum { LS_MU0 = 0, LS_MUSELF, LS_STEP, LS_RATE, LS_LP0, LS_LP1, LS_K0, LS_K1, LS_LL, LS_SALL, LS_SDIAG, LS_Q, LS_P, LS_ACC,
       LS_QMU0, LS_QMUSELF, LS_COUNT };

struct gg_latent {
  gg_config cfg;
  int N, dim, Np;
  double sigma;
  double *L, *Q, *Pm;      // positions, proposal, momenta  [N][dim]
  double* row;             // per node: dim gradient sums, ll, s  [N][dim + 2]
  double* scal;            // LS_COUNT scalars
  int* acc;                // bit mask of accepted proposals of the last resample
  hipStream_t stream;
  hipEvent_t e0, e1;
  uint32_t sweep;
  double beta;
  gg_stats st;
};

// One pass: for node i, over all j != i,
//   x = -|q_i - q_j|^2 + mu_0,  P = sigmoid(x):
//   ll_i  = sum_j A_ij x - softplus(x)            (the directed terms of row i)
//   s_i   = sum_j A_ij - P
//   g_i   = sum_j (A_ij + A_ji - 2 P) (q_i - q_j) (times -2 later: d x / d q_i = -2 (q_i - q_j), P symmetric)
template <int DIM>
__global__ __launch_bounds__(256) void k_ld_pass(int N, int Np, const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                                 const double* __restrict__ Q, const double* __restrict__ scal, int mu0_slot,
                                                 double* __restrict__ row) {
  const int i = blockIdx.x;
  const double mu0 = scal[mu0_slot];
  double qi[DIM], g[DIM], ll = 0.0, s = 0.0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) { qi[d] = Q[(size_t)i * DIM + d]; g[d] = 0.0; }
  for (int j = threadIdx.x; j < N; j += 256) {
    if (j == i) continue;
    double t[DIM], d2 = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) { t[d] = qi[d] - Q[(size_t)j * DIM + d]; d2 += t[d] * t[d]; }
    const double x = -d2 + mu0;
    const double a1 = A[(size_t)i * Np + j] ? 1.0 : 0.0, a2 = AT[(size_t)i * Np + j] ? 1.0 : 0.0;
    double P, sp;
    gg_sigmoid_softplus(x, &P, &sp);
    ll += a1 * x - sp;
    s += a1 - P;
    const double c = a1 + a2 - 2.0 * P;
#pragma unroll
    for (int d = 0; d < DIM; ++d) g[d] += c * t[d];
  }
  __shared__ double s_red[4][DIM + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < DIM + 2; ++e) {
    double v = e < DIM ? g[e] : (e == DIM ? ll : s);
    for (int dl = 32; dl >= 1; dl >>= 1) v += __shfl_down(v, dl);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < DIM + 2)
    row[(size_t)i * (DIM + 2) + threadIdx.x] = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// Sums over nodes (fixed order), the diagonal terms, the prior: ll, d ll / d mu_0, d ll / d mu_self, and the
// log probability beta ll + log prior into scal[lp_slot].  One workgroup.
template <int DIM>
__global__ __launch_bounds__(256) void k_ld_reduce(int N, const double* __restrict__ row, const uint8_t* __restrict__ dA,
                                                   const double* __restrict__ Q, double* __restrict__ scal, int mu0_slot,
                                                   int muself_slot, double sigma, double beta, int lp_slot) {
  const double mu0 = scal[mu0_slot], mus = scal[muself_slot];
  const double xd = mu0 + mus;
  const double Pd = gg_sigmoid(xd), spd = gg_softplus(xd);
  double ll = 0.0, s = 0.0, sd = 0.0, sq = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const double a = dA[i] ? 1.0 : 0.0;
    ll += row[(size_t)i * (DIM + 2) + DIM] + (a * xd - spd);
    s += row[(size_t)i * (DIM + 2) + DIM + 1] + (a - Pd);
    sd += a - Pd;
#pragma unroll
    for (int d = 0; d < DIM; ++d) { const double v = Q[(size_t)i * DIM + d]; sq += v * v; }
  }
  __shared__ double s_red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v4[4] = {ll, s, sd, sq};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    double v = v4[e];
    for (int dl = 32; dl >= 1; dl >>= 1) v += __shfl_down(v, dl);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4];
    for (int e = 0; e < 4; ++e) t[e] = (s_red[0][e] + s_red[1][e]) + (s_red[2][e] + s_red[3][e]);
    scal[LS_LL] = t[0];
    scal[LS_SALL] = t[1];
    scal[LS_SDIAG] = t[2];
    scal[lp_slot] = beta * t[0] - 0.5 * t[3] / sigma - 0.5 * mu0 * mu0 - 0.5 * mus * mus;
  }
}

// Momenta of the L update and their kinetic energy.
__global__ __launch_bounds__(256) void k_ld_momenta(int n, double* __restrict__ Pm, double* __restrict__ scal, uint64_t seed,
                                                    uint32_t sweep) {
  double k = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double p = gg_normal_at(seed, GG_SITE_LD, sweep, (uint32_t)i);
    Pm[i] = p;
    k += p * p;
  }
  __shared__ double s_red[4];
  for (int dl = 32; dl >= 1; dl >>= 1) k += __shfl_down(k, dl);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) scal[LS_K0] = 0.5 * ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
}

// p += f eps (beta (-2 g) - q / sigma);  then (drift) q += eps p.   q = Qsrc on the first call (copy of L).
template <int DIM>
__global__ __launch_bounds__(256) void k_ld_kick_drift(int N, const double* __restrict__ row, const double* Qsrc,
                                                       double* Q, double* __restrict__ Pm,
                                                       const double* __restrict__ scal, double f, double sigma, double beta,
                                                       int drift) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * DIM) return;
  const int i = idx / DIM, d = idx % DIM;
  const double eps = scal[LS_STEP];
  const double q = Qsrc[idx];
  const double grad = beta * (-2.0 * row[(size_t)i * (DIM + 2) + d]) - q / sigma;
  const double p = Pm[idx] + f * eps * grad;
  Pm[idx] = p;
  Q[idx] = drift ? q + eps * p : q;
}

// Kinetic energy at the end, Metropolis test (adjacency.py:386-389 -> hips hmc), step adaptation, acceptance.
__global__ __launch_bounds__(256) void k_ld_accept_L(int n, double* __restrict__ L, const double* __restrict__ Q,
                                                     const double* __restrict__ Pm, double* __restrict__ scal,
                                                     int* __restrict__ acc, uint64_t seed, uint32_t sweep) {
  double k = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { const double p = Pm[i]; k += p * p; }
  __shared__ double s_red[4];
  __shared__ int s_ok;
  for (int dl = 32; dl >= 1; dl >>= 1) k += __shfl_down(k, dl);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double K1 = 0.5 * ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    scal[LS_K1] = K1;
    const double u = gg_uniform_at(seed, GG_SITE_LD, sweep, (uint32_t)n);
    const int ok = gg_log(u) < (scal[LS_LP1] - scal[LS_LP0]) + (scal[LS_K0] - K1) ? 1 : 0;
    double step = scal[LS_STEP], rate = scal[LS_RATE];
    gg_hmc_adapt(ok, &step, &rate);
    scal[LS_STEP] = step;
    scal[LS_RATE] = rate;
    s_ok = ok;
    if (ok) atomicOr(acc, 1);
  }
  __syncthreads();
  if (s_ok)
    for (int i = threadIdx.x; i < n; i += 256) L[i] = Q[i];
}

// Scalar leapfrog bookkeeping of the mu_0 update (one thread): phase 0 = start (draw p, half kick, drift),
// phase 1 = middle step (full kick, drift), phase 2 = last (half kick, Metropolis test).
__global__ void k_ld_mu0_step(double* __restrict__ scal, int* __restrict__ acc, int phase, double beta, uint64_t seed,
                              uint32_t sweep, uint32_t base) {
  if (threadIdx.x || blockIdx.x) return;
  const double eps = 0.005;
  if (phase == 0) {
    const double p = gg_normal_at(seed, GG_SITE_LD, sweep, base + 1u);
    scal[LS_K0] = 0.5 * p * p;
    scal[LS_QMU0] = scal[LS_MU0];
    const double g = beta * scal[LS_SALL] - scal[LS_QMU0];
    const double p1 = p + 0.5 * eps * g;
    scal[LS_P] = p1;
    scal[LS_QMU0] += eps * p1;
  } else {
    const double g = beta * scal[LS_SALL] - scal[LS_QMU0];
    const double p1 = scal[LS_P] + (phase == 1 ? 1.0 : 0.5) * eps * g;
    scal[LS_P] = p1;
    if (phase == 1) scal[LS_QMU0] += eps * p1;
    else {
      const double u = gg_uniform_at(seed, GG_SITE_LD, sweep, base + 2u);
      if (gg_log(u) < (scal[LS_LP1] - scal[LS_LP0]) + (scal[LS_K0] - 0.5 * p1 * p1)) { scal[LS_MU0] = scal[LS_QMU0]; atomicOr(acc, 2); }
    }
  }
}

// The whole mu_self update in one workgroup: only the diagonal depends on mu_self (adjacency.py:404-421); the
// off-diagonal part of log p is the constant C = beta ll_offdiag + prior terms not involving mu_self.
__global__ __launch_bounds__(256) void k_ld_muself(int N, const uint8_t* __restrict__ dA, double* __restrict__ scal,
                                                   int* __restrict__ acc, double beta, uint64_t seed, uint32_t sweep, uint32_t base) {
  __shared__ double s_n1;
  double n1 = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) n1 += dA[i] ? 1.0 : 0.0;
  __shared__ double s_red[4];
  for (int dl = 32; dl >= 1; dl >>= 1) n1 += __shfl_down(n1, dl);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = n1;
  __syncthreads();
  if (threadIdx.x == 0) {
    s_n1 = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    const double mu0 = scal[LS_MU0], ones = s_n1, Nd = (double)N;
    // diagonal log-likelihood and its derivative as functions of q = mu_self
    auto lp_of = [&](double q, double* g) {
      const double x = mu0 + q;
      const double P = gg_sigmoid(x);
      *g = beta * (ones - Nd * P) - q;
      return beta * (ones * x - Nd * gg_softplus(x)) - 0.5 * q * q;
    };
    const double eps = 0.005;
    double q = scal[LS_MUSELF], g;
    double p = gg_normal_at(seed, GG_SITE_LD, sweep, base + 3u);
    const double K0 = 0.5 * p * p;
    const double lp0 = lp_of(q, &g);
    p += 0.5 * eps * g;
    double lp1 = lp0;
    for (int s = 0; s < LD_NSTEPS; ++s) {
      q += eps * p;
      lp1 = lp_of(q, &g);
      p += ((s < LD_NSTEPS - 1) ? 1.0 : 0.5) * eps * g;
    }
    const double u = gg_uniform_at(seed, GG_SITE_LD, sweep, base + 4u);
    if (gg_log(u) < (lp1 - lp0) + (K0 - 0.5 * p * p)) { scal[LS_MUSELF] = q; atomicOr(acc, 4); }
  }
}

// ------------------------------------------------------------------ host
template <typename T>
static int lmalloc(T** p, size_t n) {
  GG_HIP(hipMalloc((void**)p, n * sizeof(T)));
  GG_HIP(hipMemset(*p, 0, n * sizeof(T)));
  return GG_OK;
}

int gg_latent_create(gg_latent** out, const gg_config* cfg) {
  if (cfg->N <= 0 || cfg->D < 1 || cfg->D > LD_MAXDIM)
    return gg_fail(GG_ERR_UNSUPPORTED, "latent-distance model: need N > 0 and 1 <= dim <= %d (got N=%d dim=%d)", LD_MAXDIM, cfg->N, cfg->D);
  if (!(cfg->sigma_F > 0.0)) return gg_fail(GG_ERR_INVALID, "latent-distance model: sigma must be positive");
  gg_latent* e = new gg_latent();
  e->cfg = *cfg;
  e->N = cfg->N; e->dim = cfg->D; e->Np = gg_round_up(cfg->N, 64);
  e->sigma = cfg->sigma_F;
  e->beta = 1.0;
  int rc;
  GG_HIP(hipStreamCreate(&e->stream));
  GG_HIP(hipEventCreate(&e->e0));
  GG_HIP(hipEventCreate(&e->e1));
  const size_t n = (size_t)e->N * e->dim;
  if ((rc = lmalloc(&e->L, n)) || (rc = lmalloc(&e->Q, n)) || (rc = lmalloc(&e->Pm, n)) ||
      (rc = lmalloc(&e->row, (size_t)e->N * (e->dim + 2))) || (rc = lmalloc(&e->scal, (size_t)LS_COUNT)) || (rc = lmalloc(&e->acc, 1)))
    return rc;
  double init[LS_COUNT] = {0};
  init[LS_STEP] = 0.01;    // adjacency.py:205-206
  init[LS_RATE] = 0.9;
  GG_HIP(hipMemcpy(e->scal, init, sizeof(init), hipMemcpyHostToDevice));
  *out = e;
  return GG_OK;
}

int gg_latent_destroy(gg_latent* e) {
  if (!e) return GG_OK;
  void* ptrs[] = {e->L, e->Q, e->Pm, e->row, e->scal, e->acc};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (e->e0) (void)hipEventDestroy(e->e0);
  if (e->e1) (void)hipEventDestroy(e->e1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return GG_OK;
}

// state ids: GG_ST_F = L (N x dim), GG_ST_MU0 = mu_0, GG_ST_LD_AUX = {mu_self, step size, running acceptance rate}
int gg_latent_set_state(gg_latent* e, int id, const void* buf, size_t nbytes) {
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    e->sweep = *(const uint32_t*)buf;
    return GG_OK;
  }
  void* p = nullptr; size_t need = 0;
  if (id == GG_ST_F) { p = e->L; need = (size_t)e->N * e->dim * 8; }
  else if (id == GG_ST_MU0) { p = e->scal + LS_MU0; need = 8; }
  else if (id == GG_ST_LD_AUX) { p = e->scal + LS_MUSELF; need = 24; }
  else return gg_fail(GG_ERR_INVALID, "state id %d not valid for the latent-distance model", id);
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  GG_HIP(hipMemcpyAsync(p, buf, nbytes, hipMemcpyHostToDevice, e->stream));
  GG_HIP(hipStreamSynchronize(e->stream));
  return GG_OK;
}

int gg_latent_get_state(gg_latent* e, int id, void* buf, size_t nbytes) {
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    *(uint32_t*)buf = e->sweep;
    return GG_OK;
  }
  const void* p = nullptr; size_t need = 0;
  if (id == GG_ST_F) { p = e->L; need = (size_t)e->N * e->dim * 8; }
  else if (id == GG_ST_MU0) { p = e->scal + LS_MU0; need = 8; }
  else if (id == GG_ST_LD_AUX) { p = e->scal + LS_MUSELF; need = 24; }
  else return gg_fail(GG_ERR_INVALID, "state id %d not valid for the latent-distance model", id);
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  GG_HIP(hipMemcpyAsync(buf, p, nbytes, hipMemcpyDeviceToHost, e->stream));
  GG_HIP(hipStreamSynchronize(e->stream));
  return GG_OK;
}

#define LD_DISPATCH(CALL)            \
  do {                               \
    switch (e->dim) {                \
      case 1: { CALL(1); } break;    \
      case 2: { CALL(2); } break;    \
      case 3: { CALL(3); } break;    \
      default: { CALL(4); } break;   \
    }                                \
  } while (0)

// pass + reduce at positions Qp with (mu_0, mu_self) taken from the given scalar slots; log p -> scal[lp_slot]
static int ld_eval(gg_latent* e, const gg_data* d, const double* Qp, int mu0_slot, int lp_slot) {
  const int N = e->N;
#define CALL(DV)                                                                                                       \
  k_ld_pass<DV><<<N, 256, 0, e->stream>>>(N, e->Np, d->A, d->AT, Qp, e->scal, mu0_slot, e->row);                        \
  k_ld_reduce<DV><<<1, 256, 0, e->stream>>>(N, e->row, d->dA, Qp, e->scal, mu0_slot, LS_MUSELF, e->sigma, e->beta, lp_slot)
  LD_DISPATCH(CALL);
#undef CALL
  GG_HIP(hipGetLastError());
  return GG_OK;
}

void gg_latent_set_beta(gg_latent* e, double beta) { e->beta = beta; }

// parts: GG_PART_F = latent positions, GG_PART_MU0 = mu_0, GG_PART_LMBDA = mu_self (adjacency.py:360-372)
int gg_latent_resample(gg_latent* e, const gg_data* d, int parts) {
  if (!d) return gg_fail(GG_ERR_INVALID, "null data handle");
  if (d->N != e->N) return gg_fail(GG_ERR_INVALID, "data has N=%d, model has N=%d", d->N, e->N);
  const int N = e->N, n = N * e->dim;
  hipStream_t st = e->stream;
  int rc;
  GG_HIP(hipEventRecord(e->e0, st));
  GG_HIP(hipMemsetAsync(e->acc, 0, sizeof(int), st));
  const uint32_t base = (uint32_t)n;
  if (parts & GG_PART_F) {
    k_ld_momenta<<<1, 256, 0, st>>>(n, e->Pm, e->scal, e->cfg.seed, e->sweep);
    if ((rc = ld_eval(e, d, e->L, LS_MU0, LS_LP0))) return rc;
    const int nb = (n + 255) / 256;
#define CALL(DV) k_ld_kick_drift<DV><<<nb, 256, 0, st>>>(N, e->row, e->L, e->Q, e->Pm, e->scal, 0.5, e->sigma, e->beta, 1)
    LD_DISPATCH(CALL);
#undef CALL
    for (int s = 0; s < LD_NSTEPS; ++s) {
      if ((rc = ld_eval(e, d, e->Q, LS_MU0, LS_LP1))) return rc;
      const double f = (s < LD_NSTEPS - 1) ? 1.0 : 0.5;
      const int drift = (s < LD_NSTEPS - 1) ? 1 : 0;
#define CALL(DV) k_ld_kick_drift<DV><<<nb, 256, 0, st>>>(N, e->row, e->Q, e->Q, e->Pm, e->scal, f, e->sigma, e->beta, drift)
      LD_DISPATCH(CALL);
#undef CALL
    }
    k_ld_accept_L<<<1, 256, 0, st>>>(n, e->L, e->Q, e->Pm, e->scal, e->acc, e->cfg.seed, e->sweep);
    GG_HIP(hipGetLastError());
  }
  if (parts & GG_PART_MU0) {
    if ((rc = ld_eval(e, d, e->L, LS_MU0, LS_LP0))) return rc;
    k_ld_mu0_step<<<1, 64, 0, st>>>(e->scal, e->acc, 0, e->beta, e->cfg.seed, e->sweep, base);
    for (int s = 0; s < LD_NSTEPS; ++s) {
      if ((rc = ld_eval(e, d, e->L, LS_QMU0, LS_LP1))) return rc;
      k_ld_mu0_step<<<1, 64, 0, st>>>(e->scal, e->acc, (s < LD_NSTEPS - 1) ? 1 : 2, e->beta, e->cfg.seed, e->sweep, base);
    }
    GG_HIP(hipGetLastError());
  }
  if (parts & GG_PART_LMBDA) {
    k_ld_muself<<<1, 256, 0, st>>>(N, d->dA, e->scal, e->acc, e->beta, e->cfg.seed, e->sweep, base);
    GG_HIP(hipGetLastError());
  }
  GG_HIP(hipEventRecord(e->e1, st));
  GG_HIP(hipStreamSynchronize(st));
  float ms = 0;
  GG_HIP(hipEventElapsedTime(&ms, e->e0, e->e1));
  e->st.total_ms = ms;
  const int full = GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA;
  if ((parts & full) == full) { e->sweep += 1; e->st.sweeps += 1; }
  int acc = 0;
  GG_HIP(hipMemcpy(&acc, e->acc, sizeof(int), hipMemcpyDeviceToHost));
  e->st.label_changes += (uint64_t)(acc & 1);    // accepted L proposals
  return GG_OK;
}

int gg_latent_log_likelihood(gg_latent* e, const gg_data* d, double* out) {
  if (!d || !out) return gg_fail(GG_ERR_INVALID, "null argument");
  if (d->N != e->N) return gg_fail(GG_ERR_INVALID, "data has N=%d, model has N=%d", d->N, e->N);
  int rc = ld_eval(e, d, e->L, LS_MU0, LS_LP1);
  if (rc) return rc;
  GG_HIP(hipMemcpyAsync(out, e->scal + LS_LL, 8, hipMemcpyDeviceToHost, e->stream));
  GG_HIP(hipStreamSynchronize(e->stream));
  return GG_OK;
}

int gg_latent_stats(gg_latent* e, gg_stats* out) {
  *out = e->st;
  return GG_OK;
}
