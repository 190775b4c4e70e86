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
#include "spec/gg_ldw.h"

#define LD_MAXDIM 8
#define LD_NSTEPS 10

// device scalars
enum { LS_MU0 = 0, LS_MUSELF, LS_STEP, LS_RATE, LS_LP0, LS_LP1, LS_K0, LS_K1, LS_LL, LS_SALL, LS_SDIAG, LS_Q, LS_P, LS_ACC,
       LS_QMU0, LS_QMUSELF,
       // latent-distance Gaussian weights (wmode): LS_MU0 holds b; {eta, sigma, mu_self, sigma_self} are GG_ST_LD_AUX
       LS_ETA, LS_SIG, LS_WMUS, LS_WSIGS, LS_N, LS_SX, LS_SXX, LS_SS, LS_SQ, LS_COUNT };

struct gg_latent {
  gg_config cfg;
  int N, dim, Np;
  double sigma;
  double *L, *Q, *Pm;      // positions, proposal, momenta  [N][dim]
  double* Q2;              // second proposal buffer (the pass kernel drifts into the other one)
  double* row;             // per node: dim gradient sums, ll, s  [N][dim + 2]
  double* scal;            // LS_COUNT scalars
  int* acc;                // bit mask of accepted proposals of the last resample
  double* kpart;           // k_ld_momenta: per-workgroup partial kinetic energies
  unsigned* tick;          // k_ld_momenta: arrival ticket
  hipStream_t stream;
  hipEvent_t e0, e1;
  uint32_t sweep;
  int parts_done;          // parts already drawn under the current sweep index (step-wise calls, see gg_resample_part)
  double beta;
  gg_stats st;
  int wmode;               // 1: GG_LATENT_DIST_W (weights.py:613-903)
  double prior[4];         // wmode: GG_ST_NIW      = {unused, lmbda_0, nu_0, unused}     (weights.py:638)
  double prior_self[4];    // wmode: GG_ST_NIW_SELF = {mu_0, sigma_0, nu_0, kappa_0}      (weights.py:641-644)
};

// One pass: for node i, over all j != i,
//   x = -|q_i - q_j|^2 + mu_0,  P = sigmoid(x):
//   ll_i  = sum_j A_ij x - softplus(x)            (the directed terms of row i; LL = 1 only)
//   s_i   = sum_j A_ij - P
//   g_i   = sum_j (A_ij + A_ji - 2 P) (q_i - q_j) (times -2 later: d x / d q_i = -2 (q_i - q_j), P symmetric)
// The log probability is needed only at the two ends of a trajectory: the passes in between (LL = 0) skip the
// logarithm (the pass is bound by its fp64 exp + log) and, in the L update, the reduction too - with kick != 0 the
// workgroup finishes the leapfrog step of its own node (p_i += f eps grad_i, drift into the other proposal buffer).
// GRAD = 0 (the mu_0 update: only ll and s are wanted): A^T is not read and the position gradient not formed.
template <int DIM, int LL, int GRAD>
__global__ __launch_bounds__(256) void k_ld_pass(int N, int Np, const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                                 const double* __restrict__ Q, const double* __restrict__ scal, int mu0_slot,
                                                 double* __restrict__ row, int kick, double* __restrict__ Qdst,
                                                 double* __restrict__ Pm, double f, int drift, double sigma, double beta) {
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
    const double a1 = A[(size_t)i * Np + j] ? 1.0 : 0.0;
    double P, sp;
    if (LL) {
      gg_sigmoid_softplus(x, &P, &sp);
      ll += a1 * x - sp;
    } else {
      P = gg_sigmoid_only(x);
    }
    s += a1 - P;
    if (GRAD) {
      const double a2 = AT[(size_t)i * Np + j] ? 1.0 : 0.0;
      const double c = a1 + a2 - 2.0 * P;
#pragma unroll
      for (int d = 0; d < DIM; ++d) g[d] += c * t[d];
    }
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
  if (threadIdx.x < DIM + 2) {
    const int e = threadIdx.x;
    const double tot = (s_red[0][e] + s_red[1][e]) + (s_red[2][e] + s_red[3][e]);
    row[(size_t)i * (DIM + 2) + e] = tot;
    if (kick && e < DIM) {
      const int idx = i * DIM + e;
      const double eps = scal[LS_STEP];
      const double q = Q[idx];
      const double grad = beta * (-2.0 * tot) - q / sigma;
      const double p = Pm[idx] + f * eps * grad;
      Pm[idx] = p;
      if (drift) Qdst[idx] = q + eps * p;
    }
  }
}

// Scalar leapfrog bookkeeping of the mu_0 update (one thread): phase 0 = start (draw p, half kick, drift),
// phase 1 = middle step (full kick, drift), phase 2 = last (half kick, Metropolis test).
// Runs on the one thread that has just finished the reduction of the pass (k_ld_reduce).
__device__ __forceinline__ void ld_mu0_step(double* scal, int* acc, int phase, double beta, uint64_t seed, uint32_t sweep,
                                            uint32_t base) {
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

// Sums over nodes (fixed order), the diagonal terms, the prior: ll, d ll / d mu_0, d ll / d mu_self, and the
// log probability beta ll + log prior into scal[lp_slot].  One workgroup.
template <int DIM>
__global__ __launch_bounds__(256) void k_ld_reduce(int N, const double* __restrict__ row, const uint8_t* __restrict__ dA,
                                                   const double* __restrict__ Q, double* scal, int mu0_slot,
                                                   int muself_slot, double sigma, double beta, int lp_slot, int mu0_phase,
                                                   int* acc, uint64_t seed, uint32_t sweep, uint32_t base) {
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
    if (mu0_phase >= 0) ld_mu0_step(scal, acc, mu0_phase, beta, seed, sweep, base);   // the mu_0 update's leapfrog step, fused
  }
}

// Momenta of the L update and their kinetic energy.  Any number of workgroups: each writes the partial sum of its 256
// entries to kpart[blockIdx.x]; the last one to finish (ticket in *tick, reset for the next launch) adds the partials in
// index order, so K0 does not depend on the schedule.
__global__ __launch_bounds__(256) void k_ld_momenta(int n, double* __restrict__ Pm, double* __restrict__ scal, uint64_t seed,
                                                    uint32_t sweep, uint32_t site, double* __restrict__ kpart,
                                                    unsigned* __restrict__ tick) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double k = 0.0;
  if (i < n) {
    const double p = gg_normal_at(seed, site, sweep, (uint32_t)i);
    Pm[i] = p;
    k = p * p;
  }
  __shared__ double s_red[4];
  __shared__ int s_last;
  for (int dl = 32; dl >= 1; dl >>= 1) k += __shfl_down(k, dl);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(kpart + blockIdx.x, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(tick, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double t = 0.0;
  for (int w = threadIdx.x; w < (int)gridDim.x; w += 256) t += __hip_atomic_load(kpart + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int dl = 32; dl >= 1; dl >>= 1) t += __shfl_down(t, dl);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    scal[LS_K0] = 0.5 * ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    *tick = 0u;
  }
}

// Kinetic energy at the end, Metropolis test (adjacency.py:386-389 -> hips hmc), step adaptation, acceptance.
__global__ __launch_bounds__(256) void k_ld_accept_L(int n, double* __restrict__ L, const double* __restrict__ Q,
                                                     const double* __restrict__ Pm, double* __restrict__ scal,
                                                     int* __restrict__ acc, uint64_t seed, uint32_t sweep, uint32_t site,
                                                     int adapt) {
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
    const double u = gg_uniform_at(seed, site, sweep, (uint32_t)n);
    const int ok = gg_log(u) < (scal[LS_LP1] - scal[LS_LP0]) + (scal[LS_K0] - K1) ? 1 : 0;
    if (adapt) {
      double step = scal[LS_STEP], rate = scal[LS_RATE];
      gg_hmc_adapt(ok, &step, &rate);
      scal[LS_STEP] = step;
      scal[LS_RATE] = rate;
    }
    s_ok = ok;
    if (ok) { atomicOr(acc, 1); acc[1] += 1; }      // acc[1]: accepted L proposals since the last read-back
  }
  __syncthreads();
  if (s_ok)
    for (int i = threadIdx.x; i < n; i += 256) L[i] = Q[i];
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

// ------------------------------------------------------------------ latent-distance Gaussian weights (wmode)
// One pass of the weight model (weights.py:703-733): for node i over the edges of row i of W and of W^T,
//   x = w_ij + |q_i - q_j|^2,  r = x - b:
//   g_i  = sum_j (r_ij [A_ij] + r_ji [A_ji]) (q_i - q_j)                          (times -2 / sigma: d log p / d q_i)
//   STATS: ss_i = sum_j r_ij^2,  n_i = #edges,  sx_i = sum x,  sxx_i = sum x^2   (the directed terms of row i)
// Non-edges carry GG_NO_EDGE (diagonal and padding too), so the adjacency is not read: HBM-bound on 8 N^2 bytes.
// kick != 0: the workgroup finishes the leapfrog step of its own node, p_i += f eps grad_i and (drift) the new position
// q_i + eps p_i into Qdst (a second buffer: other workgroups are still reading Qsrc).  The log probability is only
// needed at the two ends of the trajectory, so the passes in between (STATS = 0) are followed by no reduction at all.
// NTL: non-temporal loads once W and W^T outgrow the Infinity Cache.
template <int DIM, int STATS, int NTL>
__global__ __launch_bounds__(256) void k_ldw_pass(int N, int Np, const int32_t* __restrict__ W, const int32_t* __restrict__ WT,
                                                  double inv_q, const double* __restrict__ Q, const double* __restrict__ scal,
                                                  double* __restrict__ row, int kick, double* __restrict__ Qdst,
                                                  double* __restrict__ Pm, double f, int drift, double beta) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int i = blockIdx.x;
  const double b = scal[LS_MU0];
  double qi[DIM], g[DIM], ss = 0.0, n = 0.0, sx = 0.0, sxx = 0.0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) { qi[d] = Q[(size_t)i * DIM + d]; g[d] = 0.0; }
  const v4i* w4 = (const v4i*)(W + (size_t)i * Np);
  const v4i* t4 = (const v4i*)(WT + (size_t)i * Np);
  auto pair = [&](int32_t w1, int32_t w2, int j) {
    if (w1 == GG_NO_EDGE && w2 == GG_NO_EDGE) return;
    double t[DIM], d2 = 0.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) { t[d] = qi[d] - Q[(size_t)j * DIM + d]; d2 += t[d] * t[d]; }
    double cf = 0.0;
    if (w1 != GG_NO_EDGE) {
      const double x = (double)w1 * inv_q + d2, r = x - b;
      if (STATS) { ss += r * r; n += 1.0; sx += x; sxx += x * x; }
      cf += r;
    }
    if (w2 != GG_NO_EDGE) cf += ((double)w2 * inv_q + d2) - b;
#pragma unroll
    for (int d = 0; d < DIM; ++d) g[d] += cf * t[d];
  };
  // four 16-byte loads of each matrix in flight per lane (the compiler issues loads right before use otherwise)
  const int n4 = (N + 3) >> 2;
  for (int j0 = threadIdx.x; j0 < n4; j0 += 1024) {
    v4i a[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j4 = j0 + u * 256;
      const v4i none = {GG_NO_EDGE, GG_NO_EDGE, GG_NO_EDGE, GG_NO_EDGE};
      if (NTL) {
        a[u] = j4 < n4 ? __builtin_nontemporal_load(w4 + j4) : none;
        c[u] = j4 < n4 ? __builtin_nontemporal_load(t4 + j4) : none;
      } else {
        a[u] = j4 < n4 ? w4[j4] : none;
        c[u] = j4 < n4 ? t4[j4] : none;
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) pair(a[u][k], c[u][k], (j0 + u * 256) * 4 + k);
  }
  constexpr int NE = STATS ? DIM + 4 : DIM;
  __shared__ double s_red[4][DIM + 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    double v = e < DIM ? g[e] : (e == DIM ? ss : (e == DIM + 1 ? n : (e == DIM + 2 ? sx : sxx)));
    for (int dl = 32; dl >= 1; dl >>= 1) v += __shfl_down(v, dl);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < NE) {
    const int e = threadIdx.x;
    const double tot = (s_red[0][e] + s_red[1][e]) + (s_red[2][e] + s_red[3][e]);
    row[(size_t)i * (DIM + 4) + e] = tot;
    if (kick && e < DIM) {
      const int idx = i * DIM + e;
      const double eps = scal[LS_STEP];
      const double q = Q[idx];
      const double grad = beta * (-2.0 * tot) / scal[LS_SIG] - q / scal[LS_ETA];
      const double p = Pm[idx] + f * eps * grad;
      Pm[idx] = p;
      if (drift) Qdst[idx] = q + eps * p;
    }
  }
}

// Sums over nodes (fixed order) and log p(L | b, A, W) = beta (-ss / 2 sigma) - sum q^2 / 2 eta - b^2 / 2 into scal[lp_slot].
template <int DIM>
__global__ __launch_bounds__(1024) void k_ldw_reduce(int N, const double* __restrict__ row, const double* __restrict__ Q,
                                                     double* __restrict__ scal, double beta, int lp_slot) {
  double v5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < N; i += 1024) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v5[e] += row[(size_t)i * (DIM + 4) + DIM + e];
#pragma unroll
    for (int d = 0; d < DIM; ++d) { const double v = Q[(size_t)i * DIM + d]; v5[4] += v * v; }
  }
  __shared__ double s_red[16][5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 5; ++e) {
    double v = v5[e];
    for (int dl = 32; dl >= 1; dl >>= 1) v += __shfl_down(v, dl);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[5];
    for (int e = 0; e < 5; ++e) {
      t[e] = 0.0;
      for (int w = 0; w < 16; ++w) t[e] += s_red[w][e];
    }
    scal[LS_SS] = t[0]; scal[LS_N] = t[1]; scal[LS_SX] = t[2]; scal[LS_SXX] = t[3]; scal[LS_SQ] = t[4];
    const double b = scal[LS_MU0];
    scal[lp_slot] = beta * (-0.5 * t[0] / scal[LS_SIG]) - 0.5 * t[4] / scal[LS_ETA] - 0.5 * b * b;
  }
}

struct gg_ldw_hyp { double lmbda_0, nu_0, smu_0, ssigma_0, snu_0, skappa_0, n_self, s1_self, s2_self; };

// The conjugate draws after the L update (weights.py:767-773), from the sums of a pass at the current L: b (:833-856),
// sigma from the residuals against the new b (:859-863; sum r^2 = sxx - 2 b sx + n b^2), the self Gaussian (:865-868;
// its statistics do not depend on the state and come exact from the data handle), eta (:870-886).  One thread.
__global__ void k_ldw_params(int n_entries, double* __restrict__ scal, gg_ldw_hyp h, double beta, uint64_t seed, uint32_t sweep,
                             int parts) {
  if (threadIdx.x || blockIdx.x) return;
  gg_stream st;
  const double n = scal[LS_N], sx = scal[LS_SX], sxx = scal[LS_SXX];
  if (parts & GG_PART_MU0) {
    gg_stream_init(&st, seed, GG_SITE_LDW, sweep, 0u);
    scal[LS_MU0] = gg_ldw_b_draw(&st, beta * n, beta * sx, scal[LS_SIG]);
  }
  if (parts & GG_PART_PARAMS) {
    const double b = scal[LS_MU0];
    const double ss = sxx - 2.0 * b * sx + n * b * b;
    gg_stream_init(&st, seed, GG_SITE_LDW, sweep, 1u);
    scal[LS_SIG] = gg_ldw_sigma_draw(&st, h.lmbda_0, h.nu_0, beta * n, beta * ss);
    gg_niw hs;
    hs.mu_0[0] = h.smu_0; hs.sigma_0[0] = h.ssigma_0; hs.nu_0 = h.snu_0; hs.kappa_0 = h.skappa_0;
    double s1 = beta * h.s1_self, s2 = beta * h.s2_self, mu, sig;
    gg_stream_init(&st, seed, GG_SITE_LDW, sweep, 2u);
    gg_niw_posterior_draw(&st, &hs, 1, beta * h.n_self, &s1, &s2, &mu, &sig);
    scal[LS_WMUS] = mu;
    scal[LS_WSIGS] = sig;
  }
  if (parts & GG_PART_LMBDA) {
    gg_stream_init(&st, seed, GG_SITE_LDW, sweep, 3u);
    scal[LS_ETA] = gg_ldw_eta_draw(&st, (double)n_entries, scal[LS_SQ]);
  }
}

// ------------------------------------------------------------------ host
template <typename T>
static int lmalloc(T** p, size_t n) {
  GG_HIP(hipMalloc((void**)p, n * sizeof(T)));
  GG_HIP(hipMemset(*p, 0, n * sizeof(T)));
  return GG_OK;
}

static int ld_init(gg_latent* e, const gg_config* cfg, int wmode);
int gg_latent_destroy(gg_latent* e);

int gg_latent_create(gg_latent** out, const gg_config* cfg) {
  if (cfg->N <= 0 || cfg->D < 1 || cfg->D > LD_MAXDIM)
    return gg_fail(GG_ERR_UNSUPPORTED, "latent-distance model: need N > 0 and 1 <= dim <= %d (got N=%d dim=%d)", LD_MAXDIM, cfg->N, cfg->D);
  const int wmode = cfg->kind == GG_LATENT_DIST_W;
  if (!wmode && !(cfg->sigma_F > 0.0)) return gg_fail(GG_ERR_INVALID, "latent-distance model: sigma must be positive");
  if (wmode && cfg->B != 1)
    return gg_fail(GG_ERR_UNSUPPORTED, "latent-distance weights: B = 1 only, as in the reference (weights.py:711), got B=%d", cfg->B);
  gg_latent* e = new gg_latent();   // (value-initialised: every pointer starts null)
  const int rc0 = ld_init(e, cfg, wmode);
  if (rc0) { gg_latent_destroy(e); return rc0; }
  *out = e;
  return GG_OK;
}

static int ld_init(gg_latent* e, const gg_config* cfg, int wmode) {
  e->cfg = *cfg;
  e->wmode = wmode;
  e->N = cfg->N; e->dim = cfg->D; e->Np = gg_round_up(cfg->N, 64);
  e->sigma = cfg->sigma_F;
  e->beta = 1.0;
  int rc;
  GG_HIP(hipStreamCreate(&e->stream));
  GG_HIP(hipEventCreate(&e->e0));
  GG_HIP(hipEventCreate(&e->e1));
  const size_t n = (size_t)e->N * e->dim;
  if ((rc = lmalloc(&e->L, n)) || (rc = lmalloc(&e->Q, n)) || (rc = lmalloc(&e->Q2, n)) || (rc = lmalloc(&e->Pm, n)) ||
      (rc = lmalloc(&e->row, (size_t)e->N * (e->dim + 4))) || (rc = lmalloc(&e->scal, (size_t)LS_COUNT)) || (rc = lmalloc(&e->acc, 2)) ||
      (rc = lmalloc(&e->kpart, (n + 255) / 256)) || (rc = lmalloc(&e->tick, 1)))
    return rc;
  double init[LS_COUNT] = {0};
  init[LS_STEP] = wmode ? 0.005 : 0.01;    // adjacency.py:205-206; weights.py:785 (fixed)
  init[LS_RATE] = 0.9;
  init[LS_MU0] = wmode ? 0.5 : 0.0;        // weights.py:621 (b)
  init[LS_ETA] = 0.01; init[LS_SIG] = 1.0; init[LS_WSIGS] = 1.0;
  const double pr[4] = {0.0, 1.0, 3.0, 1.0};   // weights.py:632-644 defaults for B = 1
  memcpy(e->prior, pr, sizeof(pr));
  memcpy(e->prior_self, pr, sizeof(pr));
  GG_HIP(hipMemcpy(e->scal, init, sizeof(init), hipMemcpyHostToDevice));
  return GG_OK;
}

int gg_latent_destroy(gg_latent* e) {
  if (!e) return GG_OK;
  void* ptrs[] = {e->L, e->Q, e->Q2, e->Pm, e->row, e->scal, e->acc, e->kpart, e->tick};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (e->e0) (void)hipEventDestroy(e->e0);
  if (e->e1) (void)hipEventDestroy(e->e1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return GG_OK;
}

// state ids: GG_ST_F = L (N x dim), GG_ST_MU0 = mu_0 (wmode: b),
// GG_ST_LD_AUX = {mu_self, step size, running acceptance rate} (wmode: {eta, sigma, mu_self, sigma_self}),
// wmode only: GG_ST_NIW = {-, lmbda_0, nu_0, -}, GG_ST_NIW_SELF = {mu_0, sigma_0, nu_0, kappa_0} (host-side hyper-parameters)
static int ld_state_slot(gg_latent* e, int id, void** dev, void** host, size_t* need) {
  *dev = nullptr; *host = nullptr;
  if (id == GG_ST_F) { *dev = e->L; *need = (size_t)e->N * e->dim * 8; }
  else if (id == GG_ST_MU0) { *dev = e->scal + LS_MU0; *need = 8; }
  else if (id == GG_ST_LD_AUX && !e->wmode) { *dev = e->scal + LS_MUSELF; *need = 24; }
  else if (id == GG_ST_LD_AUX && e->wmode) { *dev = e->scal + LS_ETA; *need = 32; }
  else if (id == GG_ST_NIW && e->wmode) { *host = e->prior; *need = 32; }
  else if (id == GG_ST_NIW_SELF && e->wmode) { *host = e->prior_self; *need = 32; }
  else return gg_fail(GG_ERR_INVALID, "state id %d not valid for the latent-distance model", id);
  return GG_OK;
}

int gg_latent_set_state(gg_latent* e, int id, const void* buf, size_t nbytes) {
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    e->sweep = *(const uint32_t*)buf;
    e->parts_done = 0;
    return GG_OK;
  }
  void *p, *hp; size_t need = 0;
  int rc = ld_state_slot(e, id, &p, &hp, &need);
  if (rc) return rc;
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  {
    const long long bad = gg_first_nonfinite((const double*)buf, nbytes / sizeof(double));   // (all of this model's state is doubles)
    if (bad >= 0) return gg_fail(GG_ERR_INVALID, "state id %d: non-finite value at index %lld", id, bad);
  }
  if (hp) { memcpy(hp, buf, nbytes); return GG_OK; }
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
  void *p, *hp; size_t need = 0;
  int rc = ld_state_slot(e, id, &p, &hp, &need);
  if (rc) return rc;
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  if (hp) { memcpy(buf, hp, nbytes); return GG_OK; }
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
      case 4: { CALL(4); } break;    \
      case 5: { CALL(5); } break;    \
      case 6: { CALL(6); } break;    \
      case 7: { CALL(7); } break;    \
      default: { CALL(8); } break;   \
    }                                \
  } while (0)

static int ld_check_data(gg_latent* e, const gg_data* d) {
  if (!d) return gg_fail(GG_ERR_INVALID, "null data handle");
  if (d->N != e->N) return gg_fail(GG_ERR_INVALID, "data has N=%d, model has N=%d", d->N, e->N);
  if (d->n_missing) return gg_fail(GG_ERR_UNSUPPORTED, "the network has %lld unobserved entries: missing data is defined for the eigenmodels only", d->n_missing);
  if (e->wmode && (!d->has_w || d->B != 1))
    return gg_fail(GG_ERR_INVALID, "latent-distance weights: the data handle must carry weights with B = 1");
  if (d->n0 != 0 || d->n1 != d->N) return gg_fail(GG_ERR_UNSUPPORTED, "latent-distance model: column-sharded data not supported");
  return GG_OK;
}

// weight model: one pass at positions Qsrc (stats: also the sums the reduction needs; kick: the fused leapfrog step)
static int ldw_pass(gg_latent* e, const gg_data* d, const double* Qsrc, int stats, int kick, double* Qdst, double f, int drift) {
  const int N = e->N;
  const double inv_q = ldexp(1.0, -d->shift);
  const int ntl = 8.0 * N * e->Np > gg_nt_threshold();
#define ARGS N, e->Np, d->Wq, d->WqT, inv_q, Qsrc, e->scal, e->row, kick, Qdst, e->Pm, f, drift, e->beta
#define CALL(DV)                                                                         \
  if (stats) { if (ntl) k_ldw_pass<DV, 1, 1><<<N, 256, 0, e->stream>>>(ARGS); else k_ldw_pass<DV, 1, 0><<<N, 256, 0, e->stream>>>(ARGS); } \
  else { if (ntl) k_ldw_pass<DV, 0, 1><<<N, 256, 0, e->stream>>>(ARGS); else k_ldw_pass<DV, 0, 0><<<N, 256, 0, e->stream>>>(ARGS); }
  LD_DISPATCH(CALL);
#undef CALL
#undef ARGS
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int ldw_reduce(gg_latent* e, const double* Qsrc, int lp_slot) {
#define CALL(DV) k_ldw_reduce<DV><<<1, 1024, 0, e->stream>>>(e->N, e->row, Qsrc, e->scal, e->beta, lp_slot)
  LD_DISPATCH(CALL);
#undef CALL
  GG_HIP(hipGetLastError());
  return GG_OK;
}

// adjacency model: one pass at positions Qsrc (ll: with the log-likelihood terms; kick: the fused leapfrog step)
static int ld_pass(gg_latent* e, const gg_data* d, const double* Qsrc, int mu0_slot, int ll, int kick, double* Qdst, double f,
                   int drift, int grad = 1) {
  const int N = e->N;
#define ARGS N, e->Np, d->A, d->AT, Qsrc, e->scal, mu0_slot, e->row, kick, Qdst, e->Pm, f, drift, e->sigma, e->beta
#define CALL(DV)                                                                                                       \
  if (!grad) { if (ll) k_ld_pass<DV, 1, 0><<<N, 256, 0, e->stream>>>(ARGS); else k_ld_pass<DV, 0, 0><<<N, 256, 0, e->stream>>>(ARGS); } \
  else { if (ll) k_ld_pass<DV, 1, 1><<<N, 256, 0, e->stream>>>(ARGS); else k_ld_pass<DV, 0, 1><<<N, 256, 0, e->stream>>>(ARGS); }
  LD_DISPATCH(CALL);
#undef CALL
#undef ARGS
  GG_HIP(hipGetLastError());
  return GG_OK;
}

// pass + reduce at positions Qp with (mu_0, mu_self) taken from the given scalar slots; log p -> scal[lp_slot]
// (need_ll = 0: only the gradient sums are wanted - the logarithm is skipped and scal[lp_slot], LS_LL are not meaningful)
static int ld_eval(gg_latent* e, const gg_data* d, const double* Qp, int mu0_slot, int lp_slot, int need_ll = 1, int grad = 1,
                   int mu0_phase = -1) {
  const int N = e->N;
  if (e->wmode) {
    int rc = ldw_pass(e, d, Qp, 1, 0, nullptr, 0.0, 0);
    return rc ? rc : ldw_reduce(e, Qp, lp_slot);
  }
  int rc = ld_pass(e, d, Qp, mu0_slot, need_ll, 0, nullptr, 0.0, 0, grad);
  if (rc) return rc;
#define CALL(DV) k_ld_reduce<DV><<<1, 256, 0, e->stream>>>(N, e->row, d->dA, Qp, e->scal, mu0_slot, LS_MUSELF, e->sigma, e->beta, lp_slot, \
                                                           mu0_phase, e->acc, e->cfg.seed, e->sweep, (uint32_t)(e->N * e->dim))
  LD_DISPATCH(CALL);
#undef CALL
  GG_HIP(hipGetLastError());
  return GG_OK;
}

void gg_latent_set_beta(gg_latent* e, double beta) { e->beta = beta; }

// The HMC transition on L shared by both models: momenta, half kick + drift, n_steps evaluations, Metropolis test.
static int ld_hmc_L(gg_latent* e, const gg_data* d) {
  const int N = e->N, n = N * e->dim;
  hipStream_t st = e->stream;
  const uint32_t site = e->wmode ? GG_SITE_LDW_HMC : GG_SITE_LD;
  int rc;
  k_ld_momenta<<<(n + 255) / 256, 256, 0, st>>>(n, e->Pm, e->scal, e->cfg.seed, e->sweep, site, e->kpart, e->tick);
  if (e->wmode) {
    // trajectory: L -> Q -> Q2 -> Q -> ...; each pass kicks the momenta and drifts into the other buffer; log p at the two ends
    const double* src = e->L;
    double* dst = e->Q;
    if ((rc = ldw_pass(e, d, src, 1, 1, dst, 0.5, 1)) || (rc = ldw_reduce(e, src, LS_LP0))) return rc;
    for (int s = 0; s < LD_NSTEPS; ++s) {
      const int last = s == LD_NSTEPS - 1;
      src = dst;
      dst = (src == e->Q) ? e->Q2 : e->Q;
      if ((rc = ldw_pass(e, d, src, last, 1, dst, last ? 0.5 : 1.0, last ? 0 : 1))) return rc;
    }
    if ((rc = ldw_reduce(e, src, LS_LP1))) return rc;
    k_ld_accept_L<<<1, 256, 0, st>>>(n, e->L, src, e->Pm, e->scal, e->acc, e->cfg.seed, e->sweep, site, 0);
    GG_HIP(hipGetLastError());
    return GG_OK;
  }
  {
    // same trajectory bookkeeping as above: L -> Q -> Q2 -> Q -> ...; reductions (log p) at the two ends only
    const double* src = e->L;
    double* dst = e->Q;
    if ((rc = ld_pass(e, d, src, LS_MU0, 1, 1, dst, 0.5, 1))) return rc;
#define RED(DV, LPS) k_ld_reduce<DV><<<1, 256, 0, st>>>(N, e->row, d->dA, src, e->scal, LS_MU0, LS_MUSELF, e->sigma, e->beta, LPS, -1, nullptr, 0, 0, 0)
#define CALL(DV) RED(DV, LS_LP0)
    LD_DISPATCH(CALL);
#undef CALL
    for (int s = 0; s < LD_NSTEPS; ++s) {
      const int last = s == LD_NSTEPS - 1;
      src = dst;
      dst = (src == e->Q) ? e->Q2 : e->Q;
      if ((rc = ld_pass(e, d, src, LS_MU0, last, 1, dst, last ? 0.5 : 1.0, last ? 0 : 1))) return rc;
    }
#define CALL(DV) RED(DV, LS_LP1)
    LD_DISPATCH(CALL);
#undef CALL
#undef RED
    k_ld_accept_L<<<1, 256, 0, st>>>(n, e->L, src, e->Pm, e->scal, e->acc, e->cfg.seed, e->sweep, site, 1);
    GG_HIP(hipGetLastError());
    return GG_OK;
  }
}

// parts, adjacency model: GG_PART_F = latent positions, GG_PART_MU0 = mu_0, GG_PART_LMBDA = mu_self (adjacency.py:360-372);
// weight model: GG_PART_F = L, GG_PART_MU0 = b, GG_PART_PARAMS = sigma + self Gaussian, GG_PART_LMBDA = eta (weights.py:767-773)
// Wait for the enqueued resamples; time of the last one, accepted L proposals of all of them.
int gg_latent_sync(gg_latent* e) {
  GG_HIP(hipStreamSynchronize(e->stream));
  float ms = 0;
  GG_HIP(hipEventElapsedTime(&ms, e->e0, e->e1));
  e->st.total_ms = ms;
  int acc[2] = {0, 0};
  GG_HIP(hipMemcpy(acc, e->acc, sizeof(acc), hipMemcpyDeviceToHost));
  e->st.label_changes += (uint64_t)acc[1];
  GG_HIP(hipMemset(e->acc + 1, 0, sizeof(int)));
  return GG_OK;
}

// defer != 0: enqueue only (a caller running several sweeps synchronises once, with gg_latent_sync).
int gg_latent_resample(gg_latent* e, const gg_data* d, int parts, int defer) {
  int rc = ld_check_data(e, d);
  if (rc) return rc;
  const int N = e->N, n = N * e->dim;
  hipStream_t st = e->stream;
  // a part asked for a second time under the same sweep index starts a new one: fresh Philox counters (graphgibbs.h)
  if (parts & e->parts_done) { e->sweep += 1; e->parts_done = 0; }
  GG_HIP(hipEventRecord(e->e0, st));
  GG_HIP(hipMemsetAsync(e->acc, 0, sizeof(int), st));
  const uint32_t base = (uint32_t)n;
  if (parts & GG_PART_F)
    if ((rc = ld_hmc_L(e, d))) return rc;
  int full;
  if (e->wmode) {
    full = GG_PART_F | GG_PART_MU0 | GG_PART_PARAMS | GG_PART_LMBDA;
    const int rest = parts & (GG_PART_MU0 | GG_PART_PARAMS | GG_PART_LMBDA);
    if (rest) {
      if ((rc = ld_eval(e, d, e->L, LS_MU0, LS_LP1))) return rc;     // sums at the (possibly new) L
      const double q1 = ldexp(1.0, -d->shift);
      gg_ldw_hyp h = {e->prior[1], e->prior[2], e->prior_self[0], e->prior_self[1], e->prior_self[2], e->prior_self[3],
                      (double)d->d1, (double)d->dS[0] * q1, (double)d->dS[1] * q1 * q1};
      k_ldw_params<<<1, 64, 0, st>>>(n, e->scal, h, e->beta, e->cfg.seed, e->sweep, rest);
      GG_HIP(hipGetLastError());
    }
  } else {
    full = GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA;
    if (parts & GG_PART_MU0) {
      // every pass is followed by the reduction, which also runs the scalar leapfrog step of mu_0 (phase 0: start, 1: middle, 2: last + test)
      if ((rc = ld_eval(e, d, e->L, LS_MU0, LS_LP0, 1, 0, 0))) return rc;
      for (int s = 0; s < LD_NSTEPS; ++s) {
        const int last = s == LD_NSTEPS - 1;
        if ((rc = ld_eval(e, d, e->L, LS_QMU0, LS_LP1, last, 0, last ? 2 : 1))) return rc;   // log p at the end only
      }
    }
    if (parts & GG_PART_LMBDA) {
      k_ld_muself<<<1, 256, 0, st>>>(N, d->dA, e->scal, e->acc, e->beta, e->cfg.seed, e->sweep, base);
      GG_HIP(hipGetLastError());
    }
  }
  GG_HIP(hipEventRecord(e->e1, st));
  if ((parts & full) == full) { e->sweep += 1; e->st.sweeps += 1; e->parts_done = 0; }
  else e->parts_done |= (parts & full);
  return defer ? GG_OK : gg_latent_sync(e);
}

int gg_latent_log_likelihood(gg_latent* e, const gg_data* d, double* out) {
  if (!out) return gg_fail(GG_ERR_INVALID, "null argument");
  int rc = ld_check_data(e, d);
  if (rc) return rc;
  if ((rc = ld_eval(e, d, e->L, LS_MU0, LS_LP1))) return rc;
  if (!e->wmode) {
    GG_HIP(hipMemcpyAsync(out, e->scal + LS_LL, 8, hipMemcpyDeviceToHost, e->stream));
    GG_HIP(hipStreamSynchronize(e->stream));
    return GG_OK;
  }
  // abstractions.py:257-274 with Mu, Sigma of weights.py:649-670: off-diagonal edges from the pass, self edges from the
  // exact diagonal sums of the data handle
  double h[LS_COUNT];
  GG_HIP(hipMemcpyAsync(h, e->scal, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  GG_HIP(hipStreamSynchronize(e->stream));
  const double l2pi = 1.8378770664093453, q1 = ldexp(1.0, -d->shift);
  const double ns = (double)d->d1, s1 = (double)d->dS[0] * q1, s2 = (double)d->dS[1] * q1 * q1;
  const double mus = h[LS_WMUS], sgs = h[LS_WSIGS];
  *out = -0.5 * h[LS_N] * (l2pi + log(h[LS_SIG])) - 0.5 * h[LS_SS] / h[LS_SIG]
         - 0.5 * ns * (l2pi + log(sgs)) - 0.5 * (s2 - 2.0 * mus * s1 + ns * mus * mus) / sgs;
  return GG_OK;
}

int gg_latent_stats(gg_latent* e, gg_stats* out) {
  *out = e->st;
  return GG_OK;
}
