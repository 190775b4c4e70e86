// gg_eigen.hip — logistic eigenmodel Gibbs sweep on gfx950
// (reference graphistician/eigenmodel.py:822-965).
//
//   resample = _resample_Omega -> _resample_F -> _resample_mu_0 -> _resample_lmbda  (eigenmodel.py:836-857)
//
// Design:
//  * Omega enters every conditional only through S = Omega + Omega^T (off-diagonal)
//    and its diagonal: eigenmodel.py:898-901 uses O[n,nn] + O[nn,n] and
//    O zcent = kappa - O mu_0; :948-958 likewise; :918-925 sums.  The engine
//    therefore draws omega_ij and omega_ji in the same thread and stores only the
//    symmetric S (fp64, N x Np) plus the diagonal: one N^2 write, row reads only.
//  * The latent-feature scan is sequential in n (eigenmodel.py:884,908).  It is
//    tiled: for a tile of T nodes the Gram statistics against F at tile start are
//    computed chip-wide (k_eig_tile_stats), then one workgroup walks the tile,
//    correcting each node's statistics by the rank-one changes of the tile's
//    earlier nodes (LDS-resident T x T sub-block of S) and drawing F[n] by a D x D
//    Cholesky.  Floating-point sums are ordered differently from the reference's
//    double loop: parity is a float tolerance (north_star), not bit-exact.
//  * All normals of a sweep are pre-drawn in parallel (counter-based RNG).
#include <math.h>
#include <stdlib.h>

#include "gg_common.h"
#include "spec/gg_pg.h"
#include "spec/gg_probit.h"
#include "spec/gg_eig.h"

#define EIG_T 64          // nodes per tile of the F scan
#define EIG_CHN 1024      // neighbours per workgroup of the tile statistics (riders of the chain launches)

struct gg_eigen {
  gg_config cfg;
  int N, D, Np, DM;
  int probit;            // GG_PROBIT_EIG: buffer S holds Z_ij + Z_ji, the pair weight is the constant 2
  double beta;           // AIS inverse temperature (logistic link only): omega ~ PG(beta, psi), kappa = beta (A - 1/2)
  int NEG, NE, NG;        // D(D+1)/2, NEG + D, 2 + 2 DM + DM(DM+1)/2
  int NEm;                // entries of a node's tile statistics, padded to DM: DM(DM+1)/2 + DM
  int NCH;
  double *F, *lmbda, *mu0, *mu_lmbda;
  double *S, *dO, *Z;
  double *Sb, *Vb;        // tile-local pair weights / mean coefficients, [N][EIG_VW] (k_eig_vblocks)
  // Several graphs per model (resample(networks), eigenmodel.py:836-857) and unobserved entries (NaN in A, :121-127): every
  // conditional is linear in the pair's summed auxiliary variables and summed centred observations, so the general form keeps
  //   a2[i][j]  = sum over graphs and OBSERVED directions (i->j, j->i) of (2 A - 1)   (int8; the logistic kappa sum is ka a2 / 2)
  //   nob[i][j] = the number of those observed entries                                (uint8; the probit pair weight)
  // and their diagonal counterparts, built by k_eig_obs when the auxiliary variables are drawn; S accumulates over the graphs.
  // One fully observed graph (gen == 0) reads A and A^T as before.
  int8_t *a2, *da2;
  uint8_t *nob;
  int gen;                // the auxiliary variables in S were drawn in the general form
  unsigned long long omega_uid;   // the network they were drawn for (one graph), 0: a list
  std::vector<unsigned long long> omega_uids;   // ... every network of the list, in order (parts drawn later must name the same list)
  double *part, *gpart, *llpart;
  int* status;
  long long* dbgbuf;   // GG_EIG_DBG: cycle counters of the chain's solver wave
  hipStream_t stream;
  // round 4: the auxiliary draws of a full sweep run on a second stream, in the order the feature scan consumes them (DESIGN.md
  // section 7): that stream's queue is kept off a few compute units (CU mask), so that the scan's one workgroup per tile - which
  // needs most of a compute unit's LDS - always finds one free instead of waiting for the draws' workgroups to drain
  hipStream_t stream2;
  hipEvent_t ev_fork, ev_chunk[16];
  int overlap;             // 1: the overlapped form is available (second stream created)
  uint32_t sweep;
  int parts_done;          // parts already drawn under the current sweep index (step-wise calls, see gg_resample_part)
  bool have_omega;
  hipEvent_t e0, e1;
  gg_stats st;
};

int gg_eigen_resample_many(gg_eigen* e, const gg_data* const* ds, int nd, int parts);

// ------------------------------------------------------------------ kernels
typedef double gg_double4 __attribute__((ext_vector_type(4)));

// psi tile = mu_0 + (F lambda) F^T for a 32 x 32 tile (eigenmodel.py:81) on the matrix cores:
// v_mfma_f64_16x16x4_f64, one 16 x 16 block per wave (4 waves = the tile), DM / 4 k-steps.
// Fragment layout, measured on gfx950 with tools/probe/mfma_f64_layout.hip: lane l supplies A[l % 16][l / 16]
// and B[l / 16][l % 16]; it receives D[4 r + l / 16][l % 16] in component r.  s_fi holds the rows of F
// scaled by lambda.
template <int DM>
__device__ __forceinline__ void eig_psi_tile_mfma(const double (*s_fi)[DM], const double (*s_fj)[DM], double mu0,
                                                  double (*s_psi)[33]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = wave >> 1, bj = wave & 1;
  gg_double4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < DM / 4; ++kk) {
    const double a = s_fi[bi * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
    const double b = s_fj[bj * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) s_psi[bi * 16 + 4 * r + (lane >> 4)][bj * 16 + (lane & 15)] = mu0 + acc[r];
}

// Where a pair's centred observation and (probit) weight come from: one fully observed graph reads its byte matrices, the
// general form (several graphs, unobserved entries) the summed int8 matrices (gg_eigen::a2 / nob).
struct eig_obs {
  const uint8_t *A, *AT;
  const int8_t* a2;
  const uint8_t* nob;
  int gen;
};
// Pair (i, j) at offset `at`: weight s and centred value a of the conditionals (eigenmodel.py:898-901, :312-321).
// logistic: s = the summed omegas (raw), a = sum of kappa = ka (A - 1/2) over the observed entries; probit: s = the number of
// observed entries (2 for one fully observed graph), a = the summed Z (raw).  ka a2 / 2 == ka (A + A^T - 1) to the bit.
__device__ __forceinline__ void eig_pair(const eig_obs& o, int probit, double ka, double raw, size_t at, double* s, double* a) {
  if (o.gen) {
    *s = probit ? (double)o.nob[at] : raw;
    *a = probit ? raw : 0.5 * ka * (double)o.a2[at];
  } else {
    *s = probit ? 2.0 : raw;
    *a = probit ? raw : ka * (double)((int)o.A[at] + (int)o.AT[at] - 1);
  }
}

// a2 / nob (and the diagonal's da2) of one graph, added to what the earlier graphs of the list left (first: stored).
__global__ void k_eig_obs(int N, int Np, const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                          const uint8_t* __restrict__ dA, int8_t* __restrict__ a2, uint8_t* __restrict__ nob,
                          int8_t* __restrict__ da2, int first) {
  const size_t at = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (at >= (size_t)N * Np) return;
  const int i = (int)(at / Np), j = (int)(at % Np);
  int v = 0, nb = 0;
  if (j < N && j != i) {
    const int a = A[at], t = AT[at];
    if (a != GG_A_MISSING) { v += 2 * a - 1; nb += 1; }
    if (t != GG_A_MISSING) { v += 2 * t - 1; nb += 1; }
  }
  a2[at] = (int8_t)((first ? 0 : (int)a2[at]) + v);
  nob[at] = (uint8_t)((first ? 0 : (int)nob[at]) + nb);
  if (j == 0) {
    const int d = dA[i];
    da2[i] = (int8_t)((first ? 0 : (int)da2[i]) + (d != GG_A_MISSING ? 2 * d - 1 : 0));
  }
}

__global__ void k_eig_normals(double* Z, int n, uint64_t seed, uint32_t site, uint32_t sweep) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) Z[i] = gg_normal_at(seed, site, sweep, (uint32_t)i);
}

__device__ __forceinline__ void eig_tile_pair(int b, int nt, int* I, int* J) {
  // b-th pair (I <= J) in row-major order of the upper triangle
  int i = 0, rem = b;
  while (rem >= nt - i) { rem -= nt - i; ++i; }
  *I = i; *J = i + rem;
}

// eigenmodel.py:859-874: omega_ij ~ PG(1, psi_ij), psi = mu_0 + (F lambda) F^T, diag psi = mu_0 (:81-82).
// gen: entries given as GG_A_MISSING are unobserved - no omega is drawn for them (their likelihood term is absent);
// accumulate: a further graph of the list (its own Philox site), added to S / dO.
// TEMP: the tempered draw PG(beta, psi), beta != 1 (its gamma sum would otherwise set the register count of the plain
// sweep's kernel, which runs at four waves per SIMD: 2.40 -> 1.96 ms at N=4096 against two).
template <int DM, bool TEMP>
__global__ __launch_bounds__(256, 4) void k_eig_omega(int N, int Np, int D, const double* __restrict__ F,
                                                   const double* __restrict__ lmbda, const double* __restrict__ mu0p,
                                                   uint64_t seed, uint32_t sweep, double* __restrict__ S,
                                                   double* __restrict__ dO, double beta, const uint8_t* __restrict__ A,
                                                   const uint8_t* __restrict__ AT, const uint8_t* __restrict__ dA, int gen,
                                                   int accumulate, uint32_t site, int pair_base) {
  __shared__ double s_fi[32][DM], s_fj[32][DM];
  __shared__ double s_t[32][33];
  __shared__ double s_psi[32][33];
  const int nt = (N + 31) / 32;
  int I, J;
  eig_tile_pair((int)blockIdx.x + pair_base, nt, &I, &J);
  const int tid = threadIdx.x;
  const double mu0 = *mu0p;
  for (int e = tid; e < 32 * DM; e += 256) {
    int r = e / DM, d = e % DM;
    int gi = I * 32 + r, gj = J * 32 + r;
    s_fi[r][d] = (d < D && gi < N) ? F[(size_t)gi * D + d] * lmbda[d] : 0.0;
    s_fj[r][d] = (d < D && gj < N) ? F[(size_t)gj * D + d] : 0.0;
  }
  __syncthreads();
  eig_psi_tile_mfma<DM>(s_fi, s_fj, mu0, s_psi);
  __syncthreads();
  {
    // The draws of a lane as ONE loop over attempts (spec/gg_pg.h gg_pg1_attempt; tempered: gg_pg_frac_attempt, a trial of one
    // term of the gamma sum): a lane whose draw is accepted goes on to its next draw at once instead of waiting for the slowest
    // lane of the wavefront at every draw (the nested rejection loops ran with a quarter of the lanes active:
    // SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = 15 of 64).  PG(1, psi): the mixture masses first, all lanes together.
    __shared__ double s_p[32][33];
    for (int e = tid; e < 1024; e += 256) {
      const int li = e >> 5, lj = e & 31;
      const int gi = I * 32 + li, gj = J * 32 + lj;
      if (gi < N && gj < N) {
        if (I == J && li == lj) {
          double dv = 0.0;
          if (!gen || dA[gi] != GG_A_MISSING) {
            gg_stream s0;
            gg_stream_init64(&s0, seed, site, sweep, (uint64_t)gi * (uint64_t)N + (uint64_t)gi);
            // tempered likelihood (AIS; build-defined, spec/gg_pg.h gg_pg_frac): omega ~ PG(beta, psi), nothing at beta = 0
            dv = TEMP ? (beta > 0.0 ? gg_pg_frac(&s0, beta, mu0) : 0.0) : gg_pg1(&s0, mu0);
          }
          dO[gi] = accumulate ? dO[gi] + dv : dv;
        } else if (!TEMP && (I < J || li < lj)) {
          s_p[li][lj] = gg_pg_mass_texpon(0.5 * gg_fabs(s_psi[li][lj]));
        }
      }
    }
    // the tile's 2 x 1024 draws dealt out through a counter in LDS: a lane takes the next draw when its own is accepted, so the
    // lanes of a wavefront end within an attempt or two of each other whatever their psi (dealt out statically, four entries a
    // lane, the wavefront ran 20 attempts against a mean of 11: entries with |psi| > 3 need more of them).  The two
    // directions of an entry may end up on different lanes: they are ADDED into the tile (0 + a + b in either order: the
    // same double).
    __shared__ int s_next;
    if (tid == 0) s_next = 0;
    for (int e = tid; e < 1024; e += 256) s_t[e >> 5][e & 31] = 0.0;
    __syncthreads();
    gg_stream st;
    int tig = 0, li = 0, lj = 0;       // (tig: PG(1, psi): a trial pending; PG(beta, psi): the term of the gamma sum)
    double Zh = 0.0, ptex = 0.0;       // (PG(beta, psi): psi^2 / 4 pi^2 and the sum so far)
    const bool none = TEMP && !(beta > 0.0);     // (beta = 0: no draws at all, the tile stays zero)
    bool busy = false;
    while (!none) {
      if (!busy) {
        const int d = atomicAdd(&s_next, 1);
        if (d >= 2048) break;
        const int e = d >> 1, dir = d & 1;
        li = e >> 5; lj = e & 31;
        const int gi = I * 32 + li, gj = J * 32 + lj;
        if (!(gi < N && gj < N && (I < J || li < lj))) continue;
        if (gen && (dir == 0 ? A : AT)[(size_t)gi * Np + gj] == GG_A_MISSING) continue;     // i -> j, j -> i not observed
        gg_stream_init64(&st, seed, site, sweep, dir == 0 ? (uint64_t)gi * (uint64_t)N + (uint64_t)gj : (uint64_t)gj * (uint64_t)N + (uint64_t)gi);
        if (TEMP) {
          const double psi = s_psi[li][lj];
          tig = 1;
          Zh = psi * psi / (4.0 * GG_PI * GG_PI);
          ptex = 0.0;
        } else {
          tig = 0;
          Zh = 0.5 * gg_fabs(s_psi[li][lj]);
          ptex = s_p[li][lj];
        }
        busy = true;
      }
      double om;
      if (TEMP ? gg_pg_frac_attempt(&st, beta, Zh, &tig, &ptex, &om) : gg_pg1_attempt(&st, Zh, ptex, &tig, &om)) {
        atomicAdd(&s_t[li][lj], om);
        busy = false;
      }
    }
  }
  __syncthreads();
  if (I == J) {
    for (int e = tid; e < 1024; e += 256) {
      const int li = e >> 5, lj = e & 31;
      if (li > lj) s_t[li][lj] = s_t[lj][li];
    }
    __syncthreads();
  }
  for (int e = tid; e < 1024; e += 256) {
    const int li = e >> 5, lj = e & 31;
    const int gi = I * 32 + li, gj = J * 32 + lj;
    if (gi < N && gj < N) { const size_t o = (size_t)gi * Np + gj; S[o] = accumulate ? S[o] + s_t[li][lj] : s_t[li][lj]; }
  }
  if (I < J) {
    for (int e = tid; e < 1024; e += 256) {
      const int lj = e >> 5, li = e & 31;   // transposed tile, coalesced over li
      const int gi = I * 32 + li, gj = J * 32 + lj;
      if (gi < N && gj < N) { const size_t o = (size_t)gj * Np + gi; S[o] = accumulate ? S[o] + s_t[li][lj] : s_t[li][lj]; }
    }
  }
}

// Probit auxiliaries (eigenmodel.py:274-293): Z_ij ~ N(psi_ij, 1) truncated to Z > 0 where A_ij = 1 and Z < 0 where
// A_ij = 0.  Z enters the conditionals only through Z_ij + Z_ji off the diagonal (:312-321, :339-346, :363-372),
// which is what this kernel stores in S (symmetric, diagonal 0): same tiling as k_eig_omega.
template <int DM>
__global__ __launch_bounds__(256, 4) void k_eig_z(int N, int Np, int D, const double* __restrict__ F,
                                               const double* __restrict__ lmbda, const double* __restrict__ mu0p,
                                               const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                               uint64_t seed, uint32_t sweep, double* __restrict__ S, int accumulate, uint32_t site,
                                               int pair_base) {
  __shared__ double s_fi[32][DM], s_fj[32][DM];
  __shared__ double s_t[32][33];
  __shared__ double s_psi[32][33];
  const int nt = (N + 31) / 32;
  int I, J;
  eig_tile_pair((int)blockIdx.x + pair_base, nt, &I, &J);
  const int tid = threadIdx.x;
  const double mu0 = *mu0p;
  for (int e = tid; e < 32 * DM; e += 256) {
    int r = e / DM, d = e % DM;
    int gi = I * 32 + r, gj = J * 32 + r;
    s_fi[r][d] = (d < D && gi < N) ? F[(size_t)gi * D + d] * lmbda[d] : 0.0;
    s_fj[r][d] = (d < D && gj < N) ? F[(size_t)gj * D + d] : 0.0;
  }
  __syncthreads();
  eig_psi_tile_mfma<DM>(s_fi, s_fj, mu0, s_psi);
  __syncthreads();
  for (int e = tid; e < 1024; e += 256) {
    const int li = e >> 5, lj = e & 31;
    const int gi = I * 32 + li, gj = J * 32 + lj;
    double val = 0.0;
    if (gi < N && gj < N && (I < J || li < lj)) {
      const double psi = s_psi[li][lj];
      const size_t at = (size_t)gi * Np + gj;
      // (an unobserved entry has no auxiliary variable: it is absent from every conditional)
      const int a1 = A[at], a2v = AT[at];
      const double z1 = a1 == GG_A_MISSING ? 0.0 : gg_truncnorm(gg_uniform_at64(seed, site, sweep, (uint64_t)gi * (uint64_t)N + (uint64_t)gj), psi, a1 ? 1 : 0);
      const double z2 = a2v == GG_A_MISSING ? 0.0 : gg_truncnorm(gg_uniform_at64(seed, site, sweep, (uint64_t)gj * (uint64_t)N + (uint64_t)gi), psi, a2v ? 1 : 0);
      val = z1 + z2;
    }
    if (I < J || li <= lj) s_t[li][lj] = val;
  }
  __syncthreads();
  if (I == J) {
    for (int e = tid; e < 1024; e += 256) {
      const int li = e >> 5, lj = e & 31;
      if (li > lj) s_t[li][lj] = s_t[lj][li];
    }
    __syncthreads();
  }
  for (int e = tid; e < 1024; e += 256) {
    const int li = e >> 5, lj = e & 31;
    const int gi = I * 32 + li, gj = J * 32 + lj;
    if (gi < N && gj < N) { const size_t o = (size_t)gi * Np + gj; S[o] = accumulate ? S[o] + s_t[li][lj] : s_t[li][lj]; }
  }
  if (I < J) {
    for (int e = tid; e < 1024; e += 256) {
      const int lj = e >> 5, li = e & 31;   // transposed tile, coalesced over li
      const int gi = I * 32 + li, gj = J * 32 + lj;
      if (gi < N && gj < N) { const size_t o = (size_t)gj * Np + gi; S[o] = accumulate ? S[o] + s_t[li][lj] : s_t[li][lj]; }
    }
  }
}

// ------------------------------------------------------------------ _resample_F (eigenmodel.py:876-908, :295-329)
//
// The scan over nodes is sequential with a continuous state, so nothing can be speculated; what CAN be chosen is what sits
// on the per-node critical path.  With the draw made by perturbation (spec/gg_eig.h) node n needs  x = Lambda_n^{-1} b_n,
//     Lambda_n = I / sigma_F + sum_{nn != n} s_{n,nn} g_nn g_nn^T,    b_n = z_n / sqrt(sigma_F) + sum_{nn != n} v_{n,nn} g_nn,
// g = F * lambda (current), v = the pair's mean coefficient with the perturbation noise folded in (gg_eig_coef).  Inside a tile
// of 64 nodes, node i's (Lambda, b) differ from what was known at tile start only by the rank-one terms of the tile's earlier
// nodes - so instead of factorising Lambda_i when its turn comes (round 1-2: a D x D Cholesky and two substitutions on one
// lane, 3400 clocks per node), the tile keeps the INVERSE of every node's precision in registers and folds each published
// feature vector u_k = g_k(new) into the later nodes by Sherman-Morrison,
//     q = P u,  t = u^T q,  c = s / (1 + s t),  P <- P - c q q^T,  b <- b + v u,
// DM lanes per node (lane j owns row j of P_i and b_i[j]).  Node k+1's own turn ("closing") then needs no update of P at all:
//     x = (P - c q q^T)(b + v u) = r + (v - c (d + v t)) q,      r = P b (prepared one step ahead),  d = u^T r,
// one matrix-vector product, two DM-lane reductions, one reciprocal.  The precisions at tile start leave out the tile's
// earlier nodes (their old features never enter), so every rank-one step ADDS a positive term: no downdates.
//   * riders (workgroups > 0 of every chain launch): Gram statistics of the NEXT tile's nodes against all nodes outside
//     both tiles (eig_tile_stats_body), the noise normals drawn on the fly;
//   * k_eig_vblocks, once per sweep: pair weights and mean coefficients of the 128 columns around every node's own tile
//     (previous tile | own tile), what the chain's prologue multiplies on the matrix cores;
//   * prologue of a chain launch: previous tile (new features) and the later nodes of the own tile (old features) are added
//     as two 64 x 64 x (NEGM + DM) products (v_mfma_f64_16x16x4_f64), then every lane factorises its node's precision
//     (redundantly within the node's DM lanes - SIMD lockstep makes that free) and solves for ITS row of the inverse.

#define EIG_VW 128   // columns per node in the tile-local blocks: [tile start - 64, tile start + 64)

// Pair weights s and mean coefficients v (with the perturbation noise) of node n against the 128 nodes of its own and the
// previous tile.  Entries outside the matrix and the diagonal are zero.
__global__ __launch_bounds__(64) void k_eig_vblocks(int N, int Np, const double* __restrict__ S, eig_obs ob,
                                                    const double* __restrict__ mu0p,
                                                    double* __restrict__ Sb, double* __restrict__ Vb, int probit, double ka,
                                                    uint64_t seed, uint32_t sweep, int n_base) {
  const int n = (int)blockIdx.x + n_base;
  const int m0 = (n / EIG_T) * EIG_T - EIG_T + 2 * (int)threadIdx.x;   // this thread's pair of columns (m0 even)
  const double mu0 = *mu0p;
  double sv[2] = {0.0, 0.0}, vv[2] = {0.0, 0.0};
  if (m0 >= 0 && m0 < N) {
    double e[2];
    gg_eig_noise_pair(seed, sweep, (uint32_t)N, (uint32_t)n, (uint32_t)(m0 >> 1), &e[0], &e[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = m0 + h;
      if (m < N && m != n) {
        const size_t at = (size_t)n * Np + m;
        double s, a;
        eig_pair(ob, probit, ka, S[at], at, &s, &a);
        sv[h] = s;
        vv[h] = gg_eig_coef(s, a, mu0, e[h]);
      }
    }
  }
  const size_t o = (size_t)n * EIG_VW + 2 * threadIdx.x;
  Sb[o] = sv[0]; Sb[o + 1] = sv[1];
  Vb[o] = vv[0]; Vb[o + 1] = vv[1];
}

// Gram statistics of a tile's nodes against the current F (eigenmodel.py:894-901), all nodes nn outside [ex_lo, n]: the
// tile whose scan is still running and the earlier nodes of n's own tile are left out (the chain adds them with their new
// features), the later nodes of the own tile enter with the features they have now - which is what node n will see:
//   G_n[a,b] = sum_nn s[n,nn] g_a g_b,   b_n[a] = sum_nn v[n,nn] g_a,   g = F[nn] * lambda.
// One workgroup per (node, chunk of EIG_CHN neighbours); a thread takes pairs of neighbours (one Philox block each).
// Output layout per (node, chunk): upper triangle of the DM x DM matrix (padded to DM), then the DM mean entries.
template <int DM>
__device__ __forceinline__ void eig_tile_stats_body(int N, int Np, int D, int t0, const double* __restrict__ S,
                                                    const eig_obs& ob,
                                                    const double* __restrict__ F, const double* __restrict__ lmbda,
                                                    const double* __restrict__ mu0p, int NCH, double* __restrict__ part,
                                                    int ex_lo, int node, int c, int probit, double ka,
                                                    uint64_t seed, uint32_t sweep) {
  constexpr int NEGM = DM * (DM + 1) / 2;
  const int n = t0 + node;
  if (n >= N) return;
  const int nth = blockDim.x;
  const double mu0 = *mu0p;
  double g[NEGM], h[DM], lam[DM];
#pragma unroll
  for (int e = 0; e < NEGM; ++e) g[e] = 0.0;
#pragma unroll
  for (int d = 0; d < DM; ++d) { h[d] = 0.0; lam[d] = d < D ? lmbda[d] : 0.0; }
  const int lo = c * EIG_CHN, hi = (lo + EIG_CHN < N) ? lo + EIG_CHN : N;   // EIG_CHN is even: lo is
  for (int nn0 = lo + 2 * (int)threadIdx.x; nn0 < hi; nn0 += 2 * nth) {
    // a pair whose both members are left out needs no normals
    const bool out0 = nn0 >= ex_lo && nn0 <= n;
    const bool out1 = nn0 + 1 >= hi || (nn0 + 1 >= ex_lo && nn0 + 1 <= n);
    if (out0 && out1) continue;
    double eps[2];
    gg_eig_noise_pair(seed, sweep, (uint32_t)N, (uint32_t)n, (uint32_t)(nn0 >> 1), &eps[0], &eps[1]);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (hh ? out1 : out0) continue;
      const int nn = nn0 + hh;
      // logistic: s = omega_ij + omega_ji, a = kappa_ij + kappa_ji; probit (eigenmodel.py:312-321): s = 2, a = Z_ij + Z_ji
      double s, a;
      eig_pair(ob, probit, ka, S[(size_t)n * Np + nn], (size_t)n * Np + nn, &s, &a);
      const double v = gg_eig_coef(s, a, mu0, eps[hh]);
      double ft[DM];
#pragma unroll
      for (int d = 0; d < DM; ++d) ft[d] = d < D ? F[(size_t)nn * D + d] * lam[d] : 0.0;
      int e = 0;
#pragma unroll
      for (int a1 = 0; a1 < DM; ++a1) {
        const double sa = s * ft[a1];
#pragma unroll
        for (int b1 = a1; b1 < DM; ++b1) g[e++] += sa * ft[b1];
        h[a1] += v * ft[a1];
      }
    }
  }
  __shared__ double s_red[16][NEGM + DM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < NEGM + DM; ++e) {
    double v = e < NEGM ? g[e] : h[e - NEGM];
    for (int dl = 32; dl >= 1; dl >>= 1) v += __shfl_down(v, dl);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < NEGM + DM) {
    const int e = threadIdx.x;
    double v = 0.0;
    for (int w = 0; w < (nth >> 6); ++w) v += s_red[w][e];
    part[((size_t)node * NCH + c) * (NEGM + DM) + e] = v;
  }
}

template <int DM>
__global__ __launch_bounds__(64 * DM) void k_eig_tile_stats(int N, int Np, int D, int t0, const double* __restrict__ S,
                                                            eig_obs ob,
                                                            const double* __restrict__ F, const double* __restrict__ lmbda,
                                                            const double* __restrict__ mu0p, int NCH,
                                                            double* __restrict__ part, int ex_lo, int probit, double ka,
                                                            uint64_t seed, uint32_t sweep) {
  eig_tile_stats_body<DM>(N, Np, D, t0, S, ob, F, lmbda, mu0p, NCH, part, ex_lo, blockIdx.x, blockIdx.y, probit, ka, seed, sweep);
}

// Order this wave's LDS operations: wait for the outstanding ones and keep the compiler from moving memory accesses across.
// (__threadfence_block() also waits for every global store in flight - hundreds of clocks on the scan's critical path.)
__device__ __forceinline__ void gg_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// x of another lane of the same DPP row (16 lanes), pattern CTRL: the two halves of the double travel separately.
template <int CTRL>
__device__ __forceinline__ double gg_dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum over the DM consecutive lanes of a node's group (DM = 4, 8, 16; groups are aligned), the same value in every lane.
// quad_perm [1,0,3,2] and [2,3,0,1] sum a quad; row_half_mirror (lane l <-> 7 - l) and row_mirror (l <-> 15 - l) pair
// a lane with one of the other quad / the other half.
template <int DM>
__device__ __forceinline__ double eig_group_sum(double x) {
  x += gg_dpp_f64<0xB1>(x);
  x += gg_dpp_f64<0x4E>(x);
  if (DM >= 8) x += gg_dpp_f64<0x141>(x);
  if (DM >= 16) x += gg_dpp_f64<0x140>(x);
  return x;
}

// The same for two values at once (the two dependent chains interleave).
template <int DM>
__device__ __forceinline__ void eig_group_sum2(double& x, double& y) {
  x += gg_dpp_f64<0xB1>(x); y += gg_dpp_f64<0xB1>(y);
  x += gg_dpp_f64<0x4E>(x); y += gg_dpp_f64<0x4E>(y);
  if (DM >= 8) { x += gg_dpp_f64<0x141>(x); y += gg_dpp_f64<0x141>(y); }
  if (DM >= 16) { x += gg_dpp_f64<0x140>(x); y += gg_dpp_f64<0x140>(y); }
}

// The sequential part of _resample_F for one tile (workgroup 0), riders behind it.  64 * DM threads = DM waves; node i of
// the tile lives in wave i % DM, lanes [(i / DM) DM, (i / DM + 1) DM) - consecutive nodes in different waves, so that while
// one wave closes node k the others fold change k - 1 into their nodes - lane j of the group owning row j of the node's
// inverse precision and a copy of its mean term.
template <int DM>
__global__ __launch_bounds__(64 * DM) void k_eig_chain(int N, int Np, int D, int t0, const double* __restrict__ S,
                                                       eig_obs ob,
                                                       double* __restrict__ F, const double* __restrict__ lmbda,
                                                       const double* __restrict__ mu0p, double sigma_F, int NCH,
                                                       const double* __restrict__ part, const double* __restrict__ Z,
                                                       const double* __restrict__ Sb, const double* __restrict__ Vb,
                                                       int* __restrict__ status, long long* __restrict__ dbg,
                                                       double* __restrict__ part_next, int probit, double ka,
                                                       uint64_t seed, uint32_t sweep) {
  constexpr int NEGM = DM * (DM + 1) / 2, NEm = NEGM + DM, TS = EIG_T + 1, NTH = 64 * DM, NW = DM, NPW = 64 / DM;
  if (blockIdx.x > 0) {
    // riders: the statistics of the NEXT tile against all nodes outside this tile and behind the node in its own
    const int idx = blockIdx.x - 1;
    eig_tile_stats_body<DM>(N, Np, D, t0 + EIG_T, S, ob, F, lmbda, mu0p, NCH, part_next, t0, idx / NCH, idx % NCH,
                            probit, ka, seed, sweep);
    return;
  }
  extern __shared__ __align__(16) unsigned char s_chain[];
  double* s_S = reinterpret_cast<double*>(s_chain);   // [T][TS]   pair weights of the block being multiplied / scanned
  double* s_V = s_S + EIG_T * TS;                     // [T][TS]   mean coefficients
  double* s_C = s_V + EIG_T * TS;                     // [T][NEm]  precision (upper triangle) and mean term per node
  double* s_g = s_C + EIG_T * NEm;                    // [T][DM]   f * lambda of the previous tile's nodes (new features)
  double* s_u = s_g + EIG_T * DM;                     // [T][DM]   published new f * lambda of this tile's nodes
  double* s_Fn = s_u + EIG_T * DM;                    // [T][DM]   new features
  double* s_q = s_Fn + EIG_T * DM;                    // [NTH]     gather scratch, one slot per lane
  int* s_pub = reinterpret_cast<int*>(s_q + NTH);      // (plain LDS accesses behind compiler barriers: a volatile pointer here becomes flat loads)
  double* s_Y = s_q + NTH + 2;                        // [T][NEm]  previous tile: g_a g_b (upper triangle) and g_a per node (prologue only)
  const int tid = threadIdx.x;
  const int T = (N - t0 < EIG_T) ? (N - t0) : EIG_T;
  const int p0 = t0 - EIG_T;       // first node of the previous tile (left out by the riders that ran beside its scan)
  const long long c_start = dbg ? clock64() : 0;
  // ---- prologue.  Every global load is issued before the first wait: the riders' partial sums, both 64 x 64 blocks of
  // pair weights / mean coefficients (previous tile: multiplied below; own tile: what the scan reads), the previous
  // tile's new features.
  constexpr int PER = EIG_T * EIG_T / NTH;            // block entries per thread
  constexpr int UC = (EIG_T * NEm + NTH - 1) / NTH;   // statistics entries per thread
  double sv0[PER], vv0[PER], sv1[PER], vv1[PER], cacc[UC];
#pragma unroll
  for (int u = 0; u < UC; ++u) cacc[u] = 0.0;
  for (int c0 = 0; c0 < NCH; c0 += 4) {
    double tmp[UC][4];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int e = tid + u * NTH;
      const int ii = e / NEm, q = e % NEm;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        tmp[u][cc] = (e < EIG_T * NEm && ii < T && c0 + cc < NCH) ? part[((size_t)ii * NCH + c0 + cc) * NEm + q] : 0.0;
    }
    if (c0 == 0) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int e = tid + u * NTH;
        const int ii = e >> 6, jj = e & 63;
        const size_t at = (size_t)(t0 + (ii < T ? ii : 0)) * EIG_VW + jj;
        sv0[u] = p0 >= 0 ? Sb[at] : 0.0;
        vv0[u] = p0 >= 0 ? Vb[at] : 0.0;
        sv1[u] = Sb[at + EIG_T];
        vv1[u] = Vb[at + EIG_T];
      }
      if (p0 >= 0)
        for (int e = tid; e < EIG_T * DM; e += NTH) {
          const int m = e / DM, q = e % DM;
          s_g[e] = q < D ? F[(size_t)(p0 + m) * D + q] * lmbda[q] : 0.0;
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < UC; ++u) cacc[u] += (tmp[u][0] + tmp[u][1]) + (tmp[u][2] + tmp[u][3]);
  }
#pragma unroll
  for (int u = 0; u < UC; ++u) { const int e = tid + u * NTH; if (e < EIG_T * NEm) s_C[e] = cacc[u]; }
  auto store_block = [&](const double* sv, const double* vv) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * NTH;
      const int ii = e >> 6, jj = e & 63;
      s_S[ii * TS + jj] = ii < T ? sv[u] : 0.0;
      s_V[ii * TS + jj] = ii < T ? vv[u] : 0.0;
    }
  };
  if (tid == 0) *s_pub = 0;
  if (p0 >= 0) {
    store_block(sv0, vv0);
    __syncthreads();
    // s_C[i][e] += sum_m X[i][m] Y[m][e] over the 64 nodes m of the previous tile, with their new features, on the matrix
    // cores (v_mfma_f64_16x16x4_f64; fragment layout as in eig_psi_tile_mfma).  Precision entries (e < NEGM): X = s,
    // Y[m][e] = g_m[a] g_m[b]; mean entries: X = v, Y[m][a] = g_m[a].
    {
      // Y[m][e]: the products the matrix cores multiply the previous tile's block with - thread (m, a) writes g_a g_b for
      // b >= a (upper triangle, row a) and the mean entry g_a (the scan's gather scratch and feature buffers are free: s_Y
      // lies over s_u .. s_q and beyond, see the LDS map)
      for (int e = tid; e < EIG_T * DM; e += NTH) {
        const int m = e / DM, a1 = e % DM;
        double g[DM];
#pragma unroll
        for (int q = 0; q < DM; ++q) g[q] = s_g[m * DM + q];
        double ga = 0.0;
#pragma unroll
        for (int q = 0; q < DM; ++q) ga = (q == a1) ? g[q] : ga;
        const int rowstart = a1 * DM - a1 * (a1 - 1) / 2;     // index of (a1, a1) in the padded upper triangle
#pragma unroll
        for (int b1 = 0; b1 < DM; ++b1)
          if (b1 >= a1) s_Y[m * NEm + rowstart + (b1 - a1)] = ga * g[b1];
        s_Y[m * NEm + NEGM + a1] = ga;
      }
    }
    __syncthreads();
    const int wv = tid >> 6, l = tid & 63;
    constexpr int ncb = (NEGM + 15) / 16;       // column blocks of the precision part
    constexpr int nblk = 4 * (ncb + 1);         // + one column block for the mean part
    constexpr int NB = nblk / (NTH / 64);       // 16 x 16 blocks per wave, accumulated side by side
    static_assert(nblk == NB * (NTH / 64), "whole blocks per wave");
    const double* X[NB];
    int yo[NB], eo[NB], ro[NB];
    bool cv[NB];
    gg_double4 acc[NB];
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      const int blk = wv + h * (NTH / 64), bi = blk & 3, cb = blk >> 2;
      const bool mean = cb == ncb;
      const int col = l & 15;
      eo[h] = mean ? NEGM + col : cb * 16 + col;
      cv[h] = mean ? col < DM : cb * 16 + col < NEGM;
      X[h] = (mean ? s_V : s_S) + (bi * 16 + (l & 15)) * TS;
      yo[h] = cv[h] ? eo[h] : 0;     // (lanes of columns that do not exist read entry 0 and store nothing)
      ro[h] = bi * 16 + (l >> 4);
      acc[h] = gg_double4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll 4
    for (int kk = 0; kk < EIG_T / 4; ++kk) {
      const int m = kk * 4 + (l >> 4);
#pragma unroll
      for (int h = 0; h < NB; ++h) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[h][m], s_Y[m * NEm + yo[h]], acc[h], 0, 0, 0);
    }
    // lane l holds C[16 bi + 4 r + l / 16][col block + l % 16] in component r
#pragma unroll
    for (int h = 0; h < NB; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (cv[h]) s_C[(ro[h] + 4 * r) * NEm + eo[h]] += acc[h][r];
    __syncthreads();
  }
  store_block(sv1, vv1);   // the own tile's block: what the scan reads
  __syncthreads();
  if (dbg && tid == 0) atomicAdd((unsigned long long*)&dbg[4], (unsigned long long)(clock64() - c_start));
  // ---- every lane: Cholesky of its node's precision (the DM lanes of a node do the same arithmetic), row j of the inverse
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = (tid & 63) / DM, j = tid % DM;
  const int i = grp * NW + wave;                  // this lane's node
  const int gbase = (tid / DM) * DM;              // first lane of the group (its slots of the gather scratch)
  const bool live = i < T;
  double Prow[DM], rj = 0.0;
  bool bad_x = false;
  const double lamj = j < D ? lmbda[j] : 0.0;
  {
    double L[DM][DM];
    const double inv_sigma_F = 1.0 / sigma_F;
    {
      int e = 0;
#pragma unroll
      for (int a1 = 0; a1 < DM; ++a1)
#pragma unroll
        for (int b1 = a1; b1 < DM; ++b1) {
          double v = live ? s_C[i * NEm + e] : 0.0;
          ++e;
          if (a1 == b1) v += inv_sigma_F;
          L[b1][a1] = v;   // lower triangle holds the precision
        }
    }
    double dmin = 1.0;     // smallest pivot: a non-positive one voids the sweep (status)
    // right-looking Cholesky; the diagonal keeps 1 / L_jj (hardware reciprocal square root + two Newton steps)
#pragma unroll
    for (int c = 0; c < DM; ++c) {
      const double dsum = L[c][c];
      dmin = __builtin_fmin(dmin, dsum);
      double inv = __builtin_amdgcn_rsq(dsum);
      inv = inv * gg_fma(-0.5 * dsum * inv, inv, 1.5);
      inv = inv * gg_fma(-0.5 * dsum * inv, inv, 1.5);
      L[c][c] = inv;
#pragma unroll
      for (int r = c + 1; r < DM; ++r) L[r][c] = L[r][c] * inv;
#pragma unroll
      for (int r = c + 1; r < DM; ++r)
#pragma unroll
        for (int k = c + 1; k <= r; ++k) L[r][k] = gg_fma(-L[r][c], L[k][c], L[r][k]);
    }
    if (live && !(dmin > 0.0)) atomicOr(status, 1);
    // row j of the inverse: L L^T p = e_j
    double y[DM];
#pragma unroll
    for (int r = 0; r < DM; ++r) {
      double v = (r == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) v = gg_fma(-L[r][k], y[k], v);
      y[r] = v * L[r][r];
    }
#pragma unroll
    for (int r = DM - 1; r >= 0; --r) {
      double v = y[r];
#pragma unroll
      for (int k = r + 1; k < DM; ++k) v = gg_fma(-L[k][r], Prow[k], v);
      Prow[r] = v * L[r][r];
    }
    // r = P b, this lane's component; b = the mean term + z / sqrt(sigma_F)
    const double isq = 1.0 / gg_sqrt(sigma_F);
#pragma unroll
    for (int m = 0; m < DM; ++m) {
      const double bm = (live && m < D) ? gg_fma(Z[(size_t)(t0 + i) * D + m], isq, s_C[i * NEm + NEGM + m]) : 0.0;
      rj = gg_fma(Prow[m], bm, rj);
    }
  }
  const long long c_scan = dbg ? clock64() : 0;
  // ---- the scan.  Step k: node k closes with change k - 1 (the features node k - 1 has just published); the later nodes
  // fold change k - 1 into their inverse.  A wave whose nodes are all closed leaves.
  const int last = T > wave ? ((T - 1 - wave) / NW) * NW + wave : -1;   // this wave's last node
  for (int k = 0; k <= last; ++k) {
    const bool act = live && i >= k;
    const bool closer = (k % NW) == wave;       // this wave holds node k
    double u[DM], uj = 0.0, s = 0.0, v = 0.0, qj = 0.0, c = 0.0;
    if (k >= 1) {
      for (int spin = 0;; ++spin) {
        asm volatile("" ::: "memory");
        if (*s_pub >= k) break;
        if (spin > (1 << 24)) { if ((tid & 63) == 0) atomicOr(status, 2); break; }   // never spin forever
        if (!closer) __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      gg_lds_wait();
#pragma unroll
      for (int m = 0; m < DM; ++m) u[m] = s_u[(k - 1) * DM + m];
      uj = s_u[(k - 1) * DM + j];
      if (act) { s = s_S[i * TS + k - 1]; v = s_V[i * TS + k - 1]; }
      // fold change k - 1 into r = P b (the P half follows below, behind the publication):
      //   r' = (P - c q q^T)(b + v u) = r + q (v - s d) / (1 + s t),   t = u^T q,  d = u^T r
      double qa = 0.0, qb = 0.0;
#pragma unroll
      for (int m = 0; m < DM / 2; ++m) { qa = gg_fma(Prow[m], u[m], qa); qb = gg_fma(Prow[DM / 2 + m], u[DM / 2 + m], qb); }
      qj = qa + qb;
      double tq = uj * qj, d = uj * rj;
      eig_group_sum2<DM>(tq, d);
      const double den = gg_fma(s, tq, 1.0);
      double rc = __builtin_amdgcn_rcp(den);
      rc = rc * gg_fma(-den, rc, 2.0);
      rc = rc * gg_fma(-den, rc, 2.0);
      c = s * rc;
      rj = gg_fma(gg_fma(-s, d, v) * rc, qj, rj);
    }
    if (closer) {
      // node k's new features are its r with every earlier change of the tile folded in
      if (act && i == k) {
        double x = rj;
        if (!(x == x)) { bad_x = true; x = 0.0; }
        s_u[k * DM + j] = x * lamj;
        s_Fn[k * DM + j] = x;
      }
      gg_lds_wait();   // LDS traffic of one wave is performed in order: the data is written before the word that announces it
      if (act && i == k && j == 0) *s_pub = k + 1;
      asm volatile("" ::: "memory");
    }
    if (k >= 1 && last > k) {
      // the later nodes of this wave: P <- P - c q q^T (lanes of closed nodes compute on, nobody reads them)
      s_q[tid] = qj;
      gg_lds_wait();
      const double cq = c * qj;
#pragma unroll
      for (int m = 0; m < DM; ++m) Prow[m] = gg_fma(-cq, s_q[gbase + m], Prow[m]);
      gg_lds_wait();
    }
  }
  if (bad_x) atomicOr(status, 1);
  __syncthreads();
  if (dbg && tid == 0) {
    atomicAdd((unsigned long long*)&dbg[2], (unsigned long long)(clock64() - c_scan));
    atomicAdd((unsigned long long*)&dbg[3], (unsigned long long)T);
    atomicAdd((unsigned long long*)&dbg[5], (unsigned long long)(c_scan - c_start));
  }
  // the tile's new features -> global memory (nothing in this launch reads them: the riders leave this tile out)
  for (int e = tid; e < T * D; e += NTH) F[(size_t)(t0 + e / D) * D + e % D] = s_Fn[(e / D) * DM + e % D];
}

// Statistics of _resample_mu_0 (eigenmodel.py:914-925) and _resample_lmbda (:938-958) over pairs i < j:
//   [0] sum S_ij   [1] sum (A_ij + A_ji - 1)   [2..2+D) sum S g   [2+D..2+2D) sum a g   then upper triangle of sum S g g^T,
// g = F_i * F_j (elementwise).  One workgroup per row i; partials reduced in k_eig_draw_globals.
template <int DM>
__global__ __launch_bounds__(256) void k_eig_global(int N, int Np, int D, const double* __restrict__ S,
                                                    eig_obs ob,
                                                    const double* __restrict__ F, double* __restrict__ gpart, int probit, double ka) {
  constexpr int NEGM = DM * (DM + 1) / 2;
  constexpr int NGM = 2 + 2 * DM + NEGM;
  double acc[NGM];
#pragma unroll
  for (int e = 0; e < NGM; ++e) acc[e] = 0.0;
  // A few rows per workgroup (<= 512 partials to sum afterwards), EIGHT at a time with a lane's F[j] held in registers across
  // them: per pair the pass then reads its entry of S and of the observations only - read row by row, F[j] came from L2 once per
  // PAIR (620 MB per launch at N=4096, the bound of the pass: 138 us).
  constexpr int RB = 8;
  __shared__ double s_fi[RB][DM];
  for (int r0 = blockIdx.x; r0 < N; r0 += RB * gridDim.x) {
    __syncthreads();
    if (threadIdx.x < RB * DM) {
      const int r = threadIdx.x / DM, d = threadIdx.x % DM, i = r0 + r * gridDim.x;
      s_fi[r][d] = (i < N && d < D) ? F[(size_t)i * D + d] : 0.0;
    }
    __syncthreads();
    for (int j = r0 + 1 + threadIdx.x; j < N; j += 256) {
      double fj[DM];
#pragma unroll
      for (int d = 0; d < DM; ++d) fj[d] = d < D ? F[(size_t)j * D + d] : 0.0;
#pragma unroll 2
      for (int r = 0; r < RB; ++r) {
        const int i = r0 + r * gridDim.x;          // (ascending in r: past the first row that is not above j, none is)
        if (i >= N || i >= j) break;
        double s, a;
        eig_pair(ob, probit, ka, S[(size_t)i * Np + j], (size_t)i * Np + j, &s, &a);
        double g[DM];
#pragma unroll
        for (int d = 0; d < DM; ++d) g[d] = s_fi[r][d] * fj[d];
        acc[0] += s;
        acc[1] += a;
        int e = 2 + 2 * DM;
#pragma unroll
        for (int a1 = 0; a1 < DM; ++a1) {
          const double sg = s * g[a1];
          acc[2 + a1] += sg;
          acc[2 + DM + a1] += a * g[a1];
#pragma unroll
          for (int b1 = a1; b1 < DM; ++b1) acc[e++] += sg * g[b1];
        }
      }
    }
  }
  __shared__ double s_red[4][NGM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < NGM; ++e) {
    double v = acc[e];
    for (int dl = 32; dl >= 1; dl >>= 1) v += __shfl_down(v, dl);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < NGM) {
    const int e = threadIdx.x;
    gpart[(size_t)blockIdx.x * NGM + e] = (s_red[0][e] + s_red[1][e]) + (s_red[2][e] + s_red[3][e]);
  }
}

template <int DM>
__global__ __launch_bounds__(256) void k_eig_draw_globals(int N, int NP, int D, const double* __restrict__ gpart,
                                                          const double* __restrict__ dO, const uint8_t* __restrict__ dA,
                                                          const int8_t* __restrict__ da2,
                                                          double* __restrict__ lmbda, double* __restrict__ mu0p,
                                                          const double* __restrict__ mu_lmbda, double mu_mu_0,
                                                          double sigma_mu0, double sigma_lmbda, int do_mu0, int do_lmbda,
                                                          uint64_t seed, uint32_t sweep, int* __restrict__ status, int probit, double ka) {
  constexpr int NEGM = DM * (DM + 1) / 2;
  constexpr int NGM = 2 + 2 * DM + NEGM;
  __shared__ double s_tot[NGM + 2];
  __shared__ double s_w[4][2];
  const int tid = threadIdx.x;
  {
    // fixed-order two-level sum of the per-row partials: 4 row quarters per entry
    __shared__ double s_q[4][NGM];
    const int e = tid & 63, q = tid >> 6;
    if (e < NGM) {
      const int lo = (int)((long long)NP * q / 4), hi = (int)((long long)NP * (q + 1) / 4);
      double v = 0.0;
      int i = lo;
      for (; i + 8 <= hi; i += 8) {        // (eight loads in flight, added in the same order as one by one)
        double w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = gpart[(size_t)(i + u) * NGM + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += w[u];
      }
      for (; i < hi; ++i) v += gpart[(size_t)i * NGM + e];
      s_q[q][e] = v;
    }
    __syncthreads();
    if (tid < NGM) s_tot[tid] = (s_q[0][tid] + s_q[1][tid]) + (s_q[2][tid] + s_q[3][tid]);
  }
  // diagonal: sum omega_ii and sum (A_ii - 1/2)
  double so = 0.0, sk = 0.0;
  if (!probit)   // (the probit bias update leaves the diagonal out altogether: eigenmodel.py:339-346)
    for (int i = tid; i < N; i += 256) { so += dO[i]; sk += da2 ? 0.5 * ka * (double)da2[i] : ka * ((dA[i] ? 1.0 : 0.0) - 0.5); }
  for (int dl = 32; dl >= 1; dl >>= 1) { so += __shfl_down(so, dl); sk += __shfl_down(sk, dl); }
  if ((tid & 63) == 0) { s_w[tid >> 6][0] = so; s_w[tid >> 6][1] = sk; }
  __syncthreads();
  if (tid == 0) {
    const double sumO_diag = (s_w[0][0] + s_w[1][0]) + (s_w[2][0] + s_w[3][0]);
    const double sumK_diag = (s_w[0][1] + s_w[1][1]) + (s_w[2][1] + s_w[3][1]);
    double mu0 = *mu0p;
    if (do_mu0) {
      // eigenmodel.py:914-932
      double prec = 1.0 / sigma_mu0 + s_tot[0] + sumO_diag;
      double spsi = 0.0;
      for (int d = 0; d < D; ++d) spsi += lmbda[d] * s_tot[2 + d];
      double h = mu_mu_0 / sigma_mu0 + (s_tot[1] + sumK_diag) - spsi;
      const double var = 1.0 / prec;
      mu0 = var * h + gg_sqrt(var) * gg_normal_at(seed, GG_SITE_EIG_MU0, sweep, 0u);
      *mu0p = mu0;
    }
    if (do_lmbda) {
      // eigenmodel.py:938-965
      double prec[GG_MAXD * GG_MAXD], h[GG_MAXD], z[GG_MAXD], x[GG_MAXD];
      int e = 2 + 2 * DM;
      for (int a1 = 0; a1 < DM; ++a1)
        for (int b1 = a1; b1 < DM; ++b1) {
          double v = s_tot[e++];
          if (a1 < D && b1 < D) {
            if (a1 == b1) v += 1.0 / sigma_lmbda;
            prec[a1 * D + b1] = v;
            prec[b1 * D + a1] = v;
          }
        }
      for (int a1 = 0; a1 < D; ++a1) {
        h[a1] = mu_lmbda[a1] / sigma_lmbda + s_tot[2 + DM + a1] - mu0 * s_tot[2 + a1];
        z[a1] = gg_normal_at(seed, GG_SITE_EIG_LMBDA, sweep, (uint32_t)a1);
      }
      if (gg_gauss_from_precision(prec, h, z, D, x)) atomicOr(status, 1);
      else
        for (int a1 = 0; a1 < D; ++a1) lmbda[a1] = x[a1];
    }
  }
}

// eigenmodel.py:110-129 with P = clip(logistic(Mu), 1e-16, 1-1e-16) (:803-811): one partial per 32 x 32 tile.
template <int DM>
__global__ __launch_bounds__(256) void k_eig_loglik(int N, int Np, int D, const uint8_t* __restrict__ A,
                                                    const uint8_t* __restrict__ dA, const double* __restrict__ F,
                                                    const double* __restrict__ lmbda, const double* __restrict__ mu0p,
                                                    double* __restrict__ llpart, int probit) {
  __shared__ double s_fi[32][DM], s_fj[32][DM];
  __shared__ double s_psi[32][33];
  __shared__ double s_w[4];
  const int I = blockIdx.y, J = blockIdx.x, tid = threadIdx.x;
  const double mu0 = *mu0p;
  for (int e = tid; e < 32 * DM; e += 256) {
    int r = e / DM, d = e % DM;
    int gi = I * 32 + r, gj = J * 32 + r;
    s_fi[r][d] = (d < D && gi < N) ? F[(size_t)gi * D + d] * lmbda[d] : 0.0;
    s_fj[r][d] = (d < D && gj < N) ? F[(size_t)gj * D + d] : 0.0;
  }
  __syncthreads();
  eig_psi_tile_mfma<DM>(s_fi, s_fj, mu0, s_psi);
  __syncthreads();
  double ll = 0.0;
  for (int e = tid; e < 1024; e += 256) {
    const int li = e >> 5, lj = e & 31;
    const int gi = I * 32 + li, gj = J * 32 + lj;
    if (gi < N && gj < N) {
      double psi = mu0;
      int a;
      if (gi == gj) {
        a = dA[gi];
      } else {
        psi = s_psi[li][lj];
        a = A[(size_t)gi * Np + gj];
      }
      if (a == GG_A_MISSING) continue;                                       // eigenmodel.py:125-127: nan_to_num(ll) = 0
      double P = probit ? gg_norm_cdf(psi) : 1.0 / (1.0 + gg_exp(-psi));   // eigenmodel.py:228-237 / :803-811
      P = P < 1e-16 ? 1e-16 : (P > 1.0 - 1e-16 ? 1.0 - 1e-16 : P);
      ll += a ? gg_log(P) : gg_log(1.0 - P);
    }
  }
  for (int dl = 32; dl >= 1; dl >>= 1) ll += __shfl_down(ll, dl);
  if ((tid & 63) == 0) s_w[tid >> 6] = ll;
  __syncthreads();
  if (tid == 0) llpart[(size_t)I * gridDim.x + J] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// ------------------------------------------------------------------ host
template <typename T>
static int emalloc(T** p, size_t n) {
  GG_HIP(hipMalloc((void**)p, n * sizeof(T)));
  GG_HIP(hipMemset(*p, 0, n * sizeof(T)));
  return GG_OK;
}

int gg_eigen_create(gg_eigen** out, const gg_config* cfg) {
  if (cfg->N <= 0 || cfg->D < 1 || cfg->D > 8)
    return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel: need N > 0 and 1 <= D <= 8 (got N=%d D=%d)", cfg->N, cfg->D);
  if (cfg->N >= (1 << 20))
    return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel: N < 2^20 (the 40-bit Philox index of a pair of nodes, spec/gg_philox.h)");
  {
    const double v[4] = {cfg->sigma_mu0, cfg->sigma_F, cfg->sigma_lmbda, cfg->mu_mu_0};
    if (gg_first_nonfinite(v, 4) >= 0 || !(v[0] > 0.0 && v[1] > 0.0 && v[2] > 0.0))
      return gg_fail(GG_ERR_INVALID, "eigenmodel: prior variances (sigma_mu0, sigma_F, sigma_lmbda) = (%g, %g, %g) must be positive and finite, mu_mu_0 finite", v[0], v[1], v[2]);
  }
  gg_eigen* e = new gg_eigen();
  e->cfg = *cfg;
  e->N = cfg->N; e->D = cfg->D; e->Np = gg_round_up(cfg->N, 64);
  e->probit = cfg->kind == GG_PROBIT_EIG ? 1 : 0;
  e->beta = 1.0;
  e->DM = cfg->D <= 4 ? 4 : 8;
  e->NEG = e->D * (e->D + 1) / 2; e->NE = e->NEG + e->D;
  e->NG = 2 + 2 * e->DM + e->DM * (e->DM + 1) / 2;
  e->NEm = e->DM * (e->DM + 1) / 2 + e->DM;
  e->NCH = (e->N + EIG_CHN - 1) / EIG_CHN;
  int rc;
  GG_HIP(hipStreamCreate(&e->stream));
  GG_HIP(hipEventCreate(&e->e0));
  GG_HIP(hipEventCreate(&e->e1));
  {
    // second stream for the auxiliary draws of a full sweep, kept off every 32nd compute unit (8 of 256): the feature scan's
    // workgroup asks for ~130 KB of LDS and would otherwise wait until the draws' four-workgroups-per-CU stream happens to
    // drain one (a stream of small workgroups backfills every hole: DESIGN.md section 5).  GG_EIG_OVERLAP=0: the serial form.
    e->overlap = 0;
    // Measured on MI355X (round 4, profiles/r04_eigen_overlap_timeline.txt): exact, and NO FASTER - 4.95 ms per sweep at N=4096,
    // D=8 with 8 compute units per XCD kept free, 5.1-5.2 with 4, 5.4 with 2, against 5.0 ms serial - so it is off unless
    // GG_EIG_OVERLAP=1: while the draws run, a scan launch takes ~100 us instead of 42 (its latency chain is clock-bound and the
    // chip is busy), and the draws, chunked and on fewer compute units, take 3.4 instead of 2.0 ms.
    const char* eo = getenv("GG_EIG_OVERLAP");
    if (eo && atoi(eo) == 1) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
        const int ncu = prop.multiProcessorCount;
        const int nw = (ncu + 31) / 32;
        std::vector<uint32_t> mask(nw, 0xffffffffu);
        // Bit i of the mask is compute unit i / 8 of XCD i % 8 (tools/probe/cu_mask.hip, measured: bits 0..7 = one CU in each
        // of the 8 XCDs; an XCD whose bits are ALL clear is not masked at all).  GG_EIG_RESERVE = r: the first r compute units
        // of every XCD stay free of the draws (default 4: 32 of 256).
        const char* er = getenv("GG_EIG_RESERVE");
        int r = er && atoi(er) > 0 ? atoi(er) : 4;
        if (r > 24) r = 24;
        for (int b = 0; b < 8 * r && b < ncu; ++b) mask[b / 32] &= ~(1u << (b % 32));
        if (ncu % 32) mask[nw - 1] &= (1u << (ncu % 32)) - 1u;
        if (hipExtStreamCreateWithCUMask(&e->stream2, (uint32_t)nw, mask.data()) == hipSuccess) e->overlap = 1;
        else (void)hipGetLastError();
      }
      if (e->overlap) {
        bool ok = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 16 && ok; ++i) ok = hipEventCreateWithFlags(&e->ev_chunk[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); e->overlap = 0; }
      }
    }
  }
  const size_t N = e->N, D = e->D;
  if ((rc = emalloc(&e->F, N * D)) || (rc = emalloc(&e->lmbda, D)) || (rc = emalloc(&e->mu0, 1)) ||
      (rc = emalloc(&e->mu_lmbda, D)) || (rc = emalloc(&e->S, N * e->Np)) || (rc = emalloc(&e->dO, N)) ||
      (rc = emalloc(&e->Z, N * D)) || (rc = emalloc(&e->part, (size_t)2 * EIG_T * e->NCH * e->NEm)) ||
      (rc = emalloc(&e->Sb, N * EIG_VW)) || (rc = emalloc(&e->Vb, N * EIG_VW)) ||
      (rc = emalloc(&e->gpart, N * e->NG)) || (rc = emalloc(&e->status, 1)) || (rc = emalloc(&e->dbgbuf, 16)))
    return rc;
  const size_t nt = (N + 31) / 32;
  if ((rc = emalloc(&e->llpart, nt * nt))) return rc;
  *out = e;
  return GG_OK;
}

int gg_eigen_destroy(gg_eigen* e) {
  if (!e) return GG_OK;
  void* ptrs[] = {e->F, e->lmbda, e->mu0, e->mu_lmbda, e->S, e->dO, e->Z, e->Sb, e->Vb, e->part, e->gpart, e->llpart, e->status, e->dbgbuf, e->a2, e->nob, e->da2};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (e->e0) (void)hipEventDestroy(e->e0);
  if (e->e1) (void)hipEventDestroy(e->e1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->stream2) (void)hipStreamDestroy(e->stream2);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  for (int i = 0; i < 16; ++i) if (e->ev_chunk[i]) (void)hipEventDestroy(e->ev_chunk[i]);
  delete e;
  return GG_OK;
}

static int eig_state_ptr(gg_eigen* e, int id, void** p, size_t* nbytes) {
  *p = nullptr;
  switch (id) {
    case GG_ST_F: *p = e->F; *nbytes = (size_t)e->N * e->D * 8; break;
    case GG_ST_LMBDA: *p = e->lmbda; *nbytes = (size_t)e->D * 8; break;
    case GG_ST_MU0: *p = e->mu0; *nbytes = 8; break;
    case GG_ST_MU_LMBDA: *p = e->mu_lmbda; *nbytes = (size_t)e->D * 8; break;
    default: break;
  }
  return *p ? GG_OK : GG_ERR_INVALID;
}

int gg_eigen_set_state(gg_eigen* e, int id, const void* buf, size_t nbytes) {
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    e->sweep = *(const uint32_t*)buf;
    e->parts_done = 0;
    return GG_OK;
  }
  void* p; size_t need;
  if (eig_state_ptr(e, id, &p, &need)) return gg_fail(GG_ERR_INVALID, "state id %d not valid for the eigenmodel", id);
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  {
    const long long bad = gg_first_nonfinite((const double*)buf, nbytes / sizeof(double));   // (all eigenmodel state is doubles)
    if (bad >= 0) return gg_fail(GG_ERR_INVALID, "state id %d: non-finite value at index %lld", id, bad);
  }
  GG_HIP(hipMemcpyAsync(p, buf, nbytes, hipMemcpyHostToDevice, e->stream));
  GG_HIP(hipStreamSynchronize(e->stream));
  return GG_OK;
}

int gg_eigen_get_state(gg_eigen* e, int id, void* buf, size_t nbytes) {
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    *(uint32_t*)buf = e->sweep;
    return GG_OK;
  }
  if (id == GG_ST_OMEGA_S) {
    // symmetrised auxiliary variables: off-diagonal omega_ij + omega_ji, diagonal omega_ii
    const size_t N = e->N;
    if (nbytes != N * N * 8) return gg_fail(GG_ERR_INVALID, "GG_ST_OMEGA_S expects N*N doubles");
    std::vector<double> hS(N * e->Np), hd(N);
    GG_HIP(hipMemcpy(hS.data(), e->S, hS.size() * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(hd.data(), e->dO, N * 8, hipMemcpyDeviceToHost));
    double* o = (double*)buf;
    for (size_t i = 0; i < N; ++i)
      for (size_t j = 0; j < N; ++j) o[i * N + j] = (i == j) ? hd[i] : hS[i * e->Np + j];
    return GG_OK;
  }
  void* p; size_t need;
  if (eig_state_ptr(e, id, &p, &need)) return gg_fail(GG_ERR_INVALID, "state id %d not valid for the eigenmodel", id);
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  GG_HIP(hipMemcpyAsync(buf, p, nbytes, hipMemcpyDeviceToHost, e->stream));
  GG_HIP(hipStreamSynchronize(e->stream));
  return GG_OK;
}

#define EIG_DISPATCH(CALL)                 \
  do {                                     \
    if (e->DM == 4) { CALL(4); }           \
    else { CALL(8); }                      \
  } while (0)

void gg_eigen_set_beta(gg_eigen* e, double beta) { e->beta = beta; }

int gg_eigen_resample(gg_eigen* e, const gg_data* d, int parts) {
  return gg_eigen_resample_many(e, &d, 1, parts);
}

// resample(networks) (eigenmodel.py:836-857, :246-272): auxiliary variables for every graph of the list, then F, mu_0 and
// lambda from the statistics summed over the graphs (every conditional is linear in them); entries given as GG_A_MISSING
// are unobserved and take part in nothing.  One fully observed graph is the plain sweep.
int gg_eigen_resample_many(gg_eigen* e, const gg_data* const* ds, int nd, int parts) {
  if (!ds || nd < 1 || !ds[0]) return gg_fail(GG_ERR_INVALID, "null data handle");
  // (a2 sums 2 A - 1 over the graphs and BOTH directions of a pair: |a2| <= 2 nd must fit int8, nob <= 2 nd uint8)
  if (nd > 63) return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel: at most 63 graphs per model (the summed observations of a pair, two per graph, are kept as int8)");
  bool general = nd > 1;
  for (int g = 0; g < nd; ++g) {
    if (!ds[g]) return gg_fail(GG_ERR_INVALID, "null data handle");
    if (ds[g]->N != e->N) return gg_fail(GG_ERR_INVALID, "data has N=%d, model has N=%d", ds[g]->N, e->N);
    if (ds[g]->n0 != 0 || ds[g]->n1 != e->N) return gg_fail(GG_ERR_INVALID, "eigenmodel: whole networks only, not column shards");
    if (ds[g]->device != e->cfg.device) return gg_fail(GG_ERR_INVALID, "data on device %d, model on device %d", ds[g]->device, e->cfg.device);
    if (ds[g]->n_missing) general = true;
  }
  const gg_data* d = ds[0];
  const int N = e->N, D = e->D, Np = e->Np;
  hipStream_t st = e->stream;
  // a part asked for a second time under the same sweep index starts a new one: fresh Philox counters (graphgibbs.h)
  if (parts & e->parts_done) { e->sweep += 1; e->parts_done = 0; }
  GG_HIP(hipEventRecord(e->e0, st));
  const bool full = (parts & (GG_PART_OMEGA | GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA)) ==
                    (GG_PART_OMEGA | GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA);
  if (general && !e->a2) {
    int rc;
    if ((rc = emalloc(&e->a2, (size_t)N * Np)) || (rc = emalloc(&e->nob, (size_t)N * Np)) || (rc = emalloc(&e->da2, (size_t)N))) return rc;
  }
  // Overlapped form (round 4; one fully observed graph, a whole sweep): the auxiliaries are drawn on the second stream in
  // chunks of tile rows, top to bottom - row t of S is complete once the tile rows 0..t of the upper triangle are done -,
  // each chunk followed by its nodes' tile-local blocks and an event; the feature scan on the main stream waits, tile by
  // tile, only for the chunk that holds the rows its riders read next.  The draws (the whole chip, divergence-bound) and the
  // scan (one workgroup's latency chain) were 2.0 + 2.7 ms in series; the scan now starts after the first chunk.
  const int nt32 = (N + 31) / 32;
  int nchunk = 0, rows_per_chunk = 0;
  const bool ovl = e->overlap && full && !general && nd == 1 && N >= 1024;
  if (ovl) {
    nchunk = 8;
    rows_per_chunk = ((nt32 + nchunk - 1) / nchunk + 1) & ~1;          // (even: a 64-node tile is two rows)
    nchunk = (nt32 + rows_per_chunk - 1) / rows_per_chunk;
    GG_HIP(hipEventRecord(e->ev_fork, st));
    GG_HIP(hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
    hipStream_t sb = e->stream2;
    const gg_data* dg = ds[0];
    const uint32_t site = (uint32_t)(e->probit ? GG_SITE_EIG_Z : GG_SITE_EIG_OMEGA);
    eig_obs obv;
    obv.A = dg->A; obv.AT = dg->AT; obv.a2 = e->a2; obv.nob = e->nob; obv.gen = 0;
    for (int c = 0; c < nchunk; ++c) {
      const int I0 = c * rows_per_chunk, I1 = (I0 + rows_per_chunk < nt32) ? I0 + rows_per_chunk : nt32;
      const long long p0 = (long long)I0 * nt32 - (long long)I0 * (I0 - 1) / 2, p1 = (long long)I1 * nt32 - (long long)I1 * (I1 - 1) / 2;
      const int np = (int)(p1 - p0), pb = (int)p0;
      if (e->probit) {
#define CALL(DMV) k_eig_z<DMV><<<np, 256, 0, sb>>>(N, Np, D, e->F, e->lmbda, e->mu0, dg->A, dg->AT, e->cfg.seed, e->sweep, e->S, 0, site, pb)
        EIG_DISPATCH(CALL);
#undef CALL
      } else {
#define CALL(DMV)                                                                                                                  \
  if (e->beta != 1.0) k_eig_omega<DMV, true><<<np, 256, 0, sb>>>(N, Np, D, e->F, e->lmbda, e->mu0, e->cfg.seed, e->sweep, e->S, e->dO, e->beta, dg->A, dg->AT, dg->dA, 0, 0, site, pb); \
  else k_eig_omega<DMV, false><<<np, 256, 0, sb>>>(N, Np, D, e->F, e->lmbda, e->mu0, e->cfg.seed, e->sweep, e->S, e->dO, e->beta, dg->A, dg->AT, dg->dA, 0, 0, site, pb)
        EIG_DISPATCH(CALL);
#undef CALL
      }
      const int n0c = I0 * 32, n1c = I1 * 32 < N ? I1 * 32 : N;
      k_eig_vblocks<<<n1c - n0c, 64, 0, sb>>>(N, Np, e->S, obv, e->mu0, e->Sb, e->Vb, e->probit, e->beta, e->cfg.seed, e->sweep, n0c);
      GG_HIP(hipGetLastError());
      GG_HIP(hipEventRecord(e->ev_chunk[c], sb));
    }
    e->have_omega = true;
    e->gen = 0;
    e->omega_uid = d->uid;
    e->omega_uids.assign(1, d->uid);
  }
  if ((parts & GG_PART_OMEGA) && !ovl) {
    const int nt = (N + 31) / 32;
    const int npairs = nt * (nt + 1) / 2;
    for (int g = 0; g < nd; ++g) {
      const gg_data* dg = ds[g];
      const int acc = g > 0 ? 1 : 0;
      // graph g draws from its own Philox site (the plain site for the first: one graph is the plain sweep)
      const uint32_t site = (uint32_t)(e->probit ? GG_SITE_EIG_Z : GG_SITE_EIG_OMEGA) + ((uint32_t)g << 8);
      if (e->probit) {
#define CALL(DMV) k_eig_z<DMV><<<npairs, 256, 0, st>>>(N, Np, D, e->F, e->lmbda, e->mu0, dg->A, dg->AT, e->cfg.seed, e->sweep, e->S, acc, site, 0)
        EIG_DISPATCH(CALL);
#undef CALL
      } else {
#define CALL(DMV)                                                                                                                  \
  if (e->beta != 1.0) k_eig_omega<DMV, true><<<npairs, 256, 0, st>>>(N, Np, D, e->F, e->lmbda, e->mu0, e->cfg.seed, e->sweep, e->S, e->dO, e->beta, dg->A, dg->AT, dg->dA, general ? 1 : 0, acc, site, 0); \
  else k_eig_omega<DMV, false><<<npairs, 256, 0, st>>>(N, Np, D, e->F, e->lmbda, e->mu0, e->cfg.seed, e->sweep, e->S, e->dO, e->beta, dg->A, dg->AT, dg->dA, general ? 1 : 0, acc, site, 0)
        EIG_DISPATCH(CALL);
#undef CALL
      }
      if (general) {
        const size_t cells = (size_t)N * Np;
        k_eig_obs<<<(unsigned)((cells + 255) / 256), 256, 0, st>>>(N, Np, dg->A, dg->AT, dg->dA, e->a2, e->nob, e->da2, acc ? 0 : 1);
      }
    }
    e->have_omega = true;
    e->gen = general ? 1 : 0;
    e->omega_uid = nd == 1 ? d->uid : 0;
    e->omega_uids.resize(nd);
    for (int g = 0; g < nd; ++g) e->omega_uids[g] = ds[g]->uid;
  }
  if ((parts & (GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA)) && !e->have_omega)
    return gg_fail(GG_ERR_INVALID, "eigenmodel: resample F / mu_0 / lambda needs the auxiliary variables (GG_PART_OMEGA) first");
  if ((parts & (GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA)) && !(parts & GG_PART_OMEGA)) {
    //  the kappa terms of F / mu_0 / lambda come from THESE networks, the auxiliaries from the ones GG_PART_OMEGA was
    // called with: they must be the same list, network for network
    bool same = (general ? 1 : 0) == e->gen && (int)e->omega_uids.size() == nd;
    for (int g = 0; same && g < nd; ++g) same = e->omega_uids[g] == ds[g]->uid;
    if (!same) return gg_fail(GG_ERR_INVALID, "eigenmodel: the auxiliary variables were drawn for a different list of networks");
  }
  eig_obs ob;
  ob.A = d->A; ob.AT = d->AT; ob.a2 = e->a2; ob.nob = e->nob; ob.gen = e->gen;
  const int8_t* da2 = e->gen ? e->da2 : nullptr;
  if (parts & GG_PART_F) {
    k_eig_normals<<<(N * D + 255) / 256, 256, 0, st>>>(e->Z, N * D, e->cfg.seed, GG_SITE_EIG_F, e->sweep);
    static int dbgon = -1;
    if (dbgon < 0) { const char* ev = getenv("GG_EIG_DBG"); dbgon = ev ? atoi(ev) : 0; }
    long long* dbg = dbgon ? e->dbgbuf : nullptr;
    if (dbg) GG_HIP(hipMemsetAsync(dbg, 0, 16 * sizeof(long long), st));
    const int DMv = e->DM, nth = 64 * DMv;
    const size_t chain_lds = (((size_t)2 * EIG_T * (EIG_T + 1) + 2 * (size_t)EIG_T * e->NEm + 3 * (size_t)EIG_T * DMv + nth + 2) * sizeof(double) + 16 + 15) & ~(size_t)15;
    static bool attr4 = false, attr8 = false;
    if (DMv == 4 && !attr4) {
      GG_HIP(hipFuncSetAttribute((const void*)k_eig_chain<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
      attr4 = true;
    }
    if (DMv == 8 && !attr8) {
      GG_HIP(hipFuncSetAttribute((const void*)k_eig_chain<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
      attr8 = true;
    }
    // the tile-local blocks of pair weights and mean coefficients (previous tile | own tile) for every node
    // (overlapped form: made chunk by chunk on the second stream, behind the draws they read)
    if (!ovl) k_eig_vblocks<<<N, 64, 0, st>>>(N, Np, e->S, ob, e->mu0, e->Sb, e->Vb, e->probit, e->beta, e->cfg.seed, e->sweep, 0);
    int waited = -1;        // chunks of the second stream the main stream has waited for
    auto need_rows = [&](int row) -> int {      // the main stream may read the rows 0..row of S (32-row units) from here on
      if (!ovl) return GG_OK;
      int c = (row < nt32 ? row : nt32 - 1) / rows_per_chunk;
      if (c > waited) { GG_HIP(hipStreamWaitEvent(st, e->ev_chunk[c], 0)); waited = c; }
      return GG_OK;
    };
    {
      static int serial_b = -1;      // GG_EIG_SERIAL_B=1 (measurement): the draws on the second stream, the scan only behind ALL of them
      if (serial_b < 0) { const char* ev = getenv("GG_EIG_SERIAL_B"); serial_b = ev ? atoi(ev) : 0; }
      int rcw = need_rows(serial_b ? nt32 - 1 : 1);
      if (rcw) return rcw;
    }
    // Tile 0's statistics in a launch of their own; every chain launch then carries, as extra workgroups, the
    // statistics of the tile after it (all nodes outside both tiles), and adds the previous tile's nodes itself.
    const size_t part_stride = (size_t)EIG_T * e->NCH * e->NEm;
    {
      dim3 grd(N < EIG_T ? N : EIG_T, e->NCH);
#define CALL(DMV) k_eig_tile_stats<DMV><<<grd, 64 * DMV, 0, st>>>(N, Np, D, 0, e->S, ob, e->F, e->lmbda, e->mu0, e->NCH, e->part, 0, e->probit, e->beta, e->cfg.seed, e->sweep)
      EIG_DISPATCH(CALL);
#undef CALL
    }
    int ti = 0;
    for (int t0 = 0; t0 < N; t0 += EIG_T, ++ti) {
      double* partb = e->part + (size_t)(ti & 1) * part_stride;
      double* partn = e->part + (size_t)((ti + 1) & 1) * part_stride;
      const int Tn = (t0 + EIG_T < N) ? ((N - t0 - EIG_T < EIG_T) ? (N - t0 - EIG_T) : EIG_T) : 0;
      const int grid = 1 + Tn * e->NCH;
      { int rcw = need_rows((t0 + 2 * EIG_T) / 32 - 1); if (rcw) return rcw; }     // (the riders read the rows of the NEXT tile)
#define CALL(DMV)                                                                                                      \
  k_eig_chain<DMV><<<grid, 64 * DMV, chain_lds, st>>>(N, Np, D, t0, e->S, ob, e->F, e->lmbda, e->mu0,                   \
                                                      e->cfg.sigma_F, e->NCH, partb, e->Z, e->Sb, e->Vb, e->status, dbg, \
                                                      partn, e->probit, e->beta, e->cfg.seed, e->sweep)
      EIG_DISPATCH(CALL);
#undef CALL
    }
  }
  const int do_mu0 = (parts & GG_PART_MU0) ? 1 : 0;
  const int do_lm = ((parts & GG_PART_LMBDA) && !e->cfg.lmbda_given) ? 1 : 0;
  if (ovl) GG_HIP(hipStreamWaitEvent(st, e->ev_chunk[nchunk - 1], 0));       // (everything the second stream did)
  if (do_mu0 || do_lm) {
#define CALL(DMV)                                                                                                   \
  k_eig_global<DMV><<<(N < 512 ? N : 512), 256, 0, st>>>(N, Np, D, e->S, ob, e->F, e->gpart, e->probit, e->beta);             \
  k_eig_draw_globals<DMV><<<1, 256, 0, st>>>(N, (N < 512 ? N : 512), D, e->gpart, e->dO, d->dA, da2, e->lmbda, e->mu0, e->mu_lmbda,            \
                                             e->cfg.mu_mu_0, e->cfg.sigma_mu0, e->cfg.sigma_lmbda, do_mu0, do_lm,   \
                                             e->cfg.seed, e->sweep, e->status, e->probit, e->beta)
    EIG_DISPATCH(CALL);
#undef CALL
  }
  GG_HIP(hipGetLastError());
  GG_HIP(hipEventRecord(e->e1, st));
  int stv = 0;
  GG_HIP(hipMemcpyAsync(&stv, e->status, sizeof(int), hipMemcpyDeviceToHost, st));
  GG_HIP(hipStreamSynchronize(st));
  float ms = 0;
  GG_HIP(hipEventElapsedTime(&ms, e->e0, e->e1));
  e->st.total_ms = ms;
  if (getenv("GG_EIG_DBG") && (parts & GG_PART_F)) {
    long long h[16];
    (void)hipMemcpy(h, e->dbgbuf, sizeof(h), hipMemcpyDeviceToHost);
    if (h[3]) fprintf(stderr, "[eig chain] per node: scan %.0f shader clocks (%lld nodes); per tile: prologue %.0f clocks (of which statistics + products %.0f)\n", (double)h[2] / h[3], h[3], (double)h[5] / ((h[3] + EIG_T - 1) / EIG_T), (double)h[4] / ((h[3] + EIG_T - 1) / EIG_T));
  }
  if (stv) {
    int z = 0;
    (void)hipMemcpy(e->status, &z, sizeof(int), hipMemcpyHostToDevice);
    return gg_fail(GG_ERR_NUMERIC, "eigenmodel: a posterior precision is not positive definite");
  }
  if (full) { e->sweep += 1; e->st.sweeps += 1; e->parts_done = 0; }
  else e->parts_done |= parts;
  return GG_OK;
}

int gg_eigen_log_likelihood(gg_eigen* e, const gg_data* d, double* out) {
  if (!d || !out) return gg_fail(GG_ERR_INVALID, "null argument");
  if (d->N != e->N) return gg_fail(GG_ERR_INVALID, "data has N=%d, model has N=%d", d->N, e->N);
  const int N = e->N, nt = (N + 31) / 32;
  dim3 grd(nt, nt);
#define CALL(DMV) k_eig_loglik<DMV><<<grd, 256, 0, e->stream>>>(N, e->Np, e->D, d->A, d->dA, e->F, e->lmbda, e->mu0, e->llpart, e->probit)
  EIG_DISPATCH(CALL);
#undef CALL
  GG_HIP(hipGetLastError());
  std::vector<double> h((size_t)nt * nt);
  GG_HIP(hipMemcpyAsync(h.data(), e->llpart, h.size() * 8, hipMemcpyDeviceToHost, e->stream));
  GG_HIP(hipStreamSynchronize(e->stream));
  double ll = 0.0;
  for (double v : h) ll += v;
  *out = ll;
  return GG_OK;
}

int gg_eigen_stats(gg_eigen* e, gg_stats* out) {
  *out = e->st;
  return GG_OK;
}
