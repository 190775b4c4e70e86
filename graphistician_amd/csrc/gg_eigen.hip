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

#define EIG_T 64          // nodes per tile of the F scan
#define EIG_CHN 4096      // neighbours per workgroup in k_eig_tile_stats (1024 threads x 4)

struct gg_eigen {
  gg_config cfg;
  int N, D, Np, DM;
  int probit;            // GG_PROBIT_EIG: buffer S holds Z_ij + Z_ji, the pair weight is the constant 2
  double beta;           // AIS inverse temperature (logistic link only): omega ~ PG(beta, psi), kappa = beta (A - 1/2)
  int NEG, NE, NG;        // D(D+1)/2, NEG + D, 2 + 2D + NEG
  int NCH;
  double *F, *lmbda, *mu0, *mu_lmbda;
  double *S, *dO, *Z;
  double *part, *gpart, *llpart;
  int* status;
  long long* dbgbuf;   // GG_EIG_DBG: cycle counters of the chain's solver wave
  hipStream_t stream;
  uint32_t sweep;
  int parts_done;          // parts already drawn under the current sweep index (step-wise calls, see gg_resample_part)
  bool have_omega;
  hipEvent_t e0, e1;
  gg_stats st;
};

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
template <int DM>
__global__ __launch_bounds__(256) void k_eig_omega(int N, int Np, int D, const double* __restrict__ F,
                                                   const double* __restrict__ lmbda, const double* __restrict__ mu0p,
                                                   uint64_t seed, uint32_t sweep, double* __restrict__ S,
                                                   double* __restrict__ dO, double beta) {
  __shared__ double s_fi[32][DM], s_fj[32][DM];
  __shared__ double s_t[32][33];
  __shared__ double s_psi[32][33];
  const int nt = (N + 31) / 32;
  int I, J;
  eig_tile_pair(blockIdx.x, nt, &I, &J);
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
    if (gi < N && gj < N) {
      if (beta != 1.0) {
        // tempered likelihood (AIS; build-defined, spec/gg_pg.h gg_pg_frac): omega ~ PG(beta, psi), nothing at beta = 0
        if (I == J && li == lj) {
          gg_stream s0;
          gg_stream_init(&s0, seed, GG_SITE_EIG_OMEGA, sweep, (uint32_t)gi * (uint32_t)N + (uint32_t)gi);
          dO[gi] = beta > 0.0 ? gg_pg_frac(&s0, beta, mu0) : 0.0;
        } else if (I < J || li < lj) {
          const double psi = s_psi[li][lj];
          gg_stream s1, s2;
          gg_stream_init(&s1, seed, GG_SITE_EIG_OMEGA, sweep, (uint32_t)gi * (uint32_t)N + (uint32_t)gj);
          gg_stream_init(&s2, seed, GG_SITE_EIG_OMEGA, sweep, (uint32_t)gj * (uint32_t)N + (uint32_t)gi);
          val = beta > 0.0 ? gg_pg_frac(&s1, beta, psi) + gg_pg_frac(&s2, beta, psi) : 0.0;
        }
      } else if (I == J && li == lj) {
        dO[gi] = gg_pg1_at(seed, sweep, (uint32_t)N, (uint32_t)gi, (uint32_t)gi, mu0);
      } else if (I < J || li < lj) {
        const double psi = s_psi[li][lj];
        // omega_ij and omega_ji see the same psi: the mixture mass is computed once for the pair
        const double Z = 0.5 * gg_fabs(psi), ptex = gg_pg_mass_texpon(Z);
        gg_stream s1, s2;
        gg_stream_init(&s1, seed, GG_SITE_EIG_OMEGA, sweep, (uint32_t)gi * (uint32_t)N + (uint32_t)gj);
        gg_stream_init(&s2, seed, GG_SITE_EIG_OMEGA, sweep, (uint32_t)gj * (uint32_t)N + (uint32_t)gi);
        val = gg_pg1_pre(&s1, Z, ptex) + gg_pg1_pre(&s2, Z, ptex);
      }
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
    if (gi < N && gj < N) S[(size_t)gi * Np + gj] = s_t[li][lj];
  }
  if (I < J) {
    for (int e = tid; e < 1024; e += 256) {
      const int lj = e >> 5, li = e & 31;   // transposed tile, coalesced over li
      const int gi = I * 32 + li, gj = J * 32 + lj;
      if (gi < N && gj < N) S[(size_t)gj * Np + gi] = s_t[li][lj];
    }
  }
}

// Probit auxiliaries (eigenmodel.py:274-293): Z_ij ~ N(psi_ij, 1) truncated to Z > 0 where A_ij = 1 and Z < 0 where
// A_ij = 0.  Z enters the conditionals only through Z_ij + Z_ji off the diagonal (:312-321, :339-346, :363-372),
// which is what this kernel stores in S (symmetric, diagonal 0): same tiling as k_eig_omega.
template <int DM>
__global__ __launch_bounds__(256) void k_eig_z(int N, int Np, int D, const double* __restrict__ F,
                                               const double* __restrict__ lmbda, const double* __restrict__ mu0p,
                                               const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                               uint64_t seed, uint32_t sweep, double* __restrict__ S) {
  __shared__ double s_fi[32][DM], s_fj[32][DM];
  __shared__ double s_t[32][33];
  __shared__ double s_psi[32][33];
  const int nt = (N + 31) / 32;
  int I, J;
  eig_tile_pair(blockIdx.x, nt, &I, &J);
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
      const double z1 = gg_truncnorm_at(seed, sweep, (uint32_t)N, (uint32_t)gi, (uint32_t)gj, psi, A[at] ? 1 : 0);
      const double z2 = gg_truncnorm_at(seed, sweep, (uint32_t)N, (uint32_t)gj, (uint32_t)gi, psi, AT[at] ? 1 : 0);
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
    if (gi < N && gj < N) S[(size_t)gi * Np + gj] = s_t[li][lj];
  }
  if (I < J) {
    for (int e = tid; e < 1024; e += 256) {
      const int lj = e >> 5, li = e & 31;   // transposed tile, coalesced over li
      const int gi = I * 32 + li, gj = J * 32 + lj;
      if (gi < N && gj < N) S[(size_t)gj * Np + gi] = s_t[li][lj];
    }
  }
}

// Gram statistics of the tile's nodes against the current F (eigenmodel.py:894-901):
//   G_n[a,b] = sum_{nn != n} S[n,nn] ft_a ft_b,  h_n[a] = sum_{nn != n} ((A[n,nn]+A[nn,n]-1) - mu_0 S[n,nn]) ft_a,
// ft = F[nn] * lambda.  One workgroup per (node, chunk of EIG_CHN neighbours).
template <int DM>
__device__ __forceinline__ void eig_tile_stats_body(int N, int Np, int D, int t0, const double* __restrict__ S,
                                                        const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                                        const double* __restrict__ F, const double* __restrict__ lmbda,
                                                        const double* __restrict__ mu0p, int NCH,
                                                        double* __restrict__ part, int ex_lo, int ex_hi, int node, int c,
                                                        int probit, double ka) {
  // nodes [ex_lo, ex_hi) (the previous tile, whose scan is still running) are left out: the chain adds them
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
  const int lo = c * EIG_CHN, hi = (lo + EIG_CHN < N) ? lo + EIG_CHN : N;
  for (int nn = lo + threadIdx.x; nn < hi; nn += nth) {
    if (nn == n || (nn >= ex_lo && nn < ex_hi)) continue;
    // logistic: s = omega_ij + omega_ji, a = kappa_ij + kappa_ji; probit (eigenmodel.py:312-321): s = 2, a = Z_ij + Z_ji
    const double raw = S[(size_t)n * Np + nn];
    const double s = probit ? 2.0 : raw;
    const double a = probit ? raw : ka * (double)((int)A[(size_t)n * Np + nn] + (int)AT[(size_t)n * Np + nn] - 1);
    const double v = a - mu0 * s;
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
  __shared__ double s_red[16][NEGM + DM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < NEGM + DM; ++e) {
    double v = e < NEGM ? g[e] : h[e - NEGM];
    for (int dl = 32; dl >= 1; dl >>= 1) v += __shfl_down(v, dl);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  // compact layout: entries of the D x D upper triangle then h, D-dependent
  const int NEG = D * (D + 1) / 2;
  if (threadIdx.x < NEGM + DM) {
    const int e = threadIdx.x;
    double v = 0.0;
    for (int w = 0; w < (nth >> 6); ++w) v += s_red[w][e];
    // map padded index -> compact index
    int out = -1;
    if (e < NEGM) {
      int a1 = 0, rem = e;
      while (rem >= DM - a1) { rem -= DM - a1; ++a1; }
      int b1 = a1 + rem;
      if (b1 < D) out = a1 * D - a1 * (a1 - 1) / 2 + (b1 - a1);
    } else if (e - NEGM < D) {
      out = NEG + (e - NEGM);
    }
    if (out >= 0) part[((size_t)node * NCH + c) * (NEG + D) + out] = v;
  }
}

// The sequential part of _resample_F for one tile (eigenmodel.py:884-908), one workgroup.
// Everything the per-node loop touches is staged in LDS first (T x T block of S and of the
// symmetrised adjacency, the tile's statistics, normals and old features).
template <int DM>
__global__ __launch_bounds__(1024) void k_eig_tile_stats(int N, int Np, int D, int t0, const double* __restrict__ S,
                                                        const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                                        const double* __restrict__ F, const double* __restrict__ lmbda,
                                                        const double* __restrict__ mu0p, int NCH,
                                                        double* __restrict__ part, int ex_lo, int ex_hi, int probit, double ka) {
  eig_tile_stats_body<DM>(N, Np, D, t0, S, A, AT, F, lmbda, mu0p, NCH, part, ex_lo, ex_hi, blockIdx.x, blockIdx.y, probit, ka);
}

// Order this wave's LDS operations: wait for the outstanding ones and keep the compiler from moving memory accesses across.
// (__threadfence_block() also waits for every global store in flight - hundreds of clocks on the scan's critical path.)
__device__ __forceinline__ void gg_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int DM, bool FULL>
__global__ __launch_bounds__(512) void k_eig_chain(int N, int Np, int Drt, int t0, const double* __restrict__ S,
                                                   const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                                   double* __restrict__ F, const double* __restrict__ lmbda,
                                                   const double* __restrict__ mu0p, double sigma_F, int NCH,
                                                   const double* __restrict__ part, const double* __restrict__ Z,
                                                   int* __restrict__ status, long long* __restrict__ dbg,
                                                   double* __restrict__ part_next, int probit, double ka) {
  constexpr int NEGM = DM * (DM + 1) / 2;
  if (blockIdx.x > 0) {
    // riders: the statistics of the NEXT tile (all nodes outside this tile and the next), computed while
    // workgroup 0 scans this one; the next launch's chain adds this tile's nodes with their new features
    const int idx = blockIdx.x - 1;
    eig_tile_stats_body<DM>(N, Np, Drt, t0 + EIG_T, S, A, AT, F, lmbda, mu0p, NCH, part_next, t0, t0 + EIG_T, idx / NCH, idx % NCH, probit, ka);
    return;
  }
  extern __shared__ __align__(16) unsigned char s_chain[];
  double* s_S = reinterpret_cast<double*>(s_chain);                 // [T][T+1]
  double* s_dP = s_S + EIG_T * (EIG_T + 1);                         // [T][NEGM]
  double* s_dF = s_dP + EIG_T * NEGM;                               // [T][DM]
  double* s_part = s_dF + EIG_T * DM;                               // [T][NEGM+DM] (compact NE used)
  double* s_Z = s_part + EIG_T * (NEGM + DM);                       // [T][DM]
  double* s_Fo = s_Z + EIG_T * DM;                                  // [T][DM]
  double* s_stat = s_Fo + EIG_T * DM;                               // [NEGM+DM]
  signed char* s_a = reinterpret_cast<signed char*>(s_stat + NEGM + DM);  // [T][T]
  const int tid = threadIdx.x;
  const int T = (N - t0 < EIG_T) ? (N - t0) : EIG_T;
  // FULL: D == DM, every bound below is a compile-time constant (no guards, no dynamic LDS indices)
  const int D = FULL ? DM : Drt;
  const int NEG = FULL ? NEGM : D * (D + 1) / 2, NE = NEG + D;
  const double mu0 = *mu0p;
  const long long c_start = dbg ? clock64() : 0;
  const int p0 = t0 - EIG_T;     // first node of the previous tile (left out by the statistics kernel)
  // block loader: rows of this tile x columns [c0, c0 + W) of S and of a = A + A^T - 1, into s_S / s_a
  auto load_block = [&](int c0, int W) {
    // (all of a thread's loads are issued before the first LDS store: 8 x 3 loads in flight instead of 3)
    constexpr int PER = EIG_T * EIG_T / 512;
    double sv[PER];
    int av[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 512;
      const int i = e / W, j = e % W;
      const bool in = e < T * W;
      const size_t at = in ? (size_t)(t0 + i) * Np + c0 + j : (size_t)t0 * Np + c0;
      sv[u] = S[at];
      av[u] = (int)A[at] + (int)AT[at] - 1;
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * 512;
      if (e < T * W) {
        const int i = e / W, j = e % W;
        s_S[i * (EIG_T + 1) + j] = sv[u];
        s_a[i * EIG_T + j] = (signed char)av[u];
      }
    }
  };
  if (p0 >= 0) load_block(p0, EIG_T); else load_block(t0, T);
  if (dbg && tid == 0) atomicAdd((unsigned long long*)&dbg[5], (unsigned long long)(clock64() - c_start));
  for (int e = tid; e < T * NE; e += 512) {
    int i = e / NE, q = e % NE;
    double acc = 0.0;
    for (int c = 0; c < NCH; ++c) acc += part[((size_t)i * NCH + c) * NE + q];
    s_part[i * (NEGM + DM) + q] = acc;
  }
  for (int e = tid; e < T * D; e += 512) {
    int i = e / D, q = e % D;
    s_Z[i * DM + q] = Z[(size_t)(t0 + i) * D + q];
    s_Fo[i * DM + q] = F[(size_t)(t0 + i) * D + q];
  }
  double lam[DM];
#pragma unroll
  for (int a1 = 0; a1 < DM; ++a1) lam[a1] = a1 < D ? lmbda[a1] : 0.0;
  if (dbg && tid == 0) atomicAdd((unsigned long long*)&dbg[6], (unsigned long long)(clock64() - c_start));
  // The statistics kernel ran while the previous tile was still being scanned and left its nodes out: add
  // them now, with their new features (s_dF is free until the scan starts: it holds f * lambda of those nodes).
  if (p0 >= 0) {
    for (int e = tid; e < EIG_T * D; e += 512) {
      int m = e / D, q = e % D;
      s_dF[m * DM + q] = F[(size_t)(p0 + m) * D + q] * lmbda[q];
    }
    __syncthreads();
    // C[i][e] = sum_m X[i][m] Y[m][e] over the 64 nodes m of the previous tile: a 64 x 64 x NE product, on the
    // matrix cores (v_mfma_f64_16x16x4_f64; fragment layout as in eig_psi_tile_mfma).  Precision entries
    // (e < NEG): X = S, Y[m][e] = g_m[a] g_m[b]; mean entries: X = a - mu_0 S, Y[m][q] = g_m[q].
    {
      const int wv = tid >> 6, l = tid & 63;
      const int ncb = (NEG + 15) / 16;            // column blocks of the precision part
      const int nblk = 4 * (ncb + 1);             // + one column block for the mean part
      for (int blk = wv; blk < nblk; blk += 8) {
        const int bi = blk & 3, cb = blk >> 2;
        const bool mean = cb == ncb;
        const int col = (mean ? 0 : cb * 16) + (l & 15);
        int a1 = 0, b1 = 0;
        bool cvalid;
        if (mean) { a1 = col; cvalid = col < D; }
        else { cvalid = col < NEG; if (cvalid) { int rem = col; while (rem >= D - a1) { rem -= D - a1; ++a1; } b1 = a1 + rem; } }
        const int row = bi * 16 + (l & 15);
        gg_double4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int kk = 0; kk < EIG_T / 4; ++kk) {
          const int m = kk * 4 + (l >> 4);
          const double raw = s_S[row * (EIG_T + 1) + m];
          const double sv = probit ? 2.0 : raw;
          const double xa = mean ? gg_fma(-mu0, sv, probit ? raw : ka * (double)s_a[row * EIG_T + m]) : sv;
          double yb = 0.0;
          if (cvalid) yb = mean ? s_dF[m * DM + a1] : s_dF[m * DM + a1] * s_dF[m * DM + b1];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, yb, acc, 0, 0, 0);
        }
        // lane l holds C[16 bi + 4 r + l / 16][col block + l % 16] in component r
        if (cvalid) {
          const int e = mean ? NEG + a1 : col;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = bi * 16 + 4 * r + (l >> 4);
            if (i < T) s_part[i * (NEGM + DM) + e] += acc[r];
          }
        }
      }
    }
    __syncthreads();
    if (dbg && tid == 0) atomicAdd((unsigned long long*)&dbg[7], (unsigned long long)(clock64() - c_start));
    load_block(t0, T);   // now the tile's own block, which the scan works on
  }
  // progress words (LDS): s_prog[0] = nodes whose rank-one change is published; s_prog[1 + w] = changes
  // updater wave w has applied
  volatile int* s_prog = reinterpret_cast<volatile int*>(s_a + EIG_T * EIG_T);
  if (tid < 16) s_prog[tid] = (tid == 2) ? 0xff : 0;   // bytes 8..15: one progress byte per updater wave (byte 8 unused: 255)
  __syncthreads();
  volatile unsigned char* s_progb = reinterpret_cast<volatile unsigned char*>(s_prog + 2);
  volatile unsigned long long* s_progq = reinterpret_cast<volatile unsigned long long*>(s_prog + 2);
  // The scan is a software pipeline inside the workgroup.  Wave 0 (the solver) runs the per-node critical
  // path only: fold the previous node's rank-one change into this node's statistics, solve the D x D
  // Gaussian conditional, publish the new rank-one change.  Waves 1-7 (updaters) push every published
  // change into the statistics of the nodes two or more positions later, in the background; each (node,
  // entry) is always owned by the same updater lane, so changes reach it in ascending order, and the solver
  // waits, before node nl, until every updater has applied change nl-2.  Accumulation order per entry is the
  // same as before: partial sums, then the earlier nodes of the tile in ascending order.
  const int wave = tid >> 6, lane = tid & 63;
  const double inv_sigma_F = 1.0 / sigma_F;
  if (dbg && tid == 0) atomicAdd((unsigned long long*)&dbg[4], (unsigned long long)(clock64() - c_start));
  if (wave == 0) {
    // The solver wave.  The per-node critical path (Cholesky + two triangular solves: ~200 dependent fp64 operations)
    // runs on lane 0; everything that is one independent operation per matrix entry -- folding the previous node's
    // rank-one change into the statistics, the new outer-product change -- is spread over the lanes, lane l owning
    // entry (l / DM, l % DM), and handed over through a small LDS scratch (same wave: program order, no barrier).
    // A wave instruction costs the same for 1 or 64 active lanes, so this takes ~250 instructions off the path.
    double* s_scr = s_stat;                        // [NEGM + DM] precision + mean term, then [2 DM] fn, fo
    const int li = lane / DM, lj = lane % DM;
    const int la = li < lj ? li : lj, lb = li < lj ? lj : li;
    const bool lact = lane < DM * DM && (FULL || lb < D);
    const int le = lact ? la * D - la * (la - 1) / 2 + (lb - la) : 0;   // compact upper-triangle index of (la, lb)
    const bool lup = lact && li <= lj;             // the lanes that write their entry
    const bool lvec = lane < D;                    // lanes that own a component of the mean term / of f
    long long c_wait = 0, c_solve = 0;
    double dPl = 0.0, dFl = 0.0;                   // the previous node's rank-one change, this lane's entry
    bool ready = true;                             // "every updater has applied change nl-2", checked one node ahead
    for (int nl = 0; nl < T; ++nl) {
      const int n = t0 + nl;
      const double* s_st = s_part + nl * (NEGM + DM);
      // pair weight and centred pair value of (this node, the previous one): tile constants, read before the wait
      double svp = 0.0, cfp = 0.0;
      if (nl >= 1) {
        const double raw = s_S[nl * (EIG_T + 1) + nl - 1];
        svp = probit ? 2.0 : raw;
        cfp = gg_fma(-mu0, svp, probit ? raw : ka * (double)s_a[nl * EIG_T + nl - 1]);
      }
      const long long c0 = dbg ? clock64() : 0;
      if (!ready) {
        for (int spin = 0;; ++spin) {
          const unsigned long long q = *s_progq;   // one load: the seven progress bytes
          int mn = 255;
#pragma unroll
          for (int w = 1; w <= 7; ++w) { const int v = (int)((q >> (8 * w)) & 255u); mn = v < mn ? v : mn; }
          if (mn >= nl - 1) break;
          if (spin > (1 << 22)) { if (lane == 0) atomicOr(status, 2); break; }   // never spin forever
          __builtin_amdgcn_s_sleep(1);
        }
      }
      gg_lds_wait();   // LDS traffic of one wave is performed in order: the data is written before the word that announces it
      const long long c1 = dbg ? clock64() : 0;
      // statistics of this node with the previous node's change folded in (same order as the updaters':
      // earlier changes first, the previous node's last), one entry per lane
      if (lup) {
        double v = gg_fma(svp, dPl, s_st[le]);
        if (li == lj) v += inv_sigma_F;
        s_scr[le] = v;
      }
      if (lvec) s_scr[NEG + lane] = gg_fma(cfp, dFl, s_st[NEG + lane]);
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      const long long cA = dbg ? clock64() : 0;
      if (lane == 0) {
        // D x D Gaussian conditional in registers (fully unrolled to DM; rows/cols >= D are identity)
        double L[DM][DM], hh[DM], x[DM];
        {
          int e = 0;
#pragma unroll
          for (int a1 = 0; a1 < DM; ++a1)
#pragma unroll
            for (int b1 = a1; b1 < DM; ++b1) {
              double v = 0.0;
              if (FULL || b1 < D) { v = s_scr[e]; ++e; }
              else if (a1 == b1) v = 1.0;
              L[b1][a1] = v;   // lower triangle holds the precision
            }
        }
#pragma unroll
        for (int a1 = 0; a1 < DM; ++a1) hh[a1] = (FULL || a1 < D) ? s_scr[NEG + a1] : 0.0;
        double zz[DM], fo[DM];
#pragma unroll
        for (int i = 0; i < DM; ++i) {
          zz[i] = (FULL || i < D) ? s_Z[nl * DM + i] : 0.0;
          fo[i] = (FULL || i < D) ? s_Fo[nl * DM + i] : 0.0;
        }
        if (dbg) { asm volatile("" ::"v"(L[0][0]), "v"(hh[0]), "v"(zz[0]), "v"(fo[0])); atomicAdd((unsigned long long*)&dbg[8], (unsigned long long)(clock64() - cA)); }
        const long long cB = dbg ? clock64() : 0;
        double dmin = 1.0;           // smallest pivot: a non-positive one voids the node (checked once, off the path)
        // Cholesky, RIGHT-looking: column j is scaled as soon as its pivot is known and its rank-one update is applied to
        // the trailing columns at once, so that the only serial chain left is pivot -> reciprocal square root -> scale ->
        // next pivot (one fma behind the scale); every other fma is independent work the scheduler can fill the chain's
        // bubbles with.  Per element the updates arrive in the same order (k ascending) as in the left-looking form
        // (gg_chol_d): the factor is the same to the bit.
#pragma unroll
        for (int j = 0; j < DM; ++j) {
          const double dsum = L[j][j];
          dmin = __builtin_fmin(dmin, dsum);
          // one reciprocal square root per column instead of D divisions (float-tolerance path)
          double inv = __builtin_amdgcn_rsq(dsum);               // hardware estimate, then two Newton steps
          inv = inv * gg_fma(-0.5 * dsum * inv, inv, 1.5);
          inv = inv * gg_fma(-0.5 * dsum * inv, inv, 1.5);
          L[j][j] = inv;           // the diagonal keeps 1 / L_jj: the solves only ever divide by it
#pragma unroll
          for (int i = j + 1; i < DM; ++i) L[i][j] = L[i][j] * inv;
#pragma unroll
          for (int i = j + 1; i < DM; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) L[i][k] = gg_fma(-L[i][j], L[k][j], L[i][k]);
        }
        if (dbg) { asm volatile("" ::"v"(L[DM - 1][DM - 1])); atomicAdd((unsigned long long*)&dbg[9], (unsigned long long)(clock64() - cB)); }
        const long long cC = dbg ? clock64() : 0;
        // y = L^{-1} h + z ; x = L^{-T} y   (gg_gauss_from_precision), both substitutions in their eager form: a finished
        // component is folded into every later right-hand side at once (forward: the same order per component as before)
        double y[DM];
#pragma unroll
        for (int i = 0; i < DM; ++i) {
          y[i] = gg_fma(hh[i], L[i][i], zz[i]);
          const double u = y[i] - zz[i];
#pragma unroll
          for (int r = i + 1; r < DM; ++r) hh[r] = gg_fma(-L[r][i], u, hh[r]);
        }
#pragma unroll
        for (int i = DM - 1; i >= 0; --i) {
          x[i] = y[i] * L[i][i];
#pragma unroll
          for (int r = 0; r < i; ++r) y[r] = gg_fma(-L[i][r], x[i], y[r]);
        }
        if (dbg) { asm volatile("" ::"v"(x[0])); atomicAdd((unsigned long long*)&dbg[10], (unsigned long long)(clock64() - cC)); }
        const bool ok = dmin > 0.0 && x[0] == x[0];   // x[0] closes the back substitution: NaN anywhere reaches it
        if (!ok) atomicOr(status, 1);
#pragma unroll
        for (int a1 = 0; a1 < DM; ++a1) {
          if (!ok) x[a1] = fo[a1];
          if (FULL || a1 < D) {
            // (the new feature stays in LDS, over the old one it replaces: the tile is written to global memory once, after
            // the scan - a global store here would sit in front of every later wait of this wave)
            s_Fo[nl * DM + a1] = x[a1];
            s_scr[a1] = x[a1] * lam[a1];            // fn
            s_scr[DM + a1] = fo[a1] * lam[a1];      // fo
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      // progress of the updaters for the NEXT node (needs change nl-1 applied everywhere): the words arrive while the
      // outer product below is formed
      const unsigned long long pq = *s_progq;
      // rank-one change of this node's outer product, seen by the later nodes of the tile: one entry per lane
      if (lact) {
        const double fna = s_scr[la], fnb = s_scr[lb], foa = s_scr[DM + la], fob = s_scr[DM + lb];
        dPl = fna * fnb - foa * fob;
        if (lup) s_dP[nl * NEGM + le] = dPl;
      }
      if (lvec) {
        dFl = s_scr[lane] - s_scr[DM + lane];
        s_dF[nl * DM + lane] = dFl;
      }
      {
        int mn = 255;
#pragma unroll
        for (int w = 1; w <= 7; ++w) { const int v = (int)((pq >> (8 * w)) & 255u); mn = v < mn ? v : mn; }
        ready = (nl + 1 < 2) || (mn >= nl);
      }
      gg_lds_wait();   // LDS traffic of one wave is performed in order: the data is written before the word that announces it
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) s_prog[0] = nl + 1;   // change nl is published
      if (dbg) { const long long c3 = clock64(); c_wait += c1 - c0; c_solve += c3 - c1; if (lane == 0) atomicAdd((unsigned long long*)&dbg[11], (unsigned long long)(cA - c1)); }
    }
    // the tile's new features -> global memory (nothing in this launch reads them: the riders leave this tile out)
    gg_lds_wait();
    for (int e = lane; e < T * D; e += 64) F[(size_t)(t0 + e / D) * D + e % D] = s_Fo[(e / D) * DM + e % D];
    if (dbg && lane == 0) { atomicAdd((unsigned long long*)&dbg[0], (unsigned long long)c_wait); atomicAdd((unsigned long long*)&dbg[2], (unsigned long long)c_solve); atomicAdd((unsigned long long*)&dbg[3], (unsigned long long)T); }
  } else {
    // updaters: every lane owns up to RU fixed (node, entry) pairs and keeps their running values in registers;
    // per change it needs two or three independent LDS reads per pair (issued as one batch), one fma, one store
    constexpr int NU = 512 - 64;
    constexpr int RU = (EIG_T * (NEGM + DM) + NU - 1) / NU;
    const int u = tid - 64;
    int ui[RU], ue[RU];
    double val[RU];
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      const int idx = u + NU * r;
      const bool v = idx < T * NE;
      ui[r] = v ? idx / NE : -1;
      ue[r] = v ? idx % NE : 0;
      val[r] = v ? s_part[ui[r] * (NEGM + DM) + ue[r]] : 0.0;
    }
    for (int k = 0; k + 2 < T; ++k) {          // change k goes to nodes k+2 .. T-1 (node k+1 is the solver's)
      for (int spin = 0; s_prog[0] < k + 1; ++spin) {
        if (spin > (1 << 22)) { if (lane == 0) atomicOr(status, 2); break; }   // never spin forever
        __builtin_amdgcn_s_sleep(1);
      }
      __threadfence_block();
      double sv[RU], dd[RU], aa[RU];
#pragma unroll
      for (int r = 0; r < RU; ++r) {
        const bool act = ui[r] >= k + 2;
        const int i = act ? ui[r] : 0, e = ue[r];
        sv[r] = s_S[i * (EIG_T + 1) + k];
        dd[r] = (e < NEG) ? s_dP[k * NEGM + e] : s_dF[k * DM + (e - NEG)];
        aa[r] = ka * (double)s_a[i * EIG_T + k];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < RU; ++r) {
        if (ui[r] >= k + 2) {
          const double svr = probit ? 2.0 : sv[r];
          const double coef = (ue[r] < NEG) ? svr : gg_fma(-mu0, svr, probit ? sv[r] : aa[r]);
          val[r] = gg_fma(coef, dd[r], val[r]);
          s_part[ui[r] * (NEGM + DM) + ue[r]] = val[r];
        }
      }
      __threadfence_block();
      if (lane == 0) s_progb[wave] = (unsigned char)(k + 1);
    }
  }
}

// Statistics of _resample_mu_0 (eigenmodel.py:914-925) and _resample_lmbda (:938-958) over pairs i < j:
//   [0] sum S_ij   [1] sum (A_ij + A_ji - 1)   [2..2+D) sum S g   [2+D..2+2D) sum a g   then upper triangle of sum S g g^T,
// g = F_i * F_j (elementwise).  One workgroup per row i; partials reduced in k_eig_draw_globals.
template <int DM>
__global__ __launch_bounds__(256) void k_eig_global(int N, int Np, int D, const double* __restrict__ S,
                                                    const uint8_t* __restrict__ A, const uint8_t* __restrict__ AT,
                                                    const double* __restrict__ F, double* __restrict__ gpart, int probit, double ka) {
  constexpr int NEGM = DM * (DM + 1) / 2;
  constexpr int NGM = 2 + 2 * DM + NEGM;
  double acc[NGM];
#pragma unroll
  for (int e = 0; e < NGM; ++e) acc[e] = 0.0;
  for (int i = blockIdx.x; i < N; i += gridDim.x) {   // a few rows per workgroup: <= 512 partials to sum afterwards
  double fi[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) fi[d] = d < D ? F[(size_t)i * D + d] : 0.0;
  for (int j = i + 1 + threadIdx.x; j < N; j += 256) {
    const double raw = S[(size_t)i * Np + j];
    const double s = probit ? 2.0 : raw;
    const double a = probit ? raw : ka * (double)((int)A[(size_t)i * Np + j] + (int)AT[(size_t)i * Np + j] - 1);
    double g[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) g[d] = d < D ? fi[d] * F[(size_t)j * D + d] : 0.0;
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
      for (int i = lo; i < hi; ++i) v += gpart[(size_t)i * NGM + e];
      s_q[q][e] = v;
    }
    __syncthreads();
    if (tid < NGM) s_tot[tid] = (s_q[0][tid] + s_q[1][tid]) + (s_q[2][tid] + s_q[3][tid]);
  }
  // diagonal: sum omega_ii and sum (A_ii - 1/2)
  double so = 0.0, sk = 0.0;
  if (!probit)   // (the probit bias update leaves the diagonal out altogether: eigenmodel.py:339-346)
    for (int i = tid; i < N; i += 256) { so += dO[i]; sk += ka * ((dA[i] ? 1.0 : 0.0) - 0.5); }
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
  if ((long long)cfg->N * cfg->N >= (1LL << 32) + 1)
    return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel: N*N must fit the 32-bit Philox index");
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
  e->NCH = (e->N + EIG_CHN - 1) / EIG_CHN;
  int rc;
  GG_HIP(hipStreamCreate(&e->stream));
  GG_HIP(hipEventCreate(&e->e0));
  GG_HIP(hipEventCreate(&e->e1));
  const size_t N = e->N, D = e->D;
  if ((rc = emalloc(&e->F, N * D)) || (rc = emalloc(&e->lmbda, D)) || (rc = emalloc(&e->mu0, 1)) ||
      (rc = emalloc(&e->mu_lmbda, D)) || (rc = emalloc(&e->S, N * e->Np)) || (rc = emalloc(&e->dO, N)) ||
      (rc = emalloc(&e->Z, N * D)) || (rc = emalloc(&e->part, (size_t)2 * EIG_T * e->NCH * e->NE)) ||
      (rc = emalloc(&e->gpart, N * e->NG)) || (rc = emalloc(&e->status, 1)) || (rc = emalloc(&e->dbgbuf, 16)))
    return rc;
  const size_t nt = (N + 31) / 32;
  if ((rc = emalloc(&e->llpart, nt * nt))) return rc;
  *out = e;
  return GG_OK;
}

int gg_eigen_destroy(gg_eigen* e) {
  if (!e) return GG_OK;
  void* ptrs[] = {e->F, e->lmbda, e->mu0, e->mu_lmbda, e->S, e->dO, e->Z, e->part, e->gpart, e->llpart, e->status};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (e->e0) (void)hipEventDestroy(e->e0);
  if (e->e1) (void)hipEventDestroy(e->e1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
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
  if (!d) return gg_fail(GG_ERR_INVALID, "null data handle");
  if (d->N != e->N) return gg_fail(GG_ERR_INVALID, "data has N=%d, model has N=%d", d->N, e->N);
  const int N = e->N, D = e->D, Np = e->Np;
  hipStream_t st = e->stream;
  // a part asked for a second time under the same sweep index starts a new one: fresh Philox counters (graphgibbs.h)
  if (parts & e->parts_done) { e->sweep += 1; e->parts_done = 0; }
  GG_HIP(hipEventRecord(e->e0, st));
  const bool full = (parts & (GG_PART_OMEGA | GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA)) ==
                    (GG_PART_OMEGA | GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA);
  if (parts & GG_PART_OMEGA) {
    const int nt = (N + 31) / 32;
    const int npairs = nt * (nt + 1) / 2;
    if (e->probit) {
#define CALL(DMV) k_eig_z<DMV><<<npairs, 256, 0, st>>>(N, Np, D, e->F, e->lmbda, e->mu0, d->A, d->AT, e->cfg.seed, e->sweep, e->S)
      EIG_DISPATCH(CALL);
#undef CALL
    } else {
#define CALL(DMV) k_eig_omega<DMV><<<npairs, 256, 0, st>>>(N, Np, D, e->F, e->lmbda, e->mu0, e->cfg.seed, e->sweep, e->S, e->dO, e->beta)
      EIG_DISPATCH(CALL);
#undef CALL
    }
    e->have_omega = true;
  }
  if ((parts & (GG_PART_F | GG_PART_MU0 | GG_PART_LMBDA)) && !e->have_omega)
    return gg_fail(GG_ERR_INVALID, "eigenmodel: resample F / mu_0 / lambda needs the auxiliary variables (GG_PART_OMEGA) first");
  if (parts & GG_PART_F) {
    k_eig_normals<<<(N * D + 255) / 256, 256, 0, st>>>(e->Z, N * D, e->cfg.seed, GG_SITE_EIG_F, e->sweep);
    static int dbgon = -1;
    if (dbgon < 0) { const char* ev = getenv("GG_EIG_DBG"); dbgon = ev ? atoi(ev) : 0; }
    long long* dbg = dbgon ? e->dbgbuf : nullptr;
    if (dbg) GG_HIP(hipMemsetAsync(dbg, 0, 16 * sizeof(long long), st));
    const int DMv = e->DM, NEGMv = DMv * (DMv + 1) / 2;
    const size_t chain_lds = (((size_t)EIG_T * (EIG_T + 1) + (size_t)EIG_T * NEGMv + 3 * (size_t)EIG_T * DMv +
                               (size_t)EIG_T * (NEGMv + DMv) + (NEGMv + DMv)) * sizeof(double) + (size_t)EIG_T * EIG_T + 64 + 15) & ~(size_t)15;
    static bool attr4 = false, attr8 = false;
    if (DMv == 4 && !attr4) {
      GG_HIP(hipFuncSetAttribute((const void*)k_eig_chain<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
      GG_HIP(hipFuncSetAttribute((const void*)k_eig_chain<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
      attr4 = true;
    }
    if (DMv == 8 && !attr8) {
      GG_HIP(hipFuncSetAttribute((const void*)k_eig_chain<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
      GG_HIP(hipFuncSetAttribute((const void*)k_eig_chain<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
      attr8 = true;
    }
    // Tile 0's statistics in a launch of their own; every chain launch then carries, as extra workgroups, the
    // statistics of the tile after it (all nodes outside both tiles), and adds the previous tile's nodes itself.
    const size_t part_stride = (size_t)EIG_T * e->NCH * e->NE;
    {
      dim3 grd(N < EIG_T ? N : EIG_T, e->NCH);
#define CALL(DMV) k_eig_tile_stats<DMV><<<grd, 1024, 0, st>>>(N, Np, D, 0, e->S, d->A, d->AT, e->F, e->lmbda, e->mu0, e->NCH, e->part, 0, 0, e->probit, e->beta)
      EIG_DISPATCH(CALL);
#undef CALL
    }
    int ti = 0;
    for (int t0 = 0; t0 < N; t0 += EIG_T, ++ti) {
      double* partb = e->part + (size_t)(ti & 1) * part_stride;
      double* partn = e->part + (size_t)((ti + 1) & 1) * part_stride;
      const int Tn = (t0 + EIG_T < N) ? ((N - t0 - EIG_T < EIG_T) ? (N - t0 - EIG_T) : EIG_T) : 0;
      const int grid = 1 + Tn * e->NCH;
#define CALL(DMV)                                                                                                      \
  if (D == DMV)                                                                                                        \
    k_eig_chain<DMV, true><<<grid, 512, chain_lds, st>>>(N, Np, D, t0, e->S, d->A, d->AT, e->F, e->lmbda, e->mu0,       \
                                                         e->cfg.sigma_F, e->NCH, partb, e->Z, e->status, dbg, partn, e->probit, e->beta); \
  else                                                                                                                 \
    k_eig_chain<DMV, false><<<grid, 512, chain_lds, st>>>(N, Np, D, t0, e->S, d->A, d->AT, e->F, e->lmbda, e->mu0,      \
                                                          e->cfg.sigma_F, e->NCH, partb, e->Z, e->status, dbg, partn, e->probit, e->beta)
      EIG_DISPATCH(CALL);
#undef CALL
    }
  }
  const int do_mu0 = (parts & GG_PART_MU0) ? 1 : 0;
  const int do_lm = ((parts & GG_PART_LMBDA) && !e->cfg.lmbda_given) ? 1 : 0;
  if (do_mu0 || do_lm) {
#define CALL(DMV)                                                                                                   \
  k_eig_global<DMV><<<(N < 512 ? N : 512), 256, 0, st>>>(N, Np, D, e->S, d->A, d->AT, e->F, e->gpart, e->probit, e->beta);    \
  k_eig_draw_globals<DMV><<<1, 256, 0, st>>>(N, (N < 512 ? N : 512), D, e->gpart, e->dO, d->dA, e->lmbda, e->mu0, e->mu_lmbda,            \
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
    if (h[3]) fprintf(stderr, "[eig chain]   solver lane per node: fold %.0f, load %.0f, cholesky %.0f, solves %.0f clocks\n", (double)h[11] / h[3], (double)h[8] / h[3], (double)h[9] / h[3], (double)h[10] / h[3]);
    if (h[3]) fprintf(stderr, "[eig chain] per node: wait %.0f fold %.0f solve %.0f shader clocks (%lld nodes); prologue %.0f clocks per tile (block %.0f, +loads %.0f, +cross %.0f)\n", (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], h[3], (double)h[4] / ((h[3] + EIG_T - 1) / EIG_T), (double)h[5] / 64, (double)h[6] / 64, (double)h[7] / 64);
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
