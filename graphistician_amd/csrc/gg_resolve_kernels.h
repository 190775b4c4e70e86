// gg_resolve_kernels.h — exact resolution of the sequential node scan.
//
// The reference visits nodes in order and each node sees the labels already
// redrawn in this scan (adjacency.py:661,695; weights.py:473,488).  Because the
// uniform of node n is a function of (seed, sweep, n) alone, every node's draw
// can be evaluated ahead of time ("speculated") against the labels at the start
// of the scan; the speculation of node n is exact unless an earlier node changed
// its label.  k_resolve walks the scan order and
//   phase A: skips to the first node whose speculated label differs from its
//            current one (every node before it is final, unchanged);
//   phase B: accepts that change, then proceeds in chunks of CH nodes whose
//            statistics live in LDS: pending changes (list L) are applied as
//            +-1 / +-q corrections read from single rows of X and X^T, the chunk
//            is re-evaluated, the first changer accepted, and so on.
// All corrections are integer, so the statistics a node sees are bit-identical
// to a from-scratch recount under the current labels: same lp, same draw as the
// sequential reference scan.  The kernel leaves when the scan is complete, when
// L is full, or when it has gone `max_idle` chunks without a change (then the
// host applies L to G chip-wide, re-speculates the remainder and re-enters at
// phase A).
#pragma once
#include "gg_block_kernels.h"

struct gg_resolve_args {
  gg_node_ctx x;
  int32_t* c;
  int32_t* cnt;
  const int32_t* cand;
  int32_t* rs;   // control block: [0] pos, [1] nL, [2] label changes (accumulated), [3] status,
                 //   [4] stalled (scan left unfinished: later pipeline kernels do nothing), [5] completed scans,
                 //   [6] sweep index (graph replay), [7] scan done, [8] next re-speculation window, [9] changes of this scan
  int32_t* L;    // [Lcap][3] = (node, old, new)
  // tail, run when the scan completes: resample_m (adjacency.py:697-702 / weights.py:493-498)
  // and the label sort the next half-sweep's bulk kernel needs
  int tail_m, tail_sort;
  int round_mode;        // device-driven restart round: runs even if the stall flag is up, no-op once the scan is done
  int window;            // phase A looks at [pos, pos + window) only (what was re-speculated); 0 = to the end
  int stage_tables;      // coefficient tables fit in LDS next to the chunk statistics
  int sweep_from_ctl;    // graph replay: sweep index = rs[6], advanced by the tail of a completed scan
  int flush_on_change;   // pipelined mode: a scan that changed any label flushes the pipeline (the next sweep's
                         // statistics were computed early under the unchanged labels)
  double* m;
  const double* pi;
  int32_t* perm;
  int32_t* lab_start;
  const int32_t* perm_prev;       // the sort the scan started from: copied when no label changed
  const int32_t* lab_start_prev;
  uint32_t site_m;
  int Lmax;
  int max_idle;
  int CH;
  int fresh;     // 1: L is empty on entry
  const uint8_t* XA;   // adjacency rows / transposed rows
  const uint8_t* XAT;
  const int32_t* XW;   // weights
  const int32_t* XWT;
  uint64_t seed;
  uint32_t site, sweep;
};

template <int NT, bool ADJ>
__device__ __forceinline__ void gg_apply_edge(int* gc, long long* gs, int a, int b, int K, const int32_t* q,
                                              bool is_edge) {
  // move one edge's statistics from block a to block b of one node/direction
  constexpr int B = ADJ ? 1 : (NT == 3 ? 1 : (NT == 6 ? 2 : (NT == 10 ? 3 : 4)));
  constexpr int NSI = NT - 1;
  if (!is_edge) return;
  atomicSub(&gc[a], 1);
  atomicAdd(&gc[b], 1);
  if (!ADJ) {
    long long v[NSI > 0 ? NSI : 1];
    int t = 0;
    for (int r = 0; r < B; ++r) v[t++] = q[r];
    for (int r = 0; r < B; ++r)
      for (int s = r; s < B; ++s) v[t++] = (long long)q[r] * (long long)q[s];
    for (int s = 0; s < NSI; ++s) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&gs[a * NSI + s]), (unsigned long long)(-v[s]));
      atomicAdd(reinterpret_cast<unsigned long long*>(&gs[b * NSI + s]), (unsigned long long)v[s]);
    }
  }
}

template <int NT, bool ADJ>
__global__ __launch_bounds__(1024) void k_resolve(gg_resolve_args g) {
  constexpr int B = ADJ ? 1 : (NT == 3 ? 1 : (NT == 6 ? 2 : (NT == 10 ? 3 : 4)));
  constexpr int NSI = NT - 1;
  const gg_node_ctx& x = g.x;
  const int N = x.N, K = x.K, Kp = x.Kp, Np = x.Np, CH = g.CH;
  const int tid = threadIdx.x, nth = blockDim.x;

  extern __shared__ __align__(16) unsigned char s_raw[];
  // carve LDS (8-byte items first)
  long long* s_gs = reinterpret_cast<long long*>(s_raw);               // [2][CH][K][NSI] (weights)
  size_t off = ADJ ? 0 : (size_t)2 * CH * K * NSI * sizeof(long long);
  int* s_gc = reinterpret_cast<int*>(s_raw + off);                     // [2][CH][K]
  off += (size_t)2 * CH * K * sizeof(int);
  int* s_cnt = reinterpret_cast<int*>(s_raw + off); off += (size_t)K * sizeof(int);
  int* s_c = reinterpret_cast<int*>(s_raw + off); off += (size_t)CH * sizeof(int);
  int* s_cand = reinterpret_cast<int*>(s_raw + off); off += (size_t)CH * sizeof(int);
  int* s_ctl = reinterpret_cast<int*>(s_raw + off); off += 8 * sizeof(int);
  int32_t* s_blkw = reinterpret_cast<int32_t*>(s_raw + off);           // [CH][CH][B] (weights)
  uint8_t* s_blka = reinterpret_cast<uint8_t*>(s_raw + off);           // [CH][CH]    (adjacency)
  off += ADJ ? (size_t)CH * CH : (size_t)CH * CH * B * sizeof(int32_t);
  off = (off + 15) & ~(size_t)15;
  double* s_T = reinterpret_cast<double*>(s_raw + off);                // Tout | Tin when staged (phase B only)
  const int ntab = K * NT * Kp;
  bool tables_staged = false;
  const double* To = g.stage_tables ? s_T : x.Tout;
  const double* Ti = g.stage_tables ? s_T + ntab : x.Tin;

  if (g.round_mode) { if (g.rs[7]) return; }
  else if (g.rs[4]) return;  // an earlier scan of this pipeline is unfinished
  if (g.sweep_from_ctl) g.sweep = (uint32_t)g.rs[6];
  int pos = g.rs[0];
  int nL = g.fresh ? 0 : g.rs[1];
  int nchg = 0;
  bool finished = false;
  for (int k = tid; k < K; k += nth) s_cnt[k] = g.cnt[k];
  if (tid == 0) s_ctl[0] = N;
  __syncthreads();

  // ---------------------------------------------------------------- phase A
  if (nL == 0) {
    int wlen = g.window;
    if (g.round_mode && g.rs[8] > 0 && g.rs[8] < wlen) wlen = g.rs[8];   // what k_speculate covered this round
    const int wend = (wlen > 0 && pos + wlen < N) ? pos + wlen : N;
    int first = N;
    for (int n = pos + tid; n < wend; n += nth)
      if (g.cand[n] != g.c[n]) { first = n; break; }
    if (first < N) atomicMin(&s_ctl[0], first);
    __syncthreads();
    int ns = s_ctl[0];
    if (ns >= N) {
      pos = wend;           // the whole window is final, unchanged
      finished = true;      // nothing to fix in this launch (the scan itself is done only if pos == N)
      if (g.round_mode && tid == 0) g.rs[8] = g.window;   // quiet stretch: widen the next window
    } else {
      int a = g.c[ns], b = g.cand[ns];
      __syncthreads();
      if (tid == 0) {
        g.c[ns] = b;
        s_cnt[a] -= 1; s_cnt[b] += 1;
        g.L[0] = ns; g.L[1] = a; g.L[2] = b;
      }
      nL = 1; nchg = 1;
      // sparse regime (the change sits far from where the scan stood): leave the chunk loop out, the next
      // round re-speculates a window after it; dense regime (burn-in): resolve chunk by chunk below
      if (g.round_mode && ns - pos >= CH) finished = true;
      if (g.round_mode) {
        int dist = ns - pos + 1, w = 4 * dist;
        w = w < 256 ? 256 : (w > g.window ? g.window : w);
        if (tid == 0) g.rs[8] = w;
      }
      pos = ns + 1;
      __syncthreads();
    }
  }

  // ---------------------------------------------------------------- phase B
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  int idle = 0;
  while (!finished && pos < N && nL < g.Lmax && idle < g.max_idle) {
    if (g.stage_tables && !tables_staged) {
      for (int idx = threadIdx.x; idx < ntab; idx += blockDim.x) {
        s_T[idx] = x.Tout[idx];
        s_T[ntab + idx] = x.Tin[idx];
      }
      tables_staged = true;   // visible after the barrier that follows the chunk loads
    }
    const int pos0 = pos;
    const int nch = (N - pos0) < CH ? (N - pos0) : CH;
    // load chunk labels, statistics (coalesced over the node index) and the diagonal block
    for (int i = tid; i < nch; i += nth) s_c[i] = g.c[pos0 + i];
    for (int idx = tid; idx < 2 * K * nch; idx += nth) {
      int i = idx % nch, j = (idx / nch) % K, d = idx / (nch * K);
      size_t gi = (size_t)j * Np + pos0 + i;
      s_gc[(d * CH + i) * K + j] = x.Gcnt[d][gi];
      if (!ADJ)
        for (int s = 0; s < NSI; ++s) s_gs[((size_t)(d * CH + i) * K + j) * NSI + s] = x.GS[d][gi * NSI + s];
    }
    for (int idx = tid; idx < nch * nch; idx += nth) {
      int i2 = idx % nch, i1 = idx / nch;
      if (ADJ) s_blka[i1 * CH + i2] = g.XA[(size_t)(pos0 + i1) * Np + pos0 + i2];
      else
        for (int r = 0; r < B; ++r)
          s_blkw[(i1 * CH + i2) * B + r] = g.XW[((size_t)(pos0 + i1) * Np + pos0 + i2) * B + r];
    }
    __syncthreads();
    // apply the pending changes of earlier nodes to the chunk
    for (int idx = tid; idx < nL * nch; idx += nth) {
      int e = idx / nch, i = idx % nch;
      int n2 = g.L[e * 3], a = g.L[e * 3 + 1], b = g.L[e * 3 + 2];
      size_t at = (size_t)n2 * Np + pos0 + i;
      int* gco = &s_gc[(0 * CH + i) * K];
      int* gci = &s_gc[(1 * CH + i) * K];
      long long* gso = ADJ ? nullptr : &s_gs[((size_t)(0 * CH + i) * K) * NSI];
      long long* gsi = ADJ ? nullptr : &s_gs[((size_t)(1 * CH + i) * K) * NSI];
      if (ADJ) {
        gg_apply_edge<NT, ADJ>(gco, gso, a, b, K, nullptr, g.XAT[at] != 0);  // edge (pos0+i) -> n2
        gg_apply_edge<NT, ADJ>(gci, gsi, a, b, K, nullptr, g.XA[at] != 0);   // edge n2 -> (pos0+i)
      } else {
        const int32_t* qo = g.XWT + at * B;
        const int32_t* qi = g.XW + at * B;
        gg_apply_edge<NT, ADJ>(gco, gso, a, b, K, qo, qo[0] != GG_NO_EDGE);
        gg_apply_edge<NT, ADJ>(gci, gsi, a, b, K, qi, qi[0] != GG_NO_EDGE);
      }
    }
    __syncthreads();

    bool changed = false;
    bool full = false;
    for (;;) {
      if (tid == 0) s_ctl[0] = CH;
      __syncthreads();
      const int i = tid / Kp, k = tid % Kp;
      const bool active = (i < nch) && (pos0 + i >= pos);
      const int ii = (i < nch) ? i : 0;
      const int n = pos0 + ii;
      const int cn = s_c[ii];
      auto ld = [&](int j, double* so, double* si) {
        int cx = ADJ ? (s_cnt[j] - (cn == j ? 1 : 0)) : 0;
        const long long* a = ADJ ? nullptr : &s_gs[((size_t)(0 * CH + ii) * K + j) * NSI];
        const long long* b = ADJ ? nullptr : &s_gs[((size_t)(1 * CH + ii) * K + j) * NSI];
        gg_stats_to_double<NT, ADJ>(s_gc[(0 * CH + ii) * K + j], s_gc[(1 * CH + ii) * K + j], a, b, cx, sc1, sc2, so, si);
      };
      double lp = gg_node_lp_core<NT, ADJ>(x, n, k, ld, To, Ti);
      double u = gg_uniform_at(g.seed, g.site, g.sweep, (uint32_t)n);
      int lab = gg_group_categorical(lp, k, K, Kp, u);
      if (active && k == 0 && lab != cn) {
        s_cand[ii] = lab;
        atomicMin(&s_ctl[0], ii);
      }
      __syncthreads();
      const int f = s_ctl[0];
      if (f >= CH) break;
      const int a = s_c[f], b = s_cand[f];
      __syncthreads();
      if (tid == 0) {
        s_c[f] = b;
        g.c[pos0 + f] = b;
        s_cnt[a] -= 1; s_cnt[b] += 1;
        g.L[nL * 3] = pos0 + f; g.L[nL * 3 + 1] = a; g.L[nL * 3 + 2] = b;
      }
      // later nodes of the chunk see the change through the LDS diagonal block
      if (tid > f && tid < nch) {
        int* gco = &s_gc[(0 * CH + tid) * K];
        int* gci = &s_gc[(1 * CH + tid) * K];
        long long* gso = ADJ ? nullptr : &s_gs[((size_t)(0 * CH + tid) * K) * NSI];
        long long* gsi = ADJ ? nullptr : &s_gs[((size_t)(1 * CH + tid) * K) * NSI];
        if (ADJ) {
          gg_apply_edge<NT, ADJ>(gco, gso, a, b, K, nullptr, s_blka[tid * CH + f] != 0);  // tid -> f
          gg_apply_edge<NT, ADJ>(gci, gsi, a, b, K, nullptr, s_blka[f * CH + tid] != 0);  // f -> tid
        } else {
          const int32_t* qo = &s_blkw[(tid * CH + f) * B];
          const int32_t* qi = &s_blkw[(f * CH + tid) * B];
          gg_apply_edge<NT, ADJ>(gco, gso, a, b, K, qo, qo[0] != GG_NO_EDGE);
          gg_apply_edge<NT, ADJ>(gci, gsi, a, b, K, qi, qi[0] != GG_NO_EDGE);
        }
      }
      nL += 1; nchg += 1; changed = true;
      pos = pos0 + f + 1;
      __syncthreads();
      if (nL >= g.Lmax) { full = true; break; }
    }
    if (!full) pos = pos0 + nch;
    idle = changed ? 0 : idle + 1;
    __syncthreads();
  }
  for (int k = tid; k < K; k += nth) g.cnt[k] = s_cnt[k];
  const bool done = pos >= N;
  __syncthreads();
  if (done) {
    double* s_gam = reinterpret_cast<double*>(s_raw);        // [K]   (chunk statistics are dead now)
    int* s_cur = reinterpret_cast<int*>(s_raw + 64 * sizeof(double));  // [K]
    if (g.tail_m) {
      if (tid < K) s_gam[tid] = gg_dirichlet_gamma_at(g.seed, g.site_m, g.sweep, tid, g.pi[tid] + (double)s_cnt[tid]);
      __syncthreads();
      if (tid < K) {
        double tot = 0.0;
        for (int j = 0; j < K; ++j) tot += s_gam[j];
        g.m[tid] = s_gam[tid] / tot;
      }
    }
    if (g.tail_sort) {
      if (g.rs[9] + nchg == 0) {
        // no label changed in this scan: the sort it started from is still the sort of the labels
        for (int n = tid; n < N; n += nth) g.perm[n] = g.perm_prev[n];
        for (int k = tid; k <= K; k += nth) g.lab_start[k] = g.lab_start_prev[k];
      } else {
        if (tid == 0) {
          int acc = 0;
          for (int k = 0; k < K; ++k) { g.lab_start[k] = acc; s_cur[k] = acc; acc += s_cnt[k]; }
          g.lab_start[K] = acc;
        }
        __syncthreads();
        for (int n = tid; n < N; n += nth) {
          int at = atomicAdd(&s_cur[g.c[n]], 1);
          g.perm[at] = n;
        }
      }
    }
  }
  if (tid == 0) {
    g.rs[0] = done ? 0 : pos;
    g.rs[1] = done ? 0 : nL;
    g.rs[2] += nchg;
    g.rs[9] += nchg;
    if (done) {
      g.rs[5] += 1;
      g.rs[7] = 1;
      if (g.sweep_from_ctl) g.rs[6] += 1;
    }
    if (!done || (g.flush_on_change && nchg > 0)) g.rs[4] = 1;
  }
}

// Apply the change list to the resident statistics of every node (chip-wide):
// after it G is again the exact block histogram under the current labels.
template <int NT, bool ADJ>
__global__ __launch_bounds__(256) void k_apply_L(gg_node_ctx x, const int32_t* __restrict__ L,
                                                 const int32_t* __restrict__ rs, int32_t* Gc0, int32_t* Gc1,
                                                 long long* Gs0, long long* Gs1, const uint8_t* XA,
                                                 const uint8_t* XAT, const int32_t* XW, const int32_t* XWT) {
  constexpr int B = ADJ ? 1 : (NT == 3 ? 1 : (NT == 6 ? 2 : (NT == 10 ? 3 : 4)));
  constexpr int NSI = NT - 1;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= x.N || rs[7]) return;
  const int nL = rs[1], Np = x.Np;
  for (int e = 0; e < nL; ++e) {
    const int n2 = L[e * 3], a = L[e * 3 + 1], b = L[e * 3 + 2];
    const size_t at = (size_t)n2 * Np + n;
    if (ADJ) {
      int xo = XAT[at] != 0, xi = XA[at] != 0;
      if (xo) { Gc0[(size_t)a * Np + n] -= 1; Gc0[(size_t)b * Np + n] += 1; }
      if (xi) { Gc1[(size_t)a * Np + n] -= 1; Gc1[(size_t)b * Np + n] += 1; }
    } else {
      for (int d = 0; d < 2; ++d) {
        const int32_t* q = (d == 0 ? XWT : XW) + at * B;
        if (q[0] == GG_NO_EDGE) continue;
        int32_t* Gc = d == 0 ? Gc0 : Gc1;
        long long* Gs = d == 0 ? Gs0 : Gs1;
        Gc[(size_t)a * Np + n] -= 1;
        Gc[(size_t)b * Np + n] += 1;
        long long v[NSI > 0 ? NSI : 1];
        int t = 0;
        for (int r = 0; r < B; ++r) v[t++] = q[r];
        for (int r = 0; r < B; ++r)
          for (int s = r; s < B; ++s) v[t++] = (long long)q[r] * (long long)q[s];
        for (int s = 0; s < NSI; ++s) {
          Gs[((size_t)a * Np + n) * NSI + s] -= v[s];
          Gs[((size_t)b * Np + n) * NSI + s] += v[s];
        }
      }
    }
  }
}
