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
  int32_t* cand;
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
  const uint32_t* XA;   // adjacency / its transpose, bit-packed (gg_abit)
  const uint32_t* XAT;
  const int32_t* XW;   // weights
  const int32_t* XWT;
  uint64_t seed;
  uint32_t site, sweep;
  long long* dbg;   // optional phase timestamps of k_round (GG_ROUND_DBG), [workgroup][8]
  int keep_L;       // incremental statistics: a completed scan leaves its change list in place for the flush into G
  unsigned* verdict;   // fused quiet-chain launch: where the resolving workgroup publishes (tag << 1 | changed)
};


// Round state of a scan in one 64-bit word (rs[16..17]) so that a restart round reads it with a single load:
//   bits 0-23 pos | 24-37 nL | 38-50 next window (0 = full) | 51 scan done | 52-63 round counter (mod 4096)
#define GG_RS_STATE 16
#define GG_RS_MIN 18      // 64-bit: smallest (node << 12 | old << 6 | new) reported by this round's workgroups
#define GG_NO_CHANGE 0xffffffffffffffffull
__host__ __device__ inline unsigned long long gg_pack_state(int pos, int nL, int wlen, int fin, int seq) {
  return (unsigned long long)(unsigned)pos | ((unsigned long long)(unsigned)nL << 24) |
         ((unsigned long long)(unsigned)wlen << 38) | ((unsigned long long)(fin ? 1 : 0) << 51) |
         ((unsigned long long)((unsigned)seq & 0xfffu) << 52);
}
__host__ __device__ inline void gg_unpack_state(unsigned long long w, int* pos, int* nL, int* wlen, int* fin, int* seq) {
  *pos = (int)(w & 0xffffffu);
  *nL = (int)((w >> 24) & 0x3fffu);
  *wlen = (int)((w >> 38) & 0x1fffu);
  *fin = (int)((w >> 51) & 1u);
  *seq = (int)((w >> 52) & 0xfffu);
}

template <int NT, bool ADJ>
__device__ __forceinline__ void gg_apply_edge(int* gc, long long* gs, int a, int b, int K, const int32_t* q,
                                              bool is_edge) {
  // move one edge's statistics from block a to block b of one node/direction
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
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

// Tail of a completed scan: resample_m (adjacency.py:697-702 / weights.py:493-498) and the label sort the next
// half-sweep's bulk kernel needs.  s_cnt = final block sizes (LDS), s_raw = at least 64 doubles + K ints of dead LDS.
__device__ __forceinline__ void gg_scan_tail(const gg_resolve_args& g, const int* s_cnt, unsigned char* s_raw,
                                             bool unchanged) {
  const int N = g.x.N, K = g.x.K;
  const int tid = threadIdx.x, nth = blockDim.x;
  double* s_gam = reinterpret_cast<double*>(s_raw);                  // [K]
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
    if (unchanged) {
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
      // (labels of 16 nodes per thread in flight together: one workgroup sorts the whole vector, and a loop that fetches,
      // counts and stores node by node pays a memory latency per node - 6.3 us for N = 4096)
      constexpr int UL = 16;
      for (int n0 = tid; n0 < N; n0 += nth * UL) {
        int lab[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) { const int n = n0 + u * nth; lab[u] = n < N ? g.c[n] : 0; }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const int n = n0 + u * nth;
          if (n < N) { const int at = atomicAdd(&s_cur[lab[u]], 1); g.perm[at] = n; }
        }
      }
    }
  }
}

template <int NT, bool ADJ>
__global__ __launch_bounds__(1024) void k_resolve(gg_resolve_args g) {
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
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
      size_t gi = (size_t)d * K * Np + (size_t)j * Np + pos0 + i;   // direction 1 follows direction 0 in the buffer
      s_gc[(d * CH + i) * K + j] = x.Gcnt[0][gi];
      if (!ADJ)
        for (int s = 0; s < NSI; ++s) s_gs[((size_t)(d * CH + i) * K + j) * NSI + s] = x.GS[0][gi * NSI + s];
    }
    for (int idx = tid; idx < nch * nch; idx += nth) {
      int i2 = idx % nch, i1 = idx / nch;
      if (ADJ) s_blka[i1 * CH + i2] = gg_abit(g.XA, (size_t)(pos0 + i1) * Np + pos0 + i2) ? 1 : 0;
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
        gg_apply_edge<NT, ADJ>(gco, gso, a, b, K, nullptr, gg_abit(g.XAT, at));  // edge (pos0+i) -> n2
        gg_apply_edge<NT, ADJ>(gci, gsi, a, b, K, nullptr, gg_abit(g.XA, at));   // edge n2 -> (pos0+i)
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
        int cx = (ADJ || gg_tJ(NT)) ? (s_cnt[j] - (cn == j ? 1 : 0)) : 0;
        const long long* a = ADJ ? nullptr : &s_gs[((size_t)(0 * CH + ii) * K + j) * NSI];
        const long long* b = ADJ ? nullptr : &s_gs[((size_t)(1 * CH + ii) * K + j) * NSI];
        gg_stats_to_double<NT, ADJ>(s_gc[(0 * CH + ii) * K + j], s_gc[(1 * CH + ii) * K + j], a, b, cx, sc1, sc2, so, si);
      };
      double lp;
      constexpr int UJR = NT == 2 ? 4 : (NT == 3 ? 2 : 1);   // 1024 threads: 128 VGPRs per lane
      if (g.stage_tables) lp = gg_node_lp_core<NT, ADJ, UJR>(x, n, k, ld, s_T, s_T + ntab);
      else lp = gg_node_lp_core<NT, ADJ, UJR>(x, n, k, ld, x.Tout, x.Tin);
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
  if (done) gg_scan_tail(g, s_cnt, s_raw, g.rs[9] + nchg == 0);
  if (tid == 0) {
    g.rs[0] = done ? 0 : pos;
    g.rs[1] = (done && !g.keep_L) ? 0 : nL;
    g.rs[2] += nchg;
    g.rs[9] += nchg;
    if (done) {
      g.rs[5] += 1;
      g.rs[7] = 1;
      if (g.sweep_from_ctl) g.rs[6] += 1;
    }
    if (!done || (g.flush_on_change && nchg > 0)) g.rs[4] = 1;
    // round state for the restart rounds that finish an incomplete scan (counter restarts at 0)
    *reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_STATE]) = gg_pack_state(done ? 0 : pos, done ? 0 : nL, 0, done ? 1 : 0, 0);
    *reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_MIN]) = GG_NO_CHANGE;
  }
}

// k_speculate and the common case of k_resolve in ONE launch (the speculative form of the adaptive scan, gg_model.hip):
// every workgroup evaluates its nodes against the labels at scan start, reports its first changer with a device-scope
// 64-bit atomicMin and draws a ticket; while waiting for nothing it also copies its share of the label sort (valid iff
// the scan turns out unchanged; rewritten by the eventual tail otherwise).  The workgroup that draws the last ticket
// sees every report:
//   * none: the scan is complete and unchanged - it publishes that (what k_resolve's tail would);
//   * otherwise the smallest one IS the next exact change of the sequential scan (every node before it was evaluated
//     under the right labels and kept its own): it is accepted - label, block counts, change list - and the scan is left
//     STALLED behind it, for the resident grid to finish (finish_scan: k_apply_L + k_scan in resume mode).
// A quiet scan costs one launch (~10 us) instead of two (9 + 8.5 us).
#ifndef GG_RIDE_UJ
#define GG_RIDE_UJ 2
#endif
#ifndef GG_RIDE_WAVES
#define GG_RIDE_WAVES 4   // waves per SIMD the riding variants are compiled for (<= 128 VGPRs): the riders need the occupancy
#endif
// The speculative scan of one workgroup of `nfront` (block indices 0 .. nfront-1 of the launch): see k_spec_resolve.
// vtag != 0: the resolving workgroup also publishes the verdict word (gg_wait_verdict).
template <int NT, bool ADJ, int UJ_, int SPLIT>
__device__ __forceinline__ void gg_spec_resolve_body(const gg_resolve_args& g, int stage_tables, int32_t* __restrict__ ap,
                                                     const unsigned nfront, const unsigned vtag) {
  const gg_node_ctx& x = g.x;
  const int tid = threadIdx.x, N = x.N, K = x.K;
  constexpr int LB = SPLIT == 2 ? 8 : 6;
  constexpr unsigned LMASK = (1u << LB) - 1u;
  __shared__ unsigned long long s_fmin;
  __shared__ int s_last, s_any;
  if (tid == 0) s_fmin = GG_NO_CHANGE;
  if (SPLIT == 2) {
    // more than 64 blocks: several passes' worth of nodes per workgroup, one fetch of a coefficient-table entry for all of
    // them (gg_spec_eval_wide); the grid is sized accordingly (launch_spec_resolve)
    constexpr int P = GG_WIDE_P(NT);
    int labs[P], nodes[P];
    bool lives[P];
    __syncthreads();   // (s_fmin is initialised before anyone reports)
    gg_spec_eval_wide<NT, ADJ, P>(x, g.seed, g.site, g.sweep, labs, nodes, lives);
    if (tid % x.Kp == 0) {
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (lives[p]) {
          const int n = nodes[p], cn = x.c[x.n0 + n];
          ap[n] = 0;
          if (labs[p] != cn) atomicMin(&s_fmin, ((unsigned long long)(unsigned)n << (2 * LB)) | ((unsigned long long)(unsigned)cn << LB) | (unsigned)labs[p]);
        }
    }
  } else {
  int n;
  bool live;
  // the node's current label: fetched now, compared after the evaluation (one dependent round trip less at the end)
  const int n_mine = (int)blockIdx.x * (256 / x.Kp) + tid / x.Kp;
  const int cn_pre = x.c[x.n0 + (n_mine < N ? n_mine : N - 1)];
  // (riding variants: shallow table batches - the registers buy the riders' occupancy; same fma order, same bits)
  const int lab = gg_spec_eval<NT, ADJ, UJ_, SPLIT>(x, 0, g.seed, g.site, g.sweep, -1, nullptr, stage_tables, &n, &live);   // (syncs inside)
  if (live && tid % x.Kp == 0) {
    ap[n] = 0;   // no entry of this scan's change list is applied to the node yet
    const int cn = cn_pre;
    // (node, old, new) with 6-bit labels, 8-bit for more than 64 blocks (SPLIT == 2)
    if (lab != cn) atomicMin(&s_fmin, ((unsigned long long)(unsigned)n << (2 * LB)) | ((unsigned long long)(unsigned)cn << LB) | (unsigned)lab);
  }
  }
  if (g.tail_sort)
    for (int n2 = blockIdx.x * 256 + tid; n2 < N; n2 += nfront * 256) g.perm[n2] = g.perm_prev[n2];
  __syncthreads();
  if (tid == 0) {
    const unsigned long long mine = s_fmin;
    if (mine != GG_NO_CHANGE) {
      unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_MIN]), mine);
      asm volatile("" ::"v"(old) : "memory");   // the ticket is not drawn before the min has been performed
    }
    // one 64-bit ticket: low word = workgroups done, high word = workgroups that reported a changer - the last one then
    // knows whether the minimum has to be fetched at all (a quiet scan saves that dependent round trip)
    const unsigned long long tk = atomicAdd(reinterpret_cast<unsigned long long*>(&g.rs[10]),
                                            1ull + (mine != GG_NO_CHANGE ? (1ull << 32) : 0ull));
    s_last = ((unsigned)(tk & 0xffffffffull) == nfront - 1u) ? 1 : 0;
    s_any = ((tk >> 32) != 0ull || mine != GG_NO_CHANGE) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) s_fmin = s_any ? atomicExch(reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_MIN]), GG_NO_CHANGE) : GG_NO_CHANGE;
  __syncthreads();
  const unsigned long long best = s_fmin;
  if (best == GG_NO_CHANGE) {
    if (g.tail_sort)
      for (int k = tid; k <= K; k += 256) g.lab_start[k] = g.lab_start_prev[k];
    if (tid == 0) {
      g.rs[0] = 0;
      g.rs[1] = 0;
      g.rs[5] += 1;
      g.rs[7] = 1;
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(&g.rs[10]), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_STATE]) = gg_pack_state(0, 0, 0, 1, 0);
      // (release: the control words written above are ordered before the verdict - a gated parameter workgroup that has
      // seen it goes on to store to the same words, gg_params_body's housekeeping)
      if (vtag) __hip_atomic_store(g.verdict, vtag << 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (tid == 0) {
    const int f = (int)(best >> (2 * LB)), ca = (int)((best >> LB) & LMASK), cb = (int)(best & LMASK);
    g.c[f] = cb;
    atomicSub(&g.cnt[ca], 1);
    atomicAdd(&g.cnt[cb], 1);
    g.L[0] = f; g.L[1] = ca; g.L[2] = cb;
    // (more than 64 blocks may be finished by restart rounds, which have nothing to do for a scan that stands behind its
    // last node: when the LAST node changed, stand on it again - its conditional does not depend on its own label, the
    // round redraws the same one, finds the window final and runs the tail; found by the release soak under GG_SCAN=0.
    // The other shapes are finished by the resident scan, which runs its tail from any position: their code is left as it
    // is - it is inlined into k_sweep_q)
    const int npos = (SPLIT == 2 && f + 1 >= N) ? N - 1 : f + 1;
    g.rs[0] = npos;
    g.rs[1] = 1;
    g.rs[2] += 1;
    g.rs[9] += 1;
    g.rs[4] = 1;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(&g.rs[10]), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_STATE]) = gg_pack_state(npos, 1, 0, 0, 0);
    if (vtag) __hip_atomic_store(g.verdict, (vtag << 1) | 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// RIDE: riders of the next sweep's bulk pass behind the front workgroups (gg_block_kernels.h).  SPLIT: compiled for the
// K > 32 conditional with unstaged tables (1) or for the other two forms (0).
template <int NT, bool ADJ, int RIDE = 0, int SPLIT = 0>
__global__ __launch_bounds__(256, ((RIDE && !SPLIT) ? GG_RIDE_WAVES : 1)) void k_spec_resolve(
    gg_resolve_args g, int stage_tables, int32_t* __restrict__ ap, gg_ride_args ride = gg_ride_args()) {
  if (g.rs[4]) return;   // behind a stalled scan of this pipeline: no-op (uniform)
  if (RIDE && (int)blockIdx.x >= ride.front) {
    gg_ride<ADJ, (gg_tB(NT)), RIDE == 2>(ride);
    return;
  }
  const unsigned nfront = RIDE ? (unsigned)ride.front : gridDim.x;
  gg_spec_resolve_body<NT, ADJ, (RIDE ? GG_RIDE_UJ : 0), SPLIT>(g, stage_tables, ap, nfront, 0u);
}

// The quiet chain's sweep in ONE launch.  Block ranges:
//   [0, nspec)                 the speculative scan of sweep s              (k_spec_resolve's workgroups)
//   [nspec, nspec + nparams)   the parameter step of sweep s+1              (k_params' workgroups; they gather and reduce their
//                              pair tables while the scan is evaluated, then wait for its verdict before they write a word)
//   [nspec + nparams, ...)     riders: the bulk pass that counts the statistics of sweep s+2
// Valid because a scan that changes no label leaves every input of the steps riding with it as it was; a scan that does
// change one raises the stall flag as ever, the parameter workgroups of sweep s+1 drop what they computed, and the host
// resumes in the serial form.  The scan workgroups come first in the grid, so they are resident before any workgroup that
// waits for them.
#ifndef GG_W_RIDER_PRIO
#define GG_W_RIDER_PRIO 0    // issue priority of the weight model's riders in the fused launch (0: left alone)
#endif
#ifndef GG_ADJ_FRONT_PRIO
#define GG_ADJ_FRONT_PRIO 0   // issue priority of the adjacency model's front workgroups in the fused launch (its chain has slack)
#endif
template <int NT, bool ADJ, int NTL, int SPLIT>
__global__ __launch_bounds__(256, (SPLIT ? 1 : GG_RIDE_WAVES)) void k_sweep_q(gg_resolve_args g, int stage_tables, int32_t* __restrict__ ap,
                                                                                 gg_params_args pg, gg_ride_args ride, int nspec, unsigned vtag) {
  if (g.rs[4]) return;
  constexpr int B = gg_tB(NT);
  const int bid = (int)blockIdx.x;
  const long long t0 = ride.dbg ? wall_clock64() : 0;
  if (ride.dbg && threadIdx.x == 0) atomicMin(reinterpret_cast<unsigned long long*>(&ride.dbg[0]), (unsigned long long)t0);
  int role;
  if (bid >= ride.front) {
    role = 3;
    if (ride.dbg && threadIdx.x == 0) atomicMin(reinterpret_cast<unsigned long long*>(&ride.dbg[4]), (unsigned long long)t0);
    if (!ADJ && GG_W_RIDER_PRIO) __builtin_amdgcn_s_setprio(GG_W_RIDER_PRIO);
    gg_ride<ADJ, B, NTL == 1>(ride);
  } else if (bid >= nspec) {
    role = 2;
    __builtin_amdgcn_s_setprio(ADJ ? GG_ADJ_FRONT_PRIO : 2);   // the latency chain goes first at every issue slot it can use
    gg_params_body<ADJ, B>(pg, bid - nspec, (unsigned)(ride.front - nspec), vtag);
  } else {
    role = 1;
    __builtin_amdgcn_s_setprio(ADJ ? GG_ADJ_FRONT_PRIO : 2);
    gg_spec_resolve_body<NT, ADJ, GG_RIDE_UJ, SPLIT>(g, stage_tables, ap, (unsigned)nspec, vtag);
  }
  if (ride.dbg && threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(&ride.dbg[role]), (unsigned long long)wall_clock64());
}

// ---- The call's LAST sweep with its epilogue in the same launch (round 3; the reference's loop `model.resample(data);
// lps.append(model.log_probability(data))`, examples/demo_model_comparison.py:69-72, one sweep per call).  Block ranges:
//   [0, nspec)              the speculative scan of the sweep                     (gg_spec_resolve_body, publishes the verdict)
//   [nspec, nspec + K K)    one workgroup per block pair: the pair's table from the statistics the riders of the previous
//                           launch counted for the NEXT sweep (= the labels this scan leaves, iff it changes none) and the
//                           pair's log-likelihood term (abstractions.py:111-115 / weights.py:316-331, as k_loglik forms it)
//   nspec + K K             resample_m (adjacency.py:697-702; as k_draw_m)
// all of them gated on the scan's verdict.  The pair workgroup that finishes last sums the terms in k_loglik's order (same
// bits), adds the diagonal term and writes, into PINNED HOST memory: the log-likelihood, the control block as the resolver
// left it and the launch's tag - the host then needs no copy of the control block, no k_draw_m and no log-likelihood launch:
// one stream synchronisation completes `resample` and answers the `log_likelihood` that follows.  A scan that changes a
// label stalls the pipeline as ever; nothing of the epilogue is written and the host takes the usual path.
struct gg_epi_args {
  double* terms;              // [K K] device scratch: the pairs' terms
  int32_t* ticket;            // device: pair workgroups done
  int32_t* bad;               // device: a covariance was not positive definite
  double* out;                // pinned host: [0] log-likelihood, [1] bad flag, [2] tag, then 24 control words (as int32) from [4] on
  int want_ll;                // 0: only resample_m and the control block's mirror
  int npair;                  // pair workgroups in the launch: K K, or 1 without the log-likelihood
  uint32_t m_site;
};

template <bool ADJ, int B>
__device__ __forceinline__ void gg_epilogue_body(const gg_params_args& g, const gg_ll_args& la, const gg_epi_args& ep, const int t,
                                                 const unsigned tag) {
  const int K = g.K, KK = K * K, tid = (int)threadIdx.x;
  constexpr int NSI = ADJ ? 0 : B + B * (B + 1) / 2;
  constexpr int NTAB = ADJ ? 1 : 1 + 2 * NSI;
  __shared__ long long s_red[4][NTAB];
  __shared__ long long s_tab[NTAB];
  __shared__ double s_part[256];
  __shared__ int s_flag;
  if (t == ep.npair) {
    // resample_m: the block sizes are final once the scan is known to have changed nothing
    if (!gg_wait_verdict(g.ctl, g.verdict, tag)) return;
    __shared__ double s_g[256];
    if (tid < K) s_g[tid] = gg_dirichlet_gamma_at(g.seed, ep.m_site, g.sweep, tid, g.pi[tid] + (double)g.cnt[tid]);
    __syncthreads();
    if (tid < K) {
      double tot = 0.0;
      for (int j = 0; j < K; ++j) tot += s_g[j];
      g.m[tid] = s_g[tid] / tot;
    }
    return;
  }
  const int a = t / K, b = t % K;
  double term = 0.0;
  bool bad = false;
  if (ep.want_ll) {
    // the pair's table, reduced as gg_params_body reduces it (the same integers whatever the order)
    const int s0 = g.lab_start[a], s1 = g.lab_start[a + 1];
    long long acc[NTAB];
#pragma unroll
    for (int q = 0; q < NTAB; ++q) acc[q] = 0;
    for (int r = s0 + tid; r < s1; r += 256) {
      const int n = g.perm[r];
      acc[0] += g.Gcnt0[(size_t)b * g.Np + n];
      if (!ADJ)
#pragma unroll
        for (int s2 = 0; s2 < NSI; ++s2) {
          const long long v = g.GS0[((size_t)b * g.Np + n) * NSI + s2];
          acc[1 + 2 * s2] += (v >> 32);
          acc[2 + 2 * s2] += (v & 0xffffffffll);
        }
      if (g.add_diag && a == b && g.dA[n]) {
        acc[0] += 1;
        if (!ADJ) {
          const int32_t* q = g.dWq + (size_t)n * B;
          int tt = 0;
          for (int r1 = 0; r1 < B; ++r1, ++tt) {
            const long long v = q[r1];
            acc[1 + 2 * tt] += (v >> 32);
            acc[2 + 2 * tt] += (v & 0xffffffffll);
          }
          for (int r1 = 0; r1 < B; ++r1)
            for (int r2 = r1; r2 < B; ++r2, ++tt) {
              const long long v = (long long)q[r1] * (long long)q[r2];
              acc[1 + 2 * tt] += (v >> 32);
              acc[2 + 2 * tt] += (v & 0xffffffffll);
            }
        }
      }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < NTAB; ++q) {
      const long long v = gg_wave_sum_ll(acc[q]);
      if (lane == 0) s_red[wave][q] = v;
    }
    __syncthreads();
    if (tid < NTAB) s_tab[tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    __syncthreads();
    if (tid == 0) {
      if (ADJ) {
        const long long n1 = s_tab[0];
        const long long pairs = (long long)la.cnt[a] * la.cnt[b] - ((a == b && la.special) ? la.cnt[a] : 0);
        term = gg_ll_bern(la.p[t], n1, pairs - n1);
      } else {
        const double sc1 = gg_pow2i(-la.shift), sc2 = gg_pow2i(-2 * la.shift);
        const int Bx = la.wide ? 1 : B;      // (wide weights: B = 2 statistics, scalar parameters)
        term = gg_ll_gauss(la.mu + (size_t)t * Bx, la.sigma + (size_t)t * Bx * Bx, B, NSI, (double)s_tab[0], s_tab + 1, sc1, sc2, &bad, la.wide);
      }
    }
  }
  if (tid == 0) {
    // The terms do not wait for the verdict (the scratch they go to is this launch's own): only the workgroup that sums
    // them does, so one lane polls instead of K K and the exchange below runs while the scan is still being resolved.
    // (exchanges at agent scope, performed at the memory side; their return values hold the wave until they have been -
    // the term is out before the ticket is drawn.  No fence: gg_block_kernels.h, k_shard_eval)
    unsigned long long o1 = atomicExch(reinterpret_cast<unsigned long long*>(&ep.terms[t]), (unsigned long long)__double_as_longlong(term));
    int o2 = bad ? atomicOr(ep.bad, 1) : 0;
    asm volatile("" ::"v"(o1), "v"(o2) : "memory");
    s_flag = atomicAdd(ep.ticket, 1) == ep.npair - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_flag) return;
  // ---- the last pair workgroup: the sum in k_loglik's order, the diagonal term, the host's copy of the control block
  double acc = 0.0;
  if (ep.want_ll)
    for (int ab = tid; ab < KK; ab += 256)
      acc += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&ep.terms[ab]), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
  s_part[tid] = acc;
  __syncthreads();
  for (int s2 = 128; s2 >= 1; s2 >>= 1) {
    if (tid < s2) s_part[tid] += s_part[tid + s2];
    __syncthreads();
  }
  if (tid == 0) {
    __hip_atomic_store(ep.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (re-armed whatever the verdict)
    s_flag = gg_poll_verdict(g.ctl, g.verdict, tag, 0) ? 1 : 0;
  }
  __syncthreads();
  if (!s_flag) {      // the scan changed a label: no record (the host copies the control block and takes its usual path)
    if (tid == 0) __hip_atomic_store(ep.bad, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (tid < 24) {
    const int v = __hip_atomic_load(&g.ctl[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    reinterpret_cast<int32_t*>(ep.out + 4)[tid] = v;
  }
  if (tid == 0) {
    double ll = s_part[0];
    int isbad = __hip_atomic_load(ep.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ep.want_ll) {
      if (ADJ) {
        if (la.diag_zero) ll += gg_ll_bern(0.0, la.d1, la.d0);
        else if (la.special) ll += gg_ll_bern(*la.p_self, la.d1, la.d0);
      } else if (la.special) {
        const double sc1 = gg_pow2i(-la.shift), sc2 = gg_pow2i(-2 * la.shift);
        long long hl[2 * (GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)];
        for (int s2 = 0; s2 < la.NSI; ++s2) { hl[2 * s2] = la.dS[s2] >> 32; hl[2 * s2 + 1] = la.dS[s2] & 0xffffffffll; }
        bool b2 = false;
        ll += gg_ll_gauss(la.mu_self, la.sigma_self, la.B, la.NSI, (double)la.d1, hl, sc1, sc2, &b2, la.wide);
        if (b2) isbad = 1;
      }
    }
    if (isbad) __hip_atomic_store(ep.bad, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ep.out[0] = ll;
    ep.out[1] = isbad ? 1.0 : 0.0;
  }
  __syncthreads();
  __threadfence_system();
  if (tid == 0) ep.out[2] = (double)tag;      // (last: the host takes the record only with its launch's tag on it)
}

template <int NT, bool ADJ, int SPLIT>
__global__ __launch_bounds__(256) void k_spec_resolve_epi(gg_resolve_args g, int stage_tables, int32_t* __restrict__ ap,
                                                          gg_params_args pg, gg_ll_args la, gg_epi_args ep, int nspec, unsigned vtag) {
  if (g.rs[4]) return;   // behind a stalled scan of this pipeline: no-op (uniform)
  if ((int)blockIdx.x >= nspec) {
    gg_epilogue_body<ADJ, (gg_tB(NT))>(pg, la, ep, (int)blockIdx.x - nspec, vtag);
    return;
  }
  gg_spec_resolve_body<NT, ADJ, 0, SPLIT>(g, stage_tables, ap, (unsigned)nspec, vtag);
}

// Apply the change list to the resident statistics of every node (chip-wide):
// after it G is again the exact block histogram under the current labels.
// Node n has entries [0, ap[n]) applied already (restart rounds apply L lazily to the nodes they
// re-speculate); afterwards L is dead and k_reset_L empties it.
template <int NT, bool ADJ>
__global__ __launch_bounds__(256) void k_apply_L(gg_node_ctx x, const int32_t* __restrict__ L,
                                                 const int32_t* __restrict__ rs, int32_t* __restrict__ ap, int32_t* Gc0,
                                                 int32_t* Gc1, long long* Gs0, long long* Gs1, const uint32_t* XA,
                                                 const uint32_t* XAT, const int32_t* XW, const int32_t* XWT, int mode) {
  // mode 0: flush between restart rounds (nothing to do once the scan is done); mode 1: end-of-scan flush of
  // the incremental-statistics pipeline (nothing to do behind a stalled scan: the slow path flushes); 2: always;
  // mode 3: the changes a completed k_scan made (rs[25] entries) into the buffer its riders counted the NEXT sweep's
  // statistics into under the labels the scan started with - every entry for every node, `ap` untouched; nothing behind a
  // stalled scan (the slow path drops what was counted ahead)
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= x.N) return;
  if (mode == 0 ? rs[7] != 0 : ((mode == 1 || mode == 3) ? rs[4] != 0 : false)) return;
  const int nL = mode == 3 ? rs[25] : rs[1], Np = x.Np;
  for (int e = mode == 3 ? 0 : ap[n]; e < nL; ++e) {
    const int n2 = L[e * 3], a = L[e * 3 + 1], b = L[e * 3 + 2];
    const size_t at = (size_t)n2 * Np + n;
    if (ADJ) {
      int xo = gg_abit(XAT, at) ? 1 : 0, xi = gg_abit(XA, at) ? 1 : 0;
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
  if (mode != 3) ap[n] = 0;
}

// k_apply_L's mode 3 as a kernel of its own (round 4): the changes a completed k_scan made (rs[GG_RS_CARRY] entries of L) moved
// in the buffer the NEXT sweep's statistics were counted into under the labels the scan started with.  k_apply_L walks the list
// once per node with plain read-modify-writes - a chain of dependent memory round trips: 125 us for 52 changes at N=4096.  Here the
// entries are dealt out over blockIdx.y and applied with integer atomics (no return value, nothing to wait for): any order gives
// the same integers.  Nothing behind a stalled scan (the slow path drops what was counted ahead).
template <int NT, bool ADJ>
__global__ __launch_bounds__(256) void k_apply_moves(gg_node_ctx x, const int32_t* __restrict__ L, const int32_t* __restrict__ rs,
                                                     int32_t* Gc0, int32_t* Gc1, long long* Gs0, long long* Gs1,
                                                     const uint32_t* XA, const uint32_t* XAT, const int32_t* XW, const int32_t* XWT) {
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= x.N || rs[4] != 0) return;
  const int nL = rs[25 /* GG_RS_CARRY, gg_scan_kernels.h */], Np = x.Np;
  for (int e = blockIdx.y; e < nL; e += gridDim.y) {
    const int n2 = L[e * 3], a = L[e * 3 + 1], b = L[e * 3 + 2];
    const size_t at = (size_t)n2 * Np + n;
    if (ADJ) {
      if (gg_abit(XAT, at)) { atomicSub(&Gc0[(size_t)a * Np + n], 1); atomicAdd(&Gc0[(size_t)b * Np + n], 1); }
      if (gg_abit(XA, at)) { atomicSub(&Gc1[(size_t)a * Np + n], 1); atomicAdd(&Gc1[(size_t)b * Np + n], 1); }
    } else {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int32_t* q = (d == 0 ? XWT : XW) + at * B;
        int32_t qq[B];
#pragma unroll
        for (int r = 0; r < B; ++r) qq[r] = q[r];
        if (qq[0] == GG_NO_EDGE) continue;
        int32_t* Gc = d == 0 ? Gc0 : Gc1;
        long long* Gs = d == 0 ? Gs0 : Gs1;
        atomicSub(&Gc[(size_t)a * Np + n], 1);
        atomicAdd(&Gc[(size_t)b * Np + n], 1);
        long long v[NSI > 0 ? NSI : 1];
        int tt = 0;
#pragma unroll
        for (int r = 0; r < B; ++r) v[tt++] = qq[r];
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int s = r; s < B; ++s) v[tt++] = (long long)qq[r] * (long long)qq[s];
#pragma unroll
        for (int s = 0; s < NSI; ++s) {
          atomicAdd(reinterpret_cast<unsigned long long*>(&Gs[((size_t)a * Np + n) * NSI + s]), (unsigned long long)(-v[s]));
          atomicAdd(reinterpret_cast<unsigned long long*>(&Gs[((size_t)b * Np + n) * NSI + s]), (unsigned long long)v[s]);
        }
      }
    }
  }
}

#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ void k_reset_L(int32_t* rs, int mode) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(mode == 0 ? rs[7] != 0 : (mode == 1 ? rs[4] != 0 : false))) {
    rs[1] = 0;
    int pos, nL, wlen, fin, seq;
    unsigned long long* st = reinterpret_cast<unsigned long long*>(&rs[GG_RS_STATE]);
    gg_unpack_state(*st, &pos, &nL, &wlen, &fin, &seq);
    *st = gg_pack_state(pos, 0, wlen, fin, seq);
  }
}
#endif

// One device-driven restart round in a single launch.  The scan stands at pos = rs[0] with the change
// list L[0, nL) (nL = rs[1]); the round covers the window [pos, pos + wlen):
//   every workgroup (groups of 256 / Kp nodes, grid-strided): brings its nodes' statistics up to date by
//     applying the entries L[ap[n], nL) they have not seen yet (+-1 / +-q corrections from single rows of
//     X and X^T, in LDS, written back to G), re-evaluates their conditionals and draws -> cand;
//   the workgroup that finishes last (ticket in rs[10]) plays the resolver: the first node of the
//     window whose draw differs from its label is the next exact change of the scan (every node
//     before it saw all earlier changes); it is appended to L and the scan moves behind it.  A window
//     without a change is final; the scan's tail runs when the last node is passed.
// Statistics of nodes outside the window stay stale until a round reaches them (or k_apply_L
// flushes L); nothing else reads G before the next bulk pass rebuilds it.
template <int NT, bool ADJ>
__global__ __launch_bounds__(256) void k_round(gg_resolve_args g, int32_t* __restrict__ ap, int32_t* Gc0, int32_t* Gc1,
                                               long long* Gs0, long long* Gs1, int stage_tables, int round_id) {
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
  constexpr int U = 4;   // corrections in flight per thread
#define GG_TS(slot) do { if (g.dbg && threadIdx.x == 0) g.dbg[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
  const long long ts0 = g.dbg ? wall_clock64() : 0;
  const gg_node_ctx& x = g.x;
  const int N = x.N, K = x.K, Kp = x.Kp, Np = x.Np;
  const int per = 256 / Kp;
  const int tid = threadIdx.x;

  extern __shared__ __align__(16) unsigned char s_raw[];
  // [dir][j][i] integer statistics, then their spec-order doubles [dir][j][i][NT], then tables when staged
  long long* s_gs = reinterpret_cast<long long*>(s_raw);
  // Weights with Kp = 64 and unstaged tables: the integer statistics are dead once they are converted and written back,
  // which is before the split chains need their meeting place (gg_lp_split4) - the two share one region, or B = 4 with
  // K > 40 would not fit in 160 KB of LDS.
  const size_t gs_bytes = ADJ ? 0 : (size_t)2 * K * per * NSI * sizeof(long long);
  const size_t meet_bytes = (size_t)4 * 2 * NT * 64 * sizeof(double);
  const bool alias_meet = !ADJ && !stage_tables && Kp == 64;
  size_t off = alias_meet ? (gs_bytes > meet_bytes ? gs_bytes : meet_bytes) : gs_bytes;
  double* s_st = reinterpret_cast<double*>(s_raw + off); off += (size_t)2 * K * per * NT * sizeof(double);
  double* s_T = alias_meet ? reinterpret_cast<double*>(s_raw) : reinterpret_cast<double*>(s_raw + off);
  const int ntab = K * NT * Kp;
  if (stage_tables) off += (size_t)2 * ntab * sizeof(double);
  else if (Kp == 64 && !alias_meet) off += meet_bytes;   // the split chains meet here (gg_lp_split4)
  int* s_gc = reinterpret_cast<int*>(s_raw + off); off += (size_t)2 * K * per * sizeof(int);
  int* s_ap = reinterpret_cast<int*>(s_raw + off); off += (size_t)per * sizeof(int);
  off = (off + 7) & ~(size_t)7;   // (s_ctl[0..1] is a 64-bit word for the LDS atomics: one node per workgroup - 129..256 blocks - left it on a 4-byte boundary)
  int* s_ctl = reinterpret_cast<int*>(s_raw + off); off += 8 * sizeof(int);
  int* s_cnt = reinterpret_cast<int*>(s_raw + off); off += (size_t)K * sizeof(int);   // [K] (resolver only)
  off = (off + 7) & ~(size_t)7;
  double* s_cat = reinterpret_cast<double*>(s_raw + off);   // more than 64 blocks: scratch of the cross-wave categorical
  // (node, old, new) payloads: 6-bit labels, 8-bit for more than 64 blocks
  const int LB = Kp > 64 ? 8 : 6;
  const unsigned LMASK = (1u << LB) - 1u;
  unsigned long long* s_min = reinterpret_cast<unsigned long long*>(s_ctl);   // s_ctl[0..1]

  // The whole round state is one 64-bit word: a workgroup that starts after this round's resolver has
  // published the next state sees a different round counter and leaves; one that starts before sees this
  // round's state, under which every active workgroup has already drawn its ticket, so it is inactive.
  int pos, nL, w8, fin, seq;
  gg_unpack_state(*reinterpret_cast<const volatile unsigned long long*>(&g.rs[GG_RS_STATE]), &pos, &nL, &w8, &fin, &seq);
  const long long ts1 = g.dbg ? wall_clock64() : 0;
  if (seq != (round_id & 0xfff) || fin) return;   // stale launch, or the scan is done: no-op
  int wlen = g.window;
  if (w8 > 0 && w8 < wlen) wlen = w8;
  const int wend = (pos + wlen < N) ? pos + wlen : N;
  const int nact = (wend - pos + per - 1) / per;
  const int nwg = nact < (int)gridDim.x ? nact : (int)gridDim.x;
  if ((int)blockIdx.x >= nwg) return;
  if (g.dbg && tid == 0) { g.dbg[(size_t)blockIdx.x * 8] = ts0; g.dbg[(size_t)blockIdx.x * 8 + 1] = ts1; }

  const size_t dstride = (size_t)K * Np;   // direction 1 follows direction 0 in the buffer (Gc1 == Gc0 + dstride)
  // Every launch starts with a cold L2: touch the coefficient tables now (one load per 128-byte line) so
  // that they arrive while the statistics are being brought up to date, not when the conditional needs them.
  long long pf = 0;
  if (!stage_tables) {
    const long long* to = reinterpret_cast<const long long*>(x.Tout);
    const long long* ti = reinterpret_cast<const long long*>(x.Tin);
    for (int idx = tid * 16; idx < ntab; idx += 256 * 16) pf ^= to[idx] ^ ti[idx];
  }
  if (stage_tables) {
    for (int idx = tid; idx < ntab; idx += 256) {
      s_T[idx] = x.Tout[idx];
      s_T[ntab + idx] = x.Tin[idx];
    }
  }
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  if (tid == 0) *s_min = GG_NO_CHANGE;
  for (int grp = blockIdx.x; grp < nact; grp += gridDim.x) {
    const int n0 = pos + grp * per;
    __syncthreads();   // the previous group's LDS is dead
    {
      // 2 K per <= 512 entries: at most two per thread; all their loads are issued before the first LDS store
      int gcv[2];
      long long gsv[2][NSI > 0 ? NSI : 1];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * 256;
        const bool in = idx < 2 * K * per;
        const int ii = in ? idx : 0;
        const int i = ii % per, j = (ii / per) % K, d = ii / (per * K);
        const int n = (n0 + i < N) ? n0 + i : N - 1;
        const size_t gi = d * dstride + (size_t)j * Np + n;
        gcv[u] = Gc0[gi];
        if (!ADJ)
#pragma unroll
          for (int s = 0; s < NSI; ++s) gsv[u][s] = Gs0[gi * NSI + s];
      }
      const int apv = (tid < per && n0 + tid < wend) ? ap[n0 + tid] : nL;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * 256;
        if (idx < 2 * K * per) {
          s_gc[idx] = gcv[u];
          if (!ADJ)
#pragma unroll
            for (int s = 0; s < NSI; ++s) s_gs[(size_t)idx * NSI + s] = gsv[u][s];
        }
      }
      if (tid < per) s_ap[tid] = apv;
    }
    if (tid == 0) s_ctl[5] = nL;
    __syncthreads();
    if (tid < per) atomicMin(&s_ctl[5], s_ap[tid]);
    __syncthreads();
    const int e0 = s_ctl[5];
    GG_TS(2);
    // pending corrections: item = (entry e, node i, direction d); U independent items per thread in flight.
    // d = 0: edge (n0+i) -> n2 read from X^T row n2; d = 1: edge n2 -> (n0+i) read from X row n2
    const int nitems = (nL - e0) * per * 2;
    for (int base = tid; base < nitems; base += 256 * U) {
      int ia[U], ib[U], ii[U], dd[U];
      size_t at[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * 256;
        const int d = idx & 1, i = (idx >> 1) % per, e = e0 + (idx >> 1) / per;
        ok[u] = idx < nitems && e >= s_ap[i];
        const int ee = ok[u] ? e : 0;
        const int n2 = g.L[ee * 3];
        ia[u] = g.L[ee * 3 + 1]; ib[u] = g.L[ee * 3 + 2];
        ii[u] = i; dd[u] = d;
        at[u] = ok[u] ? (size_t)n2 * Np + n0 + i : 0;
      }
      if (ADJ) {
        uint8_t xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = gg_abit(dd[u] == 0 ? g.XAT : g.XA, at[u]) ? 1 : 0;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ok[u] && xv[u] != 0) {
            atomicSub(&s_gc[(dd[u] * K + ia[u]) * per + ii[u]], 1);
            atomicAdd(&s_gc[(dd[u] * K + ib[u]) * per + ii[u]], 1);
          }
      } else {
        int32_t q[U][B];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t* src = (dd[u] == 0 ? g.XWT : g.XW) + at[u] * B;
#pragma unroll
          for (int r = 0; r < B; ++r) q[u][r] = src[r];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ok[u] && q[u][0] != GG_NO_EDGE) {
            atomicSub(&s_gc[(dd[u] * K + ia[u]) * per + ii[u]], 1);
            atomicAdd(&s_gc[(dd[u] * K + ib[u]) * per + ii[u]], 1);
            long long v[NSI > 0 ? NSI : 1];
            int t = 0;
#pragma unroll
            for (int r = 0; r < B; ++r) v[t++] = q[u][r];
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
              for (int s2 = r; s2 < B; ++s2) v[t++] = (long long)q[u][r] * (long long)q[u][s2];
#pragma unroll
            for (int s2 = 0; s2 < NSI; ++s2) {
              atomicAdd(reinterpret_cast<unsigned long long*>(&s_gs[((size_t)(dd[u] * K + ia[u]) * per + ii[u]) * NSI + s2]),
                        (unsigned long long)(-v[s2]));
              atomicAdd(reinterpret_cast<unsigned long long*>(&s_gs[((size_t)(dd[u] * K + ib[u]) * per + ii[u]) * NSI + s2]),
                        (unsigned long long)v[s2]);
            }
          }
      }
    }
    __syncthreads();
    GG_TS(3);
    // write back what changed, convert to spec-order doubles
    for (int idx = tid; idx < 2 * K * per; idx += 256) {
      const int i = idx % per, j = (idx / per) % K, d = idx / (per * K);
      const bool live = n0 + i < wend;
      const int n = (n0 + i < N) ? n0 + i : N - 1;
      const size_t gi = d * dstride + (size_t)j * Np + n;
      const int gc = s_gc[idx];
      const bool dirty = live && s_ap[i] < nL;
      if (dirty) Gc0[gi] = gc;
      int ts;
      double* dst = s_st + gg_st_slot(!stage_tables && Kp == 64, idx / per, i, per, NT, &ts);
      if (ADJ) {
        const int cx = x.cnt[j] - (x.c[x.n0 + n] == j ? 1 : 0);
        dst[0] = (double)gc;
        dst[ts] = (double)(cx - gc);
      } else {
        dst[0] = (double)gc;
#pragma unroll
        for (int s = 0; s < NSI; ++s) {
          const long long v = s_gs[(size_t)idx * NSI + s];
          if (dirty) Gs0[gi * NSI + s] = v;
          dst[(1 + s) * ts] = (double)v * (s < B ? sc1 : sc2);
        }
        if (gg_tJ(NT)) dst[(NT - 1) * ts] = (double)(x.cnt[j] - (x.c[x.n0 + n] == j ? 1 : 0));
      }
    }
    if (tid < per && n0 + tid < wend) ap[n0 + tid] = nL;
    __syncthreads();
    GG_TS(4);
    asm volatile("" ::"v"(pf));   // (keeps the prefetch loads alive)
    {
      const int i = tid / Kp, k = tid % Kp;
      const int n = n0 + i;
      const bool live = n < wend;
      const int nn = (n < N) ? n : N - 1;
      auto ld = [&](int j, double* so, double* si) {
        const double* a = s_st + (size_t)((0 * K + j) * per + i) * NT;
        const double* b = s_st + (size_t)((1 * K + j) * per + i) * NT;
#pragma unroll
        for (int t = 0; t < NT; ++t) { so[t] = a[t]; si[t] = b[t]; }
      };
      double lp;
      if (stage_tables) lp = gg_node_lp_core<NT, ADJ>(x, nn, k, ld, s_T, s_T + ntab);
      else if (Kp == 64) lp = gg_lp_split4<NT, ADJ>(x, s_st, s_T, nn);
      else lp = gg_node_lp_core<NT, ADJ>(x, nn, k, ld, x.Tout, x.Tin);
      double u = gg_uniform_at(g.seed, g.site, g.sweep, (uint32_t)(x.n0 + nn));
      // (more than 64 blocks: a node's lanes span two or four wavefronts - round 3: the restart rounds serve that shape too)
      int lab = Kp > 64 ? gg_group_categorical_wide(lp, k, K, Kp, u, s_cat) : gg_group_categorical(lp, k, K, Kp, u);
      if (live && k == 0) {
        const int cn = x.c[x.n0 + n];
        if (lab != cn)
          atomicMin(s_min, ((unsigned long long)(unsigned)n << (2 * LB)) | ((unsigned long long)(unsigned)cn << LB) | (unsigned)lab);
      }
    }
  }
  // ---- last workgroup out resolves the window.  Each workgroup reports its first changer with a
  // device-scope atomic min and only then (the min has returned) draws its ticket: whoever sees the last
  // ticket sees every report.
  GG_TS(5);
  __syncthreads();
  if (tid == 0) {
    const unsigned long long mine = *s_min;
    if (mine != GG_NO_CHANGE) {
      unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_MIN]), mine);
      asm volatile("" ::"v"(old) : "memory");   // the ticket is not drawn before the min has been performed
    }
    s_ctl[6] = (atomicAdd(&g.rs[10], 1) == nwg - 1) ? 1 : 0;
  }
  __syncthreads();
  GG_TS(6);
  if (!s_ctl[6]) return;
  // resolver (uniform values, one lane writes)
  unsigned long long best = GG_NO_CHANGE;
  if (tid == 0) {
    best = atomicExch(reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_MIN]), GG_NO_CHANGE);
    *s_min = best;
  }
  __syncthreads();
  best = *s_min;
  int newpos, nchg = 0, wnext;
  if (best == GG_NO_CHANGE) {
    newpos = wend;                              // the whole window is final, unchanged
    wnext = 0;                                  // quiet stretch: full window next
  } else {
    const int ns = (int)(best >> (2 * LB)), a = (int)((best >> LB) & LMASK), b = (int)(best & LMASK);
    if (tid == 0) {
      g.c[ns] = b;
      atomicSub(&g.cnt[a], 1);
      atomicAdd(&g.cnt[b], 1);
      g.L[nL * 3] = ns; g.L[nL * 3 + 1] = a; g.L[nL * 3 + 2] = b;
    }
    int w = 4 * (ns - pos + 1);
    wnext = w < 256 ? 256 : (w > g.window ? g.window : w);
    nchg = 1;
    newpos = ns + 1;
  }
  const bool done = newpos >= N;
  if (done) {
    __threadfence();
    __syncthreads();
    for (int k = tid; k < K; k += 256) s_cnt[k] = __hip_atomic_load(&g.cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    gg_scan_tail(g, s_cnt, s_raw, g.rs[9] + nchg == 0);
  }
  if (tid == 0) {
    g.rs[0] = done ? 0 : newpos;
    g.rs[1] = (done && !g.keep_L) ? 0 : nL + nchg;
    g.rs[2] += nchg;
    g.rs[9] += nchg;
    g.rs[8] = wnext;
    __hip_atomic_store(&g.rs[10], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done) { g.rs[5] += 1; g.rs[7] = 1; }
    else g.rs[4] = 1;
    *reinterpret_cast<volatile unsigned long long*>(&g.rs[GG_RS_STATE]) =
        gg_pack_state(done ? 0 : newpos, done ? 0 : nL + nchg, wnext, done ? 1 : 0, round_id + 1);
    if (g.dbg) { g.dbg[(size_t)blockIdx.x * 8 + 7] = wall_clock64(); g.dbg[512 * 8] = blockIdx.x; g.dbg[512 * 8 + 1] = nwg; }
  }
#undef GG_TS
}
