// gg_scan_kernels.h — the whole sequential node scan (adjacency.py:650-695, weights.py:415-488) in ONE launch.
//
// k_speculate + k_resolve finish a scan in-stream only when label changes are rare; otherwise the host has to
// drive restart rounds (k_round), one launch per accepted change.  k_scan keeps the grid resident instead:
//   * G workgroups (<= 256, one per CU); workgroup b owns node groups b, b + G, ... (256 / Kp nodes each) and
//     works on band r = groups [r G, (r+1) G) together with all the others;
//   * per iteration every workgroup re-evaluates the nodes of its group that lie at or behind the scan position
//     (statistics and, when they fit, coefficient tables stay in LDS), reports its first changer with a
//     device-scope atomicMin, and meets the others at a grid barrier (one counter, bounded spin);
//   * the minimum is the next exact change of the scan; the candidates of the OTHER workgroups are kept as
//     hypotheses (below), every workgroup applies all of them to its own group's statistics (one bit / one
//     int32 per node and direction from row f of X and X^T), the scan position moves behind the minimum,
//     and the loop continues; a band without a surprise is final and the grid moves on.
// Hypotheses.  A round's exchange shows every workgroup ALL first changers, not only the earliest.  The later
// ones were computed without the earlier ones, so they are not the chain yet - but a change elsewhere rarely
// flips a decision (that is what the speculative form rests on), so they are taken as a hypothesis H: node n
// sees the statistics of accepted changes + every h in H with position < n, is evaluated under them, and what a
// workgroup reports is its first SURPRISE - the first live node whose decision differs from what H says about
// it (a new changer, a hypothesised change that does not happen, or one that goes elsewhere).  The earliest
// surprise p* was evaluated under hypotheses that all lie before it and are thereby confirmed (induction
// over positions: the first is exact, so is the next, ...): everything of H before p* and the decision at p*
// are the chain, exactly; the other surprises amend H.  A round without a surprise confirms all of H.
// Rounds per band: 1 + the largest number of changers in one workgroup's group (+ the rare flips) instead of
// 1 + the number of changes.  Never more rounds than before (H empty = the old protocol).
// Cost: one barrier round (a few us) per round instead of a kernel launch (and, before, a host round
// trip) per accepted change, and no stall of the optimistic sweep pipeline.  Same statistics, same conditional, same
// draw as the sequential loop: bit-exact (tests/test_gpu_parity_sbm.py run through this kernel).
//
// Grid barrier: all G workgroups must be resident.  G <= 256 workgroups of 256 threads with <= 64 KB LDS is far
// below what the chip holds (even with the other model's kernels in flight), and every spin is bounded: on a
// timeout the kernel raises status bit 2 and leaves, the host reports it (no hang).
#pragma once
#include "gg_resolve_kernels.h"

#ifndef GG_SCAN_MOVES_U
#define GG_SCAN_MOVES_U 4   /* (8: band starts 3 % shorter at N=65536, every round's moves 0.4 us longer - N=4096 mixing chains 3 % slower) */
#endif
#define GG_SCAN_MAXWG 256
#define GG_BAR_SPIN_LIMIT (1 << 19)   // about a quarter of a second of polling: far beyond any kernel the grid may have to wait behind
#ifndef GG_SCAN_POLL_SLEEP
#define GG_SCAN_POLL_SLEEP 4
#endif

struct gg_scan_args {
  gg_resolve_args r;          // labels, tables, data rows, RNG addressing, tail outputs (perm / lab_start)
  int32_t* ap;                // entries of L already contained in a node's written-back statistics
  int32_t* Gc0;               // statistics buffers (direction 1 follows direction 0)
  long long* Gs0;
  unsigned long long* bar;    // G exchange slots (tag << 48 | payload), zeroed by k_params before every scan
  int Lcap;
  int G;                      // workgroups in the grid
  int resume;                 // 1: finish a scan that k_resolve left stalled at rs[0] (G is exact for the current labels, L empty)
  int spin_limit;             // polls before an exchange gives up (GG_BAR_SPIN_LIMIT; GG_SCAN_SPIN overrides it for tests)
  int mf_small;               // 9..32 blocks, tables in LDS: the conditional on the matrix cores (gg_lp_mfma_small; its meeting place is in the LDS request)
  // RIDE: the NEXT sweep's bulk pass rides behind the G scan workgroups, counting into the next statistics buffer under the
  // labels this scan STARTS with; the scan's change list (rs[GG_RS_CARRY] entries of g.L when it ends) then turns those
  // statistics into the ones under the labels it ends with (k_apply_L mode 3): integer moves, exact.
  gg_ride_args ride;
  int32_t* Gn0;               // the next sweep's statistics buffer (what the riders fill), same layout as Gc0 / Gs0
  long long* Gsn0;
  double* epi_out;      // pinned host record for the call's last scan (control block + tag, as k_spec_resolve_epi leaves it), or null
  unsigned epi_tag;
};
#define GG_RS_CARRY 25        // control block: entries of g.L that hold the changes of the last completed k_scan since its last flush

// Grid-wide exchange that doubles as the barrier: workgroup b publishes (round tag << 48 | payload) in its own
// slot with one write-through store; everybody polls all G slots (one coalesced load per poll, lane t reads
// slot t) until every tag equals this round's, and takes the minimum payload.  No read-modify-write atomics, no
// counter to reset; the slots (2 x GG_SCAN_MAXWG) are zeroed (tag 0 = never) by k_params before every scan.
// Returns false on timeout.
#define GG_SLOT_NONE 0xffffffffffffull
// fenced = false (the common case): nothing but the slots themselves crosses workgroups - every change travels in the
// payload - so no release / acquire fence is needed (an agent-scope fence pair costs 6-18 us with 64-256 workgroups, the
// bare exchange 1-3 us: tools/probe/xcd_barrier.hip).  fenced = true: plain global writes made before the exchange
// (statistics flushed into G) must be visible to the other workgroups after it.
__device__ __forceinline__ bool gg_grid_exchange(unsigned long long* slots, int G, unsigned tag, unsigned long long mine,
                                                 bool fenced, unsigned long long* s_min, int32_t* rs, unsigned long long* best,
                                                 int spin_limit) {
  const int tid = threadIdx.x;
  // two slot arrays, alternating by round: a workgroup can be one round ahead of the slowest one (never two), and
  // must not overwrite the value the slow ones are still polling for
  slots += (size_t)(tag & 1u) * GG_SCAN_MAXWG * GG_SLOT_PAD;
  __syncthreads();
  if (tid == 0) {
    *s_min = GG_SLOT_NONE;
    if (fenced) __threadfence();   // plain global writes of this workgroup come first
    __hip_atomic_store(&slots[(size_t)blockIdx.x * GG_SLOT_PAD], ((unsigned long long)(tag & 0xffffu) << 48) | (mine & GG_SLOT_NONE), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  bool okall = false;
  // lane t watches slot t and stops loading it once its tag has arrived: the polls are memory-side loads that compete
  // with the other model's bulk pass, and most slots are ready long before the last one
  bool seen = tid >= G;
  unsigned long long v = GG_SLOT_NONE;
  for (int spin = 0; spin < spin_limit; ++spin) {
    if (!seen) {
      const unsigned long long w = __hip_atomic_load(&slots[(size_t)tid * GG_SLOT_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(w >> 48) == (tag & 0xffffu)) { seen = true; v = w & GG_SLOT_NONE; }
    }
    if (__syncthreads_and(seen ? 1 : 0)) {
      if (v != GG_SLOT_NONE) atomicMin(s_min, v);
      okall = true;
      break;
    }
    __builtin_amdgcn_s_sleep(GG_SCAN_POLL_SLEEP);
  }
  if (!okall && tid == 0) atomicOr(&rs[3], 2);
  if (fenced) __threadfence();
  __syncthreads();
  *best = *s_min;
  __syncthreads();
  return okall;
}

// The round's exchange carries up to GG_SCAN_NW surprises per workgroup: words 0..NW-1 of its slot line, each tagged on its
// own (no ordering between them is needed), in position order - word 0 is the workgroup's earliest.  Lane t ends up with
// the words of slot t in v[]; *best is the minimum over all word 0s.  Slot order = position order within a band
// (workgroup t owns group band G + t).
#ifndef GG_SCAN_NW
#define GG_SCAN_NW 2
#endif
__device__ __forceinline__ bool gg_grid_exchange_multi(unsigned long long* slots, int G, unsigned tag, const unsigned long long* s_mine,
                                                       unsigned long long* s_min, int32_t* rs, unsigned long long* best,
                                                       int spin_limit, unsigned long long (&v)[GG_SCAN_NW]) {
  const int tid = threadIdx.x;
  slots += (size_t)(tag & 1u) * GG_SCAN_MAXWG * GG_SLOT_PAD;
  __syncthreads();
  if (tid == 0) *s_min = GG_SLOT_NONE;
  if (tid < GG_SCAN_NW)
    __hip_atomic_store(&slots[(size_t)blockIdx.x * GG_SLOT_PAD + tid], ((unsigned long long)(tag & 0xffffu) << 48) | (s_mine[tid] & GG_SLOT_NONE),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  bool okall = false;
  unsigned seen = tid >= G ? (1u << GG_SCAN_NW) - 1u : 0u;
#pragma unroll
  for (int w = 0; w < GG_SCAN_NW; ++w) v[w] = GG_SLOT_NONE;
  for (int spin = 0; spin < spin_limit; ++spin) {
    if (seen != (1u << GG_SCAN_NW) - 1u) {
      unsigned long long wd[GG_SCAN_NW];
#pragma unroll
      for (int w = 0; w < GG_SCAN_NW; ++w)
        wd[w] = __hip_atomic_load(&slots[(size_t)tid * GG_SLOT_PAD + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int w = 0; w < GG_SCAN_NW; ++w)
        if (!(seen & (1u << w)) && (unsigned)(wd[w] >> 48) == (tag & 0xffffu)) { seen |= 1u << w; v[w] = wd[w] & GG_SLOT_NONE; }
    }
    if (__syncthreads_and(seen == (1u << GG_SCAN_NW) - 1u ? 1 : 0)) {
      if (v[0] != GG_SLOT_NONE) atomicMin(s_min, v[0]);
      okall = true;
      break;
    }
    __builtin_amdgcn_s_sleep(GG_SCAN_POLL_SLEEP);
  }
  if (!okall && tid == 0) atomicOr(&rs[3], 2);
  __syncthreads();
  *best = *s_min;
  __syncthreads();
  return okall;
}

// MODE 1: the Kp = 64 evaluation without both table layouts in LDS (gg_lp_split4) - its own instantiation, so that the other
// shapes do not carry its registers.  MODE 2 ("wide"): Kp = 128 or 256 (65 <= K <= 256 blocks) - a node's lanes span two or
// four wavefronts: the categorical draw scans through LDS (gg_group_categorical_wide) and the (node, old, new) payloads carry
// 8-bit labels.  Two workgroups per CU in every case (both models' scan grids resident together).
template <int NT, bool ADJ, int MODE, int RIDE = 0>
__global__ __launch_bounds__(256, 2) void k_scan(gg_scan_args a) {
  constexpr bool WIDE = MODE == 2;
  if (RIDE && (int)blockIdx.x >= a.G) {
    // riders (the scan workgroups come first in the grid: resident before anything that shares the chip with them)
    if (!a.resume && a.r.rs[4]) return;
    gg_ride<ADJ, (gg_tB(NT)), RIDE == 2>(a.ride);
    return;
  }
  constexpr int LB = WIDE ? 8 : 6;                       // bits per label in a payload
  constexpr unsigned LMASK = (1u << LB) - 1u;
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
  constexpr int U = GG_SCAN_MOVES_U;   // (moves of a list in flight per thread: apply_moves)
  const gg_resolve_args& g = a.r;
  const gg_node_ctx& x = g.x;
  const long long t_kernel0 = g.dbg ? wall_clock64() : 0;
  if (!a.resume && g.rs[4]) return;   // an earlier scan of this pipeline is unfinished (uniform: set by a previous launch)
  const int N = x.N, K = x.K, Kp = x.Kp, Np = x.Np;
  // nodes per group: one pass of 256 / Kp nodes; more than 64 blocks with NT <= 3: TWO passes' worth (GG_SCAN_WIDE_P) - a lane
  // keeps the chains of two nodes and a coefficient-table entry fetched from L2 serves both (the quiet scan's
  // gg_spec_eval_wide takes four; here both models' grids must fit a CU's LDS together).  Half the bands, half the rounds.
  // 33..64 blocks (MODE 1), NT <= 3: 8 nodes per group (adjacency) or 16 (one-dimensional weights), evaluated together on the
  // matrix cores (gg_lp_mfma, round 4; before: two passes of gg_lp_split4 over four nodes each - N=65536, K=64: 67 rounds of
  // 15.9 us of evaluation).
  // MODE 3 = MODE 1 with 4 GG_SCAN_ADJ_PW = 16 nodes per group whatever NT: the adjacency model's scan under pair scheduling
  // (gg_model.hip, launch_scan) - it is resident while the weight model's pass streams, never next to the weight model's scan,
  // so its LDS need not leave room for that one, and half the bands and rounds shorten the time it takes compute units away
  // from the pass (32 nodes per group: measured no better).
  constexpr bool M13 = MODE == 1 || MODE == 3;
  constexpr int PW = WIDE ? GG_SCAN_WIDE_P(NT) : (MODE == 3 ? GG_SCAN_ADJ_PW : (MODE == 1 ? GG_SCAN_SPLIT_P(NT) : 1));
  const int per = (256 / Kp) * PW;
  const int tid = threadIdx.x, b = blockIdx.x, G = a.G;
  const int ngroups = (N + per - 1) / per;
  const int stage_tables = g.stage_tables;

  extern __shared__ __align__(16) unsigned char s_raw[];
  long long* s_gs = reinterpret_cast<long long*>(s_raw);
  size_t off = ADJ ? 0 : (size_t)2 * K * per * NSI * sizeof(long long);
  double* s_st = reinterpret_cast<double*>(s_raw + off); off += (size_t)2 * K * per * NT * sizeof(double);
  // stage_tables: 1 the two lane-major layouts of the coefficient table in LDS, 2 one padded copy of T[a][b][t] (half the
  // bytes: K = 32..64), 0 none (table entries from global memory)
  double* s_T = reinterpret_cast<double*>(s_raw + off);
  const int ntab = K * NT * Kp;
  const int RS = (K * NT) | 1;
  if (stage_tables == 1) off += (size_t)2 * ntab * sizeof(double);
  else if (stage_tables == 2) off += (size_t)K * RS * sizeof(double);
  double* s_acc = reinterpret_cast<double*>(s_raw + off);     // gg_lp_split4: where the chains of the four waves meet
  constexpr bool split = M13;         // (the host: stage_tables != 1 && Kp == 64)
  if (split) off += MODE == 3 ? (size_t)4 * GG_SCAN_ADJ_PW * 64 * sizeof(double) : GG_SCAN_MEET_BYTES(NT);
  else if (a.mf_small) off += (size_t)2 * NT * 256 * sizeof(double);   // (gg_lp_mfma_small: [chain][node][Kp], per Kp = 256)                   // (4 PW nodes at once: gg_lp_mfma leaves lp[node][64] here)
  int* s_gc = reinterpret_cast<int*>(s_raw + off); off += (size_t)2 * K * per * sizeof(int);
  int* s_c = reinterpret_cast<int*>(s_raw + off); off += (size_t)per * sizeof(int);
  int* s_cnt = reinterpret_cast<int*>(s_raw + off); off += (size_t)K * sizeof(int);
  off = (off + 7) & ~(size_t)7;
  unsigned long long* s_min = reinterpret_cast<unsigned long long*>(s_raw + off); off += 2 * sizeof(unsigned long long);
  int* s_L = reinterpret_cast<int*>(s_raw + off); off += (size_t)a.Lcap * 3 * sizeof(int);   // [Lcap][3]: this workgroup's own copy of the change list
  off = (off + 7) & ~(size_t)7;
  unsigned long long* s_dev = reinterpret_cast<unsigned long long*>(s_raw + off); off += 128 * sizeof(unsigned long long);   // deviation of node i of the group (or none)
  unsigned long long* s_mine = reinterpret_cast<unsigned long long*>(s_raw + off); off += 8 * sizeof(unsigned long long);    // the first GG_SCAN_NW of them
  int* s_S = reinterpret_cast<int*>(s_raw + off); off += 256 * 3 * sizeof(int);      // its surprises, compacted: (node, from, to)
  int* s_cntn = reinterpret_cast<int*>(s_raw + off); off += (size_t)per * K * sizeof(int);   // block sizes as node i of the group sees them
  int* s_misc = reinterpret_cast<int*>(s_raw + off); off += 16 * sizeof(int);
  // (words 0..8 are scratch of the folds; an 8-byte aligned double inside words 10..13 holds 0.0 for gg_lp_mfma)
  double* zero8 = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(s_misc + 10) + 7) & ~(uintptr_t)7);
  unsigned short* s_H = reinterpret_cast<unsigned short*>(s_raw + off); off += (size_t)GG_SCAN_MAXWG * per * sizeof(short);   // [G per] hypothesis per node of the band: 0xffff none, else (old << 8 | new) (old != new: never 0xffff)
  off = (off + 3) & ~(size_t)3;
  unsigned* s_Hm = reinterpret_cast<unsigned*>(s_raw + off); off += (size_t)(GG_SCAN_MAXWG * per / 32 + 2) * sizeof(unsigned);   // one bit per node of the band: a hypothesis is in place
  off = (off + 7) & ~(size_t)7;
  double* s_cat = reinterpret_cast<double*>(s_raw + off);     // WIDE: scratch of the cross-wave categorical (2 x 256 doubles + 16 ints)

  const size_t dstride = (size_t)K * Np;
  // The first band's statistics: 2 K per <= 512 entries, at most two per thread.  Their loads are issued before the
  // tables are staged (all of a thread's loads in flight together) and stored to LDS when the band loop starts.
  const int band_first = ((a.resume ? g.rs[0] : 0) / per) / G;
  int pf_gc[2];
  long long pf_gs[2][NSI > 0 ? NSI : 1];
  {
    const int grp0 = band_first * G + b;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * 256;
      const bool in = idx < 2 * K * per && grp0 < ngroups;
      const int ii = in ? idx : 0;
      const int i = ii % per, j = (ii / per) % K, d = ii / (per * K);
      int n = (in ? grp0 * per : 0) + i;
      n = n < N ? n : N - 1;
      const size_t gi = d * dstride + (size_t)j * Np + n;
      pf_gc[u] = a.Gc0[gi];
      if (!ADJ)
#pragma unroll
        for (int s2 = 0; s2 < NSI; ++s2) pf_gs[u][s2] = a.Gs0[gi * NSI + s2];
    }
  }
  for (int k = tid; k < K; k += 256) s_cnt[k] = g.cnt[k];
  if (tid == 0) *zero8 = 0.0;
  if (stage_tables == 1) {
    for (int idx = tid; idx < ntab; idx += 256) { s_T[idx] = x.Tout[idx]; s_T[ntab + idx] = x.Tin[idx]; }
  } else if (stage_tables == 2) {
    for (int idx = tid; idx < K * K * NT; idx += 256) s_T[(size_t)(idx / (K * NT)) * RS + idx % (K * NT)] = x.T[idx];
  }
  // nothing of this scan's list is in G yet.  Every node's entry is zeroed by the workgroup that OWNS the node's group -
  // the one that writes it again at the end of the band - so that no address has two unordered writers (the common
  // exchange carries no fence; the flush path, which reads and writes other workgroups' entries, is fenced on both sides).
  for (int grp = b; grp < ngroups; grp += G)
    if (tid < per && grp * per + tid < N) a.ap[grp * per + tid] = 0;
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  const double logm_k = x.logm[(tid % Kp) < K ? tid % Kp : 0];   // (the same in every round: not behind each round's chains)
  const int kk_mf = 16 * (tid >> 6) + (tid & 15);                // (the label of this lane's results in gg_lp_mfma)
  const double logm_mf = x.logm[kk_mf < K ? kk_mf : 0];

  int pos = a.resume ? g.rs[0] : 0, nL = 0, nchg = 0;   // (resume: uniform, written by an earlier launch)
  const long long t_init0 = g.dbg ? wall_clock64() : 0;
  unsigned long long it = 0;     // barrier rounds so far (identical in every workgroup)
  int wi = 0;                    // evaluation rounds so far: selects the rotating first-changer word
  bool alive = true, fatal = false;   // fatal: gave up inside the chip-wide flush (G half updated: not resumable)
  bool flushed = false;               // the list was flushed at least once in this scan (same in every workgroup)
  long long t_eval = 0, t_xchg = 0, t_apply = 0, t_load = 0, t_e1 = 0, t_e2 = 0, t_e3 = 0, t_a1 = 0, t_a2 = 0;
  const long long t_begin = g.dbg ? wall_clock64() : 0;

  // move one edge's statistics of node i (LDS) from block ba to block bb
  auto move_edge = [&](int d, int i, int ba, int bb, const int32_t* q) {
    atomicSub(&s_gc[(d * K + ba) * per + i], 1);
    atomicAdd(&s_gc[(d * K + bb) * per + i], 1);
    if (!ADJ) {
      long long v[NSI > 0 ? NSI : 1];
      int t = 0;
#pragma unroll
      for (int r = 0; r < B; ++r) v[t++] = q[r];
#pragma unroll
      for (int r = 0; r < B; ++r)
#pragma unroll
        for (int s2 = r; s2 < B; ++s2) v[t++] = (long long)q[r] * (long long)q[s2];
#pragma unroll
      for (int s2 = 0; s2 < NSI; ++s2) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&s_gs[((size_t)(d * K + ba) * per + i) * NSI + s2]), (unsigned long long)(-v[s2]));
        atomicAdd(reinterpret_cast<unsigned long long*>(&s_gs[((size_t)(d * K + bb) * per + i) * NSI + s2]), (unsigned long long)v[s2]);
      }
    }
  };

  // the moves (f: from -> to) of a list applied to the statistics of my group's nodes (first node n0); behind_only: only to
  // the nodes behind f - a node sees the changes at positions before its own - together with the block sizes they see
  auto apply_moves = [&](const int* list, int count, int n0, bool behind_only) {
    if (ADJ && per <= 32 && (32 % per) == 0) {
      // Adjacency: the group's per nodes are per consecutive bits of ONE word of row f of the bit matrix (groups start at
      // multiples of per, per divides 32, rows are whole words): one load per (move, direction) instead of one per (move,
      // direction, node) - the same word, per times over.  Two items per thread in flight.
      const int nit = count * 2;
      for (int base = tid; base < nit; base += 512) {
        uint32_t wb[2];
        int n2v[2], iav[2], ibv[2], dv[2];
        bool okv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int it = base + u * 256;
          okv[u] = it < nit;
          const int e = okv[u] ? it >> 1 : 0;
          dv[u] = it & 1;
          n2v[u] = list[e * 3]; iav[u] = list[e * 3 + 1]; ibv[u] = list[e * 3 + 2];
          const size_t at = (size_t)n2v[u] * Np + n0;
          wb[u] = okv[u] ? (dv[u] == 0 ? g.XAT : g.XA)[at >> 5] >> (at & 31) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (!okv[u]) continue;
          // nodes of the group that exist and (behind_only) lie behind the moved node
          uint32_t live = per == 32 ? 0xffffffffu : ((1u << per) - 1u);
          if (n0 + per > N) live &= (N - n0) > 0 ? ((1u << (N - n0)) - 1u) : 0u;
          if (behind_only) {
            const int first = n2v[u] + 1 - n0;                 // first node index behind f
            if (first >= per) live = 0u;
            else if (first > 0) live &= ~((1u << first) - 1u);
          }
          uint32_t edges = wb[u] & live;
          while (edges) {
            const int i = __ffs(edges) - 1;
            edges &= edges - 1u;
            move_edge(dv[u], i, iav[u], ibv[u], nullptr);
          }
          if (behind_only && dv[u] == 0) {
            uint32_t lv = live;
            while (lv) {
              const int i = __ffs(lv) - 1;
              lv &= lv - 1u;
              atomicSub(&s_cntn[i * K + iav[u]], 1); atomicAdd(&s_cntn[i * K + ibv[u]], 1);
            }
          }
        }
      }
      return;
    }
    const int nitems = count * per * 2;
    for (int base = tid; base < nitems; base += 256 * U) {
      int ia[U], ib[U], ii[U], dd[U];
      size_t at[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * 256;
        const int d = idx & 1, i = (idx >> 1) % per, e = (idx >> 1) / per;
        const int ee = idx < nitems ? e : 0;
        const int n2 = list[ee * 3];
        ok[u] = idx < nitems && n0 + i < N && (!behind_only || n0 + i > n2);
        ia[u] = list[ee * 3 + 1]; ib[u] = list[ee * 3 + 2];
        ii[u] = i; dd[u] = d;
        at[u] = ok[u] ? (size_t)n2 * Np + n0 + i : 0;
      }
      if (ADJ) {
        uint8_t xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = gg_abit(dd[u] == 0 ? g.XAT : g.XA, at[u]) ? 1 : 0;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ok[u] && xv[u] != 0) move_edge(dd[u], ii[u], ia[u], ib[u], nullptr);
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
          if (ok[u] && q[u][0] != GG_NO_EDGE) move_edge(dd[u], ii[u], ia[u], ib[u], q[u]);
      }
      if (behind_only) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ok[u] && dd[u] == 0) { atomicSub(&s_cntn[ii[u] * K + ia[u]], 1); atomicAdd(&s_cntn[ii[u] * K + ib[u]], 1); }
      }
    }
  };

  for (int band = band_first; band * G < ngroups && alive; ++band) {
    const int grp = band * G + b;
    const bool has = grp < ngroups;
    const int n0 = grp * per;
    if (pos < band * G * per) pos = band * G * per;
    __syncthreads();
    if (has) {
      // the group's statistics as the bulk pass left them, plus every change accepted so far
      if (PW == 1 && band == band_first) {   // (two passes' worth per group: its 2 K per entries are more than two per thread)
#pragma unroll
        for (int u = 0; u < 2; ++u) {        // prefetched at kernel start
          const int idx = tid + u * 256;
          if (idx < 2 * K * per) {
            s_gc[idx] = pf_gc[u];
            if (!ADJ)
#pragma unroll
              for (int s2 = 0; s2 < NSI; ++s2) s_gs[(size_t)idx * NSI + s2] = pf_gs[u][s2];
          }
        }
      } else {
        // four entries per thread in flight together (2 K per = 2048 entries at K = 64 with 16 nodes per group: eight per
        // thread, each a 4-byte and NSI 8-byte loads - one at a time they were eight memory round trips per band start)
        constexpr int UL = 4;
        for (int base = tid; base < 2 * K * per; base += 256 * UL) {
          int gcv[UL];
          long long gsv[UL][NSI > 0 ? NSI : 1];
#pragma unroll
          for (int u = 0; u < UL; ++u) {
            const int idx = base + u * 256;
            const int ii = idx < 2 * K * per ? idx : 0;
            const int i = ii % per, j = (ii / per) % K, d = ii / (per * K);
            const int n = (n0 + i < N) ? n0 + i : N - 1;
            const size_t gi = d * dstride + (size_t)j * Np + n;
            gcv[u] = a.Gc0[gi];
            if (!ADJ)
#pragma unroll
              for (int s = 0; s < NSI; ++s) gsv[u][s] = a.Gs0[gi * NSI + s];
          }
#pragma unroll
          for (int u = 0; u < UL; ++u) {
            const int idx = base + u * 256;
            if (idx < 2 * K * per) {
              s_gc[idx] = gcv[u];
              if (!ADJ)
#pragma unroll
                for (int s = 0; s < NSI; ++s) s_gs[(size_t)idx * NSI + s] = gsv[u][s];
            }
          }
        }
      }
      if (tid < per) s_c[tid] = (n0 + tid < N) ? g.c[n0 + tid] : 0;
      __syncthreads();
      apply_moves(s_L, nL, n0, false);
    }
    if (g.dbg) t_load += wall_clock64() - t_begin - t_load - t_eval - t_xchg - t_apply;
    // the band starts without hypotheses; every node of the group sees the block sizes of the accepted labels
    const int band0 = band * G * per;
    const int band1 = (band0 + G * per < N) ? band0 + G * per : N;
    for (int idx = tid; idx < G * per; idx += 256) s_H[idx] = (unsigned short)0xffffu;
    for (int idx = tid; idx < (G * per + 31) / 32; idx += 256) s_Hm[idx] = 0u;
    for (int idx = tid; idx < per * K; idx += 256) s_cntn[idx] = s_cnt[idx % K];
    // the node's uniform is a function of (seed, site, sweep, node): drawn once per band, not once per round (ten Philox
    // rounds are a tenth of a round's evaluation)
    double u_bandp[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
      const int nb = n0 + (M13 ? 4 * p + (tid >> 6) : (tid / Kp) * PW + p);
      u_bandp[p] = gg_uniform_at(g.seed, g.site, g.sweep, (uint32_t)(x.n0 + ((has && nb < N) ? nb : N - 1)));
    }
    const double u_band = u_bandp[0];
    // the ordinary lane layout (node tid / Kp, label tid % Kp; MODE 0): the self connection's chain, once per band - its three
    // dependent loads (diagonal entry, weight, pair (k, k)'s coefficients) were behind every round's chains
    double self0 = 0.0;
    if (MODE == 0 && !x.special) {
      const int nb = n0 + tid / Kp, kb = tid % Kp;
      self0 = gg_node_self_acc<NT, ADJ>(x, (has && nb < N) ? nb : N - 1, kb < K ? kb : 0);
    }
    const double* self0_p = (MODE == 0 && !x.special) ? &self0 : nullptr;
    // gg_lp_mfma's lane layout: the self chains of this lane's (node, label) entries, once per band
    constexpr int NRT_MF = PW > 4 ? PW / 4 : 1;   // (16-row tiles of the group)
    double self_mf[4 * NRT_MF];
#pragma unroll
    for (int r = 0; r < 4 * NRT_MF; ++r) self_mf[r] = 0.0;
    if (M13 && PW > 1 && !x.special) {
#pragma unroll
      for (int r = 0; r < 4 * NRT_MF; ++r) {
        const int il = 16 * (r >> 2) + 4 * (r & 3) + ((tid >> 4) & 3);
        const int nb = n0 + il;
        self_mf[r] = gg_node_self_acc<NT, ADJ>(x, (has && il < per && nb < N) ? nb : N - 1, kk_mf < K ? kk_mf : 0);
      }
    }
    // ---- iterate until the band holds no further surprise
    // Spec-order doubles of the statistics: all of them when the band starts and after a round with several surprises;
    // after a single one (f: ca -> cb) only the entries of blocks ca and cb can differ - edges to / from f moved between
    // them and their sizes changed - so only those 4 * per entries are converted again.
    auto convert = [&](int idx) {
      const int i = idx % per, j = (idx / per) % K;
      const int gc = s_gc[idx];
      int ts;
      double* dst = s_st + gg_st_slot(split || WIDE, idx / per, i, per, NT, &ts);   // (WIDE: [dir][j][t][node] - a lane's nodes are one read)
      if (ADJ) {
        const int cx = s_cntn[i * K + j] - (s_c[i] == j ? 1 : 0);
        dst[0] = (double)gc;
        dst[ts] = (double)(cx - gc);
      } else {
        dst[0] = (double)gc;
#pragma unroll
        for (int s = 0; s < NSI; ++s) dst[(1 + s) * ts] = (double)s_gs[(size_t)idx * NSI + s] * (s < B ? sc1 : sc2);
        if (gg_tJ(NT)) dst[(NT - 1) * ts] = (double)(s_cntn[i * K + j] - (s_c[i] == j ? 1 : 0));
      }
    };
    int redo_a = -1, redo_b = -1;     // blocks whose entries the last round touched (-1: convert everything)
    const int wv = tid >> 6, ln = tid & 63;
    for (;;) {
      const long long c0 = g.dbg ? wall_clock64() : 0;
      // (the evaluation's first batch of coefficients is on its way while the statistics are converted)
      double b0_mf[GG_MFMA_RING][2 * NT];
      if (M13 && PW > 1) {
#pragma unroll
        for (int u = 0; u < GG_MFMA_RING; ++u) gg_lp_mfma_b1<NT>(x, u, b0_mf[u]);
      }
      __syncthreads();
      if (tid == 0) *s_min = GG_NO_CHANGE;
      if (has) {
        if (redo_a < 0) {
          for (int idx = tid; idx < 2 * K * per; idx += 256) convert(idx);
        } else {
          for (int e = tid; e < 4 * per; e += 256) {
            const int i = e % per, js = (e / per) & 1, d = e / (2 * per);
            convert((d * K + (js ? redo_b : redo_a)) * per + i);
          }
        }
      }
      __syncthreads();
      const long long ca = g.dbg ? wall_clock64() : 0;
      long long cb = 0;
      if (WIDE) {
        // lane (i0, k) keeps the chains of nodes i0 PW + p: one fetch of a coefficient-table entry (L2: the tables are never
        // staged at this size) for all of them; statistics [dir][j][t][node]: the lane's nodes are one LDS read per (j, t)
        const int i0 = tid / Kp, k = tid % Kp;
        double acc_o[PW][NT], acc_i[PW][NT];
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
          for (int t = 0; t < NT; ++t) { acc_o[p][t] = 0.0; acc_i[p][t] = 0.0; }
        if (k < K) {
          constexpr int UJ = NT <= 3 ? 4 : 2;
          auto step = [&](int j, const double* to, const double* ti) {
            const double* pa = s_st + ((size_t)(0 * K + j) * NT) * per + i0 * PW;
            const double* pb = s_st + ((size_t)(1 * K + j) * NT) * per + i0 * PW;
            double av[NT][PW], bv[NT][PW];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int p = 0; p < PW; ++p) { av[t][p] = pa[(size_t)t * per + p]; bv[t][p] = pb[(size_t)t * per + p]; }
#pragma unroll
            for (int p = 0; p < PW; ++p)
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                acc_o[p][t] = gg_fma(to[t], av[t][p], acc_o[p][t]);
                acc_i[p][t] = gg_fma(ti[t], bv[t][p], acc_i[p][t]);
              }
          };
          int j = 0;
          for (; j + UJ <= K; j += UJ) {
            double to[UJ][NT], ti[UJ][NT];
#pragma unroll
            for (int u = 0; u < UJ; ++u)
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                to[u][t] = x.Tout[((size_t)(j + u) * NT + t) * Kp + k];
                ti[u][t] = x.Tin[((size_t)(j + u) * NT + t) * Kp + k];
              }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < UJ; ++u) step(j + u, to[u], ti[u]);
          }
          for (; j < K; ++j) {
            double to[NT], ti[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) { to[t] = x.Tout[((size_t)j * NT + t) * Kp + k]; ti[t] = x.Tin[((size_t)j * NT + t) * Kp + k]; }
            step(j, to, ti);
          }
        }
        if (g.dbg) cb = wall_clock64();
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          const int i = i0 * PW + p;
          const int n = n0 + i;
          const bool live = has && n < N && n >= pos;
          const int nn = (has && n < N) ? n : N - 1;
          const double lp = gg_node_lp_finish<NT, ADJ>(x, nn, k, acc_o[p], acc_i[p], &logm_k);
          const int lab = gg_group_categorical_wide(lp, k, K, Kp, u_bandp[p], s_cat);
          if (k == 0) {
            unsigned long long dv = GG_SLOT_NONE;
            if (live) {
              const int cn = s_c[i];
              const int h = s_H[n - band0];
              const int expct = h != 0xffff ? (h & 0xff) : cn;
              if (lab != expct) dv = ((unsigned long long)(unsigned)n << (2 * LB)) | ((unsigned long long)(unsigned)cn << LB) | (unsigned)lab;
            }
            s_dev[i] = dv;
          }
        }
      } else if (M13 && PW > 1) {
        // the group's 4 PW nodes at once on the matrix cores (gg_lp_mfma: the chains of gg_block_lp as f64 MFMA steps, bit for
        // bit); wave w then draws for node 4 p + w, p = 0 .. PW - 1
        const int w4 = tid >> 6, k = tid & 63;
        gg_lp_mfma<NT, ADJ, (PW > 1 ? 4 * PW : 8)>(x, s_st, s_acc, logm_mf, b0_mf, self_mf, (int)(zero8 - s_st));   // (PW == 1: this branch is not taken)
        __syncthreads();
        if (g.dbg) cb = wall_clock64();
        double lpp[PW];
        int labp[PW];
#pragma unroll
        for (int p = 0; p < PW; ++p) lpp[p] = s_acc[(4 * p + w4) * 64 + k];
        gg_group_categorical64<PW>(lpp, k, K, u_bandp, labp);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          const int i = 4 * p + w4;
          const int n = n0 + i;
          const bool live = has && n < N && n >= pos;
          const int lab = labp[p];
          if (k == 0) {
            unsigned long long dv = GG_SLOT_NONE;
            if (live) {
              const int cn = s_c[i];
              const int h = s_H[n - band0];
              const int expct = h != 0xffff ? (h & 0xff) : cn;
              if (lab != expct) dv = ((unsigned long long)(unsigned)n << (2 * LB)) | ((unsigned long long)(unsigned)cn << LB) | (unsigned)lab;
            }
            s_dev[i] = dv;
          }
        }
      } else {
        const int i = tid / Kp, k = tid % Kp;
        const int n = n0 + i;
        const bool live = has && n < N && n >= pos;
        const int nn = (has && n < N) ? n : N - 1;
        auto ld = [&](int j, double* so, double* si) {
          const double* pa = s_st + (size_t)((0 * K + j) * per + i) * NT;
          const double* pb = s_st + (size_t)((1 * K + j) * per + i) * NT;
#pragma unroll
          for (int t = 0; t < NT; ++t) { so[t] = pa[t]; si[t] = pb[t]; }
        };
        double lp;
        if (MODE == 0 && a.mf_small) {
          gg_lp_mfma_small<NT, ADJ>(x, s_st, s_T, s_T + ntab, s_acc, per, Kp, (int)(zero8 - s_st));
          __syncthreads();
          double ao[NT], ai[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            ao[t] = s_acc[((size_t)t * per + i) * Kp + k];
            ai[t] = s_acc[((size_t)(NT + t) * per + i) * Kp + k];
          }
          lp = gg_node_lp_finish<NT, ADJ>(x, nn, k, ao, ai, &logm_k, self0_p);
        } else if (stage_tables == 1) lp = gg_node_lp_core<NT, ADJ>(x, nn, k, ld, s_T, s_T + ntab, &logm_k, self0_p);
        else if (split) lp = gg_lp_split4<NT, ADJ>(x, s_st, s_acc, nn, stage_tables == 2 ? s_T : nullptr, RS);
        else if (stage_tables == 2) lp = gg_node_lp_core_t1<NT, ADJ>(x, nn, k, ld, s_T, RS, &logm_k, self0_p);
        else lp = gg_node_lp_core<NT, ADJ>(x, nn, k, ld, x.Tout, x.Tin, &logm_k, self0_p);
        if (g.dbg) cb = wall_clock64();
        const int lab = gg_group_categorical(lp, k, K, Kp, u_band);
        if (k == 0) {
          // what the hypotheses say about this node: it moves to h's new label, or it stays
          unsigned long long dv = GG_SLOT_NONE;
          if (live) {
            const int cn = s_c[i];
            const int h = s_H[n - band0];
            const int expct = h != 0xffff ? (h & 0xff) : cn;
            if (lab != expct) dv = ((unsigned long long)(unsigned)n << (2 * LB)) | ((unsigned long long)(unsigned)cn << LB) | (unsigned)lab;
          }
          s_dev[i] = dv;
        }
      }
      __syncthreads();
      if (tid < 64) {
        // this workgroup's first GG_SCAN_NW surprises, in position order (the later ones were evaluated without the earlier
        // ones of this round: hypotheses like everybody else's)
        if (tid < GG_SCAN_NW) s_mine[tid] = GG_SLOT_NONE;
        int have = 0;
        for (int base = 0; base < per; base += 64) {
          const unsigned long long dv = base + tid < per ? s_dev[base + tid] : GG_SLOT_NONE;
          const unsigned long long bal = __ballot(dv != GG_SLOT_NONE);
          const int rk = have + __popcll(bal & ((1ull << tid) - 1ull));
          if (dv != GG_SLOT_NONE && rk < GG_SCAN_NW) s_mine[rk] = dv;
          have += __popcll(bal);
        }
      }
      const long long c1 = g.dbg ? wall_clock64() : 0;
      unsigned long long best;
      unsigned long long sv[GG_SCAN_NW];
      if (!gg_grid_exchange_multi(a.bar, G, (unsigned)(it + 1), s_mine, s_min, g.rs, &best, a.spin_limit, sv)) {
        alive = false;
        break;
      }
      ++it; ++wi;
      const long long c2 = g.dbg ? wall_clock64() : 0;
      t_eval += c1 - c0; t_xchg += c2 - c1;
      t_e1 += ca - c0; t_e2 += cb - ca; t_e3 += c1 - cb;
      const bool quiet = best == GG_SLOT_NONE;   // no surprise anywhere: every remaining hypothesis is the chain
      // ---- the surprises, in position order (= slot order), amend the hypotheses
      int nS = 0;
      if (!quiet) {
        int cnt_s = 0;
#pragma unroll
        for (int w = 0; w < GG_SCAN_NW; ++w) cnt_s += sv[w] != GG_SLOT_NONE ? 1 : 0;
        int incl = cnt_s;
        for (int dlt = 1; dlt < 64; dlt <<= 1) { const int o = __shfl_up(incl, dlt); if (ln >= dlt) incl += o; }
        if (ln == 63) s_misc[wv] = incl;
        __syncthreads();
        int offS = incl - cnt_s;
        for (int w = 0; w < wv; ++w) offS += s_misc[w];
        nS = s_misc[0] + s_misc[1] + s_misc[2] + s_misc[3];
        if (nS > 256) nS = 256;     // (what does not fit is reported again next round: it is not in H)
#pragma unroll
        for (int w = 0; w < GG_SCAN_NW; ++w) {
          const unsigned long long v = sv[w];
          if (v == GG_SLOT_NONE || offS >= 256) continue;
          const int f = (int)(v >> (2 * LB)), old = (int)((v >> LB) & LMASK), nw = (int)(v & LMASK);
          const int h = s_H[f - band0];
          // statistics behind f hold f in block `from` (the hypothesised label, or the old one): it goes to `nw`
          s_S[offS * 3] = f; s_S[offS * 3 + 1] = h != 0xffff ? (h & 0xff) : old; s_S[offS * 3 + 2] = nw;
          s_H[f - band0] = (nw == old) ? (unsigned short)0xffffu : (unsigned short)((old << 8) | nw);
          if (nw == old) atomicAnd(&s_Hm[(f - band0) >> 5], ~(1u << ((f - band0) & 31)));
          else atomicOr(&s_Hm[(f - band0) >> 5], 1u << ((f - band0) & 31));
          ++offS;
        }
        __syncthreads();
      }
      const long long cf = g.dbg ? wall_clock64() : 0;
      // ---- accept: the hypotheses at positions [pos, hi) - up to and including the earliest surprise - are the chain
      const int hi = quiet ? band1 : (int)(best >> (2 * LB)) + 1;
      bool cut = false;       // the list filled up before the end of the range
      {
        // (mask words of 32 nodes, contiguous runs per thread so that thread order is position order; one word per thread
        // unless K <= 4, whose bands have up to 1024 words)
        const int p0 = pos - band0, p1 = hi - band0;          // the range, relative to the band
        const int w_first = p0 >> 5, w_last = p1 > p0 ? ((p1 - 1) >> 5) : w_first - 1;
        const int wpt = (w_last - w_first + 1 + 255) / 256;   // words per thread
        auto word_mask = [&](int wd) {
          unsigned mk = s_Hm[wd];
          if (wd == w_first) mk &= ~0u << (p0 & 31);
          if (wd == w_last && (p1 & 31)) mk &= ~0u >> (32 - (p1 & 31));
          return mk;
        };
        int cnt_t = 0;
        for (int r = 0; r < wpt; ++r) { const int wd = w_first + tid * wpt + r; if (wd <= w_last) cnt_t += __popc(word_mask(wd)); }
        int incl = cnt_t;
        for (int dlt = 1; dlt < 64; dlt <<= 1) { const int o = __shfl_up(incl, dlt); if (ln >= dlt) incl += o; }
        if (ln == 63) s_misc[4 + wv] = incl;
        __syncthreads();
        int e = incl - cnt_t;
        for (int w = 0; w < wv; ++w) e += s_misc[4 + w];
        const int total = s_misc[4] + s_misc[5] + s_misc[6] + s_misc[7];
        const int room = a.Lcap - nL;
        const int take = total < room ? total : room;
        for (int r = 0; r < wpt; ++r) {
          const int wd = w_first + tid * wpt + r;
          if (wd > w_last) break;
          unsigned mk = word_mask(wd);
          while (mk) {
            const int bit = __ffs(mk) - 1;
            mk &= mk - 1u;
            const int n = band0 + (wd << 5) + bit;
            const int h = s_H[n - band0];
            if (e < take) {
              const int old = h >> 8, nw = h & 0xff;
              int* dstL = s_L + (size_t)(nL + e) * 3;
              dstL[0] = n; dstL[1] = old; dstL[2] = nw;
              atomicSub(&s_cnt[old], 1); atomicAdd(&s_cnt[nw], 1);
              if (b == 0) {   // for what runs after this kernel (labels; the list, for a scan that is handed to restart rounds)
                g.c[n] = nw;
                int32_t* dstG = g.L + (size_t)(nL + e) * 3;
                dstG[0] = n; dstG[1] = old; dstG[2] = nw;
              }
              if (e == take - 1) s_misc[8] = n;
            }
            ++e;
          }
        }
        __syncthreads();
        cut = take < total;
        if (take > 0) pos = cut ? s_misc[8] + 1 : hi;
        else if (!cut) pos = hi;
        nL += take; nchg += take;
      }
      const long long cg = g.dbg ? wall_clock64() : 0;
      if (g.dbg) { t_a1 += cf - c2; t_a2 += cg - cf; }
      // ---- statistics of my group's nodes behind each surprise: the edges to / from f move between blocks
      if (has && nS > 0) apply_moves(s_S, nS, n0, true);
      if (nS == 1) { redo_a = s_S[1]; redo_b = s_S[2]; } else redo_a = redo_b = -1;
      if (g.dbg) { __syncthreads(); t_apply += wall_clock64() - c2; }
      if (nL >= a.Lcap) {
        // the list is full: bring the statistics of every node that is not in LDS right now up to date in G
        // (earlier bands: as the bulk pass left them + ap entries; later bands: as the bulk pass left them), then
        // start a fresh list.  Grid-wide, followed by a barrier.
        {
          // (earlier bands' ap entries were plain stores of other workgroups: fenced meeting first)
          unsigned long long dummy;
          if (!gg_grid_exchange(a.bar, G, (unsigned)(it + 1), GG_SLOT_NONE, true, s_min, g.rs, &dummy, a.spin_limit)) { alive = false; break; }
          ++it;
        }
        for (int n = b * 256 + tid; n < N; n += G * 256) {
          if (RIDE) {
            // the riders count the next sweep's statistics under the labels this scan started with: the list that is about
            // to be dropped goes into that buffer too (atomics: the riders are adding to it at this moment)
            for (int e = 0; e < nL; ++e) {
              const int n2 = s_L[e * 3], ba = s_L[e * 3 + 1], bb = s_L[e * 3 + 2];
              const size_t at = (size_t)n2 * Np + n;
              for (int d = 0; d < 2; ++d) {
                if (ADJ) {
                  if (gg_abit(d == 0 ? g.XAT : g.XA, at)) { atomicSub(&a.Gn0[d * dstride + (size_t)ba * Np + n], 1); atomicAdd(&a.Gn0[d * dstride + (size_t)bb * Np + n], 1); }
                } else {
                  const int32_t* q = (d == 0 ? g.XWT : g.XW) + at * B;
                  if (q[0] == GG_NO_EDGE) continue;
                  atomicSub(&a.Gn0[d * dstride + (size_t)ba * Np + n], 1);
                  atomicAdd(&a.Gn0[d * dstride + (size_t)bb * Np + n], 1);
                  long long v[NSI > 0 ? NSI : 1];
                  int t = 0;
                  for (int r = 0; r < B; ++r) v[t++] = q[r];
                  for (int r = 0; r < B; ++r)
                    for (int s2 = r; s2 < B; ++s2) v[t++] = (long long)q[r] * (long long)q[s2];
                  for (int s2 = 0; s2 < NSI; ++s2) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(&a.Gsn0[(d * dstride + (size_t)ba * Np + n) * NSI + s2]), (unsigned long long)(-v[s2]));
                    atomicAdd(reinterpret_cast<unsigned long long*>(&a.Gsn0[(d * dstride + (size_t)bb * Np + n) * NSI + s2]), (unsigned long long)v[s2]);
                  }
                }
              }
            }
          }
          const int ng = n / per;
          if (ng >= band * G && ng < (band + 1) * G) { a.ap[n] = 0; continue; }   // in some workgroup's LDS
          for (int e = a.ap[n]; e < nL; ++e) {
            const int n2 = s_L[e * 3], ba = s_L[e * 3 + 1], bb = s_L[e * 3 + 2];
            const size_t at = (size_t)n2 * Np + n;
            for (int d = 0; d < 2; ++d) {
              if (ADJ) {
                if (gg_abit(d == 0 ? g.XAT : g.XA, at)) { a.Gc0[d * dstride + (size_t)ba * Np + n] -= 1; a.Gc0[d * dstride + (size_t)bb * Np + n] += 1; }
              } else {
                const int32_t* q = (d == 0 ? g.XWT : g.XW) + at * B;
                if (q[0] == GG_NO_EDGE) continue;
                a.Gc0[d * dstride + (size_t)ba * Np + n] -= 1;
                a.Gc0[d * dstride + (size_t)bb * Np + n] += 1;
                long long v[NSI > 0 ? NSI : 1];
                int t = 0;
                for (int r = 0; r < B; ++r) v[t++] = q[r];
                for (int r = 0; r < B; ++r)
                  for (int s2 = r; s2 < B; ++s2) v[t++] = (long long)q[r] * (long long)q[s2];
                for (int s2 = 0; s2 < NSI; ++s2) {
                  a.Gs0[(d * dstride + (size_t)ba * Np + n) * NSI + s2] -= v[s2];
                  a.Gs0[(d * dstride + (size_t)bb * Np + n) * NSI + s2] += v[s2];
                }
              }
            }
          }
          a.ap[n] = 0;
        }
        {
          unsigned long long dummy;
          if (!gg_grid_exchange(a.bar, G, (unsigned)(it + 1), GG_SLOT_NONE, true, s_min, g.rs, &dummy, a.spin_limit)) { alive = false; fatal = true; break; }
        }
        ++it;
        nL = 0;
        flushed = true;
      }
      if (quiet && !cut) break;   // the band is final
    }
    if (!alive) break;
    // the band is final: write the group's statistics back (exact under the labels so far) for the incremental path
    __syncthreads();
    if (has && g.keep_L) {
      for (int idx = tid; idx < 2 * K * per; idx += 256) {
        const int i = idx % per, j = (idx / per) % K, d = idx / (per * K);
        if (n0 + i >= N) continue;
        const size_t gi = d * dstride + (size_t)j * Np + n0 + i;
        a.Gc0[gi] = s_gc[idx];
        if (!ADJ)
          for (int s = 0; s < NSI; ++s) a.Gs0[gi * NSI + s] = s_gs[(size_t)idx * NSI + s];
      }
      if (tid < per && n0 + tid < N) a.ap[n0 + tid] = nL;
    }
  }
  // ---- tail: workgroup 0 publishes the result of the scan
  // An unchanged scan keeps its sort: every workgroup copies its share of perm (the tail's longest part, 16 loads deep
  // when workgroup 0 does it alone); workgroup 0 then only publishes.  (nchg is the same in every workgroup.)
  const bool unchanged = alive && g.tail_sort && g.rs[9] + nchg == 0;
  if (unchanged)
    for (int n = b * 256 + tid; n < N; n += G * 256) g.perm[n] = g.perm_prev[n];
  if (b != 0) return;
  if (!alive) {
    // An exchange timed out (the grid could not assemble: the device is shared with other processes, say).  What
    // workgroup 0 has recorded - labels, change list, position - is a valid prefix of the scan whoever gave up first:
    // a change is only recorded after every workgroup has published its candidate for it.  Leave the scan as a STALLED
    // one at that position (the state the restart rounds resume from: G as the bulk pass left it or as written back,
    // ap entries, L) and let the host carry on with the launch-per-change form.  Status bit 2 tells it why.
    // After a flush the statistics of the band that was in LDS at the time are not in G any more (the flush skips them and
    // restarts the list), and the workgroups that gave up may stand one accepted change apart: there is no consistent
    // stalled scan to hand to the restart rounds.  Reported as an error instead of resuming from stale statistics
    // (needs > 1024 changes in one scan AND a grid that cannot assemble).
    // Round 4: that case is recoverable after all - what workgroup 0 has recorded (labels, block sizes, position) is a valid
    // prefix of the scan there too; only the statistics are not.  Status bit 4 beside the stalled state with an EMPTY list tells
    // the host to count the statistics again under the labels as they stand (one bulk pass) before the restart rounds go on
    // from the position (gg_model.hip, model_sync: "recount").
    const bool recount = fatal || flushed;
    if (recount && tid == 0) atomicOr(&g.rs[3], 4);
    if (recount) nL = 0;
    __syncthreads();
    for (int k = tid; k < K; k += 256) g.cnt[k] = s_cnt[k];
    if (tid == 0) {
      // (the last node itself changed and the confirming exchange gave up: stand on that node again - its conditional
      // does not depend on its own label, so the restart round redraws the same label, finds the window final and runs
      // the tail; a stalled scan at pos == N would give the rounds nothing to do)
      const int rpos = pos < N ? pos : N - 1;
      g.rs[0] = rpos;
      g.rs[1] = nL;
      g.rs[2] += nchg;
      g.rs[9] += nchg;
      g.rs[4] = 1;
      g.rs[GG_RS_CARRY] = -1;
      // round state for the restart rounds, exactly as k_resolve leaves it for an incomplete scan
      *reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_STATE]) = gg_pack_state(rpos, nL, 0, 0, 0);
      *reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_MIN]) = GG_NO_CHANGE;
      g.rs[10] = 0;
    }
    return;
  }
  __syncthreads();
  for (int k = tid; k < K; k += 256) g.cnt[k] = s_cnt[k];
  __threadfence();
  __syncthreads();
  const long long t_tail0 = g.dbg ? wall_clock64() : 0;
  if (unchanged) {
    for (int k = tid; k <= K; k += 256) g.lab_start[k] = g.lab_start_prev[k];
    if (g.tail_m) { gg_resolve_args gm = g; gm.tail_sort = 0; gg_scan_tail(gm, s_cnt, s_raw, true); }
  } else {
    gg_scan_tail(g, s_cnt, s_raw, g.rs[9] + nchg == 0);
  }
  if (g.dbg) __syncthreads();
  if (g.dbg && tid == 0) {
    g.dbg[0] = t_eval; g.dbg[1] = t_xchg; g.dbg[2] = t_apply; g.dbg[3] = (long long)it; g.dbg[4] = t_load;
    g.dbg[5] = wall_clock64() - t_begin;
    g.dbg[6] = wall_clock64() - t_tail0;
    g.dbg[7] = t_init0 - t_kernel0;
    g.dbg[8] = t_e1; g.dbg[9] = t_e2; g.dbg[10] = t_e3; g.dbg[11] = t_a1; g.dbg[12] = t_a2;
  }
  if (tid == 0) {
    g.rs[0] = 0;
    g.rs[1] = g.keep_L ? nL : 0;
    g.rs[GG_RS_ROUNDS] += (int)it;     // exchange rounds of this scan (gg_stats.resolve_rounds)
    g.rs[GG_RS_CARRY] = nL;            // g.L[0, nL): the changes since the scan's last flush (what k_apply_L mode 3 applies)
    g.rs[2] += nchg;
    g.rs[9] += nchg;
    g.rs[5] += 1;
    g.rs[7] = 1;
    *reinterpret_cast<unsigned long long*>(&g.rs[GG_RS_STATE]) = gg_pack_state(0, g.keep_L ? nL : 0, 0, 1, 0);
  }
  if (a.epi_out) {
    // the call's last scan: the control block as it stands now goes to pinned host memory, the launch's tag last - the host
    // watches for it instead of copying the block and synchronising the stream (gg_model.hip, fetch_ctl)
    __syncthreads();
    if (tid < 24) reinterpret_cast<int32_t*>(a.epi_out + 4)[tid] = __hip_atomic_load(&g.rs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    __threadfence_system();
    if (tid == 0) a.epi_out[2] = (double)a.epi_tag;
  }
}
