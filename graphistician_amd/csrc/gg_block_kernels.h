// gg_block_kernels.h — HIP kernels of the block-model half-sweep (SBM adjacency,
// SBM Gaussian weights and their K = 1 special cases Beta-Bernoulli / NIW).
//
// One half-sweep (adjacency.py:615-618 or weights.py:394-397) is
//   sort labels -> BULK statistics -> pair tables -> draw parameters -> tables
//   -> speculate -> resolve [-> apply_L -> speculate -> resolve]* -> draw m
// BULK is the only kernel that touches the N x N data and is HBM-bound; every
// other kernel works on O(N K) statistics.
//
// Statistics G[dir][k][n] (dir 0 = out-edges of n, 1 = in-edges of n) are block
// histograms of node n's row/column: exact integers, so any summation order
// gives the same bits and the sequential scan of the reference can be resolved
// speculatively (kernel k_resolve) without changing a single draw.
#pragma once
#include <type_traits>
#include "gg_common.h"
#include "spec/gg_shcov.h"

// ------------------------------------------------------------------ statistics per (node, block) as a template constant
// Every scan kernel is instantiated on NT, the number of spec-order doubles per (direction, block) of a node's conditional
// (gg_lp.h): 2 = adjacency {ones, zeros}; 3 / 6 / 10 / 15 = Gaussian weights of dimension B = 1..4 {n, s1_r, s2_rs};
// 4 / 7 / 11 / 16 = the JOINT model (one label vector for adjacency and weights, stochastic_block_model.py:252-302):
// the Gaussian statistics followed by the number of potential partners in the block, whose coefficient is log(1 - p)
// (gg_joint_table).  Integer statistics kept per node: NSI = B + B (B + 1) / 2 in both weight forms, none for the adjacency.
__host__ __device__ constexpr bool gg_tJ(int NT) { return NT == 4 || NT == 7 || NT == 11 || NT == 16; }
__host__ __device__ constexpr int gg_tB(int NT) {
  return (NT == 2 || NT == 3 || NT == 4) ? 1 : ((NT == 6 || NT == 7) ? 2 : ((NT == 10 || NT == 11) ? 3 : 4));
}
__host__ __device__ constexpr int gg_tNSI(int NT) { return NT == 2 ? 1 : NT - 1 - (gg_tJ(NT) ? 1 : 0); }

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ long long gg_shfl_down_ll(long long v, int delta) {
  int lo = __shfl_down((int)(v & 0xffffffffll), delta);
  int hi = __shfl_down((int)(v >> 32), delta);
  return ((long long)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ long long gg_wave_sum_ll(long long v) {
  for (int d = 32; d >= 1; d >>= 1) v += gg_shfl_down_ll(v, d);
  return v;
}

// ------------------------------------------------------------------ label sort
// Counting sort of the labels: perm groups node indices by block, lab_start[k]
// is the first position of block k, cnt[k] its size.  One workgroup.
#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ __launch_bounds__(1024) void k_sort_labels(const int32_t* __restrict__ c, int N, int K,
                                                      int32_t* __restrict__ perm, int32_t* __restrict__ lab_start,
                                                      int32_t* __restrict__ cnt) {
  extern __shared__ int s_hist[];  // [K] counts, [K] cursors
  int* s_cur = s_hist + K;
  for (int k = threadIdx.x; k < 2 * K; k += blockDim.x) s_hist[k] = 0;
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) atomicAdd(&s_hist[c[n]], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int k = 0; k < K; ++k) {
      lab_start[k] = acc;
      cnt[k] = s_hist[k];
      s_cur[k] = acc;
      acc += s_hist[k];
    }
    lab_start[K] = acc;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    int pos = atomicAdd(&s_cur[c[n]], 1);
    perm[pos] = n;
  }
}
#endif

// ------------------------------------------------------------------ BULK kernels
// G[k][n] = sum over rows i with label k of X[i][n]  (column sums grouped by the
// row's label).  Run on X it yields the in-edge histograms of every node, run on
// the stored transpose the out-edge histograms.  Rows are visited grouped by
// label (perm), so a wave accumulates in registers only; a workgroup covers one
// (label, row sub-chunk, column strip) and adds its partial sums to G with
// integer atomics (order independent).  Row reads are whole 1 KiB wave
// transactions of a contiguous row segment: coalesced regardless of perm.
//
// slot -> (label k, sub-chunk): labels with more than RW rows span several slots.
// Every lane of the calling wavefront must be active (the slots of 64 labels at a time are counted by a wave-wide prefix sum:
// the serial walk this replaces was K dependent loads per work item - 6 us at 100 blocks, 13 us at 256).
// SERIAL: the walk - for the WEIGHT riders of the fused kernel's translation unit (gg_sweepq.hip, GG_FIND_SLOT_SERIAL_W):
// there the prefix sum moved the rider loop's schedule (32.2 -> 38.8 us per launch at 16 blocks), and that kernel only runs
// shapes with few blocks; its adjacency instantiation gains from the prefix sum (23.5 -> 20.1 us alone) and keeps it.
#ifndef GG_FIND_SLOT_SERIAL_W
#define GG_FIND_SLOT_SERIAL_W false
#endif
template <bool SERIAL = false>
__device__ __forceinline__ bool gg_find_slot(const int32_t* __restrict__ lab_start, int K, int RW, int slot,
                                             int* k_out, int* row0, int* nrows) {
  if (SERIAL) {
  for (int k = 0; k < K; ++k) {
    int s0 = lab_start[k], n = lab_start[k + 1] - s0;
    int ns = (n + RW - 1) / RW;
    if (slot < ns) {
      *k_out = k;
      *row0 = s0 + slot * RW;
      int rem = n - slot * RW;
      *nrows = rem < RW ? rem : RW;
      return true;
    }
    slot -= ns;
  }
  return false;
  }
  const int lane = (int)(threadIdx.x & 63u);
  int base = 0;                                   // slots of the labels before the current group of 64
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int k = k0 + lane;
    int s0 = 0, n = 0;
    if (k < K) { s0 = lab_start[k]; n = lab_start[k + 1] - s0; }
    const int ns = (n + RW - 1) / RW;
    int incl = ns;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    const int total = __shfl(incl, 63);
    if (slot < base + total) {
      const unsigned long long bal = __ballot(base + incl > slot);   // (the first such lane holds the slot's label)
      const int src = __ffsll((long long)bal) - 1;
      const int sub = slot - base - __shfl(incl - ns, src);
      const int s0s = __shfl(s0, src), ns_n = __shfl(n, src);
      *k_out = k0 + src;
      *row0 = s0s + sub * RW;
      const int rem = ns_n - sub * RW;
      *nrows = rem < RW ? rem : RW;
      return true;
    }
    base += total;
  }
  return false;
}

// Adjacency: 16 uint8 per lane per row, SWAR byte counters (<= 255 rows per wave).
// blockIdx.z selects the direction: 0 = out-statistics (reads X^T), 1 = in-statistics (reads X).
// Single entry of a bit-packed adjacency matrix: `at` is the linear index row * Np + col of the byte layout (Np is a
// multiple of 64, so rows are whole words).
__device__ __forceinline__ bool gg_abit(const uint32_t* __restrict__ X, size_t at) { return (X[at >> 5] >> (at & 31)) & 1u; }

// One (column strip, label slot, direction) work item of the adjacency pass; a workgroup of 256 threads.
// The matrices are bit-packed (gg_data::Ab, AbT): a strip is 1024 columns = 32 words per row, so half a wave reads one row
// (128 contiguous bytes) and the other half the next; every lane keeps BIT-SLICED counters for its 32 columns - plane b
// holds bit b of each column's count, a row's word is added with a ripple of and / xor - and expands them to integers at
// the end.  8 planes: at most 255 rows per lane (RW <= 1020 rows per slot, 8 rows in flight per workgroup).
template <bool NTL>
__device__ __forceinline__ void gg_bulk_adj_body(const uint32_t* __restrict__ XT, const uint32_t* __restrict__ X1,
                                                 int Np, const int32_t* __restrict__ perm,
                                                 const int32_t* __restrict__ lab_start, int K, int RW,
                                                 int32_t* __restrict__ G0, int32_t* __restrict__ G1, int dir, int strip,
                                                 int slot) {
  __shared__ unsigned int s_bins[1024];
  const uint32_t* __restrict__ X = dir == 0 ? XT : X1;
  int32_t* __restrict__ G = dir == 0 ? G0 : G1;
  int k, row0, nrows;
  if (!gg_find_slot(lab_start, K, RW, slot, &k, &row0, &nrows)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Wb = Np >> 5;                       // words per row
  const int wcol = strip * 32 + (lane & 31);    // this lane's word within the row
  const int rsub = wave * 2 + (lane >> 5);      // which of the 8 rows in flight is this lane's
  for (int i = tid; i < 1024; i += 256) s_bins[i] = 0;
  __syncthreads();
  unsigned pl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto add = [&](unsigned w) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned t = pl[b] & w;
      pl[b] ^= w;
      w = t;
    }
  };
  if (wcol < Wb) {
    int r = rsub;
    for (; r + 24 < nrows; r += 32) {      // 4 rows in flight per lane
      const int i0 = perm[row0 + r], i1 = perm[row0 + r + 8], i2 = perm[row0 + r + 16], i3 = perm[row0 + r + 24];
      const unsigned w0 = X[(size_t)i0 * Wb + wcol], w1 = X[(size_t)i1 * Wb + wcol], w2 = X[(size_t)i2 * Wb + wcol],
                     w3 = X[(size_t)i3 * Wb + wcol];
      add(w0); add(w1); add(w2); add(w3);
    }
    for (; r < nrows; r += 8) add(X[(size_t)perm[row0 + r] * Wb + wcol]);
#pragma unroll 4
    for (int c = 0; c < 32; ++c) {
      unsigned v = 0;
#pragma unroll
      for (int b = 0; b < 8; ++b) v |= ((pl[b] >> c) & 1u) << b;
      if (v) atomicAdd(&s_bins[(lane & 31) * 32 + c], v);
    }
  }
  __syncthreads();
  for (int i = tid; i < 1024; i += 256) {
    unsigned int v = s_bins[i];
    int col = strip * 1024 + i;
    if (v && col < Np) atomicAdd(&G[(size_t)k * Np + col], (int)v);
  }
}

template <bool NTL>
__global__ __launch_bounds__(256) void k_bulk_adj(const uint32_t* __restrict__ XT, const uint32_t* __restrict__ X1,
                                                  int Np, const int32_t* __restrict__ perm,
                                                  const int32_t* __restrict__ lab_start, int K, int RW,
                                                  int32_t* __restrict__ G0, int32_t* __restrict__ G1, int dir0,
                                                  const int32_t* __restrict__ ctl) {
  if (ctl && ctl[4]) return;  // pipeline stalled by an earlier scan: leave everything untouched
  gg_bulk_adj_body<NTL>(XT, X1, Np, perm, lab_start, K, RW, G0, G1, dir0 + (int)blockIdx.z, (int)blockIdx.x, (int)blockIdx.y);
}

// streaming reads of the bulk passes: -DGG_NT_LOADS marks them non-temporal (each byte is used once per pass)
// Streaming reads of the bulk passes.  NT = non-temporal: every byte is used once per pass, so when a pass streams
// more than the 256 MiB Infinity Cache can hold the loads should not allocate on the way (measured on MI355X:
// k_bulk_w 6.2 instead of 5.55 TB/s at N=65536; at N=4096, where the matrices stay cache-resident from sweep to
// sweep, the same hint costs 9 %, so the host picks per size).
typedef int gg_int4v __attribute__((ext_vector_type(4)));
typedef unsigned int gg_uint4v __attribute__((ext_vector_type(4)));
template <bool NTL>
__device__ __forceinline__ int4 gg_stream_load(const int4* p) {
  if (NTL) {
    const gg_int4v v = __builtin_nontemporal_load(reinterpret_cast<const gg_int4v*>(p));
    return make_int4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
template <bool NTL>
__device__ __forceinline__ uint4 gg_stream_load(const uint4* p) {
  if (NTL) {
    const gg_uint4v v = __builtin_nontemporal_load(reinterpret_cast<const gg_uint4v*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *p;
}

// Weights: 4 int32 per lane per row = 4 / B elements; per element edge count,
// sum q_r, sum q_r q_s in 64-bit integers.
// One (column strip, label slot, direction) work item of the weight pass; a workgroup of 256 threads.
// Row element of the weight pass as an int4 (component 3 unused for B = 3, whose elements are 12 bytes: three dwords).
struct __attribute__((packed, aligned(4))) gg_i3 { int x, y, z; };
template <int B, bool NTL>
__device__ __forceinline__ int4 gg_load_welem(const int32_t* __restrict__ p) {
  if (B == 3) {
    const gg_i3 v = *reinterpret_cast<const gg_i3*>(p);
    return make_int4(v.x, v.y, v.z, 0);
  }
  return gg_stream_load<NTL>(reinterpret_cast<const int4*>(p));
}

template <int B, int U, bool NTL>
__device__ __forceinline__ void gg_bulk_w_body(const int32_t* __restrict__ XT, const int32_t* __restrict__ X1,
                                               int Np, const int32_t* __restrict__ perm,
                                               const int32_t* __restrict__ lab_start, int K, int RW,
                                               int32_t* __restrict__ Gc0, int32_t* __restrict__ Gc1,
                                               long long* __restrict__ Gs0, long long* __restrict__ Gs1, int dir, int strip,
                                               int slot) {
  constexpr int EPL = 4 / B;               // elements per lane
  constexpr int NSI = B + B * (B + 1) / 2; // integer statistics per element
  constexpr int SC = 64 * EPL;             // strip columns
  __shared__ int s_cnt[SC];
  __shared__ unsigned long long s_s[SC * NSI];
  const int32_t* __restrict__ X = dir == 0 ? XT : X1;
  int32_t* __restrict__ Gcnt = dir == 0 ? Gc0 : Gc1;
  long long* __restrict__ GS = dir == 0 ? Gs0 : Gs1;
  int k, row0, nrows;
  if (!gg_find_slot<GG_FIND_SLOT_SERIAL_W>(lab_start, K, RW, slot, &k, &row0, &nrows)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col0 = strip * SC + lane * EPL;
  for (int i = tid; i < SC; i += 256) s_cnt[i] = 0;
  for (int i = tid; i < SC * NSI; i += 256) s_s[i] = 0ull;
  __syncthreads();
  if (col0 < Np) {
    // Per-wave partial sums are kept in the cheapest exact type: a wave sees at most RW/4 <= 255
    // rows and |q| <= 2^23, so sum q fits int32 for <= 127 rows and sum q_r q_s (< 2^46 each) is
    // exact in fp64 (< 2^53) for <= 127 rows; longer runs flush into the 64-bit accumulators.
    int cnt[EPL];
    int s1[EPL][B];
    double s2[EPL][NSI - B];
    long long st[EPL][NSI];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      cnt[e] = 0;
#pragma unroll
      for (int r = 0; r < B; ++r) s1[e][r] = 0;
#pragma unroll
      for (int t = 0; t < NSI - B; ++t) s2[e][t] = 0.0;
#pragma unroll
      for (int s = 0; s < NSI; ++s) st[e][s] = 0;
    }
    auto accumulate = [&](const int4& v) {
      const int q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const bool valid = q[e * B] != GG_NO_EDGE;
        cnt[e] += valid ? 1 : 0;
        int x[B];
        double xd[B];
#pragma unroll
        for (int r = 0; r < B; ++r) { x[r] = valid ? q[e * B + r] : 0; xd[r] = (double)x[r]; }
#pragma unroll
        for (int r = 0; r < B; ++r) s1[e][r] += x[r];
        int t = 0;
#pragma unroll
        for (int r = 0; r < B; ++r)
#pragma unroll
          for (int s = r; s < B; ++s) { s2[e][t] = gg_fma(xd[r], xd[s], s2[e][t]); ++t; }
      }
    };
    auto flush = [&]() {
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
#pragma unroll
        for (int r = 0; r < B; ++r) { st[e][r] += (long long)s1[e][r]; s1[e][r] = 0; }
#pragma unroll
        for (int t = 0; t < NSI - B; ++t) { st[e][B + t] += (long long)s2[e][t]; s2[e][t] = 0.0; }
      }
    };
    int r = wave;
    int since = 0;
    for (; r + 4 * (U - 1) < nrows; r += 4 * U) {   // U rows in flight per wave
      int4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int iu = perm[row0 + r + 4 * u];
        v[u] = gg_load_welem<B, NTL>(X + ((size_t)iu * Np + col0) * B);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) accumulate(v[u]);
      since += U;
      if (since > 120 - U) { flush(); since = 0; }
    }
    for (; r < nrows; r += 4) {
      int i0 = perm[row0 + r];
      int4 v0 = gg_load_welem<B, NTL>(X + ((size_t)i0 * Np + col0) * B);
      accumulate(v0);
      since += 1;
      if (since > 120) { flush(); since = 0; }
    }
    flush();
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      if (cnt[e]) {
        atomicAdd(&s_cnt[lane * EPL + e], cnt[e]);
#pragma unroll
        for (int s = 0; s < NSI; ++s) atomicAdd(&s_s[(lane * EPL + e) * NSI + s], (unsigned long long)st[e][s]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < SC; i += 256) {
    int col = strip * SC + i;
    int v = s_cnt[i];
    if (v && col < Np) {
      atomicAdd(&Gcnt[(size_t)k * Np + col], v);
      for (int s = 0; s < NSI; ++s)
        atomicAdd(reinterpret_cast<unsigned long long*>(&GS[((size_t)k * Np + col) * NSI + s]), s_s[i * NSI + s]);
    }
  }
}

#ifndef GG_BULK_W_WAVES
#define GG_BULK_W_WAVES 1
#endif
template <int B, int U, bool NTL>
__global__ __launch_bounds__(256, GG_BULK_W_WAVES) void k_bulk_w(const int32_t* __restrict__ XT, const int32_t* __restrict__ X1,
                                                int Np, const int32_t* __restrict__ perm,
                                                const int32_t* __restrict__ lab_start, int K, int RW,
                                                int32_t* __restrict__ Gc0, int32_t* __restrict__ Gc1,
                                                long long* __restrict__ Gs0, long long* __restrict__ Gs1, int dir0,
                                                const int32_t* __restrict__ ctl) {
  if (ctl && ctl[4]) return;
  gg_bulk_w_body<B, U, NTL>(XT, X1, Np, perm, lab_start, K, RW, Gc0, Gc1, Gs0, Gs1, dir0 + (int)blockIdx.z, (int)blockIdx.x,
                            (int)blockIdx.y);
}

// ------------------------------------------------------------------ riders
// The statistics of sweep s+1 are those the bulk pass would count under the labels sweep s STARTS with whenever the scan of
// sweep s changes no label - the steady state of a converged chain.  While a chain is quiet the engine therefore does not
// launch the bulk pass of sweep s+1 on its own: its work items ride as extra workgroups ("riders", block indices behind
// the front workgroups) in the two launches of sweep s's latency chain - k_params and k_spec_resolve - and stream X and
// X^T while those few hundred workgroups wait on dependent loads.  Every sweep still streams all of the data once; a scan
// that does change a label stalls the pipeline as before, and the riders' statistics are discarded.
struct gg_ride_args {
  const void* XT;            // adjacency: bit-packed A^T / A (uint32 words); weights: int32 Wq^T / Wq
  const void* X1;
  int Np, K, RW;
  const int32_t* perm;       // label sort the sweep starts with
  const int32_t* lab_start;
  int32_t *Gc0, *Gc1;        // statistics buffer of the NEXT sweep (zeroed by the previous sweep's k_params)
  long long *Gs0, *Gs1;
  int strips, nslots;        // work item w = (dir * nslots + slot) * strips + strip
  int w0, w1;                // this launch carries items [w0, w1)
  int front;                 // block index of the first rider
  int nriders;               // rider workgroups in the launch (persistent: item w, w + nriders, ...)
  int delay;                 // riders sleep this many x 127 x 64 clocks before their first load (GG_RIDER_DELAY; tuning)
  long long* dbg;            // GG_SWQ_DBG: [0] first workgroup start, [1] last scan end, [2] last parameter end, [3] last rider end,
                             //             [4] first rider start, [5] verdict seen by the parameter workgroups (max)
};

// Riders are persistent: `nriders` workgroups walk the work items with that stride.  Their number bounds the bytes the pass
// keeps in flight (workgroups x 4 waves x 4 KB), and with it the loaded latency every dependent load of the front
// workgroups sees: enough in flight to saturate HBM, not four times that (measured: DESIGN.md section 5).
#ifndef GG_RIDE_U
#define GG_RIDE_U 8   // rows a rider wave keeps in flight (measured at N=4096: 8 rows x 256-row items 31.8 us per sweep of the weight model alone, 4 x 192: 35.0)
#endif
template <bool ADJ, int B, bool NTL>
__device__ __forceinline__ void gg_ride(const gg_ride_args& r) {
  for (int i = 0; i < r.delay; ++i) __builtin_amdgcn_s_sleep(127);
  for (int w = r.w0 + ((int)blockIdx.x - r.front); w < r.w1; w += r.nriders) {
    const int strip = w % r.strips, sl = (w / r.strips) % r.nslots, dir = w / (r.strips * r.nslots);
    if (ADJ)
      gg_bulk_adj_body<NTL>(reinterpret_cast<const uint32_t*>(r.XT), reinterpret_cast<const uint32_t*>(r.X1), r.Np, r.perm, r.lab_start,
                            r.K, r.RW, r.Gc0, r.Gc1, dir, strip, sl);
    else
      gg_bulk_w_body<B, GG_RIDE_U, NTL>(reinterpret_cast<const int32_t*>(r.XT), reinterpret_cast<const int32_t*>(r.X1), r.Np, r.perm,
                                r.lab_start, r.K, r.RW, r.Gc0, r.Gc1, r.Gs0, r.Gs1, dir, strip, sl);
    __syncthreads();   // the body's LDS bins are reused by the next item
  }
}

// ------------------------------------------------------------------ pair tables
// tab[a][b][.] = sum over nodes n of block a of the out-statistics G[0][b][n]:
// the K x K masked sums of adjacency.py:630-634 / weights.py:405-408 in one
// pass over O(N K) numbers.  Weight sums are kept as (hi, lo) 32-bit halves so
// that N^2 terms of 46 bits cannot overflow (gg_hilo_to_double rounds once).
// Layout: adjacency NTAB = 1 {ones}; weights NTAB = 1 + 2 NSI {cnt, hi_s, lo_s ...}.
// When the diagonal is not special-cased it belongs to pair (c_i, c_i).
#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ __launch_bounds__(256) void k_pair_tables(const int32_t* __restrict__ Gcnt0,
                                                     const long long* __restrict__ GS0, int Np, int K, int NSI,
                                                     const int32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ lab_start, int add_diag,
                                                     const uint8_t* __restrict__ dA,
                                                     const int32_t* __restrict__ dWq, int B,
                                                     long long* __restrict__ tab) {
  const int a = blockIdx.x / K, b = blockIdx.x % K;
  const int NTAB = GS0 ? 1 + 2 * NSI : 1;
  const int s0 = lab_start[a], s1 = lab_start[a + 1];
  long long acc[1 + 2 * (GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)];
  for (int t = 0; t < NTAB; ++t) acc[t] = 0;
  for (int r = s0 + threadIdx.x; r < s1; r += blockDim.x) {
    int n = perm[r];
    acc[0] += Gcnt0[(size_t)b * Np + n];
    if (GS0)
      for (int s = 0; s < NSI; ++s) {
        long long v = GS0[((size_t)b * Np + n) * NSI + s];
        acc[1 + 2 * s] += (v >> 32);
        acc[2 + 2 * s] += (v & 0xffffffffll);
      }
    if (add_diag && a == b && dA[n]) {
      acc[0] += 1;
      if (GS0) {
        const int32_t* q = dWq + (size_t)n * B;
        int t = 0;
        for (int r1 = 0; r1 < B; ++r1, ++t) {
          long long v = q[r1];
          acc[1 + 2 * t] += (v >> 32);
          acc[2 + 2 * t] += (v & 0xffffffffll);
        }
        for (int r1 = 0; r1 < B; ++r1)
          for (int r2 = r1; r2 < B; ++r2, ++t) {
            long long v = (long long)q[r1] * (long long)q[r2];
            acc[1 + 2 * t] += (v >> 32);
            acc[2 + 2 * t] += (v & 0xffffffffll);
          }
      }
    }
  }
  __shared__ long long s_red[4][1 + 2 * (GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = 0; t < NTAB; ++t) {
    long long v = gg_wave_sum_ll(acc[t]);
    if (lane == 0) s_red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < NTAB) {
    long long v = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_red[w][threadIdx.x];
    tab[(size_t)blockIdx.x * NTAB + threadIdx.x] = v;
  }
}
#endif

// ------------------------------------------------------------------ parameter draws
struct gg_adj_hyp {
  double tau1, tau0;
  int special;       // special_case_self_conns (BB: self_connection)
  int bb;            // Beta-Bernoulli semantics (adjacency.py:153-181)
  long long d1, d0;  // diagonal edge / non-edge counts
  int ngraphs;       // adjacency matrices the counts were summed over (resample_p(As), adjacency.py:630-634): 1 in a sweep
};

struct gg_w_hyp {
  gg_niw niw, niw_self;
  int special;
  int shift;
  long long d1;
  long long dS[GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2];
};

__device__ inline void gg_unpack_wstats(const long long* hl /* hi,lo pairs */, int B, int shift, double* s1, double* s2) {
  const double sc1 = gg_pow2i(-shift), sc2 = gg_pow2i(-2 * shift);
  int t = 0;
  for (int r = 0; r < B; ++r, ++t) s1[r] = gg_hilo_to_double(hl[2 * t], hl[2 * t + 1]) * sc1;
  for (int r = 0; r < B; ++r)
    for (int s = r; s < B; ++s, ++t) {
      double v = gg_hilo_to_double(hl[2 * t], hl[2 * t + 1]) * sc2;
      s2[r * B + s] = v;
      s2[s * B + r] = v;
    }
}

// adjacency.py:697-702 / weights.py:493-498 resample_m.
// ctl: the pipeline's control block, or null - behind a stalled scan the launch is a no-op like every other kernel of the
// pipeline (more than 64 blocks, speculative form: a draw from the half-scanned block sizes would hand the resumed scan the
// wrong log m).
#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ void k_draw_m(const int32_t* __restrict__ cnt, const double* __restrict__ pi, int K, uint64_t seed,
                         uint32_t site, uint32_t sweep, double* __restrict__ m, const int32_t* __restrict__ ctl = nullptr) {
  extern __shared__ double s_g[];
  if (ctl && ctl[4]) return;
  int k = threadIdx.x;
  if (k < K) s_g[k] = gg_dirichlet_gamma_at(seed, site, sweep, k, pi[k] + (double)cnt[k]);
  __syncthreads();
  if (k < K) {
    double tot = 0.0;
    for (int j = 0; j < K; ++j) tot += s_g[j];
    m[k] = s_g[k] / tot;
  }
}
#endif

// log m for more than 64 blocks (the parameter kernel's last workgroup has 192 lanes for it): all of them, after it.
#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ void k_logm(const double* __restrict__ m, double* __restrict__ logm, int K, const int32_t* __restrict__ ctl = nullptr) {
  if (ctl && ctl[4]) return;
  const int k = threadIdx.x;
  if (k < K) logm[k] = gg_log(m[k]);
}
#endif

// ------------------------------------------------------------------ fused parameter step
// One workgroup per block pair (a, b) (+ one for the self parameter and log m):
//   1. reduce the pair's sufficient statistics from the out-histograms (k_pair_tables),
//   2. thread 0 draws the parameter (k_draw_p / k_draw_niw),
//   3. and writes the pair's coefficient-table entries (k_prep_tables);
// every thread also zeroes a slice of the *other* statistics buffer, which the
// next half-sweep's bulk kernel accumulates into.
#define GG_SLOT_PAD 16   // 8-byte words between two exchange slots of k_scan: every slot on its own 128-byte line
struct gg_params_args {
  int is_adj, K, Kp, B, NSI, Np, N;
  int do_tables, do_draw, do_prep;
  int add_diag;
  const int32_t* Gcnt0;
  const long long* GS0;
  const int32_t* perm;
  const int32_t* lab_start;
  const int32_t* cnt;
  const uint8_t* dA;
  const int32_t* dWq;
  long long* tab;
  gg_adj_hyp ah;
  gg_w_hyp wh;
  uint64_t seed;
  uint32_t sweep;
  double *m, *logm, *p, *p_self, *mu, *sigma, *mu_self, *sigma_self, *T, *Tout, *Tin;
  int32_t* ctl;
  unsigned* verdict;              // verdict word of the fused launch (its own cache line)
  unsigned long long* zero_ptr;   // other statistics buffer (8-byte words) or null
  size_t zero_words;
  // deferred resample_m of the PREVIOUS sweep (adjacency.py:697-702): drawn here, off the scan's
  // critical path, before log m is formed
  int draw_m;
  uint32_t m_site, m_sweep;
  const double* pi;
  double beta;   // tempering of the parameter posterior: counts weighted by beta (graph_models.py:180-181,366-367)
  int sweep_from_ctl;   // graph replay: the sweep index lives in ctl[6]; the deferred m belongs to sweep - 1
  int n0, Nl;           // sharded chain: only nodes [n0, n0 + Nl) are stored locally (tables are then partial)
  int tables_from_tab;  // sharded chain: the all-reduced tables are already in `tab`
  unsigned long long* bar;   // k_scan's barrier counter + first-changer words, re-armed for the scan that follows
  int32_t* ap;               // k_scan's per-node "entries of the change list already applied" (or null), zeroed with them: a resident
  int ap_n;                  // grid whose workgroups are not all placed when it gives up (a device shared with another process)
                             // has late workgroups that leave at once - they used to leave their groups' entries as they were
  int shared;                // SBMGaussianWeightSharedCov: draw the means given the shared covariance, leave the
                             // covariance draw and the coefficient tables to k_shared_cov
  double* csc;               // [K*K][B*B + 1] centred scatter and tempered count of every block pair's off-diagonal edges (shared-covariance kind)
};

#ifndef GG_VERDICT_FIRST_SLEEP
#define GG_VERDICT_FIRST_SLEEP 2   // x 127 x 64 clocks before the first poll
#endif
// Verdict word of the fused quiet-chain launch (k_sweep_q, gg_resolve_kernels.h): the workgroup that resolves the scan of
// sweep s publishes (launch tag << 1 | scan changed a label); the parameter workgroups of sweep s+1, which ride in the
// same launch, wait for it before they write anything (an unchanged scan is what makes their input statistics valid).
// The word lives on a cache line of its own (gg_params_args::verdict / gg_resolve_args::verdict): agent-scope polls of one
// address are served by ONE memory channel, past the per-XCD L2s - a few hundred workgroups polling the line the scan's
// atomics live on, or polling at full rate, slow those atomics and every stream that crosses that channel (measured:
// scan 11 -> 35 us, riders 30 -> 59 us).  So: a first sleep about as long as the scan takes, then a poll every ~1 us.
// One thread polls (presleep: 127 x 64-clock sleeps before the first poll); true = the scan changed nothing.
__device__ __forceinline__ bool gg_poll_verdict(int32_t* ctl, unsigned* verdict, unsigned tag, int presleep) {
  for (int i = 0; i < presleep; ++i) __builtin_amdgcn_s_sleep(127);
  for (int spin = 0; spin < (1 << 19); ++spin) {
    // (relaxed polls: an acquire load would invalidate the compute unit's caches, shared with the scan's workgroups, at
    // every poll; nothing this workgroup reads afterwards was written by the scan)
    const unsigned v = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((v >> 1) == tag) {
      // one acquire fence, once per sweep, pairs with the resolver's release store: what this workgroup writes from here on
      // (control words the resolver has written too) is ordered behind the resolver's writes
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return (v & 1u) == 0u;
    }
    __builtin_amdgcn_s_sleep(40);
  }
  atomicOr(&ctl[3], 8);   // the resolving workgroups are dispatched ahead of this one: only a hung device gets here
  return false;
}
// The same for a whole workgroup (uniform result).
__device__ __forceinline__ bool gg_wait_verdict(int32_t* ctl, unsigned* verdict, unsigned tag) {
  __shared__ int s_verdict_ok;
  if (threadIdx.x == 0) s_verdict_ok = gg_poll_verdict(ctl, verdict, tag, GG_VERDICT_FIRST_SLEEP) ? 1 : 0;
  __syncthreads();
  return s_verdict_ok != 0;
}

// The parameter step of one workgroup (`bid` of `nfront`).  gate_tag != 0: nothing is written before the verdict of the
// scan that runs in the same launch has arrived, and nothing at all if that scan changed a label.
template <bool ADJ, int B, bool JOINT = false>
__device__ __forceinline__ void gg_params_body(gg_params_args& g, const int bid, const unsigned nfront, const unsigned gate_tag) {
  const int K = g.K;
  constexpr int NSI = ADJ ? 0 : B + B * (B + 1) / 2;
  constexpr int NTAB = ADJ ? 1 : 1 + 2 * NSI;
  constexpr int NTG = 1 + B + B * (B + 1) / 2;            // Gaussian coefficients
  constexpr int NT = ADJ ? 2 : NTG + (JOINT ? 1 : 0);     // + the joint model's partner coefficient
  auto housekeeping = [&]() {
    if (g.do_prep && bid == 0 && threadIdx.x == 0) {   // a new scan follows: reset its per-scan words
      g.ctl[7] = 0;
      g.ctl[8] = 0;
      g.ctl[9] = 0;
    }
    // k_scan's exchange slots (2 x 256, one per 128-byte line): tag 0 = never written
    if (g.do_prep && g.bar)
      for (unsigned sl = bid * blockDim.x + threadIdx.x; sl < 512u * 4u; sl += nfront * blockDim.x) g.bar[(size_t)(sl >> 2) * GG_SLOT_PAD + (sl & 3u)] = 0ull;   // (k_scan uses words 0..3 of a slot line)
    if (g.do_prep && g.ap)
      for (unsigned i = bid * blockDim.x + threadIdx.x; i < (unsigned)g.ap_n; i += nfront * blockDim.x) g.ap[i] = 0;
    if (g.zero_ptr) {
      size_t i = (size_t)bid * blockDim.x + threadIdx.x;
      size_t stride = (size_t)nfront * blockDim.x;
      for (; i < g.zero_words; i += stride) g.zero_ptr[i] = 0ull;
    }
  };
  if (!gate_tag) housekeeping();
  __shared__ long long s_red[4][NTAB];
  __shared__ long long s_tab[NTAB];
  __shared__ long long s_tabd[NTAB];   // diagonal edges of pair (a, a), shared-covariance kind without the special case
  if (threadIdx.x < NTAB) s_tabd[threadIdx.x] = 0;
  const int t = bid;
  if (t < K * K) {
    const int a = t / K, b = t % K;
    if (g.do_tables) {
      const int s0 = g.lab_start[a], s1 = g.lab_start[a + 1];
      long long acc[NTAB];
      // shared-covariance kind without the self-connection special case: the covariance is centred on the
      // off-diagonal edges only (weights.py:545-548), so the diagonal's share of pair (a, a) is also kept apart
      const bool keep_diag = !ADJ && g.shared && g.add_diag && a == b;
      long long accd[ADJ ? 1 : NTAB];
#pragma unroll
      for (int q = 0; q < NTAB; ++q) acc[q] = 0;
#pragma unroll
      for (int q = 0; q < (ADJ ? 1 : NTAB); ++q) accd[q] = 0;
      for (int r = s0 + threadIdx.x; r < s1; r += blockDim.x) {
        int n = g.perm[r] - g.n0;
        if (n < 0 || n >= g.Nl) continue;   // not an owned node
        acc[0] += g.Gcnt0[(size_t)b * g.Np + n];
        if (!ADJ)
#pragma unroll
          for (int s = 0; s < NSI; ++s) {
            long long v = g.GS0[((size_t)b * g.Np + n) * NSI + s];
            acc[1 + 2 * s] += (v >> 32);
            acc[2 + 2 * s] += (v & 0xffffffffll);
          }
        if (g.add_diag && a == b && g.dA[n]) {
          acc[0] += 1;
          if (!ADJ) {
            accd[0] += 1;
            const int32_t* q = g.dWq + (size_t)n * B;
            int tt = 0;
            for (int r1 = 0; r1 < B; ++r1, ++tt) {
              long long v = q[r1];
              acc[1 + 2 * tt] += (v >> 32);
              acc[2 + 2 * tt] += (v & 0xffffffffll);
              accd[1 + 2 * tt] += (v >> 32);
              accd[2 + 2 * tt] += (v & 0xffffffffll);
            }
            for (int r1 = 0; r1 < B; ++r1)
              for (int r2 = r1; r2 < B; ++r2, ++tt) {
                long long v = (long long)q[r1] * (long long)q[r2];
                acc[1 + 2 * tt] += (v >> 32);
                acc[2 + 2 * tt] += (v & 0xffffffffll);
                accd[1 + 2 * tt] += (v >> 32);
                accd[2 + 2 * tt] += (v & 0xffffffffll);
              }
          }
        }
      }
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int q = 0; q < NTAB; ++q) {
        long long v = gg_wave_sum_ll(acc[q]);
        if (lane == 0) s_red[wave][q] = v;
      }
      __syncthreads();
      if (threadIdx.x < NTAB) {
        long long v = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_red[w][threadIdx.x];
        s_tab[threadIdx.x] = v;
        g.tab[(size_t)t * NTAB + threadIdx.x] = v;
      }
      __syncthreads();
      if (!ADJ && keep_diag) {     // the diagonal's share, reduced the same way (workgroup-uniform branch)
#pragma unroll
        for (int q = 0; q < NTAB; ++q) {
          long long v = gg_wave_sum_ll(accd[q]);
          if (lane == 0) s_red[wave][q] = v;
        }
        __syncthreads();
        if (threadIdx.x < NTAB) {
          long long v = 0;
          for (int w = 0; w < (int)(blockDim.x >> 6); ++w) v += s_red[w][threadIdx.x];
          s_tabd[threadIdx.x] = v;
        }
        __syncthreads();
      }
    } else if (g.tables_from_tab) {
      if (threadIdx.x < NTAB) s_tab[threadIdx.x] = g.tab[(size_t)t * NTAB + threadIdx.x];
      __syncthreads();
    }
    // Gated form: the draw itself (one lane, a few microseconds of dependent arithmetic) does not wait for the verdict, only
    // its writes do - except for the shared-covariance kind, whose draw reads state the same step writes.
    const bool early_gate = gate_tag && !ADJ && g.shared;
    if (early_gate) {
      if (!gg_wait_verdict(g.ctl, g.verdict, gate_tag)) return;   // (workgroup-uniform)
      housekeeping();
    }
    __shared__ int s_commit;
    if (threadIdx.x == 0) {
      double e[GG_MAXNT];
      bool bad = false, commit = true;
      if (ADJ) {
        double pv;
        if (g.do_draw) {
          long long n1 = s_tab[0];
          long long pairs = (long long)g.cnt[a] * (long long)g.cnt[b];
          if (a == b && g.ah.special) pairs -= g.cnt[a];
          pairs *= (long long)g.ah.ngraphs;      // (every graph of the list offers every pair: adjacency.py:633)
          long long n0 = pairs - n1;
          gg_stream s;
          gg_stream_init(&s, g.seed, g.ah.bb ? GG_SITE_BB_P : GG_SITE_ADJ_P, g.sweep, (uint32_t)t);
          pv = gg_beta(&s, g.ah.tau1 + g.beta * (double)n1, g.ah.tau0 + g.beta * (double)n0);
        } else {
          pv = g.p[t];
        }
        e[0] = gg_log(pv);
        e[1] = gg_log(1.0 - pv);
        if (gate_tag) commit = gg_poll_verdict(g.ctl, g.verdict, gate_tag, 0);
        if (commit && g.do_draw) g.p[t] = pv;
      } else {
        // wide weights (spec/gg_lp.h): the data are the two limbs of a ONE-dimensional weight (template B = 2, g.B = 1) - the
        // statistics collapse to (n, sum w, sum w^2) for the draw, the parameters are scalars, the table spreads them again
        const bool wide1 = (B == 2) && g.B == 1;
        double mo[GG_MAXB], so[GG_MAXB * GG_MAXB];
        if (g.do_draw && g.shared) {
          // weights.py:529-535: mean of the pair given the shared covariance; its centred scatter for :545-548
          double s1[GG_MAXB], s2[GG_MAXB * GG_MAXB], C[GG_MAXB * GG_MAXB];
          gg_unpack_wstats(s_tab + 1, B, g.wh.shift, s1, s2);
          for (int i = 0; i < B; ++i) s1[i] *= g.beta;
          for (int i = 0; i < B * B; ++i) s2[i] *= g.beta;
          const double nn = g.beta * (double)s_tab[0];
          gg_stream s;
          gg_stream_init(&s, g.seed, GG_SITE_W_NIW, g.sweep, (uint32_t)t);
          if (gg_fixed_cov_mean_draw(&s, g.wh.niw.mu_0, g.sigma + (size_t)t * B * B, B, nn, s1, mo)) atomicOr(&g.ctl[3], 1);
          for (int i = 0; i < B; ++i) g.mu[(size_t)t * B + i] = mo[i];
          // centred scatter of the pair's OFF-DIAGONAL edges (weights.py:545-548): their statistics are the exact
          // integer difference of the pair's and its diagonal's (zero when self connections are special-cased)
          long long off[NTAB];
          for (int q = 0; q < NTAB; ++q) off[q] = s_tab[q] - s_tabd[q];
          gg_unpack_wstats(off + 1, B, g.wh.shift, s1, s2);
          for (int i = 0; i < B; ++i) s1[i] *= g.beta;
          for (int i = 0; i < B * B; ++i) s2[i] *= g.beta;
          const double noff = g.beta * (double)off[0];
          gg_centered_scatter(noff, s1, s2, mo, B, C);
          for (int i = 0; i < B * B; ++i) g.csc[(size_t)t * (B * B + 1) + i] = C[i];
          g.csc[(size_t)t * (B * B + 1) + B * B] = noff;
        } else if (g.do_draw) {
          double s1[GG_MAXB], s2[GG_MAXB * GG_MAXB];
          gg_unpack_wstats(s_tab + 1, B, g.wh.shift, s1, s2);
          gg_stream s;
          gg_stream_init(&s, g.seed, GG_SITE_W_NIW, g.sweep, (uint32_t)t);
          if (wide1) { double w1, w2; gg_wide_collapse(s1, s2, &w1, &w2); s1[0] = w1; s2[0] = w2; }
          const int Bd = wide1 ? 1 : B;
          // tempered posterior (build-defined for the weights: Gaussian likelihood to the power beta scales the
          // sufficient statistics, as graph_models.py:366-367 does for the adjacency counts); beta = 1: x * 1.0 == x
          for (int i = 0; i < Bd; ++i) s1[i] *= g.beta;
          for (int i = 0; i < Bd * Bd; ++i) s2[i] *= g.beta;
          gg_niw_posterior_draw(&s, &g.wh.niw, Bd, g.beta * (double)s_tab[0], s1, s2, mo, so);
        } else if (wide1) {
          mo[0] = g.mu[t]; so[0] = g.sigma[t];
        } else {
          for (int i = 0; i < B; ++i) mo[i] = g.mu[(size_t)t * B + i];
          for (int i = 0; i < B * B; ++i) so[i] = g.sigma[(size_t)t * B * B + i];
        }
        if (g.do_prep && !(g.shared && g.do_draw) && (wide1 ? gg_gauss_table_wide(mo, so, e) : gg_gauss_table(mo, so, B, e))) {
          bad = true;
          for (int i = 0; i < NT; ++i) e[i] = gg_nan();
        }
        double pj = 0.0;
        if (JOINT) {
          // stochastic_block_model.py:232-250 resample_p: edges / non-edges of the block pair, the diagonal included
          if (g.do_draw) {
            const long long n1 = s_tab[0];
            const long long pairs = (long long)g.cnt[a] * (long long)g.cnt[b];
            gg_stream sp;
            gg_stream_init(&sp, g.seed, GG_SITE_ADJ_P, g.sweep, (uint32_t)t);
            pj = gg_beta(&sp, g.ah.tau1 + g.beta * (double)n1, g.ah.tau0 + g.beta * (double)(pairs - n1));
          } else {
            pj = g.p[t];
          }
          if (g.do_prep && !bad) gg_joint_table(e, NTG, pj);
        }
        if (gate_tag && !early_gate) commit = gg_poll_verdict(g.ctl, g.verdict, gate_tag, 0);
        if (commit) {
          if (bad) atomicOr(&g.ctl[3], 1);
          if (JOINT && g.do_draw) g.p[t] = pj;
          if (g.do_draw && !g.shared) {
            if (wide1) { g.mu[t] = mo[0]; g.sigma[t] = so[0]; }
            else {
              for (int i = 0; i < B; ++i) g.mu[(size_t)t * B + i] = mo[i];
              for (int i = 0; i < B * B; ++i) g.sigma[(size_t)t * B * B + i] = so[i];
            }
          }
        }
      }
      if (commit && g.do_prep && !(!ADJ && g.shared && g.do_draw))
        for (int i = 0; i < NT; ++i) {
          g.T[(size_t)t * NT + i] = e[i];
          g.Tout[((size_t)b * NT + i) * g.Kp + a] = e[i];
          g.Tin[((size_t)a * NT + i) * g.Kp + b] = e[i];
        }
      s_commit = commit ? 1 : 0;
    }
    if (gate_tag && !early_gate) {
      __syncthreads();
      if (!s_commit) return;
      housekeeping();
    }
  } else {
    // last workgroup: the self parameter and log m
    if (gate_tag) {
      if (!gg_wait_verdict(g.ctl, g.verdict, gate_tag)) return;
      housekeeping();
    }
    if (threadIdx.x == 0 && g.do_draw) {
      if (ADJ) {
        if (g.ah.special) {
          gg_stream s;
          gg_stream_init(&s, g.seed, g.ah.bb ? GG_SITE_BB_PSELF : GG_SITE_ADJ_PSELF, g.sweep, 0u);
          *g.p_self = gg_beta(&s, g.ah.tau1 + g.beta * (double)g.ah.d1, g.ah.tau0 + g.beta * (double)g.ah.d0);
        }
      } else if (g.wh.special) {
        long long hl[2 * (GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)];
        for (int i = 0; i < NSI; ++i) { hl[2 * i] = g.wh.dS[i] >> 32; hl[2 * i + 1] = g.wh.dS[i] & 0xffffffffll; }
        double s1[GG_MAXB], s2[GG_MAXB * GG_MAXB], mo[GG_MAXB], so[GG_MAXB * GG_MAXB];
        gg_unpack_wstats(hl, B, g.wh.shift, s1, s2);
        gg_stream s;
        gg_stream_init(&s, g.seed, GG_SITE_W_NIW_SELF, g.sweep, 0u);
        const bool wide1 = (B == 2) && g.B == 1;          // (wide weights: see the pair branch above)
        if (wide1) { double w1, w2; gg_wide_collapse(s1, s2, &w1, &w2); s1[0] = w1; s2[0] = w2; }
        const int Bd = wide1 ? 1 : B;
        for (int i = 0; i < Bd; ++i) s1[i] *= g.beta;
        for (int i = 0; i < Bd * Bd; ++i) s2[i] *= g.beta;
        gg_niw_posterior_draw(&s, &g.wh.niw_self, Bd, g.beta * (double)g.wh.d1, s1, s2, mo, so);
        for (int i = 0; i < Bd; ++i) g.mu_self[i] = mo[i];
        for (int i = 0; i < Bd * Bd; ++i) g.sigma_self[i] = so[i];
      }
    }
    __shared__ double s_gam[64];
    const int kk = (int)threadIdx.x - 64;   // wave 1 handles m, wave 0 the self parameter
    if (g.draw_m) {
      if (kk >= 0 && kk < K) s_gam[kk] = gg_dirichlet_gamma_at(g.seed, g.m_site, g.m_sweep, kk, g.pi[kk] + (double)g.cnt[kk]);
    }
    __syncthreads();
    if (kk >= 0 && kk < K) {
      double mk;
      if (g.draw_m) {
        double tot = 0.0;
        for (int j = 0; j < K; ++j) tot += s_gam[j];
        mk = s_gam[kk] / tot;
        g.m[kk] = mk;
      } else {
        mk = g.m[kk];
      }
      if (g.do_prep) g.logm[kk] = gg_log(mk);
    }
  }
}

template <bool ADJ, int B, int RIDE = 0, bool JOINT = false>   // RIDE: 0 none, 1 riders with plain loads, 2 riders with non-temporal loads
__global__ __launch_bounds__(256) void k_params(gg_params_args g, gg_ride_args ride = gg_ride_args()) {
  if (g.ctl[4]) return;
  if (RIDE && (int)blockIdx.x >= ride.front) {
    gg_ride<ADJ, B, RIDE == 2>(ride);
    return;
  }
  const unsigned nfront = RIDE ? (unsigned)ride.front : gridDim.x;   // the workgroups that do k_params' own work
  gg_params_body<ADJ, B, JOINT>(g, (int)blockIdx.x, nfront, 0u);
}

// The parameter step for more than 64 blocks (round 3): K K + 1 workgroups of 256 threads are 10 001 workgroups at 100 blocks,
// each reducing a pair's table over the ~N / K = 41 nodes of a block - one wavefront's worth.  Here a WAVE owns a pair (four
// pairs per workgroup, no workgroup barrier anywhere): the same integer tables, the same draws from the same Philox streams as
// gg_params_body (plain SBM kinds: not the joint model, not the shared covariance), and the last workgroup forms log m for all
// K <= 256 blocks itself (no k_logm launch behind it).  nfront = ceil(K K / 4) + 1 workgroups.
template <bool ADJ, int B>
__device__ __forceinline__ void gg_params_body_w(gg_params_args& g, const int bid, const unsigned nfront) {
  const int K = g.K, KK = K * K;
  constexpr int NSI = ADJ ? 0 : B + B * (B + 1) / 2;
  constexpr int NTAB = ADJ ? 1 : 1 + 2 * NSI;
  constexpr int NT = ADJ ? 2 : 1 + B + B * (B + 1) / 2;
  // housekeeping, as gg_params_body's
  if (g.do_prep && bid == 0 && threadIdx.x == 0) { g.ctl[7] = 0; g.ctl[8] = 0; g.ctl[9] = 0; }
  if (g.do_prep && g.bar)
    for (unsigned sl = bid * blockDim.x + threadIdx.x; sl < 512u * 4u; sl += nfront * blockDim.x) g.bar[(size_t)(sl >> 2) * GG_SLOT_PAD + (sl & 3u)] = 0ull;
  if (g.do_prep && g.ap)
    for (unsigned i = bid * blockDim.x + threadIdx.x; i < (unsigned)g.ap_n; i += nfront * blockDim.x) g.ap[i] = 0;
  if (g.zero_ptr) {
    size_t i = (size_t)bid * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)nfront * blockDim.x;
    for (; i < g.zero_words; i += stride) g.zero_ptr[i] = 0ull;
  }
  const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
  if (bid == (int)nfront - 1) {
    // last workgroup: the self parameter (one lane) and log m
    if (threadIdx.x == 0 && g.do_draw) {
      if (ADJ) {
        if (g.ah.special) {
          gg_stream s;
          gg_stream_init(&s, g.seed, g.ah.bb ? GG_SITE_BB_PSELF : GG_SITE_ADJ_PSELF, g.sweep, 0u);
          *g.p_self = gg_beta(&s, g.ah.tau1 + g.beta * (double)g.ah.d1, g.ah.tau0 + g.beta * (double)g.ah.d0);
        }
      } else if (g.wh.special) {
        long long hl[2 * (GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)];
        for (int i = 0; i < NSI; ++i) { hl[2 * i] = g.wh.dS[i] >> 32; hl[2 * i + 1] = g.wh.dS[i] & 0xffffffffll; }
        double s1[GG_MAXB], s2[GG_MAXB * GG_MAXB], mo[GG_MAXB], so[GG_MAXB * GG_MAXB];
        gg_unpack_wstats(hl, B, g.wh.shift, s1, s2);
        gg_stream s;
        gg_stream_init(&s, g.seed, GG_SITE_W_NIW_SELF, g.sweep, 0u);
        for (int i = 0; i < B; ++i) s1[i] *= g.beta;
        for (int i = 0; i < B * B; ++i) s2[i] *= g.beta;
        gg_niw_posterior_draw(&s, &g.wh.niw_self, B, g.beta * (double)g.wh.d1, s1, s2, mo, so);
        for (int i = 0; i < B; ++i) g.mu_self[i] = mo[i];
        for (int i = 0; i < B * B; ++i) g.sigma_self[i] = so[i];
      }
    }
    if (g.do_prep && (int)threadIdx.x < K) g.logm[threadIdx.x] = gg_log(g.m[threadIdx.x]);   // (m: drawn by k_draw_m in front of this launch)
    return;
  }
  const int t = bid * 4 + wave;
  if (t >= KK) return;                 // (wave-uniform)
  const int a = t / K, b = t % K;
  long long tabv[NTAB];
#pragma unroll
  for (int q = 0; q < NTAB; ++q) tabv[q] = 0;
  if (g.do_tables) {
    const int s0 = g.lab_start[a], s1 = g.lab_start[a + 1];
    for (int r = s0 + lane; r < s1; r += 64) {
      const int n = g.perm[r] - g.n0;
      if (n < 0 || n >= g.Nl) continue;   // not an owned node
      tabv[0] += g.Gcnt0[(size_t)b * g.Np + n];
      if (!ADJ)
#pragma unroll
        for (int s = 0; s < NSI; ++s) {
          const long long v = g.GS0[((size_t)b * g.Np + n) * NSI + s];
          tabv[1 + 2 * s] += (v >> 32);
          tabv[2 + 2 * s] += (v & 0xffffffffll);
        }
      if (g.add_diag && a == b && g.dA[n]) {
        tabv[0] += 1;
        if (!ADJ) {
          const int32_t* q = g.dWq + (size_t)n * B;
          int tt = 0;
          for (int r1 = 0; r1 < B; ++r1, ++tt) {
            const long long v = q[r1];
            tabv[1 + 2 * tt] += (v >> 32);
            tabv[2 + 2 * tt] += (v & 0xffffffffll);
          }
          for (int r1 = 0; r1 < B; ++r1)
            for (int r2 = r1; r2 < B; ++r2, ++tt) {
              const long long v = (long long)q[r1] * (long long)q[r2];
              tabv[1 + 2 * tt] += (v >> 32);
              tabv[2 + 2 * tt] += (v & 0xffffffffll);
            }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NTAB; ++q) tabv[q] = gg_wave_sum_ll(tabv[q]);     // (lane 0 holds the sums)
    if (lane == 0)
#pragma unroll
      for (int q = 0; q < NTAB; ++q) g.tab[(size_t)t * NTAB + q] = tabv[q];
  } else if (g.tables_from_tab) {
#pragma unroll
    for (int q = 0; q < NTAB; ++q) tabv[q] = g.tab[(size_t)t * NTAB + q];
  }
  if (lane != 0) return;
  double e[GG_MAXNT];
  bool bad = false;
  if (ADJ) {
    double pv;
    if (g.do_draw) {
      const long long n1 = tabv[0];
      long long pairs = (long long)g.cnt[a] * (long long)g.cnt[b];
      if (a == b && g.ah.special) pairs -= g.cnt[a];
      pairs *= (long long)g.ah.ngraphs;
      const long long n0 = pairs - n1;
      gg_stream s;
      gg_stream_init(&s, g.seed, g.ah.bb ? GG_SITE_BB_P : GG_SITE_ADJ_P, g.sweep, (uint32_t)t);
      pv = gg_beta(&s, g.ah.tau1 + g.beta * (double)n1, g.ah.tau0 + g.beta * (double)n0);
      g.p[t] = pv;
    } else {
      pv = g.p[t];
    }
    e[0] = gg_log(pv);
    e[1] = gg_log(1.0 - pv);
  } else {
    double mo[GG_MAXB], so[GG_MAXB * GG_MAXB];
    if (g.do_draw) {
      double s1[GG_MAXB], s2[GG_MAXB * GG_MAXB];
      gg_unpack_wstats(tabv + 1, B, g.wh.shift, s1, s2);
      gg_stream s;
      gg_stream_init(&s, g.seed, GG_SITE_W_NIW, g.sweep, (uint32_t)t);
      for (int i = 0; i < B; ++i) s1[i] *= g.beta;
      for (int i = 0; i < B * B; ++i) s2[i] *= g.beta;
      gg_niw_posterior_draw(&s, &g.wh.niw, B, g.beta * (double)tabv[0], s1, s2, mo, so);
      for (int i = 0; i < B; ++i) g.mu[(size_t)t * B + i] = mo[i];
      for (int i = 0; i < B * B; ++i) g.sigma[(size_t)t * B * B + i] = so[i];
    } else {
      for (int i = 0; i < B; ++i) mo[i] = g.mu[(size_t)t * B + i];
      for (int i = 0; i < B * B; ++i) so[i] = g.sigma[(size_t)t * B * B + i];
    }
    if (g.do_prep && gg_gauss_table(mo, so, B, e)) {
      bad = true;
      for (int i = 0; i < NT; ++i) e[i] = gg_nan();
    }
    if (bad) atomicOr(&g.ctl[3], 1);
  }
  if (g.do_prep)
    for (int i = 0; i < NT; ++i) {
      g.T[(size_t)t * NT + i] = e[i];
      g.Tout[((size_t)b * NT + i) * g.Kp + a] = e[i];
      g.Tin[((size_t)a * NT + i) * g.Kp + b] = e[i];
    }
}

template <bool ADJ, int B, int RIDE = 0>   // RIDE as k_params'
__global__ __launch_bounds__(256) void k_params_w(gg_params_args g, gg_ride_args ride = gg_ride_args()) {
  if (g.ctl[4]) return;
  if (RIDE && (int)blockIdx.x >= ride.front) {
    gg_ride<ADJ, B, RIDE == 2>(ride);
    return;
  }
  const unsigned nfront = RIDE ? (unsigned)ride.front : gridDim.x;
  gg_params_body_w<ADJ, B>(g, (int)blockIdx.x, nfront);
}

#define GG_RS_SPOS 22        // control block: scan position of the sharded chain (global node index) - for the host; the device
                             //                reads gg_shard_state
#define GG_RS_ROUNDS 14      //                rounds of the sharded scan that did something
#ifndef GG_SHARD_M
#define GG_SHARD_M 64        // surprises a rank reports per round (k_shard_eval); (-DGG_SHARD_M=512 on ONE rank emulates what 8 ranks report per round: tools/scripts/shard_model.sh)
#endif
#define GG_SHARD_NONE 0x7fffffff

// ------------------------------------------------------------------ per-node conditional
struct gg_node_ctx {
  double beta;     // AIS inverse temperature (1 = plain Gibbs)
  int n0;          // global index of local node 0 (0 unless the chain is sharded); N = number of LOCAL nodes
  int N, Np, K, Kp, B, shift;
  int special;     // special_case_self_conns
  const int32_t* c;
  const int32_t* cnt;
  const int32_t* cntn;   // [N][K] block sizes as node n sees them (sharded scan with hypotheses), or null: cnt for every node
  const double* logm;
  const double* T;
  const double* Tout;
  const double* Tin;
  const int32_t* Gcnt[2];
  const long long* GS[2];
  const uint8_t* dA;
  const int32_t* dWq;
};

// Statistics -> double, spec order {n | ones, zeros}, {s1_r}, {s2_rs}.
template <int NT>
struct gg_statvec { double v[NT]; };

// The categorical draw of gg_lp.h::gg_categorical executed by a group of Kp lanes
// (lane k holds lp[k]; lanes k >= K hold -inf).  Returns the label on every lane.
__device__ __forceinline__ int gg_group_categorical(double lp, int k, int K, int Kp, double u) {
  double mx = lp;
  for (int s = Kp >> 1; s >= 1; s >>= 1) {
    double o = __shfl_xor(mx, s, Kp);
    mx = o > mx ? o : mx;
  }
  double x = (k < K) ? gg_exp(lp - mx) : 0.0;
  for (int s = 1; s < Kp; s <<= 1) {
    double o = __shfl_up(x, s, Kp);
    if (k >= s) x = x + o;
  }
  double tot = __shfl(x, K - 1, Kp);
  double thr = u * tot;
  bool flag = (k < K) && (thr > x);
  unsigned long long bal = __ballot(flag);
  int lane = threadIdx.x & 63;
  int g0 = lane & ~(Kp - 1);
  unsigned long long mask = (Kp >= 64) ? ~0ull : (((1ull << Kp) - 1ull) << g0);
  int idx = __popcll(bal & mask);
  return idx > K - 1 ? K - 1 : idx;
}

// The same draw for Kp = 128 or 256 (65 <= K <= 256): the group spans two or four wavefronts, so the maximum and the scan
// go through LDS.  The scan is the Kogge-Stone scan of gg_lp.h::gg_categorical over Kp elements, step by step
// (y[k] = x[k] + x[k - s]): the same additions in the same order as the oracle's, hence the same bits.  Every thread of
// the workgroup calls it (it synchronises); scr: 2 * blockDim doubles + 16 ints of LDS nobody else uses meanwhile.
__device__ __forceinline__ int gg_group_categorical_wide(double lp, int k, int K, int Kp, double u, double* scr) {
  const int tid = (int)threadIdx.x, BD = (int)blockDim.x;
  const int g0 = tid - k;                       // first thread of my group
  int* iscr = reinterpret_cast<int*>(scr + 2 * BD);
  // maximum over the group: butterfly through LDS (any order gives the same maximum)
  double mx = lp;
  scr[tid] = mx;
  __syncthreads();
  for (int s = Kp >> 1; s >= 1; s >>= 1) {
    const double o = scr[g0 + (k ^ s)];
    __syncthreads();
    mx = o > mx ? o : mx;
    scr[tid] = mx;
    __syncthreads();
  }
  double x = (k < K) ? gg_exp(lp - mx) : 0.0;
  int cur = 0;
  scr[tid] = x;
  __syncthreads();
  for (int s = 1; s < Kp; s <<= 1) {
    double v = scr[cur * BD + tid];
    if (k >= s) v = v + scr[cur * BD + tid - s];
    cur ^= 1;
    scr[cur * BD + tid] = v;
    __syncthreads();
  }
  x = scr[cur * BD + tid];
  const double tot = scr[cur * BD + g0 + K - 1];
  const double thr = u * tot;
  const bool flag = (k < K) && (thr > x);
  const unsigned long long bal = __ballot(flag);
  if ((tid & 63) == 0) iscr[tid >> 6] = __popcll(bal);
  __syncthreads();
  int idx = 0;
  for (int w = 0; w < (Kp >> 6); ++w) idx += iscr[(g0 >> 6) + w];
  __syncthreads();      // (iscr and scr are free again)
  return idx > K - 1 ? K - 1 : idx;
}

// The self connection's chain of node n under label k (gg_block_lp's last term; models that do not special-case the diagonal):
// depends on the node's diagonal entry and the coefficients of pair (k, k) only - the same in every round of a scan.
template <int NT, bool ADJ>
__device__ __forceinline__ double gg_node_self_acc(const gg_node_ctx& x, int n, int k) {
  constexpr int B = gg_tB(NT);
  const int K = x.K;
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  double self[NT];
  if (ADJ) {
    double a = x.dA[n] ? 1.0 : 0.0;
    self[0] = a; self[1] = 1.0 - a;
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) self[t] = 0.0;
    if (x.dA[n]) {
      const int32_t* q = x.dWq + (size_t)n * B;
      self[0] = 1.0;
      int t = 1;
      for (int r = 0; r < B; ++r) self[t++] = (double)q[r] * sc1;
      for (int r = 0; r < B; ++r)
        for (int s = r; s < B; ++s) self[t++] = (double)((long long)q[r] * (long long)q[s]) * sc2;
    }
    if (gg_tJ(NT)) self[NT - 1] = 1.0;   // the node is its own one potential partner (stochastic_block_model.py:291-296)
  }
  double acc = 0.0;
#pragma unroll
  for (int t = 0; t < NT; ++t) acc = gg_fma(x.T[((size_t)k * K + k) * NT + t], self[t], acc);
  return acc;
}

// Combination of the finished chains of node n, label k (the tail of gg_lp.h::gg_block_lp).
template <int NT, bool ADJ>
// self_pre: the self connection's chain when the caller has it already (it does not change during a band of a scan).
__device__ __forceinline__ double gg_node_lp_finish(const gg_node_ctx& x, int n, int k, const double* acc_o,
                                                    const double* acc_i, const double* logm_k = nullptr,
                                                    const double* self_pre = nullptr) {
  if (k >= x.K) return -gg_inf();
  const int K = x.K;
  double lp = logm_k ? *logm_k : x.logm[k];
#pragma unroll
  for (int t = 0; t < NT; ++t) lp = gg_fma(x.beta, acc_o[t], lp);
#pragma unroll
  for (int t = 0; t < NT; ++t) lp = gg_fma(x.beta, acc_i[t], lp);
  if (!x.special) lp = gg_fma(x.beta, self_pre ? *self_pre : gg_node_self_acc<NT, ADJ>(x, n, k), lp);
  return lp;
}

// lp[k] of one node (gg_lp.h::gg_block_lp order).  `ld(j, so, si)` supplies the
// node's out / in statistics towards block j as doubles in spec order.
template <int NT, bool ADJ, int UJ_ = 0, class Loader>
__device__ __forceinline__ double gg_node_lp_core(const gg_node_ctx& x, int n, int k, Loader ld,
                                                  const double* __restrict__ Tout, const double* __restrict__ Tin,
                                                  const double* logm_k = nullptr, const double* self_pre = nullptr) {
  if (k >= x.K) return -gg_inf();
  const int K = x.K, Kp = x.Kp;
  double acc_o[NT], acc_i[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc_o[t] = 0.0; acc_i[t] = 0.0; }
  // The table entries of UJ consecutive j are fetched as one batch before any of them is used (the
  // compiler's own schedule waits for every load right where it issues it); the fma order is unchanged.
  constexpr int UJ = UJ_ > 0 ? UJ_ : (NT <= 3 ? 8 : (NT <= 6 ? 4 : 2));   // (UJ_: callers short of registers ask for less)
  int j = 0;
  for (; j + UJ <= K; j += UJ) {
    double to[UJ][NT], ti[UJ][NT];
#pragma unroll
    for (int u = 0; u < UJ; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        to[u][t] = Tout[((size_t)(j + u) * NT + t) * Kp + k];
        ti[u][t] = Tin[((size_t)(j + u) * NT + t) * Kp + k];
      }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < UJ; ++u) {
      double so[NT], si[NT];
      ld(j + u, so, si);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc_o[t] = gg_fma(to[u][t], so[t], acc_o[t]);
        acc_i[t] = gg_fma(ti[u][t], si[t], acc_i[t]);
      }
    }
  }
  for (; j < K; ++j) {
    double so[NT], si[NT];
    ld(j, so, si);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc_o[t] = gg_fma(Tout[((size_t)j * NT + t) * Kp + k], so[t], acc_o[t]);
      acc_i[t] = gg_fma(Tin[((size_t)j * NT + t) * Kp + k], si[t], acc_i[t]);
    }
  }
  return gg_node_lp_finish<NT, ADJ>(x, n, k, acc_o, acc_i, logm_k, self_pre);
}

// The same with ONE copy of the table T[a][b][t] (rows padded to RS doubles, RS odd: lane k's reads of row k fall into
// different LDS banks) instead of the two lane-major layouts: out-coefficient (j, t) of lane k = T[k][j][t], in-coefficient
// = T[j][k][t].  Half the LDS of the two layouts: what lets K = 32..64 keep the table next to the statistics (k_scan).
template <int NT, bool ADJ, class Loader>
__device__ __forceinline__ double gg_node_lp_core_t1(const gg_node_ctx& x, int n, int k, Loader ld,
                                                     const double* __restrict__ T1, int RS, const double* logm_k = nullptr,
                                                     const double* self_pre = nullptr) {
  if (k >= x.K) return -gg_inf();
  const int K = x.K;
  double acc_o[NT], acc_i[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc_o[t] = 0.0; acc_i[t] = 0.0; }
  const double* To = T1 + (size_t)k * RS;
  const double* Ti = T1 + (size_t)k * NT;
  constexpr int UJ = NT <= 3 ? 4 : 2;
  int j = 0;
  for (; j + UJ <= K; j += UJ) {
    double to[UJ][NT], ti[UJ][NT];
#pragma unroll
    for (int u = 0; u < UJ; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        to[u][t] = To[(j + u) * NT + t];
        ti[u][t] = Ti[(size_t)(j + u) * RS + t];
      }
#pragma unroll
    for (int u = 0; u < UJ; ++u) {
      double so[NT], si[NT];
      ld(j + u, so, si);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc_o[t] = gg_fma(to[u][t], so[t], acc_o[t]);
        acc_i[t] = gg_fma(ti[u][t], si[t], acc_i[t]);
      }
    }
  }
  for (; j < K; ++j) {
    double so[NT], si[NT];
    ld(j, so, si);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc_o[t] = gg_fma(To[j * NT + t], so[t], acc_o[t]);
      acc_i[t] = gg_fma(Ti[(size_t)j * RS + t], si[t], acc_i[t]);
    }
  }
  return gg_node_lp_finish<NT, ADJ>(x, n, k, acc_o, acc_i, logm_k, self_pre);
}

// Where the spec-order doubles of (direction d, block j, node i of the workgroup) live in LDS: element t is at
// base + t * stride.  Normal layout [dir][j][node][NT]; split layout (Kp = 64, tables not staged: gg_lp_split4)
// [dir][j][NT][node], so that a chain reads its statistic for all four nodes with one wide load.
__device__ __forceinline__ size_t gg_st_slot(bool split, int dj /* d * K + j */, int i, int per, int NT, int* stride) {
  *stride = split ? per : 1;
  return split ? ((size_t)dj * NT) * per + i : ((size_t)dj * per + i) * NT;
}

// The same conditional for a workgroup of 4 waves that holds 4 nodes with Kp = 64 (33 <= K <= 64), when the
// coefficient tables do not fit in LDS: the 2 NT independent fma chains (direction, statistic) are dealt
// out to the waves, each wave running its chains for all 4 nodes, so every table entry is fetched once per
// workgroup instead of once per node; the chains meet in LDS and wave i combines those of node i in the
// order of gg_block_lp.  Same chains, same order, same bits.  s_st: [dir][j][NT][4] (gg_st_slot with split = true),
// s_acc: [4][2 NT][64].
// PER: nodes per (j, t) of the statistics' split layout [dir][j][t][PER] (4, or 8 when a group of the resident scan holds two
// passes' worth: the caller then hands in s_st + 4 p for pass p; a compile-time stride - as a run-time one it cost the chains
// a quarter more).
template <int NT, bool ADJ, int PER = 4>
__device__ __forceinline__ double gg_lp_split4(const gg_node_ctx& x, const double* s_st, double* s_acc, int n,
                                               const double* T1 = nullptr, int RS = 0) {
  constexpr int NCH = 2 * NT;
  constexpr int CPW = (NCH + 3) / 4;                                         // chains per wave
#ifndef GG_SPLIT_UJ2
#define GG_SPLIT_UJ2 16
#endif
  constexpr int UJ = CPW == 1 ? 64 : (CPW == 2 ? GG_SPLIT_UJ2 : (CPW <= 4 ? 16 : 8));   // table entries in flight per chain
  const int w = threadIdx.x >> 6, k = threadIdx.x & 63, K = x.K;
  // values the tail needs, fetched now so that their latency hides behind the chains
  const double logm_k = x.logm[k < K ? k : 0];
  double acc[CPW][4];
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = 0.0;
  if (k < K) {
    // One batch of UJ consecutive j: the table entries are fetched first (all in flight together), then the chains run.
    // FULL batches carry no bounds checks: with a check per step every step becomes its own basic block, whose LDS read
    // of the four nodes' statistics is waited for before its four fma - 100 clocks per step, 8 of 10 us of a K = 64
    // evaluation (measured); without them the block is straight-line code and the reads run ahead of the arithmetic.
    auto batch = [&](int j, auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      double tv[CPW][UJ];
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int ch = w + 4 * c;
        if (ch < NCH) {
          const int d = ch / NT, t = ch % NT;
          if (T1) {
            // one padded copy of T[a][b][t] in LDS (gg_node_lp_core_t1): out = T[k][j][t], in = T[j][k][t]
            const double* Tb = T1 + (d == 0 ? (size_t)k * RS : (size_t)k * NT) + t;
            const int js = d == 0 ? NT : RS;
#pragma unroll
            for (int u = 0; u < UJ; ++u) {
              const int jj = (FULL || j + u < K) ? j + u : K - 1;
              tv[c][u] = Tb[(size_t)jj * js];
            }
          } else {
            const double* Tb = (d == 0 ? x.Tout : x.Tin) + (size_t)t * 64 + k;
#pragma unroll
            for (int u = 0; u < UJ; ++u) {
              const int jj = (FULL || j + u < K) ? j + u : K - 1;
              tv[c][u] = Tb[(size_t)jj * NT * 64];
            }
          }
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int ch = w + 4 * c;
        if (ch < NCH) {
          const int d = ch / NT, t = ch % NT;
          // split layout of the statistics: [dir][j][t][node] - the four nodes of a (j, t) are one 32-byte read
          const double* sb = s_st + ((size_t)d * K * NT + t) * PER;
          // the four nodes' statistics of UB steps are read before the first of them is used (left to itself the compiler
          // reads each step's 32 bytes right in front of its four fma and waits for them: one LDS latency per step)
          constexpr int UB = UJ < 4 ? UJ : 4;
#pragma unroll
          for (int u0 = 0; u0 < UJ; u0 += UB) {
            double spv[UB][4];
#pragma unroll
            for (int b2 = 0; b2 < UB; ++b2) {
              const int jj = (FULL || j + u0 + b2 < K) ? j + u0 + b2 : K - 1;
              const double* sp = sb + (size_t)jj * NT * PER;
#pragma unroll
              for (int i = 0; i < 4; ++i) spv[b2][i] = sp[i];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int b2 = 0; b2 < UB; ++b2) {
              if (FULL || j + u0 + b2 < K) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[c][i] = gg_fma(tv[c][u0 + b2], spv[b2][i], acc[c][i]);
              }
            }
          }
        }
      }
    };
    int j = 0;
    for (; j + UJ <= K; j += UJ) batch(j, std::true_type());
    if (j < K) batch(j, std::false_type());
  }
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int ch = w + 4 * c;
    if (ch < NCH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s_acc[(i * NCH + ch) * 64 + k] = acc[c][i];
    }
  }
  __syncthreads();
  double acc_o[NT], acc_i[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc_o[t] = s_acc[(w * NCH + t) * 64 + k];
    acc_i[t] = s_acc[(w * NCH + NT + t) * 64 + k];
  }
  return gg_node_lp_finish<NT, ADJ>(x, n, k, acc_o, acc_i, &logm_k);
}

// The same conditional on the matrix cores (33 <= K <= 64, a workgroup of 4 waves, PER = 8 or 16 nodes at once).
// Chain c = (direction, statistic) of gg_block_lp is a matrix product: Acc_c[node][k] = sum_j stat_c[node][j] * T_c[j][k], and
// v_mfma_f64_16x16x4_f64 IS a chain of four fused multiply-adds in ascending order of its inner index, bit for bit
// (tools/probe/mfma_f64_exact.hip: 4.7 M outputs at inner lengths 4, 16, 64, all identical to fma chains, none of the other
// candidate orders), so instruction after instruction over j = 0, 4, 8, ... produces the spec's chain - same order, same bits.
// Wave w owns the 16 blocks k = 16 w .. 16 w + 15 for all nodes and all 2 NT chains: per step of four j a lane supplies one
// statistic (node l % 16, block j = 4 s + l / 16: one LDS read per chain from the split layout [dir][j][t][PER]) and one
// coefficient (block j, label k: Tout / Tin as they lie in global memory, 128-byte segments) and the matrix pipe does the 16 x 16
// x 4 products - against gg_lp_split4, where every fma needed its coefficient loaded per lane AND a broadcast LDS read of the
// statistic, a coefficient now serves 16 nodes and a statistic 16 labels.  The coefficients of the next batch of steps are in
// flight while the current batch multiplies.  Results: lane l, element r = (node 4 r + l / 16, label 16 w + l % 16); each lane
// finishes its (node, label) entries (gg_node_lp_finish) and leaves lp in s_lp[node][64], where the caller's waves pick a node
// each for the categorical draw.  Padding (j >= K in the last step, nodes >= PER) multiplies zeros: fma(0, 0, acc) = acc.
typedef double gg_d4 __attribute__((ext_vector_type(4)));
#define GG_MFMA_RING 4     /* steps (of four j) whose coefficients are in flight ahead of the matrix pipe */
// The coefficients of step s as this lane supplies them (the caller fetches the first GG_MFMA_RING steps ahead of the evaluation -
// the tables do not change during a scan).
template <int NT>
__device__ __forceinline__ void gg_lp_mfma_b1(const gg_node_ctx& x, int step, double (&b)[2 * NT]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, K = x.K;
  const int j = 4 * step + (l >> 4), kk = 16 * w + (l & 15);
  // (a step beyond K reads block K - 1's coefficients again: its statistics are supplied as zeros, and 0 * finite adds nothing -
  // no select on the loaded value, which would make the wave wait for the load where it is issued)
  const size_t at = ((size_t)(j < K ? j : K - 1) * NT) * 64 + kk;
#pragma unroll
  for (int c = 0; c < 2 * NT; ++c) b[c] = (c < NT ? x.Tout : x.Tin)[at + (size_t)(c % NT) * 64];
}
// selfacc[r]: gg_node_self_acc of (node 4 r + l / 16, label kk) - constant over a band's rounds, so the caller computes it once
// per band instead of behind every round's chains (unused when the model special-cases the diagonal).
// s_st[zoff]: an LDS double that holds 0.0 - what a lane reads for a statistic that does not exist (j >= K, node >= PER): an
// address select instead of a select on the loaded value, so that a step's reads are all in flight before the first is waited
// for.  (An OFFSET from s_st, not a pointer: a select between two pointers loses the LDS address space, the reads become flat
// loads, and a flat load is waited for together with every global load in flight - the coming steps' coefficients.)
// Schedule (fixed with scheduling barriers - left to itself the scheduler sinks the loads below the matrix instructions to save
// registers and then waits for them at once): step i multiplies while the statistics of step i + 1 are being read and the
// coefficients of steps i + 1 .. i + 4 are in flight (a ring of four: an L2 round trip is longer than two steps' 12 matrix
// instructions, measured).
template <int NT, bool ADJ, int PER>
__device__ __forceinline__ void gg_lp_mfma(const gg_node_ctx& x, const double* s_st, double* s_lp, double logm_kk,
                                           const double (&b0)[GG_MFMA_RING][2 * NT], const double (&selfacc)[4 * (PER > 16 ? PER / 16 : 1)],
                                           int zoff) {
  static_assert(PER == 8 || PER == 16 || PER == 32, "one or two 16-row tiles of nodes");
  constexpr int NCH = 2 * NT;
  constexpr int R = GG_MFMA_RING;
  constexpr int NRT = PER > 16 ? PER / 16 : 1;     // row tiles: the coefficients of a step serve all of them
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, K = x.K;
  const int jr = l >> 4, c16 = l & 15;
  const int kk = 16 * w + c16;                // the label this lane's results belong to
  const int S = (K + 3) >> 2;
  gg_d4 acc[NRT][NCH];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[rt][c] = gg_d4{0.0, 0.0, 0.0, 0.0};
  double bq[R][NCH], av[2][NRT][NCH];
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int c = 0; c < NCH; ++c) bq[u][c] = b0[u][c];
  auto load_a = [&](int step, double (&a)[NRT][NCH]) {
    const int j = 4 * step + jr;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const int node = 16 * rt + c16;
      const bool in = node < PER && j < K;
      const int base = in ? j * NT * PER + node : zoff;
      const int d1 = in ? K * NT * PER : 0, t1 = in ? PER : 0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) a[rt][c] = s_st[base + (c / NT) * d1 + (c % NT) * t1];
    }
  };
  load_a(0, av[0]);
  for (int s0 = 0; s0 < S; s0 += R) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int step = s0 + i;
      load_a(step + 1, av[(i + 1) & 1]);      // (beyond the last step: zeros)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          acc[rt][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i & 1][rt][c], bq[i][c], acc[rt][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (step + R < S) gg_lp_mfma_b1<NT>(x, step + R, bq[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the tail of gg_block_lp (gg_node_lp_finish, with the self chain handed in)
  const bool special = x.special != 0;
  const double beta = x.beta;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * rt + 4 * r + jr;
      double lp = logm_kk;
#pragma unroll
      for (int t = 0; t < NT; ++t) lp = gg_fma(beta, acc[rt][t][r], lp);
#pragma unroll
      for (int t = 0; t < NT; ++t) lp = gg_fma(beta, acc[rt][NT + t][r], lp);
      if (!special) lp = gg_fma(beta, selfacc[4 * rt + r], lp);
      if (kk >= K) lp = -gg_inf();
      if (i < PER) s_lp[i * 64 + kk] = lp;
    }
}

// The same for 9..32 blocks with the coefficient tables in LDS (k_scan, MODE 0, stage_tables == 1): a group is per = 256 / Kp
// nodes (16 or 8) and Kp = 16 or 32 labels - one or two 16 x 16 tiles per chain.  The 2 NT x (Kp / 16) (chain, tile) items are
// dealt out over the four waves; an item is ceil(K / 4) matrix instructions in a row (the chain of gg_block_lp, bit for bit:
// tools/probe/mfma_f64_exact.hip), its statistics read from the normal layout [dir][j][node][NT], its coefficients from the
// lane-major tables [j][t][Kp].  The finished chains meet in s_acc[chain][node][Kp]; after a barrier lane (node, k) of the
// ordinary layout picks its 2 NT values up and finishes as gg_node_lp_core does.  Against the vector form (one fma, one
// coefficient read and one broadcast statistic read per (j, t, direction) and lane): 1.78 against 1.70 us per round at K = 16 -
// no gain (both are latency, not arithmetic); kept behind GG_MFMA_SMALL=1, off by default.
template <int NT, bool ADJ>
__device__ __forceinline__ void gg_lp_mfma_small(const gg_node_ctx& x, const double* s_st, const double* s_To, const double* s_Ti,
                                                 double* s_acc, int per, int Kp, int zoff) {
  constexpr int NCH = 2 * NT;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, K = x.K;
  const int jr = l >> 4, c16 = l & 15;
  const int ntile = Kp >> 4, S = (K + 3) >> 2;
  const bool node_in = c16 < per;
  for (int item = w; item < NCH * ntile; item += 4) {
    const int c = item / ntile, tile = item - c * ntile;
    const int d = c / NT, t = c - d * NT;
    const double* Tb = (d == 0 ? s_To : s_Ti) + (size_t)t * Kp + 16 * tile + c16;
    gg_d4 acc = gg_d4{0.0, 0.0, 0.0, 0.0};
    for (int s0 = 0; s0 < S; s0 += 4) {
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = 4 * (s0 + u) + jr;
        const bool in = node_in && j < K;
        av[u] = s_st[in ? ((d * K + j) * per + c16) * NT + t : zoff];
        bv[u] = Tb[(size_t)(j < K ? j : K - 1) * NT * Kp];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u < S) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);   // (uniform)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int node = 4 * r + jr;
      if (node < per) s_acc[((size_t)c * per + node) * Kp + 16 * tile + c16] = acc[r];
    }
  }
}

// gg_group_categorical for P nodes held by one wave (Kp = 64: lane k holds lp[p][k]), the P draws' reductions and scans
// interleaved - each is a chain of a dozen dependent cross-lane steps, and P chains in flight cost little more than one.
template <int P>
__device__ __forceinline__ void gg_group_categorical64(const double (&lp)[P], int k, int K, const double (&u)[P], int (&lab)[P]) {
  double mx[P], xs[P];
#pragma unroll
  for (int p = 0; p < P; ++p) mx[p] = lp[p];
  for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const double o = __shfl_xor(mx[p], s, 64);
      mx[p] = o > mx[p] ? o : mx[p];
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) xs[p] = (k < K) ? gg_exp(lp[p] - mx[p]) : 0.0;
  for (int s = 1; s < 64; s <<= 1) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const double o = __shfl_up(xs[p], s, 64);
      if (k >= s) xs[p] = xs[p] + o;
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const double tot = __shfl(xs[p], K - 1, 64);
    const double thr = u[p] * tot;
    const bool flag = (k < K) && (thr > xs[p]);
    const int idx = __popcll(__ballot(flag));
    lab[p] = idx > K - 1 ? K - 1 : idx;
  }
}

// Integer statistics -> spec-order doubles.
template <int NT, bool ADJ>
__device__ __forceinline__ void gg_stats_to_double(int cnt_o, int cnt_i, const long long* so_i, const long long* si_i,
                                                   int cx, double sc1, double sc2, double* so, double* si) {
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
  if (ADJ) {
    so[0] = (double)cnt_o; so[1] = (double)(cx - cnt_o);
    si[0] = (double)cnt_i; si[1] = (double)(cx - cnt_i);
  } else {
    so[0] = (double)cnt_o;
    si[0] = (double)cnt_i;
#pragma unroll
    for (int s = 0; s < NSI; ++s) {
      double sc = s < B ? sc1 : sc2;
      so[1 + s] = (double)so_i[s] * sc;
      si[1 + s] = (double)si_i[s] * sc;
    }
    if (gg_tJ(NT)) { so[NT - 1] = (double)cx; si[NT - 1] = (double)cx; }   // joint model: potential partners in the block
  }
}

template <int NT, bool ADJ>
__device__ __forceinline__ double gg_node_lp_global(const gg_node_ctx& x, int n, int k) {
  const int cn = x.c[x.n0 + n];
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  constexpr int NSI = gg_tNSI(NT);
  auto ld = [&](int j, double* so, double* si) {
    const size_t gi = (size_t)j * x.Np + n;
    int cx = (ADJ || gg_tJ(NT)) ? ((x.cntn ? x.cntn[(size_t)n * x.K + j] : x.cnt[j]) - (cn == j ? 1 : 0)) : 0;
    long long a[NSI > 0 ? NSI : 1], b[NSI > 0 ? NSI : 1];
    if (!ADJ) {
#pragma unroll
      for (int s = 0; s < NSI; ++s) { a[s] = x.GS[0][gi * NSI + s]; b[s] = x.GS[1][gi * NSI + s]; }
    }
    gg_stats_to_double<NT, ADJ>(x.Gcnt[0][gi], x.Gcnt[1][gi], a, b, cx, sc1, sc2, so, si);
  };
  return gg_node_lp_core<NT, ADJ>(x, n, k, ld, x.Tout, x.Tin);
}

// Speculative evaluation: every node n >= n_begin draws its label as if no
// earlier node of this scan changed.  Lane = (node, k): Kp lanes per node, 256 / Kp
// nodes per workgroup.  The workgroup first stages its nodes' statistics (one
// coalesced round of loads) and, when they fit, the coefficient tables in LDS:
// the per-j loop then never waits on global memory.
// The evaluation itself (all 256 threads of the workgroup): returns the drawn label of node *n_out for lane (i, k).
// UJ_: table entries fetched per batch (0: the default depth; riding kernels ask for less).  SPLIT: -1 = decide at run time
// whether the K > 32 form (gg_lp_split4, 128+ registers of table entries in flight) runs; 0 / 1 = compiled for one of the
// two only, so that the kernel's register allocation is not that of the form it never runs.
template <int NT, bool ADJ, int UJ_ = 0, int SPLIT = -1>
__device__ __forceinline__ int gg_spec_eval(const gg_node_ctx& x, int n_begin, uint64_t seed, uint32_t site, uint32_t sweep,
                                            int lp_node, double* __restrict__ lp_out, int stage_tables, int* n_out,
                                            bool* live_out) {
  constexpr int NSI = ADJ ? 0 : gg_tNSI(NT);
  extern __shared__ __align__(16) unsigned char s_spec[];
  const int K = x.K, Kp = x.Kp;
  const int per = 256 / Kp;
  // statistics of the workgroup's nodes, already converted to spec-order doubles: [dir][j][node][NT]
  double* s_st = reinterpret_cast<double*>(s_spec);
  double* s_T = s_st + (size_t)2 * K * per * NT;   // Tout | Tin (when staged)
  const int n0 = n_begin + blockIdx.x * per;
  const int tid = threadIdx.x;
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  for (int idx = tid; idx < 2 * K * per; idx += 256) {
    const int i = idx % per, j = (idx / per) % K, d = idx / (per * K);
    const int n = (n0 + i < x.N) ? n0 + i : x.N - 1;
    const size_t gi = (size_t)j * x.Np + n;
    const size_t gd = (size_t)d * K * x.Np + gi;   // direction 1 follows direction 0 in the buffer
    const int gc = x.Gcnt[0][gd];
    int ts;
    double* dst = s_st + gg_st_slot(!stage_tables && Kp == 64, idx / per, i, per, NT, &ts);
    if (ADJ) {
      const int cx = (x.cntn ? x.cntn[(size_t)n * K + j] : x.cnt[j]) - (x.c[x.n0 + n] == j ? 1 : 0);
      dst[0] = (double)gc;
      dst[ts] = (double)(cx - gc);
    } else {
      constexpr int B = gg_tB(NT);
      dst[0] = (double)gc;
#pragma unroll
      for (int s = 0; s < NSI; ++s) dst[(1 + s) * ts] = (double)x.GS[0][gd * NSI + s] * (s < B ? sc1 : sc2);
      if (gg_tJ(NT)) dst[(NT - 1) * ts] = (double)((x.cntn ? x.cntn[(size_t)n * K + j] : x.cnt[j]) - (x.c[x.n0 + n] == j ? 1 : 0));
    }
  }
  const int ntab = K * NT * Kp;
  if (stage_tables) {
    for (int idx = tid; idx < ntab; idx += 256) {
      s_T[idx] = x.Tout[idx];
      s_T[ntab + idx] = x.Tin[idx];
    }
  }
  __syncthreads();
  const int i = tid / Kp, k = tid % Kp;
  const int n = n0 + i;
  const bool live = n < x.N;
  const int nn = live ? n : x.N - 1;
  auto ld = [&](int j, double* so, double* si) {
    const double* a = s_st + (size_t)((0 * K + j) * per + i) * NT;
    const double* b = s_st + (size_t)((1 * K + j) * per + i) * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) { so[t] = a[t]; si[t] = b[t]; }
  };
  // (two call sites: the table pointers keep their address space, LDS or global, instead of becoming flat)
  double lp;
  if (SPLIT != 1 && stage_tables) lp = gg_node_lp_core<NT, ADJ, UJ_>(x, nn, k, ld, s_T, s_T + ntab);
  else if (SPLIT < 0 ? Kp == 64 : SPLIT == 1) lp = gg_lp_split4<NT, ADJ>(x, s_st, s_T /* chains meet here */, nn);
  else lp = gg_node_lp_core<NT, ADJ, UJ_>(x, nn, k, ld, x.Tout, x.Tin);
  if (lp_out && nn == lp_node && live && k < K) lp_out[k] = lp;
  double u = gg_uniform_at(seed, site, sweep, (uint32_t)(x.n0 + nn));
  *n_out = n;
  *live_out = live;
  if (SPLIT == 2) {
    // more than 64 blocks: a node's lanes span two or four wavefronts and the categorical scans through LDS (scratch behind
    // the statistics; the tables are never staged at this size)
    double* s_cat = s_st + (size_t)2 * K * per * NT;
    return gg_group_categorical_wide(lp, k, K, Kp, u, s_cat);
  }
  return gg_group_categorical(lp, k, K, Kp, u);
}

// 33..64 blocks, the sharded scan's evaluation (k_shard_eval): EIGHT nodes per workgroup in two passes of four - the statistics of
// all eight are staged once ([dir][j][t][8]), gg_lp_split4 runs per pass on its half; the launch has half the workgroups and
// (the table entries of a pass still come from L2 per pass) the staging, the uniform and the launch's fixed part are paid once
// for eight nodes.  Same chains, same bits.  Nodes of the workgroup: n_begin + 8 blockIdx.x + 4 p + wave.
template <int NT, bool ADJ>
__device__ __forceinline__ void gg_spec_eval_split8(const gg_node_ctx& x, int n_begin, uint64_t seed, uint32_t site, uint32_t sweep,
                                                    int* labs, int* nodes, bool* lives) {
  constexpr int NSI = ADJ ? 0 : gg_tNSI(NT);
  extern __shared__ __align__(16) unsigned char s_spec[];
  const int K = x.K;
  double* s_st = reinterpret_cast<double*>(s_spec);
  double* s_acc = s_st + (size_t)2 * K * 8 * NT;      // the split chains of a pass meet here
  const int n0 = n_begin + (int)blockIdx.x * 8;
  const int tid = threadIdx.x;
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  // gg_lp_mfma's inputs that do not depend on the statistics, requested first: the first steps' coefficients, log m and the self
  // chains of this lane's (node, label) entries
  double b0[GG_MFMA_RING][2 * NT], selfacc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < GG_MFMA_RING; ++u) gg_lp_mfma_b1<NT>(x, u, b0[u]);
  const int kk = 16 * (tid >> 6) + (tid & 15);
  const double logm_kk = x.logm[kk < K ? kk : 0];
  if (!x.special) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {       // (rows 8..15 of the tile are padding)
      const int nb = n0 + 4 * r + ((tid >> 4) & 3);
      selfacc[r] = gg_node_self_acc<NT, ADJ>(x, nb < x.N ? nb : x.N - 1, kk < K ? kk : 0);
    }
  }
  for (int idx = tid; idx < 2 * K * 8; idx += 256) {
    const int i = idx % 8, j = (idx / 8) % K, d = idx / (8 * K);
    const int n = (n0 + i < x.N) ? n0 + i : x.N - 1;
    const size_t gd = (size_t)d * K * x.Np + (size_t)j * x.Np + n;
    const int gc = x.Gcnt[0][gd];
    int ts;
    double* dst = s_st + gg_st_slot(true, idx / 8, i, 8, NT, &ts);
    if (ADJ) {
      const int cx = (x.cntn ? x.cntn[(size_t)n * K + j] : x.cnt[j]) - (x.c[x.n0 + n] == j ? 1 : 0);
      dst[0] = (double)gc;
      dst[ts] = (double)(cx - gc);
    } else {
      constexpr int B = gg_tB(NT);
      dst[0] = (double)gc;
#pragma unroll
      for (int s = 0; s < NSI; ++s) dst[(1 + s) * ts] = (double)x.GS[0][gd * NSI + s] * (s < B ? sc1 : sc2);
      if (gg_tJ(NT)) dst[(NT - 1) * ts] = (double)((x.cntn ? x.cntn[(size_t)n * K + j] : x.cnt[j]) - (x.c[x.n0 + n] == j ? 1 : 0));
    }
  }
  // (round 4) the eight nodes' chains on the matrix cores: gg_lp_mfma leaves lp[node][64] in the meeting place; s_acc[8 * 64] is
  // the zero it reads for statistics that do not exist
  if (tid == 0) s_acc[8 * 64] = 0.0;
  __syncthreads();
  gg_lp_mfma<NT, ADJ, 8>(x, s_st, s_acc, logm_kk, b0, selfacc, (int)(s_acc + 8 * 64 - s_st));
  __syncthreads();
  const int w = tid >> 6, k = tid & 63;
  double lpp[2], up[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int n = n0 + 4 * p + w;
    const bool live = n < x.N;
    const int nn = live ? n : x.N - 1;
    lpp[p] = s_acc[(4 * p + w) * 64 + k];
    up[p] = gg_uniform_at(seed, site, sweep, (uint32_t)(x.n0 + nn));
    nodes[p] = n;
    lives[p] = live;
  }
  gg_group_categorical64<2>(lpp, k, K, up, *reinterpret_cast<int(*)[2]>(labs));
}

// More than 64 blocks, the quiet scan's evaluation (k_spec_resolve / k_spec_resolve_epi, SPLIT == 2): a node's lanes span two or
// four wavefronts, so a workgroup holds only 256 / Kp = 2 or 1 nodes per pass - and every pass reads both coefficient tables
// (2 K NT Kp doubles: 600 KB at 100 blocks) from L2.  Here a workgroup evaluates GG_WIDE_P(NT) passes' worth of nodes at once:
// lane (i, k) keeps the chains of nodes i, i + per, i + 2 per, ... and every table entry it fetches serves all of them.
// Same chains per node, same order, same bits as gg_spec_eval.  Nodes of the workgroup: [blockIdx.x per P, + per P).
#ifndef GG_WIDE_P3
#define GG_WIDE_P3 4
#endif
#ifndef GG_WIDE_UJ3
#define GG_WIDE_UJ3 4
#endif
#define GG_WIDE_P(NT) ((NT) <= 3 ? GG_WIDE_P3 : ((NT) <= 6 ? 2 : 1))
#define GG_SCAN_WIDE_P(NT) ((NT) <= 3 ? 2 : 1)   // the resident scan's (k_scan, MODE 2): both models' grids share a CU's LDS
// ... and for 33..64 blocks (MODE 1): 4 P nodes per group, evaluated together by gg_lp_mfma - 8 for the adjacency model, 16 for
// one-dimensional weights (whose evaluation is the longer one; the two grids' LDS still fit one CU together: gg_model.hip)
#ifndef GG_SCAN_SPLIT_P3
#define GG_SCAN_SPLIT_P3 4
#endif
#define GG_SCAN_SPLIT_P(NT) ((NT) == 3 ? GG_SCAN_SPLIT_P3 : ((NT) < 3 ? 2 : 1))
#ifndef GG_SCAN_ADJ_PW
#define GG_SCAN_ADJ_PW 4      /* k_scan MODE 3 (the adjacency model's scan under pair scheduling): 4 x this many nodes per group.
                                 8 (32 nodes, two row tiles per step of coefficients: gg_lp_mfma<.., 32>) is exact as well and measured
                                 no faster - 27 rounds of 19 us instead of 46 of 11.5 alone, 6.9-7.2 against 6.9 ms per sweep of the pair */
#endif
// LDS where the evaluation's partial results meet (k_scan, MODE 1): lp[4 P][64] of gg_lp_mfma, or the split chains of gg_lp_split4
#define GG_SCAN_MEET_BYTES(NT) ((size_t)(GG_SCAN_SPLIT_P(NT) > 1 ? 4 * GG_SCAN_SPLIT_P(NT) * 64 : 4 * 2 * (NT) * 64) * sizeof(double))
template <int NT, bool ADJ, int P>
__device__ __forceinline__ void gg_spec_eval_wide(const gg_node_ctx& x, uint64_t seed, uint32_t site, uint32_t sweep, int* labs,
                                                  int* nodes, bool* lives) {
  constexpr int NSI = ADJ ? 0 : gg_tNSI(NT);
  extern __shared__ __align__(16) unsigned char s_spec[];
  const int K = x.K, Kp = x.Kp;
  const int per = 256 / Kp, nwg = per * P;
  // statistics [dir][j][t][i0][p] (node of the workgroup = p per + i0): the P nodes a lane serves are one contiguous read per
  // (j, t) - the reads are wave-uniform broadcasts, and their NUMBER is what the LDS pipe charges for
  double* s_st = reinterpret_cast<double*>(s_spec);
  double* s_cat = s_st + (size_t)2 * K * nwg * NT;                       // the cross-wave categorical's scratch
  const int n0 = (int)blockIdx.x * nwg;
  const int tid = threadIdx.x;
  const double sc1 = gg_pow2i(-x.shift), sc2 = gg_pow2i(-2 * x.shift);
  for (int idx = tid; idx < 2 * K * nwg; idx += 256) {
    const int i = idx % nwg, j = (idx / nwg) % K, d = idx / (nwg * K);
    const int n = (n0 + i < x.N) ? n0 + i : x.N - 1;
    const size_t gd = (size_t)d * K * x.Np + (size_t)j * x.Np + n;        // direction 1 follows direction 0 in the buffer
    const int gc = x.Gcnt[0][gd];
    double* dst = s_st + (((size_t)(d * K + j) * NT) * per + (i % per)) * P + i / per;     // + t * per * P
    const size_t ts = (size_t)per * P;                                                        // stride of t
    if (ADJ) {
      const int cx = (x.cntn ? x.cntn[(size_t)n * K + j] : x.cnt[j]) - (x.c[x.n0 + n] == j ? 1 : 0);
      dst[0] = (double)gc;
      dst[ts] = (double)(cx - gc);
    } else {
      constexpr int B = gg_tB(NT);
      dst[0] = (double)gc;
#pragma unroll
      for (int s = 0; s < NSI; ++s) dst[(1 + s) * ts] = (double)x.GS[0][gd * NSI + s] * (s < B ? sc1 : sc2);
      if (gg_tJ(NT)) dst[(NT - 1) * ts] = (double)((x.cntn ? x.cntn[(size_t)n * K + j] : x.cnt[j]) - (x.c[x.n0 + n] == j ? 1 : 0));
    }
  }
  __syncthreads();
  const int i0 = tid / Kp, k = tid % Kp;
  double acc_o[P][NT], acc_i[P][NT];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc_o[p][t] = 0.0; acc_i[p][t] = 0.0; }
  if (k < K) {
    constexpr int UJ = NT <= 3 ? GG_WIDE_UJ3 : 2;
    auto step = [&](int j, const double* to, const double* ti) {
      const double* a = s_st + (((size_t)(0 * K + j) * NT) * per + i0) * P;
      const double* b = s_st + (((size_t)(1 * K + j) * NT) * per + i0) * P;
      double av[NT][P], bv[NT][P];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < P; ++p) { av[t][p] = a[(size_t)t * per * P + p]; bv[t][p] = b[(size_t)t * per * P + p]; }
#pragma unroll
      for (int p = 0; p < P; ++p)
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
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int n = n0 + p * per + i0;
    const bool live = n < x.N;
    const int nn = live ? n : x.N - 1;
    const double lp = gg_node_lp_finish<NT, ADJ>(x, nn, k, acc_o[p], acc_i[p]);
    const double u = gg_uniform_at(seed, site, sweep, (uint32_t)(x.n0 + nn));
    labs[p] = gg_group_categorical_wide(lp, k, K, Kp, u, s_cat);    // (synchronises: every thread, every pass)
    nodes[p] = n;
    lives[p] = live;
  }
}

template <int NT, bool ADJ>
__global__ __launch_bounds__(256) void k_speculate(gg_node_ctx x, int n_begin, uint64_t seed, uint32_t site,
                                                   uint32_t sweep, int32_t* __restrict__ cand, int lp_node,
                                                   double* __restrict__ lp_out, const int32_t* __restrict__ ctl,
                                                   int stage_tables, int round_mode, int32_t* __restrict__ ap) {
  if (round_mode) {            // device-driven restart round: start where the scan stands, unless it is done
    if (ctl[7]) return;
    n_begin = ctl[0];
    const int first = (int)blockIdx.x * (256 / x.Kp);
    if (n_begin + first >= x.N) return;
    if (ctl[8] > 0 && first >= ctl[8]) return;   // adaptive window chosen by the last resolve
  } else if (ctl && ctl[4]) {
    return;
  }
  int n;
  bool live;
  // (more than 64 blocks: the evaluation with the cross-wave categorical - gg_node_lp and the step-wise forms reach this kernel)
  const int lab = x.Kp > 64 ? gg_spec_eval<NT, ADJ, 0, 2>(x, n_begin, seed, site, sweep, lp_node, lp_out, stage_tables, &n, &live)
                            : gg_spec_eval<NT, ADJ>(x, n_begin, seed, site, sweep, lp_node, lp_out, stage_tables, &n, &live);
  if (live && threadIdx.x % x.Kp == 0) {
    cand[n] = lab;
    if (ap) ap[n] = 0;   // no entry of this scan's change list is applied to the node yet
  }
}

// SBMGaussianWeightSharedCov, second half of resample_mu_and_Sig (weights.py:545-552): the shared covariance
// Sigma ~ IW(lmbda_0 + sum over pairs of their centred scatter, nu_0 + #off-diagonal edges), summed in pair order by
// one lane (same order as the oracle: bit-exact), then every pair gets it and its coefficient table.
template <int B>
__global__ __launch_bounds__(256) void k_shared_cov(gg_params_args g) {
  if (g.ctl[4]) return;
  constexpr int NSI = B + B * (B + 1) / 2;
  constexpr int NTAB = 1 + 2 * NSI;
  constexpr int NT = 1 + B + B * (B + 1) / 2;
  __shared__ double s_sig[B * B];
  const int K = g.K, KK = K * K;
  if (threadIdx.x == 0) {
    double S[B * B], sig[B * B];
    for (int i = 0; i < B * B; ++i) S[i] = g.wh.niw.sigma_0[i];
    double ntot = 0.0;
    for (int ab = 0; ab < KK; ++ab) {
      for (int i = 0; i < B * B; ++i) S[i] += g.csc[(size_t)ab * (B * B + 1) + i];
      ntot += g.csc[(size_t)ab * (B * B + 1) + B * B];      // tempered number of off-diagonal edges of the pair
    }
    gg_stream s;
    gg_stream_init(&s, g.seed, GG_SITE_W_SHCOV, g.sweep, 0u);
    gg_iw_draw(&s, S, g.wh.niw.nu_0 + ntot, B, sig);
    for (int i = 0; i < B * B; ++i) s_sig[i] = sig[i];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < KK; t += 256) {
    const int a = t / K, b = t % K;
    double so[B * B], mo[B], e[GG_MAXNT];
    for (int i = 0; i < B * B; ++i) { so[i] = s_sig[i]; g.sigma[(size_t)t * B * B + i] = so[i]; }
    if (!g.do_prep) continue;
    for (int i = 0; i < B; ++i) mo[i] = g.mu[(size_t)t * B + i];
    if (gg_gauss_table(mo, so, B, e)) {
      atomicOr(&g.ctl[3], 1);
      for (int i = 0; i < NT; ++i) e[i] = gg_nan();
    }
    for (int i = 0; i < NT; ++i) {
      g.T[(size_t)t * NT + i] = e[i];
      g.Tout[((size_t)b * NT + i) * g.Kp + a] = e[i];
      g.Tin[((size_t)a * NT + i) * g.Kp + b] = e[i];
    }
  }
}

// ---- the sharded scan's round (gg_shard_sweep), with hypotheses exactly as in k_scan (gg_scan_kernels.h): every rank keeps
// the same hypothesis H per node of the band, its own nodes' statistics hold every accepted or hypothesised change at a
// position before theirs, and what a rank reports are the first GG_SHARD_M SURPRISES among its own nodes - decisions that
// differ from what H says.  Everything of H before the earliest reported surprise, and that one itself, is the sequential
// chain (a rank's unreported surprises lie behind its reported ones, so behind the earliest report).  One all-gather per ROUND.
//
// Round 4: ownership is block-interleaved (gg_own: blocks of 2^lb nodes dealt out round-robin), so a band's nodes - and the
// evaluation of a round - spread over ALL ranks instead of sitting on the one rank that owns that stretch of the node range; and
// the band is as wide as the hypotheses can carry (W nodes, chosen per scan by the host from the previous scan's change count,
// up to GG_SHARD_WMAX): a round costs one collective whatever its width, and a band needs 1 + (depth of mutually dependent
// decisions) rounds, not one per 4096 nodes.  H lives in global memory, ONE copy: the evaluation only reads it, and in the
// apply launch every workgroup amends, accepts and resets its own slice of the band (a report carries the label its node
// was hypothesised to have, so nobody needs H to price a move).
//
// A round is TWO launches around the all-gather:
//   k_shard_eval   the rank's own nodes of the band behind the scan position evaluated under the hypotheses; the workgroup
//                  that finishes last compacts the first GG_SHARD_M surprises, in position order, into `trip`
//                  (2 ints each: node, old << 16 | hypothesised << 8 | drawn)
//   [all-gather of trip]
//   k_shard_apply  every workgroup: the earliest report; its slice of the band amended / accepted (labels, block sizes) / reset;
//                  the round's moves applied to the statistics of the rank's nodes behind each moved node.
// The scan state (position, band start) is kept TWICE and used alternately (round r reads copy r & 1, workgroup 0 of
// k_shard_apply writes copy (r + 1) & 1), so that no workgroup of a launch reads a word another one writes.
#define GG_SHARD_WMAX 65536      // widest band (nodes): H holds one hypothesis per node of the band
struct gg_shard_state {
  int32_t st[2][2];   // [copy][spos, wbase]
  int32_t ticket;     // workgroups of the running k_shard_eval that are done
  int32_t pad[3];
};

// (two workgroups per CU where the statistics allow it: one stages its nodes while the other multiplies)
template <int NT, bool ADJ>
__global__ __launch_bounds__(256, (NT <= 6 ? 2 : 1)) void k_shard_eval(gg_node_ctx x, gg_own own, int W, uint64_t seed, uint32_t site, uint32_t sweep,
                                                    int32_t* __restrict__ cand, const int32_t* __restrict__ c,
                                                    const int32_t* __restrict__ H, gg_shard_state* __restrict__ S, int round,
                                                    int ntot, int stage_tables, int32_t* __restrict__ trip) {
  __shared__ int s_scan[64];
  __shared__ int s_last, s_found;
  const int t = (int)threadIdx.x;
  const int spos = S->st[round & 1][0], wbase = S->st[round & 1][1];
  int lo = 0, hi = 0;
  if (spos < ntot) {
    const int wend = wbase + W < ntot ? wbase + W : ntot;   // the band the scan stands in
    lo = gg_own_below(own, spos); hi = gg_own_below(own, wend);
  }
  // Workgroups start at multiples of their node count in the LOCAL numbering: the nodes of one workgroup then lie in one
  // ownership block (2^lb >= 8 nodes), where local and global index differ by a constant - which is all the evaluation needs
  // to know about ownership (its x.n0).  Nodes in front of lo are evaluated and dropped.
  const int per = (x.Kp == 64 && NT <= 3) ? 8 : 256 / x.Kp;
  const int lo_al = lo & ~(per - 1);
  const int first = lo_al + (int)blockIdx.x * per;
  // (an atomic exchange at agent scope publishes a decision: performed at the memory side, and its return value holds the wave
  // until it has been - the decision is out before the workgroup's ticket is drawn.  No fence: a release fence per workgroup
  // writes back every dirty line of the XCD's L2, megabytes of statistics - measured at +115 us per round)
  if (hi > lo && first < hi) {                                     // (uniform over the workgroup)
    gg_node_ctx xl = x;
    xl.n0 = gg_own_glob(own, first) - first;
    if (x.Kp == 64 && NT <= 3) {                                   // eight nodes per workgroup, two passes of four
      int labs[2], nodes[2];
      bool lives[2];
      gg_spec_eval_split8<NT, ADJ>(xl, lo_al, seed, site, sweep, labs, nodes, lives);
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (lives[p] && nodes[p] >= lo && nodes[p] < hi && (t & 63) == 0) {
          const int old = atomicExch(&cand[nodes[p]], labs[p]);
          asm volatile("" ::"v"(old) : "memory");
        }
    } else {
      int n;
      bool live;
      const int lab = x.Kp > 64 ? gg_spec_eval<NT, ADJ, 0, 2>(xl, lo_al, seed, site, sweep, -1, nullptr, stage_tables, &n, &live)
                                : gg_spec_eval<NT, ADJ>(xl, lo_al, seed, site, sweep, -1, nullptr, stage_tables, &n, &live);
      if (live && n >= lo && n < hi && t % x.Kp == 0) {
        const int old = atomicExch(&cand[n], lab);
        asm volatile("" ::"v"(old) : "memory");
      }
    }
  }
  // last workgroup out: it sees every decision (read with agent-scope loads, not through a stale cache line)
  __syncthreads();
  if (t == 0) {
    const int tk = atomicAdd(&S->ticket, 1);
    s_last = tk == (int)gridDim.x - 1 ? 1 : 0;
    if (s_last) __hip_atomic_store(&S->ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_found = 0;
  }
  __syncthreads();
  if (!s_last) return;
  for (int i = t; i < GG_SHARD_M; i += 256) { trip[2 * i] = GG_SHARD_NONE; trip[2 * i + 1] = 0; }
  __syncthreads();
  // The first GG_SHARD_M deviations in position order, 4096 own nodes at a time.  Thread t looks at the nodes base + 256 i + t
  // (i < 16), all of its loads in flight together - the decisions are read with agent-scope loads, a memory-side round trip
  // each; a node's rank is the number of deviations found before its chunk, in the stripes before its own and in the lanes
  // before it in its stripe.
  constexpr int NS = 16;
  const int wv = t >> 6, ln = t & 63;
  for (int base = lo; base < hi; base += NS * 256) {
    int cv[NS], fv[NS], ov[NS], gv[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int nl = base + i * 256 + t;
      const bool in = nl < hi;
      cv[i] = in ? __hip_atomic_load(&cand[nl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      gv[i] = in ? gg_own_glob(own, nl) : 0;
      const int hv = in ? (int)H[gv[i] - wbase] : -1;
      ov[i] = in ? c[gv[i]] : 0;
      fv[i] = hv >= 0 ? (hv & 0xff) : ov[i];        // the label the node has for everybody behind it
    }
    unsigned dev = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const bool d = base + i * 256 + t < hi && cv[i] != fv[i];
      dev |= d ? (1u << i) : 0u;
      const unsigned long long bal = __ballot(d);
      if (ln == 0) s_scan[i * 4 + wv] = __popcll(bal);
    }
    __syncthreads();
    const int found0 = s_found;
    int tot = 0;
#pragma unroll
    for (int q = 0; q < NS * 4; ++q) tot += s_scan[q];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const bool d = (dev >> i) & 1u;
      const unsigned long long bal = __ballot(d);
      if (d) {
        int rk = found0 + __popcll(bal & ((1ull << ln) - 1ull));
        for (int q = 0; q < i * 4 + wv; ++q) rk += s_scan[q];
        if (rk < GG_SHARD_M) { trip[2 * rk] = gv[i]; trip[2 * rk + 1] = (ov[i] << 16) | (fv[i] << 8) | cv[i]; }
      }
    }
    __syncthreads();
    if (t == 0) s_found = found0 + tot;
    __syncthreads();
    if (found0 + tot >= GG_SHARD_M) break;          // (uniform)
  }
}

// all: the gathered reports [world][GG_SHARD_M][2] (node, old << 16 | hypothesised << 8 | drawn).
template <int NT, bool ADJ>
__global__ __launch_bounds__(256) void k_shard_apply(gg_node_ctx x, gg_own own, int W, int32_t* c, int32_t* cnt, int32_t* H,
                                                     int32_t* rs, gg_shard_state* S, int round, int ntot,
                                                     const int32_t* __restrict__ all, int world, int32_t* __restrict__ cntn,
                                                     int32_t* Gc0, int32_t* Gc1, long long* Gs0, long long* Gs1,
                                                     const uint32_t* XA, const uint32_t* XAT, const int32_t* XW, const int32_t* XWT,
                                                     int fold1) {
  constexpr int B = gg_tB(NT);
  constexpr int NSI = gg_tNSI(NT);
  extern __shared__ int s_mv[];                // [world][GG_SHARD_M][4]: node, from, to, old
  __shared__ int s_pstar, s_pe;
  const int t = (int)threadIdx.x;
  const int spos = S->st[round & 1][0], wbase = S->st[round & 1][1];
  const bool first = blockIdx.x == 0 && blockIdx.y == 0;
  if (spos >= ntot) {          // the scan is over: the state is handed on as it is
    if (first && t == 0) { S->st[(round + 1) & 1][0] = spos; S->st[(round + 1) & 1][1] = wbase; }
    return;
  }
  const int nrep = world * GG_SHARD_M;
  if (t == 0) s_pstar = GG_SHARD_NONE;
  __syncthreads();
  // the moves the reports imply: node f leaves the block its hypothesis put it in (or its old one) for the one it draws
  for (int e = t; e < nrep; e += 256) {
    const int f = all[2 * e], pk = all[2 * e + 1];
    s_mv[4 * e] = f; s_mv[4 * e + 1] = (pk >> 8) & 0xff; s_mv[4 * e + 2] = pk & 0xff; s_mv[4 * e + 3] = (pk >> 16) & 0xff;
    if (f != GG_SHARD_NONE) atomicMin(&s_pstar, f);
  }
  __syncthreads();
  for (int e = t; e < nrep; e += 256)
    if (s_mv[4 * e] != GG_SHARD_NONE && s_mv[4 * e] == s_pstar) s_pe = e;      // (a node is reported by its one owner)
  __syncthreads();
  const int bend = wbase + W < ntot ? wbase + W : ntot;
  const bool quiet = s_pstar == GG_SHARD_NONE;
  const int hi = quiet ? bend : s_pstar + 1;
  {
    // ---- fold, accept, reset: this workgroup's slice of the band (what it writes here is read by LATER launches only,
    // or by itself behind its own barrier)
    const int nwg = fold1 ? 1 : (int)(gridDim.x * gridDim.y), wg = fold1 ? (first ? 0 : 1) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int slice = ((W + nwg - 1) / nwg + 63) & ~63;
    const int b0 = wbase + wg * slice, b1 = b0 + slice < bend ? b0 + slice : bend;
    if (b0 < b1) {                             // (uniform over the workgroup)
      // No word of H is read in the launch that writes it: every report lies at or behind the earliest one, p*, so the
      // accepted stretch [spos, p*) holds hypotheses of EARLIER rounds only (read from H), p* itself is accepted from its
      // report, and the reports behind it amend H for later rounds.
      int mine = 0;
      const int a0 = b0 > spos ? b0 : spos, a1 = b1 < hi ? b1 : hi;
      for (int n = a0 + t; n < a1; n += 256) {
        int h = H[n - wbase];
        if (!quiet && n == hi - 1) {           // p*: what its report says
          const int e = s_pe, old = s_mv[4 * e + 3], nw = s_mv[4 * e + 2];
          h = (nw == old) ? -1 : ((old << 8) | nw);
        }
        if (h < 0) continue;
        c[n] = h & 0xff;
        atomicSub(&cnt[h >> 8], 1);
        atomicAdd(&cnt[h & 0xff], 1);
        ++mine;
      }
      // (the writes below touch words OTHER threads of this workgroup have just read - the two loops deal the band's nodes out
      // differently: without this barrier a reset could overtake the read of the very hypothesis it clears, and the node's
      // accepted change was lost on that rank only: found as an intermittent mismatch with three ranks and one wide band)
      __syncthreads();
      for (int e = t; e < nrep; e += 256) {
        const int f = s_mv[4 * e];
        if (f == GG_SHARD_NONE || f < b0 || f >= b1) continue;
        const int old = s_mv[4 * e + 3], nw = s_mv[4 * e + 2];
        H[f - wbase] = (nw == old) ? -1 : ((old << 8) | nw);
      }
      // (a final band starts the next one without hypotheses)
      if (quiet) for (int n = b0 + t; n < b1; n += 256) H[n - wbase] = -1;
      if (mine) atomicAdd(&rs[2], mine);
    }
    if (first && t == 0) {
      rs[GG_RS_ROUNDS] += 1;
      rs[GG_RS_SPOS] = hi;        // (for the host: the device reads the state copies)
      S->st[(round + 1) & 1][0] = hi;
      S->st[(round + 1) & 1][1] = quiet ? bend : wbase;
    }
  }
  // ---- the moves applied to the statistics (and the block sizes) of this rank's nodes BEHIND the moved node
  const int n = blockIdx.x * blockDim.x + t;
  if (n >= x.N) return;
  const int gn = gg_own_glob(own, n);
  const int Np = x.Np, K = x.K;
  for (int e = blockIdx.y; e < nrep; e += gridDim.y) {
    const int f = s_mv[4 * e], a = s_mv[4 * e + 1], b = s_mv[4 * e + 2];
    if (f == GG_SHARD_NONE || gn <= f) continue;
    atomicSub(&cntn[(size_t)n * K + a], 1);
    atomicAdd(&cntn[(size_t)n * K + b], 1);
    const size_t at = (size_t)f * Np + n;
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

// start of a sharded scan: no hypotheses, every own node sees the block sizes of the labels as they are
#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ void k_shard_init(const int32_t* __restrict__ cnt, int32_t* __restrict__ cntn, int32_t* __restrict__ H, int Nl, int K,
                             int32_t* __restrict__ rs, gg_shard_state* __restrict__ S) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { rs[GG_RS_ROUNDS] = 0; rs[GG_RS_SPOS] = 0; S->st[0][0] = 0; S->st[0][1] = 0; S->ticket = 0; }
  if (i < (size_t)Nl * K) cntn[i] = cnt[i % K];
  if (i < GG_SHARD_WMAX) H[i] = -1;
}
// sharded chain: the label sort with LOCAL node indices (-1 for a node another rank owns), what the pair tables' reduction walks
__global__ void k_perm_local(const int32_t* __restrict__ perm, int32_t* __restrict__ perm_loc, int N, gg_own own) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const int n = perm[r];
  perm_loc[r] = gg_own_has(own, n) ? gg_own_below(own, n) : -1;
}
#endif

// ------------------------------------------------------------------ log-likelihood from the K x K tables, on the device
// abstractions.py:111-115 (adjacency: sum_ab n1 log p + n0 log(1 - clip(p)) + diagonal) and weights.py:316-331 (weights:
// sum_ab [n g + h.s1 + Q.s2] + the self Gaussian on the diagonal edges), evaluated by one workgroup from the pair tables
// k_params has just reduced: the caller gets ONE double back instead of fetching tables and parameters (the demo loop
// `resample; log_probability`, examples/demo_model_comparison.py:69-72, pays this per iteration).  Pair terms are summed by a
// fixed-shape tree, so the value does not depend on scheduling.
struct gg_ll_args {
  int is_adj, K, B, NSI, NTAB, special, shift;
  const long long* tab;
  const int32_t* cnt;
  const double *p, *p_self, *mu, *sigma, *mu_self, *sigma_self;
  long long d1, d0;
  long long dS[GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2];
  int joint;            // GG_SBM_JOINT: Bernoulli terms of every block pair (diagonal included) on top of the Gaussian ones
  int wide;             // wide weights: B = 2 statistics (the limbs), scalar parameters (spec/gg_lp.h)
  int diag_zero;        // adjacency.py:525-526 with the special case: the diagonal has probability 0
  double* out;          // out[0] = log-likelihood, out[1] = 1.0 if a covariance was not positive definite
};

__device__ __forceinline__ double gg_ll_bern(double P, long long n1, long long n0) {
  const double Pc = P < 1e-64 ? 1e-64 : (P > 1 - 1e-64 ? 1 - 1e-64 : P);   // safe_P, abstractions.py:92-94
  double v = 0.0;
  if (n1) v += (double)n1 * log(P);
  if (n0) v += (double)n0 * log(1.0 - Pc);
  return v;
}

// (B: dimension of the STATISTICS; wide weights - the limbs of a one-dimensional weight, spec/gg_lp.h - have B = 2 statistics and
// scalar parameters: wide != 0)
__device__ __forceinline__ double gg_ll_gauss(const double* mu, const double* sig, int B, int NSI, double n, const long long* hl,
                                              double sc1, double sc2, bool* bad, int wide = 0) {
  if (n == 0) return 0.0;
  double e[GG_MAXNT];
  if (wide ? gg_gauss_table_wide(mu, sig, e) : gg_gauss_table(mu, sig, B, e)) { *bad = true; return 0.0; }
  double v = n * e[0];
  for (int s = 0; s < NSI; ++s) v += e[1 + s] * (gg_hilo_to_double(hl[2 * s], hl[2 * s + 1]) * (s < B ? sc1 : sc2));
  return v;
}

#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ __launch_bounds__(256) void k_loglik(gg_ll_args g) {
  __shared__ double s_part[256];
  __shared__ int s_bad;
  const int K = g.K, KK = K * K, tid = threadIdx.x;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  const double sc1 = gg_pow2i(-g.shift), sc2 = gg_pow2i(-2 * g.shift);
  double acc = 0.0;
  bool bad = false;
  for (int ab = tid; ab < KK; ab += 256) {       // (fixed assignment of pairs to lanes)
    const int a = ab / K, b = ab % K;
    if (g.is_adj) {
      const long long n1 = g.tab[ab];
      const long long pairs = (long long)g.cnt[a] * g.cnt[b] - ((a == b && g.special) ? g.cnt[a] : 0);
      acc += gg_ll_bern(g.p[ab], n1, pairs - n1);
    } else {
      const long long* e = g.tab + (size_t)ab * g.NTAB;
      const int Bx = g.wide ? 1 : g.B;      // (stride of the parameters)
      acc += gg_ll_gauss(g.mu + (size_t)ab * Bx, g.sigma + (size_t)ab * Bx * Bx, g.B, g.NSI, (double)e[0], e + 1, sc1, sc2, &bad, g.wide);
      if (g.joint) acc += gg_ll_bern(g.p[ab], e[0], (long long)g.cnt[a] * g.cnt[b] - e[0]);
    }
  }
  if (bad) atomicOr(&s_bad, 1);
  s_part[tid] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) s_part[tid] += s_part[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    double ll = s_part[0];
    if (g.is_adj) {
      if (g.diag_zero) ll += gg_ll_bern(0.0, g.d1, g.d0);
      else if (g.special) ll += gg_ll_bern(*g.p_self, g.d1, g.d0);
    } else if (g.special) {
      long long hl[2 * (GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)];
      for (int s = 0; s < g.NSI; ++s) { hl[2 * s] = g.dS[s] >> 32; hl[2 * s + 1] = g.dS[s] & 0xffffffffll; }
      bool b2 = false;
      ll += gg_ll_gauss(g.mu_self, g.sigma_self, g.B, g.NSI, (double)g.d1, hl, sc1, sc2, &b2, g.wide);
      if (b2) s_bad = 1;
    }
    g.out[0] = ll;
    g.out[1] = s_bad ? 1.0 : 0.0;
  }
}
#endif

// ------------------------------------------------------------------ predictive log-likelihood of a held-out node
// abstractions.py:157-182 (adjacency) / :313-357 (Gaussian weights): the log-likelihood of an (N+1)-th node's row and
// column under predictive parameters.  For the block models those parameters depend on a draw only through the new node's
// block c2 ~ Cat(m) (adjacency.py:562-573, weights.py:375-389; K = 1 kinds: not at all), so the M-draw Monte-Carlo average
// needs ll[k] for each block k once: workgroup k sums the 2N edge terms under "the new node is in block k" - labels and
// parameters never leave the device - and the caller averages exp(ll[c2_m]) over its M draws.
//   adjacency: ll[k] = sum_n a_row[n] log P[k,c_n] + (1 - a_row[n]) log(1 - P[k,c_n]) + (same with P[c_n,k] on the column)
//              + the self entry: P[k,k] (0 when self connections are not allowed; the K = 1 kinds use p_self)
//   weights:   ll[k] = sum_{n: a_row[n]} log N(w_row[n]; mu[k,c_n], Sigma[k,c_n]) + sum_{n: a_col[n]} log N(w_col[n]; mu[c_n,k], ..)
//              + the self entry under the self Gaussian (special case) or pair (k,k)
// P is clipped to [1e-64, 1 - 1e-64] as the reference does (abstractions.py:171-174).
struct gg_pred_args {
  int is_adj, K, B, N, special, allow_self, use_self_param;
  const int32_t* c;
  const double *p, *p_self, *mu, *sigma, *mu_self, *sigma_self;
  const uint8_t *a_row, *a_col;     // [N + 1]
  const double *w_row, *w_col;      // [N + 1][B]
  double* out;                      // [K]; out[K] = 1.0 if a covariance was not positive definite
};

#ifndef GG_ONLY_SWEEPQ   // (non-template kernels: defined once, in gg_model.hip)
__global__ __launch_bounds__(256) void k_pred_ll(gg_pred_args g) {
  extern __shared__ double s_tabs[];      // adjacency: [4][K] log terms; weights: [2][K][NT] Gaussian coefficient tables
  __shared__ double s_part[256];
  __shared__ int s_bad;
  const int K = g.K, B = g.B, N = g.N, k = blockIdx.x, tid = threadIdx.x;
  const int NT = 1 + B + B * (B + 1) / 2;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  auto clipP = [](double P) { return P < 1e-64 ? 1e-64 : (P > 1 - 1e-64 ? 1 - 1e-64 : P); };
  if (g.is_adj) {
    for (int j = tid; j < K; j += 256) {
      const double pr = clipP(g.p[(size_t)k * K + j]), pc = clipP(g.p[(size_t)j * K + k]);
      s_tabs[j] = log(pr); s_tabs[K + j] = log(1.0 - pr); s_tabs[2 * K + j] = log(pc); s_tabs[3 * K + j] = log(1.0 - pc);
    }
  } else {
    for (int j = tid; j < 2 * K; j += 256) {
      const int jj = j % K, dir = j / K;
      const size_t ab = dir == 0 ? (size_t)k * K + jj : (size_t)jj * K + k;
      if (gg_gauss_table(g.mu + ab * B, g.sigma + ab * B * B, B, s_tabs + (size_t)j * NT)) atomicOr(&s_bad, 1);
    }
  }
  __syncthreads();
  double acc = 0.0;
  for (int n = tid; n < N; n += 256) {
    const int j = g.c[n];
    if (g.is_adj) {
      acc += g.a_row[n] ? s_tabs[j] : s_tabs[K + j];
      acc += g.a_col[n] ? s_tabs[2 * K + j] : s_tabs[3 * K + j];
    } else {
      for (int dir = 0; dir < 2; ++dir) {
        if (!(dir == 0 ? g.a_row[n] : g.a_col[n])) continue;
        const double* w = (dir == 0 ? g.w_row : g.w_col) + (size_t)n * B;
        const double* e = s_tabs + ((size_t)dir * K + j) * NT;
        double v = e[0];
        int t = 1 + B;
        for (int r = 0; r < B; ++r) v += e[1 + r] * w[r];
        for (int r = 0; r < B; ++r)
          for (int q = r; q < B; ++q) v += e[t++] * w[r] * w[q];
        acc += v;
      }
    }
  }
  s_part[tid] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) s_part[tid] += s_part[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    double ll = s_part[0];
    // the new node's self entry (index N of the row; the column's copy is not counted twice: abstractions.py:162,170)
    if (g.is_adj) {
      double ps = g.use_self_param ? *g.p_self : g.p[(size_t)k * K + k];
      if (!g.allow_self) ps = 0.0;
      ps = clipP(ps);
      ll += g.a_row[N] ? log(ps) : log(1.0 - ps);
    } else if (g.a_row[N]) {
      double e[GG_MAXNT];
      const double* mu = g.use_self_param ? g.mu_self : g.mu + ((size_t)k * K + k) * B;
      const double* sg = g.use_self_param ? g.sigma_self : g.sigma + ((size_t)k * K + k) * B * B;
      if (gg_gauss_table(mu, sg, B, e)) s_bad = 1;
      const double* w = g.w_row + (size_t)N * B;
      double v = e[0];
      int t = 1 + B;
      for (int r = 0; r < B; ++r) v += e[1 + r] * w[r];
      for (int r = 0; r < B; ++r)
        for (int q = r; q < B; ++q) v += e[t++] * w[r] * w[q];
      ll += v;
    }
    g.out[k] = ll;
    if (s_bad) g.out[K] = 1.0;
  }
}
#endif
