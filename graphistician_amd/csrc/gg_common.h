// gg_common.h — shared declarations of the HIP engine (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/graphgibbs.h"
#include "spec/gg_philox.h"
#include "spec/gg_math.h"
#include "spec/gg_draws.h"
#include "spec/gg_lp.h"

#define GG_NO_EDGE INT32_MIN
#define GG_WAVE 64

extern thread_local std::string gg_err_text;
int gg_fail(int code, const char* fmt, ...);

#define GG_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      (void)hipGetLastError();   /* the error is reported here: do not leave it sticking to the next, unrelated call */ \
      return gg_fail(GG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    }                                                                                         \
  } while (0)

static inline int gg_round_up(int x, int m) { return (x + m - 1) / m * m; }

// Bytes a streaming pass may touch before its loads become non-temporal (beyond what the 256 MiB Infinity Cache keeps
// from sweep to sweep, see gg_stream_load).  GG_NT_BYTES overrides the threshold: 0 forces the non-temporal
// instantiations at any size, which is how the parity tests reach them with small networks.  Read at every launch.
static inline double gg_nt_threshold() {
  const char* e = getenv("GG_NT_BYTES");
  return e ? atof(e) : 200e6;
}

// Host-side validation of double-valued state handed over the C ABI: a NaN or an infinity would reach samplers whose
// rejection loops only end on comparisons that succeed (gamma, Polya-gamma, truncated normal): a hung GPU, not an error.
static inline long long gg_first_nonfinite(const double* v, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!(v[i] - v[i] == 0.0)) return (long long)i;
  return -1;
}
// Symmetric positive definite B x B (row-major)?  Plain Cholesky pivots.
static inline bool gg_is_spd(const double* S, int B) {
  double L[16] = {0};
  if (B > 4) return false;
  for (int j = 0; j < B; ++j) {
    double d = S[j * B + j];
    for (int k = 0; k < j; ++k) d -= L[j * B + k] * L[j * B + k];
    if (!(d > 0.0)) return false;
    const double dj = __builtin_sqrt(d);
    L[j * B + j] = dj;
    for (int i = j + 1; i < B; ++i) {
      {
        const double lo = S[i * B + j], up = S[j * B + i], df = lo > up ? lo - up : up - lo;
        const double mag = (lo < 0 ? -lo : lo) + (up < 0 ? -up : up);
        if (df > 1e-8 * mag) return false;   // (the kernels read the lower triangle; rounding-level asymmetry is let through)
      }
      double v = S[i * B + j];
      for (int k = 0; k < j; ++k) v -= L[i * B + k] * L[j * B + k];
      L[i * B + j] = v / dj;
    }
  }
  return true;
}
static inline int gg_ceil_log2(long long n) { int b = 0; while ((1LL << b) < n) ++b; return b; }
static inline int gg_qbits(int N) { int b = (62 - gg_ceil_log2(N)) / 2; return b > 23 ? 23 : b; }

// Number of integer weight statistics per (node, block): B first moments + B(B+1)/2 second moments.
static inline int gg_nsi(int B) { return B + B * (B + 1) / 2; }

// Which nodes a rank of a sharded chain owns.  lb < 0: the contiguous range [n0, n0 + Nl).  lb >= 0:
// blocks of 2^lb consecutive nodes dealt out round-robin - node n belongs to rank (n >> lb) % world - so that every band of the
// sequential scan spreads over all ranks; a rank's nodes keep their global order in its local numbering, local block i being
// global block i * world + rank.  An unsharded model owns everything: {0, N, -1, 0, 1}.
struct gg_own {
  int n0, Nl, lb, rank, world;
};
// global index of local node nl
GG_HD int gg_own_glob(const gg_own& o, int nl) {
  if (o.lb < 0) return o.n0 + nl;
  return ((((nl >> o.lb) * o.world) + o.rank) << o.lb) | (nl & ((1 << o.lb) - 1));
}
// number of owned nodes with a global index below n (= the local index of n when n is owned)
GG_HD int gg_own_below(const gg_own& o, int n) {
  int v;
  if (o.lb < 0) {
    v = n - o.n0;
    v = v < 0 ? 0 : v;
  } else {
    const int nb = n >> o.lb, q = nb / o.world, r = nb - q * o.world;
    v = (q << o.lb) + (r > o.rank ? (1 << o.lb) : (r == o.rank ? (n & ((1 << o.lb) - 1)) : 0));
  }
  return v > o.Nl ? o.Nl : v;
}
GG_HD bool gg_own_has(const gg_own& o, int n) {
  if (o.lb < 0) return n >= o.n0 && n < o.n0 + o.Nl;
  const int nb = n >> o.lb;
  return nb % o.world == o.rank;
}
// (the same ownership on both sides of a call?)
static inline bool gg_own_same(const gg_own& a, const gg_own& b) {
  if (a.Nl != b.Nl || (a.lb < 0) != (b.lb < 0)) return false;
  return a.lb < 0 ? a.n0 == b.n0 : (a.lb == b.lb && a.rank == b.rank && a.world == b.world);
}
// block-interleaved ownership of N nodes; returns false for arguments that describe none
static inline bool gg_own_blocks(int N, int block, int rank, int world, gg_own* o) {
  if (block < 8 || (block & (block - 1)) || world < 1 || rank < 0 || rank >= world) return false;
  o->n0 = 0; o->lb = 0; o->rank = rank; o->world = world;
  while ((1 << o->lb) < block) o->lb += 1;
  o->Nl = 0x7fffffff;
  o->Nl = gg_own_below(*o, N);
  if (world == 1) *o = gg_own{0, N, -1, 0, 1};     // (one rank owns everything, in order: the plain range)
  return true;
}

// Device-resident network.  Layout:
//   A, AT   : uint8  [N][Np]      off-diagonal adjacency and its transpose (diag = 0, pad = 0)
//   Ab, AbT : uint32 [N][Np/32]   the same, bit-packed (block adjacency models: 1/8 of the bytes per pass)
//   Wq, WqT : int32  [N][Np][B]   fixed-point weights, GG_NO_EDGE in component 0 where A == 0 / diag / pad
//   dA, dWq : the diagonal, kept apart because the reference special-cases self connections
#define GG_A_MISSING 2   // byte value of an unobserved adjacency entry (host input and the device byte matrices)

struct gg_data {
  int device;
  int N, B, Np;      // N rows (all nodes); Np = padded number of stored columns
  int n0, n1;        // stored columns = nodes [n0, n1) (0, N unless this is a shard; -1, -1 for a block-interleaved shard)
  gg_own own;        // the stored columns, in general: local column nl holds node gg_own_glob(own, nl)
  int has_w;
  int wide;          // 1: W is one-dimensional, stored as two 24-bit limbs in the B = 2 layout (gg_data_create_wide; spec/gg_lp.h)
  int shift;
  long long n_edges;
  // dynamic range of the stored weights (gg_data_quantisation): the largest |w| on an edge fixes ONE shift for the whole
  // matrix, so an edge whose |w| is far below it keeps few significant bits.  w_low = the MEDIAN of |w| over the edges'
  // components, low_bits = floor(log2(w_low 2^shift)) (what a typical weight keeps; < 0: not measured - device-generated data)
  double w_max, w_low;
  int low_bits;
  long long n_missing;   // entries of A given as GG_A_MISSING (unobserved: eigenmodel.py:121-127); 0 for every other network
  uint8_t *A, *AT, *dA;
  // the same off-diagonal adjacency and its transpose, one BIT per entry ([N][Np / 32] words, bit j of row i = A[i][j]):
  // what the block adjacency models read (their bulk pass counts by positional popcount, their scans test single bits);
  // the byte matrices stay for the kernels that want whole rows as bytes (eigenmodel, latent-distance) and for download
  uint32_t *Ab, *AbT;
  int32_t *Wq, *WqT, *dWq;
  // label-independent diagonal statistics (exact)
  long long d1;                              // number of self edges
  long long dS[GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2];  // sum q, sum q q^T over self edges (fits: N * 2^46)
  hipStream_t stream;
  // Lifetime: a model with sweeps enqueued on this network (gg_resample_async) holds a reference until it has been
  // synchronised; gg_data_destroy on a referenced handle only marks it, the last release frees it.  uid is unique per
  // created network (a new allocation at a recycled address is never mistaken for the old one).
  int refs, dead;
  unsigned long long uid;
  // Pair scheduling (gg_model.hip, launch_scan): the weight model on this network records an event behind each of its resident
  // scans; the adjacency model's scan of the same sweep waits for it - it then assembles on the empty chip just before the weight
  // model's next pass instead of meeting the weight model's scan.  A scheduling hint only: no result depends on it.
  // Events are kept per SWEEP INDEX (a ring of GG_PAIR_RING): the adjacency model's sweep s waits for the weight model's sweep s
  // or for nothing - never for "the latest", which from a second host thread could be the weight model's last sweep of a long
  // call (the whole adjacency chain would then run behind the whole weight chain).
#define GG_PAIR_RING 256
  struct pair_slot { hipEvent_t ev; uint32_t sweep; int set; };
  mutable pair_slot wscan[GG_PAIR_RING] = {}, wpass[GG_PAIR_RING] = {};
};
// (record on `st` for sweep `sweep` / the event of exactly that sweep, or null)
static inline int gg_pair_record(gg_data::pair_slot* ring, uint32_t sweep, hipStream_t st) {
  gg_data::pair_slot& s = ring[sweep % GG_PAIR_RING];
  s.set = 0;
  if (!s.ev) GG_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  GG_HIP(hipEventRecord(s.ev, st));
  s.sweep = sweep;
  s.set = 1;
  return GG_OK;
}
static inline hipEvent_t gg_pair_lookup(const gg_data::pair_slot* ring, uint32_t sweep) {
  const gg_data::pair_slot& s = ring[sweep % GG_PAIR_RING];
  return (s.set && s.sweep == sweep) ? s.ev : nullptr;
}

int gg_ensure_device(int device);
void gg_data_retain(const gg_data* d);
void gg_data_release(const gg_data* d);
