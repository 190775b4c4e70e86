// gg_common.h — shared declarations of the HIP engine (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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
static inline int gg_ceil_log2(long long n) { int b = 0; while ((1LL << b) < n) ++b; return b; }
static inline int gg_qbits(int N) { int b = (62 - gg_ceil_log2(N)) / 2; return b > 23 ? 23 : b; }

// Number of integer weight statistics per (node, block): B first moments + B(B+1)/2 second moments.
static inline int gg_nsi(int B) { return B + B * (B + 1) / 2; }

// Device-resident network.  Layout:
//   A, AT   : uint8  [N][Np]      off-diagonal adjacency and its transpose (diag = 0, pad = 0)
//   Wq, WqT : int32  [N][Np][B]   fixed-point weights, GG_NO_EDGE in component 0 where A == 0 / diag / pad
//   dA, dWq : the diagonal, kept apart because the reference special-cases self connections
struct gg_data {
  int device;
  int N, B, Np;      // N rows (all nodes); Np = padded number of stored columns
  int n0, n1;        // stored columns = nodes [n0, n1) (0, N unless this is a shard)
  int has_w;
  int shift;
  long long n_edges;
  uint8_t *A, *AT, *dA;
  int32_t *Wq, *WqT, *dWq;
  // label-independent diagonal statistics (exact)
  long long d1;                              // number of self edges
  long long dS[GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2];  // sum q, sum q q^T over self edges (fits: N * 2^46)
  hipStream_t stream;
};

int gg_ensure_device(int device);
