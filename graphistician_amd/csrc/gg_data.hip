// gg_data.hip — device-resident network: upload, fixed-point quantisation,
// stored transposes, synthetic generation.
//
// Replaces the (A, W) ndarray pair the reference hands to every call
// (networks.py:28,69; data conventions SURVEY.md §8b).  The Gibbs scan needs row
// n and column n of the data for every node, so both the matrix and its
// transpose are kept resident: every access in the sweep is then a contiguous
// row read.
#include <math.h>
#include <stdarg.h>

#include <algorithm>
#include <mutex>

#include "gg_common.h"

static std::mutex g_data_mu;   // guards gg_data::refs / dead (see gg_data_destroy)
static int data_refs(gg_data* d) { std::lock_guard<std::mutex> lk(g_data_mu); return d->refs; }

thread_local std::string gg_err_text;

static unsigned long long gg_next_data_uid() {
  static unsigned long long next = 0;
  return __atomic_add_fetch(&next, 1ULL, __ATOMIC_RELAXED);
}

int gg_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  gg_err_text = buf;
  return code;
}

int gg_ensure_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return gg_fail(GG_ERR_NO_DEVICE, "no HIP device available (%s); the engine has no CPU fallback",
                   e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return gg_fail(GG_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  GG_HIP(hipSetDevice(device));
  return GG_OK;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <typename T, int B>
__global__ void k_transpose(const T* __restrict__ src, T* __restrict__ dst, int N, int Np) {
  __shared__ T tile[32][33 * B];
  int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int i = by + r, j = bx + tx;
    for (int b = 0; b < B; ++b) {
      T v = (T)0;
      if (i < N && j < N) v = src[((size_t)i * Np + j) * B + b];
      tile[r][tx * B + b] = v;
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    int i = bx + r, j = by + tx;  // dst[i][j] = src[j][i]
    if (i < N && j < N)
      for (int b = 0; b < B; ++b) dst[((size_t)i * Np + j) * B + b] = tile[tx][r * B + b];
  }
}

__global__ void k_fill_i32(int32_t* p, size_t n, int B, int32_t v0) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (i % B == 0) ? v0 : 0;
}

// Synthetic planted SBM (B = 1): A_ij ~ Bern(p[c_i,c_j]) (diag: p_self), W_ij ~ N(mu, sigma)
// for edges; one Philox block per (i, j).  Mirrors adjacency.py:558-560 and weights.py:360-373.
__global__ void k_generate_sbm(uint8_t* A, int32_t* Wq, uint8_t* dA, int32_t* dWq, int N, int Np, int K,
                               const int32_t* __restrict__ c, const double* __restrict__ p, double p_self,
                               const double* __restrict__ mu, const double* __restrict__ sd, double mu_self,
                               double sd_self, uint64_t seed, double scale, int qmax) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= N) return;
  gg_u32x4 r = gg_philox4x32_10((uint32_t)j, (uint32_t)i, 0u, GG_SITE_DATA, (uint32_t)seed, (uint32_t)(seed >> 32));
  int ab = c[i] * K + c[j];
  double pe = (i == j) ? p_self : p[ab];
  double u = gg_u53(r.v[0], r.v[1]);
  int a = u < pe;
  // one normal by inversion-free Box-Muller-less route: polar needs a loop; use a second block if rejected
  gg_stream s;
  gg_stream_init(&s, seed, GG_SITE_DATA, 1u + (uint32_t)i, (uint32_t)j);
  double z = gg_normal(&s);
  double w = (i == j) ? (mu_self + sd_self * z) : (mu[ab] + sd[ab] * z);
  double qd = __builtin_rint(w * scale);
  if (qd > (double)qmax) qd = (double)qmax;
  if (qd < -(double)qmax) qd = -(double)qmax;
  int32_t q = a ? (int32_t)qd : GG_NO_EDGE;
  if (i == j) {
    dA[i] = (uint8_t)a;
    dWq[i] = q;
    A[(size_t)i * Np + j] = 0;
    Wq[(size_t)i * Np + j] = GG_NO_EDGE;
  } else {
    A[(size_t)i * Np + j] = (uint8_t)a;
    Wq[(size_t)i * Np + j] = q;
  }
}

// Column shard [n0, n1) of the same synthetic network: entry (i, nl) of the shard of A is the edge
// i -> n0+nl, of the shard of A^T the edge n0+nl -> i; both use the Philox counters of k_generate_sbm,
// so the union of the shards IS the network k_generate_sbm produces.
__device__ __forceinline__ int32_t gg_gen_edge(int i, int j, int K, const int32_t* c, const double* p, double p_self,
                                               const double* mu, const double* sd, double mu_self, double sd_self,
                                               uint64_t seed, double scale, int qmax, int* a_out) {
  gg_u32x4 r = gg_philox4x32_10((uint32_t)j, (uint32_t)i, 0u, GG_SITE_DATA, (uint32_t)seed, (uint32_t)(seed >> 32));
  const int ab = c[i] * K + c[j];
  const double pe = (i == j) ? p_self : p[ab];
  const int a = gg_u53(r.v[0], r.v[1]) < pe;
  gg_stream s;
  gg_stream_init(&s, seed, GG_SITE_DATA, 1u + (uint32_t)i, (uint32_t)j);
  const double z = gg_normal(&s);
  const double w = (i == j) ? (mu_self + sd_self * z) : (mu[ab] + sd[ab] * z);
  double qd = __builtin_rint(w * scale);
  if (qd > (double)qmax) qd = (double)qmax;
  if (qd < -(double)qmax) qd = -(double)qmax;
  *a_out = a;
  return a ? (int32_t)qd : GG_NO_EDGE;
}

__global__ void k_generate_sbm_shard(uint8_t* A, uint8_t* AT, int32_t* Wq, int32_t* WqT, uint8_t* dA, int32_t* dWq, int N,
                                     int Np, gg_own own, int Nl, int K, const int32_t* __restrict__ c,
                                     const double* __restrict__ p, double p_self, const double* __restrict__ mu,
                                     const double* __restrict__ sd, double mu_self, double sd_self, uint64_t seed,
                                     double scale, int qmax) {
  const int nl = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (nl >= Nl) return;
  const int n = gg_own_glob(own, nl);
  int a_in, a_out;
  const int32_t q_in = gg_gen_edge(i, n, K, c, p, p_self, mu, sd, mu_self, sd_self, seed, scale, qmax, &a_in);
  if (i == n) {
    dA[nl] = (uint8_t)a_in;
    dWq[nl] = q_in;
    return;   // the diagonal stays 0 / NO_EDGE in the shards
  }
  const int32_t q_out = gg_gen_edge(n, i, K, c, p, p_self, mu, sd, mu_self, sd_self, seed, scale, qmax, &a_out);
  A[(size_t)i * Np + nl] = (uint8_t)a_in;
  Wq[(size_t)i * Np + nl] = q_in;
  AT[(size_t)i * Np + nl] = (uint8_t)a_out;
  WqT[(size_t)i * Np + nl] = q_out;
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
static int alloc_data(gg_data* d) {
  size_t nA = (size_t)d->N * d->Np;
  GG_HIP(hipMalloc(&d->A, nA));
  GG_HIP(hipMalloc(&d->AT, nA));
  GG_HIP(hipMalloc(&d->dA, (size_t)d->N));
  GG_HIP(hipMalloc((void**)&d->Ab, nA / 8));
  GG_HIP(hipMalloc((void**)&d->AbT, nA / 8));
  GG_HIP(hipMemsetAsync(d->A, 0, nA, d->stream));
  GG_HIP(hipMemsetAsync(d->AT, 0, nA, d->stream));
  if (d->has_w) {
    size_t nW = nA * d->B;
    GG_HIP(hipMalloc(&d->Wq, nW * sizeof(int32_t)));
    GG_HIP(hipMalloc(&d->WqT, nW * sizeof(int32_t)));
    GG_HIP(hipMalloc(&d->dWq, (size_t)d->N * d->B * sizeof(int32_t)));
    k_fill_i32<<<2048, 256, 0, d->stream>>>(d->Wq, nW, d->B, GG_NO_EDGE);
    k_fill_i32<<<2048, 256, 0, d->stream>>>(d->WqT, nW, d->B, GG_NO_EDGE);
  }
  return GG_OK;
}

static void data_free(gg_data* d);

// Stream + device buffers of a fresh handle; on failure the handle (and whatever was allocated) is freed here.
static int data_open(gg_data* d) {
  int rc = GG_OK;
  if (hipStreamCreate(&d->stream) != hipSuccess) { d->stream = nullptr; rc = gg_fail(GG_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(hipGetLastError())); }
  if (!rc) rc = alloc_data(d);
  if (rc) data_free(d);
  return rc;
}

// 32 adjacency bytes -> one word of the bit matrix (bit j of the word = byte j != 0).
__global__ void k_pack_bits(const uint8_t* __restrict__ A, uint32_t* __restrict__ Ab, size_t nwords) {
  size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; w < nwords; w += stride) {
    const uint4* src = reinterpret_cast<const uint4*>(A + w * 32);
    const uint4 lo = src[0], hi = src[1];
    const unsigned v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned bits = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) bits |= (((v[q] >> (8 * b)) & 255u) ? 1u : 0u) << (4 * q + b);
    Ab[w] = bits;
  }
}

static int pack_bits(gg_data* d) {
  const size_t nwords = (size_t)d->N * d->Np / 32;
  k_pack_bits<<<2048, 256, 0, d->stream>>>(d->A, d->Ab, nwords);
  k_pack_bits<<<2048, 256, 0, d->stream>>>(d->AT, d->AbT, nwords);
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int make_transposes(gg_data* d) {
  dim3 blk(32, 8), grd((d->N + 31) / 32, (d->N + 31) / 32);
  k_transpose<uint8_t, 1><<<grd, blk, 0, d->stream>>>(d->A, d->AT, d->N, d->Np);
  if (d->has_w) {
    switch (d->B) {
      case 1: k_transpose<int32_t, 1><<<grd, blk, 0, d->stream>>>(d->Wq, d->WqT, d->N, d->Np); break;
      case 2: k_transpose<int32_t, 2><<<grd, blk, 0, d->stream>>>(d->Wq, d->WqT, d->N, d->Np); break;
      case 3: k_transpose<int32_t, 3><<<grd, blk, 0, d->stream>>>(d->Wq, d->WqT, d->N, d->Np); break;
      case 4: k_transpose<int32_t, 4><<<grd, blk, 0, d->stream>>>(d->Wq, d->WqT, d->N, d->Np); break;
    }
  }
  GG_HIP(hipGetLastError());
  return pack_bits(d);
}

static void diag_stats_host(gg_data* d, const uint8_t* dA, const int32_t* dWq) {
  d->d1 = 0;
  for (int s = 0; s < gg_nsi(GG_MAXB); ++s) d->dS[s] = 0;
  int B = d->B;
  for (int i = 0; i < d->N; ++i) {
    if (dA[i] != 1) continue;      // (0: no self edge; GG_A_MISSING: unobserved)
    d->d1 += 1;
    if (!d->has_w) continue;
    const int32_t* q = dWq + (size_t)i * B;
    for (int r = 0; r < B; ++r) d->dS[r] += q[r];
    int t = B;
    for (int r = 0; r < B; ++r)
      for (int s = r; s < B; ++s) d->dS[t++] += (long long)q[r] * (long long)q[s];
  }
}

// Host staging of a whole (A, W) into an allocated handle (gg_data_create, gg_data_update).
static int fill_full(gg_data* d, const uint8_t* A, const double* W) {
  const int N = d->N, B = d->B;
  const int Bin = d->wide ? 1 : B;      // dimension of W as the caller hands it (wide: one weight, stored as two limbs)
  // host staging: off-diagonal copy, diagonal, quantisation (same arithmetic as the
  // oracle's orc_quantise: shift = qbits - frexp-exponent of max |w| over edges).
  size_t nA = (size_t)N * d->Np;
  std::vector<uint8_t> hA(nA, 0), hdA(N, 0);
  long long nedges = 0, nmiss = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const uint8_t raw = A[(size_t)i * N + j];
      const uint8_t a = raw == GG_A_MISSING ? (uint8_t)GG_A_MISSING : (raw ? 1 : 0);
      nedges += a == 1;
      nmiss += a == GG_A_MISSING;
      if (i == j) hdA[i] = a; else hA[(size_t)i * d->Np + j] = a;
    }
  d->n_edges = nedges;
  d->n_missing = nmiss;
  if (nmiss && W) return gg_fail(GG_ERR_UNSUPPORTED, "unobserved adjacency entries (%lld of them) with weights: missing data is defined for the eigenmodels' adjacency only (eigenmodel.py:121-127)", nmiss);
  GG_HIP(hipMemcpyAsync(d->A, hA.data(), nA, hipMemcpyHostToDevice, d->stream));
  GG_HIP(hipMemcpyAsync(d->dA, hdA.data(), N, hipMemcpyHostToDevice, d->stream));
  std::vector<int32_t> hW, hdW;
  if (W) {
    double wmax = 0.0;
    for (size_t ij = 0; ij < (size_t)N * N; ++ij) {
      if (!A[ij]) continue;
      for (int b = 0; b < Bin; ++b) {
        double a = fabs(W[ij * Bin + b]);
        if (!(a <= 1.7e308)) return gg_fail(GG_ERR_NUMERIC, "non-finite weight on an edge");
        if (a > wmax) wmax = a;
      }
    }
    if (wmax == 0.0) wmax = 1.0;
    int e;
    frexp(wmax, &e);
    d->shift = gg_qbits(N) - e;
    {
      // Dynamic-range guard: the shift is set by the largest |w|; report what a TYPICAL weight keeps: the
      // median of |w| over the edges' non-zero components - the sums the sampler forms (w, w w^T) carry the absolute error, so it
      // is the data's scale that has to stand clear of the step, not its smallest entries (all of them up to 2^22 values, a strided sample beyond).
      std::vector<float> mags;
      size_t nval = 0;
      for (size_t ij = 0; ij < (size_t)N * N; ++ij) if (A[ij] == 1 || (A[ij] && A[ij] != GG_A_MISSING)) nval += Bin;
      const size_t stride = nval > ((size_t)1 << 22) ? nval / ((size_t)1 << 22) : 1;
      size_t seen = 0;
      for (size_t ij = 0; ij < (size_t)N * N; ++ij) {
        if (!A[ij] || A[ij] == GG_A_MISSING) continue;
        for (int b = 0; b < Bin; ++b, ++seen)
          if (seen % stride == 0) { const double a = fabs(W[ij * Bin + b]); if (a > 0.0) mags.push_back((float)a); }
      }
      d->w_max = wmax; d->w_low = 0.0; d->low_bits = -1;
      if (!mags.empty()) {
        const size_t k = mags.size() / 2;
        std::nth_element(mags.begin(), mags.begin() + k, mags.end());
        d->w_low = mags[k];
        int el;
        frexp(ldexp(d->w_low, d->shift), &el);      // w_low 2^shift in [2^(el-1), 2^el)
        d->low_bits = el - 1 < 0 ? 0 : el - 1;
        if (d->wide) d->low_bits += GG_WIDE_LIMB;    // (the second limb carries 24 bits more)
      }
    }
    hW.assign(nA * B, 0);
    hdW.assign((size_t)N * B, 0);
    for (size_t x = 0; x < nA; ++x) hW[x * B] = GG_NO_EDGE;
    for (int i = 0; i < N; ++i) {
      hdW[(size_t)i * B] = GG_NO_EDGE;
      for (int j = 0; j < N; ++j) {
        size_t ij = (size_t)i * N + j;
        if (!A[ij]) continue;
        int32_t* dst = (i == j) ? &hdW[(size_t)i * B] : &hW[((size_t)i * d->Np + j) * B];
        if (d->wide) {
          // two limbs: q0 = rint(w 2^s), q1 = rint((w - q0 2^-s) 2^(s+24)) - the remainder is exact in double, |q1| <= 2^23
          const double w = W[ij];
          const long long q0 = llrint(ldexp(w, d->shift));
          const double rem = w - ldexp((double)q0, -d->shift);
          dst[0] = (int32_t)q0;
          dst[1] = (int32_t)llrint(ldexp(rem, d->shift + GG_WIDE_LIMB));
        } else
          for (int b = 0; b < B; ++b) dst[b] = (int32_t)llrint(ldexp(W[ij * B + b], d->shift));
      }
    }
    GG_HIP(hipMemcpyAsync(d->Wq, hW.data(), hW.size() * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    GG_HIP(hipMemcpyAsync(d->dWq, hdW.data(), hdW.size() * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
  }
  diag_stats_host(d, hdA.data(), W ? hdW.data() : nullptr);
  int rc = make_transposes(d);
  if (rc) return rc;
  GG_HIP(hipStreamSynchronize(d->stream));
  return GG_OK;
}


extern "C" {

int gg_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) n = 0;
  *count = n;
  return GG_OK;
}

const char* gg_last_error(void) { return gg_err_text.c_str(); }
const char* gg_version(void) { return "graphgibbs-mi355x 0.4 (gfx950)"; }

int gg_data_create(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, int32_t device) {
  if (!out || !A || N <= 0) return gg_fail(GG_ERR_INVALID, "gg_data_create: bad arguments");
  if (W && (B < 1 || B > GG_MAXB)) return gg_fail(GG_ERR_UNSUPPORTED, "weight dimension B=%d not in 1..%d", B, GG_MAXB);
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  gg_data* d = new gg_data();
  memset(d, 0, sizeof(*d));
  d->uid = gg_next_data_uid();
  d->device = device;
  d->N = N;
  d->n0 = 0; d->n1 = N;
  d->own = gg_own{0, N, -1, 0, 1};
  d->B = W ? B : 1;
  d->Np = gg_round_up(N, 64);
  d->has_w = W ? 1 : 0;
  if ((rc = data_open(d))) return rc;
  rc = fill_full(d, A, W);
  if (rc) { gg_data_destroy(d); return rc; }
  *out = d;
  return GG_OK;
}

// Re-upload a new (A, W) of the same shape into an existing handle: the reference's loops that draw fresh data every
// iteration (Geweke tests: test/test_logistic_eigenmodel_geweke.py:35-47 `A = model.rvs()`; AIS / predictive checks
// with held-out entries) then pay one staging copy instead of allocating and freeing every device buffer.
int gg_data_update(gg_data* d, const uint8_t* A, const double* W) {
  if (!d || !A) return gg_fail(GG_ERR_INVALID, "gg_data_update: null argument");
  if (d->n0 != 0 || d->n1 != d->N) return gg_fail(GG_ERR_INVALID, "gg_data_update: not for column shards");
  if ((W != nullptr) != (d->has_w != 0)) return gg_fail(GG_ERR_INVALID, "gg_data_update: the handle was created %s weights", d->has_w ? "with" : "without");
  if (data_refs(d) > 0)
    return gg_fail(GG_ERR_INVALID, "gg_data_update: a model still has sweeps enqueued on this network (gg_sync it first)");
  int rc = gg_ensure_device(d->device);
  if (rc) return rc;
  d->uid = gg_next_data_uid();   // a different network from here on (statistics carried by a model belong to the old one)
  return fill_full(d, A, W);
}

static int create_sbm_impl(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                           const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                           int32_t device, gg_own own);

int gg_data_create_sbm(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                       const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                       int32_t device) {
  return create_sbm_impl(out, N, K, c, p, p_self, mu, sigma, mu_self, sigma_self, seed, device, gg_own{0, N, -1, 0, 1});
}

int gg_data_create_sbm_shard(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                             const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                             int32_t n0, int32_t n1, int32_t device) {
  if (n0 < 0 || n1 > N || n0 >= n1) return gg_fail(GG_ERR_INVALID, "gg_data_create_sbm_shard: bad node range");
  return create_sbm_impl(out, N, K, c, p, p_self, mu, sigma, mu_self, sigma_self, seed, device, gg_own{n0, n1 - n0, -1, 0, 1});
}

int gg_data_create_sbm_shard_blocks(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                                    const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                                    int32_t block, int32_t rank, int32_t world, int32_t device) {
  gg_own own;
  if (N <= 0 || !gg_own_blocks(N, block, rank, world, &own))
    return gg_fail(GG_ERR_INVALID, "gg_data_create_sbm_shard_blocks: block %d (a power of two >= 8), rank %d of %d", block, rank, world);
  if (own.Nl < 1) return gg_fail(GG_ERR_INVALID, "gg_data_create_sbm_shard_blocks: rank %d of %d owns no node of %d in blocks of %d", rank, world, N, block);
  return create_sbm_impl(out, N, K, c, p, p_self, mu, sigma, mu_self, sigma_self, seed, device, own);
}

static int create_sbm_impl(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                           const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                           int32_t device, gg_own own) {
  if (!out || N <= 0 || K <= 0 || !c || !p || !mu || !sigma) return gg_fail(GG_ERR_INVALID, "gg_data_create_sbm: bad arguments");
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  const bool shard = !(own.lb < 0 && own.n0 == 0 && own.Nl == N);
  const int Nl = own.Nl;
  gg_data* d = new gg_data();
  memset(d, 0, sizeof(*d));
  d->uid = gg_next_data_uid();
  d->device = device; d->N = N; d->B = 1; d->Np = gg_round_up(Nl, 64); d->has_w = 1;
  d->own = own;
  if (own.lb < 0) { d->n0 = own.n0; d->n1 = own.n0 + Nl; } else { d->n0 = d->n1 = -1; }
  if ((rc = data_open(d))) return rc;
  // fixed-point scale from a bound on |w|: max |mu| + 8 sd (values beyond are clamped; probability < 1e-15 per edge)
  double wb = fabs(mu_self) + 8.0 * sqrt(sigma_self);
  std::vector<double> sd((size_t)K * K);
  for (int ab = 0; ab < K * K; ++ab) {
    sd[ab] = sqrt(sigma[ab]);
    double v = fabs(mu[ab]) + 8.0 * sd[ab];
    if (v > wb) wb = v;
  }
  int e;
  frexp(wb, &e);
  int qb = gg_qbits(N);
  d->shift = qb - e;
  int32_t* dc; double *dp, *dmu, *dsd;
  GG_HIP(hipMalloc(&dc, (size_t)N * 4));
  GG_HIP(hipMalloc(&dp, (size_t)K * K * 8));
  GG_HIP(hipMalloc(&dmu, (size_t)K * K * 8));
  GG_HIP(hipMalloc(&dsd, (size_t)K * K * 8));
  GG_HIP(hipMemcpyAsync(dc, c, (size_t)N * 4, hipMemcpyHostToDevice, d->stream));
  GG_HIP(hipMemcpyAsync(dp, p, (size_t)K * K * 8, hipMemcpyHostToDevice, d->stream));
  GG_HIP(hipMemcpyAsync(dmu, mu, (size_t)K * K * 8, hipMemcpyHostToDevice, d->stream));
  GG_HIP(hipMemcpyAsync(dsd, sd.data(), (size_t)K * K * 8, hipMemcpyHostToDevice, d->stream));
  if (!shard) {
    dim3 grd((N + 255) / 256, N);
    k_generate_sbm<<<grd, 256, 0, d->stream>>>(d->A, d->Wq, d->dA, d->dWq, N, d->Np, K, dc, dp, p_self, dmu, dsd, mu_self,
                                               sqrt(sigma_self), seed, ldexp(1.0, d->shift), 1 << qb);
    GG_HIP(hipGetLastError());
    rc = make_transposes(d);
    if (rc) return rc;
  } else {
    dim3 grd((Nl + 255) / 256, N);
    k_generate_sbm_shard<<<grd, 256, 0, d->stream>>>(d->A, d->AT, d->Wq, d->WqT, d->dA, d->dWq, N, d->Np, own, Nl, K, dc, dp,
                                                     p_self, dmu, dsd, mu_self, sqrt(sigma_self), seed,
                                                     ldexp(1.0, d->shift), 1 << qb);
    GG_HIP(hipGetLastError());
    if ((rc = pack_bits(d))) return rc;
  }
  std::vector<uint8_t> hdA(Nl);
  std::vector<int32_t> hdW(Nl);
  GG_HIP(hipMemcpyAsync(hdA.data(), d->dA, Nl, hipMemcpyDeviceToHost, d->stream));
  GG_HIP(hipMemcpyAsync(hdW.data(), d->dWq, (size_t)Nl * 4, hipMemcpyDeviceToHost, d->stream));
  GG_HIP(hipStreamSynchronize(d->stream));
  {
    const int Nsave = d->N;
    d->N = Nl;
    diag_stats_host(d, hdA.data(), hdW.data());
    d->N = Nsave;
  }
  d->n_edges = -1;
  hipFree(dc); hipFree(dp); hipFree(dmu); hipFree(dsd);
  *out = d;
  return GG_OK;
}

static int create_shard_impl(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, gg_own own, int32_t device);

int gg_data_create_shard(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, int32_t n0, int32_t n1,
                         int32_t device) {
  if (!out || !A || N <= 0 || n0 < 0 || n1 > N || n0 >= n1) return gg_fail(GG_ERR_INVALID, "gg_data_create_shard: bad arguments");
  return create_shard_impl(out, N, B, A, W, gg_own{n0, n1 - n0, -1, 0, 1}, device);
}

int gg_data_create_shard_blocks(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, int32_t block, int32_t rank,
                                int32_t world, int32_t device) {
  gg_own own;
  if (!out || !A || N <= 0 || !gg_own_blocks(N, block, rank, world, &own))
    return gg_fail(GG_ERR_INVALID, "gg_data_create_shard_blocks: block %d (a power of two >= 8), rank %d of %d", block, rank, world);
  if (own.Nl < 1) return gg_fail(GG_ERR_INVALID, "gg_data_create_shard_blocks: rank %d of %d owns no node of %d in blocks of %d", rank, world, N, block);
  return create_shard_impl(out, N, B, A, W, own, device);
}

static int create_shard_impl(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, gg_own own, int32_t device) {
  if (W && (B < 1 || B > GG_MAXB)) return gg_fail(GG_ERR_UNSUPPORTED, "weight dimension B=%d not in 1..%d", B, GG_MAXB);
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  gg_data* d = new gg_data();
  memset(d, 0, sizeof(*d));
  d->uid = gg_next_data_uid();
  d->device = device;
  d->N = N;
  d->own = own;
  if (own.lb < 0) { d->n0 = own.n0; d->n1 = own.n0 + own.Nl; } else { d->n0 = d->n1 = -1; }
  d->B = W ? B : 1;
  const int Nl = own.Nl;
  d->Np = gg_round_up(Nl, 64);
  d->has_w = W ? 1 : 0;
  if ((rc = data_open(d))) return rc;
  const size_t nA = (size_t)N * d->Np;
  // column shard of A (in-edges of the owned nodes) and of A^T (their out-edges), diagonal apart
  std::vector<uint8_t> hA(nA, 0), hAT(nA, 0), hdA(Nl, 0);
  long long nedges = 0;
  for (int i = 0; i < N; ++i)
    for (int nl = 0; nl < Nl; ++nl) {
      const int n = gg_own_glob(own, nl);
      const uint8_t a_in = A[(size_t)i * N + n] ? 1 : 0;    // i -> n
      const uint8_t a_out = A[(size_t)n * N + i] ? 1 : 0;   // n -> i
      if (i == n) { hdA[nl] = a_in; nedges += a_in; continue; }
      hA[(size_t)i * d->Np + nl] = a_in;
      hAT[(size_t)i * d->Np + nl] = a_out;
      nedges += a_in;
    }
  d->n_edges = nedges;
  GG_HIP(hipMemcpyAsync(d->A, hA.data(), nA, hipMemcpyHostToDevice, d->stream));
  GG_HIP(hipMemcpyAsync(d->AT, hAT.data(), nA, hipMemcpyHostToDevice, d->stream));
  GG_HIP(hipMemcpyAsync(d->dA, hdA.data(), Nl, hipMemcpyHostToDevice, d->stream));
  { int rcp = pack_bits(d); if (rcp) { gg_data_destroy(d); return rcp; } }
  std::vector<int32_t> hW, hWT, hdW;
  if (W) {
    double wmax = 0.0;   // global scale: identical on every rank
    for (size_t ij = 0; ij < (size_t)N * N; ++ij) {
      if (!A[ij]) continue;
      for (int b = 0; b < B; ++b) {
        double a = fabs(W[ij * B + b]);
        if (!(a <= 1.7e308)) { gg_data_destroy(d); return gg_fail(GG_ERR_NUMERIC, "non-finite weight on an edge"); }
        if (a > wmax) wmax = a;
      }
    }
    if (wmax == 0.0) wmax = 1.0;
    int e;
    frexp(wmax, &e);
    d->shift = gg_qbits(N) - e;
    {
      // Dynamic-range guard: the shift is set by the largest |w|; report what a TYPICAL weight keeps: the
      // median of |w| over the edges' non-zero components - the sums the sampler forms (w, w w^T) carry the absolute error, so it
      // is the data's scale that has to stand clear of the step, not its smallest entries (all of them up to 2^22 values, a strided sample beyond).
      std::vector<float> mags;
      size_t nval = 0;
      for (size_t ij = 0; ij < (size_t)N * N; ++ij) if (A[ij] == 1 || (A[ij] && A[ij] != GG_A_MISSING)) nval += B;
      const size_t stride = nval > ((size_t)1 << 22) ? nval / ((size_t)1 << 22) : 1;
      size_t seen = 0;
      for (size_t ij = 0; ij < (size_t)N * N; ++ij) {
        if (!A[ij] || A[ij] == GG_A_MISSING) continue;
        for (int b = 0; b < B; ++b, ++seen)
          if (seen % stride == 0) { const double a = fabs(W[ij * B + b]); if (a > 0.0) mags.push_back((float)a); }
      }
      d->w_max = wmax; d->w_low = 0.0; d->low_bits = -1;
      if (!mags.empty()) {
        const size_t k = mags.size() / 2;
        std::nth_element(mags.begin(), mags.begin() + k, mags.end());
        d->w_low = mags[k];
        int el;
        frexp(ldexp(d->w_low, d->shift), &el);      // w_low 2^shift in [2^(el-1), 2^el)
        d->low_bits = el - 1 < 0 ? 0 : el - 1;
      }
    }
    hW.assign(nA * B, 0); hWT.assign(nA * B, 0); hdW.assign((size_t)Nl * B, 0);
    for (size_t x = 0; x < nA; ++x) { hW[x * B] = GG_NO_EDGE; hWT[x * B] = GG_NO_EDGE; }
    for (int nl = 0; nl < Nl; ++nl) hdW[(size_t)nl * B] = GG_NO_EDGE;
    for (int i = 0; i < N; ++i)
      for (int nl = 0; nl < Nl; ++nl) {
        const int n = gg_own_glob(own, nl);
        const size_t in = (size_t)i * N + n, on = (size_t)n * N + i;
        if (i == n) {
          if (A[in]) for (int b = 0; b < B; ++b) hdW[(size_t)nl * B + b] = (int32_t)llrint(ldexp(W[in * B + b], d->shift));
          continue;
        }
        if (A[in]) for (int b = 0; b < B; ++b) hW[((size_t)i * d->Np + nl) * B + b] = (int32_t)llrint(ldexp(W[in * B + b], d->shift));
        if (A[on]) for (int b = 0; b < B; ++b) hWT[((size_t)i * d->Np + nl) * B + b] = (int32_t)llrint(ldexp(W[on * B + b], d->shift));
      }
    GG_HIP(hipMemcpyAsync(d->Wq, hW.data(), hW.size() * 4, hipMemcpyHostToDevice, d->stream));
    GG_HIP(hipMemcpyAsync(d->WqT, hWT.data(), hWT.size() * 4, hipMemcpyHostToDevice, d->stream));
    GG_HIP(hipMemcpyAsync(d->dWq, hdW.data(), hdW.size() * 4, hipMemcpyHostToDevice, d->stream));
  }
  {  // diagonal statistics of the owned nodes (the driver all-reduces them)
    const int Nsave = d->N;
    d->N = Nl;
    diag_stats_host(d, hdA.data(), W ? hdW.data() : nullptr);
    d->N = Nsave;
  }
  GG_HIP(hipStreamSynchronize(d->stream));
  *out = d;
  return GG_OK;
}

// Who frees a handle that is destroyed while sweeps are still enqueued on it is decided under one lock (the
// lock-free rendezvous let the loser read a handle the winner had deleted): destroy with references outstanding only marks
// it, and the release that takes the count to zero frees it.  These are per-call operations, not per sweep.

int gg_data_create_wide(gg_data** out, int32_t N, const uint8_t* A, const double* W, int32_t device) {
  if (!out || !A || !W || N <= 0) return gg_fail(GG_ERR_INVALID, "gg_data_create_wide: bad arguments (W is required)");
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  gg_data* d = new gg_data();
  memset(d, 0, sizeof(*d));
  d->uid = gg_next_data_uid();
  d->device = device; d->N = N; d->B = 2; d->Np = gg_round_up(N, 64); d->has_w = 1; d->wide = 1;
  d->n0 = 0; d->n1 = N;
  d->own = gg_own{0, N, -1, 0, 1};
  if ((rc = data_open(d))) return rc;
  rc = fill_full(d, A, W);
  if (rc) { gg_data_destroy(d); return rc; }
  *out = d;
  return GG_OK;
}

int gg_data_destroy(gg_data* d) {
  if (!d) return GG_OK;
  {
    std::lock_guard<std::mutex> lk(g_data_mu);
    if (d->refs > 0) { d->dead = 1; return GG_OK; }   // sweeps are still enqueued on it: freed by the last release
  }
  data_free(d);
  return GG_OK;
}

}  // extern "C"

void gg_data_retain(const gg_data* d) {
  if (!d) return;
  std::lock_guard<std::mutex> lk(g_data_mu);
  const_cast<gg_data*>(d)->refs += 1;
}
void gg_data_release(const gg_data* d) {
  if (!d) return;
  gg_data* w = const_cast<gg_data*>(d);
  bool last;
  {
    std::lock_guard<std::mutex> lk(g_data_mu);
    w->refs -= 1;
    last = w->refs == 0 && w->dead;
  }
  if (last) data_free(w);
}

static void data_free(gg_data* d) {
  hipSetDevice(d->device);
  if (d->A) hipFree(d->A);
  if (d->AT) hipFree(d->AT);
  if (d->Ab) hipFree(d->Ab);
  if (d->AbT) hipFree(d->AbT);
  if (d->dA) hipFree(d->dA);
  if (d->Wq) hipFree(d->Wq);
  if (d->WqT) hipFree(d->WqT);
  if (d->dWq) hipFree(d->dWq);
  if (d->stream) hipStreamDestroy(d->stream);
  for (int i = 0; i < GG_PAIR_RING; ++i) {
    if (d->wscan[i].ev) hipEventDestroy(d->wscan[i].ev);
    if (d->wpass[i].ev) hipEventDestroy(d->wpass[i].ev);
  }
  delete d;
}

extern "C" {

int gg_data_info(const gg_data* d, int32_t* N, int32_t* B, int32_t* shift, int64_t* n_edges) {
  if (!d) return gg_fail(GG_ERR_INVALID, "null data");
  if (N) *N = d->N;
  if (B) *B = d->wide ? 1 : d->B;      // (the weights' dimension as the caller knows it)
  if (shift) *shift = d->shift;
  if (n_edges) *n_edges = d->n_edges;
  return GG_OK;
}

int gg_data_quantisation(const gg_data* d, double* w_max, double* w_low, int32_t* low_bits, int32_t* shift) {
  if (!d) return gg_fail(GG_ERR_INVALID, "null data");
  if (w_max) *w_max = d->w_max;
  if (w_low) *w_low = d->w_low;
  if (low_bits) *low_bits = (d->has_w && d->w_max > 0.0) ? d->low_bits : -1;     // (w_max == 0: never measured)
  if (shift) *shift = d->shift;
  return GG_OK;
}

int gg_data_download(const gg_data* d, uint8_t* A, double* W) {
  if (!d || !A) return gg_fail(GG_ERR_INVALID, "gg_data_download: bad arguments");
  if (d->n0 != 0 || d->n1 != d->N) return gg_fail(GG_ERR_INVALID, "gg_data_download: not for column shards");
  GG_HIP(hipSetDevice(d->device));
  int N = d->N, Np = d->Np, B = d->B;
  std::vector<uint8_t> hA((size_t)N * Np), hdA(N);
  GG_HIP(hipMemcpy(hA.data(), d->A, hA.size(), hipMemcpyDeviceToHost));
  GG_HIP(hipMemcpy(hdA.data(), d->dA, N, hipMemcpyDeviceToHost));
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) A[(size_t)i * N + j] = (i == j) ? hdA[i] : hA[(size_t)i * Np + j];
  if (W && d->has_w) {
    std::vector<int32_t> hW((size_t)N * Np * B), hdW((size_t)N * B);
    GG_HIP(hipMemcpy(hW.data(), d->Wq, hW.size() * 4, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(hdW.data(), d->dWq, hdW.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const int32_t* q = (i == j) ? &hdW[(size_t)i * B] : &hW[((size_t)i * Np + j) * B];
        if (d->wide)      // (N, N, 1): the two limbs put together again - exact, 48 bits fit a double
          W[(size_t)i * N + j] = (q[0] == GG_NO_EDGE) ? 0.0 : ldexp((double)q[0], -d->shift) + ldexp((double)q[1], -d->shift - GG_WIDE_LIMB);
        else
          for (int b = 0; b < B; ++b)
            W[((size_t)i * N + j) * B + b] = (q[0] == GG_NO_EDGE) ? 0.0 : ldexp((double)q[b], -d->shift);
      }
  }
  return GG_OK;
}

}  // extern "C"
