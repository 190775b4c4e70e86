// Probe: what one step of k_eig_chain's scan costs a wave, piece by piece.  One workgroup of 8 waves (DM = 8 lanes per node,
// node i in wave i % 8), every vector "published" beforehand (nobody waits for anybody), 64 steps; FLAGS switch pieces off.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

enum { F_POLL = 1, F_UREAD = 2, F_SV = 4, F_Q = 8, F_SUMS = 16, F_RCP = 32, F_PUB = 64, F_PUPD = 128, F_ALL = 255 };

template <int FLAGS, int WAVES>
__global__ __launch_bounds__(512) void k_step(long long* out, int steps) {
  constexpr int DM = 8, T = 64, TS = 65;
  extern __shared__ __align__(16) double lds[];
  double* s_S = lds; double* s_V = s_S + T * TS; double* s_u = s_V + T * TS; double* s_q = s_u + T * DM; double* s_Fn = s_q + 512;
  int* s_pub = reinterpret_cast<int*>(s_Fn + T * DM);
  const int tid = threadIdx.x;
  for (int e = tid; e < T * TS; e += blockDim.x) { s_S[e] = 0.01 + 1e-5 * e; s_V[e] = 0.02 - 1e-5 * e; }
  for (int e = tid; e < T * DM; e += blockDim.x) s_u[e] = 0.05 + 1e-4 * e;
  if (tid == 0) *s_pub = T + 1;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = (tid & 63) / DM, j = tid % DM;
  const int i = grp * WAVES + wave, gbase = (tid / DM) * DM;
  double Prow[DM], rj = 0.1 + 0.01 * j;
#pragma unroll
  for (int m = 0; m < DM; ++m) Prow[m] = (m == j ? 1.0 : 0.01) + 1e-4 * i;
  const double lamj = 0.7;
  const long long t0 = clock64();
  for (int rep = 0; rep < steps / T; ++rep)
  for (int k = 1; k < T; ++k) {
    const bool closer = (k % WAVES) == wave;
    double u[DM], uj = 0.3, s = 0.01, v = 0.02, qj = 0.1, c = 0.01;
    if (FLAGS & F_POLL) {
      for (int spin = 0;; ++spin) {
        asm volatile("" ::: "memory");
        if (*s_pub >= k) break;
        if (!closer) __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      lds_wait();
    }
    if (FLAGS & F_UREAD) {
#pragma unroll
      for (int m = 0; m < DM; ++m) u[m] = s_u[(k - 1) * DM + m];
      uj = s_u[(k - 1) * DM + j];
    } else {
#pragma unroll
      for (int m = 0; m < DM; ++m) u[m] = 0.05 + 0.01 * m + rj * 1e-9;
    }
    if (FLAGS & F_SV) { s = s_S[i * TS + k - 1]; v = s_V[i * TS + k - 1]; }
    if (FLAGS & F_Q) {
      double qa = 0.0, qb = 0.0;
#pragma unroll
      for (int m = 0; m < DM / 2; ++m) { qa = __builtin_fma(Prow[m], u[m], qa); qb = __builtin_fma(Prow[DM / 2 + m], u[DM / 2 + m], qb); }
      qj = qa + qb;
    } else qj = u[0] * Prow[0];
    double tq = uj * qj, d = uj * rj;
    if (FLAGS & F_SUMS) {
      tq += dpp_f64<0xB1>(tq); d += dpp_f64<0xB1>(d);
      tq += dpp_f64<0x4E>(tq); d += dpp_f64<0x4E>(d);
      tq += dpp_f64<0x141>(tq); d += dpp_f64<0x141>(d);
    }
    double rc = 0.9;
    if (FLAGS & F_RCP) {
      const double den = __builtin_fma(s, tq, 1.0);
      rc = __builtin_amdgcn_rcp(den);
      rc = rc * __builtin_fma(-den, rc, 2.0);
      rc = rc * __builtin_fma(-den, rc, 2.0);
    } else rc = 1.0 - s * tq;
    c = s * rc;
    rj = __builtin_fma(__builtin_fma(-s, d, v) * rc, qj, rj);
    if ((FLAGS & F_PUB) && closer) {
      if (i == k) { s_Fn[k * DM + j] = rj; s_q[256 + (k & 31) * DM + j] = rj * lamj; }
      lds_wait();
      if (i == k && j == 0) s_Fn[T * DM - 1] = (double)k;
      asm volatile("" ::: "memory");
    }
    if (FLAGS & F_PUPD) {
      s_q[tid & 255] = qj;    // (8 waves share 256 slots here: the probe only times the traffic)
      lds_wait();
      const double cq = c * qj;
#pragma unroll
      for (int m = 0; m < DM; ++m) Prow[m] = __builtin_fma(-cq, s_q[(gbase & 255) + m], Prow[m]);
      lds_wait();
    } else {
      const double cq = c * qj;
#pragma unroll
      for (int m = 0; m < DM; ++m) Prow[m] = __builtin_fma(-cq, u[m], Prow[m]);
    }
  }
  const long long t1 = clock64();
  double acc = rj;
#pragma unroll
  for (int m = 0; m < DM; ++m) acc += Prow[m];
  if ((tid & 63) == 0) out[wave] = t1 - t0;
  if (acc == 1.2345) out[63] = 1;
}

// The restructured step: ONE LDS batch per step - the published vector (its last component doubles as the flag), this lane's
// own component, the pair's coefficients (prefetched: they are static) and the group's q of the PREVIOUS step, whose rank-one
// update of P is applied first; the new q is written at the end of the step and read back by the next batch.
template <int WAVES, int DEFER, int SENT>
__global__ __launch_bounds__(512) void k_step2(long long* out, int steps) {
  constexpr int DM = 8, T = 64, TS = 65;
  extern __shared__ __align__(16) double lds[];
  double* s_S = lds; double* s_V = s_S + T * TS; double* s_u = s_V + T * TS; double* s_q = s_u + T * DM; double* s_Fn = s_q + 512;
  int* s_pub = reinterpret_cast<int*>(s_Fn + T * DM);
  const int tid = threadIdx.x;
  for (int e = tid; e < T * TS; e += blockDim.x) { s_S[e] = 0.01 + 1e-5 * e; s_V[e] = 0.02 - 1e-5 * e; }
  for (int e = tid; e < T * DM; e += blockDim.x) s_u[e] = 0.05 + 1e-4 * e;
  for (int e = tid; e < 512; e += blockDim.x) s_q[e] = 0.01;
  if (tid == 0) *s_pub = T + 1;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), grp = (tid & 63) / DM, j = tid % DM;
  const int i = grp * WAVES + wave, gbase = (tid / DM) * DM;
  double Prow[DM], rj = 0.1 + 0.01 * j, cq_prev = 0.0;
#pragma unroll
  for (int m = 0; m < DM; ++m) Prow[m] = (m == j ? 1.0 : 0.01) + 1e-4 * i;
  const double lamj = 0.7;
  double s_next = s_S[i * TS], v_next = s_V[i * TS];
  const long long t0 = clock64();
  for (int rep = 0; rep < steps / T; ++rep)
  for (int k = 1; k < T; ++k) {
    const bool closer = (k % WAVES) == wave;
    double u[DM], uj, qg[DM];
    const double s = s_next, v = v_next;
    if (!SENT) {
      for (int spin = 0;; ++spin) {
        asm volatile("" ::: "memory");
        if (*s_pub >= k) break;
        if (!closer) __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
      lds_wait();
    }
    for (int spin = 0;; ++spin) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int m = 0; m < DM; ++m) u[m] = s_u[(k - 1) * DM + m];
      uj = s_u[(k - 1) * DM + j];
      if (DEFER) {
#pragma unroll
        for (int m = 0; m < DM; ++m) qg[m] = s_q[(gbase & 255) + m];
      }
      s_next = s_S[i * TS + k]; v_next = s_V[i * TS + k];       // the next step's coefficients ride in this batch
      asm volatile("" ::"v"(uj), "v"(s_next), "v"(v_next), "v"(qg[0]));
      if (!SENT || (unsigned)__double2hiint(u[DM - 1]) != 0x7FF8DEADu) break;
    }
    if (DEFER) {
#pragma unroll
      for (int m = 0; m < DM; ++m) Prow[m] = __builtin_fma(-cq_prev, qg[m], Prow[m]);
    }
    double qa = 0.0, qb = 0.0;
#pragma unroll
    for (int m = 0; m < DM / 2; ++m) { qa = __builtin_fma(Prow[m], u[m], qa); qb = __builtin_fma(Prow[DM / 2 + m], u[DM / 2 + m], qb); }
    const double qj = qa + qb;
    double tq = uj * qj, d = uj * rj;
    tq += dpp_f64<0xB1>(tq); d += dpp_f64<0xB1>(d);
    tq += dpp_f64<0x4E>(tq); d += dpp_f64<0x4E>(d);
    tq += dpp_f64<0x141>(tq); d += dpp_f64<0x141>(d);
    const double den = __builtin_fma(s, tq, 1.0);
    double rc = __builtin_amdgcn_rcp(den);
    rc = rc * __builtin_fma(-den, rc, 2.0);
    rc = rc * __builtin_fma(-den, rc, 2.0);
    rj = __builtin_fma(__builtin_fma(-s, d, v) * rc, qj, rj);
    if (closer && i == k) { s_Fn[k * DM + j] = rj; s_q[256 + (k & 31) * DM + j] = rj * lamj; }
    if (!SENT && closer) {
      lds_wait();
      if (i == k && j == 0) s_Fn[T * DM - 1] = (double)k;
    }
    asm volatile("" ::: "memory");
    const double cq = s * rc * qj;
    if (DEFER) {
      s_q[tid & 255] = qj;
      cq_prev = cq;
    } else {
      s_q[tid & 255] = qj;
      lds_wait();
#pragma unroll
      for (int m = 0; m < DM; ++m) Prow[m] = __builtin_fma(-cq, s_q[(gbase & 255) + m], Prow[m]);
      lds_wait();
    }
  }
  const long long t1 = clock64();
  double acc = rj;
#pragma unroll
  for (int m = 0; m < DM; ++m) acc += Prow[m];
  if ((tid & 63) == 0) out[wave] = t1 - t0;
  if (acc == 1.2345) out[63] = 1;
}

template <int WAVES, int DEFER, int SENT>
static void run2(const char* what) {
  long long* d; (void)hipMalloc(&d, 64 * 8);
  const int steps = 64 * 40;
  const size_t lds = (2 * 64 * 65 + 64 * 8 + 512 + 64 * 8 + 8) * 8;
  (void)hipFuncSetAttribute((const void*)k_step2<WAVES, DEFER, SENT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_step2<WAVES, DEFER, SENT><<<1, 64 * WAVES, lds>>>(d, steps);
  k_step2<WAVES, DEFER, SENT><<<1, 64 * WAVES, lds>>>(d, steps);
  long long h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s %d waves: %7.1f clocks per step (wave 0), %7.1f (last wave)\n", what, WAVES, (double)h[0] / (steps - steps / 64), (double)h[WAVES - 1] / (steps - steps / 64));
  (void)hipFree(d);
}

template <int FLAGS, int WAVES>
static void run(const char* what) {
  long long* d; (void)hipMalloc(&d, 64 * 8);
  const int steps = 64 * 40;
  const size_t lds = (2 * 64 * 65 + 64 * 8 + 512 + 64 * 8 + 8) * 8;
  (void)hipFuncSetAttribute((const void*)k_step<FLAGS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_step<FLAGS, WAVES><<<1, 64 * WAVES, lds>>>(d, steps);
  k_step<FLAGS, WAVES><<<1, 64 * WAVES, lds>>>(d, steps);
  long long h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s %d waves: %7.1f clocks per step (wave 0), %7.1f (last wave)\n", what, WAVES, (double)h[0] / (steps - steps / 64), (double)h[WAVES - 1] / (steps - steps / 64));
  (void)hipFree(d);
}

int main() {
  run<F_ALL, 8>("everything");
  run<F_ALL & ~F_POLL, 8>("without the flag poll");
  run<F_ALL & ~F_UREAD, 8>("without the read of u");
  run<F_ALL & ~F_SV, 8>("without the reads of s, v");
  run<F_ALL & ~F_Q, 8>("without q = P u");
  run<F_ALL & ~F_SUMS, 8>("without the two group sums");
  run<F_ALL & ~F_RCP, 8>("without the reciprocal");
  run<F_ALL & ~F_PUB, 8>("without the publication");
  run<F_ALL & ~F_PUPD, 8>("without the LDS gather of the P step");
  run<F_Q | F_SUMS | F_RCP, 8>("arithmetic only (no LDS at all)");
  run2<8, 0, 0>("step2: prefetched s/v, flag poll, gather in step");
  run2<8, 1, 0>("step2: + gather deferred into the next batch");
  run2<8, 0, 1>("step2: + vector is its own flag");
  run2<8, 1, 1>("step2: both (one LDS batch per step)");
  run<F_ALL, 4>("everything");
  run<F_ALL, 2>("everything");
  run<F_ALL, 1>("everything");
  run<F_Q | F_SUMS | F_RCP, 1>("arithmetic only (no LDS at all)");
  return 0;
}
