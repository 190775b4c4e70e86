// gg_math.h — reproducible double-precision elementary functions.
//
// The block-assignment draw (adjacency.py:695, weights.py:488 via pybasicbayes
// sample_discrete_from_log) exponentiates log-probabilities; for the HIP engine
// and the CPU oracle to pick the same label from the same uniform, exp/log must
// return identical bits on x86-64 and gfx950.  libm and ocml do not promise that,
// so both sides compile these routines: only IEEE-754 +,-,*,/ , fma and integer
// bit manipulation, in a fixed order.  Build with -ffp-contract=off on both
// compilers; fused operations are written as explicit gg_fma().
//
// Algorithms: argument reduction + minimax polynomial as in the classical
// fdlibm exp/log (error < 1 ulp); constants are the published fdlibm ones.
#pragma once
#include <stdint.h>
#include "gg_philox.h"

GG_HD double gg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
GG_HD double gg_sqrt(double x) { return __builtin_sqrt(x); }
GG_HD double gg_fabs(double x) { return __builtin_fabs(x); }

GG_HD uint64_t gg_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
GG_HD double gg_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

GG_HD double gg_inf() { return gg_u2d(0x7FF0000000000000ull); }
GG_HD double gg_nan() { return gg_u2d(0x7FF8000000000000ull); }
GG_HD bool gg_isnan(double x) { return x != x; }

// 2^k for -1022 <= k <= 1023
GG_HD double gg_pow2i(int k) { return gg_u2d((uint64_t)(k + 1023) << 52); }

GG_HD double gg_exp(double x) {
  const double ln2HI = 6.93147180369123816490e-01;
  const double ln2LO = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;
  if (gg_isnan(x)) return x;
  if (x > 709.782712893383973096) return gg_inf();
  if (x < -745.13321910194110842) return 0.0;
  double ax = gg_fabs(x);
  if (ax < 3.7252902984619140625e-09) return 1.0 + x;     // |x| < 2^-28
  int k;
  double hi, lo;
  if (ax > 0.34657359027997264) {                          // |x| > 0.5 ln2
    double t = invln2 * x + (x < 0.0 ? -0.5 : 0.5);
    k = (int)t;
    double dk = (double)k;
    hi = x - dk * ln2HI;
    lo = dk * ln2LO;
  } else {
    k = 0; hi = x; lo = 0.0;
  }
  double r = hi - lo;
  double t = r * r;
  double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  double y;
  if (k == 0) return 1.0 - ((r * c) / (c - 2.0) - r);
  y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (k >= -1021) {
    if (k == 1024) return y * 2.0 * gg_pow2i(1023);
    return y * gg_pow2i(k);
  }
  return y * gg_pow2i(k + 1000) * gg_pow2i(-1000);
}

GG_HD double gg_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01;
  const double Lg2 = 3.999999999940941908e-01;
  const double Lg3 = 2.857142874366239149e-01;
  const double Lg4 = 2.222219843214978396e-01;
  const double Lg5 = 1.818357216161805012e-01;
  const double Lg6 = 1.531383769920937332e-01;
  const double Lg7 = 1.479819860511658591e-01;
  if (gg_isnan(x)) return x;
  if (x < 0.0) return gg_nan();
  if (x == 0.0) return -gg_inf();
  if (x == gg_inf()) return x;
  int k = 0;
  uint64_t u = gg_d2u(x);
  if ((u >> 52) == 0) {                     // subnormal: scale up by 2^54
    x = x * 18014398509481984.0;
    u = gg_d2u(x);
    k -= 54;
  }
  int e = (int)(u >> 52) - 1023;
  uint64_t mant = u & 0x000FFFFFFFFFFFFFull;
  // normalise to [sqrt(2)/2, sqrt(2))
  if (mant > 0x6A09E667F3BCCull) {          // 1+f > sqrt(2)
    e += 1;
    x = gg_u2d(mant | 0x3FE0000000000000ull);
  } else {
    x = gg_u2d(mant | 0x3FF0000000000000ull);
  }
  k += e;
  double f = x - 1.0;
  double dk = (double)k;
  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  double R = t2 + t1;
  double hfsq = 0.5 * f * f;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// x^y for x > 0 via exp(y log x); used only for the gamma(a<1) boost where a few
// ulp are irrelevant but host/device agreement is not.
GG_HD double gg_pow_pos(double x, double y) { return gg_exp(y * gg_log(x)); }

// Correctly rounded double of the exact integer hi*2^32 + lo, for |hi| < 2^52,
// 0 <= lo < 2^52: both conversions are exact and the fma rounds once.  The
// engines keep K x K weight tables as such (hi, lo) pairs so that the value
// handed to the conjugate update is independent of the reduction order.
GG_HD double gg_hilo_to_double(int64_t hi, int64_t lo) {
  return gg_fma((double)hi, 4294967296.0, (double)lo);
}
