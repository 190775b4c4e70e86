// gg_latent.h — scalar arithmetic of the latent-distance model (adjacency.py:191-437), shared by the HIP engine
// and the CPU oracle: numerically stable Bernoulli-logit terms and the step-size adaptation of the HMC update.
#pragma once
#include <math.h>
#include "gg_math.h"

// log(1 + e^x) and 1 / (1 + e^-x) without overflow (the reference writes log(P), log(1 - P): adjacency.py:262-266)
GG_HD double gg_softplus(double x) { return x > 0.0 ? x + gg_log(1.0 + gg_exp(-x)) : gg_log(1.0 + gg_exp(x)); }
GG_HD double gg_sigmoid(double x) {
  if (x >= 0.0) return 1.0 / (1.0 + gg_exp(-x));
  const double e = gg_exp(x);
  return e / (1.0 + e);
}

// Both at once from a single exponential: e = exp(-|x|), softplus = max(x, 0) + log(1 + e), sigmoid = 1 / (1 + e) or
// e / (1 + e).  (Same values as the two functions above.)
// (exp / log here are the platform's: libm on the host, ocml on the device, <= 1 ulp apart.  The model's parity is a
// float tolerance anyway (its sums are ordered differently), and the pass is bound by these two calls: the
// branch-free ocml versions run about twice as fast as the bit-reproducible gg_exp / gg_log.)
GG_HD void gg_sigmoid_softplus(double x, double* P, double* sp) {
  const double e = exp(x > 0.0 ? -x : x);
  const double d = 1.0 + e;
  *sp = (x > 0.0 ? x : 0.0) + log(d);
  *P = x >= 0.0 ? 1.0 / d : e / d;
}

// The sigmoid alone, same arithmetic as in gg_sigmoid_softplus (gradient-only passes skip the logarithm).
GG_HD double gg_sigmoid_only(double x) {
  const double e = exp(x > 0.0 ? -x : x);
  const double d = 1.0 + e;
  return x >= 0.0 ? 1.0 / d : e / d;
}

// hips.inference.hmc, adaptive_step_sz=True, as recollected (the package is not in the image: parity unpinned):
// running acceptance rate smoothed with constant 0.95; the step grows 2 % while the rate is above the 0.9
// target and shrinks 2 % otherwise, clipped to [1e-3, 1].
GG_HD void gg_hmc_adapt(int accepted, double* step_sz, double* accept_rate) {
  const double r = 0.95 * *accept_rate + 0.05 * (accepted ? 1.0 : 0.0);
  double s = *step_sz * (r > 0.9 ? 1.02 : 0.98);
  s = s < 1e-3 ? 1e-3 : (s > 1.0 ? 1.0 : s);
  *step_sz = s;
  *accept_rate = r;
}
