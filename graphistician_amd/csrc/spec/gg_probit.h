// gg_probit.h — arithmetic of the probit eigenmodel's auxiliary variables (shared by the HIP engine and the
// CPU oracle).  Reference: eigenmodel.py:274-293 (_resample_Z) -> internals/utils.py:53-76 (sample_truncnorm):
//   cdf_lb = Phi(lb - mu), cdf_ub = Phi(ub - mu), c = cdf_lb + u (cdf_ub - cdf_lb), c clipped to [1e-15, 1 - 1e-15],
//   z = mu - sqrt(2) erfcinv(2 c) = mu + Phi^{-1}(c),
// with (lb, ub) = (0, inf) where A = 1 and (-inf, 0) where A = 0.  scipy's erfcinv is not available on either
// side, so Phi^{-1} is restated here: Wichura's algorithm AS 241 (PPND16, Appl. Statist. 37 (1988) 477-484,
// relative accuracy about 1e-16), written on gg_log / gg_sqrt so that host and device agree bit for bit given c.
// Phi itself is erfc (libm / ocml, <= 1 ulp apart): parity for this model is a float tolerance.
#pragma once
#include "gg_math.h"
#include "gg_philox.h"
#include "gg_pg.h"

// Phi^{-1}(p), 0 < p < 1 (AS 241, PPND16).
GG_HD double gg_ndtri(double p) {
  const double q = p - 0.5;
  if ((q < 0 ? -q : q) <= 0.425) {
    const double r = 0.180625 - q * q;
    const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r +
                            4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r +
                          1.3314166789178437745e2) * r + 3.3871328727963666080e0);
    const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r +
                            2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r +
                          4.2313330701600911252e1) * r + 1.0);
    return q * num / den;
  }
  double r = q < 0 ? p : 1.0 - p;
  r = gg_sqrt(-gg_log(r));
  double val;
  if (r <= 5.0) {
    r -= 1.6;
    const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                            1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                          4.63033784615654529590e0) * r + 1.42343711074968357734e0);
    const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                            1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                          2.05319162663775882187e0) * r + 1.0);
    val = num / den;
  } else {
    r -= 5.0;
    const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                            2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                          5.46378491116411436990e0) * r + 6.65790464350110377720e0);
    const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                            7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                          5.99832206555887937690e-1) * r + 1.0);
    val = num / den;
  }
  return q < 0 ? -val : val;
}

// z ~ N(mu, 1) truncated to z > 0 (a != 0) or z < 0 (a == 0), by inversion with the reference's clipping.
GG_HD double gg_truncnorm(double u, double mu, int a) {
  const double c0 = gg_norm_cdf(-mu);                       // Phi(0 - mu)
  double c = a ? c0 + u * (1.0 - c0) : u * c0;
  c = c < 1e-15 ? 1e-15 : (c > 1.0 - 1e-15 ? 1.0 - 1e-15 : c);
  return mu + gg_ndtri(c);
}

GG_HD double gg_truncnorm_at(uint64_t seed, uint32_t sweep, uint32_t N, uint32_t i, uint32_t j, double mu, int a) {
  return gg_truncnorm(gg_uniform_at(seed, GG_SITE_EIG_Z, sweep, i * N + j), mu, a);
}
