// gg_philox.h — counter-based RNG shared by the HIP engine and the CPU oracle.
//
// Philox4x32-10 (Salmon et al., SC'11).  The reference library draws every
// random number from the global numpy Mersenne-Twister stream
// (e.g. graphistician/adjacency.py:638,702; SURVEY.md §8b "Threading"), which a
// parallel sampler cannot reproduce; the build replaces it by a stateless
// generator addressed by (seed, site, sweep, index) so that a node's uniform is
// known before the node is visited and host and device agree bit for bit.
//
// Header-only, compiles unchanged under g++ and hipcc (host + device).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GG_HD __host__ __device__ inline
#else
#define GG_HD inline
#endif

// Sites: one id per place in the sweep that consumes randomness.
enum gg_site {
  GG_SITE_ADJ_P = 1,       // Beta draws of p[c1,c2]           adjacency.py:638
  GG_SITE_ADJ_PSELF = 2,   // Beta draw of p_self               adjacency.py:648
  GG_SITE_ADJ_C = 3,       // categorical draw of c[n]          adjacency.py:695
  GG_SITE_ADJ_M = 4,       // Dirichlet draw of m               adjacency.py:702
  GG_SITE_W_NIW = 5,       // NIW draws of (mu,Sigma)[c1,c2]    weights.py:408
  GG_SITE_W_NIW_SELF = 6,  // NIW draw of the self Gaussian     weights.py:413
  GG_SITE_W_C = 7,         // categorical draw of c[n]          weights.py:488
  GG_SITE_W_M = 8,         // Dirichlet draw of m               weights.py:498
  GG_SITE_BB_P = 9,        // Beta-Bernoulli p                  adjacency.py:171-181
  GG_SITE_BB_PSELF = 10,
  GG_SITE_EIG_OMEGA = 11,  // PG(1,psi) draws                   eigenmodel.py:867
  GG_SITE_EIG_F = 12,      // latent feature draws              eigenmodel.py:908
  GG_SITE_EIG_MU0 = 13,    //                                   eigenmodel.py:932
  GG_SITE_EIG_LMBDA = 14,  //                                   eigenmodel.py:965
  GG_SITE_INIT = 15,       // prior draws at construction
  GG_SITE_DATA = 16,       // synthetic data generation (rvs)
  GG_SITE_EIG_Z = 17,      // truncated-normal auxiliaries of the probit eigenmodel   eigenmodel.py:291
  GG_SITE_LD = 18,         // latent-distance HMC: momenta and acceptance uniforms       adjacency.py:374-421
  GG_SITE_W_SHCOV = 19,    // shared covariance of SBMGaussianWeightSharedCov            weights.py:548
  GG_SITE_LDW_HMC = 20,    // latent-distance weights: HMC momenta and acceptance uniform  weights.py:777-792
  GG_SITE_LDW = 21,        // latent-distance weights: b, sigma, self Gaussian, eta draws  weights.py:833-886
  GG_SITE_EIG_FN = 22,     // eigenmodels: per-pair normals of the perturbation draw of F[n] (spec/gg_eig.h)   eigenmodel.py:908, :329
  GG_SITE_EIG_F_REF = 23   // oracle only: the D normals of F[n] drawn the reference's way (mean + chol(Sigma_post) z)   eigenmodel.py:903-908
};

struct gg_u32x4 { uint32_t v[4]; };

GG_HD void gg_mulhilo32(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
  uint64_t p = (uint64_t)a * (uint64_t)b;
  *hi = (uint32_t)(p >> 32);
  *lo = (uint32_t)p;
}

GG_HD gg_u32x4 gg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    gg_mulhilo32(0xD2511F53u, c0, &hi0, &lo0);
    gg_mulhilo32(0xCD9E8D57u, c2, &hi1, &lo1);
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  gg_u32x4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// 53-bit uniform in [0,1) from two 32-bit words.
GG_HD double gg_u53(uint32_t a, uint32_t b) {
  uint64_t hi = (uint64_t)(a >> 5);   // 27 bits
  uint64_t lo = (uint64_t)(b >> 6);   // 26 bits
  return (double)((hi << 26) | lo) * (1.0 / 9007199254740992.0);
}

// The single uniform consumed by a categorical draw at (site, sweep, index).
GG_HD double gg_uniform_at(uint64_t seed, uint32_t site, uint32_t sweep, uint32_t index) {
  gg_u32x4 r = gg_philox4x32_10(0u, index, sweep, site, (uint32_t)seed, (uint32_t)(seed >> 32));
  return gg_u53(r.v[0], r.v[1]);
}

// A sequential stream addressed by (seed, site, sweep, index): used by draws
// that need an unbounded number of variates (rejection samplers).
struct gg_stream {
  uint32_t k0, k1;
  uint32_t c0, c1, c2, c3;
  uint32_t w2, w3;   // second half of the last Philox block
  int have;          // 1: (w2, w3) not consumed yet
};

GG_HD void gg_stream_init(gg_stream* s, uint64_t seed, uint32_t site, uint32_t sweep, uint32_t index) {
  s->k0 = (uint32_t)seed;
  s->k1 = (uint32_t)(seed >> 32);
  s->c0 = 0u; s->c1 = index; s->c2 = sweep; s->c3 = site;
  s->have = 0;
  s->w2 = s->w3 = 0u;
}

// The same addressing for a 64-bit index - a PAIR of nodes, i N + j, which passes 2^32 at N = 65536 (the eigenmodels' auxiliary
// variables): the low word goes where the 32-bit index goes, the high word into the top byte of the word that counts the
// stream's blocks (no stream draws 2^24 blocks), so an index below 2^32 addresses exactly the stream it always did and
// indices up to 2^40 (N < 2^20) never collide.
GG_HD void gg_stream_init64(gg_stream* s, uint64_t seed, uint32_t site, uint32_t sweep, uint64_t index) {
  gg_stream_init(s, seed, site, sweep, (uint32_t)index);
  s->c0 = (uint32_t)(index >> 32) << 24;
}
GG_HD double gg_uniform_at64(uint64_t seed, uint32_t site, uint32_t sweep, uint64_t index) {
  gg_u32x4 r = gg_philox4x32_10((uint32_t)(index >> 32) << 24, (uint32_t)index, sweep, site, (uint32_t)seed, (uint32_t)(seed >> 32));
  return gg_u53(r.v[0], r.v[1]);
}

// Words are consumed in pairs: block b yields uniforms (w0,w1) then (w2,w3).
GG_HD double gg_stream_uniform(gg_stream* s) {
  if (s->have) {
    s->have = 0;
    return gg_u53(s->w2, s->w3);
  }
  gg_u32x4 r = gg_philox4x32_10(s->c0, s->c1, s->c2, s->c3, s->k0, s->k1);
  s->c0 += 1u;
  s->w2 = r.v[2]; s->w3 = r.v[3];
  s->have = 1;
  return gg_u53(r.v[0], r.v[1]);
}

// Uniform in (0,1): never returns 0 (safe under log).
GG_HD double gg_stream_uniform_pos(gg_stream* s) {
  double u = gg_stream_uniform(s);
  return u + (0.5 / 9007199254740992.0);
}
