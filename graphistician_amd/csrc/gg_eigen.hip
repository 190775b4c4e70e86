// gg_eigen.hip — logistic eigenmodel Gibbs (eigenmodel.py:822-965).  Placeholder
// until the PG(1,z) / latent-feature kernels land: every entry point reports
// GG_ERR_UNSUPPORTED (no CPU fallback).
#include "gg_common.h"

struct gg_eigen { int dummy; };

int gg_eigen_create(gg_eigen** out, const gg_config* cfg) {
  (void)out; (void)cfg;
  return gg_fail(GG_ERR_UNSUPPORTED, "logistic eigenmodel: not implemented in this build");
}
int gg_eigen_destroy(gg_eigen* e) { delete e; return GG_OK; }
int gg_eigen_set_state(gg_eigen*, int, const void*, size_t) { return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel"); }
int gg_eigen_get_state(gg_eigen*, int, void*, size_t) { return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel"); }
int gg_eigen_resample(gg_eigen*, const gg_data*, int) { return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel"); }
int gg_eigen_log_likelihood(gg_eigen*, const gg_data*, double*) { return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel"); }
int gg_eigen_stats(gg_eigen*, gg_stats*) { return gg_fail(GG_ERR_UNSUPPORTED, "eigenmodel"); }
