/* graphgibbs.h — C ABI of the MI355X Gibbs engine (libgraphgibbs.so).
 *
 * Drop-in boundary for graphistician's Aldous-Hoover Gibbs hot path.  The
 * reference is pure Python; the binding a maintainer adds is a ctypes stub
 * (INTEGRATION.md).  Each entry point names the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions: every function returns 0 on success or a negative gg_status;
 * nothing throws across the ABI; gg_last_error() gives the text of the last
 * failure on the calling thread.  Plain pointers and sizes only.  A handle is
 * not re-entrant; distinct handles are independent.  There is NO CPU fallback:
 * without a HIP device every call that needs one fails with GG_ERR_NO_DEVICE.
 */
#ifndef GRAPHGIBBS_H
#define GRAPHGIBBS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gg_data gg_data;   /* device-resident network (A, A^T, W, W^T, diagonals) */
typedef struct gg_model gg_model; /* one component distribution with device-resident state */

enum gg_status {
  GG_OK = 0,
  GG_ERR_NO_DEVICE = -1,
  GG_ERR_INVALID = -2,
  GG_ERR_HIP = -3,
  GG_ERR_UNSUPPORTED = -4,
  GG_ERR_NUMERIC = -5
};

/* Component kinds.  Reference classes:
 *  GG_SBM_ADJ       adjacency.py:440-702  SBMAdjacencyDistribution
 *  GG_BB_ADJ        adjacency.py:110-188  BetaBernoulliAdjacencyDistribution
 *  GG_BERN_ADJ      adjacency.py:65-107   BernoulliAdjacencyDistribution (scalar p, p_self)
 *  GG_SBM_GAUSS_W   weights.py:172-498    SBMGaussianWeightDistribution
 *  GG_NIW_W         weights.py:68-162     NIWGaussianWeightDistribution
 *  GG_FIXED_W       weights.py:9-65       FixedGaussianWeightDistribution
 *  GG_LOGISTIC_EIG  eigenmodel.py:822-965 _GibbsLogisticEigenmodel
 *  GG_PROBIT_EIG    eigenmodel.py:239-379 _GibbsProbitEigenmodel (same state ids; the auxiliaries are Z, not omega)
 *  GG_SBM_GAUSS_SHCOV_W  weights.py:501-610  SBMGaussianWeightSharedCov (one covariance for all block pairs; GG_ST_NIW packs
 *                   the mean prior mu_0, the inverse-Wishart scale lmbda_0 and nu_0; GG_ST_SIGMA holds the shared
 *                   covariance in every pair)
 *  GG_SBM_JOINT     stochastic_block_model.py:202-309, 589-678  GaussianStochasticBlockModel (legacy): ONE label vector drives
 *                   adjacency and weights - resample = weight models of every block pair (:220-226) -> resample_p (:232-250)
 *                   -> resample_c with Bernoulli + Gaussian terms (:252-302) -> resample_m (:304-309).  State: GG_ST_C, M, PI,
 *                   P, MU, SIGMA, NIW; no special-cased self connections (the diagonal belongs to pair (c_n, c_n)); B <= 2.
 *  GG_LATENT_DIST   adjacency.py:191-437  LatentDistanceAdjacencyDistribution (HMC; D = dim, sigma_F = sigma;
 *                   parts: GG_PART_F = L, GG_PART_MU0 = mu_0, GG_PART_LMBDA = mu_self)
 *  GG_LATENT_DIST_W weights.py:613-903    LatentDistanceGaussianWeightDistribution (B = 1; D = dim; state: GG_ST_F = L,
 *                   GG_ST_MU0 = b, GG_ST_LD_AUX = {eta, sigma, mu_self, sigma_self}, GG_ST_NIW = {-, lmbda_0, nu_0, -},
 *                   GG_ST_NIW_SELF = {mu_0, sigma_0, nu_0, kappa_0}; parts: GG_PART_F = L (HMC), GG_PART_MU0 = b,
 *                   GG_PART_PARAMS = sigma + self Gaussian, GG_PART_LMBDA = eta)
 */
enum gg_kind {
  GG_SBM_ADJ = 1,
  GG_BB_ADJ = 2,
  GG_BERN_ADJ = 3,
  GG_SBM_GAUSS_W = 4,
  GG_NIW_W = 5,
  GG_FIXED_W = 6,
  GG_LOGISTIC_EIG = 7,
  GG_PROBIT_EIG = 8,
  GG_LATENT_DIST = 9,
  GG_SBM_GAUSS_SHCOV_W = 10,
  GG_LATENT_DIST_W = 11,
  GG_SBM_JOINT = 12
};

typedef struct gg_config {
  uint32_t struct_size;           /* = sizeof(gg_config) as the CALLER compiled / declared it; gg_model_create refuses a
                                     mismatch (a stale binding would otherwise be read past its end) */
  int32_t kind;                   /* gg_kind */
  int32_t N;                      /* nodes */
  int32_t K;                      /* blocks C, 1..256 (1 for non-SBM kinds) */
  int32_t B;                      /* weight dimension (weights kinds), 1..4 */
  int32_t D;                      /* latent dimension (eigenmodel) */
  int32_t allow_self_connections; /* adjacency.py:461 */
  int32_t special_case_self_conns;/* adjacency.py:462, weights.py:189; BB: self_connection (adjacency.py:119-125) */
  int32_t device;                 /* HIP device ordinal */
  int32_t reserved0;              /* keeps seed 8-byte aligned without hidden padding; set to 0 */
  uint64_t seed;                  /* Philox key */
  double tau1, tau0;              /* Beta prior (adjacency.py:460) */
  double sigma_mu0, sigma_F, sigma_lmbda, mu_mu_0; /* eigenmodel.py:41-56 */
  int32_t lmbda_given;            /* eigenmodel.py:65-67 */
  int32_t wide_weights;           /* 1: reference-precision weights for GG_SBM_GAUSS_W / GG_NIW_W with B = 1 (at most 64 blocks): the
                                     network must come from gg_data_create_wide - 48 bits per weight in two limbs instead of 24 */
  int32_t shard_begin, shard_end; /* sharded single chain: this rank owns nodes [shard_begin, shard_end); 0,0 = all */
  /* sharded single chain, block-interleaved ownership (shard_block > 0; shard_begin / shard_end are then ignored): the nodes are
   * dealt out in blocks of shard_block consecutive nodes (a power of two >= 8), block b to rank b % shard_world - every band of
   * the sequential scan then spreads over all ranks */
  int32_t shard_block, shard_rank, shard_world;
  int32_t reserved1;              /* set to 0 */
} gg_config;

/* State ids for gg_get_state / gg_set_state.  Shapes in comments, C order. */
enum gg_state_id {
  GG_ST_C = 1,          /* int32  (N)        block labels                 adjacency.py:496, weights.py:206 */
  GG_ST_M = 2,          /* double (K)        block probabilities          adjacency.py:486, weights.py:205 */
  GG_ST_PI = 3,         /* double (K)        Dirichlet prior              adjacency.py:469-473 */
  GG_ST_P = 4,          /* double (K,K)      connection probabilities     adjacency.py:507 */
  GG_ST_P_SELF = 5,     /* double (1)                                     adjacency.py:512 */
  GG_ST_MU = 6,         /* double (K,K,B)    Gaussian means               weights.py:231-235 */
  GG_ST_SIGMA = 7,      /* double (K,K,B,B)  Gaussian covariances         weights.py:237-241 */
  GG_ST_MU_SELF = 8,    /* double (B)                                     weights.py:226-228 */
  GG_ST_SIGMA_SELF = 9, /* double (B,B) */
  GG_ST_NIW = 10,       /* double (B + B*B + 2) = mu_0, Sigma_0, nu_0, kappa_0   weights.py:208-222 */
  GG_ST_NIW_SELF = 11,  /* same layout, prior of the self Gaussian */
  GG_ST_SWEEP = 12,     /* uint32 (1)        sweep counter = Philox counter word */
  GG_ST_F = 13,         /* double (N,D)      latent features              eigenmodel.py:62 */
  GG_ST_LMBDA = 14,     /* double (D)                                     eigenmodel.py:64-73 */
  GG_ST_MU0 = 15,       /* double (1)                                     eigenmodel.py:57 */
  GG_ST_MU_LMBDA = 16,  /* double (D)        prior mean of lambda         eigenmodel.py:71 */
  GG_ST_LD_AUX = 18,    /* double (3)        latent-distance model: mu_self, HMC step size, running acceptance rate (adjacency.py:201-206); (4) for GG_LATENT_DIST_W */
  GG_ST_OMEGA_S = 17    /* double (N,N) read-only: omega_ij + omega_ji off the diagonal, omega_ii on it (eigenmodel.py:859-874) */
};

/* Parts of a sweep, for step-wise parity tests (gg_resample_part). */
enum gg_part {
  GG_PART_PARAMS = 1, /* resample_p (adjacency.py:620) / resample_mu_and_Sig (weights.py:399) */
  GG_PART_C = 2,      /* resample_c (adjacency.py:650, weights.py:415) */
  GG_PART_M = 4,      /* resample_m (adjacency.py:697, weights.py:493) */
  GG_PART_OMEGA = 8,  /* _resample_Omega (eigenmodel.py:859) */
  GG_PART_F = 16,     /* _resample_F     (eigenmodel.py:876) */
  GG_PART_MU0 = 32,   /* _resample_mu_0  (eigenmodel.py:910) */
  GG_PART_LMBDA = 64, /* _resample_lmbda (eigenmodel.py:934) */
  GG_PART_ALL = 127
};

typedef struct gg_stats {
  uint32_t struct_size;         /* IN: sizeof(gg_stats) as the CALLER compiled / declared it - gg_last_stats writes no more than
                                   that many bytes and refuses 0 (a binding built against an older, shorter struct is not overrun) */
  uint32_t reserved;
  uint64_t sweeps;              /* sweeps run on this handle */
  uint64_t label_changes;       /* labels changed by resample_c since creation */
  uint64_t resolve_rounds;      /* rounds of the node scans: 1 per quiet scan, 1 per grid-wide exchange of a resident scan */
  double bulk_bytes;            /* algorithmic bytes streamed by the bulk statistics kernels, last call */
  double bulk_ms;               /* HIP-event time of those kernels, last call */
  uint64_t bulk_launches;       /* number of bulk kernel launches, last call */
  double total_ms;              /* HIP-event time of the whole last call - measured only while timing is on (gg_set_timing > 0), 0 otherwise */
  /* sweeps ENQUEUED since creation, by execution form (block models; a sweep voided by a stalled scan and run again is
   * counted again).  fused: one launch per sweep (scan | gated parameter step | riders); two_launch: parameter step +
   * riders, scan + riders (the first / last sweep of a run, K > 32 without staged tables); serial: pass, parameter step,
   * scan as separate launches (chains that keep changing labels, wide and joint models, GG_AHEAD=0) */
  uint64_t sweeps_fused, sweeps_two_launch, sweeps_serial;
} gg_stats;

/* ---- library ------------------------------------------------------------- */
int gg_device_count(int* count);
const char* gg_last_error(void);
const char* gg_version(void);

/* ---- data: replaces the (A, W) ndarray pair the reference passes to every
 * resample / log_likelihood call (networks.py:28,69).  A is (N,N) C-order,
 * A[i][j] != 0 means an edge i -> j (adjacency.py:686-692), except the value 2 (GG_A_MISSING): an UNOBSERVED entry - the
 * reference's NaN, defined for the eigenmodels only (eigenmodel.py:121-127; every other kind returns GG_ERR_UNSUPPORTED for a
 * network that has one, and so does a network with weights); W is (N,N,B) C-order
 * float64 or NULL for adjacency-only use.  The engine copies; the caller keeps
 * ownership.  Weights are stored as fixed point. */
int gg_data_create(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, int32_t device);
/* A model that has sweeps enqueued on a network (gg_resample_async) holds a reference to it until it is synchronised:
 * destroying the handle earlier only marks it, the last reference frees it. */
/* The same with REFERENCE-PRECISION weights (B = 1): W is kept as two 24-bit limbs, w = (q0 + q1 2^-24) 2^-shift - 48 bits
 * under the one scale instead of 24 - for data whose dynamic range the default storage cannot carry (a few large weights
 * among many small ones: gg_data_quantisation).  The engine's statistics of the two limbs are exact integers as before, and
 * they determine sum w and sum w^2 exactly; a model uses such data iff it was created with
 * gg_config.wide_weights = 1.  Twice the weight bytes per pass. */
int gg_data_create_wide(gg_data** out, int32_t N, const uint8_t* A, const double* W, int32_t device);
int gg_data_destroy(gg_data* d);
/* Replace the network held by an existing handle with another (A, W) of the same N, B (W NULL iff it was created without
 * weights): one staging copy instead of freeing and allocating every device buffer - for callers that draw fresh data
 * every iteration (the Geweke loops of test/test_logistic_eigenmodel_geweke.py:35-47).  Refused while a model still has
 * sweeps enqueued on the handle. */
int gg_data_update(gg_data* d, const uint8_t* A, const double* W);
int gg_data_info(const gg_data* d, int32_t* N, int32_t* B, int32_t* shift, int64_t* n_edges);
/* Dynamic range of the stored weights.  W is kept as 24-bit fixed point with ONE shift for the whole matrix, fixed by the
 * largest |w| on an edge (exact, order-independent integer sums: DESIGN.md section 2) - 24 bits ABSOLUTE where the
 * reference's float64 (abstractions.py:261) keeps 53 relative.  w_max: that largest |w|; w_low: the median of |w| over
 * the edges' components; low_bits: the significant bits a weight of size w_low keeps, floor(log2(w_low 2^shift)) (-1: not
 * measured - no weights, or data generated on the device).  low_bits < 12: a few outliers have set the scale and the bulk of
 * the data is stored coarsely - the Python binding warns; rescale, clip or transform the weights or the wide storage
 * (gg_data_create_wide). */
int gg_data_quantisation(const gg_data* d, double* w_max, double* w_low, int32_t* low_bits, int32_t* shift);
/* Synthetic planted-SBM network generated on the device (bench / large N):
 * mirrors SBMAdjacencyDistribution.rvs (adjacency.py:558-560) and
 * SBMGaussianWeightDistribution.rvs (weights.py:360-373) for B = 1. */
int gg_data_create_sbm(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                       const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                       int32_t device);
/* Column shard for a chain sharded over several GPUs: this rank owns nodes
 * [n0, n1) and stores A[:, n0:n1] and A[n0:n1, :]^T (same for W), i.e. everything the per-node statistics
 * of its own nodes need.  A and W are the FULL (N,N[,B]) host arrays (the fixed-point scale is global). */
int gg_data_create_shard(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, int32_t n0, int32_t n1,
                         int32_t device);
/* The column shard [n0, n1) of the network gg_data_create_sbm generates (same Philox counters). */
int gg_data_create_sbm_shard(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                             const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                             int32_t n0, int32_t n1, int32_t device);
/* Block-interleaved column shards (gg_config.shard_block): rank `rank` of `world` owns the blocks b = rank, rank + world, ... of
 * `block` consecutive nodes (a power of two >= 8) and stores their columns of A / A^T (W / W^T) side by side in block order. */
int gg_data_create_shard_blocks(gg_data** out, int32_t N, int32_t B, const uint8_t* A, const double* W, int32_t block, int32_t rank,
                                int32_t world, int32_t device);
int gg_data_create_sbm_shard_blocks(gg_data** out, int32_t N, int32_t K, const int32_t* c, const double* p, double p_self,
                                    const double* mu, const double* sigma, double mu_self, double sigma_self, uint64_t seed,
                                    int32_t block, int32_t rank, int32_t world, int32_t device);
/* 64-bit content hash of a host buffer on `threads` host threads (0 = all): what the Python binding keys its device copies of
 * the caller's (A, W) arrays with (networks.py:28,69: the reference reads its arrays on every call, so an in-place edit must be
 * seen).  The digest does not depend on the number of threads; a change confined to one 8-byte word always changes it. */
int gg_hash64(const void* buf, size_t nbytes, int32_t threads, uint64_t* out);
/* Copy the stored network back (de-quantised weights); W may be NULL. */
int gg_data_download(const gg_data* d, uint8_t* A, double* W);

/* ---- models ---------------------------------------------------------------- */
int gg_model_create(gg_model** out, const gg_config* cfg);
int gg_model_destroy(gg_model* m);
int gg_set_state(gg_model* m, int32_t id, const void* buf, size_t nbytes);
int gg_get_state(gg_model* m, int32_t id, void* buf, size_t nbytes);

/* n_sweeps x resample(data): adjacency.py:615-618 / weights.py:394-397 /
 * adjacency.py:153-181 / weights.py:154-162 / eigenmodel.py:836-857.
 * beta is the AIS inverse temperature (graph_models.py:50-64); 1.0 = plain Gibbs.  Defined by the reference for the
 * ER / SBM adjacency models; build-defined for the weight kinds (statistics x beta), the latent-distance models (likelihood
 * term x beta) and GG_LOGISTIC_EIG (omega ~ PG(beta, psi), kappa x beta); GG_PROBIT_EIG returns GG_ERR_UNSUPPORTED for beta != 1. */
int gg_resample(gg_model* m, const gg_data* d, int32_t n_sweeps, double beta);
/* Step-wise parts (the reference's public resample_p / resample_c / resample_m, _resample_Omega / _resample_F / ...).
 * Sweep-index rule: every draw is addressed by (seed, site, sweep index, element).  Parts called one after the other
 * share the current sweep index until one of them is asked for a second time (or, for the block models, until
 * GG_PART_M - the last step of a sweep - has run): then the index advances, so that repeating a part never repeats
 * its variates.  A full gg_resample after step-wise parts starts at a fresh index as well.  GG_ST_SWEEP reads / sets
 * the index (setting it forgets which parts were drawn under it). */
int gg_resample_part(gg_model* m, const gg_data* d, int32_t parts, double beta);
/* The reference's list-taking forms: SBMAdjacencyDistribution.resample_p(As) (adjacency.py:620-648; GG_PART_PARAMS: the
 * block-pair counts summed over the nd graphs, one Beta draw per pair) and the eigenmodels' resample(networks)
 * (eigenmodel.py:836-857, :246-272: auxiliary variables per graph, every conditional from the summed statistics).
 * nd = 1 is gg_resample_part. */
int gg_resample_part_many(gg_model* m, const gg_data* const* ds, int32_t nd, int32_t parts, double beta);
/* Stream-ordered form: gg_resample_async enqueues n_sweeps sweeps on the model's HIP stream and
 * returns; gg_sync (and every call that reads state) waits for them.  gg_resample = async + sync.
 * Distinct models run on distinct streams, so the adjacency and weight halves of
 * FactorizedNetworkDistribution.resample (networks.py:69-71), which share only the data, overlap. */
int gg_resample_async(gg_model* m, const gg_data* d, int32_t n_sweeps, double beta);
int gg_sync(gg_model* m);
/* gg_sync for n independent models at once, each driven from its own host thread so that the work the
 * host enqueues while finishing their scans overlaps on the device (the two halves of networks.py:69-71). */
int gg_sync_many(gg_model** models, int32_t n);
/* n_sweeps x resample for n independent models (models[i] on data[i]), one host thread per model. */
int gg_resample_many(gg_model** models, const gg_data* const* data, int32_t n, int32_t n_sweeps, double beta);

/* abstractions.py:111-115 (adjacency kinds), weights.py:316-331 /
 * abstractions.py:257-274 (weight kinds), eigenmodel.py:110-129. */
int gg_log_likelihood(gg_model* m, const gg_data* d, double* out);
/* the same for n independent models (models[i] on data[i] -> out[i]); the block models' passes are all enqueued
 * before the first is waited for: the two halves of FactorizedNetworkDistribution.log_likelihood (networks.py:28-31)
 * run concurrently. */
int gg_log_likelihood_many(gg_model** models, const gg_data* const* data, int32_t n, double* out);

/* log_prior of n models at once (models[i] -> out[i]): SBMAdjacencyDistribution.log_prior (adjacency.py:538-555),
 * SBMGaussianWeightDistribution.log_prior (weights.py:333-357), SBMGaussianWeightSharedCov.log_prior (weights.py:554-577) from
 * the state where it lives - what `log_probability` (abstractions.py:39-40) adds to the log-likelihood once per sweep in the
 * reference's loops.  Other kinds: GG_ERR_UNSUPPORTED (their priors are O(1) host formulas in the Python classes). */
int gg_log_prior_many(gg_model** models, int32_t n, double* out);

/* approx_predictive_ll (abstractions.py:157-182, 313-357; networks.py:60-67), the held-out evaluation of
 * examples/demo_model_comparison.py:61-74: log-likelihood of an (N+1)-th node's row and column, a_row / a_col uint8 [N+1]
 * (entry N = the self connection), w_row / w_col double [N+1][B] (weight kinds; NULL otherwise), for each block k the new
 * node may belong to -> ll_k[K] (K = 1 kinds: one value).  The caller draws c2 ~ Cat(m) M times and averages exp(ll_k[c2]). */
int gg_predictive_ll(gg_model* m, const uint8_t* a_row, const uint8_t* a_col, const double* w_row, const double* w_col, double* ll_k);

/* Conditional log-probabilities lp[K] of node n under the current state
 * (the vector handed to sample_discrete_from_log at adjacency.py:695 /
 * weights.py:488) — for parity tests. */
int gg_node_lp(gg_model* m, const gg_data* d, int32_t n, double* lp);

/* out->struct_size must be set by the caller (see gg_stats). */
int gg_last_stats(gg_model* m, gg_stats* out);
/* Device time of the bulk statistics kernel alone (HIP events on the engine stream), for the roofline line. */
int gg_bench_bulk(gg_model* m, const gg_data* d, int32_t reps, double* ms_per_launch, double* bytes_per_launch);
/* Measured ceiling beside the nominal HBM peak: streams `bytes` of device memory `reps`
 * times with 16-byte loads; mode 0 = read only, 1 = copy.  Average device ms per pass. */
int gg_bench_stream(int32_t device, int64_t bytes, int32_t reps, int32_t mode, double* ms_per_pass);
/* on = n > 0: record HIP events (on the engine stream) around the bulk kernel of every n-th sweep, so that
 * gg_last_stats reports bulk_ms / bulk_launches / bulk_bytes for the sampled launches; 0 = off. */
int gg_set_timing(gg_model* m, int32_t on);

/* ---- one chain sharded over the GPUs of a node (block models GG_SBM_ADJ / GG_SBM_GAUSS_W, beta = 1) ---------------------
 * SURVEY.md section 8(e), DESIGN.md section 8.  Rank r owns nodes [shard_begin, shard_end) (gg_config) and holds the matching
 * column shard of the network (gg_data_create_shard / gg_data_create_sbm_shard); every rank holds all labels.  All traffic
 * is integer (exact, order independent), so the chain equals the single-GPU chain bit for bit for any number of ranks.
 * Collectives go through a communicator:
 *   gg_comm_create_rccl   RCCL on device buffers, enqueued on the engine stream (xGMI inside a node); the 128-byte unique id
 *                         comes from gg_comm_unique_id on one rank and reaches the others through whatever launched them;
 *   gg_comm_create_host   a pair of host callbacks (in-place SUM of int64; all-gather of int32) - the protocol's test
 *                         harness (gloo through ctypes), buffers staged through the host.
 * gg_shard_sweep runs n half-sweeps: per half-sweep one all-reduce of the K x K integer tables, then the sequential scan
 * (adjacency.py:650-695, weights.py:415-488) in rounds - every rank evaluates its own nodes of the current band (as wide as the
 * previous scan's change count lets the reports describe, 4096 .. 65536 nodes) under the changes hypothesised so far and reports
 * its first 64 surprises (node, labels); one all-gather per round; every rank accepts the hypotheses up to the earliest surprise
 * and amends the rest - enqueued in batches on the engine stream; the host reads the scan position once
 * per batch, never once per change.  Ownership: a contiguous range (shard_begin / shard_end) or, so that every band spreads over
 * all ranks, blocks dealt out round-robin (shard_block / shard_rank / shard_world, gg_data_create_shard_blocks). */
typedef struct gg_comm gg_comm;
typedef int (*gg_host_allreduce_fn)(void* user, int64_t* buf, int64_t n);                          /* in place, SUM over ranks */
typedef int (*gg_host_allgather_fn)(void* user, const int32_t* send, int32_t* recv, int64_t n);    /* recv[world][n] */
int gg_comm_unique_id(uint8_t* out128);
int gg_comm_create_rccl(gg_comm** out, int32_t rank, int32_t world, const uint8_t* id128, int32_t device);
int gg_comm_create_host(gg_comm** out, int32_t rank, int32_t world, gg_host_allreduce_fn ar, gg_host_allgather_fn ag, void* user,
                        int32_t device);
int gg_comm_destroy(gg_comm* c);
/* collectives: all-reduces and all-gathers that went over the wire (none with one rank); rounds: rounds of the sharded scans
 * driven through the communicator (each is one all-gather when world > 1).  Any output pointer may be NULL. */
int gg_comm_info(const gg_comm* c, int32_t* rank, int32_t* world, int32_t* is_rccl, uint64_t* collectives, uint64_t* rounds);
int gg_shard_sweep(gg_model* m, const gg_data* d, gg_comm* c, int32_t n_half_sweeps);
/* log-likelihood from the all-reduced tables (abstractions.py:111-115, weights.py:316-331) */
int gg_shard_log_likelihood(gg_model* m, const gg_data* d, gg_comm* c, double* out);

/* Device self-test of the shared arithmetic spec: evaluates exp/log/sqrt/div
 * on the device for n inputs (parity tests compare with the host build). */
int gg_selftest_math(int32_t device, int32_t op, const double* x, const double* y, double* out, int64_t n);
/* Conjugate draws executed on the device (same streams as the host spec). */
int gg_selftest_draws(int32_t device, int32_t what, uint64_t seed, const double* params, int32_t nparams,
                      double* out, int64_t n, int32_t stride);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHGIBBS_H */
