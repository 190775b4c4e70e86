// gg_model.hip — host orchestration of the block-model components behind the C ABI.
//
// Mirrors, per kind (include/graphgibbs.h):
//   resample        adjacency.py:615-618, :153-181; weights.py:394-397, :154-162
//   log_likelihood  abstractions.py:111-115; weights.py:316-331
// State lives on the device between calls; the Python classes read it through
// gg_get_state.  No CPU fallback: every entry point needs a HIP device.
#include <math.h>
#include <stdlib.h>

#include "gg_common.h"
#include <thread>
#include <chrono>
#include <atomic>
#include <rccl/rccl.h>
#include "gg_block_kernels.h"
#include "gg_resolve_kernels.h"
#include "gg_scan_kernels.h"
#include "gg_sweepq.h"

struct gg_eigen;  // gg_eigen.hip
int gg_eigen_create(gg_eigen** out, const gg_config* cfg);
int gg_eigen_destroy(gg_eigen* e);
int gg_eigen_set_state(gg_eigen* e, int id, const void* buf, size_t nbytes);
int gg_eigen_get_state(gg_eigen* e, int id, void* buf, size_t nbytes);
int gg_eigen_resample(gg_eigen* e, const gg_data* d, int parts);
int gg_eigen_resample_many(gg_eigen* e, const gg_data* const* ds, int nd, int parts);
void gg_eigen_set_beta(gg_eigen* e, double beta);
int gg_eigen_log_likelihood(gg_eigen* e, const gg_data* d, double* out);
int gg_eigen_stats(gg_eigen* e, gg_stats* out);
struct gg_latent;  // gg_latent.hip
int gg_latent_create(gg_latent** out, const gg_config* cfg);
int gg_latent_destroy(gg_latent* e);
int gg_latent_set_state(gg_latent* e, int id, const void* buf, size_t nbytes);
int gg_latent_get_state(gg_latent* e, int id, void* buf, size_t nbytes);
void gg_latent_set_beta(gg_latent* e, double beta);
int gg_latent_resample(gg_latent* e, const gg_data* d, int parts, int defer);
int gg_latent_sync(gg_latent* e);
int gg_latent_log_likelihood(gg_latent* e, const gg_data* d, double* out);
int gg_latent_stats(gg_latent* e, gg_stats* out);

struct gg_model {
  gg_config cfg;
  int kind;
  bool is_adj, is_block, sbm;
  bool joint;                           // GG_SBM_JOINT: a weight model that also carries p and the adjacency terms of the conditional
  bool wide1;                           // wide weights (gg_config.wide_weights): the data are the two 24-bit limbs of a one-dimensional
                                        // weight - B = 2 through every statistics kernel, Bx = 1 for the parameters (spec/gg_lp.h)
  int Bx;                               // dimension of the Gaussian PARAMETERS (= B unless wide1)
  int N, K, Kp, B, NT, NSI, Np, NTAB;   // N = all nodes (labels, rows); Np = padded number of OWNED nodes
  int Nl, n0;                           // owned nodes [n0, n0 + Nl) (all of them unless the chain is sharded; n0 = 0 when block-interleaved)
  gg_own own;                           // the owned nodes, in general (gg_common.h)
  bool sharded;
  int ngraphs = 1;                      // graphs whose tables the parameter step is drawing from (gg_resample_part_many)
  long long* tab_sum = nullptr;         // their summed K x K tables
  unsigned char* pred_buf = nullptr;    // gg_predictive_ll: grow-only device scratch (the reference calls it once per held-out node)
  size_t pred_cap = 0;
  long long* sh_buf;                    // sharded chain: device staging of the diagonal statistics (all-reduce)
  int32_t *sh_trip, *sh_all;            // this rank's first changer {node, old, new}; the gathered ones [world][3]
  long long sh_diag[1 + GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2];   // global diagonal statistics of network sh_diag_uid
  unsigned long long sh_diag_uid;
  int sh_world;
  gg_eigen* eig;
  gg_latent* ld;                // GG_LATENT_DIST, GG_LATENT_DIST_W: everything lives in gg_latent.hip
  // device state
  int32_t *c, *cand, *cnt;
  // label sort, double buffered: a completed scan writes the sort of its final labels into the other
  // buffer, so that the bulk kernel of the following sweep (launched early, see enqueue_sweep_pipe)
  // can still read the old one
  int32_t *permb[2], *labb[2];
  int pcur;
  double *m, *logm, *pi;
  double *p, *p_self;
  double *mu, *sigma, *mu_self, *sigma_self;
  double *T, *Tout, *Tin;
  // statistics: three buffers [Gcnt out | Gcnt in | GS out | GS in], used in rotation.  Sweep s reads Gbuf[cur]; in the
  // "ahead" form (quiet chains, enqueue_sweep_ahead) the riders of its two launches count sweep s+1's statistics into
  // Gbuf[cur+1] while k_params zeroes Gbuf[cur+2] for the riders of sweep s+1; the serial form (enqueue_sweep_serial)
  // counts into Gbuf[cur] itself and zeroes Gbuf[cur+1].
  unsigned char* Gbuf[3];
  size_t Gbytes;
  int cur;
  bool zeroed[3];
  bool perm_valid;
  long long* tab;
  int32_t *rs, *L;   // rs = control block (gg_resolve_kernels.h)
  double* lp_out;
  double* ll_out;              // k_loglik's result {log-likelihood, not-positive-definite flag}
  bool ll_host_form;
  bool m_pending;              // resample_m of sweep m_sweep not drawn yet (done by the next k_params or at sync)
  uint32_t m_sweep;
  double beta;                 // inverse temperature of the sweeps being enqueued / pending
  int pending;                 // optimistic sweeps enqueued and not yet confirmed
  const gg_data* pend_data;
  // host mirrors / hypers
  gg_niw niw, niw_self;
  int32_t* h_rs;  // pinned copy of the control block
  long long* h_tab;   // pinned staging of the K x K tables + block counts (gg_log_likelihood)
  hipStream_t stream;
  bool can_ahead;               // this model may run the ahead form (scan model, not sharded, GG_AHEAD != 0)
  int scan_ride;                // the resident-grid scan carries the next sweep's bulk pass as riders: 0 never (default; wide, sharded, joint), 1 while the chain changes
                                // few labels per sweep (GG_SCAN_RIDE=1), 2 always (GG_SCAN_RIDE=2: tests).  Exact, and slower: see launch_scan
  long long ride_chg_seen;      // label changes at the last completed batch, and what that batch changed per sweep (-1: no batch yet)
  double chg_per_sweep;
  bool ahead;                   // Gbuf[cur] holds the exact statistics of network ahead_uid under the CURRENT labels
  unsigned long long ahead_uid; // (counted by the riders of the last completed sweep, whose scan changed no label)
  unsigned* verdict;            // verdict word of the fused launch, alone on its cache line
  long long* swq_dbg;           // GG_SWQ_DBG=1: timestamps of the last fused launch (gg_ride_args::dbg)
  unsigned vtag;                // tag of the last fused launch's verdict word (k_sweep_q)
  bool pre;                     // ahead form: the parameter step of the NEXT sweep to enqueue has run already (inside the last k_sweep_q)
  double ride_split;            // share of the next sweep's bulk work items that rides in the k_params launch (rest: k_spec_resolve)
  int cur0, pcur0;              // buffer indices when the pending batch started
  uint32_t sweep;
  int parts_done;               // step-wise calls: parts already drawn under the current sweep index (gg_resample_part)
  int Lmax, Lcap, CH, max_idle;
  long long* dbg;               // GG_ROUND_DBG: phase timestamps of k_round
  unsigned long long* bar;      // k_scan: exchange slots
  double* csc;                  // shared-covariance kind: centred scatter of every block pair
  int round_hint;               // restart rounds the last unfinished scan needed: size of the first batch next time
  int depth;                    // sweeps enqueued ahead of confirmation: halves to 1 when a scan stalls, doubles on clean batches
  int todo;                     // sweeps requested but not enqueued yet (depth-limited)
  bool use_scan;                // the scan runs as one resident-grid launch (k_scan) instead of speculate + resolve (+ rounds)
  int scan_pref;                // GG_SCAN: 1 always, 0 never, unset (-1): adaptive - k_scan while the chain keeps changing labels
  long long chg_seen;           // label changes (control word 2) at the last decision
  long long clean_run;          // consecutive sweeps without a label change while in k_scan mode
  int32_t* sh_cntn;             // sharded scan: [Nl][K] block sizes as own node n sees them
  int32_t* sh_H;                //               [GG_SHARD_WMAX] hypothesis per node of the band
  int32_t* perm_loc;            //               the label sort with local node indices (k_perm_local)
  int sh_band;                  //               width of the scan's band chosen for the next scan (0: not yet)
  gg_shard_state* sh_state;     //               scan position / band start (two copies) + the evaluation launch's ticket
  int sh_hint;                  //               rounds the last scan of this model needed (+ margin): the first batch of the next
  bool wide;                    // 65..256 blocks: a node's lanes span two or four wavefronts - the resident scan's wide instantiation is the
                                // only scan form (no speculative / restart-round fallback, no node shards, no gg_node_lp)
  bool ev_call;                 // this call is bracketed by ev_call0 / ev_call1 (gg_set_timing > 0)
  bool timing_dirty;            // the last call's HIP-event times have not been read yet (gg_last_stats does: each query costs
                                // microseconds, and a caller that does not ask for them should not wait for them)
  int ahead_cap;                // speculative form: sweeps enqueued ahead of the last confirmed one (64, doubling per clean batch up to 512)
  int scan_pref_saved;          // the preference a k_scan timeout suspended (0: none suspended)
  double* h_epi;                // pinned: the record the epilogue launch of a call's last sweep leaves (k_spec_resolve_epi)
  double* h_lp;                 // pinned: where k_log_prior leaves its value
  double* epi_terms; int32_t* epi_words;   // its device scratch: the pairs' log-likelihood terms; ticket, bad flag
  bool epi_hit;                 // the control block just fetched came from the epilogue's record
  unsigned epi_tag;             // tag of the epilogue launch the current batch ends with (0: none - the control block is copied)
  bool epi_ll;                  // ... and it evaluates the log-likelihood
  bool ll_asked; int calls_since_ll;   // the caller's pattern: a log_likelihood every ll_period resample calls (the epilogue of
  int ll_period;                       // the call it is expected behind computes it: 1 = the demo loop, 2 = the AIS loop, ...)
  bool ll_from_cache;           // the pending log_likelihood call is answered from ll_val
  bool ll_cached; double ll_val; bool ll_bad; uint64_t ll_uid;   // the log-likelihood of the state as it is, from the last epilogue
  int32_t done_base;            // control word 5 (scans completed, a running count on the device) when the current batch was enqueued
  uint64_t scan_retry_at;       // st.sweeps count from which the suspended preference is tried again
  uint64_t scan_backoff;        // sweeps to wait after the next timeout (doubles with every timeout, at most 2^20)
  int scan_spin;                // polls before a k_scan exchange gives up (GG_SCAN_SPIN overrides the default for tests)
  long long scan_hold;          // clean sweeps required before leaving k_scan mode (doubles when the chain stalls again soon after)
  long long since_revert;       // sweeps completed in speculate + resolve mode since the last switch back
  int32_t* ap;                  // restart rounds: entries of L already applied to node n's statistics
  size_t resolve_lds;
  bool resolve_stage;          // coefficient tables staged in the resolver's LDS
  bool shard_from_tab;         // enqueue_params: take the (all-reduced) tables from `tab`
  // timing
  int timing;        // 0 = off; n > 0: HIP events around the bulk kernel of every n-th sweep
  std::vector<hipEvent_t> ev;
  size_t ev_used;
  std::vector<uint32_t> ev_sweep;   // sweep index of each timed launch (event pair); ~0u = the launch was a no-op
  std::vector<double> ev_bytes;
  hipEvent_t ev_call0, ev_call1;
  gg_stats st;
};

// The network a model has sweeps enqueued on is held by reference until they are confirmed (gg_common.h).
static void set_pend_data(gg_model* m, const gg_data* d) {
  if (m->pend_data == d) return;
  if (d) gg_data_retain(d);
  if (m->pend_data) gg_data_release(m->pend_data);
  m->pend_data = d;
}

static int model_kind_ok(int kind) {
  return kind == GG_SBM_ADJ || kind == GG_BB_ADJ || kind == GG_BERN_ADJ || kind == GG_SBM_GAUSS_W || kind == GG_NIW_W ||
         kind == GG_FIXED_W || kind == GG_SBM_GAUSS_SHCOV_W || kind == GG_SBM_JOINT;
}

template <typename T>
static int dmalloc(T** p, size_t n) {
  GG_HIP(hipMalloc((void**)p, n * sizeof(T)));
  GG_HIP(hipMemset(*p, 0, n * sizeof(T)));
  return GG_OK;
}

static size_t resolve_lds_bytes(const gg_model* m, int CH) {
  size_t b = 0;
  if (!m->is_adj) b += (size_t)2 * CH * m->K * m->NSI * sizeof(long long);
  b += (size_t)2 * CH * m->K * sizeof(int);
  b += (size_t)m->K * sizeof(int) + 2 * (size_t)CH * sizeof(int) + 8 * sizeof(int);
  b += m->is_adj ? (size_t)CH * CH : (size_t)CH * CH * m->B * sizeof(int32_t);
  return (b + 15) & ~(size_t)15;
}

static size_t table_bytes(const gg_model* m) { return (size_t)2 * m->K * m->NT * m->Kp * sizeof(double); }

// Dynamic LDS of the scan kernels (bytes) and whether the coefficient tables are staged in it.  One place for the
// launch sites and for the capacity check at model creation: 150 KB is what a workgroup may ask for next to the few
// static words these kernels keep.
#define GG_LDS_LIMIT ((size_t)150 * 1024)
// nodes a workgroup of the quiet scan evaluates (k_spec_resolve, k_spec_resolve_epi, k_sweep_q): one pass of 256 / Kp nodes, or
// GG_WIDE_P passes' worth for more than 64 blocks (gg_spec_eval_wide)
static int spec_nodes_per_wg(const gg_model* m) { return (256 / m->Kp) * (m->Kp > 64 ? GG_WIDE_P(m->NT) : 1); }

static size_t lds_spec(const gg_model* m, int* stage_out) {
  const int K = m->K, per = spec_nodes_per_wg(m);
  size_t lds = (size_t)2 * K * per * m->NT * sizeof(double);
  const int stage = (lds + table_bytes(m) <= 64 * 1024) ? 1 : 0;
  if (stage) lds += table_bytes(m);
  else if (m->Kp == 64) lds += (size_t)4 * 2 * m->NT * 64 * sizeof(double);   // gg_lp_split4's meeting place
  if (m->Kp > 64) lds += (size_t)2 * 256 * sizeof(double) + 16 * sizeof(int) + 8;   // the cross-wave categorical's scratch
  if (stage_out) *stage_out = stage;
  return lds;
}
static size_t lds_round(const gg_model* m, int* stage_out) {
  const int K = m->K, per = 256 / m->Kp;
  const size_t gsb = m->is_adj ? 0 : (size_t)2 * K * per * m->NSI * sizeof(long long);
  const size_t meet = (size_t)4 * 2 * m->NT * 64 * sizeof(double);
  size_t lds = (size_t)2 * K * per * m->NT * sizeof(double) + (size_t)2 * K * per * sizeof(int) +
               (size_t)(per + 8 + K) * sizeof(int) + 64;
  const int stage = (lds + gsb + table_bytes(m) <= 64 * 1024) ? 1 : 0;
  if (stage) lds += gsb + table_bytes(m);
  else if (m->Kp == 64 && !m->is_adj) lds += gsb > meet ? gsb : meet;   // k_round: the meeting place reuses the dead integer statistics
  else if (m->Kp == 64) lds += gsb + meet;
  else lds += gsb;
  if (m->Kp > 64) lds += (size_t)2 * 256 * sizeof(double) + 16 * sizeof(int) + 16;   // the cross-wave categorical's scratch
  if (stage_out) *stage_out = stage;
  return lds;
}
// nodes of a group of the resident scan (k_scan): 256 / Kp, twice that for more than 64 blocks with NT <= 3
// (and for 33..64 blocks: Kp = 64 never has both table layouts in LDS, so it always runs the split form, MODE 1)
static int scan_nodes_per_wg(const gg_model* m) {
  return (256 / m->Kp) * (m->Kp > 64 ? GG_SCAN_WIDE_P(m->NT) : (m->Kp == 64 ? GG_SCAN_SPLIT_P(m->NT) : 1));
}

// 9..32 blocks: the resident scan's conditional on the matrix cores when the tables are staged - GG_MFMA_SMALL=1; off by default:
// exact (tests) but no faster (N=4096, K=16: 1.78 against 1.70 us of a round's 8.6 - with the tables and the statistics both
// in LDS the vector form's conditional is the latency of its LDS round trips, and the matrix form adds a barrier and a round
// trip through the meeting place for the arithmetic it saves)
static bool scan_mf_small(const gg_model* m) {
  const char* e = getenv("GG_MFMA_SMALL");
  return (m->Kp == 16 || m->Kp == 32) && e && atoi(e) != 0;
}
static size_t lds_scan(const gg_model* m, int lcap, int* stage_out, int per_override = 0) {
  const int K = m->K, per = per_override ? per_override : scan_nodes_per_wg(m);
  size_t lds = (size_t)2 * K * per * m->NT * sizeof(double) + (size_t)2 * K * per * sizeof(int) +
               (size_t)(per + K + 8) * sizeof(int) + 64 + (size_t)lcap * 3 * sizeof(int);
  // deviations, the round's surprises, per-node block sizes, scratch, one hypothesis per node of a band
  lds += 136 * sizeof(unsigned long long) + 256 * 3 * sizeof(int) + (size_t)per * K * sizeof(int) + 16 * sizeof(int) + 16 +
         (size_t)GG_SCAN_MAXWG * per * sizeof(short) + (size_t)(GG_SCAN_MAXWG * per / 32 + 2) * sizeof(unsigned);
  if (!m->is_adj) lds += (size_t)2 * K * per * m->NSI * sizeof(long long);
  if (m->Kp > 64) lds += (size_t)2 * 256 * sizeof(double) + 16 * sizeof(int) + 8;   // the cross-wave categorical's scratch
  // coefficient table: both lane-major layouts when that stays below 64 KB (1); else one padded copy of T[a][b][t] when
  // that stays below 72 KB (2: K up to 32 at B = 1); else from global memory (0).  (At K = 64 the one copy fits the 150 KB
  // but gains 10 % of a round's evaluation and leaves no room for anything else on the CU: not taken.)  Kp = 64 without layout 1 evaluates with
  // gg_lp_split4, which needs its meeting place.
  const size_t split_acc = m->Kp == 64 ? (per_override ? (size_t)per_override * 64 * sizeof(double) : GG_SCAN_MEET_BYTES(m->NT)) : 0;
  const size_t one_copy = (size_t)K * ((K * m->NT) | 1) * sizeof(double);
  int stage = 0;
  // (9..32 blocks with both table layouts in LDS: the conditional runs on the matrix cores - gg_lp_mfma_small - and its chains
  // meet in 2 NT x 256 doubles more)
  const size_t small_meet = scan_mf_small(m) ? (size_t)2 * m->NT * 256 * sizeof(double) : 0;
  if (lds + small_meet + table_bytes(m) <= 64 * 1024) { stage = 1; lds += small_meet + table_bytes(m); }
  else if (lds + split_acc + one_copy <= 72 * 1024) { stage = 2; lds += split_acc + one_copy; }   // (two workgroups per CU)
  else lds += split_acc;
  if (stage_out) *stage_out = stage;
  return lds;
}


extern "C" {

int gg_model_create(gg_model** out, const gg_config* cfg) {
  if (!out || !cfg) return gg_fail(GG_ERR_INVALID, "gg_model_create: null argument");
  if (cfg->struct_size != (uint32_t)sizeof(gg_config))
    return gg_fail(GG_ERR_INVALID, "gg_config.struct_size = %u but this library's gg_config has %zu bytes: the caller's binding of "
                   "include/graphgibbs.h is out of date (or struct_size was not set)", cfg->struct_size, sizeof(gg_config));
  if (cfg->kind == GG_LATENT_DIST || cfg->kind == GG_LATENT_DIST_W) {
    int rc = gg_ensure_device(cfg->device);
    if (rc) return rc;
    gg_model* m = new gg_model();
    m->cfg = *cfg; m->kind = cfg->kind;
    rc = gg_latent_create(&m->ld, cfg);
    if (rc) { delete m; return rc; }
    *out = m;
    return GG_OK;
  }
  if (cfg->kind == GG_LOGISTIC_EIG || cfg->kind == GG_PROBIT_EIG) {
    int rc = gg_ensure_device(cfg->device);
    if (rc) return rc;
    gg_model* m = new gg_model();
    m->cfg = *cfg; m->kind = cfg->kind;
    rc = gg_eigen_create(&m->eig, cfg);
    if (rc) { delete m; return rc; }
    *out = m;
    return GG_OK;
  }
  if (!model_kind_ok(cfg->kind)) return gg_fail(GG_ERR_INVALID, "unknown model kind %d", cfg->kind);
  if (cfg->N <= 0) return gg_fail(GG_ERR_INVALID, "N must be positive");
  int rc = gg_ensure_device(cfg->device);
  if (rc) return rc;
  gg_model* m = new gg_model();
  m->cfg = *cfg;
  m->kind = cfg->kind;
  m->is_adj = (cfg->kind == GG_SBM_ADJ || cfg->kind == GG_BB_ADJ || cfg->kind == GG_BERN_ADJ);
  m->sbm = (cfg->kind == GG_SBM_ADJ || cfg->kind == GG_SBM_GAUSS_W || cfg->kind == GG_SBM_GAUSS_SHCOV_W || cfg->kind == GG_SBM_JOINT);
  m->joint = cfg->kind == GG_SBM_JOINT;
  m->is_block = true;
  m->N = cfg->N;
  m->K = m->sbm ? cfg->K : 1;
  if (m->K < 1 || m->K > 256) { delete m; return gg_fail(GG_ERR_UNSUPPORTED, "number of blocks C=%d not in 1..256", m->K); }
  m->B = m->is_adj ? 1 : cfg->B;
  if (!m->is_adj && !(m->B >= 1 && m->B <= GG_MAXB)) {
    delete m;
    return gg_fail(GG_ERR_UNSUPPORTED, "weight dimension B=%d: the HIP engine supports B in 1..%d", cfg->B, GG_MAXB);
  }
  m->Bx = m->B;
  m->wide1 = false;
  if (cfg->wide_weights) {
    // reference-precision weights: 48 bits in two limbs (spec/gg_lp.h)
    if (!(cfg->kind == GG_SBM_GAUSS_W || cfg->kind == GG_NIW_W) || cfg->B != 1 || m->K > 64) {
      const int kd = cfg->kind, bb = cfg->B, kk = m->K;
      delete m;
      return gg_fail(GG_ERR_UNSUPPORTED, "wide_weights: for GG_SBM_GAUSS_W / GG_NIW_W with B = 1 and at most 64 blocks (kind %d, B = %d, C = %d)", kd, bb, kk);
    }
    m->wide1 = true; m->B = 2; m->Bx = 1;
  }
  if (m->joint && m->B > 2) {
    delete m;
    return gg_fail(GG_ERR_UNSUPPORTED, "GG_SBM_JOINT: weight dimension B=%d (the joint model's scan kernels are built for B <= 2)", cfg->B);
  }
  if ((cfg->kind == GG_SBM_ADJ || cfg->kind == GG_BB_ADJ || cfg->kind == GG_SBM_JOINT) &&
      !(cfg->tau1 > 0.0 && cfg->tau0 > 0.0 && cfg->tau1 - cfg->tau1 == 0.0 && cfg->tau0 - cfg->tau0 == 0.0)) {
    delete m;
    return gg_fail(GG_ERR_INVALID, "Beta prior (tau1, tau0) = (%g, %g) must be positive and finite", cfg->tau1, cfg->tau0);
  }
  m->Kp = gg_next_pow2(m->K < 2 ? 2 : m->K);
  m->NT = m->is_adj ? 2 : gg_nt_gauss(m->B) + (m->joint ? 1 : 0);
  m->NSI = m->is_adj ? 0 : gg_nsi(m->B);
  m->NTAB = m->is_adj ? 1 : 1 + 2 * m->NSI;
  m->n0 = 0; m->Nl = m->N; m->sharded = false;
  m->own = gg_own{0, m->N, -1, 0, 1};
  if (cfg->shard_block > 0) {
    if (!gg_own_blocks(m->N, cfg->shard_block, cfg->shard_rank, cfg->shard_world, &m->own) || m->own.Nl < 1) {
      const int bl = cfg->shard_block, rk = cfg->shard_rank, wd = cfg->shard_world, NN = m->N;
      delete m;
      return gg_fail(GG_ERR_INVALID, "shard_block %d (a power of two >= 8), rank %d of %d: no such share of %d nodes", bl, rk, wd, NN);
    }
    m->Nl = m->own.Nl;
    m->sharded = m->own.lb >= 0;      // (gg_own_blocks turns one rank's share into the plain range [0, N))
  } else if (cfg->shard_end > cfg->shard_begin) {
    if (cfg->shard_begin < 0 || cfg->shard_end > m->N) { delete m; return gg_fail(GG_ERR_INVALID, "shard [%d,%d) outside [0,%d)", cfg->shard_begin, cfg->shard_end, m->N); }
    m->n0 = cfg->shard_begin; m->Nl = cfg->shard_end - cfg->shard_begin;
    m->sharded = !(m->n0 == 0 && m->Nl == m->N);
    m->own = gg_own{m->n0, m->Nl, -1, 0, 1};
  }
  m->Np = gg_round_up(m->Nl, 64);
  // NIW / non-SBM weight kinds always keep a separate self Gaussian (weights.py:91-94)
  if (m->kind == GG_NIW_W || m->kind == GG_FIXED_W) m->cfg.special_case_self_conns = 1;
  if (m->joint) m->cfg.special_case_self_conns = 0;   // stochastic_block_model.py:291-296: the diagonal belongs to pair (c_n, c_n)
  if (m->joint && m->sharded) { delete m; return gg_fail(GG_ERR_UNSUPPORTED, "GG_SBM_JOINT: not for node shards"); }
  m->wide = m->K > 64;
  // (more than 64 blocks: the joint model runs the same wide forms - its one extra statistic per block, the partner count, is
  // carried by every evaluation the plain kinds have; round 4)
  if (m->kind == GG_SBM_GAUSS_SHCOV_W && !m->cfg.special_case_self_conns && m->sharded) {
    delete m;
    return gg_fail(GG_ERR_UNSUPPORTED, "SBMGaussianWeightSharedCov without special_case_self_conns: the node-sharded chain does not carry the diagonal's share of the pair tables");
  }
  {
    // GG_ADJ_PRIO=1: the adjacency model (the short half of a network's sweep) on a queue of higher priority.  Measured at
    // N=65536, K=64 and N=4096, K=16: no difference either way (a queue's priority does not let a large workgroup into a compute
    // unit that small ones keep backfilling; what does is the order of the launches - pair scheduling, launch_scan) - off.
    int lo = 0, hi = 0;
    const char* e = getenv("GG_ADJ_PRIO");
    const bool want = m->is_adj && e && atoi(e) != 0;
    if (want && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo)
      GG_HIP(hipStreamCreateWithPriority(&m->stream, hipStreamDefault, hi));
    else
      GG_HIP(hipStreamCreate(&m->stream));
  }
  const int N = m->N, K = m->K, B = m->B, NT = m->NT;
  const size_t KK = (size_t)K * K;
  if ((rc = dmalloc(&m->c, N)) || (rc = dmalloc(&m->cand, N)) || (rc = dmalloc(&m->cnt, K)) ||
      (rc = dmalloc(&m->permb[0], N)) || (rc = dmalloc(&m->permb[1], N)) || (rc = dmalloc(&m->labb[0], K + 1)) ||
      (rc = dmalloc(&m->labb[1], K + 1)) || (rc = dmalloc(&m->m, K)) ||
      (rc = dmalloc(&m->logm, K)) || (rc = dmalloc(&m->pi, K)) || (rc = dmalloc(&m->T, KK * NT)) ||
      (rc = dmalloc(&m->Tout, (size_t)K * NT * m->Kp)) || (rc = dmalloc(&m->Tin, (size_t)K * NT * m->Kp)) ||
      (rc = dmalloc(&m->tab, KK * m->NTAB)) || (rc = dmalloc(&m->rs, 32)) || (rc = dmalloc(&m->lp_out, K)) || (rc = dmalloc(&m->ll_out, 2)))
    return rc;
  {
    // the 64-bit word the workgroups report their first changer to starts (and is always left) at "none"
    const unsigned long long none = GG_NO_CHANGE;
    GG_HIP(hipMemcpy(m->rs + GG_RS_MIN, &none, sizeof(none), hipMemcpyHostToDevice));
  }
  m->Gbytes = (size_t)2 * K * m->Np * (sizeof(int32_t) + (size_t)m->NSI * sizeof(long long));
  for (int b = 0; b < 3; ++b) {
    if ((rc = dmalloc(&m->Gbuf[b], m->Gbytes))) return rc;
    m->zeroed[b] = true;
  }
  if (m->is_adj || m->joint) {
    if ((rc = dmalloc(&m->p, KK)) || (rc = dmalloc(&m->p_self, 1))) return rc;
  }
  if (!m->is_adj) {
    if ((rc = dmalloc(&m->mu, KK * B)) || (rc = dmalloc(&m->sigma, KK * B * B)) || (rc = dmalloc(&m->mu_self, B)) ||
        (rc = dmalloc(&m->sigma_self, (size_t)B * B)) || (rc = dmalloc(&m->csc, KK * (B * B + 1))))
      return rc;
  }
  m->beta = 1.0;
  m->Lmax = N < 4096 ? N : 4096;
  if (m->Lmax < 1) m->Lmax = 1;
  m->Lcap = m->Lmax + 4096;   // the first resolve fills at most Lmax entries, restart rounds append behind them
  if ((rc = dmalloc(&m->L, (size_t)m->Lcap * 3))) return rc;
  if ((rc = dmalloc(&m->ap, (size_t)m->Nl + 1))) return rc;
  m->dbg = nullptr;
  m->ahead = false; m->ahead_uid = 0;
  m->bar = nullptr; m->use_scan = false;
  m->depth = 1 << 20; m->todo = 0; m->round_hint = 0;
  if ((rc = dmalloc(&m->bar, (size_t)512 * GG_SLOT_PAD))) return rc;
  if (getenv("GG_ROUND_DBG") && (rc = dmalloc(&m->dbg, (size_t)512 * 8 + 8))) return rc;
  if ((rc = dmalloc(&m->verdict, 64))) return rc;
  m->swq_dbg = nullptr;
  if (getenv("GG_SWQ_DBG") && (rc = dmalloc(&m->swq_dbg, 8))) return rc;
  GG_HIP(hipHostMalloc((void**)&m->h_rs, 32 * sizeof(int32_t)));
  GG_HIP(hipHostMalloc((void**)&m->h_tab, ((size_t)m->K * m->K * m->NTAB + m->K) * sizeof(long long)));
  GG_HIP(hipHostMalloc((void**)&m->h_epi, 16 * sizeof(double), hipHostMallocCoherent));   // (watched by the host while the kernel runs)
  memset(m->h_epi, 0, 16 * sizeof(double));
  if ((rc = dmalloc(&m->epi_terms, (size_t)m->K * m->K)) || (rc = dmalloc(&m->epi_words, 16))) return rc;
  // resolve chunk: one Kp-lane group per node, LDS budget 96 KiB
  int CH = 1024 / m->Kp;
  if (CH > 64) CH = 64;
  while (CH > 1 && resolve_lds_bytes(m, CH) > 96 * 1024) CH >>= 1;
  m->CH = CH;
  m->resolve_lds = resolve_lds_bytes(m, CH);
  m->resolve_stage = (m->resolve_lds + table_bytes(m) <= 150 * 1024);
  if (m->resolve_stage) m->resolve_lds += table_bytes(m);
  m->max_idle = 1;
  GG_HIP(hipEventCreate(&m->ev_call0));
  GG_HIP(hipEventCreate(&m->ev_call1));
  {
    // The scan as one resident-grid launch (k_scan): exact, never stalls the sweep pipeline, ~6.5 us per accepted change
    // (evaluate 3.3 + fence-free exchange 2.3 + apply 0.7) against ~17 us for a restart round, but a quiet scan costs
    // 15 us in situ against 11 for k_spec_resolve, with more interference on the other model's streaming.
    // Default: adaptive - k_scan while the chain keeps changing labels, the speculative one-launch form once it has been
    // quiet for a while (model_sync decides batch by batch; both give the same chain bit for bit).  GG_SCAN=1 / 0 force either.
    const char* es = getenv("GG_SCAN");
    const bool can_scan = m->sbm && m->K > 1 && !m->sharded;
    m->scan_pref = !can_scan ? 0 : (es ? (atoi(es) == 1 ? 1 : 0) : -1);
    // (more than 64 blocks, round 3: every form - the resident scan's wide instantiation, the speculative one-launch scan
    // k_spec_resolve<.., 2> for quiet chains, restart rounds as the fallback; the two-kernel form k_speculate + k_resolve is
    // the one they do not have, k_spec_resolve stands in for it)
    m->use_scan = m->scan_pref == 1;
    m->chg_seen = 0; m->clean_run = 0; m->scan_hold = 4; m->since_revert = 1 << 30;
    m->scan_pref_saved = 0; m->scan_retry_at = 0; m->ahead_cap = 64;
    { const char* er = getenv("GG_SCAN_RETRY"); m->scan_backoff = (er && atoll(er) > 0) ? (uint64_t)atoll(er) : 1024; }
    { const char* esp = getenv("GG_SCAN_SPIN"); m->scan_spin = esp && atoi(esp) > 0 ? atoi(esp) : GG_BAR_SPIN_LIMIT; }
    // The ahead form (enqueue_sweep_ahead): while the chain is quiet, the next sweep's bulk pass rides in this sweep's
    // k_params and k_spec_resolve launches.  GG_AHEAD=0 keeps every sweep in the serial form (bulk -> params -> scan).
    {
      const char* er2 = getenv("GG_SCAN_RIDE");
      m->scan_ride = (m->wide || m->joint || m->sharded) ? 0 : (er2 ? atoi(er2) : 0);
      m->ride_chg_seen = 0; m->chg_per_sweep = -1.0;
    }
    const char* ea = getenv("GG_AHEAD");
    m->can_ahead = can_scan && m->scan_pref < 0 && !(ea && atoi(ea) == 0) && !m->joint;   // (the joint kind runs the serial form)
    const char* ers = getenv("GG_RIDE_SPLIT");
    m->ride_split = ers ? atof(ers) : -1.0;   // (default chosen below, once the scan forms of this shape are known)
    if (can_scan) {
      // what fits in LDS decides which scan forms this shape (K, B) can run
      const bool fit_scan = lds_scan(m, m->Lcap < 1024 ? m->Lcap : 1024, nullptr) <= GG_LDS_LIMIT;
      const bool fit_round = lds_round(m, nullptr) <= GG_LDS_LIMIT;
      if (lds_spec(m, nullptr) > GG_LDS_LIMIT || (!fit_scan && !fit_round)) {
        const size_t need = lds_round(m, nullptr);
        gg_model_destroy(m);
        return gg_fail(GG_ERR_UNSUPPORTED, "K=%d with B=%d needs %zu KB of LDS per workgroup for the node scan (150 KB available)",
                       cfg->K, cfg->B, need / 1024);
      }
      if (!fit_scan) { m->scan_pref = 0; m->use_scan = false; m->can_ahead = false; }   // restart rounds only (their meeting place reuses dead statistics)
    }
    if (!(m->ride_split >= 0.0 && m->ride_split <= 1.0)) {
      // Riders get the occupancy of the launch they ride in: k_params and the lean k_spec_resolve (K <= 32, or tables
      // staged in LDS) leave them what the plain bulk kernel has; the K > 32 conditional with its table entries in
      // registers does not, so there the whole pass rides with k_params.
      int stage = 0;
      lds_spec(m, &stage);
      m->ride_split = ((!stage && m->Kp == 64) || m->wide) ? 1.0 : 0.5;
    }
  }
  // defaults: uniform m, pi = 1, p = 0.5, standard NIW
  std::vector<double> ones(K, 1.0), um(K, 1.0 / K), half(KK, 0.5);
  GG_HIP(hipMemcpy(m->pi, ones.data(), K * 8, hipMemcpyHostToDevice));
  GG_HIP(hipMemcpy(m->m, um.data(), K * 8, hipMemcpyHostToDevice));
  if (m->is_adj || m->joint) {
    GG_HIP(hipMemcpy(m->p, half.data(), KK * 8, hipMemcpyHostToDevice));
    double h = 0.5;
    GG_HIP(hipMemcpy(m->p_self, &h, 8, hipMemcpyHostToDevice));
  }
  if (!m->is_adj) {
    const int Bx = m->Bx;      // (the parameters' dimension: 1 for wide weights, whose buffers are sized for B = 2)
    memset(&m->niw, 0, sizeof(gg_niw));
    for (int i = 0; i < Bx; ++i) m->niw.sigma_0[i * Bx + i] = 1.0;
    m->niw.nu_0 = Bx + 2; m->niw.kappa_0 = 1.0;
    m->niw_self = m->niw;
    std::vector<double> sig(KK * Bx * Bx, 0.0), sself((size_t)Bx * Bx, 0.0);
    for (size_t ab = 0; ab < KK; ++ab)
      for (int i = 0; i < Bx; ++i) sig[ab * Bx * Bx + i * Bx + i] = 1.0;
    for (int i = 0; i < Bx; ++i) sself[i * Bx + i] = 1.0;
    GG_HIP(hipMemcpy(m->sigma, sig.data(), sig.size() * 8, hipMemcpyHostToDevice));
    GG_HIP(hipMemcpy(m->sigma_self, sself.data(), sself.size() * 8, hipMemcpyHostToDevice));
  }
  *out = m;
  return GG_OK;
}

int gg_model_destroy(gg_model* m) {
  if (!m) return GG_OK;
  hipSetDevice(m->cfg.device);
  if (m->eig) { gg_eigen_destroy(m->eig); delete m; return GG_OK; }
  if (m->ld) { gg_latent_destroy(m->ld); delete m; return GG_OK; }
  // (always: a call that ended on its epilogue's record may still have the pair scheduling's hand-over delay or a no-op
  // launch behind it on the stream)
  if (m->stream) hipStreamSynchronize(m->stream);
  if (m->pend_data) set_pend_data(m, nullptr);
  void* ptrs[] = {m->c, m->cand, m->cnt, m->permb[0], m->permb[1], m->labb[0], m->labb[1], m->m, m->logm, m->pi, m->p, m->p_self, m->mu, m->sigma,
                  m->mu_self, m->sigma_self, m->T, m->Tout, m->Tin, m->Gbuf[0], m->Gbuf[1], m->Gbuf[2], m->tab,
                  m->rs, m->L, m->lp_out, m->ll_out, m->ap, m->dbg, m->bar, m->csc, m->swq_dbg, m->verdict, m->sh_buf, m->sh_trip, m->sh_all,
                  m->sh_cntn, m->sh_H, m->sh_state, m->pred_buf, m->tab_sum, m->perm_loc};
  for (void* p : ptrs) if (p) hipFree(p);
  if (m->h_rs) hipHostFree(m->h_rs);
  if (m->h_tab) hipHostFree(m->h_tab);
  if (m->h_epi) hipHostFree(m->h_epi);
  if (m->h_lp) hipHostFree(m->h_lp);
  if (m->epi_terms) hipFree(m->epi_terms);
  if (m->epi_words) hipFree(m->epi_words);
  for (hipEvent_t e : m->ev) hipEventDestroy(e);
  if (m->ev_call0) hipEventDestroy(m->ev_call0);
  if (m->ev_call1) hipEventDestroy(m->ev_call1);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
  return GG_OK;
}

static int state_ptr(gg_model* m, int id, void** p, size_t* nbytes) {
  const size_t K = m->K, KK = K * K, B = m->Bx;
  *p = nullptr;
  switch (id) {
    case GG_ST_C: *p = m->c; *nbytes = (size_t)m->N * 4; break;
    case GG_ST_M: *p = m->m; *nbytes = K * 8; break;
    case GG_ST_PI: *p = m->pi; *nbytes = K * 8; break;
    case GG_ST_P: if (m->is_adj || m->joint) { *p = m->p; *nbytes = KK * 8; } break;
    case GG_ST_P_SELF: if (m->is_adj) { *p = m->p_self; *nbytes = 8; } break;
    case GG_ST_MU: if (!m->is_adj) { *p = m->mu; *nbytes = KK * B * 8; } break;
    case GG_ST_SIGMA: if (!m->is_adj) { *p = m->sigma; *nbytes = KK * B * B * 8; } break;
    case GG_ST_MU_SELF: if (!m->is_adj) { *p = m->mu_self; *nbytes = B * 8; } break;
    case GG_ST_SIGMA_SELF: if (!m->is_adj) { *p = m->sigma_self; *nbytes = B * B * 8; } break;
    default: break;
  }
  return *p ? GG_OK : GG_ERR_INVALID;
}

static int model_sync(gg_model* m, bool ctl_fresh = false);
// Forget statistics counted ahead for the next sweep (labels or data are about to change from outside, or the buffer
// is about to be reused): Gbuf[cur] then holds stale numbers and must be zeroed before the next bulk pass adds to it.
static void drop_ahead(gg_model* m) {
  if (m->ahead) { m->ahead = false; m->zeroed[m->cur] = false; }
}

int gg_set_state(gg_model* m, int32_t id, const void* buf, size_t nbytes) {
  if (m) m->ll_cached = false;   // (the state changes: what an epilogue evaluated no longer describes it)
  if (!m || !buf) return gg_fail(GG_ERR_INVALID, "gg_set_state: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig) return gg_eigen_set_state(m->eig, id, buf, nbytes);
  if (m->ld) return gg_latent_set_state(m->ld, id, buf, nbytes);
  { int rc0 = model_sync(m); if (rc0) return rc0; }
  const size_t B = m->Bx;
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    m->sweep = *(const uint32_t*)buf;
    m->parts_done = 0;
    return GG_OK;
  }
  if (id == GG_ST_NIW || id == GG_ST_NIW_SELF) {
    if (m->is_adj) return gg_fail(GG_ERR_INVALID, "NIW hypers on an adjacency model");
    size_t need = (B + B * B + 2) * 8;
    if (nbytes != need) return gg_fail(GG_ERR_INVALID, "NIW hypers: expected %zu bytes, got %zu", need, nbytes);
    const double* d = (const double*)buf;
    if (gg_first_nonfinite(d, B + B * B + 2) >= 0) return gg_fail(GG_ERR_INVALID, "NIW hypers: non-finite value");
    if (!gg_is_spd(d + B, (int)B)) return gg_fail(GG_ERR_INVALID, "NIW hypers: Sigma_0 is not symmetric positive definite");
    if (!(d[B + B * B] > (double)B - 1.0)) return gg_fail(GG_ERR_INVALID, "NIW hypers: nu_0 = %g must exceed B - 1 = %d", d[B + B * B], (int)B - 1);
    if (!(d[B + B * B + 1] > 0.0)) return gg_fail(GG_ERR_INVALID, "NIW hypers: kappa_0 = %g must be positive", d[B + B * B + 1]);
    gg_niw* h = id == GG_ST_NIW ? &m->niw : &m->niw_self;
    memset(h, 0, sizeof(gg_niw));
    for (size_t i = 0; i < B; ++i) h->mu_0[i] = d[i];
    for (size_t i = 0; i < B * B; ++i) h->sigma_0[i] = d[B + i];
    h->nu_0 = d[B + B * B];
    h->kappa_0 = d[B + B * B + 1];
    return GG_OK;
  }
  void* p; size_t need;
  if (state_ptr(m, id, &p, &need)) return gg_fail(GG_ERR_INVALID, "state id %d not valid for model kind %d", id, m->kind);
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  if (id == GG_ST_C) {
    const int32_t* c = (const int32_t*)buf;
    for (int n = 0; n < m->N; ++n)
      if (c[n] < 0 || c[n] >= m->K) return gg_fail(GG_ERR_INVALID, "label c[%d]=%d out of range [0,%d)", n, c[n], m->K);
    m->perm_valid = false;
    drop_ahead(m);          // statistics counted ahead belong to the old labels
  } else {
    // every other state is doubles: values the samplers cannot digest are refused here (gg_common.h)
    const double* v = (const double*)buf;
    const size_t n = nbytes / sizeof(double);
    const long long bad = gg_first_nonfinite(v, n);
    if (bad >= 0) return gg_fail(GG_ERR_INVALID, "state id %d: non-finite value at index %lld", id, bad);
    if (id == GG_ST_PI || id == GG_ST_M || id == GG_ST_P || id == GG_ST_P_SELF) {
      for (size_t i = 0; i < n; ++i) {
        if (id == GG_ST_PI && !(v[i] > 0.0)) return gg_fail(GG_ERR_INVALID, "Dirichlet parameter pi[%zu] = %g must be positive", i, v[i]);
        if (id != GG_ST_PI && !(v[i] >= 0.0 && v[i] <= 1.0)) return gg_fail(GG_ERR_INVALID, "state id %d: probability %g at index %zu outside [0, 1]", id, v[i], i);
      }
    }
    if (id == GG_ST_SIGMA || id == GG_ST_SIGMA_SELF) {
      for (size_t i = 0; i < n; i += B * B)
        if (!gg_is_spd(v + i, (int)B)) return gg_fail(GG_ERR_INVALID, "state id %d: covariance %zu is not symmetric positive definite", id, i / (B * B));
    }
  }
  GG_HIP(hipMemcpyAsync(p, buf, nbytes, hipMemcpyHostToDevice, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  return GG_OK;
}

int gg_get_state(gg_model* m, int32_t id, void* buf, size_t nbytes) {
  if (!m || !buf) return gg_fail(GG_ERR_INVALID, "gg_get_state: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig) return gg_eigen_get_state(m->eig, id, buf, nbytes);
  if (m->ld) return gg_latent_get_state(m->ld, id, buf, nbytes);
  { int rc0 = model_sync(m); if (rc0) return rc0; }
  const size_t B = m->Bx;
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    *(uint32_t*)buf = m->sweep;
    return GG_OK;
  }
  if (id == GG_ST_NIW || id == GG_ST_NIW_SELF) {
    if (m->is_adj) return gg_fail(GG_ERR_INVALID, "NIW hypers on an adjacency model");
    size_t need = (B + B * B + 2) * 8;
    if (nbytes != need) return gg_fail(GG_ERR_INVALID, "NIW hypers: expected %zu bytes, got %zu", need, nbytes);
    double* d = (double*)buf;
    const gg_niw* h = id == GG_ST_NIW ? &m->niw : &m->niw_self;
    for (size_t i = 0; i < B; ++i) d[i] = h->mu_0[i];
    for (size_t i = 0; i < B * B; ++i) d[B + i] = h->sigma_0[i];
    d[B + B * B] = h->nu_0;
    d[B + B * B + 1] = h->kappa_0;
    return GG_OK;
  }
  void* p; size_t need;
  if (state_ptr(m, id, &p, &need)) return gg_fail(GG_ERR_INVALID, "state id %d not valid for model kind %d", id, m->kind);
  if (nbytes != need) return gg_fail(GG_ERR_INVALID, "state id %d: expected %zu bytes, got %zu", id, need, nbytes);
  GG_HIP(hipMemcpyAsync(buf, p, nbytes, hipMemcpyDeviceToHost, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  return GG_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// sweep pieces
// ---------------------------------------------------------------------------
static int check_data(const gg_model* m, const gg_data* d) {
  if (!d) return gg_fail(GG_ERR_INVALID, "null data handle");
  if (d->N != m->N) return gg_fail(GG_ERR_INVALID, "data has N=%d, model has N=%d", d->N, m->N);
  if (!gg_own_same(d->own, m->own)) {
    if (d->own.lb < 0 && m->own.lb < 0)
      return gg_fail(GG_ERR_INVALID, "data holds nodes [%d,%d), model owns [%d,%d)", d->n0, d->n1, m->n0, m->n0 + m->Nl);
    return gg_fail(GG_ERR_INVALID, "data and model do not describe the same share of the nodes (data: %d nodes, blocks of %d, rank %d of %d; "
                                   "model: %d nodes, blocks of %d, rank %d of %d; block 0 = a contiguous range)", d->own.Nl,
                   d->own.lb < 0 ? 0 : 1 << d->own.lb, d->own.rank, d->own.world, m->own.Nl, m->own.lb < 0 ? 0 : 1 << m->own.lb,
                   m->own.rank, m->own.world);
  }
  if (d->device != m->cfg.device) return gg_fail(GG_ERR_INVALID, "data on device %d, model on device %d", d->device, m->cfg.device);
  if (d->n_missing) return gg_fail(GG_ERR_UNSUPPORTED, "the network has %lld unobserved entries: missing data is defined for the eigenmodels only (eigenmodel.py:121-127)", d->n_missing);
  if (!m->is_adj) {
    if (!d->has_w) return gg_fail(GG_ERR_INVALID, "weight model needs data created with W");
    if (d->B != m->B) return gg_fail(GG_ERR_INVALID, "data has B=%d, model has B=%d", d->B, m->B);
    if ((d->wide != 0) != m->wide1)
      return gg_fail(GG_ERR_INVALID, m->wide1 ? "the model was created with wide_weights: it needs data created with gg_data_create_wide"
                                              : "the data hold wide (two-limb) weights: the model must be created with wide_weights");
  }
  return GG_OK;
}

static hipEvent_t next_event(gg_model* m) {
  if (m->ev_used == m->ev.size()) {
    hipEvent_t e;
    // (device-scope release only: a timing sample must not make the launch it brackets write its cache lines back to host-visible memory)
    if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) { (void)hipGetLastError(); hipEventCreate(&e); }
    m->ev.push_back(e);
  }
  return m->ev[m->ev_used++];
}

static int32_t* gcnt(const gg_model* m, int buf, int dir) {
  return reinterpret_cast<int32_t*>(m->Gbuf[buf]) + (size_t)dir * m->K * m->Np;
}
static long long* gs(const gg_model* m, int buf, int dir) {
  if (m->is_adj) return nullptr;
  unsigned char* base = m->Gbuf[buf] + (size_t)2 * m->K * m->Np * sizeof(int32_t);
  return reinterpret_cast<long long*>(base) + (size_t)dir * m->K * m->Np * m->NSI;
}

static gg_node_ctx node_ctx(const gg_model* m, const gg_data* d, int buf) {
  gg_node_ctx x;
  x.beta = m->beta;
  x.n0 = m->own.lb < 0 ? m->n0 : 0;      // (block-interleaved shards: the sharded kernels set it per workgroup)
  x.N = m->Nl; x.Np = m->Np; x.K = m->K; x.Kp = m->Kp; x.B = m->B; x.shift = m->is_adj ? 0 : d->shift;
  x.special = m->cfg.special_case_self_conns;
  x.c = m->c; x.cnt = m->cnt; x.cntn = nullptr; x.logm = m->logm; x.T = m->T; x.Tout = m->Tout; x.Tin = m->Tin;
  x.Gcnt[0] = gcnt(m, buf, 0); x.Gcnt[1] = gcnt(m, buf, 1); x.GS[0] = gs(m, buf, 0); x.GS[1] = gs(m, buf, 1);
  x.dA = d->dA; x.dWq = d->dWq;
  return x;
}

// rows per workgroup of the bulk kernel: enough workgroups to fill 256 CUs,
// at most 1020 rows (SWAR byte counters: 255 rows per wave).
static int bulk_rw(const gg_model* m, int strips, int ndirs) {
  static int override_rw = -1;
  if (override_rw < 0) { const char* e = getenv("GG_BULK_RW"); override_rw = e ? atoi(e) : 0; }
  if (override_rw > 0) return override_rw > 1020 ? 1020 : (override_rw + 3) / 4 * 4;
  {
    static int rw_a = -1, rw_w = -1;   // per-half overrides (tuning)
    if (rw_a < 0) { const char* e = getenv("GG_BULK_RW_A"); rw_a = e ? atoi(e) : 0; }
    if (rw_w < 0) { const char* e = getenv("GG_BULK_RW_W"); rw_w = e ? atoi(e) : 0; }
    const int o = m->is_adj ? rw_a : rw_w;
    if (o > 0) return o > 1020 ? 1020 : (o + 3) / 4 * 4;
  }
  (void)strips; (void)ndirs;
  // Rows per workgroup, measured on MI355X inside the whole sweep (round 2, riders of the ahead form, the weight model alone:
  // 128 rows 33.7 us per sweep, 192: 35.0, 256: 31.8 with 8 rows in flight per wave, 384: 34.5, 512: 38.0; the plain kernel
  // of the serial form is within 2 % over 160-256).  Larger N keeps the same number of row slots per label.
  long long rw = (long long)m->N / 16;
  rw = (rw + 3) / 4 * 4;
  if (rw < 16) rw = 16;
  if (rw > 1020) rw = 1020;
  return (int)rw;
}

static int ensure_perm(gg_model* m) {
  if (!m->perm_valid) {
    k_sort_labels<<<1, 1024, 2 * m->K * sizeof(int), m->stream>>>(m->c, m->N, m->K, m->permb[m->pcur], m->labb[m->pcur], m->cnt);
    GG_HIP(hipGetLastError());
    m->perm_valid = true;
  }
  return GG_OK;
}

static int ensure_zero(gg_model* m, int buf) {
  if (!m->zeroed[buf]) {
    GG_HIP(hipMemsetAsync(m->Gbuf[buf], 0, m->Gbytes, m->stream));
    m->zeroed[buf] = true;
  }
  return GG_OK;
}

// Launch shape of a bulk pass over this model's matrices: column strips, label slots, rows per slot, streaming hint.
struct bulk_shape { int strips, nslots, RW; bool ntl; };
static bulk_shape bulk_shape_of(const gg_model* m, int ndirs) {
  bulk_shape b;
  const int SC = m->is_adj ? 1024 : 64 * (4 / m->B);
  b.strips = (m->Np + SC - 1) / SC;
  b.RW = bulk_rw(m, b.strips, ndirs);
  b.nslots = (m->N + b.RW - 1) / b.RW + m->K;
  // non-temporal loads once a pass streams more than the Infinity Cache holds (see gg_stream_load)
  b.ntl = (double)ndirs * m->N * (double)m->Np * (m->is_adj ? 0.125 : 4.0 * m->B) > gg_nt_threshold();
  return b;
}

// BULK statistics under the current labels into buffer `buf`.  dirs: bit 0 = out (reads X^T), bit 1 = in (reads X).
static int pair_sched_on() { const char* e = getenv("GG_PAIR_SCHED"); return e ? atoi(e) : 1; }   // (read at every launch: the tests switch it)
// the adjacency model's pass behind the weight model's pass as well (GG_PAIR_PASS, default 1): it then streams while the weight
// model's scan - latency-bound, the memory system idle - runs, instead of sharing the pass's bandwidth
static int pair_pass_on() { const char* e = getenv("GG_PAIR_PASS"); return e ? atoi(e) : 1; }   // (read at every launch: the tests switch it)
static int pair_adj16_on() { const char* e = getenv("GG_PAIR_ADJ16"); return e ? atoi(e) : 1; }   // (read at every launch: the tests switch it)
static int pair_min_n() {
  static int n = -1;
  if (n < 0) { const char* e = getenv("GG_PAIR_MIN_N"); n = e ? atoi(e) : 16384; }
  return n;
}
// how long the weight model's stream idles behind its scan, in ticks of the 100 MHz wall clock (GG_PAIR_DELAY_US, default 12)
static long long pair_delay_ticks() {
  static long long t = -1;
  if (t < 0) { const char* e = getenv("GG_PAIR_DELAY_US"); t = (long long)((e ? atof(e) : 12.0) * 100.0); }
  return t;
}
__global__ void k_delay(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

static int enqueue_bulk(gg_model* m, const gg_data* d, int buf, int dirs, bool pipelined, uint32_t sweep = 0) {
  const int N = m->N, K = m->K, Np = m->Np;
  int rc;
  hipStream_t st = m->stream;
  if ((rc = ensure_perm(m)) || (rc = ensure_zero(m, buf))) return rc;
  const int32_t* perm = m->permb[m->pcur];
  const int32_t* lab_start = m->labb[m->pcur];
  const int dir0 = (dirs & 1) ? 0 : 1;
  const int ndirs = (dirs == 3) ? 2 : 1;
  const int32_t* ctl = pipelined ? m->rs : nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = m->timing > 0 && (sweep % (uint32_t)m->timing) == 0 && m->ev_used < 4096;
  // pair scheduling (launch_scan): the adjacency model's pass waits for the weight model's parameter step - it then streams under
  // the weight model's scan (before the sampling events: the wait is not the pass)
  const bool pair = pair_sched_on() && pair_pass_on() && m->N >= pair_min_n() && m->Kp == 64 && m->use_scan && !m->sharded && pipelined;   // (pipelined: a sweep's pass)
  if (pair && m->is_adj) { hipEvent_t ev = gg_pair_lookup(d->wpass, sweep); if (ev) GG_HIP(hipStreamWaitEvent(st, ev, 0)); }
  if (timed) { e0 = next_event(m); e1 = next_event(m); GG_HIP(hipEventRecord(e0, st)); }
  const bulk_shape bs = bulk_shape_of(m, ndirs);
  dim3 grd(bs.strips, bs.nslots, ndirs);
  const int RW = bs.RW;
  if (m->is_adj) {
    if (bs.ntl)
      k_bulk_adj<true><<<grd, 256, 0, st>>>(d->AbT, d->Ab, Np, perm, lab_start, K, RW, gcnt(m, buf, 0), gcnt(m, buf, 1), dir0, ctl);
    else
      k_bulk_adj<false><<<grd, 256, 0, st>>>(d->AbT, d->Ab, Np, perm, lab_start, K, RW, gcnt(m, buf, 0), gcnt(m, buf, 1), dir0, ctl);
    if (timed) m->ev_bytes.push_back((double)ndirs * N * (double)N / 8.0);
  } else {
    const bool ntl = bs.ntl;
#define BULKW(BV, UV)                                                                                                 \
  do {                                                                                                                \
    if (ntl) k_bulk_w<BV, UV, true><<<grd, 256, 0, st>>>(d->WqT, d->Wq, Np, perm, lab_start, K, RW, gcnt(m, buf, 0),    \
                                                         gcnt(m, buf, 1), gs(m, buf, 0), gs(m, buf, 1), dir0, ctl);    \
    else k_bulk_w<BV, UV, false><<<grd, 256, 0, st>>>(d->WqT, d->Wq, Np, perm, lab_start, K, RW, gcnt(m, buf, 0),      \
                                                      gcnt(m, buf, 1), gs(m, buf, 0), gs(m, buf, 1), dir0, ctl);      \
  } while (0)
    switch (m->B) {
      case 1: BULKW(1, 4); break;
      case 2: BULKW(2, 4); break;
      case 3: BULKW(3, 4); break;
      case 4: BULKW(4, 4); break;
    }
#undef BULKW
    if (timed) m->ev_bytes.push_back((double)ndirs * N * (double)N * 4.0 * m->B);
  }
  if (timed) {
    m->ev_sweep.push_back(sweep);
    GG_HIP(hipEventRecord(e1, st));
  }
  GG_HIP(hipGetLastError());
  m->zeroed[buf] = false;
  return GG_OK;
}

// The next sweep's bulk pass as riders (gg_block_kernels.h): work items [part == 0: the first ride_split share;
// part == 1: the rest] of the pass over both directions, counted into buffer `nbuf` under the label sort `P`.
static gg_ride_args make_ride(const gg_model* m, const gg_data* d, int nbuf, int P, int part, int front, int* nriders, int* ntl,
                              double* bytes) {
  gg_ride_args r;
  memset(&r, 0, sizeof(r));
  const bulk_shape bs = bulk_shape_of(m, 2);
  r.XT = m->is_adj ? (const void*)d->AbT : (const void*)d->WqT;
  r.X1 = m->is_adj ? (const void*)d->Ab : (const void*)d->Wq;
  r.Np = m->Np; r.K = m->K; r.RW = bs.RW;
  r.perm = m->permb[P]; r.lab_start = m->labb[P];
  r.Gc0 = gcnt(m, nbuf, 0); r.Gc1 = gcnt(m, nbuf, 1); r.Gs0 = gs(m, nbuf, 0); r.Gs1 = gs(m, nbuf, 1);
  r.strips = bs.strips; r.nslots = bs.nslots;
  const long long W = (long long)bs.strips * bs.nslots * 2;
  long long ws = (long long)((double)W * m->ride_split + 0.5);
  ws = ws < 0 ? 0 : (ws > W ? W : ws);
  if (part == 2) ws = W;              // the whole pass in one launch
  r.w0 = (part == 0 || part == 2) ? 0 : (int)ws;
  r.w1 = (part == 0 || part == 2) ? (int)ws : (int)W;
  r.front = front;
  {
    static int cap = -1, cap_a = -1;   // GG_RIDERS: most rider workgroups per launch (a cap below ~512 starves the pass, above it changes nothing: measured)
    if (cap < 0) { const char* e = getenv("GG_RIDERS"); cap = e && atoi(e) > 0 ? atoi(e) : 2048; }
    if (cap_a < 0) { const char* e = getenv("GG_RIDERS_A"); cap_a = e && atoi(e) > 0 ? atoi(e) : 0; }
    // (the adjacency model's pass is small and its chain has slack: with 11/16 of its items as workgroups, each walking
    // one or two, the weight model's riders find their slots sooner - 36.6 against 37.8 us per sweep at N=4096, measured
    // over 128..256; fewer than 5/8 and the adjacency chain becomes the slower one)
    int cp = cap;
    if (m->is_adj) { cp = cap_a > 0 ? cap_a : (r.w1 - r.w0) * 11 / 16; if (cp < 1) cp = 1; if (cp > cap) cp = cap; }
    r.nriders = (r.w1 - r.w0) < cp ? (r.w1 - r.w0) : cp;
  }
  {
    static int dly = -1;
    if (dly < 0) { const char* e = getenv("GG_RIDER_DELAY"); dly = e ? atoi(e) : 0; }
    r.delay = part == 2 ? dly : 0;
  }
  *nriders = r.nriders;
  *ntl = bs.ntl ? 1 : 0;
  if (bytes) *bytes = 2.0 * m->N * (double)m->N * (m->is_adj ? 0.125 : 4.0 * m->B) * (double)(r.w1 - r.w0) / (double)(W > 0 ? W : 1);
  return r;
}

// fused: pair tables (from buffer `buf`) -> parameter draw -> coefficient tables; optionally zero the other buffer
// zero_buf >= 0: that statistics buffer is zeroed by the kernel's workgroups (unless it is known to be zero already);
// ride_next >= 0 (ahead form): the first share of the NEXT sweep's bulk pass rides behind them, counting into that buffer.
static gg_params_args fill_params_args(gg_model* m, const gg_data* d, int buf, bool tables, bool draw, bool prep, int zero_buf,
                                       uint32_t sweep, int pidx) {
  gg_params_args g;
  memset(&g, 0, sizeof(g));
  g.is_adj = m->is_adj ? 1 : 0; g.K = m->K; g.Kp = m->Kp; g.B = m->Bx; g.NSI = m->NSI; g.Np = m->Np; g.N = m->N;   // (g.B: the parameters' dimension - the kernels' template argument is the statistics')
  g.n0 = m->n0; g.Nl = m->Nl;
  if (m->sharded) g.n0 = 0;      // (the pair tables walk perm_loc: local indices, -1 for another rank's node)
  g.do_tables = tables; g.do_draw = draw; g.do_prep = prep;
  g.add_diag = m->cfg.special_case_self_conns ? 0 : 1;
  g.Gcnt0 = gcnt(m, buf, 0); g.GS0 = gs(m, buf, 0);
  g.perm = m->sharded ? m->perm_loc : m->permb[pidx]; g.lab_start = m->labb[pidx]; g.cnt = m->cnt;
  g.dA = d ? d->dA : nullptr; g.dWq = d ? d->dWq : nullptr;
  g.tab = m->tab;
  g.ah.tau1 = m->cfg.tau1; g.ah.tau0 = m->cfg.tau0;
  g.ah.special = m->cfg.special_case_self_conns;
  g.ah.bb = (m->kind == GG_BB_ADJ);
  g.ah.ngraphs = m->ngraphs;
  if (d) { g.ah.d1 = d->d1; g.ah.d0 = (long long)m->N * m->ngraphs - d->d1; }
  if (!m->is_adj) {
    g.wh.niw = m->niw; g.wh.niw_self = m->niw_self;
    g.wh.special = m->cfg.special_case_self_conns;
    if (d) {
      g.wh.shift = d->shift; g.wh.d1 = d->d1;
      for (int i = 0; i < gg_nsi(GG_MAXB); ++i) g.wh.dS[i] = d->dS[i];
    }
  }
  g.seed = m->cfg.seed; g.sweep = sweep;
  g.m = m->m; g.logm = m->logm; g.p = m->p; g.p_self = m->p_self; g.mu = m->mu; g.sigma = m->sigma;
  g.mu_self = m->mu_self; g.sigma_self = m->sigma_self; g.T = m->T; g.Tout = m->Tout; g.Tin = m->Tin;
  g.ctl = m->rs;
  g.verdict = m->verdict;
  g.bar = m->bar;
  // (the serial form's parameter step in front of a resident scan: see gg_params_args::ap)
  g.ap = (prep && m->use_scan && m->bar && !m->sharded) ? m->ap : nullptr; g.ap_n = m->N;
  g.pi = m->pi;
  g.beta = m->beta;
  g.sweep_from_ctl = 0;
  g.tables_from_tab = m->shard_from_tab ? 1 : 0;
  if (m->m_pending && prep && !m->wide) {     // (wide: k_draw_m, launched by enqueue_params - the in-kernel draw is one wave of 64)
    g.draw_m = 1; g.m_site = m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M; g.m_sweep = m->m_sweep;
    m->m_pending = false;
  }
  if (zero_buf >= 0 && !m->zeroed[zero_buf]) {
    g.zero_ptr = reinterpret_cast<unsigned long long*>(m->Gbuf[zero_buf]);
    g.zero_words = m->Gbytes / 8;
    m->zeroed[zero_buf] = true;
  }
  g.shared = (m->kind == GG_SBM_GAUSS_SHCOV_W) ? 1 : 0;
  g.csc = m->csc;
  return g;
}

static int enqueue_params(gg_model* m, const gg_data* d, int buf, bool tables, bool draw, bool prep, int zero_buf,
                          uint32_t sweep, int pidx = -1, int ride_next = -1, int ride_part = 0) {
  if (pidx < 0) pidx = m->pcur;
  if (m->wide && m->m_pending && prep) {
    // resample_m of the previous sweep (deferred to this parameter step, which takes log m from it): more than 64 blocks do
    // not fit the one wave the parameter kernel has for it
    k_draw_m<<<1, 256, 256 * sizeof(double), m->stream>>>(m->cnt, m->pi, m->K, m->cfg.seed,
                                                          m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M, m->m_sweep, m->m, m->rs);
    GG_HIP(hipGetLastError());
    m->m_pending = false;
  }
  if (m->sharded && tables) {
    if (!m->perm_loc) { int rcp = dmalloc(&m->perm_loc, (size_t)m->N); if (rcp) return rcp; }
    k_perm_local<<<(m->N + 255) / 256, 256, 0, m->stream>>>(m->permb[pidx], m->perm_loc, m->N, m->own);
    GG_HIP(hipGetLastError());
  }
  gg_params_args g = fill_params_args(m, d, buf, tables, draw, prep, zero_buf, sweep, pidx);
  // more than 64 blocks (plain SBM kinds): a wave per block pair, four pairs per workgroup, log m in the same launch
  static int pw_on = -1;
  if (pw_on < 0) { const char* e = getenv("GG_PARAMS_W"); pw_on = e ? atoi(e) : 1; }
  const bool wave_pairs = pw_on && m->wide && !m->joint && !g.shared;
  const int nwg = wave_pairs ? (m->K * m->K + 3) / 4 + 1 : m->K * m->K + 1;
  if (ride_next >= 0) {
    int nr, ntl;
    double bytes;
    const gg_ride_args r = make_ride(m, d, ride_next, pidx, ride_part, nwg, &nr, &ntl, &bytes);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool timed = m->timing > 0 && (sweep % (uint32_t)m->timing) == 0 && m->ev_used < 4096;
    if (timed) { e0 = next_event(m); e1 = next_event(m); GG_HIP(hipEventRecord(e0, m->stream)); }
#define PRM(ADJV, BV)                                                                          \
  do {                                                                                         \
    if (wave_pairs) {                                                                          \
      if (ntl) k_params_w<ADJV, BV, 2><<<nwg + nr, 256, 0, m->stream>>>(g, r);                 \
      else k_params_w<ADJV, BV, 1><<<nwg + nr, 256, 0, m->stream>>>(g, r);                     \
    } else if (ntl) k_params<ADJV, BV, 2><<<nwg + nr, 256, 0, m->stream>>>(g, r);              \
    else k_params<ADJV, BV, 1><<<nwg + nr, 256, 0, m->stream>>>(g, r);                         \
  } while (0)
    if (m->is_adj) PRM(true, 1);
    else if (m->B == 1) PRM(false, 1);
    else if (m->B == 2) PRM(false, 2);
    else if (m->B == 3) PRM(false, 3);
    else PRM(false, 4);
#undef PRM
    if (timed) { m->ev_bytes.push_back(bytes); m->ev_sweep.push_back(sweep); GG_HIP(hipEventRecord(e1, m->stream)); }
    m->zeroed[ride_next] = false;
  } else if (wave_pairs) {
    if (m->is_adj) k_params_w<true, 1><<<nwg, 256, 0, m->stream>>>(g);
    else if (m->B == 1) k_params_w<false, 1><<<nwg, 256, 0, m->stream>>>(g);
    else if (m->B == 2) k_params_w<false, 2><<<nwg, 256, 0, m->stream>>>(g);
    else if (m->B == 3) k_params_w<false, 3><<<nwg, 256, 0, m->stream>>>(g);
    else k_params_w<false, 4><<<nwg, 256, 0, m->stream>>>(g);
  } else if (m->is_adj) k_params<true, 1><<<nwg, 256, 0, m->stream>>>(g);
  else if (m->joint && m->B == 1) k_params<false, 1, 0, true><<<nwg, 256, 0, m->stream>>>(g);
  else if (m->joint) k_params<false, 2, 0, true><<<nwg, 256, 0, m->stream>>>(g);
  else if (m->B == 1) k_params<false, 1><<<nwg, 256, 0, m->stream>>>(g);
  else if (m->B == 2) k_params<false, 2><<<nwg, 256, 0, m->stream>>>(g);
  else if (m->B == 3) k_params<false, 3><<<nwg, 256, 0, m->stream>>>(g);
  else k_params<false, 4><<<nwg, 256, 0, m->stream>>>(g);
  GG_HIP(hipGetLastError());
  if (m->wide && prep && !wave_pairs) {     // (k_params_w forms log m itself)
    k_logm<<<1, 256, 0, m->stream>>>(m->m, m->logm, m->K, m->rs);
    GG_HIP(hipGetLastError());
  }
  if (g.shared && draw) {   // weights.py:545-552: the shared covariance and, with it, the coefficient tables
    if (m->B == 1) k_shared_cov<1><<<1, 256, 0, m->stream>>>(g);
    else if (m->B == 2) k_shared_cov<2><<<1, 256, 0, m->stream>>>(g);
    else if (m->B == 3) k_shared_cov<3><<<1, 256, 0, m->stream>>>(g);
    else k_shared_cov<4><<<1, 256, 0, m->stream>>>(g);
    GG_HIP(hipGetLastError());
  }
  return GG_OK;
}

static int launch_speculate(gg_model* m, const gg_data* d, int buf, int n_begin, int lp_node, uint32_t sweep, bool pipelined,
                            int round_window = 0, int explicit_count = 0) {
  gg_node_ctx x = node_ctx(m, d, buf);
  const int K = m->K, per = 256 / m->Kp;
  const int count = round_window > 0 ? round_window : (explicit_count > 0 ? explicit_count : (m->Nl - n_begin));
  int nb = (count + per - 1) / per;
  if (nb < 1) nb = 1;
  uint32_t site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C;
  double* lpo = lp_node >= 0 ? m->lp_out : nullptr;
  const int32_t* ctl = (pipelined || round_window > 0) ? m->rs : nullptr;
  size_t lds = (size_t)2 * K * per * m->NT * sizeof(double);
  const int stage = (lds + table_bytes(m) <= 64 * 1024) ? 1 : 0;
  if (stage) lds += table_bytes(m);
  else if (m->Kp == 64) lds += (size_t)4 * 2 * m->NT * 64 * sizeof(double);   // gg_lp_split4's meeting place
  if (m->Kp > 64) lds += (size_t)2 * 256 * sizeof(double) + 16 * sizeof(int) + 8;   // the cross-wave categorical's scratch
#define SPEC(NTV, ADJV)                                                                                                    \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set && lds > 48 * 1024) {                                                                                    \
      GG_HIP(hipFuncSetAttribute((const void*)k_speculate<NTV, ADJV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    k_speculate<NTV, ADJV><<<nb, 256, lds, m->stream>>>(x, n_begin, m->cfg.seed, site, sweep, m->cand, lp_node, lpo, ctl,    \
                                                        stage, round_window > 0 ? 1 : 0, m->ap);                          \
  } while (0)
  if (m->is_adj) SPEC(2, true);
  else if (m->joint && m->B == 1) SPEC(4, false);
  else if (m->joint) SPEC(7, false);
  else if (m->B == 1) SPEC(3, false);
  else if (m->B == 2) SPEC(6, false);
  else if (m->B == 3) SPEC(10, false);
  else SPEC(15, false);
#undef SPEC
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static gg_resolve_args fill_spec_args(gg_model* m, const gg_data* d, int buf, uint32_t sweep, int pw) {
  gg_resolve_args g;
  memset(&g, 0, sizeof(g));
  g.x = node_ctx(m, d, buf);
  g.c = m->c; g.cnt = m->cnt; g.cand = m->cand; g.rs = m->rs; g.L = m->L;
  g.tail_m = 0; g.tail_sort = 1;
  g.m = m->m; g.pi = m->pi; g.perm = m->permb[pw]; g.lab_start = m->labb[pw];
  g.perm_prev = m->permb[1 - pw]; g.lab_start_prev = m->labb[1 - pw];
  g.flush_on_change = 0; g.sweep_from_ctl = 0; g.round_mode = 0; g.window = 0; g.stage_tables = 0;
  g.site_m = m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M;
  g.Lmax = m->Lmax; g.max_idle = m->max_idle; g.CH = m->CH; g.fresh = 1;
  g.XA = d->Ab; g.XAT = d->AbT; g.XW = d->Wq; g.XWT = d->WqT;
  g.seed = m->cfg.seed; g.site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C; g.sweep = sweep;
  g.dbg = nullptr;
  g.keep_L = 0;
  g.verdict = m->verdict;
  return g;
}

// The quiet chain's sweep in one launch (k_sweep_q): scan of sweep `sweep` on buffer `a` | parameter step of sweep + 1
// on buffer `b` (gated on the scan's verdict) | riders counting the statistics of sweep + 2 into buffer `c`.
static int launch_sweep_q(gg_model* m, const gg_data* d, int a, int b, int c, uint32_t sweep, int P) {
  gg_resolve_args g = fill_spec_args(m, d, a, sweep, 1 - P);
  // (the parameter step and the riders read the label sort the scan starts from: the one it ends with is being written
  // by this very launch and is the same iff the scan changes nothing - the only case in which either result is used)
  gg_params_args pg = fill_params_args(m, d, b, true, true, true, a /* zeroed after the verdict, for the next launch's riders */,
                                       sweep + 1u, P);
  const int per = 256 / m->Kp;
  const int nspec = (m->Nl + per - 1) / per;
  const int nparams = m->K * m->K + 1;
  int stage;
  const size_t lds = lds_spec(m, &stage);
  int nr = 0, ntl = 0;
  double bytes = 0;
  gg_ride_args r = make_ride(m, d, c, P, 2, nspec + nparams, &nr, &ntl, &bytes);
  if (m->swq_dbg) {
    const long long init[8] = {0x7fffffffffffffffLL, 0, 0, 0, 0x7fffffffffffffffLL, 0, 0, 0};
    GG_HIP(hipMemcpyAsync(m->swq_dbg, init, sizeof(init), hipMemcpyHostToDevice, m->stream));
    r.dbg = m->swq_dbg;
  }
  const bool split64 = !stage && m->Kp == 64;
  m->vtag += 1u;
  if (m->vtag == 0u || m->vtag >= 0x7fffffffu) m->vtag = 1u;
  const unsigned vtag = m->vtag;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = m->timing > 0 && (sweep % (uint32_t)m->timing) == 0 && m->ev_used < 4096;
  if (timed) { e0 = next_event(m); e1 = next_event(m); GG_HIP(hipEventRecord(e0, m->stream)); }
  // (the fused kernel is compiled in a translation unit of its own, gg_sweepq.hip: its rider loop's schedule then depends on
  // nothing that changes in this file)
  {
    const int rcq = gg_launch_sweep_q(m->is_adj ? 2 : m->NT, m->is_adj ? 1 : 0, ntl ? 1 : 0, split64 ? 1 : 0, nspec + nparams + nr, lds, m->stream,
                                      g, stage, m->ap, pg, r, nspec, vtag);
    if (rcq) return rcq;
  }
  if (timed) { m->ev_bytes.push_back(bytes); m->ev_sweep.push_back(sweep); GG_HIP(hipEventRecord(e1, m->stream)); }
  GG_HIP(hipGetLastError());
  if (pg.shared) {
    // weights.py:545-552: the shared covariance of sweep + 1 and, with it, the coefficient tables - the second half of the
    // parameter step, as behind every k_params (enqueue_params).  A no-op behind a scan that changed a label (stall flag),
    // like the gated first half inside the launch.
    if (m->B == 1) k_shared_cov<1><<<1, 256, 0, m->stream>>>(pg);
    else if (m->B == 2) k_shared_cov<2><<<1, 256, 0, m->stream>>>(pg);
    else if (m->B == 3) k_shared_cov<3><<<1, 256, 0, m->stream>>>(pg);
    else k_shared_cov<4><<<1, 256, 0, m->stream>>>(pg);
    GG_HIP(hipGetLastError());
  }
  m->zeroed[c] = false;
  return GG_OK;
}

// k_speculate + the common case of k_resolve in one launch (k_spec_resolve): the speculative form of the adaptive scan.
// ride_next >= 0 (ahead form): the rest of the next sweep's bulk pass rides behind its workgroups (label sort 1 - pw, the
// one this sweep started with).
static int launch_spec_resolve(gg_model* m, const gg_data* d, int buf, uint32_t sweep, int pw, int ride_next = -1) {
  gg_resolve_args g = fill_spec_args(m, d, buf, sweep, pw);
  const int per = spec_nodes_per_wg(m);
  const int nb = (m->Nl + per - 1) / per;
  int stage;
  const size_t lds = lds_spec(m, &stage);
  int nr = 0, ntl = 0;
  double bytes = 0;
  gg_ride_args r;
  memset(&r, 0, sizeof(r));
  if (ride_next >= 0) r = make_ride(m, d, ride_next, 1 - pw, 1, nb, &nr, &ntl, &bytes);
  const int ride = ride_next < 0 ? 0 : (ntl ? 2 : 1);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = ride && m->timing > 0 && (sweep % (uint32_t)m->timing) == 0 && m->ev_used < 4096;
  if (timed) { e0 = next_event(m); e1 = next_event(m); GG_HIP(hipEventRecord(e0, m->stream)); }
  const bool split64 = !stage && m->Kp == 64;   // the K > 32 conditional (gg_lp_split4): its own instantiation
#define SPR1(NTV, ADJV, RV, SV)                                                                                            \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set && lds > 40 * 1024) {                                                                                    \
      GG_HIP(hipFuncSetAttribute((const void*)k_spec_resolve<NTV, ADJV, RV, SV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    k_spec_resolve<NTV, ADJV, RV, SV><<<nb + nr, 256, lds, m->stream>>>(g, stage, m->ap, r);                                \
  } while (0)
#define SPR(NTV, ADJV)                                                                                                     \
  do {                                                                                                                     \
    if (m->Kp > 64) {                                                                                                      \
      if (ride == 0) SPR1(NTV, ADJV, 0, 2);                                                                                \
      else if (ride == 1) SPR1(NTV, ADJV, 1, 2);                                                                           \
      else SPR1(NTV, ADJV, 2, 2);                                                                                          \
    } else if (split64) {                                                                                                  \
      if (ride == 0) SPR1(NTV, ADJV, 0, 1);                                                                                \
      else if (ride == 1) SPR1(NTV, ADJV, 1, 1);                                                                           \
      else SPR1(NTV, ADJV, 2, 1);                                                                                          \
    } else {                                                                                                               \
      if (ride == 0) SPR1(NTV, ADJV, 0, 0);                                                                                \
      else if (ride == 1) SPR1(NTV, ADJV, 1, 0);                                                                           \
      else SPR1(NTV, ADJV, 2, 0);                                                                                          \
    }                                                                                                                      \
  } while (0)
  if (m->is_adj) SPR(2, true);
  else if (m->joint && m->B == 1) SPR(4, false);
  else if (m->joint) SPR(7, false);
  else if (m->B == 1) SPR(3, false);
  else if (m->B == 2) SPR(6, false);
  else if (m->B == 3) SPR(10, false);
  else SPR(15, false);
#undef SPR
#undef SPR1
  if (timed) { m->ev_bytes.push_back(bytes); m->ev_sweep.push_back(sweep); GG_HIP(hipEventRecord(e1, m->stream)); }
  GG_HIP(hipGetLastError());
  if (ride_next >= 0) m->zeroed[ride_next] = false;
  return GG_OK;
}

static bool ll_is_host_form(const gg_model* m) {
  // (the one combination whose diagonal needs per-node terms, adjacency.py:525-526 without the special case: host closed form)
  return m->is_adj && !m->cfg.allow_self_connections && m->kind == GG_SBM_ADJ && !m->cfg.special_case_self_conns;
}

static gg_ll_args fill_ll_args(const gg_model* m, const gg_data* d, double* out) {
  gg_ll_args g;
  memset(&g, 0, sizeof(g));
  g.is_adj = m->is_adj ? 1 : 0; g.K = m->K; g.B = m->B; g.NSI = m->NSI; g.NTAB = m->NTAB;
  g.special = m->cfg.special_case_self_conns ? 1 : 0;
  g.shift = m->is_adj ? 0 : d->shift;
  g.tab = m->tab; g.cnt = m->cnt;
  g.p = m->p; g.p_self = m->p_self; g.mu = m->mu; g.sigma = m->sigma; g.mu_self = m->mu_self; g.sigma_self = m->sigma_self;
  g.d1 = d->d1; g.d0 = (long long)m->N - d->d1;
  for (int i = 0; i < gg_nsi(GG_MAXB); ++i) g.dS[i] = d->dS[i];
  g.diag_zero = (m->is_adj && !m->cfg.allow_self_connections && m->kind == GG_SBM_ADJ) ? 1 : 0;
  g.joint = m->joint ? 1 : 0;
  g.wide = m->wide1 ? 1 : 0;
  g.out = out;
  return g;
}

// The epilogue form fits: a plain SBM half (not joint, not the shared-covariance kind, one graph), unsharded.
static bool epi_ok(const gg_model* m) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("GG_EPILOGUE"); on = e ? atoi(e) : 1; }
  return on && m->sbm && m->K > 1 && !m->joint && !m->sharded && m->kind != GG_SBM_GAUSS_SHCOV_W && m->ngraphs == 1;
}

// The call's last sweep in the speculative form: k_spec_resolve_epi = the scan | (when the caller keeps asking for it) the
// log-likelihood of the state the sweep leaves, from the statistics in buffer `nbuf` | resample_m | the host's copy of the
// control block (gg_resolve_kernels.h).  P: the label sort the scan starts from.
static int launch_spec_resolve_epi(gg_model* m, const gg_data* d, int buf, int nbuf, uint32_t sweep, int P) {
  gg_resolve_args g = fill_spec_args(m, d, buf, sweep, 1 - P);
  gg_params_args pg = fill_params_args(m, d, nbuf, true, false, false, -1, sweep, P);
  const bool want_ll = m->ll_asked && m->calls_since_ll == m->ll_period && !ll_is_host_form(m);
  gg_ll_args la = fill_ll_args(m, d, nullptr);
  gg_epi_args ep;
  memset(&ep, 0, sizeof(ep));
  ep.terms = m->epi_terms; ep.ticket = m->epi_words; ep.bad = m->epi_words + 1; ep.out = m->h_epi;
  ep.want_ll = want_ll ? 1 : 0;
  ep.npair = want_ll ? m->K * m->K : 1;
  ep.m_site = m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M;
  const int per = spec_nodes_per_wg(m);
  const int nspec = (m->Nl + per - 1) / per;
  int stage;
  const size_t lds = lds_spec(m, &stage);
  const bool split64 = !stage && m->Kp == 64;
  m->vtag += 1u;
  if (m->vtag == 0u || m->vtag >= 0x7fffffffu) m->vtag = 1u;
  const unsigned vtag = m->vtag;
#define SPE1(NTV, ADJV, SV)                                                                                                \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set && lds > 40 * 1024) {                                                                                    \
      GG_HIP(hipFuncSetAttribute((const void*)k_spec_resolve_epi<NTV, ADJV, SV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    k_spec_resolve_epi<NTV, ADJV, SV><<<nspec + ep.npair + 1, 256, lds, m->stream>>>(g, stage, m->ap, pg, la, ep, nspec, vtag); \
  } while (0)
#define SPE(NTV, ADJV)                                                                                                     \
  do {                                                                                                                     \
    if (m->Kp > 64) SPE1(NTV, ADJV, 2);                                                                                    \
    else if (split64) SPE1(NTV, ADJV, 1);                                                                                  \
    else SPE1(NTV, ADJV, 0);                                                                                               \
  } while (0)
  if (m->is_adj) SPE(2, true);
  else if (m->B == 1) SPE(3, false);
  else if (m->B == 2) SPE(6, false);
  else if (m->B == 3) SPE(10, false);
  else SPE(15, false);
#undef SPE
#undef SPE1
  GG_HIP(hipGetLastError());
  m->epi_tag = vtag;
  m->epi_ll = want_ll;
  return GG_OK;
}

// The tail of a completed scan sorts the final labels into perm buffer `pw` (default: the other one).
static int launch_resolve(gg_model* m, const gg_data* d, int buf, uint32_t sweep, bool tail_m, bool tail_sort,
                          int pw = -1, int round_window = 0) {
  if (pw < 0) pw = 1 - m->pcur;
  gg_resolve_args g;
  g.x = node_ctx(m, d, buf);
  g.c = m->c; g.cnt = m->cnt; g.cand = m->cand; g.rs = m->rs; g.L = m->L;
  g.tail_m = tail_m ? 1 : 0; g.tail_sort = tail_sort ? 1 : 0;
  g.m = m->m; g.pi = m->pi; g.perm = m->permb[pw]; g.lab_start = m->labb[pw];
  g.perm_prev = m->permb[1 - pw]; g.lab_start_prev = m->labb[1 - pw];
  g.flush_on_change = 0;
  g.sweep_from_ctl = 0;
  g.round_mode = round_window > 0 ? 1 : 0;
  g.window = round_window;
  g.stage_tables = m->resolve_stage ? 1 : 0;
  g.site_m = m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M;
  g.Lmax = m->Lmax; g.max_idle = m->max_idle; g.CH = m->CH; g.fresh = 1;
  g.XA = d->Ab; g.XAT = d->AbT; g.XW = d->Wq; g.XWT = d->WqT;
  g.seed = m->cfg.seed; g.site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C; g.sweep = sweep;
  g.dbg = nullptr;
  g.keep_L = 0;
  size_t lds = m->resolve_lds;
#define RES(NTV, ADJV)                                                                                           \
  do {                                                                                                           \
    static bool attr_set = false;                                                                                \
    if (!attr_set && lds > 48 * 1024) {                                                                          \
      GG_HIP(hipFuncSetAttribute((const void*)k_resolve<NTV, ADJV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      attr_set = true;                                                                                           \
    }                                                                                                            \
    k_resolve<NTV, ADJV><<<1, 1024, lds, m->stream>>>(g);                                                        \
  } while (0)
  if (m->is_adj) RES(2, true);
  else if (m->joint && m->B == 1) RES(4, false);
  else if (m->joint) RES(7, false);
  else if (m->B == 1) RES(3, false);
  else if (m->B == 2) RES(6, false);
  else if (m->B == 3) RES(10, false);
  else RES(15, false);
#undef RES
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int launch_apply_L(gg_model* m, const gg_data* d, int buf, int mode = 0, bool reset = true) {
  gg_node_ctx x = node_ctx(m, d, buf);
  int nb = (m->Nl + 255) / 256;
  if (mode == 3) {      // the scan's change list moved in the statistics counted ahead: entries dealt out over blockIdx.y, atomics
    int ny = (int)(m->chg_per_sweep > 0.0 ? m->chg_per_sweep / 3.0 : 4.0);
    ny = ny < 2 ? 2 : (ny > 32 ? 32 : ny);
    dim3 grd(nb, ny);
#define APM(NTV, ADJV)                                                                                             \
    k_apply_moves<NTV, ADJV><<<grd, 256, 0, m->stream>>>(x, m->L, m->rs, gcnt(m, buf, 0), gcnt(m, buf, 1), gs(m, buf, 0),  \
                                                         gs(m, buf, 1), d->Ab, d->AbT, d->Wq, d->WqT)
    if (m->is_adj) APM(2, true);
    else if (m->joint && m->B == 1) APM(4, false);
    else if (m->joint) APM(7, false);
    else if (m->B == 1) APM(3, false);
    else if (m->B == 2) APM(6, false);
    else if (m->B == 3) APM(10, false);
    else APM(15, false);
#undef APM
    GG_HIP(hipGetLastError());
    return GG_OK;
  }
#define APL(NTV, ADJV)                                                                                             \
  k_apply_L<NTV, ADJV><<<nb, 256, 0, m->stream>>>(x, m->L, m->rs, m->ap, gcnt(m, buf, 0), gcnt(m, buf, 1),          \
                                                  gs(m, buf, 0), gs(m, buf, 1), d->Ab, d->AbT, d->Wq, d->WqT, mode)
  if (m->is_adj) APL(2, true);
  else if (m->joint && m->B == 1) APL(4, false);
  else if (m->joint) APL(7, false);
  else if (m->B == 1) APL(3, false);
  else if (m->B == 2) APL(6, false);
  else if (m->B == 3) APL(10, false);
  else APL(15, false);
#undef APL
  GG_HIP(hipGetLastError());
  if (reset) {
    k_reset_L<<<1, 64, 0, m->stream>>>(m->rs, mode);
    GG_HIP(hipGetLastError());
  }
  return GG_OK;
}

// One restart round (k_round): lazy apply + re-speculation of the next window + resolution, one launch.
static int launch_round(gg_model* m, const gg_data* d, int buf, uint32_t sweep, bool tail_m, bool tail_sort, int pw,
                        int window, int round_id) {
  gg_resolve_args g;
  g.x = node_ctx(m, d, buf);
  g.c = m->c; g.cnt = m->cnt; g.cand = m->cand; g.rs = m->rs; g.L = m->L;
  g.tail_m = tail_m ? 1 : 0; g.tail_sort = tail_sort ? 1 : 0;
  g.m = m->m; g.pi = m->pi; g.perm = m->permb[pw]; g.lab_start = m->labb[pw];
  g.perm_prev = m->permb[1 - pw]; g.lab_start_prev = m->labb[1 - pw];
  g.flush_on_change = 0; g.sweep_from_ctl = 0; g.round_mode = 1;
  g.window = window;
  g.stage_tables = 0;
  g.site_m = m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M;
  g.Lmax = m->Lcap; g.max_idle = m->max_idle; g.CH = m->CH; g.fresh = 0;
  g.XA = d->Ab; g.XAT = d->AbT; g.XW = d->Wq; g.XWT = d->WqT;
  g.seed = m->cfg.seed; g.site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C; g.sweep = sweep;
  g.dbg = m->dbg;
  g.keep_L = 0;
  const int per = 256 / m->Kp;
  int nb = (window + per - 1) / per;
  if (nb > 512) nb = 512;   // groups beyond the grid are strided over
  int stage;
  const size_t lds = lds_round(m, &stage);
  if (lds > GG_LDS_LIMIT)
    return gg_fail(GG_ERR_UNSUPPORTED, "restart rounds need %zu KB of LDS for K=%d, B=%d", lds / 1024, m->K, m->B);
#define RND(NTV, ADJV)                                                                                            \
  do {                                                                                                            \
    static bool attr_set = false;                                                                                 \
    if (!attr_set && lds > 48 * 1024) {                                                                           \
      GG_HIP(hipFuncSetAttribute((const void*)k_round<NTV, ADJV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    k_round<NTV, ADJV><<<nb, 256, lds, m->stream>>>(g, m->ap, gcnt(m, buf, 0), gcnt(m, buf, 1), gs(m, buf, 0),      \
                                                    gs(m, buf, 1), stage, round_id);                              \
  } while (0)
  if (m->is_adj) RND(2, true);
  else if (m->joint && m->B == 1) RND(4, false);
  else if (m->joint) RND(7, false);
  else if (m->B == 1) RND(3, false);
  else if (m->B == 2) RND(6, false);
  else if (m->B == 3) RND(10, false);
  else RND(15, false);
#undef RND
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int read_ctl(gg_model* m) {
  GG_HIP(hipMemcpyAsync(m->h_rs, m->rs, 24 * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  return GG_OK;
}

// The control block at the end of a batch: from the record the batch's epilogue launch left in pinned memory when there is
// one with this batch's tag on it (no copy to enqueue, one stream synchronisation), else copied.
static int fetch_ctl(gg_model* m) {
  m->epi_hit = false;
  if (m->epi_tag) {
    // The record is written to pinned memory by the launch itself, its tag last (behind a system-scope fence): the host
    // watches for the tag instead of waiting for the stream - a stream synchronisation returns some 10 us after the device
    // is done.  Once the tag is there, nothing in flight reads the network any more (every workgroup that does has drawn
    // its ticket before the record is written; what may still run is resample_m).  A launch that leaves no record (the
    // scan changed a label, or the launch was a no-op behind a stalled one) is noticed by the stream running empty.
    const volatile double* rec = m->h_epi;
    for (unsigned spin = 1;; ++spin) {
      if ((unsigned)rec[2] == m->epi_tag) break;
      if ((spin & 4095u) == 0u) {
        if (hipStreamQuery(m->stream) != hipErrorNotReady) break;   // (drained, or failed: read_ctl reports it)
        if (spin > (1u << 16)) std::this_thread::yield();           // (a long batch: do not hold the core against other threads)
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if ((unsigned)m->h_epi[2] == m->epi_tag) {
      memcpy(m->h_rs, m->h_epi + 4, 24 * sizeof(int32_t));
      m->epi_hit = true;
      return GG_OK;
    }
  }
  return read_ctl(m);
}

static int write_ctl_word(gg_model* m, int idx, int32_t v) {
  GG_HIP(hipMemcpyAsync(m->rs + idx, &v, sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  return GG_OK;
}

// The whole scan in one launch (k_scan: resident grid, one barrier round per accepted change).
// ride_buf >= 0: the NEXT sweep's statistics are being counted under the labels this scan starts with, into that buffer - by
// riders of this very launch (riders_here), or already, by the riders of this sweep's parameter step (the scan then only
// needs the buffer for the one case the change list cannot carry: a list that overflows inside the scan is flushed into it).
static int launch_scan(gg_model* m, const gg_data* d, int buf, uint32_t sweep, bool tail_sort, int pw, bool resume = false, int ride_buf = -1,
                       bool record = false, bool riders_here = true) {
  gg_scan_args a;
  memset(&a, 0, sizeof(a));
  if (record) {    // the call's last scan leaves the control block in pinned memory (the host watches for the tag)
    m->vtag += 1u;
    if (m->vtag == 0u || m->vtag >= 0x7fffffffu) m->vtag = 1u;
    a.epi_out = m->h_epi; a.epi_tag = m->vtag;
    m->epi_tag = m->vtag; m->epi_ll = false;
  }
  a.resume = resume ? 1 : 0;
  a.spin_limit = m->scan_spin;
  gg_resolve_args& g = a.r;
  g.x = node_ctx(m, d, buf);
  g.c = m->c; g.cnt = m->cnt; g.cand = m->cand; g.rs = m->rs; g.L = m->L;
  g.tail_m = 0; g.tail_sort = tail_sort ? 1 : 0;
  g.m = m->m; g.pi = m->pi; g.perm = m->permb[pw]; g.lab_start = m->labb[pw];
  g.perm_prev = m->permb[1 - pw]; g.lab_start_prev = m->labb[1 - pw];
  g.flush_on_change = 0; g.sweep_from_ctl = 0; g.round_mode = 0; g.window = 0;
  g.site_m = m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M;
  g.Lmax = m->Lcap; g.max_idle = m->max_idle; g.CH = m->CH; g.fresh = 1;
  g.XA = d->Ab; g.XAT = d->AbT; g.XW = d->Wq; g.XWT = d->WqT;
  g.seed = m->cfg.seed; g.site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C; g.sweep = sweep;
  g.dbg = m->dbg;
  g.keep_L = 0;
  a.ap = m->ap; a.Gc0 = gcnt(m, buf, 0); a.Gs0 = gs(m, buf, 0); a.bar = m->bar;
  a.Lcap = m->Lcap < 1024 ? m->Lcap : 1024;   // every workgroup keeps its own copy of the list in LDS (12 KB)
  // (with riders every workgroup of the launch is given the scan's LDS: a shorter list - flushed more often in a burn-in
  // sweep, never in a steady one - lets a third workgroup onto the CU)
  if (ride_buf >= 0 && riders_here && a.Lcap > 256) a.Lcap = 256;
  // 33..64 blocks: the weight model's group holds 16 nodes (gg_lp_mfma) - 116 KB of statistics and hypotheses; with a list of 256
  // entries its workgroup (116 KB at K = 64, B = 1) and the adjacency model's (38 KB) still fit one CU's 160 KB TOGETHER, which
  // is what keeps two resident grids launched at the same moment from starving each other (each CU can take one of each,
  // whatever the order the dispatcher places them in)
  if (m->Kp == 64 && a.Lcap > 256) a.Lcap = 256;
  { const char* el = getenv("GG_SCAN_LCAP"); if (el && atoi(el) >= 2 && atoi(el) < a.Lcap) a.Lcap = atoi(el); }   // (tests: flush the list often)
  // (pair scheduling, below: the adjacency model's grid is never resident next to the weight model's - 16 nodes per group, MODE 3)
  // (33..64 blocks only: there a scan is a millisecond of rounds; with 16 blocks it is a tenth of the pass and ordering the two
  // chains costs more than their meeting - N=32768, K=16: 600 against 610 sweeps/s)
  const bool pair = pair_sched_on() && m->N >= pair_min_n() && m->Kp == 64 && !resume && ride_buf < 0 && !m->sharded;
  const bool adj16 = pair && m->is_adj && m->Kp == 64 && pair_adj16_on();
  const int per = adj16 ? 4 * GG_SCAN_ADJ_PW : scan_nodes_per_wg(m);
  const int ngroups = (m->Nl + per - 1) / per;
  a.G = ngroups < GG_SCAN_MAXWG ? ngroups : GG_SCAN_MAXWG;
  int stage;
  size_t lds = lds_scan(m, a.Lcap, &stage, adj16 ? 4 * GG_SCAN_ADJ_PW : 0);
  if (lds > GG_LDS_LIMIT)
    return gg_fail(GG_ERR_UNSUPPORTED, "the resident-grid scan needs %zu KB of LDS for K=%d, B=%d", lds / 1024, m->K, m->B);
  g.stage_tables = stage;
  a.mf_small = (stage == 1 && scan_mf_small(m)) ? 1 : 0;
  int nr = 0, ntl = 0;
  double ride_bytes = 0.0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool timed = false;
  if (ride_buf >= 0) {
    // the NEXT sweep's bulk pass behind the scan workgroups, under the label sort this scan starts from (1 - pw)
    a.ride = make_ride(m, d, ride_buf, 1 - pw, 2, a.G, &nr, &ntl, &ride_bytes);
    a.Gn0 = gcnt(m, ride_buf, 0); a.Gsn0 = gs(m, ride_buf, 0);
    if (!riders_here) { nr = 0; ntl = 0; a.ride.nriders = 0; }      // (the pass rode with the parameter step)
    timed = riders_here && m->timing > 0 && (sweep % (uint32_t)m->timing) == 0 && m->ev_used < 4096;
    if (timed) { e0 = next_event(m); e1 = next_event(m); GG_HIP(hipEventRecord(e0, m->stream)); }
  }
  // Pair scheduling at sizes where the weight model's pass fills the chip for milliseconds (N >= GG_PAIR_MIN_N, default 16384;
  // GG_PAIR_SCHED=0 turns it off).  A resident grid cannot assemble while a pass keeps backfilling the compute units, so the
  // adjacency model's scan used to wait out the weight model's pass and then share the chip with the weight model's scan -
  // each grid's rounds then wait for workgroups that share a CU with the other's (N=65536, K=64: the weight model's scan 2.6 ms
  // instead of 0.8).  Now the adjacency model's scan waits for the weight model's scan of the same sweep instead, and the weight model's
  // stream idles for a few microseconds behind its scan (k_delay): the adjacency model's grid assembles on the empty chip, the
  // pass then fills the room around it, and the weight model's next scan has the chip to itself.
  if (pair && m->is_adj) { hipEvent_t ev = gg_pair_lookup(d->wscan, sweep); if (ev) GG_HIP(hipStreamWaitEvent(m->stream, ev, 0)); }
#define SCN1(NTV, ADJV, SPV)                                                                                       \
  do {                                                                                                            \
    static bool attr_set = false;                                                                                 \
    if (!attr_set && lds > 48 * 1024) {                                                                           \
      GG_HIP(hipFuncSetAttribute((const void*)k_scan<NTV, ADJV, SPV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    k_scan<NTV, ADJV, SPV><<<a.G, 256, lds, m->stream>>>(a);                                                       \
  } while (0)
#define SCNR(NTV, ADJV, SPV, RV)                                                                                   \
  do {                                                                                                            \
    static bool attr_set = false;                                                                                 \
    if (!attr_set && lds > 48 * 1024) {                                                                           \
      GG_HIP(hipFuncSetAttribute((const void*)k_scan<NTV, ADJV, SPV, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    k_scan<NTV, ADJV, SPV, RV><<<a.G + nr, 256, lds, m->stream>>>(a);                                              \
  } while (0)
#define SCN(NTV, ADJV)                                                                                             \
  do {                                                                                                            \
    if (ride_buf >= 0) {                                                                                          \
      if (stage != 1 && m->Kp == 64) { if (ntl) SCNR(NTV, ADJV, 1, 2); else SCNR(NTV, ADJV, 1, 1); }              \
      else { if (ntl) SCNR(NTV, ADJV, 0, 2); else SCNR(NTV, ADJV, 0, 1); }                                        \
    } else if (m->Kp > 64) SCN1(NTV, ADJV, 2);                                                                    \
    else if (stage != 1 && m->Kp == 64) SCN1(NTV, ADJV, 1);                                                       \
    else SCN1(NTV, ADJV, 0);                                                                                      \
  } while (0)
  if (adj16) SCN1(2, true, 3);
  else if (m->is_adj) SCN(2, true);
  else if (m->joint && m->B == 1) SCN(4, false);
  else if (m->joint) SCN(7, false);
  else if (m->B == 1) SCN(3, false);
  else if (m->B == 2) SCN(6, false);
  else if (m->B == 3) SCN(10, false);
  else SCN(15, false);
#undef SCN1
#undef SCNR
#undef SCN
  GG_HIP(hipGetLastError());
  if (ride_buf >= 0) {
    if (timed) { m->ev_bytes.push_back(ride_bytes); m->ev_sweep.push_back(sweep); GG_HIP(hipEventRecord(e1, m->stream)); }
    m->zeroed[ride_buf] = false;
  }
  if (pair && !m->is_adj) {
    { const int rcp = gg_pair_record(d->wscan, sweep, m->stream); if (rcp) return rcp; }
    k_delay<<<1, 64, 0, m->stream>>>(pair_delay_ticks());
    GG_HIP(hipGetLastError());
  }
  return GG_OK;
}

// GG_PRIDE (default 1): while a chain keeps changing labels, the next sweep's pass rides with THIS sweep's parameter step
// (VERDICT r3 #3), the resident scan runs alone behind it, and the scan's change list then moves the statistics counted ahead.
static int pride_on() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("GG_PRIDE"); on = e ? atoi(e) : 0; }
  return on;
}

// A k_scan exchange timed out: the model continues with restart rounds, and tries the resident grid again after
// scan_backoff further sweeps (1024, doubling with every timeout) - one transient (a device shared with another
// process for a moment) must not cost the 2.6x per-change penalty for the rest of the model's life.
static void scan_suspend(gg_model* m) {
  if (m->scan_pref == 0) return;
  m->scan_pref_saved = m->scan_pref;
  m->scan_pref = 0; m->use_scan = false;
  m->scan_retry_at = m->st.sweeps + m->scan_backoff;
  if (m->scan_backoff < (1ull << 20)) m->scan_backoff *= 2;
}
static void scan_maybe_resume(gg_model* m) {
  if (m->scan_pref_saved == 0 || m->scan_pref != 0 || m->st.sweeps < m->scan_retry_at) return;
  m->scan_pref = m->scan_pref_saved;
  m->scan_pref_saved = 0;
  m->use_scan = m->scan_pref == 1;
  m->clean_run = 0; m->since_revert = 1 << 30;
}

// Finish a scan the pipeline left unfinished.  Restart rounds are device-driven: each round is ONE launch
// (k_round: bring the next window's statistics up to date with the change list, re-speculate it, resolve it)
// that reads the scan position from the control block and turns into a no-op once the scan is done, so a
// batch of rounds is enqueued without host synchronisation; the host only checks between batches, and
// flushes the change list into G chip-wide (k_apply_L) before a batch that could overflow it.
// The round that completes the scan sorts the labels into perm buffer `pw`.
// The statistics of a stalled sweep counted again under the labels as they stand (a resident scan gave up behind a flush of
// its change list: k_scan's give-up path): the sort of the current labels into slot P_s - the one the stalled sweep's pass used,
// not needed any more: the scan's tail writes the final sort into the other one -, the buffer zeroed, one pass without the
// pipeline's gate; nothing of the (empty) list counts as applied.
static int recount_stats(gg_model* m, const gg_data* d, int buf, int P_s) {
  const int pcur_keep = m->pcur;
  m->pcur = P_s;
  m->perm_valid = false;
  m->zeroed[buf] = false;
  GG_HIP(hipMemsetAsync(m->ap, 0, (size_t)m->N * sizeof(int32_t), m->stream));
  const int rc = enqueue_bulk(m, d, buf, 3, false);
  m->pcur = pcur_keep;
  m->perm_valid = false;
  return rc;
}
static bool gave_up_behind_flush(const gg_model* m) {
  return (m->h_rs[3] & 4) && !(m->h_rs[3] & ~6) && m->h_rs[4] && m->h_rs[0] != 0;
}

static int finish_scan(gg_model* m, const gg_data* d, int buf, uint32_t sweep, bool tail_m, bool tail_sort, int pw) {
  int rc;
  if (m->scan_pref != 0 && !tail_m) {
    // The resident-grid scan finishes it (6.5 us per further change instead of a launch per change): flush the changes
    // accepted so far into G (exact statistics under the current labels, empty list), then k_scan from the position
    // the scan stalled at.
    if ((rc = read_ctl(m))) return rc;
    if (m->h_rs[4] && m->h_rs[0] != 0) {
      if ((rc = launch_apply_L(m, d, buf))) return rc;
      if ((rc = launch_scan(m, d, buf, sweep, tail_sort, pw, true))) return rc;   // (its rounds are counted on the device)
      if ((rc = read_ctl(m))) return rc;
    }
    if (gave_up_behind_flush(m)) {
      // (the resumed resident scan gave up behind a flush: as in model_sync - count again, restart rounds from the position)
      if ((rc = write_ctl_word(m, 3, 0))) return rc;
      fprintf(stderr, "graphgibbs: the resident-grid scan timed out at node %d after flushing its change list; statistics are counted again, this model continues with restart rounds for %llu sweeps\n", m->h_rs[0], (unsigned long long)m->scan_backoff);
      scan_suspend(m);
      m->h_rs[3] = 0;
      if ((rc = recount_stats(m, d, buf, 1 - pw))) return rc;
    } else if (m->h_rs[3] == 2) {
      // the resident grid timed out and left a consistent stalled scan: restart rounds from here on (see model_sync)
      if ((rc = write_ctl_word(m, 3, 0))) return rc;
      fprintf(stderr, "graphgibbs: the resident-grid scan timed out; this model continues with restart rounds for %llu sweeps\n", (unsigned long long)m->scan_backoff);
      scan_suspend(m);
    } else {
      if (m->h_rs[3]) return GG_OK;          // (status: reported by the caller's next control-block check)
      if (m->h_rs[4]) { if ((rc = write_ctl_word(m, 4, 0))) return rc; }
      return GG_OK;
    }
  }
  // at most 512 node groups per round (two workgroups per CU): beyond that a round's latency grows with the window
  int window = m->N < 4096 ? m->N : 4096;
  if (window > 512 * (256 / m->Kp)) window = 512 * (256 / m->Kp);
  // first batch: what the previous unfinished scan needed plus a margin (rounds behind the end of the scan are
  // cheap no-ops, a second batch costs a host round trip)
  int batch = m->round_hint > 4 ? m->round_hint : 4;
  long long rounds = 0;
  int seq_first = -1;
  for (;;) {
    if ((rc = read_ctl(m))) return rc;
    if (m->dbg && rounds > 0) {
      // phase times of the last round that ran: resolver workgroup and the mean over the active ones (us)
      std::vector<long long> h(512 * 8 + 8);
      GG_HIP(hipMemcpy(h.data(), m->dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
      const int last = (int)h[512 * 8], nwg = (int)h[512 * 8 + 1];
      const long long* t = &h[(size_t)last * 8];
      fprintf(stderr, "[round dbg] %s nwg=%d resolver wg %d: ctl %.2f load %.2f apply %.2f conv %.2f lp %.2f ticket %.2f resolve %.2f | total %.2f us\n",
              m->is_adj ? "adj" : "w", nwg, last, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
              (t[5] - t[4]) * 0.01, (t[6] - t[5]) * 0.01, (t[7] - t[6]) * 0.01, (t[7] - t[0]) * 0.01);
      long long first_start = t[0], mx[8] = {0};
      for (int w = 0; w < nwg; ++w) {
        if (h[(size_t)w * 8] < first_start) first_start = h[(size_t)w * 8];
        for (int q = 1; q < 7; ++q) { long long dlt = h[(size_t)w * 8 + q] - h[(size_t)w * 8 + q - 1]; if (dlt > mx[q]) mx[q] = dlt; }
      }
      fprintf(stderr, "[round dbg]   max over wgs: ctl %.2f load %.2f apply %.2f conv %.2f lp %.2f ticket %.2f ; first wg start -> resolver end %.2f us\n",
              mx[1] * 0.01, mx[2] * 0.01, mx[3] * 0.01, mx[4] * 0.01, mx[5] * 0.01, mx[6] * 0.01, (t[7] - first_start) * 0.01);
    }
    if (!m->h_rs[4] || m->h_rs[0] == 0) {   // scan complete (a set flag with pos == 0 is a pipeline flush, not a stall)
      if (m->h_rs[4]) { if ((rc = write_ctl_word(m, 4, 0))) return rc; }
      if (seq_first >= 0) {
        unsigned long long st;
        memcpy(&st, &m->h_rs[GG_RS_STATE], sizeof(st));
        int q0, q1, q2, q3, seq_now;
        gg_unpack_state(st, &q0, &q1, &q2, &q3, &seq_now);
        const int used = (seq_now - seq_first) & 0xfff;           // rounds that actually ran
        int hint = used + used / 4 + 2;
        m->round_hint = hint > 256 ? 256 : hint;
      }
      return GG_OK;
    }
    if (m->h_rs[1] + batch > m->Lcap) {
      if ((rc = launch_apply_L(m, d, buf))) return rc;
    }
    int base;
    {
      unsigned long long st;
      memcpy(&st, &m->h_rs[GG_RS_STATE], sizeof(st));
      int q0, q1, q2, q3;
      gg_unpack_state(st, &q0, &q1, &q2, &q3, &base);
    }
    if (seq_first < 0) seq_first = base;
    for (int r = 0; r < batch; ++r)
      if ((rc = launch_round(m, d, buf, sweep, tail_m, tail_sort, pw, window, base + r))) return rc;
    rounds += batch;
    m->st.resolve_rounds += batch;
    if (batch < 64) batch *= 2;
    if (rounds > 4LL * m->N + 64) return gg_fail(GG_ERR_HIP, "resolve did not make progress (pos=%d)", m->h_rs[0]);
  }
}

// ---- one full sweep, enqueued optimistically (no host sync) ---------------------------------
// Serial form, everything on the model's stream:
//   bulk -> params (+ zero the next buffer) -> scan (k_scan, or k_spec_resolve while the chain is quiet and the ahead
//   form is off) [K = 1 kinds: bulk -> params]
static int enqueue_sweep_serial(gg_model* m, const gg_data* d, uint32_t sweep, bool last = false) {
  int rc;
  const bool scan = m->sbm && m->K > 1;
  const int buf = m->cur, nxt = (buf + 1) % 3, P = m->pcur;
  bool rode = false;
  if (m->ahead && m->ahead_uid == d->uid) {
    // the statistics of this sweep were counted ahead by the riders of the previous one (which changed no label, or whose
    // resident scan's change list was applied to them)
    m->ahead = false;
  } else {
    drop_ahead(m);
    if ((rc = enqueue_bulk(m, d, buf, scan ? 3 : 1, true, sweep))) return rc;
  }
  // A chain that keeps changing labels (resident-scan form): the next sweep's pass cannot ride in a quiet chain's fused launch -
  // its labels change - and riding in the scan's own launch loses (below).  It rides with the PARAMETER STEP instead, as the
  // quiet call's first launch already does: [k_params of sweep s | riders = the pass of sweep s + 1 under the labels scan s
  // starts with], then k_scan ALONE - nothing saturates the memory system under its exchanges -, then k_apply_L (mode 3) moves
  // the statistics counted ahead by the scan's change list (integer moves commute with the riders' adds; a list that overflows
  // inside the scan is flushed into that buffer by the scan itself).  31 + 11.6 + 50 us of a mixing sweep become 32 + 50.
  const int c3 = (buf + 2) % 3;
  const bool pride = scan && m->use_scan && pride_on() && m->can_ahead && m->Kp <= 64 && !m->joint && m->scan_ride == 0 &&
                     m->chg_per_sweep >= 0.0 && m->chg_per_sweep <= 256.0 && m->ngraphs == 1;
  if (pride) {
    if ((rc = ensure_perm(m)) || (rc = ensure_zero(m, nxt))) return rc;
    if ((rc = enqueue_params(m, d, buf, true, true, true, c3, sweep, P, nxt, 2))) return rc;
  } else if ((rc = enqueue_params(m, d, buf, true, true, scan, nxt, sweep, P))) return rc;
  if (scan && m->use_scan && !m->is_adj && !m->sharded && pair_sched_on() && pair_pass_on() && m->N >= pair_min_n() && m->Kp == 64) {
    // (pair scheduling: the adjacency model's next pass waits for this point - it streams under the weight model's scan,
    // which leaves the memory system idle, not under its pass or its parameter step, which read at the memory system's rate)
    { const int rcp = gg_pair_record(d->wpass, sweep, m->stream); if (rcp) return rcp; }
  }
  if (scan) {
    if (pride) {
      static int rec_on = -1;
      if (rec_on < 0) { const char* e = getenv("GG_EPILOGUE"); rec_on = e ? atoi(e) : 1; }
      if ((rc = launch_scan(m, d, buf, sweep, true, 1 - P, false, nxt, last && rec_on, false))) return rc;
      if ((rc = launch_apply_L(m, d, nxt, 3, false))) return rc;
      rode = true;
    } else if (m->use_scan) {
      // GG_SCAN_RIDE (off by default): the next sweep's bulk pass rides in the scan's launch, counting under the labels the
      // scan starts with, and the scan's change list then moves those statistics to the labels it ends with - exact
      // (tests/test_gpu_edge_cases.py::test_resident_grid_scan_is_the_same_chain, ride = 2), and measured SLOWER on MI355X
      // (round 3): the scan's exchange rounds are memory-side polls, and under a pass that saturates the memory system each
      // of them takes several times as long - N=4096, K=16, 60 changes per sweep: 192 us per sweep against 95 serial;
      // N=65536, K=64: 11.6 ms against 8.6.  The quiet chain's fused launch overlaps the same pass with ONE verdict word
      // polled behind a sleep; a scan that needs 5-7 grid-wide exchanges does not survive the loaded latency.
      // (not while the chain relabels hundreds of nodes per sweep - burn-in: the riding launch keeps a short change list,
      // and a list that overflows is flushed chip-wide)
      const bool ride = m->Kp <= 64 && m->zeroed[nxt] &&
                        (m->scan_ride == 2 || (m->scan_ride == 1 && m->chg_per_sweep >= 0.0 && m->chg_per_sweep <= 128.0));
      static int rec_on = -1;
      if (rec_on < 0) { const char* e = getenv("GG_EPILOGUE"); rec_on = e ? atoi(e) : 1; }
      if ((rc = launch_scan(m, d, buf, sweep, true, 1 - P, false, ride ? nxt : -1, last && !ride && rec_on))) return rc;
      if (ride) {
        if ((rc = launch_apply_L(m, d, nxt, 3, false))) return rc;
        rode = true;
      }
    } else if (m->scan_pref < 0 || m->wide) {
      // adaptive form, quiet phase: speculation and the resolution of the common case in one launch; a scan that does
      // change a label is left stalled behind its first change for the resident grid (finish_scan) - or, for more than 64
      // blocks without the resident grid (GG_SCAN=0, or suspended after a timeout), for the restart rounds
      if ((rc = launch_spec_resolve(m, d, buf, sweep, 1 - P))) return rc;
    } else {
      if ((rc = launch_speculate(m, d, buf, 0, -1, sweep, true))) return rc;
      if ((rc = launch_resolve(m, d, buf, sweep, false, true, 1 - P))) return rc;
    }
    if (!m->use_scan) m->st.resolve_rounds += 1;   // (k_scan counts its exchange rounds on the device: gg_last_stats)
    m->m_pending = true;   // drawn by the next sweep's k_params (or at the next sync)
    m->m_sweep = sweep;
    m->pcur = 1 - P;
  } else if (m->sbm) {
    // K == 1 SBM: resample_c returns early (adjacency.py:657-658) but m is still drawn
    k_draw_m<<<1, m->K > 64 ? 256 : 64, (m->K > 64 ? 256 : 64) * sizeof(double), m->stream>>>(m->cnt, m->pi, m->K, m->cfg.seed,
                                                        m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M, sweep, m->m);
    GG_HIP(hipGetLastError());
  }
  m->cur = nxt;
  if (rode) { m->ahead = true; m->ahead_uid = d->uid; }   // (a scan that times out stalls the pipeline: model_sync drops it)
  m->st.sweeps_serial += 1;
  return GG_OK;
}

// Ahead form (scan models while the chain is quiet).  A sweep that is followed by another one in the
// same call runs as ONE launch:
//   k_sweep_q(s):  scan of sweep s  |  parameter step of sweep s+1 (gated on the scan's verdict)  |  riders: bulk pass of s+2
// preceded, for the first sweep of a run, by
//   k_params(s) + riders: the whole bulk pass of sweep s+1
// and the last sweep of a call by a plain k_spec_resolve(s): what a caller can observe after n sweeps is the state after
// exactly n sweeps.  The riders count under the labels the sweep STARTS with: exactly the statistics a later sweep needs
// iff no scan in between changes a label.  A scan that does change one stalls the pipeline (every later launch is a no-op,
// the gated parameter step of the next sweep writes nothing), the host finishes it with the resident grid and continues in
// the serial form; what the riders counted is dropped.  Every sweep still streams all of X and X^T once.
static int enqueue_sweep_ahead(gg_model* m, const gg_data* d, uint32_t sweep, bool more) {
  int rc;
  bool whole = true;      // the next sweep's whole pass rode with this sweep's parameter step
  const int a = m->cur, b = (a + 1) % 3, c = (a + 2) % 3, P = m->pcur;
  {
    // K > 32 without the tables in LDS: the scan's conditional keeps its table entries in registers (178 of them), the
    // fused launch is compiled for two waves per SIMD, and riders that share it stream at half their speed.  Every sweep
    // then runs as the two-launch form: k_params with the whole next pass as riders (full occupancy), k_spec_resolve alone.
    static int fuse64 = -1;   // GG_FUSE64=1: the fused launch for these shapes as well
    if (fuse64 < 0) { const char* e = getenv("GG_FUSE64"); fuse64 = e ? atoi(e) : 0; }
    int stage = 0;
    lds_spec(m, &stage);
    if ((!stage && m->Kp == 64 && !fuse64) || m->wide) more = false;   // (more than 64 blocks: the two-launch form as well)
  }
  if (!m->pre) {
    if (!(m->ahead && m->ahead_uid == d->uid)) {
      // prologue: nothing was counted ahead for this sweep (first sweep, new data or labels, after a stall)
      drop_ahead(m);
      if ((rc = enqueue_bulk(m, d, a, 3, true, sweep))) return rc;
    }
    if ((rc = ensure_perm(m)) || (rc = ensure_zero(m, b))) return rc;
    // parameter step of this sweep; the whole bulk pass of the next one rides along (also when the scan follows in a launch
    // of its own: 98 against 104 us per one-sweep call with half of the pass in each of the two launches; GG_ONE_RIDE=0)
    static int one_ride = -1;
    if (one_ride < 0) { const char* e = getenv("GG_ONE_RIDE"); one_ride = e ? atoi(e) : 1; }
    whole = more || one_ride;
    if ((rc = enqueue_params(m, d, a, true, true, true, c, sweep, P, b, whole ? 2 : 0))) return rc;
  }
  m->m_pending = true;     // resample_m of this sweep: drawn by the next parameter step (or at the next sync)
  m->m_sweep = sweep;
  if (more) {
    if ((rc = launch_sweep_q(m, d, a, b, c, sweep, P))) return rc;
    if (m->pre) m->st.sweeps_fused += 1; else m->st.sweeps_two_launch += 1;   // (the first sweep of a run: k_params + riders in front)
    m->pre = true;
  } else {
    // last sweep of the call: the scan alone (+ the rest of the next sweep's pass when it did not ride above) - with the
    // call's epilogue in the same launch where that form fits (resample_m, the control block for the host, and the
    // log-likelihood of the state the call leaves when the caller keeps asking for it)
    if (epi_ok(m) && (m->pre || whole)) {
      if ((rc = launch_spec_resolve_epi(m, d, a, b, sweep, P))) return rc;
      m->m_pending = false;     // (drawn by the epilogue; a scan that stalls: model_sync's slow path draws it)
    } else if ((rc = launch_spec_resolve(m, d, a, sweep, 1 - P, (m->pre || whole) ? -1 : b))) return rc;
    m->st.sweeps_two_launch += 1;
    m->pre = false;
  }
  m->st.resolve_rounds += 1;
  m->pcur = 1 - P;
  m->cur = b;
  m->ahead = true;          // valid once the batch is confirmed clean (model_sync drops it after a stall)
  m->ahead_uid = d->uid;
  return GG_OK;
}

static int enqueue_sweep(gg_model* m, const gg_data* d, uint32_t sweep, bool more) {
  const bool learn_params = (m->kind != GG_BERN_ADJ && m->kind != GG_FIXED_W);
  if (!learn_params) return GG_OK;
  int rc;
  m->epi_tag = 0;     // (set by the launch that ends a batch with the epilogue record)
  if (m->can_ahead && !m->use_scan && m->scan_pref < 0) rc = enqueue_sweep_ahead(m, d, sweep, more);
  else rc = enqueue_sweep_serial(m, d, sweep, !more);
  if (rc) return rc;
  if (!more && m->m_pending) {
    // The call's last sweep: resample_m (adjacency.py:697-702) is enqueued right behind its scan - a no-op if that scan
    // stalls (model_sync then draws m once the scan is finished) - instead of being launched after the control block has
    // been read: a one-sweep call ends with that read, and the next call's first kernel no longer queues behind this one.
    k_draw_m<<<1, m->K > 64 ? 256 : 64, (m->K > 64 ? 256 : 64) * sizeof(double), m->stream>>>(m->cnt, m->pi, m->K, m->cfg.seed,
                                                        m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M, m->m_sweep, m->m, m->rs);
    GG_HIP(hipGetLastError());
    m->m_pending = false;
  }
  return GG_OK;
}

static void begin_call(gg_model* m) {
  m->timing_dirty = false;
  m->ev_used = 0;
  m->ev_sweep.clear(); m->ev_bytes.clear();
  m->st.bulk_bytes = 0; m->st.bulk_ms = 0; m->st.bulk_launches = 0; m->st.total_ms = 0;
  // (the call's own event bracket only when timing was asked for: two records cost what a small kernel costs)
  m->ev_call = m->timing > 0;
  if (m->ev_call) hipEventRecord(m->ev_call0, m->stream);
}

static int collect_timing(gg_model* m) {
  float ms = 0;
  if (m->ev_call) GG_HIP(hipEventElapsedTime(&ms, m->ev_call0, m->ev_call1));
  m->st.total_ms = ms;
  m->st.bulk_ms = 0; m->st.bulk_launches = 0; m->st.bulk_bytes = 0;
  for (size_t i = 0; i + 1 < m->ev_used; i += 2) {
    if (i / 2 < m->ev_sweep.size() && m->ev_sweep[i / 2] == ~0u) continue;   // the launch was a no-op behind a stalled scan
    float t = 0;
    GG_HIP(hipEventElapsedTime(&t, m->ev[i], m->ev[i + 1]));
    m->st.bulk_ms += t;
    m->st.bulk_launches += 1;
    if (i / 2 < m->ev_bytes.size()) m->st.bulk_bytes += m->ev_bytes[i / 2];
  }
  return GG_OK;
}

// Confirm the optimistic sweeps: if a scan stalled (or flushed the pipeline), finish it on the slow
// path and re-enqueue the sweeps that became no-ops behind it.
// ctl_fresh: the caller has already recorded ev_call1, copied the control block to h_rs and waited for it
// (gg_sync_many overlaps that round trip over all its models).
static int model_sync(gg_model* m, bool ctl_fresh) {
  if (m->eig || m->ld || (m->pending == 0 && m->todo == 0)) return GG_OK;
  int rc;
  const gg_data* d = m->pend_data;
  const bool scan = m->sbm && m->K > 1;
  if (!ctl_fresh && m->ev_call) GG_HIP(hipEventRecord(m->ev_call1, m->stream));
  bool enq_after_read = false;     // something was enqueued behind the last control-block read
  for (;;) {
    bool abort0 = false, recount = false;
    if (ctl_fresh) ctl_fresh = false;
    else if ((rc = fetch_ctl(m))) return rc;
    if (gave_up_behind_flush(m)) {
      // A k_scan exchange timed out AFTER the scan had flushed its change list (or inside the flush): labels, block sizes and
      // position are a valid prefix of the scan, the statistics are not (k_scan's give-up path).  Count them again under the
      // labels as they stand - one bulk pass, below - and let the restart rounds go on from the position with an empty list.
      if ((rc = write_ctl_word(m, 3, 0))) return rc;
      if (m->scan_pref != 0) fprintf(stderr, "graphgibbs: the resident-grid scan timed out at node %d after flushing its change list; statistics are counted again, this model continues with restart rounds for %llu sweeps\n", m->h_rs[0], (unsigned long long)m->scan_backoff);
      scan_suspend(m);
      m->h_rs[3] = 0;
      recount = true;
    }
    if (m->h_rs[3] == 2) {
      // A k_scan exchange timed out (the resident grid could not assemble - a device shared with other processes, say)
      // and left the scan as a stalled one at a consistent position: carry on with the launch-per-change form, for good.
      if ((rc = write_ctl_word(m, 3, 0))) return rc;
      if (m->scan_pref != 0) fprintf(stderr, "graphgibbs: the resident-grid scan timed out at node %d (%d changes in its list); this model continues with restart rounds for %llu sweeps\n", m->h_rs[0], m->h_rs[1], (unsigned long long)m->scan_backoff);
      scan_suspend(m);
      m->h_rs[3] = 0;
      abort0 = m->h_rs[4] && m->h_rs[0] == 0;   // it gave up before its first decision: still a stalled scan, at node 0
    }
    if (m->h_rs[3]) {
      const int st3 = m->h_rs[3];
      write_ctl_word(m, 3, 0);
      m->pending = 0;
      m->todo = 0;
      m->ahead = false;
      m->pre = false;
      m->zeroed[0] = m->zeroed[1] = m->zeroed[2] = false;
      (void)hipStreamSynchronize(m->stream);   // (what was enqueued behind the failed launch runs as no-ops: let it, before the caller tears anything down)
      set_pend_data(m, nullptr);
      if (st3 & 8) return gg_fail(GG_ERR_HIP, "k_sweep_q: the scan's verdict did not arrive (device hung?); set GG_AHEAD=0");
      if (st3 & 2) return gg_fail(GG_ERR_HIP, "k_scan: grid barrier timed out (the resident grid did not assemble); set GG_SCAN=0");
      if (st3 & 4) return gg_fail(GG_ERR_HIP, "k_scan: grid barrier timed out inside the statistics flush; set GG_SCAN=0");
      return gg_fail(GG_ERR_NUMERIC, "a block covariance is not positive definite");
    }
    if (!m->h_rs[4]) {  // nothing stalled or flushed: every enqueued sweep completed
      if (scan && m->pending > 0) {
        m->chg_per_sweep = (double)((long long)m->h_rs[2] - m->ride_chg_seen) / (double)m->pending;
        m->ride_chg_seen = (long long)m->h_rs[2];
      }
      if (m->scan_pref < 0) {
        // adaptive scan form: stay with the resident-grid scan while labels keep changing, go back to speculate +
        // resolve after scan_hold sweeps in a row without a change
        const long long chg = (long long)m->h_rs[2] - m->chg_seen;
        m->chg_seen = (long long)m->h_rs[2];
        if (chg != 0) { m->use_scan = true; m->clean_run = 0; }
        else if (m->use_scan && (m->clean_run += m->pending) >= m->scan_hold) { m->use_scan = false; m->clean_run = 0; m->since_revert = 0; }
        else if (!m->use_scan) m->since_revert += m->pending;
      }
      m->sweep += (uint32_t)m->pending;
      m->st.sweeps += (uint64_t)m->pending;
      m->pending = 0;
      if (m->epi_hit && m->epi_ll && m->todo == 0) {
        // the epilogue's log-likelihood belongs to the state this call leaves
        m->ll_cached = true; m->ll_val = m->h_epi[0]; m->ll_bad = m->h_epi[1] != 0.0; m->ll_uid = d->uid;
      }
      m->epi_tag = 0; m->epi_hit = false;
      scan_maybe_resume(m);
      if (m->todo == 0) break;
      // a clean batch: look further ahead again
      if (m->depth < (1 << 20)) m->depth *= 2;
      // (adaptive form, speculate + resolve: a stall turns everything enqueued behind it into no-op launches, ~15 us per
      // sweep - do not run much further ahead than a stall costs anyway: 64 sweeps, doubling with every clean batch up to
      // 512 (a chain that has been quiet that long rarely stalls); the resident scan cannot stall)
      int take = m->todo < m->depth ? m->todo : m->depth;
      if (m->scan_pref < 0 && !m->use_scan) {
        if (m->ahead_cap < 512) m->ahead_cap *= 2;
        if (take > m->ahead_cap) take = m->ahead_cap;
      }
      m->cur0 = m->cur; m->pcur0 = m->pcur;
      m->done_base = m->h_rs[5];   // (every scan of the batch completed: the count just read is the next batch's base)
      // (`more` also for the last sweep of the batch when the call has further sweeps to enqueue: the quiet form's pipeline
      // - parameters of the next sweep, statistics of the one after - is not drained at a batch boundary)
      for (int i = 0; i < take; ++i)
        if ((rc = enqueue_sweep(m, d, m->sweep + (uint32_t)i, i + 1 < take || m->todo > take))) return rc;
      m->pending = take;
      m->todo -= take;
      if (m->ev_call) GG_HIP(hipEventRecord(m->ev_call1, m->stream));
      continue;
    }
    // A scan stalled: the sweeps enqueued behind it ran as no-ops and must be enqueued again.  A chain that keeps
    // changing labels stalls every sweep, so stop running ahead (depth 1) until batches come back clean.
    if (m->scan_pref >= 0) m->depth = 1;   // (the adaptive form switches to the resident scan below, which cannot stall)
    m->ahead_cap = 64;
    if (m->scan_pref < 0) {   // the chain is changing labels: k_scan from the next sweep on
      // (a stall soon after the last switch back: the chain changes labels now and then - hold the resident form longer)
      // Break-even: a quiet sweep costs ~4 us more in the resident form, a stall ~400 us in the other one (slow path,
      // host round trips, the no-op launches behind it) = ~100 quiet sweeps.  So once a chain has surprised the
      // speculative form after a switch back, the resident form is held for 4 x the quiet run that preceded the surprise
      // (at least 64, at most 4096 sweeps); the first stall of all (burn-in) leaves the short initial hold alone.
      // Round 4, re-measured: with the quiet chain's sweep fused into one launch (36-38 us) a quiet sweep costs ~18 us more in
      // the resident form (the adjacency model beside a fused weight model: 56 us), a stall 300-600 us (bench blocks with one):
      // break-even ~25 quiet sweeps, not 100 - and the old rule held a chain that had been quiet for q sweeps in the
      // resident form for 4 q more (chains with one change in warm-up spent 80 of 140 timed sweeps there).  Now: a surprise
      // after a SHORT quiet run (q < hold_q = 32: the chain changes labels every few sweeps) holds the resident form for
      // hold_long = 32 clean sweeps; after a long one only hold_min = 16 (past the burst a change tends to start).
      // Driver's seed mix (tools/scripts/hold_ab.sh): 16.7 k -> 18.6 k sweeps/s at 20-sweep calls, 17.6 -> 18.6 k at 200.
      if (!m->use_scan && m->since_revert < (1 << 29)) {
        static int hold_q = -1, hold_long = -1, hold_min = -1, hold_old = -1;
        if (hold_q < 0) {
          const char* e;
          hold_q = (e = getenv("GG_SCAN_HOLD_Q")) ? atoi(e) : 32;
          hold_long = (e = getenv("GG_SCAN_HOLD_LONG")) ? atoi(e) : 32;
          hold_min = (e = getenv("GG_SCAN_HOLD_MIN")) ? atoi(e) : 16;
          hold_old = (e = getenv("GG_SCAN_HOLD_OLD")) ? atoi(e) : 0;
        }
        if (hold_old) {
          long long h = 4 * m->since_revert;
          m->scan_hold = h < 64 ? 64 : (h > 4096 ? 4096 : h);
        } else {
          m->scan_hold = m->since_revert < hold_q ? hold_long : hold_min;
        }
      }
      m->use_scan = true; m->clean_run = 0; m->chg_seen = (long long)m->h_rs[2];
    }
    // ctl[5] scans completed before the flag went up.  pos != 0: the next one is stuck mid-scan.
    const int done = scan ? (int)((uint32_t)m->h_rs[5] - (uint32_t)m->done_base) : m->pending;
    const bool mid = m->h_rs[0] != 0 || abort0;
    const int completed = mid ? done + 1 : done;       // after the slow path below
    if (mid) {
      const int s_idx = done;                          // index of the stalled sweep within the batch
      const int buf_s = (m->cur0 + s_idx) % 3, P_s = (m->pcur0 + s_idx) & 1;
      if (abort0) {
        // nothing of this scan is decided yet (statistics, parameters and tables of the sweep are in place): run it in
        // the speculative form; what that leaves unfinished is finished below like any other stalled scan
        if ((rc = write_ctl_word(m, 4, 0))) return rc;
        if (m->wide) {
          if ((rc = launch_spec_resolve(m, d, buf_s, m->sweep + (uint32_t)s_idx, 1 - P_s))) return rc;
        } else {
          if ((rc = launch_speculate(m, d, buf_s, 0, -1, m->sweep + (uint32_t)s_idx, true))) return rc;
          if ((rc = launch_resolve(m, d, buf_s, m->sweep + (uint32_t)s_idx, false, true, 1 - P_s))) return rc;
        }
      }
      if (recount && (rc = recount_stats(m, d, buf_s, P_s))) return rc;
      if ((rc = finish_scan(m, d, buf_s, m->sweep + (uint32_t)s_idx, false, true, 1 - P_s))) return rc;
    } else {
      if ((rc = write_ctl_word(m, 4, 0))) return rc;
    }
    m->m_pending = true;
    m->m_sweep = m->sweep + (uint32_t)(completed - 1);
    const int remaining = m->pending - completed;
    for (uint32_t& sw : m->ev_sweep)    // timed launches of the sweeps behind the stall did nothing
      if (sw != ~0u && sw >= m->sweep + (uint32_t)completed) sw = ~0u;
    m->sweep += (uint32_t)completed;
    m->st.sweeps += (uint64_t)completed;
    m->cur = (m->cur0 + completed) % 3;
    m->pcur = (m->pcur0 + completed) & 1;
    // Whatever the sweeps behind the stall were to count or zero did not happen, and what the riders of the stalled sweep
    // counted ahead belongs to labels that have changed since: every buffer is zeroed again before its next use.
    m->ahead = false;
    m->pre = false;
    m->zeroed[0] = m->zeroed[1] = m->zeroed[2] = false;
    m->pending = 0;
    // word 5 (completed scans, a running count): what the slow path left is the next batch's base (finish_scan returns
    // with a fresh copy of the control block)
    m->done_base = m->h_rs[5];
    m->cur0 = m->cur; m->pcur0 = m->pcur;
    m->todo += remaining;
    const int take = m->todo < m->depth ? m->todo : m->depth;
    for (int i = 0; i < take; ++i)
      if ((rc = enqueue_sweep(m, d, m->sweep + (uint32_t)i, i + 1 < take))) return rc;
    m->pending = take;
    m->todo -= take;
    if (take == 0) break;
    if (m->ev_call) GG_HIP(hipEventRecord(m->ev_call1, m->stream));
  }
  m->pending = 0;
  if (m->m_pending) {
    k_draw_m<<<1, m->K > 64 ? 256 : 64, (m->K > 64 ? 256 : 64) * sizeof(double), m->stream>>>(m->cnt, m->pi, m->K, m->cfg.seed,
                                                        m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M, m->m_sweep, m->m);
    GG_HIP(hipGetLastError());
    m->m_pending = false;
    enq_after_read = true;
  }
  if (scan) m->done_base = m->h_rs[5];   // (the running count of completed scans: base of the next batch - no copy, no round trip)
  m->st.label_changes = (uint64_t)m->h_rs[2];
  if (m->dbg && m->use_scan) {
    long long h[13];
    GG_HIP(hipMemcpy(h, m->dbg, sizeof(h), hipMemcpyDeviceToHost));
    if (h[3]) fprintf(stderr, "[scan dbg] %s last scan: %lld rounds; per round eval %.2f (convert %.2f, conditional %.2f, draw + report %.2f) exchange %.2f apply %.2f (fold %.2f, accept %.2f, moves %.2f) us; start-up %.2f group loads %.2f tail %.2f us; kernel %.2f us\n",
                      m->is_adj ? "adj" : "w", h[3], h[0] * 0.01 / h[3], h[8] * 0.01 / h[3], h[9] * 0.01 / h[3], h[10] * 0.01 / h[3],
                      h[1] * 0.01 / h[3], h[2] * 0.01 / h[3], h[11] * 0.01 / h[3], h[12] * 0.01 / h[3], (h[2] - h[11] - h[12]) * 0.01 / h[3],
                      h[7] * 0.01, h[4] * 0.01, h[6] * 0.01, (h[5] + h[7]) * 0.01);
  }
  // (a clean call has nothing in flight any more: its control block was read after everything it enqueued)
  if (m->ev_call) GG_HIP(hipEventSynchronize(m->ev_call1));
  (void)enq_after_read;
  if (m->swq_dbg) {
    long long h[8];
    GG_HIP(hipMemcpy(h, m->swq_dbg, sizeof(h), hipMemcpyDeviceToHost));
    if (h[1]) fprintf(stderr, "[swq dbg] %s last fused launch: scan done %.2f us, parameters done %.2f us, first rider starts %.2f us, riders done %.2f us (after the first workgroup started)\n",
                      m->is_adj ? "adj" : "w", (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, (h[4] - h[0]) * 0.01, (h[3] - h[0]) * 0.01);
  }
  set_pend_data(m, nullptr);   // nothing of this model is in flight on the network any more
  m->timing_dirty = true;
  return GG_OK;
}

extern "C" {

int gg_set_timing(gg_model* m, int32_t on) {
  if (!m) return gg_fail(GG_ERR_INVALID, "null model");
  m->timing = on < 0 ? 0 : on;
  return GG_OK;
}

// Enqueue n sweeps and return without waiting; gg_sync (or any state read) completes them.
// more_after: the caller enqueues further sweeps of this model before it synchronises (gg_resample_many's interleaving)
static int resample_async_impl(gg_model* m, const gg_data* d, int32_t n_sweeps, double beta, bool more_after) {
  if (!m) return gg_fail(GG_ERR_INVALID, "null model");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (!(beta >= 0.0 && beta <= 1.0)) return gg_fail(GG_ERR_INVALID, "beta must lie in [0, 1]");
  if (beta != 1.0 && m->eig && m->kind == GG_PROBIT_EIG)
    return gg_fail(GG_ERR_UNSUPPORTED, "tempered conditionals: not defined for the probit eigenmodel (the truncated-normal augmentation has no tempered counterpart)");
  if (m->eig) {
    gg_eigen_set_beta(m->eig, beta);   // logistic link: PG(beta, psi) augmentation (build-defined; the reference leaves it a TODO, graph_models.py:478)
    for (int s = 0; s < n_sweeps; ++s) {
      int rc = gg_eigen_resample(m->eig, d, GG_PART_ALL);
      if (rc) return rc;
    }
    return GG_OK;
  }
  if (m->ld) {
    gg_latent_set_beta(m->ld, beta);
    for (int s = 0; s < n_sweeps; ++s) {
      int rc = gg_latent_resample(m->ld, d, GG_PART_ALL, 1);
      if (rc) return rc;
    }
    return gg_latent_sync(m->ld);
  }
  if (m->sharded) return gg_fail(GG_ERR_INVALID, "this model owns a shard of the nodes: drive it with the gg_shard_* step API");
  int rc = check_data(m, d);
  if (rc) return rc;
  if (n_sweeps <= 0) return GG_OK;
  if ((m->pending || m->todo) && (m->pend_data != d || m->beta != beta) && (rc = model_sync(m))) return rc;
  m->ll_cached = false;
  if (m->pending == 0 && m->todo == 0) {
    if (m->parts_done) { m->sweep += 1; m->parts_done = 0; }   // step-wise parts were drawn under this index already
    if (m->calls_since_ll < 1000) m->calls_since_ll += 1;
    begin_call(m);
    m->cur0 = m->cur; m->pcur0 = m->pcur;
  }
  set_pend_data(m, d);
  m->beta = beta;
  for (int s = 0; s < n_sweeps; ++s) {
    const bool spec_form = m->scan_pref < 0 && !m->use_scan;
    if (m->todo > 0 || m->pending >= m->depth || (spec_form && m->pending >= m->ahead_cap)) { m->todo += 1; continue; }   // not further ahead than `depth`
    // (`more`: another sweep of this call follows - the quiet form then runs the sweep as one fused launch; also when the
    // next one is only enqueued at the sync, so that the pipeline is not drained at the batch boundary)
    const bool more = s + 1 < n_sweeps || more_after;
    if ((rc = enqueue_sweep(m, d, m->sweep + (uint32_t)m->pending, more))) return rc;
    m->pending += 1;
  }
  return GG_OK;
}

int gg_resample_async(gg_model* m, const gg_data* d, int32_t n_sweeps, double beta) {
  return resample_async_impl(m, d, n_sweeps, beta, false);
}

int gg_sync(gg_model* m) {
  if (!m) return gg_fail(GG_ERR_INVALID, "null model");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig || m->ld) return GG_OK;
  return model_sync(m);
}

// Completes the pending sweeps of several independent models.  A scan that the optimistic pipeline left
// unfinished is finished by host-enqueued restart rounds, which keep only part of the chip busy; driving
// every model from its own host thread lets the rounds (and bulk passes) of different models overlap.
// GG_CALL_DBG=1: host-side phases of the short-call path (enqueue | control-block copies issued | streams drained | bookkeeping),
// averaged and printed every 256 calls
static double g_cd[5]; static long long g_cdn; static int g_cd_on = -1;
static inline double cd_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int gg_sync_many(gg_model** ms, int32_t n) {
  if (n < 0 || (n > 0 && !ms)) return gg_fail(GG_ERR_INVALID, "gg_sync_many: bad arguments");
  if (n == 0) return GG_OK;
  if (n == 1) return gg_sync(ms[0]);
  const double cd0 = g_cd_on > 0 ? cd_now() : 0;
  // First the common case without threads: the control blocks of all models are fetched with one overlapped round trip;
  // if no scan stalled and nothing is left to enqueue, finishing a model is bookkeeping (a thread costs more than that).
  std::vector<char> fresh((size_t)n, 0);
  bool slow = false;
  for (int i = 0; i < n; ++i) {
    gg_model* m = ms[i];
    if (!m) return gg_fail(GG_ERR_INVALID, "null model");
    if (m->eig || m->ld || (m->pending == 0 && m->todo == 0)) continue;
    GG_HIP(hipSetDevice(m->cfg.device));
    if (m->ev_call) GG_HIP(hipEventRecord(m->ev_call1, m->stream));
    if (!m->epi_tag) GG_HIP(hipMemcpyAsync(m->h_rs, m->rs, 24 * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    fresh[(size_t)i] = 1;
  }
  const double cd1 = g_cd_on > 0 ? cd_now() : 0;
  for (int i = 0; i < n; ++i) {
    gg_model* m = ms[i];
    if (!fresh[(size_t)i]) continue;
    GG_HIP(hipSetDevice(m->cfg.device));
    if (m->epi_tag) { const int rc = fetch_ctl(m); if (rc) return rc; }   // (the epilogue's record, or the copy if it left none)
    else { m->epi_hit = false; GG_HIP(hipStreamSynchronize(m->stream)); }
    if (m->h_rs[4] || m->h_rs[3] || m->todo > 0) slow = true;
  }
  const double cd2 = g_cd_on > 0 ? cd_now() : 0;
  if (!slow) {
    for (int i = 0; i < n; ++i) {
      GG_HIP(hipSetDevice(ms[i]->cfg.device));
      const int rc = model_sync(ms[i], fresh[(size_t)i] != 0);
      if (rc) return rc;
    }
    if (g_cd_on > 0) { g_cd[1] += cd1 - cd0; g_cd[2] += cd2 - cd1; g_cd[3] += cd_now() - cd2; }
    return GG_OK;
  }
  std::vector<int> rcs((size_t)n, GG_OK);
  std::vector<std::string> errs((size_t)n);
  std::vector<std::thread> th;
  auto one = [&](int i) {
    int rc = GG_OK;
    if (hipSetDevice(ms[i]->cfg.device) != hipSuccess) rc = gg_fail(GG_ERR_HIP, "hipSetDevice failed");
    else if (!(ms[i]->eig || ms[i]->ld)) rc = model_sync(ms[i], fresh[(size_t)i] != 0);
    rcs[(size_t)i] = rc;
    if (rc) errs[(size_t)i] = gg_err_text;   // the error text is thread-local
  };
  for (int i = 1; i < n; ++i) th.emplace_back(one, i);
  one(0);
  for (auto& t : th) t.join();
  for (int i = 0; i < n; ++i)
    if (rcs[(size_t)i]) { gg_err_text = errs[(size_t)i]; return rcs[(size_t)i]; }
  return GG_OK;
}

// n_sweeps x resample(data) for several independent models, each enqueued and completed from its own host
// thread: the launch cost of one model's sweeps no longer delays the other's (networks.py:69-71 run n times,
// the halves being independent given the data).
int gg_resample_many(gg_model** ms, const gg_data* const* ds, int32_t n, int32_t n_sweeps, double beta) {
  if (n < 0 || (n > 0 && (!ms || !ds))) return gg_fail(GG_ERR_INVALID, "gg_resample_many: bad arguments");
  if (n == 0) return GG_OK;
  static int thr = -1;
  if (thr < 0) { const char* et = getenv("GG_MANY_INTERLEAVE"); thr = et ? atoi(et) : 32; }
  // (networks whose scans are scheduled as a pair - launch_scan - are always enqueued from this thread, sweep by sweep: the
  // adjacency model's sweep s then finds the weight model's events of sweep s recorded; a sweep takes milliseconds there, what a
  // second host thread would hide is nothing)
  bool paired = false;
  if (pair_sched_on())
    for (int i = 0; i < n; ++i) paired = paired || (ms[i] && !ms[i]->eig && !ms[i]->ld && ms[i]->N >= pair_min_n() && ms[i]->Kp == 64);
  if (n_sweeps <= thr || paired) {
    // short runs: enqueueing from this thread costs less than starting one per model - sweep by sweep, alternating
    // between the models, so that none of them waits for another's whole batch; completion is shared (gg_sync_many)
    // (the weight models first: theirs is the longer chain - every launch of it that waits behind an adjacency model's
    // enqueue is time on the critical path, 13 us at the start of a call)
    std::vector<int> order;
    for (int i = 0; i < n; ++i) if (ms[i] && !ms[i]->is_adj) order.push_back(i);
    for (int i = 0; i < n; ++i) if (!ms[i] || ms[i]->is_adj) order.push_back(i);
    if (g_cd_on < 0) { const char* e = getenv("GG_CALL_DBG"); g_cd_on = e ? atoi(e) : 0; }
    const double cd0 = g_cd_on > 0 ? cd_now() : 0;
    for (int s = 0; s < n_sweeps; ++s)
      for (int i : order) {
        const int rc = resample_async_impl(ms[i], ds[i], 1, beta, s + 1 < n_sweeps);
        if (rc) return rc;
      }
    if (g_cd_on > 0) g_cd[0] += cd_now() - cd0;
    const int rcs = gg_sync_many(ms, n);
    if (g_cd_on > 0 && ++g_cdn % 256 == 0) {
      fprintf(stderr, "[call dbg] per call: enqueue %.1f us | copies issued %.1f | drained %.1f | bookkeeping %.1f\n", g_cd[0] / 256, g_cd[1] / 256,
              g_cd[2] / 256, g_cd[3] / 256);
      g_cd[0] = g_cd[1] = g_cd[2] = g_cd[3] = 0;
    }
    return rcs;
  }
  std::vector<int> rcs((size_t)n, GG_OK);
  std::vector<std::string> errs((size_t)n);
  auto run = [&](int i) {
    int rc = gg_resample_async(ms[i], ds[i], n_sweeps, beta);
    if (!rc) rc = gg_sync(ms[i]);
    rcs[(size_t)i] = rc;
    if (rc) errs[(size_t)i] = gg_err_text;   // the error text is thread-local
  };
  std::vector<std::thread> th;
  for (int i = 1; i < n; ++i) th.emplace_back(run, i);
  run(0);
  for (auto& t : th) t.join();
  for (int i = 0; i < n; ++i)
    if (rcs[(size_t)i]) { gg_err_text = errs[(size_t)i]; return rcs[(size_t)i]; }
  return GG_OK;
}

int gg_resample(gg_model* m, const gg_data* d, int32_t n_sweeps, double beta) {
  int rc = gg_resample_async(m, d, n_sweeps, beta);
  if (rc) return rc;
  return gg_sync(m);
}

// Step-wise parts (parity tests, reference-style resample_p / resample_c / resample_m calls): synchronous.
int gg_resample_part(gg_model* m, const gg_data* d, int32_t parts, double beta) {
  if (m) m->ll_cached = false;   // (the state changes: what an epilogue evaluated no longer describes it)
  if (!m) return gg_fail(GG_ERR_INVALID, "null model");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (!(beta >= 0.0 && beta <= 1.0)) return gg_fail(GG_ERR_INVALID, "beta must lie in [0, 1]");
  if (beta != 1.0 && m->eig && m->kind == GG_PROBIT_EIG)
    return gg_fail(GG_ERR_UNSUPPORTED, "tempered conditionals: not defined for the probit eigenmodel (the truncated-normal augmentation has no tempered counterpart)");
  if (m->eig) { gg_eigen_set_beta(m->eig, beta); return gg_eigen_resample(m->eig, d, parts); }
  if (m->ld) { gg_latent_set_beta(m->ld, beta); return gg_latent_resample(m->ld, d, parts, 0); }
  int rc = model_sync(m);
  if (rc) return rc;
  m->beta = beta;
  if (parts != GG_PART_M) rc = check_data(m, d);  // resample_m needs the labels only (adjacency.py:697-702)
  if (rc) return rc;
  // Philox counters are addressed by (seed, site, sweep, index): a part asked for again under the same sweep index
  // would redraw the very same variates (a deterministic map, not a Markov kernel), so it starts a new sweep index.
  parts &= (GG_PART_PARAMS | GG_PART_C | GG_PART_M);
  if (parts & m->parts_done) { m->sweep += 1; m->parts_done = 0; }
  begin_call(m);
  const bool learn_params = (m->kind != GG_BERN_ADJ && m->kind != GG_FIXED_W);
  const bool do_params = (parts & GG_PART_PARAMS) && learn_params;
  const bool do_c = (parts & GG_PART_C) && m->sbm && m->K > 1;
  const bool do_m = (parts & GG_PART_M) && m->sbm;
  const int buf = m->cur;
  if (do_params || do_c) {
    // statistics under the current labels: counted ahead by the last sweep's riders, or one bulk pass now
    const bool have = m->ahead && m->ahead_uid == d->uid;
    if (!have) { drop_ahead(m); if ((rc = enqueue_bulk(m, d, buf, do_c ? 3 : 1, false))) return rc; }
    if ((rc = enqueue_params(m, d, buf, do_params, do_params, do_c, -1, m->sweep))) return rc;
    if (do_c) { m->ahead = false; m->zeroed[buf] = false; }   // the scan below may change labels: not the next sweep's statistics
  }
  if (do_c) {
    const int pw = 1 - m->pcur;
    if (m->wide) {
      // (more than 64 blocks: the one-launch speculative scan stands in for the two kernels below)
      if ((rc = launch_spec_resolve(m, d, buf, m->sweep, pw))) return rc;
    } else {
      if ((rc = launch_speculate(m, d, buf, 0, -1, m->sweep, false))) return rc;
      if ((rc = launch_resolve(m, d, buf, m->sweep, false, true, pw))) return rc;
    }
    m->st.resolve_rounds += 1;
    if ((rc = finish_scan(m, d, buf, m->sweep, false, true, pw))) return rc;
    m->pcur = pw;   // the completed scan left the sort of the final labels there
    m->done_base = m->h_rs[5];
    m->st.label_changes = (uint64_t)m->h_rs[2];
  }
  if (do_m) {
    if ((rc = ensure_perm(m))) return rc;
    k_draw_m<<<1, m->K > 64 ? 256 : 64, (m->K > 64 ? 256 : 64) * sizeof(double), m->stream>>>(m->cnt, m->pi, m->K, m->cfg.seed,
                                                        m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M, m->sweep, m->m);
    GG_HIP(hipGetLastError());
  }
  GG_HIP(hipEventRecord(m->ev_call1, m->stream));
  if ((rc = read_ctl(m))) return rc;
  if (m->h_rs[3]) {
    write_ctl_word(m, 3, 0);
    return gg_fail(GG_ERR_NUMERIC, "a block covariance is not positive definite");
  }
  // resample_m is the last step of a sweep (adjacency.py:615-618, weights.py:394-397): it closes the sweep index
  if (parts & GG_PART_M) { m->sweep += 1; m->parts_done = 0; }
  else m->parts_done |= parts;
  m->timing_dirty = true;
  return GG_OK;
}

int gg_node_lp(gg_model* m, const gg_data* d, int32_t n, double* lp) {
  if (!m || !lp) return gg_fail(GG_ERR_INVALID, "gg_node_lp: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig || m->ld) return gg_fail(GG_ERR_UNSUPPORTED, "gg_node_lp: block models only");
  int rc = model_sync(m);
  if (rc) return rc;
  if ((rc = check_data(m, d))) return rc;
  if (n < 0 || n >= m->N) return gg_fail(GG_ERR_INVALID, "node %d out of range", n);
  m->beta = 1.0;
  begin_call(m);
  const int buf = m->cur;
  if (!(m->ahead && m->ahead_uid == d->uid)) { drop_ahead(m); if ((rc = enqueue_bulk(m, d, buf, 3, false))) return rc; }
  if ((rc = enqueue_params(m, d, buf, false, false, true, -1, m->sweep))) return rc;
  if ((rc = launch_speculate(m, d, buf, 0, n, m->sweep, false))) return rc;
  GG_HIP(hipEventRecord(m->ev_call1, m->stream));
  if ((rc = read_ctl(m))) return rc;
  if (m->h_rs[3]) {
    write_ctl_word(m, 3, 0);
    return gg_fail(GG_ERR_NUMERIC, "a block covariance is not positive definite");
  }
  GG_HIP(hipMemcpy(lp, m->lp_out, (size_t)m->K * 8, hipMemcpyDeviceToHost));
  m->timing_dirty = true;
  return GG_OK;
}

// Closed forms from the K x K tables (shared by the single-GPU and the sharded path).
static int loglik_from_tables(gg_model* m, const gg_data* d, const long long* tabp, const int32_t* cntp, double* out) {
  const int K = m->K, B = m->B, Bx = m->Bx;
  const size_t KK = (size_t)K * K;
  const long long* tab = tabp;
  const int32_t* cnt = cntp;
  const int special = m->cfg.special_case_self_conns;
  double ll = 0.0;
  if (m->is_adj) {
    std::vector<double> p(KK);
    double p_self = 0;
    GG_HIP(hipMemcpy(p.data(), m->p, KK * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(&p_self, m->p_self, 8, hipMemcpyDeviceToHost));
    auto term = [](double P, long long n1, long long n0) {
      double Pc = P < 1e-64 ? 1e-64 : (P > 1 - 1e-64 ? 1 - 1e-64 : P);  // safe_P, abstractions.py:92-94
      double v = 0.0;
      if (n1) v += (double)n1 * log(P);
      if (n0) v += (double)n0 * log(1.0 - Pc);
      return v;
    };
    for (int a = 0; a < K; ++a)
      for (int b = 0; b < K; ++b) {
        long long n1 = tab[(size_t)a * K + b];
        long long pairs = (long long)cnt[a] * cnt[b] - ((a == b && special) ? cnt[a] : 0);
        ll += term(p[(size_t)a * K + b], n1, pairs - n1);
      }
    long long d1 = d->d1, d0 = (long long)m->N - d->d1;
    if (!m->cfg.allow_self_connections && m->kind == GG_SBM_ADJ) {
      // adjacency.py:525-526: diagonal probability 0
      if (!special) {
        // remove the diagonal terms counted in the (a,a) tables above and re-add them with P = 0
        std::vector<int32_t> c(m->N);
        std::vector<uint8_t> dA(m->N);
        GG_HIP(hipMemcpy(c.data(), m->c, (size_t)m->N * 4, hipMemcpyDeviceToHost));
        GG_HIP(hipMemcpy(dA.data(), d->dA, m->N, hipMemcpyDeviceToHost));
        for (int n = 0; n < m->N; ++n) ll -= term(p[(size_t)c[n] * K + c[n]], dA[n] ? 1 : 0, dA[n] ? 0 : 1);
      }
      ll += term(0.0, d1, d0);
    } else if (special) {
      ll += term(p_self, d1, d0);
    }
  } else {
    const int NSI = m->NSI;
    std::vector<double> mu(KK * Bx), sigma(KK * Bx * Bx), mus(Bx), sigs((size_t)Bx * Bx);
    GG_HIP(hipMemcpy(mu.data(), m->mu, mu.size() * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(sigma.data(), m->sigma, sigma.size() * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(mus.data(), m->mu_self, mus.size() * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(sigs.data(), m->sigma_self, sigs.size() * 8, hipMemcpyDeviceToHost));
    const double sc1 = ldexp(1.0, -d->shift), sc2 = ldexp(1.0, -2 * d->shift);
    auto ll_of = [&](const double* mu_, const double* sig_, double n, const long long* hl) -> double {
      if (n == 0) return 0.0;
      double e[GG_MAXNT];
      if (m->wide1 ? gg_gauss_table_wide(mu_, sig_, e) : gg_gauss_table(mu_, sig_, B, e)) return NAN;
      double v = n * e[0];
      for (int s = 0; s < NSI; ++s) {
        double x = gg_hilo_to_double(hl[2 * s], hl[2 * s + 1]) * (s < B ? sc1 : sc2);
        v += e[1 + s] * x;
      }
      return v;
    };
    for (size_t ab = 0; ab < KK; ++ab) {
      const long long* e = &tab[ab * m->NTAB];
      ll += ll_of(&mu[ab * Bx], &sigma[ab * Bx * Bx], (double)e[0], e + 1);
    }
    if (special) {
      long long hl[2 * (GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2)];
      for (int s = 0; s < NSI; ++s) { hl[2 * s] = d->dS[s] >> 32; hl[2 * s + 1] = d->dS[s] & 0xffffffffll; }
      ll += ll_of(mus.data(), sigs.data(), (double)d->d1, hl);
    }
  }
  *out = ll;
  return GG_OK;
}

// Closed forms from the K x K tables:
//   adjacency (abstractions.py:111-115): sum_ab n1 log p + n0 log(1 - clip(p)) + diagonal terms;
//   weights (weights.py:316-331): sum_ab [ n g + h.s1 + Q.s2 ] + self Gaussian on the diagonal edges.
// Block kinds: enqueue the statistics pass, the K x K tables and their copy to pinned staging ...
static int ll_enqueue(gg_model* m, const gg_data* d) {
  int rc = model_sync(m);
  if (rc) return rc;
  if ((rc = check_data(m, d))) return rc;
  // the caller's pattern: asked right after a resample call -> the next call's epilogue evaluates it (launch_spec_resolve_epi)
  // (how many resample calls lay between the last two requests: the next one is expected after as many)
  if (m->calls_since_ll >= 1) { m->ll_asked = true; m->ll_period = m->calls_since_ll; }
  m->calls_since_ll = 0;
  m->ll_from_cache = m->ll_cached && m->ll_uid == d->uid;
  if (m->ll_from_cache) return GG_OK;   // the last call's epilogue left the value of exactly this state
  const int buf = m->cur;
  // (no event bracket around this call: two records and an elapsed-time query cost as much as the kernels they would time)
  m->ev_used = 0; m->ev_sweep.clear(); m->ev_bytes.clear();
  // The statistics of the current labels may have been counted already, for the next sweep, by the riders of the last
  // one (ahead form): the demo loop `resample; log_probability` (examples/demo_model_comparison.py:69-72) then pays
  // no second pass over the data - the tables come from numbers that are resident anyway.  Otherwise: one bulk pass
  // (out-direction only) whose buffer is not reused.
  if (!(m->ahead && m->ahead_uid == d->uid)) {
    drop_ahead(m);
    if ((rc = enqueue_bulk(m, d, buf, 1, false))) return rc;
  }
  if ((rc = enqueue_params(m, d, buf, true, false, false, -1, m->sweep))) return rc;
  const size_t KK = (size_t)m->K * m->K;
  const bool host_form = ll_is_host_form(m);
  if (host_form) {
    GG_HIP(hipMemcpyAsync(m->h_tab, m->tab, KK * m->NTAB * 8, hipMemcpyDeviceToHost, m->stream));
    GG_HIP(hipMemcpyAsync(m->h_tab + KK * m->NTAB, m->cnt, m->K * 4, hipMemcpyDeviceToHost, m->stream));
  } else {
    // (the two result words go straight into the pinned, device-mapped staging: no copy to enqueue behind the kernel)
    gg_ll_args g = fill_ll_args(m, d, reinterpret_cast<double*>(m->h_tab));
    k_loglik<<<1, 256, 0, m->stream>>>(g);
    GG_HIP(hipGetLastError());
  }
  m->ll_host_form = host_form;
  return GG_OK;
}

// ... and, once it has arrived, the value (or, for the one host-form combination, the closed form from the tables).
static int ll_finish(gg_model* m, const gg_data* d, double* out) {
  if (m->ll_from_cache) { *out = m->ll_bad ? NAN : m->ll_val; return GG_OK; }
  GG_HIP(hipStreamSynchronize(m->stream));
  if (!m->ll_host_form) {
    const double* r = reinterpret_cast<const double*>(m->h_tab);
    *out = r[1] != 0.0 ? NAN : r[0];     // (a covariance that is not positive definite: NaN, as the host form returns)
    return GG_OK;
  }
  const size_t KK = (size_t)m->K * m->K;
  return loglik_from_tables(m, d, m->h_tab, reinterpret_cast<const int32_t*>(m->h_tab + KK * m->NTAB), out);
}

int gg_log_likelihood(gg_model* m, const gg_data* d, double* out) {
  if (!m || !out) return gg_fail(GG_ERR_INVALID, "gg_log_likelihood: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig) return gg_eigen_log_likelihood(m->eig, d, out);
  if (m->ld) return gg_latent_log_likelihood(m->ld, d, out);
  int rc = ll_enqueue(m, d);
  return rc ? rc : ll_finish(m, d, out);
}

// gg_log_likelihood for n independent models at once: the block models' passes are all enqueued (each on its own
// stream) before the first is waited for, so that the two halves of FactorizedNetworkDistribution.log_likelihood
// (networks.py:28-31) stream A and W concurrently; the other kinds follow one after the other.
int gg_log_likelihood_many(gg_model** ms, const gg_data* const* ds, int32_t n, double* out) {
  if (n < 0 || (n > 0 && (!ms || !ds || !out))) return gg_fail(GG_ERR_INVALID, "gg_log_likelihood_many: bad arguments");
  for (int i = 0; i < n; ++i) {
    if (!ms[i]) return gg_fail(GG_ERR_INVALID, "gg_log_likelihood_many: null model");
    if (ms[i]->eig || ms[i]->ld) continue;
    GG_HIP(hipSetDevice(ms[i]->cfg.device));
    int rc = ll_enqueue(ms[i], ds[i]);
    if (rc) return rc;
  }
  for (int i = 0; i < n; ++i) {
    GG_HIP(hipSetDevice(ms[i]->cfg.device));
    int rc = (ms[i]->eig || ms[i]->ld) ? gg_log_likelihood(ms[i], ds[i], out + i) : ll_finish(ms[i], ds[i], out + i);
    if (rc) return rc;
  }
  return GG_OK;
}

// ---- log_prior on the device -----------------------------------------------------------------------------------------
// SBMAdjacencyDistribution.log_prior (adjacency.py:538-555): Dirichlet(pi) on m + Beta(tau1, tau0) on every p[a][b] (+ p_self
// with the special case) + sum_n log m[c_n];  SBMGaussianWeightDistribution.log_prior (weights.py:333-357): Dirichlet + the NIW
// density of every pair's (mu, Sigma) (+ the self Gaussian) + sum_n log m[c_n];  SBMGaussianWeightSharedCov (weights.py:554-577):
// Dirichlet + labels only.  The reference's loop `lps.append(model.log_probability(data))` asks for it once per sweep; on the
// host it costs nine state reads and a dozen scipy calls (0.75 ms at K = 16, ten sweeps' worth) - here one small launch per
// model from the state where it lives, the value written to pinned memory.
struct gg_lprior_args {
  int is_adj, K, B, special, shcov;
  double tau1, tau0;
  const double *m, *pi, *p, *p_self, *mu, *sigma, *mu_self, *sigma_self;
  gg_niw niw, niw_self;      // (host structs, by value: sigma_0 packed B x B, as gg_set_state stores it)
  const int32_t* cnt;
  double* out;
};

__device__ static double gg_dev_niw_logp(const double* mu, const double* sigma, const gg_niw& niw, int B) {
  const double* mu0 = niw.mu_0;
  const double* psi = niw.sigma_0;
  const double nu = niw.nu_0, kappa = niw.kappa_0;
  double P[GG_MAXB * GG_MAXB], Pp[GG_MAXB * GG_MAXB], logdet = 0.0, logdet_psi = 0.0;
  if (gg_spd_inv_logdet(sigma, B, P, &logdet) || gg_spd_inv_logdet(psi, B, Pp, &logdet_psi)) return nan("");
  double tr = 0.0, quad = 0.0;
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < B; ++j) {
      tr += psi[i * B + j] * P[j * B + i];
      quad += (mu[i] - mu0[i]) * P[i * B + j] * (mu[j] - mu0[j]);
    }
  double mg = 0.25 * B * (B - 1) * 1.1447298858494002;          // log Gamma_B(nu / 2) = B (B - 1) / 4 log pi + sum_j lgamma(nu / 2 + (1 - j) / 2)
  for (int j = 1; j <= B; ++j) mg += lgamma(0.5 * nu + 0.5 * (1 - j));
  const double lp_iw = 0.5 * nu * logdet_psi - 0.5 * nu * B * 0.6931471805599453 - mg - 0.5 * (nu + B + 1) * logdet - 0.5 * tr;
  const double lp_n = -0.5 * B * 1.8378770664093453 - 0.5 * (logdet - B * log(kappa)) - 0.5 * kappa * quad;
  return lp_iw + lp_n;
}

__global__ __launch_bounds__(256) void k_log_prior(gg_lprior_args g) {
  __shared__ double s_red[256];
  const int t = (int)threadIdx.x, K = g.K;
  double acc = 0.0;
  for (int k = t; k < K; k += 256) {
    const double lm = log(g.m[k]);
    acc += (g.pi[k] - 1.0) * lm - lgamma(g.pi[k]);
    if (g.cnt[k] > 0) acc += (double)g.cnt[k] * lm;            // sum_n log m[c_n]
  }
  if (g.is_adj) {
    const double bl = lgamma(g.tau1) + lgamma(g.tau0) - lgamma(g.tau1 + g.tau0);
    for (int ab = t; ab < K * K; ab += 256) acc += (g.tau1 - 1.0) * log(g.p[ab]) + (g.tau0 - 1.0) * log1p(-g.p[ab]) - bl;
    if (g.special && t == 0) acc += (g.tau1 - 1.0) * log(g.p_self[0]) + (g.tau0 - 1.0) * log1p(-g.p_self[0]) - bl;
  } else if (!g.shcov) {
    const int B = g.B;
    for (int ab = t; ab < K * K; ab += 256) acc += gg_dev_niw_logp(g.mu + (size_t)ab * B, g.sigma + (size_t)ab * B * B, g.niw, B);
    if (g.special && t == 0) acc += gg_dev_niw_logp(g.mu_self, g.sigma_self, g.niw_self, B);
  }
  s_red[t] = acc;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {          // fixed tree: the value does not depend on scheduling
    if (t < s) s_red[t] += s_red[t + s];
    __syncthreads();
  }
  if (t == 0) {
    double sp = 0.0;
    for (int k = 0; k < K; ++k) sp += g.pi[k];
    g.out[0] = s_red[0] + lgamma(sp);
  }
}

static int lprior_enqueue(gg_model* m) {
  if (m->eig || m->ld || !(m->kind == GG_SBM_ADJ || m->kind == GG_SBM_GAUSS_W || m->kind == GG_SBM_GAUSS_SHCOV_W))
    return gg_fail(GG_ERR_UNSUPPORTED, "gg_log_prior: SBM adjacency / weight kinds only (the other kinds' priors are O(1) host formulas)");
  GG_HIP(hipSetDevice(m->cfg.device));
  int rc = model_sync(m);
  if (rc) return rc;
  if ((rc = ensure_perm(m))) return rc;          // (block sizes of the labels as they are)
  if (!m->h_lp) GG_HIP(hipHostMalloc((void**)&m->h_lp, 2 * sizeof(double), hipHostMallocDefault));
  gg_lprior_args g;
  memset(&g, 0, sizeof(g));
  g.is_adj = m->is_adj ? 1 : 0; g.K = m->K; g.B = m->Bx; g.special = m->cfg.special_case_self_conns; g.shcov = m->kind == GG_SBM_GAUSS_SHCOV_W;
  g.tau1 = m->cfg.tau1; g.tau0 = m->cfg.tau0;
  g.m = m->m; g.pi = m->pi; g.p = m->p; g.p_self = m->p_self; g.mu = m->mu; g.sigma = m->sigma; g.mu_self = m->mu_self;
  g.sigma_self = m->sigma_self; g.niw = m->niw; g.niw_self = m->niw_self; g.cnt = m->cnt; g.out = m->h_lp;
  k_log_prior<<<1, 256, 0, m->stream>>>(g);
  GG_HIP(hipGetLastError());
  return GG_OK;
}

extern "C" int gg_log_prior_many(gg_model** ms, int32_t n, double* out) {
  if (n < 0 || (n > 0 && (!ms || !out))) return gg_fail(GG_ERR_INVALID, "gg_log_prior_many: bad arguments");
  for (int i = 0; i < n; ++i) {
    if (!ms[i]) return gg_fail(GG_ERR_INVALID, "gg_log_prior_many: null model");
    int rc = lprior_enqueue(ms[i]);
    if (rc) return rc;
  }
  for (int i = 0; i < n; ++i) {
    GG_HIP(hipSetDevice(ms[i]->cfg.device));
    GG_HIP(hipStreamSynchronize(ms[i]->stream));
    out[i] = ms[i]->h_lp[0];
  }
  return GG_OK;
}

// Microbenchmark of the bulk statistics kernel alone (both directions, current labels): average
// device time per launch over `reps` launches, measured with HIP events on the model's stream.
int gg_bench_bulk(gg_model* m, const gg_data* d, int32_t reps, double* ms_per_launch, double* bytes_per_launch) {
  if (!m || !ms_per_launch) return gg_fail(GG_ERR_INVALID, "gg_bench_bulk: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig || m->ld) return gg_fail(GG_ERR_UNSUPPORTED, "gg_bench_bulk: block models only");
  int rc = model_sync(m);
  if (rc) return rc;
  if ((rc = check_data(m, d))) return rc;
  drop_ahead(m);
  const int t0 = m->timing;
  m->timing = 0;
  const int buf = m->cur;
  if ((rc = enqueue_bulk(m, d, buf, 3, false))) return rc;  // warm-up
  GG_HIP(hipEventRecord(m->ev_call0, m->stream));
  for (int r = 0; r < reps; ++r) {
    m->zeroed[buf] = true;  // accumulate on top: contents are irrelevant here
    if ((rc = enqueue_bulk(m, d, buf, 3, false))) return rc;
  }
  GG_HIP(hipEventRecord(m->ev_call1, m->stream));
  GG_HIP(hipEventSynchronize(m->ev_call1));
  float ms = 0;
  GG_HIP(hipEventElapsedTime(&ms, m->ev_call0, m->ev_call1));
  *ms_per_launch = ms / (reps > 0 ? reps : 1);
  if (bytes_per_launch) *bytes_per_launch = 2.0 * m->N * (double)m->N * (m->is_adj ? 0.125 : 4.0 * m->B);
  m->zeroed[buf] = false;
  m->timing = t0;
  m->st.bulk_bytes = 0; m->st.bulk_launches = 0;
  return GG_OK;
}

// ---- sharded single chain (include/graphgibbs.h) ----------------------------------
// Communicator: RCCL (device buffers, collectives enqueued on the engine stream: xGMI inside a node) or a pair of host
// callbacks (the protocol test harness: gloo through ctypes; buffers are staged through the host).
struct gg_comm {
  int rank, world, device;
  ncclComm_t nccl;
  gg_host_allreduce_fn ar;
  gg_host_allgather_fn ag;
  void* user;
  std::vector<long long> h64;
  std::vector<int32_t> h32s, h32r;
  unsigned long long collectives;   // collectives issued through this communicator (none with one rank: nothing travels)
  unsigned long long rounds;        // rounds of the sharded scans driven through it (each is one all-gather when world > 1)
};

static int comm_allreduce_i64(gg_comm* c, long long* dbuf, size_t n, hipStream_t st) {
  if (c->world == 1) return GG_OK;      // (one rank: nothing travels, nothing is counted - ADVICE r3)
  c->collectives += 1;
  if (c->nccl) {
    const ncclResult_t r = ncclAllReduce(dbuf, dbuf, n, ncclInt64, ncclSum, c->nccl, st);
    if (r != ncclSuccess) return gg_fail(GG_ERR_HIP, "ncclAllReduce failed: %s", ncclGetErrorString(r));
    return GG_OK;
  }
  c->h64.resize(n);
  GG_HIP(hipMemcpyAsync(c->h64.data(), dbuf, n * 8, hipMemcpyDeviceToHost, st));
  GG_HIP(hipStreamSynchronize(st));
  if (c->ar(c->user, reinterpret_cast<int64_t*>(c->h64.data()), (int64_t)n)) return gg_fail(GG_ERR_HIP, "host all-reduce callback failed");
  GG_HIP(hipMemcpyAsync(dbuf, c->h64.data(), n * 8, hipMemcpyHostToDevice, st));
  GG_HIP(hipStreamSynchronize(st));
  return GG_OK;
}

static int comm_allgather_i32(gg_comm* c, const int32_t* dsend, int32_t* drecv, size_t n, hipStream_t st) {
  if (c->world > 1) c->collectives += 1;
  if (c->world == 1) {
    GG_HIP(hipMemcpyAsync(drecv, dsend, n * 4, hipMemcpyDeviceToDevice, st));
    return GG_OK;
  }
  if (c->nccl) {
    const ncclResult_t r = ncclAllGather(dsend, drecv, n, ncclInt32, c->nccl, st);
    if (r != ncclSuccess) return gg_fail(GG_ERR_HIP, "ncclAllGather failed: %s", ncclGetErrorString(r));
    return GG_OK;
  }
  c->h32s.resize(n); c->h32r.resize(n * c->world);
  GG_HIP(hipMemcpyAsync(c->h32s.data(), dsend, n * 4, hipMemcpyDeviceToHost, st));
  GG_HIP(hipStreamSynchronize(st));
  if (c->ag(c->user, c->h32s.data(), c->h32r.data(), (int64_t)n)) return gg_fail(GG_ERR_HIP, "host all-gather callback failed");
  GG_HIP(hipMemcpyAsync(drecv, c->h32r.data(), n * 4 * c->world, hipMemcpyHostToDevice, st));
  GG_HIP(hipStreamSynchronize(st));
  return GG_OK;
}

int gg_comm_unique_id(uint8_t* out128) {
  if (!out128) return gg_fail(GG_ERR_INVALID, "gg_comm_unique_id: null argument");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return gg_fail(GG_ERR_HIP, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, 128);
  return GG_OK;
}

int gg_comm_create_rccl(gg_comm** out, int32_t rank, int32_t world, const uint8_t* id128, int32_t device) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return gg_fail(GG_ERR_INVALID, "gg_comm_create_rccl: bad arguments");
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  gg_comm* c = new gg_comm();
  c->rank = rank; c->world = world; c->device = device; c->nccl = nullptr; c->ar = nullptr; c->ag = nullptr; c->user = nullptr;
  c->collectives = 0; c->rounds = 0;
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  const ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) { delete c; return gg_fail(GG_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r)); }
  *out = c;
  return GG_OK;
}

int gg_comm_create_host(gg_comm** out, int32_t rank, int32_t world, gg_host_allreduce_fn ar, gg_host_allgather_fn ag, void* user,
                        int32_t device) {
  if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && (!ar || !ag)))
    return gg_fail(GG_ERR_INVALID, "gg_comm_create_host: bad arguments");
  int rc = gg_ensure_device(device);
  if (rc) return rc;
  gg_comm* c = new gg_comm();
  c->rank = rank; c->world = world; c->device = device; c->nccl = nullptr; c->ar = ar; c->ag = ag; c->user = user;
  c->collectives = 0; c->rounds = 0;
  *out = c;
  return GG_OK;
}

int gg_comm_destroy(gg_comm* c) {
  if (!c) return GG_OK;
  if (c->nccl) { hipSetDevice(c->device); ncclCommDestroy(c->nccl); }
  delete c;
  return GG_OK;
}

int gg_comm_info(const gg_comm* c, int32_t* rank, int32_t* world, int32_t* is_rccl, uint64_t* collectives, uint64_t* rounds) {
  if (!c) return gg_fail(GG_ERR_INVALID, "null communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (is_rccl) *is_rccl = c->nccl ? 1 : 0;
  if (collectives) *collectives = c->collectives;
  if (rounds) *rounds = c->rounds;
  return GG_OK;
}

static int shard_check(gg_model* m, const gg_data* d, const gg_comm* c) {
  if (!m || !c) return gg_fail(GG_ERR_INVALID, "null model or communicator");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig || m->ld) return gg_fail(GG_ERR_UNSUPPORTED, "sharding: block models only");
  if (m->kind != GG_SBM_ADJ && m->kind != GG_SBM_GAUSS_W) return gg_fail(GG_ERR_UNSUPPORTED, "sharding: SBM kinds only");
  if (c->device != m->cfg.device) return gg_fail(GG_ERR_INVALID, "communicator on device %d, model on device %d", c->device, m->cfg.device);
  return d ? check_data(m, d) : GG_OK;
}

// Global diagonal statistics (label independent): all-reduced once per network, then cached in the model.
static int shard_global_diag(gg_model* m, const gg_data* d, gg_comm* c) {
  if (m->sh_diag_uid == d->uid) return GG_OK;
  const int nd = 1 + gg_nsi(GG_MAXB);
  long long h[1 + GG_MAXB + GG_MAXB * (GG_MAXB + 1) / 2];
  h[0] = d->d1;
  for (int i = 0; i < gg_nsi(GG_MAXB); ++i) h[1 + i] = d->dS[i];
  GG_HIP(hipMemcpyAsync(m->sh_buf, h, nd * 8, hipMemcpyHostToDevice, m->stream));
  int rc = comm_allreduce_i64(c, m->sh_buf, nd, m->stream);
  if (rc) return rc;
  GG_HIP(hipMemcpyAsync(m->sh_diag, m->sh_buf, nd * 8, hipMemcpyDeviceToHost, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  m->sh_diag_uid = d->uid;
  return GG_OK;
}

// K x K tables of the whole network under the current labels: local bulk pass + pair tables, all-reduced in place (device).
static int shard_global_tables(gg_model* m, const gg_data* d, gg_comm* c, int dirs) {
  int rc;
  const int buf = m->cur;
  if ((rc = enqueue_bulk(m, d, buf, dirs, false))) return rc;
  if ((rc = enqueue_params(m, d, buf, true, false, false, -1, m->sweep))) return rc;
  return comm_allreduce_i64(c, m->tab, (size_t)m->K * m->K * m->NTAB, m->stream);
}

// One round of the sharded scan = these two launches around the all-gather (gg_block_kernels.h).  W: width of the band.
static int launch_shard_eval(gg_model* m, const gg_data* d, int buf, int round, int W) {
  gg_node_ctx x = node_ctx(m, d, buf);
  x.cntn = m->sh_cntn;   // block sizes as each own node sees them (hypotheses at earlier positions included)
  // (33..64 blocks, NT <= 3: eight nodes per workgroup in two passes of four - gg_spec_eval_split8)
  const int K = m->K, per = (m->Kp == 64 && m->NT <= 3) ? 8 : 256 / m->Kp;
  // own nodes of a band: at most ceil(W / (block * world)) own blocks (+1 for a band that starts inside one)
  int count = m->Nl;
  if (m->own.lb >= 0) {
    const long long span = (long long)m->own.world << m->own.lb;
    const long long c2 = ((long long)W / span + 2) << m->own.lb;
    if (c2 < count) count = (int)c2;
  } else if (W < count) count = W;
  int nb = (count + per - 1) / per + 1;       // (+1: workgroups start at a multiple of `per` at or before the first node)
  if (nb < 1) nb = 1;                         // (a rank without nodes in the band still reports "no surprise")
  const uint32_t site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C;
  size_t lds = (size_t)2 * K * per * m->NT * sizeof(double);
  const int stage = (lds + table_bytes(m) <= 64 * 1024) ? 1 : 0;
  if (stage) lds += table_bytes(m);
  else if (m->Kp == 64) lds += (size_t)4 * 2 * m->NT * 64 * sizeof(double);   // gg_lp_split4's meeting place
  if (m->Kp > 64) lds += (size_t)2 * 256 * sizeof(double) + 16 * sizeof(int) + 8;   // the cross-wave categorical's scratch
#define SHE(NTV, ADJV)                                                                                                     \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set && lds > 48 * 1024) {                                                                                    \
      GG_HIP(hipFuncSetAttribute((const void*)k_shard_eval<NTV, ADJV>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024)); \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    k_shard_eval<NTV, ADJV><<<nb, 256, lds, m->stream>>>(x, m->own, W, m->cfg.seed, site, m->sweep, m->cand, m->c, m->sh_H,  \
                                                         m->sh_state, round, m->N, stage, m->sh_trip);                     \
  } while (0)
  if (m->is_adj) SHE(2, true);
  else if (m->B == 1) SHE(3, false);
  else if (m->B == 2) SHE(6, false);
  else if (m->B == 3) SHE(10, false);
  else SHE(15, false);
#undef SHE
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int launch_shard_apply(gg_model* m, const gg_data* d, int buf, int round, int W, const int32_t* all, int world) {
  gg_node_ctx x = node_ctx(m, d, buf);
  dim3 grd((m->Nl + 255) / 256 > 0 ? (m->Nl + 255) / 256 : 1, 8);
  const size_t lds = (size_t)4 * GG_SHARD_M * world * sizeof(int);
#define SHA(NTV, ADJV)                                                                                                 \
  k_shard_apply<NTV, ADJV><<<grd, 256, lds, m->stream>>>(x, m->own, W, m->c, m->cnt, m->sh_H, m->rs, m->sh_state, round, m->N, all, \
                                                         world, m->sh_cntn, gcnt(m, buf, 0), gcnt(m, buf, 1), gs(m, buf, 0), \
                                                         gs(m, buf, 1), d->Ab, d->AbT, d->Wq, d->WqT, fold1)
  static int fold1 = -1;      // GG_SHARD_FOLD1=1 (diagnosis): the whole band folded / accepted by ONE workgroup
  if (fold1 < 0) { const char* e = getenv("GG_SHARD_FOLD1"); fold1 = e ? atoi(e) : 0; }
  if (m->is_adj) SHA(2, true);
  else if (m->B == 1) SHA(3, false);
  else if (m->B == 2) SHA(6, false);
  else if (m->B == 3) SHA(10, false);
  else SHA(15, false);
#undef SHA
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int shard_alloc(gg_model* m, const gg_comm* c) {
  int rc;
  if (!m->sh_buf) {
    if ((rc = dmalloc(&m->sh_buf, 64)) || (rc = dmalloc(&m->sh_trip, (size_t)2 * GG_SHARD_M + 4)) ||
        (rc = dmalloc(&m->sh_all, (size_t)2 * GG_SHARD_M * c->world + 4)))
      return rc;
    m->sh_world = c->world;
  }
  return GG_OK;
}

// Width of the band of the next scan.  A round costs one collective and two launches whatever the band's width, and a band is
// closed after 1 + (depth of mutually dependent decisions) rounds, so the band should be as wide as the reports can describe:
// a rank reports GG_SHARD_M surprises per round, and the first round of a band meets every node that changes its label - the
// previous scan's count (replicated: the same on every rank) sizes the band so that a round's reports hold about half of
// what they can.  A burn-in scan (most labels change) runs 4096-node bands, a converged chain one band for the whole scan.
// GG_SHARD_BAND fixes it (tests cross several widths: the chain does not depend on it).
static int shard_band(const gg_model* m, int world, long long last_changes) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("GG_SHARD_BAND"); forced = e ? atoi(e) : 0; }
  long long W;
  if (forced > 0) W = forced;
  else {
    const long long room = (long long)GG_SHARD_M * world / 2;
    W = last_changes <= 0 ? GG_SHARD_WMAX : (long long)m->N * room / last_changes;
  }
  long long p = 256;                       // (a power of two: band boundaries fall on ownership-block boundaries)
  while (p * 2 <= W && p * 2 <= GG_SHARD_WMAX) p *= 2;
  if (forced <= 0 && p < 4096) p = 4096;
  return (int)p;
}

// n half-sweeps (adjacency.py:615-618 / weights.py:394-397) of a chain whose nodes are sharded over the communicator's
// ranks.  Per half-sweep: one all-reduce of the integer tables; the scan in rounds of
//   speculate (own nodes of the band, under the hypotheses) -> first surprises -> ALL-GATHER of (node, labels) x GG_SHARD_M
//   -> fold / accept up to the earliest surprise -> moves of the own nodes' statistics         (k_shard_* kernels),
// all enqueued on the engine stream in batches, the host reading the scan position once per batch.  Every rank issues the
// same collectives (the position it reads is the same everywhere), and rounds beyond the scan's end are no-ops.
int gg_shard_sweep(gg_model* m, const gg_data* d, gg_comm* c, int32_t n) {
  if (m) m->ll_cached = false;   // (the state changes: what an epilogue evaluated no longer describes it)
  int rc = shard_check(m, d, c);
  if (rc) return rc;
  if (!m->sharded && c->world > 1) return gg_fail(GG_ERR_INVALID, "gg_shard_sweep: the model was not created with a node shard");
  if (m->own.lb >= 0 && m->own.world != c->world)
    return gg_fail(GG_ERR_INVALID, "gg_shard_sweep: the model's nodes are dealt out to %d ranks, the communicator has %d", m->own.world, c->world);
  if ((rc = shard_alloc(m, c))) return rc;
  if (!m->sh_cntn) {
    if ((rc = dmalloc(&m->sh_cntn, (size_t)(m->Nl > 0 ? m->Nl : 1) * m->K)) || (rc = dmalloc(&m->sh_H, (size_t)GG_SHARD_WMAX)) ||
        (rc = dmalloc(&m->sh_state, 1)))
      return rc;
  }
  if (m->sh_world != c->world) return gg_fail(GG_ERR_INVALID, "gg_shard_sweep: the communicator's size changed");
  drop_ahead(m);
  m->beta = 1.0;
  if ((rc = shard_global_diag(m, d, c))) return rc;
  gg_data dg = *d;     // kernel arguments take the GLOBAL diagonal statistics
  dg.d1 = m->sh_diag[0];
  for (int i = 0; i < gg_nsi(GG_MAXB); ++i) dg.dS[i] = m->sh_diag[1 + i];
  const int buf = m->cur;
  for (int s = 0; s < n; ++s) {
    if ((rc = shard_global_tables(m, d, c, 3))) return rc;
    m->shard_from_tab = true;
    rc = enqueue_params(m, &dg, buf, false, true, true, -1, m->sweep);
    m->shard_from_tab = false;
    if (rc) return rc;
    if (m->K > 1) {
      {
        const size_t cells = (size_t)m->Nl * m->K > (size_t)GG_SHARD_WMAX ? (size_t)m->Nl * m->K : (size_t)GG_SHARD_WMAX;
        k_shard_init<<<(unsigned)((cells + 255) / 256), 256, 0, m->stream>>>(m->cnt, m->sh_cntn, m->sh_H, m->Nl, m->K, m->rs, m->sh_state);
        GG_HIP(hipGetLastError());
      }
      if (m->sh_band <= 0) m->sh_band = shard_band(m, c->world, (long long)m->N);   // (first scan: nothing known - burn-in width)
      const int W = m->sh_band;
      const long long chg0 = (long long)(uint32_t)m->h_rs[2];      // (the running count the previous scan left: replicated)
      // Rounds are enqueued in batches without host synchronisation; a round beyond the scan's end is a no-op on the device
      // but still a collective on the wire, so the first batch is what the previous scan needed plus a margin (the same
      // number on every rank: it comes from the replicated control block), later ones a quarter of that.
      int batch = m->sh_hint > 0 ? m->sh_hint : 8;
      int round = 0;      // (selects the copy of the scan state a round reads: the same sequence on every rank)
      // one rank: the "gathered" reports are the rank's own - no copy
      const int32_t* all = c->world == 1 ? m->sh_trip : m->sh_all;
      for (;;) {
        for (int r = 0; r < batch; ++r, ++round) {
          // one round: own nodes of the band behind the scan position evaluated under the hypotheses, the first GG_SHARD_M
          // surprises compacted by the launch's last workgroup -> ALL-GATHER -> every workgroup amends / accepts / resets its
          // slice of the band up to the earliest surprise, moves of the owned nodes' statistics
          if ((rc = launch_shard_eval(m, d, buf, round, W))) return rc;
          if (c->world > 1) { if ((rc = comm_allgather_i32(c, m->sh_trip, m->sh_all, 2 * GG_SHARD_M, m->stream))) return rc; }
          c->rounds += 1;
          if ((rc = launch_shard_apply(m, d, buf, round, W, all, c->world))) return rc;
        }
        GG_HIP(hipMemcpyAsync(m->h_rs, m->rs, 24 * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
        GG_HIP(hipStreamSynchronize(m->stream));
        if (m->h_rs[3]) { write_ctl_word(m, 3, 0); return gg_fail(GG_ERR_NUMERIC, "a block covariance is not positive definite"); }
        if (m->h_rs[GG_RS_SPOS] >= m->N) break;
        batch = m->sh_hint / 4 > 4 ? m->sh_hint / 4 : 4;
      }
      m->st.label_changes = (uint64_t)m->h_rs[2];
      m->st.resolve_rounds += (uint64_t)m->h_rs[GG_RS_ROUNDS];
      {
        const int used = m->h_rs[GG_RS_ROUNDS];
        const int hint = used + (used / 8 > 1 ? used / 8 : 1);
        m->sh_hint = hint < 2 ? 2 : (hint > 4096 ? 4096 : hint);
      }
      m->sh_band = shard_band(m, c->world, (long long)(uint32_t)m->h_rs[2] - chg0);
      m->perm_valid = false;
    }
    k_draw_m<<<1, m->K > 64 ? 256 : 64, (m->K > 64 ? 256 : 64) * sizeof(double), m->stream>>>(m->cnt, m->pi, m->K, m->cfg.seed,
                                                        m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M, m->sweep, m->m);
    GG_HIP(hipGetLastError());
    m->sweep += 1;
    m->st.sweeps += 1;
    m->zeroed[buf] = false;
  }
  GG_HIP(hipStreamSynchronize(m->stream));
  return GG_OK;
}

// log-likelihood of the sharded chain (abstractions.py:111-115, weights.py:316-331) from the all-reduced tables.
int gg_shard_log_likelihood(gg_model* m, const gg_data* d, gg_comm* c, double* out) {
  int rc = shard_check(m, d, c);
  if (rc) return rc;
  if (!out) return gg_fail(GG_ERR_INVALID, "gg_shard_log_likelihood: null output");
  if ((rc = shard_alloc(m, c))) return rc;
  drop_ahead(m);
  if ((rc = shard_global_diag(m, d, c))) return rc;
  if ((rc = shard_global_tables(m, d, c, 1))) return rc;
  const size_t KK = (size_t)m->K * m->K;
  GG_HIP(hipMemcpyAsync(m->h_tab, m->tab, KK * m->NTAB * 8, hipMemcpyDeviceToHost, m->stream));
  GG_HIP(hipMemcpyAsync(m->h_tab + KK * m->NTAB, m->cnt, m->K * 4, hipMemcpyDeviceToHost, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  gg_data dg = *d;
  dg.d1 = m->sh_diag[0];
  for (int i = 0; i < gg_nsi(GG_MAXB); ++i) dg.dS[i] = m->sh_diag[1 + i];
  return loglik_from_tables(m, &dg, m->h_tab, reinterpret_cast<const int32_t*>(m->h_tab + KK * m->NTAB), out);
}

// Predictive log-likelihood of a held-out (N+1)-th node, per block of the new node (k_pred_ll).
int gg_predictive_ll(gg_model* m, const uint8_t* a_row, const uint8_t* a_col, const double* w_row, const double* w_col, double* ll_k) {
  if (!m || !a_row || !a_col || !ll_k) return gg_fail(GG_ERR_INVALID, "gg_predictive_ll: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig || m->ld) return gg_fail(GG_ERR_UNSUPPORTED, "gg_predictive_ll: block models only (the other kinds draw latent features per sample: host code)");
  if (m->sharded) return gg_fail(GG_ERR_UNSUPPORTED, "gg_predictive_ll: not for node shards");
  if (m->joint) return gg_fail(GG_ERR_UNSUPPORTED, "gg_predictive_ll: not for the joint model (the reference defines no predictive parameters for it)");
  if (!m->is_adj && (!w_row || !w_col)) return gg_fail(GG_ERR_INVALID, "gg_predictive_ll: weight kinds need the weight row and column");
  int rc = model_sync(m);
  if (rc) return rc;
  const int N = m->N, K = m->K, B = m->Bx;      // (the weights as the caller hands them: doubles, the parameters' dimension)
  const size_t na = (size_t)N + 1, nw = ((size_t)N + 1) * B;
  const size_t bytes = 2 * na + (m->is_adj ? 0 : 2 * nw * 8) + 16;
  const size_t need = bytes + (size_t)(K + 1) * 8 + 64;
  if (m->pred_cap < need) {   // (no allocation, no free and so no implicit device synchronisation per call: ADVICE r2)
    if (m->pred_buf) { GG_HIP(hipStreamSynchronize(m->stream)); GG_HIP(hipFree(m->pred_buf)); m->pred_buf = nullptr; m->pred_cap = 0; }
    GG_HIP(hipMalloc((void**)&m->pred_buf, need));
    m->pred_cap = need;
  }
  unsigned char* dbuf = m->pred_buf;
  // layout: [w_row | w_col] (doubles first: alignment) | a_row | a_col | out[K + 1]
  double* d_w = reinterpret_cast<double*>(dbuf);
  unsigned char* d_a = dbuf + (m->is_adj ? 0 : 2 * nw * 8);
  double* d_out = reinterpret_cast<double*>(dbuf + ((bytes + 7) & ~(size_t)7));
  if (!m->is_adj) {
    if (gg_first_nonfinite(w_row, nw) >= 0 || gg_first_nonfinite(w_col, nw) >= 0) return gg_fail(GG_ERR_INVALID, "gg_predictive_ll: non-finite weight");
    GG_HIP(hipMemcpyAsync(d_w, w_row, nw * 8, hipMemcpyHostToDevice, m->stream));
    GG_HIP(hipMemcpyAsync(d_w + nw, w_col, nw * 8, hipMemcpyHostToDevice, m->stream));
  }
  GG_HIP(hipMemcpyAsync(d_a, a_row, na, hipMemcpyHostToDevice, m->stream));
  GG_HIP(hipMemcpyAsync(d_a + na, a_col, na, hipMemcpyHostToDevice, m->stream));
  GG_HIP(hipMemsetAsync(d_out, 0, (size_t)(K + 1) * 8, m->stream));
  gg_pred_args g;
  memset(&g, 0, sizeof(g));
  g.is_adj = m->is_adj ? 1 : 0; g.K = K; g.B = B; g.N = N;
  g.special = m->cfg.special_case_self_conns ? 1 : 0;
  g.allow_self = (m->kind == GG_SBM_ADJ && !m->cfg.allow_self_connections) ? 0 : 1;
  // the new node's self entry: the SBM adjacency model uses p[c2, c2] (adjacency.py:562-573), the K = 1 adjacency kinds their
  // p_self; the weight kinds their self Gaussian when self connections are special-cased (weights.py:144-152, 375-389)
  g.use_self_param = m->is_adj ? (m->kind != GG_SBM_ADJ && g.special) : g.special;
  g.c = m->c; g.p = m->p; g.p_self = m->p_self; g.mu = m->mu; g.sigma = m->sigma; g.mu_self = m->mu_self; g.sigma_self = m->sigma_self;
  g.a_row = d_a; g.a_col = d_a + na; g.w_row = d_w; g.w_col = d_w + nw;
  g.out = d_out;
  const size_t lds = m->is_adj ? (size_t)4 * K * 8 : (size_t)2 * K * m->NT * 8;
  k_pred_ll<<<K, 256, lds, m->stream>>>(g);
  hipError_t e = hipGetLastError();
  std::vector<double> h((size_t)K + 1);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost, m->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
  if (e != hipSuccess) return gg_fail(GG_ERR_HIP, "gg_predictive_ll failed: %s", hipGetErrorString(e));
  if (h[(size_t)K] != 0.0) return gg_fail(GG_ERR_NUMERIC, "gg_predictive_ll: a block covariance is not positive definite");
  for (int k = 0; k < K; ++k) ll_k[k] = h[(size_t)k];
  return GG_OK;
}

__global__ void k_tab_accumulate(long long* __restrict__ dst, const long long* __restrict__ src, int n, int first) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = first ? src[i] : dst[i] + src[i];
}

// The reference's list-taking forms.  SBM adjacency: resample_p(As) (adjacency.py:620-648) - the block-pair counts are summed
// over the graphs, every graph offering every pair, then ONE Beta draw per pair (and one for p_self from the summed
// diagonals).  Eigenmodels: resample(networks) (eigenmodel.py:836-857, :246-272) - auxiliary variables per graph, every
// conditional from the statistics summed over the graphs (gg_eigen.hip).
int gg_resample_part_many(gg_model* m, const gg_data* const* ds, int32_t nd, int32_t parts, double beta) {
  if (m) m->ll_cached = false;   // (the state changes: what an epilogue evaluated no longer describes it)
  if (!m || !ds || nd < 1) return gg_fail(GG_ERR_INVALID, "gg_resample_part_many: null argument or empty list");
  for (int i = 0; i < nd; ++i) if (!ds[i]) return gg_fail(GG_ERR_INVALID, "gg_resample_part_many: null data handle in the list");
  if (nd == 1) return gg_resample_part(m, ds[0], parts, beta);
  GG_HIP(hipSetDevice(m->cfg.device));
  if (!(beta >= 0.0 && beta <= 1.0)) return gg_fail(GG_ERR_INVALID, "beta must lie in [0, 1]");
  if (m->eig) {
    if (beta != 1.0 && m->kind == GG_PROBIT_EIG)
      return gg_fail(GG_ERR_UNSUPPORTED, "tempered conditionals: not defined for the probit eigenmodel");
    gg_eigen_set_beta(m->eig, beta);
    return gg_eigen_resample_many(m->eig, ds, nd, parts);
  }
  if (m->kind != GG_SBM_ADJ || m->sharded || parts != GG_PART_PARAMS)
    return gg_fail(GG_ERR_UNSUPPORTED, "gg_resample_part_many: several graphs are defined for SBM adjacency resample_p (GG_PART_PARAMS) and for the eigenmodels");
  int rc = model_sync(m);
  if (rc) return rc;
  for (int i = 0; i < nd; ++i) if ((rc = check_data(m, ds[i]))) return rc;
  m->beta = beta;
  if (GG_PART_PARAMS & m->parts_done) { m->sweep += 1; m->parts_done = 0; }
  begin_call(m);
  const size_t nt = (size_t)m->K * m->K * m->NTAB;
  if (!m->tab_sum && (rc = dmalloc(&m->tab_sum, nt))) return rc;
  drop_ahead(m);
  const int buf = m->cur;
  long long d1 = 0;
  for (int i = 0; i < nd; ++i) {
    // this graph's K x K tables under the current labels (one out-direction pass), added to the sum
    if ((rc = enqueue_bulk(m, ds[i], buf, 1, false))) return rc;      // (zeroes the buffer first)
    if ((rc = enqueue_params(m, ds[i], buf, true, false, false, -1, m->sweep))) return rc;
    k_tab_accumulate<<<(unsigned)((nt + 255) / 256), 256, 0, m->stream>>>(m->tab_sum, m->tab, (int)nt, i == 0 ? 1 : 0);
    GG_HIP(hipGetLastError());
    d1 += ds[i]->d1;
  }
  GG_HIP(hipMemcpyAsync(m->tab, m->tab_sum, nt * sizeof(long long), hipMemcpyDeviceToDevice, m->stream));
  gg_data dg = *ds[0];      // kernel arguments take the summed diagonal counts
  dg.d1 = d1;
  m->ngraphs = nd;
  m->shard_from_tab = true;
  rc = enqueue_params(m, &dg, buf, false, true, false, -1, m->sweep);
  m->shard_from_tab = false;
  m->ngraphs = 1;
  if (rc) return rc;
  GG_HIP(hipEventRecord(m->ev_call1, m->stream));
  if ((rc = read_ctl(m))) return rc;
  m->parts_done |= GG_PART_PARAMS;
  m->timing_dirty = true;
  return GG_OK;
}

static int last_stats_full(gg_model* m, gg_stats* out) {
  if (m->eig) return gg_eigen_stats(m->eig, out);
  if (m->ld) return gg_latent_stats(m->ld, out);
  int rc = model_sync(m);
  if (rc) return rc;
  if (m->timing_dirty) { m->timing_dirty = false; if ((rc = collect_timing(m))) return rc; }
  *out = m->st;
  if (m->sbm && !m->sharded && m->K > 1) {
    // rounds of the resident scans (one grid-wide exchange each), counted by k_scan itself
    if ((rc = read_ctl(m))) return rc;
    out->resolve_rounds += (uint64_t)(uint32_t)m->h_rs[GG_RS_ROUNDS];
  }
  return GG_OK;
}

// The caller states the size of ITS gg_stats (the struct grew, and a binding built against the shorter one was
// written past its end): no more than that is written; fields beyond what the library knows stay untouched.
int gg_last_stats(gg_model* m, gg_stats* out) {
  if (!m || !out) return gg_fail(GG_ERR_INVALID, "gg_last_stats: null argument");
  const uint32_t want = out->struct_size;
  if (want < 2 * sizeof(uint32_t) || want > 4096)
    return gg_fail(GG_ERR_INVALID, "gg_last_stats: gg_stats.struct_size = %u: set it to sizeof(gg_stats) before the call", want);
  gg_stats full;
  memset(&full, 0, sizeof(full));
  int rc = last_stats_full(m, &full);
  if (rc) return rc;
  full.struct_size = want < sizeof(gg_stats) ? want : (uint32_t)sizeof(gg_stats);
  full.reserved = 0;
  memcpy(out, &full, full.struct_size);
  return GG_OK;
}

}  // extern "C"
