// gg_model.hip — host orchestration of the block-model components behind the C ABI.
//
// Mirrors, per kind (include/graphgibbs.h):
//   resample        adjacency.py:615-618, :153-181; weights.py:394-397, :154-162
//   log_likelihood  abstractions.py:111-115; weights.py:316-331
// State lives on the device between calls; the Python classes read it through
// gg_get_state.  No CPU fallback: every entry point needs a HIP device.
#include <math.h>

#include "gg_common.h"
#include "gg_block_kernels.h"
#include "gg_resolve_kernels.h"

struct gg_eigen;  // gg_eigen.hip
int gg_eigen_create(gg_eigen** out, const gg_config* cfg);
int gg_eigen_destroy(gg_eigen* e);
int gg_eigen_set_state(gg_eigen* e, int id, const void* buf, size_t nbytes);
int gg_eigen_get_state(gg_eigen* e, int id, void* buf, size_t nbytes);
int gg_eigen_resample(gg_eigen* e, const gg_data* d, int parts);
int gg_eigen_log_likelihood(gg_eigen* e, const gg_data* d, double* out);
int gg_eigen_stats(gg_eigen* e, gg_stats* out);

struct gg_model {
  gg_config cfg;
  int kind;
  bool is_adj, is_block, sbm;
  int N, K, Kp, B, NT, NSI, Np, NTAB;
  gg_eigen* eig;
  // device state
  int32_t *c, *cand, *cnt, *perm, *lab_start;
  double *m, *logm, *pi;
  double *p, *p_self;
  double *mu, *sigma, *mu_self, *sigma_self;
  double *T, *Tout, *Tin;
  int32_t* Gcnt[2];
  long long* GS[2];
  long long* tab;
  int32_t *rs, *L;
  int* status;
  double* lp_out;
  // host mirrors / hypers
  gg_niw niw, niw_self;
  int32_t* h_rs;  // pinned
  hipStream_t stream;
  uint32_t sweep;
  int Lmax, CH, max_idle;
  size_t resolve_lds;
  bool g_valid;  // G consistent with the labels for the bound data
  // timing
  bool timing;
  std::vector<hipEvent_t> ev;
  size_t ev_used;
  hipEvent_t ev_call0, ev_call1;
  gg_stats st;
};

static int model_kind_ok(int kind) {
  return kind == GG_SBM_ADJ || kind == GG_BB_ADJ || kind == GG_BERN_ADJ || kind == GG_SBM_GAUSS_W || kind == GG_NIW_W ||
         kind == GG_FIXED_W;
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

extern "C" {

int gg_model_create(gg_model** out, const gg_config* cfg) {
  if (!out || !cfg) return gg_fail(GG_ERR_INVALID, "gg_model_create: null argument");
  if (cfg->kind == GG_LOGISTIC_EIG) {
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
  m->sbm = (cfg->kind == GG_SBM_ADJ || cfg->kind == GG_SBM_GAUSS_W);
  m->is_block = true;
  m->N = cfg->N;
  m->K = m->sbm ? cfg->K : 1;
  if (m->K < 1 || m->K > 64) { delete m; return gg_fail(GG_ERR_UNSUPPORTED, "number of blocks C=%d not in 1..64", m->K); }
  m->B = m->is_adj ? 1 : cfg->B;
  if (!m->is_adj && !(m->B == 1 || m->B == 2 || m->B == 4)) {
    delete m;
    return gg_fail(GG_ERR_UNSUPPORTED, "weight dimension B=%d: the HIP engine supports B in {1,2,4}", cfg->B);
  }
  m->Kp = gg_next_pow2(m->K < 2 ? 2 : m->K);
  m->NT = m->is_adj ? 2 : gg_nt_gauss(m->B);
  m->NSI = m->is_adj ? 0 : gg_nsi(m->B);
  m->NTAB = m->is_adj ? 1 : 1 + 2 * m->NSI;
  m->Np = gg_round_up(m->N, 64);
  // NIW / non-SBM weight kinds always keep a separate self Gaussian (weights.py:91-94)
  if (m->kind == GG_NIW_W || m->kind == GG_FIXED_W) m->cfg.special_case_self_conns = 1;
  GG_HIP(hipStreamCreate(&m->stream));
  const int N = m->N, K = m->K, B = m->B, NT = m->NT;
  const size_t KK = (size_t)K * K;
  if ((rc = dmalloc(&m->c, N)) || (rc = dmalloc(&m->cand, N)) || (rc = dmalloc(&m->cnt, K)) ||
      (rc = dmalloc(&m->perm, N)) || (rc = dmalloc(&m->lab_start, K + 1)) || (rc = dmalloc(&m->m, K)) ||
      (rc = dmalloc(&m->logm, K)) || (rc = dmalloc(&m->pi, K)) || (rc = dmalloc(&m->T, KK * NT)) ||
      (rc = dmalloc(&m->Tout, (size_t)K * NT * m->Kp)) || (rc = dmalloc(&m->Tin, (size_t)K * NT * m->Kp)) ||
      (rc = dmalloc(&m->tab, KK * m->NTAB)) || (rc = dmalloc(&m->rs, 8)) || (rc = dmalloc(&m->status, 1)) ||
      (rc = dmalloc(&m->lp_out, K)))
    return rc;
  for (int d = 0; d < 2; ++d) {
    if ((rc = dmalloc(&m->Gcnt[d], (size_t)K * m->Np))) return rc;
    if (!m->is_adj && (rc = dmalloc(&m->GS[d], (size_t)K * m->Np * m->NSI))) return rc;
  }
  if (m->is_adj) {
    if ((rc = dmalloc(&m->p, KK)) || (rc = dmalloc(&m->p_self, 1))) return rc;
  } else {
    if ((rc = dmalloc(&m->mu, KK * B)) || (rc = dmalloc(&m->sigma, KK * B * B)) || (rc = dmalloc(&m->mu_self, B)) ||
        (rc = dmalloc(&m->sigma_self, (size_t)B * B)))
      return rc;
  }
  m->Lmax = N < 4096 ? N : 4096;
  if (m->Lmax < 1) m->Lmax = 1;
  if ((rc = dmalloc(&m->L, (size_t)m->Lmax * 3))) return rc;
  GG_HIP(hipHostMalloc((void**)&m->h_rs, 8 * sizeof(int32_t)));
  // resolve chunk: one Kp-lane group per node, LDS budget 96 KiB
  int CH = 1024 / m->Kp;
  if (CH > 64) CH = 64;
  while (CH > 1 && resolve_lds_bytes(m, CH) > 96 * 1024) CH >>= 1;
  m->CH = CH;
  m->resolve_lds = resolve_lds_bytes(m, CH);
  m->max_idle = 2;
  GG_HIP(hipEventCreate(&m->ev_call0));
  GG_HIP(hipEventCreate(&m->ev_call1));
  // defaults: uniform m, pi = 1, p = 0.5, standard NIW
  std::vector<double> ones(K, 1.0), um(K, 1.0 / K), half(KK, 0.5);
  GG_HIP(hipMemcpy(m->pi, ones.data(), K * 8, hipMemcpyHostToDevice));
  GG_HIP(hipMemcpy(m->m, um.data(), K * 8, hipMemcpyHostToDevice));
  if (m->is_adj) {
    GG_HIP(hipMemcpy(m->p, half.data(), KK * 8, hipMemcpyHostToDevice));
    double h = 0.5;
    GG_HIP(hipMemcpy(m->p_self, &h, 8, hipMemcpyHostToDevice));
  } else {
    memset(&m->niw, 0, sizeof(gg_niw));
    for (int i = 0; i < B; ++i) m->niw.sigma_0[i * B + i] = 1.0;
    m->niw.nu_0 = B + 2; m->niw.kappa_0 = 1.0;
    m->niw_self = m->niw;
    std::vector<double> sig(KK * B * B, 0.0), sself((size_t)B * B, 0.0);
    for (size_t ab = 0; ab < KK; ++ab)
      for (int i = 0; i < B; ++i) sig[ab * B * B + i * B + i] = 1.0;
    for (int i = 0; i < B; ++i) sself[i * B + i] = 1.0;
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
  void* ptrs[] = {m->c, m->cand, m->cnt, m->perm, m->lab_start, m->m, m->logm, m->pi, m->p, m->p_self, m->mu, m->sigma,
                  m->mu_self, m->sigma_self, m->T, m->Tout, m->Tin, m->Gcnt[0], m->Gcnt[1], m->GS[0], m->GS[1], m->tab,
                  m->rs, m->L, m->status, m->lp_out};
  for (void* p : ptrs) if (p) hipFree(p);
  if (m->h_rs) hipHostFree(m->h_rs);
  for (hipEvent_t e : m->ev) hipEventDestroy(e);
  if (m->ev_call0) hipEventDestroy(m->ev_call0);
  if (m->ev_call1) hipEventDestroy(m->ev_call1);
  if (m->stream) hipStreamDestroy(m->stream);
  delete m;
  return GG_OK;
}

static int state_ptr(gg_model* m, int id, void** p, size_t* nbytes) {
  const size_t K = m->K, KK = K * K, B = m->B;
  *p = nullptr;
  switch (id) {
    case GG_ST_C: *p = m->c; *nbytes = (size_t)m->N * 4; break;
    case GG_ST_M: *p = m->m; *nbytes = K * 8; break;
    case GG_ST_PI: *p = m->pi; *nbytes = K * 8; break;
    case GG_ST_P: if (m->is_adj) { *p = m->p; *nbytes = KK * 8; } break;
    case GG_ST_P_SELF: if (m->is_adj) { *p = m->p_self; *nbytes = 8; } break;
    case GG_ST_MU: if (!m->is_adj) { *p = m->mu; *nbytes = KK * B * 8; } break;
    case GG_ST_SIGMA: if (!m->is_adj) { *p = m->sigma; *nbytes = KK * B * B * 8; } break;
    case GG_ST_MU_SELF: if (!m->is_adj) { *p = m->mu_self; *nbytes = B * 8; } break;
    case GG_ST_SIGMA_SELF: if (!m->is_adj) { *p = m->sigma_self; *nbytes = B * B * 8; } break;
    default: break;
  }
  return *p ? GG_OK : GG_ERR_INVALID;
}

int gg_set_state(gg_model* m, int32_t id, const void* buf, size_t nbytes) {
  if (!m || !buf) return gg_fail(GG_ERR_INVALID, "gg_set_state: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig) return gg_eigen_set_state(m->eig, id, buf, nbytes);
  const size_t B = m->B;
  if (id == GG_ST_SWEEP) {
    if (nbytes != 4) return gg_fail(GG_ERR_INVALID, "GG_ST_SWEEP expects 4 bytes");
    m->sweep = *(const uint32_t*)buf;
    return GG_OK;
  }
  if (id == GG_ST_NIW || id == GG_ST_NIW_SELF) {
    if (m->is_adj) return gg_fail(GG_ERR_INVALID, "NIW hypers on an adjacency model");
    size_t need = (B + B * B + 2) * 8;
    if (nbytes != need) return gg_fail(GG_ERR_INVALID, "NIW hypers: expected %zu bytes, got %zu", need, nbytes);
    const double* d = (const double*)buf;
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
    m->g_valid = false;
  }
  GG_HIP(hipMemcpyAsync(p, buf, nbytes, hipMemcpyHostToDevice, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  return GG_OK;
}

int gg_get_state(gg_model* m, int32_t id, void* buf, size_t nbytes) {
  if (!m || !buf) return gg_fail(GG_ERR_INVALID, "gg_get_state: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig) return gg_eigen_get_state(m->eig, id, buf, nbytes);
  const size_t B = m->B;
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
  if (d->device != m->cfg.device) return gg_fail(GG_ERR_INVALID, "data on device %d, model on device %d", d->device, m->cfg.device);
  if (!m->is_adj) {
    if (!d->has_w) return gg_fail(GG_ERR_INVALID, "weight model needs data created with W");
    if (d->B != m->B) return gg_fail(GG_ERR_INVALID, "data has B=%d, model has B=%d", d->B, m->B);
  }
  return GG_OK;
}

static hipEvent_t next_event(gg_model* m) {
  if (m->ev_used == m->ev.size()) {
    hipEvent_t e;
    hipEventCreate(&e);
    m->ev.push_back(e);
  }
  return m->ev[m->ev_used++];
}

static gg_node_ctx node_ctx(const gg_model* m, const gg_data* d) {
  gg_node_ctx x;
  x.N = m->N; x.Np = m->Np; x.K = m->K; x.Kp = m->Kp; x.B = m->B; x.shift = m->is_adj ? 0 : d->shift;
  x.special = m->cfg.special_case_self_conns;
  x.c = m->c; x.cnt = m->cnt; x.logm = m->logm; x.T = m->T; x.Tout = m->Tout; x.Tin = m->Tin;
  x.Gcnt[0] = m->Gcnt[0]; x.Gcnt[1] = m->Gcnt[1]; x.GS[0] = m->GS[0]; x.GS[1] = m->GS[1];
  x.dA = d->dA; x.dWq = d->dWq;
  return x;
}

// rows per workgroup of the bulk kernel: enough workgroups to fill 256 CUs,
// at most 1020 rows (SWAR byte counters: 255 rows per wave).
static int bulk_rw(const gg_model* m, int strips) {
  long long target_wg = 2048;
  long long rows_slots = target_wg / (strips > 0 ? strips : 1);
  if (rows_slots < 1) rows_slots = 1;
  long long rw = (m->N + rows_slots - 1) / rows_slots;
  rw = (rw + 3) / 4 * 4;
  if (rw < 16) rw = 16;
  if (rw > 1020) rw = 1020;
  return (int)rw;
}

// sort + BULK statistics under the current labels.  dirs: bit 0 = out (reads X^T), bit 1 = in (reads X).
static int run_bulk(gg_model* m, const gg_data* d, int dirs) {
  const int N = m->N, K = m->K, Np = m->Np;
  k_sort_labels<<<1, 1024, 2 * K * sizeof(int), m->stream>>>(m->c, N, K, m->perm, m->lab_start, m->cnt);
  for (int dir = 0; dir < 2; ++dir) {
    if (!(dirs & (1 << dir))) continue;
    GG_HIP(hipMemsetAsync(m->Gcnt[dir], 0, (size_t)K * Np * sizeof(int32_t), m->stream));
    if (!m->is_adj) GG_HIP(hipMemsetAsync(m->GS[dir], 0, (size_t)K * Np * m->NSI * sizeof(long long), m->stream));
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (m->timing) { e0 = next_event(m); e1 = next_event(m); GG_HIP(hipEventRecord(e0, m->stream)); }
  for (int dir = 0; dir < 2; ++dir) {
    if (!(dirs & (1 << dir))) continue;
    if (m->is_adj) {
      int strips = (Np + 1023) / 1024;
      int RW = bulk_rw(m, strips);
      dim3 grd(strips, (N + RW - 1) / RW + K);
      const uint8_t* X = dir == 0 ? d->AT : d->A;
      k_bulk_adj<<<grd, 256, 0, m->stream>>>(X, Np, m->perm, m->lab_start, K, RW, m->Gcnt[dir]);
      m->st.bulk_bytes += (double)N * (double)N * 1.0;
    } else {
      const int32_t* X = dir == 0 ? d->WqT : d->Wq;
      int SC = 64 * (4 / m->B);
      int strips = (Np + SC - 1) / SC;
      int RW = bulk_rw(m, strips);
      dim3 grd(strips, (N + RW - 1) / RW + K);
      switch (m->B) {
        case 1: k_bulk_w<1><<<grd, 256, 0, m->stream>>>(X, Np, m->perm, m->lab_start, K, RW, m->Gcnt[dir], m->GS[dir]); break;
        case 2: k_bulk_w<2><<<grd, 256, 0, m->stream>>>(X, Np, m->perm, m->lab_start, K, RW, m->Gcnt[dir], m->GS[dir]); break;
        case 4: k_bulk_w<4><<<grd, 256, 0, m->stream>>>(X, Np, m->perm, m->lab_start, K, RW, m->Gcnt[dir], m->GS[dir]); break;
      }
      m->st.bulk_bytes += (double)N * (double)N * 4.0 * m->B;
    }
    m->st.bulk_launches += 1;
  }
  if (m->timing) GG_HIP(hipEventRecord(e1, m->stream));
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int run_pair_tables(gg_model* m, const gg_data* d) {
  int add_diag = m->cfg.special_case_self_conns ? 0 : 1;
  k_pair_tables<<<m->K * m->K, 256, 0, m->stream>>>(m->Gcnt[0], m->is_adj ? nullptr : m->GS[0], m->Np, m->K, m->NSI,
                                                    m->perm, m->lab_start, add_diag, d->dA, d->dWq, m->B, m->tab);
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int run_draw_params(gg_model* m, const gg_data* d) {
  const int K = m->K;
  int nthr = K * K + 1;
  if (m->is_adj) {
    gg_adj_hyp h;
    h.tau1 = m->cfg.tau1; h.tau0 = m->cfg.tau0;
    h.special = m->cfg.special_case_self_conns;
    h.bb = (m->kind == GG_BB_ADJ);
    h.d1 = d->d1; h.d0 = (long long)m->N - d->d1;
    k_draw_p<<<(nthr + 63) / 64, 64, 0, m->stream>>>(m->tab, m->cnt, K, h, m->cfg.seed, m->sweep, m->p, m->p_self);
  } else {
    gg_w_hyp h;
    h.niw = m->niw; h.niw_self = m->niw_self;
    h.special = m->cfg.special_case_self_conns;
    h.shift = d->shift; h.d1 = d->d1;
    for (int i = 0; i < gg_nsi(GG_MAXB); ++i) h.dS[i] = d->dS[i];
    k_draw_niw<<<(nthr + 63) / 64, 64, 0, m->stream>>>(m->tab, K, m->B, h, m->cfg.seed, m->sweep, m->mu, m->sigma,
                                                       m->mu_self, m->sigma_self);
  }
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int run_prep_tables(gg_model* m) {
  int nthr = m->K * m->K + m->K;
  k_prep_tables<<<(nthr + 63) / 64, 64, 0, m->stream>>>(m->is_adj ? 1 : 0, m->K, m->Kp, m->B, m->m, m->p, m->mu, m->sigma,
                                                        m->logm, m->T, m->Tout, m->Tin, m->status);
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int launch_speculate(gg_model* m, const gg_data* d, int n_begin, int lp_node) {
  gg_node_ctx x = node_ctx(m, d);
  int per = 256 / m->Kp;
  int nb = (m->N - n_begin + per - 1) / per;
  if (nb < 1) nb = 1;
  uint32_t site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C;
  double* lpo = lp_node >= 0 ? m->lp_out : nullptr;
#define SPEC(NTV, ADJV) k_speculate<NTV, ADJV><<<nb, 256, 0, m->stream>>>(x, n_begin, m->cfg.seed, site, m->sweep, m->cand, lp_node, lpo)
  if (m->is_adj) SPEC(2, true);
  else if (m->B == 1) SPEC(3, false);
  else if (m->B == 2) SPEC(6, false);
  else SPEC(15, false);
#undef SPEC
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int launch_resolve(gg_model* m, const gg_data* d, int fresh) {
  gg_resolve_args g;
  g.x = node_ctx(m, d);
  g.c = m->c; g.cnt = m->cnt; g.cand = m->cand; g.rs = m->rs; g.L = m->L;
  g.Lmax = m->Lmax; g.max_idle = m->max_idle; g.CH = m->CH; g.fresh = fresh;
  g.XA = d->A; g.XAT = d->AT; g.XW = d->Wq; g.XWT = d->WqT;
  g.seed = m->cfg.seed; g.site = m->is_adj ? GG_SITE_ADJ_C : GG_SITE_W_C; g.sweep = m->sweep;
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
  else if (m->B == 1) RES(3, false);
  else if (m->B == 2) RES(6, false);
  else RES(15, false);
#undef RES
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int launch_apply_L(gg_model* m, const gg_data* d) {
  gg_node_ctx x = node_ctx(m, d);
  int nb = (m->N + 255) / 256;
#define APL(NTV, ADJV)                                                                                             \
  k_apply_L<NTV, ADJV><<<nb, 256, 0, m->stream>>>(x, m->L, m->rs, m->Gcnt[0], m->Gcnt[1], m->GS[0], m->GS[1], d->A, d->AT, \
                                                  d->Wq, d->WqT)
  if (m->is_adj) APL(2, true);
  else if (m->B == 1) APL(3, false);
  else if (m->B == 2) APL(6, false);
  else APL(15, false);
#undef APL
  GG_HIP(hipGetLastError());
  return GG_OK;
}

// resample_c: speculate, then resolve rounds until the scan is complete.
static int run_resample_c(gg_model* m, const gg_data* d) {
  if (m->K == 1) return GG_OK;  // adjacency.py:657-658, weights.py:421-422
  int rc;
  int32_t init[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  GG_HIP(hipMemcpyAsync(m->rs, init, sizeof(init), hipMemcpyHostToDevice, m->stream));
  int pos = 0;
  for (int round = 0; pos < m->N; ++round) {
    if (round > 0) {
      if ((rc = launch_apply_L(m, d))) return rc;
    }
    if ((rc = launch_speculate(m, d, pos, -1))) return rc;
    if ((rc = launch_resolve(m, d, 1))) return rc;
    GG_HIP(hipMemcpyAsync(m->h_rs, m->rs, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    GG_HIP(hipStreamSynchronize(m->stream));
    pos = m->h_rs[0];
    m->st.resolve_rounds += 1;
    if (round > 2 * m->N + 8) return gg_fail(GG_ERR_HIP, "resolve did not make progress (pos=%d)", pos);
  }
  m->st.label_changes += (uint64_t)m->h_rs[2];
  if (m->h_rs[2] > 0) m->g_valid = false;
  return GG_OK;
}

static int run_draw_m(gg_model* m) {
  uint32_t site = m->is_adj ? GG_SITE_ADJ_M : GG_SITE_W_M;
  k_draw_m<<<1, 64, 64 * sizeof(double), m->stream>>>(m->cnt, m->pi, m->K, m->cfg.seed, site, m->sweep, m->m);
  GG_HIP(hipGetLastError());
  return GG_OK;
}

static int check_status(gg_model* m) {
  int st = 0;
  GG_HIP(hipMemcpyAsync(&st, m->status, sizeof(int), hipMemcpyDeviceToHost, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  if (st) {
    int z = 0;
    hipMemcpy(m->status, &z, sizeof(int), hipMemcpyHostToDevice);
    return gg_fail(GG_ERR_NUMERIC, "a block covariance is not positive definite");
  }
  return GG_OK;
}

static int resample_parts(gg_model* m, const gg_data* d, int parts) {
  int rc;
  const bool learn_params = (m->kind != GG_BERN_ADJ && m->kind != GG_FIXED_W);
  const bool do_params = (parts & GG_PART_PARAMS) && learn_params;
  const bool do_c = (parts & GG_PART_C) && m->sbm && m->K > 1;
  const bool do_m = (parts & GG_PART_M) && m->sbm;
  if (do_params || do_c) {
    // out-statistics feed the pair tables; the scan needs both directions
    if ((rc = run_bulk(m, d, do_c ? 3 : 1))) return rc;
  }
  if (do_params) {
    if ((rc = run_pair_tables(m, d))) return rc;
    if ((rc = run_draw_params(m, d))) return rc;
  }
  if (do_c) {
    if ((rc = run_prep_tables(m))) return rc;
    if ((rc = run_resample_c(m, d))) return rc;
  }
  if (do_m) {
    if (!do_c && !do_params) {
      // block sizes under the current labels
      k_sort_labels<<<1, 1024, 2 * m->K * sizeof(int), m->stream>>>(m->c, m->N, m->K, m->perm, m->lab_start, m->cnt);
    }
    if ((rc = run_draw_m(m))) return rc;
  }
  return GG_OK;
}

static void begin_call(gg_model* m) {
  m->ev_used = 0;
  m->st.bulk_bytes = 0; m->st.bulk_ms = 0; m->st.bulk_launches = 0; m->st.total_ms = 0;
  hipEventRecord(m->ev_call0, m->stream);
}

static int end_call(gg_model* m) {
  GG_HIP(hipEventRecord(m->ev_call1, m->stream));
  GG_HIP(hipStreamSynchronize(m->stream));
  float ms = 0;
  GG_HIP(hipEventElapsedTime(&ms, m->ev_call0, m->ev_call1));
  m->st.total_ms = ms;
  for (size_t i = 0; i + 1 < m->ev_used; i += 2) {
    float t = 0;
    GG_HIP(hipEventElapsedTime(&t, m->ev[i], m->ev[i + 1]));
    m->st.bulk_ms += t;
  }
  return GG_OK;
}

extern "C" {

int gg_set_timing(gg_model* m, int32_t on) {
  if (!m) return gg_fail(GG_ERR_INVALID, "null model");
  m->timing = on != 0;
  return GG_OK;
}

int gg_resample_part(gg_model* m, const gg_data* d, int32_t parts, double beta) {
  if (!m) return gg_fail(GG_ERR_INVALID, "null model");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (beta != 1.0) return gg_fail(GG_ERR_UNSUPPORTED, "tempered conditionals (beta != 1) are not implemented yet");
  if (m->eig) return gg_eigen_resample(m->eig, d, parts);
  int rc = GG_OK;
  if (parts != GG_PART_M) rc = check_data(m, d);  // resample_m needs the labels only (adjacency.py:697-702)
  if (rc) return rc;
  begin_call(m);
  rc = resample_parts(m, d, parts);
  if (rc) return rc;
  if ((rc = end_call(m))) return rc;
  if (!m->is_adj && (parts & GG_PART_C) && (rc = check_status(m))) return rc;
  return GG_OK;
}

int gg_resample(gg_model* m, const gg_data* d, int32_t n_sweeps, double beta) {
  if (!m) return gg_fail(GG_ERR_INVALID, "null model");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (beta != 1.0) return gg_fail(GG_ERR_UNSUPPORTED, "tempered conditionals (beta != 1) are not implemented yet");
  if (m->eig) {
    for (int s = 0; s < n_sweeps; ++s) {
      int rc = gg_eigen_resample(m->eig, d, GG_PART_ALL);
      if (rc) return rc;
    }
    return GG_OK;
  }
  int rc = check_data(m, d);
  if (rc) return rc;
  begin_call(m);
  for (int s = 0; s < n_sweeps; ++s) {
    if ((rc = resample_parts(m, d, GG_PART_ALL))) return rc;
    m->sweep += 1;
    m->st.sweeps += 1;
  }
  if ((rc = end_call(m))) return rc;
  if (!m->is_adj && (rc = check_status(m))) return rc;
  return GG_OK;
}

int gg_node_lp(gg_model* m, const gg_data* d, int32_t n, double* lp) {
  if (!m || !lp) return gg_fail(GG_ERR_INVALID, "gg_node_lp: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig) return gg_fail(GG_ERR_UNSUPPORTED, "gg_node_lp: block models only");
  int rc = check_data(m, d);
  if (rc) return rc;
  if (n < 0 || n >= m->N) return gg_fail(GG_ERR_INVALID, "node %d out of range", n);
  begin_call(m);
  if ((rc = run_bulk(m, d, 3))) return rc;
  if ((rc = run_prep_tables(m))) return rc;
  if ((rc = launch_speculate(m, d, 0, n))) return rc;
  if ((rc = end_call(m))) return rc;
  if (!m->is_adj && (rc = check_status(m))) return rc;
  GG_HIP(hipMemcpy(lp, m->lp_out, (size_t)m->K * 8, hipMemcpyDeviceToHost));
  return GG_OK;
}

// Closed forms from the K x K tables:
//   adjacency (abstractions.py:111-115): sum_ab n1 log p + n0 log(1 - clip(p)) + diagonal terms;
//   weights (weights.py:316-331): sum_ab [ n g + h.s1 + Q.s2 ] + self Gaussian on the diagonal edges.
int gg_log_likelihood(gg_model* m, const gg_data* d, double* out) {
  if (!m || !out) return gg_fail(GG_ERR_INVALID, "gg_log_likelihood: null argument");
  GG_HIP(hipSetDevice(m->cfg.device));
  if (m->eig) return gg_eigen_log_likelihood(m->eig, d, out);
  int rc = check_data(m, d);
  if (rc) return rc;
  begin_call(m);
  if ((rc = run_bulk(m, d, 1))) return rc;
  if ((rc = run_pair_tables(m, d))) return rc;
  const int K = m->K, B = m->B;
  const size_t KK = (size_t)K * K;
  std::vector<long long> tab(KK * m->NTAB);
  std::vector<int32_t> cnt(K);
  GG_HIP(hipMemcpyAsync(tab.data(), m->tab, tab.size() * 8, hipMemcpyDeviceToHost, m->stream));
  GG_HIP(hipMemcpyAsync(cnt.data(), m->cnt, K * 4, hipMemcpyDeviceToHost, m->stream));
  if ((rc = end_call(m))) return rc;
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
        // with the diagonal forced to zero probability the (a,a) table must not hold it
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
    const int NT = m->NT, NSI = m->NSI;
    std::vector<double> mu(KK * B), sigma(KK * B * B), mus(B), sigs((size_t)B * B);
    GG_HIP(hipMemcpy(mu.data(), m->mu, mu.size() * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(sigma.data(), m->sigma, sigma.size() * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(mus.data(), m->mu_self, mus.size() * 8, hipMemcpyDeviceToHost));
    GG_HIP(hipMemcpy(sigs.data(), m->sigma_self, sigs.size() * 8, hipMemcpyDeviceToHost));
    const double sc1 = ldexp(1.0, -d->shift), sc2 = ldexp(1.0, -2 * d->shift);
    auto ll_of = [&](const double* mu_, const double* sig_, double n, const long long* hl) -> double {
      if (n == 0) return 0.0;
      double e[GG_MAXNT];
      if (gg_gauss_table(mu_, sig_, B, e)) return NAN;
      double v = n * e[0];
      for (int s = 0; s < NSI; ++s) {
        double x = gg_hilo_to_double(hl[2 * s], hl[2 * s + 1]) * (s < B ? sc1 : sc2);
        v += e[1 + s] * x;
      }
      return v;
    };
    (void)NT;
    for (size_t ab = 0; ab < KK; ++ab) {
      const long long* e = &tab[ab * m->NTAB];
      ll += ll_of(&mu[ab * B], &sigma[ab * B * B], (double)e[0], e + 1);
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

int gg_last_stats(gg_model* m, gg_stats* out) {
  if (!m || !out) return gg_fail(GG_ERR_INVALID, "gg_last_stats: null argument");
  if (m->eig) return gg_eigen_stats(m->eig, out);
  *out = m->st;
  return GG_OK;
}

}  // extern "C"
