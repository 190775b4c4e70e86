"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY — see the header of oracle_sbm.cpp.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package graphistician_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

PART_PARAMS, PART_C, PART_M = 1, 2, 4
PART_ALL = 7


def build(force=False):
    """Compile the oracle with g++ (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_sbm_adj_loglik.restype = C.c_double
        _lib.orc_const_adj_loglik.restype = C.c_double
        _lib.orc_sbm_w_loglik.restype = C.c_double
        _lib.orc_uniform_at.restype = C.c_double
        _lib.orc_normal_at.restype = C.c_double
        for name in ("orc_eig_loglik",):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = C.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, _p(a, C.c_uint8)


def _i32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return _p(a, C.c_int32)


def _f64(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous, (a.dtype, a.flags)
    return _p(a, C.c_double)


def pack_niw(mu_0, sigma_0, nu_0, kappa_0):
    mu_0 = np.asarray(mu_0, dtype=np.float64).ravel()
    sigma_0 = np.asarray(sigma_0, dtype=np.float64).ravel()
    return np.ascontiguousarray(np.concatenate([mu_0, sigma_0, [float(nu_0), float(kappa_0)]]))


def quantise(A, W):
    """Fixed-point weight storage: returns (q int32 (N,N,B), shift)."""
    N = A.shape[0]
    W = np.ascontiguousarray(W, dtype=np.float64)
    B = W.shape[2]
    A8, pA = _u8(A != 0)
    q = np.empty((N, N, B), dtype=np.int32)
    shift = C.c_int(0)
    rc = lib().orc_quantise(N, B, pA, _f64(W), _i32(q), C.byref(shift))
    if rc:
        raise ValueError("non-finite weight on an edge")
    return q, shift.value


def quantise_wide(A, W):
    """Wide weight storage (spec/gg_lp.h): W (N, N[, 1]) as two 24-bit limbs, w = (q0 + q1 2^-24) 2^-shift -> (q int32 (N,N,2), shift)."""
    N = A.shape[0]
    W = np.ascontiguousarray(np.asarray(W, dtype=np.float64).reshape(N, N))
    A8, pA = _u8(A != 0)
    q = np.empty((N, N, 2), dtype=np.int32)
    shift = C.c_int(0)
    rc = lib().orc_quantise_wide(N, pA, _f64(W), _i32(q), C.byref(shift))
    if rc:
        raise ValueError("non-finite weight on an edge")
    return q, shift.value


class _wide_mode(object):
    """with _wide_mode(flag): the Gaussian-weight oracle functions treat q as the two limbs of a one-dimensional weight."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        if self.on:
            lib().orc_set_wide(1)

    def __exit__(self, *exc):
        if self.on:
            lib().orc_set_wide(0)


def dequantise(q, shift):
    return np.ldexp(q.astype(np.float64), -shift)


# ---------------------------------------------------------------- SBM adjacency
def sbm_adj_sweep(A, c, m, p, p_self, pi, tau1, tau0, special, seed, sweep, parts=PART_ALL, beta=1.0):
    """In-place sweep of (c, m, p); returns new p_self.  beta = AIS inverse temperature."""
    N, K = A.shape[0], m.shape[0]
    A8, pA = _u8(A != 0)
    ps = C.c_double(float(p_self))
    lib().orc_sbm_adj_sweep_beta(N, K, pA, _i32(c), _f64(m), _f64(p), C.byref(ps),
                                 _f64(np.ascontiguousarray(pi, dtype=np.float64)), C.c_double(tau1), C.c_double(tau0),
                                 int(special), C.c_uint64(seed), C.c_uint32(sweep), int(parts), C.c_double(beta))
    return ps.value


def sbm_adj_node_lp(A, c, m, p, special, n1, literal=False):
    N, K = A.shape[0], m.shape[0]
    A8, pA = _u8(A != 0)
    lp = np.zeros(K)
    f = lib().orc_sbm_adj_node_lp_literal if literal else lib().orc_sbm_adj_node_lp
    f(N, K, pA, _i32(c), _f64(m), _f64(p), int(special), int(n1), _f64(lp))
    return lp


def categorical(lp, u):
    lp = np.ascontiguousarray(lp, dtype=np.float64)
    return lib().orc_categorical(_f64(lp), lp.shape[0], C.c_double(u))


def sbm_adj_loglik(A, c, p, p_self, allow_self=True, special=True):
    N, K = A.shape[0], p.shape[0]
    A8, pA = _u8(A != 0)
    return lib().orc_sbm_adj_loglik(N, K, pA, _i32(c), _f64(p), C.c_double(p_self), int(allow_self), int(special))


def bb_adj_sweep(A, tau1, tau0, self_connection, seed, sweep, beta=1.0):
    N = A.shape[0]
    A8, pA = _u8(A != 0)
    p, ps = C.c_double(0), C.c_double(0)
    lib().orc_bb_adj_sweep_beta(N, pA, C.byref(p), C.byref(ps), C.c_double(tau1), C.c_double(tau0), int(self_connection),
                                C.c_uint64(seed), C.c_uint32(sweep), C.c_double(beta))
    return p.value, ps.value


def const_adj_loglik(A, p, p_self):
    N = A.shape[0]
    A8, pA = _u8(A != 0)
    return lib().orc_const_adj_loglik(N, pA, C.c_double(p), C.c_double(p_self))


# ---------------------------------------------------------------- Gaussian weights
def sbm_w_sweep(A, q, shift, c, m, mu, sigma, mu_self, sigma_self, pi, niw, niw_self, special, seed, sweep,
                parts=PART_ALL, beta=1.0, wide=False):
    """wide: q holds the two limbs of a one-dimensional weight (quantise_wide); mu (K,K,1), sigma (K,K,1,1), one-dimensional priors."""
    N, K, B = A.shape[0], m.shape[0], q.shape[2]
    A8, pA = _u8(A != 0)
    with _wide_mode(wide):
        rc = lib().orc_sbm_w_sweep_beta(N, K, B, pA, _i32(q), int(shift), _i32(c), _f64(m), _f64(mu), _f64(sigma),
                                        _f64(mu_self), _f64(sigma_self), _f64(np.ascontiguousarray(pi, dtype=np.float64)),
                                        _f64(niw), _f64(niw_self), int(special), C.c_uint64(seed), C.c_uint32(sweep), int(parts),
                                        C.c_double(beta))
    if rc:
        raise RuntimeError("oracle sbm_w_sweep failed rc=%d" % rc)


def sbm_wsh_sweep(A, q, shift, c, m, mu, sigma, mu_self, sigma_self, pi, prior, niw_self, seed, sweep, parts=PART_ALL, beta=1.0,
                  special=True):
    """SBMGaussianWeightSharedCov (weights.py:501-552); sigma (K,K,B,B) holds the shared covariance in every pair;
    prior = pack_niw(mu_0, lmbda_0, nu_0, unused)."""
    N, K, B = A.shape[0], m.shape[0], q.shape[2]
    A8, pA = _u8(A != 0)
    rc = lib().orc_sbm_wsh_sweep(N, K, B, pA, _i32(q), int(shift), _i32(c), _f64(m), _f64(mu), _f64(sigma), _f64(mu_self),
                                 _f64(sigma_self), _f64(np.ascontiguousarray(pi, dtype=np.float64)), _f64(prior), _f64(niw_self),
                                 int(bool(special)), C.c_uint64(seed), C.c_uint32(sweep), int(parts), C.c_double(beta))
    if rc:
        raise RuntimeError("oracle sbm_wsh_sweep failed rc=%d" % rc)


def sbm_w_node_lp(A, q, shift, c, m, mu, sigma, special, n1, literal=False, wide=False):
    N, K, B = A.shape[0], m.shape[0], q.shape[2]
    A8, pA = _u8(A != 0)
    lp = np.zeros(K)
    assert not (wide and literal)
    f = lib().orc_sbm_w_node_lp_literal if literal else lib().orc_sbm_w_node_lp
    with _wide_mode(wide):
        rc = f(N, K, B, pA, _i32(q), int(shift), _i32(c), _f64(m), _f64(mu), _f64(sigma), int(special), int(n1), _f64(lp))
    if rc:
        raise RuntimeError("oracle sbm_w_node_lp failed rc=%d" % rc)
    return lp


def sbm_w_loglik(A, q, shift, c, mu, sigma, mu_self, sigma_self, special=True, wide=False):
    N, B = A.shape[0], q.shape[2]
    K = mu.shape[0]
    A8, pA = _u8(A != 0)
    with _wide_mode(wide):
        return lib().orc_sbm_w_loglik(N, K, B, pA, _i32(q), int(shift), _i32(c), _f64(mu), _f64(sigma), _f64(mu_self),
                                      _f64(sigma_self), int(special))


# ---------------------------------------------------------------- joint SBM (stochastic_block_model.py:202-309)
def sbm_joint_sweep(A, q, shift, c, m, p, mu, sigma, pi, niw, tau1, tau0, seed, sweep, parts=PART_ALL, beta=1.0):
    """In-place sweep of (c, m, p, mu, sigma) of the joint model (one label vector for adjacency and weights)."""
    N, K, B = A.shape[0], m.shape[0], q.shape[2]
    A8, pA = _u8(A != 0)
    rc = lib().orc_sbm_joint_sweep(N, K, B, pA, _i32(q), int(shift), _i32(c), _f64(m), _f64(p), _f64(mu), _f64(sigma),
                                   _f64(np.ascontiguousarray(pi, dtype=np.float64)), _f64(niw), C.c_double(tau1), C.c_double(tau0),
                                   C.c_uint64(seed), C.c_uint32(sweep), int(parts), C.c_double(beta))
    if rc:
        raise RuntimeError("oracle sbm_joint_sweep failed rc=%d" % rc)


def sbm_joint_node_lp(A, q, shift, c, m, p, mu, sigma, n1, literal=False):
    N, K, B = A.shape[0], m.shape[0], q.shape[2]
    A8, pA = _u8(A != 0)
    lp = np.zeros(K)
    f = lib().orc_sbm_joint_node_lp_literal if literal else lib().orc_sbm_joint_node_lp
    rc = f(N, K, B, pA, _i32(q), int(shift), _i32(c), _f64(m), _f64(p), _f64(mu), _f64(sigma), int(n1), _f64(lp))
    if rc:
        raise RuntimeError("oracle sbm_joint_node_lp failed rc=%d" % rc)
    return lp


def sbm_joint_loglik(A, q, shift, c, p, mu, sigma):
    N, B, K = A.shape[0], q.shape[2], p.shape[0]
    A8, pA = _u8(A != 0)
    lib().orc_sbm_joint_loglik.restype = C.c_double
    return lib().orc_sbm_joint_loglik(N, K, B, pA, _i32(q), int(shift), _i32(c), _f64(p), _f64(mu), _f64(sigma))


def niw_w_sweep(A, q, shift, mu, sigma, mu_self, sigma_self, niw, niw_self, seed, sweep, beta=1.0):
    N, B = A.shape[0], q.shape[2]
    A8, pA = _u8(A != 0)
    rc = lib().orc_niw_w_sweep_beta(N, B, pA, _i32(q), int(shift), _f64(mu), _f64(sigma), _f64(mu_self), _f64(sigma_self),
                                    _f64(niw), _f64(niw_self), C.c_uint64(seed), C.c_uint32(sweep), C.c_double(beta))
    if rc:
        raise RuntimeError("oracle niw_w_sweep failed rc=%d" % rc)


# ---------------------------------------------------------------- spec primitives
def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(*[C.c_uint32(x) for x in ctr], *[C.c_uint32(x) for x in key], out)
    return [int(x) for x in out]


def uniform_at(seed, site, sweep, index):
    return lib().orc_uniform_at(C.c_uint64(seed), C.c_uint32(site), C.c_uint32(sweep), C.c_uint32(index))


def uniform_at64(seed, site, sweep, index, stream=False):
    """The uniform addressed by a 64-bit index (spec/gg_philox.h: gg_uniform_at64; stream=True: the first uniform of
    gg_stream_init64's stream)."""
    f = lib().orc_stream_first64 if stream else lib().orc_uniform_at64
    f.restype = C.c_double
    return f(C.c_uint64(seed), C.c_uint32(site), C.c_uint32(sweep), C.c_uint64(index))


def normal_at(seed, site, sweep, index):
    return lib().orc_normal_at(C.c_uint64(seed), C.c_uint32(site), C.c_uint32(sweep), C.c_uint32(index))


def exp_vec(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().orc_exp_vec(_f64(x), _f64(y), C.c_longlong(x.size))
    return y


def log_vec(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().orc_log_vec(_f64(x), _f64(y), C.c_longlong(x.size))
    return y


def normal_vec(seed, n):
    y = np.empty(n)
    lib().orc_normal_vec(C.c_uint64(seed), _f64(y), C.c_longlong(n))
    return y


def gamma_vec(seed, a, n):
    y = np.empty(n)
    lib().orc_gamma_vec(C.c_uint64(seed), C.c_double(a), _f64(y), C.c_longlong(n))
    return y


def beta_vec(seed, a, b, n):
    y = np.empty(n)
    lib().orc_beta_vec(C.c_uint64(seed), C.c_double(a), C.c_double(b), _f64(y), C.c_longlong(n))
    return y


def dirichlet_vec(seed, alpha, n):
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    y = np.empty((n, alpha.shape[0]))
    lib().orc_dirichlet_vec(C.c_uint64(seed), _f64(alpha), alpha.shape[0], _f64(y), C.c_longlong(n))
    return y


def niw_vec(seed, niw, B, n_obs, s1, s2, n):
    s1 = np.ascontiguousarray(s1, dtype=np.float64)
    s2 = np.ascontiguousarray(s2, dtype=np.float64)
    mu = np.empty((n, B))
    sig = np.empty((n, B, B))
    lib().orc_niw_vec(C.c_uint64(seed), _f64(niw), B, C.c_double(n_obs), _f64(s1), _f64(s2), _f64(mu), _f64(sig),
                      C.c_longlong(n))
    return mu, sig


def gauss_table(mu, sigma):
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    sigma = np.ascontiguousarray(sigma, dtype=np.float64)
    B = mu.shape[0]
    out = np.zeros(1 + B + B * (B + 1) // 2)
    rc = lib().orc_gauss_table(_f64(mu), _f64(sigma), B, _f64(out))
    if rc:
        raise ValueError("sigma not SPD")
    return out


# ---------------------------------------------------------------- logistic eigenmodel
PART_OMEGA, PART_F, PART_MU0, PART_LMBDA = 8, 16, 32, 64
PART_EIG_ALL = 120


def eig_sweep(A, F, lmbda, mu_0, mu_lmbda, mu_mu_0, sigma_mu0, sigma_F, sigma_lmbda, lmbda_given, seed, sweep,
              parts=PART_EIG_ALL, Omega=None, beta=1.0, f_draw="perturbation"):
    """In-place sweep of (F, lmbda); returns (mu_0, Omega).  beta < 1: tempered likelihood (build-defined, PG(beta, psi)).
    f_draw="reference": F[n] is drawn the reference's way - Sigma_post = inv(prec), Sigma_post h + chol(Sigma_post) z
    (eigenmodel.py:903-908) - instead of by perturbation (spec/gg_eig.h); the same conditional law, another stream."""
    N, D = F.shape
    A8, pA = _u8(A != 0)
    if Omega is None:
        Omega = np.zeros((N, N))
    if f_draw == "reference" and (parts & PART_F):
        args = (mu_lmbda, mu_mu_0, sigma_mu0, sigma_F, sigma_lmbda, lmbda_given, seed, sweep)
        if parts & PART_OMEGA:
            eig_sweep(A, F, lmbda, mu_0, *args, parts=PART_OMEGA, Omega=Omega, beta=beta)
        eig_F_window(A, Omega, F, lmbda, mu_0, sigma_F, seed, sweep, 0, N, beta=beta, ref_draw=True)
        rest = parts & (PART_MU0 | PART_LMBDA)
        if rest:
            mu_0, _ = eig_sweep(A, F, lmbda, mu_0, *args, parts=rest, Omega=Omega, beta=beta)
        return mu_0, Omega
    assert f_draw in ("perturbation", "reference")
    mu = C.c_double(float(mu_0))
    rc = lib().orc_eig_sweep(N, D, pA, _f64(F), _f64(lmbda), C.byref(mu), _f64(Omega),
                             _f64(np.ascontiguousarray(mu_lmbda, dtype=np.float64)), C.c_double(mu_mu_0),
                             C.c_double(sigma_mu0), C.c_double(sigma_F), C.c_double(sigma_lmbda), int(lmbda_given),
                             C.c_uint64(seed), C.c_uint32(sweep), int(parts), C.c_double(beta))
    if rc:
        raise RuntimeError("oracle eig_sweep failed rc=%d" % rc)
    return mu.value, Omega


def eig_F_window(A, Omega, F, lmbda, mu_0, sigma_F, seed, sweep, n_begin, n_end, beta=1.0, ref_draw=False):
    """The nodes [n_begin, n_end) of _resample_F's sequential scan (eigenmodel.py:876-908), F updated in place: F must hold the
    new features of the nodes before n_begin and the old ones from there on.  Omega: dense (N, N) auxiliaries; only the sums
    Omega[n, nn] + Omega[nn, n] of the window's rows and columns are read.  ref_draw: the reference's form of the draw
    (inverse, mean, Cholesky factor of the covariance; eigenmodel.py:903-908) instead of the perturbation draw."""
    N, D = F.shape
    A8, pA = _u8(A != 0)
    rc = lib().orc_eig_resample_F_window(N, D, pA, _f64(Omega), _f64(F), _f64(lmbda), C.c_double(float(mu_0)), C.c_double(sigma_F),
                                         C.c_uint64(seed), C.c_uint32(sweep), C.c_double(beta), int(n_begin), int(n_end), 1 if ref_draw else 0)
    if rc:
        raise RuntimeError("oracle eig_F_window failed rc=%d" % rc)


def eig_omega_block(A, F, lmbda, mu_0, seed, sweep, idx, beta=1.0):
    """omega_{ij} of _resample_Omega (eigenmodel.py:859-874) for i, j in idx only -> (len(idx), len(idx))."""
    N, D = F.shape
    A8, pA = _u8(A != 0)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.zeros((idx.size, idx.size))
    lib().orc_eig_omega_block(N, D, pA, _f64(F), _f64(lmbda), C.c_double(float(mu_0)), C.c_uint64(seed), C.c_uint32(sweep),
                              C.c_double(beta), _i32(idx), int(idx.size), _f64(out))
    return out


def _adj_missing(A):
    """uint8 adjacency with NaN (missing data, eigenmodel.py:121-127) as the byte 2."""
    A = np.asarray(A)
    if A.dtype.kind == "f" and np.isnan(A).any():
        out = (np.where(np.isnan(A), 0, A) != 0).astype(np.uint8)
        out[np.isnan(A)] = 2
        return np.ascontiguousarray(out)
    return np.ascontiguousarray(A != 0, dtype=np.uint8)


def eig_sweep_multi(As, F, lmbda, mu_0, mu_lmbda, mu_mu_0, sigma_mu0, sigma_F, sigma_lmbda, lmbda_given, seed, sweep,
                    parts=PART_EIG_ALL, aux=None, beta=1.0, probit=False):
    """resample(networks) (eigenmodel.py:836-857 / :246-272) for a list of graphs; NaN entries are unobserved.  In-place sweep
    of (F, lmbda); returns (mu_0, aux) with aux = the per-graph auxiliary variables, shape (nG, N, N)."""
    N, D = F.shape
    A8 = np.ascontiguousarray(np.stack([_adj_missing(A) for A in As]))
    nG = A8.shape[0]
    if aux is None:
        aux = np.zeros((nG, N, N))
    mu = C.c_double(float(mu_0))
    common = [_f64(np.ascontiguousarray(mu_lmbda, dtype=np.float64)), C.c_double(mu_mu_0), C.c_double(sigma_mu0), C.c_double(sigma_F),
              C.c_double(sigma_lmbda), int(lmbda_given), C.c_uint64(seed), C.c_uint32(sweep), int(parts)]
    pA = A8.ctypes.data_as(C.POINTER(C.c_uint8))
    if probit:
        rc = lib().orc_probit_sweep_multi(nG, N, D, pA, _f64(F), _f64(lmbda), C.byref(mu), _f64(aux), *common)
    else:
        rc = lib().orc_eig_sweep_multi(nG, N, D, pA, _f64(F), _f64(lmbda), C.byref(mu), _f64(aux), *(common + [C.c_double(beta)]))
    if rc:
        raise RuntimeError("oracle eig_sweep_multi failed rc=%d" % rc)
    return mu.value, aux


def eig_loglik(A, F, lmbda, mu_0):
    N, D = F.shape
    A8 = _adj_missing(A)
    pA = A8.ctypes.data_as(C.POINTER(C.c_uint8))
    lib().orc_eig_loglik.restype = C.c_double
    return lib().orc_eig_loglik(N, D, pA, _f64(F), _f64(lmbda), C.c_double(mu_0))


def probit_sweep(A, F, lmbda, mu_0, mu_lmbda, mu_mu_0, sigma_mu0, sigma_F, sigma_lmbda, lmbda_given, seed, sweep,
                 parts=PART_EIG_ALL, Z=None):
    """Probit eigenmodel (eigenmodel.py:246-379): in-place sweep of (F, lmbda); returns (mu_0, Z)."""
    N, D = F.shape
    A8, pA = _u8(A != 0)
    if Z is None:
        Z = np.zeros((N, N))
    mu = C.c_double(float(mu_0))
    rc = lib().orc_probit_sweep(N, D, pA, _f64(F), _f64(lmbda), C.byref(mu), _f64(Z),
                                _f64(np.ascontiguousarray(mu_lmbda, dtype=np.float64)), C.c_double(mu_mu_0),
                                C.c_double(sigma_mu0), C.c_double(sigma_F), C.c_double(sigma_lmbda), int(lmbda_given),
                                C.c_uint64(seed), C.c_uint32(sweep), int(parts))
    if rc:
        raise RuntimeError("oracle probit_sweep failed rc=%d" % rc)
    return mu.value, Z


def probit_loglik(A, F, lmbda, mu_0):
    N, D = F.shape
    A8 = _adj_missing(A)
    pA = A8.ctypes.data_as(C.POINTER(C.c_uint8))
    lib().orc_probit_loglik.restype = C.c_double
    return lib().orc_probit_loglik(N, D, pA, _f64(F), _f64(lmbda), C.c_double(mu_0))


def ndtri_vec(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty_like(p)
    lib().orc_ndtri_vec(_f64(p), _f64(out), C.c_longlong(p.size))
    return out


def truncnorm_vec(u, mu, a):
    u = np.ascontiguousarray(u, dtype=np.float64)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty_like(u)
    lib().orc_truncnorm_vec(_f64(u), _f64(mu), _i32(a), _f64(out), C.c_longlong(u.size))
    return out


def pg_frac_vec(seed, b, z, n):
    out = np.empty(n)
    lib().orc_pg_frac_vec(C.c_uint64(seed), C.c_double(b), C.c_double(z), _f64(out), C.c_longlong(n))
    return out


def sbm_adj_resample_p_multi(As, c, p, p_self, tau1, tau0, special, seed, sweep, beta=1.0):
    """resample_p(As) for a list of graphs (adjacency.py:620-648): p is updated in place, returns p_self."""
    A8 = np.ascontiguousarray(np.stack([np.asarray(A) != 0 for A in As]).astype(np.uint8))
    nG, N = A8.shape[0], A8.shape[1]
    K = p.shape[0]
    ps = C.c_double(float(p_self))
    cc = np.ascontiguousarray(c, dtype=np.int32)
    rc = lib().orc_sbm_adj_resample_p_multi(nG, N, K, A8.ctypes.data_as(C.POINTER(C.c_uint8)), _i32(cc), _f64(p), C.byref(ps),
                                            C.c_double(tau1), C.c_double(tau0), int(special), C.c_uint64(seed), C.c_uint32(sweep),
                                            C.c_double(beta))
    assert rc == 0
    return ps.value


def eig_noise_row(seed, sweep, N, n):
    """eps[n, :] of the perturbation draw of F[n] (spec/gg_eig.h)."""
    out = np.empty(N)
    lib().orc_eig_noise_row(C.c_uint64(seed), C.c_uint32(sweep), int(N), int(n), _f64(out))
    return out


def pg1_vec(seed, z, n):
    out = np.empty(n)
    lib().orc_pg1_vec(C.c_uint64(seed), C.c_double(z), _f64(out), C.c_longlong(n))
    return out


# ---------------------------------------------------------------- optimised CPU flavour (oracle_fast.cpp, OpenMP)
class FastSweeper(object):
    """Same chain as sbm_adj_sweep / sbm_w_sweep (bit for bit), organised for speed: per-node block statistics
    built once per half-sweep in an OpenMP pass, O(K^2) per scanned node, O(N) per label change."""

    def __init__(self, A, q=None, shift=0, threads=None):
        if threads is None:   # the CPU share of this process, at most 16 (a one-GPU box's share)
            threads = max(1, min(len(os.sched_getaffinity(0)), 16))
        self.A8 = np.ascontiguousarray(A != 0, dtype=np.uint8)
        self.q = q
        N = A.shape[0]
        B = q.shape[2] if q is not None else 1
        lib().orc_fast_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_fast_create(N, B, _p(self.A8, C.c_uint8), _i32(q) if q is not None else None, int(shift), int(threads)))
        self.threads = int(threads)

    def adj_sweep(self, c, m, p, p_self, pi, tau1, tau0, special, seed, sweep):
        ps = np.array([p_self], dtype=np.float64)
        rc = lib().orc_fast_adj_sweep(self.h, m.shape[0], _i32(c), _f64(m), _f64(p), _f64(ps),
                                      _f64(np.ascontiguousarray(pi, dtype=np.float64)), C.c_double(tau1), C.c_double(tau0),
                                      int(special), C.c_uint64(seed), C.c_uint32(sweep))
        if rc:
            raise RuntimeError("oracle fast adj sweep failed rc=%d" % rc)
        return float(ps[0])

    def w_sweep(self, c, m, mu, sigma, mu_self, sigma_self, pi, niw, niw_self, special, seed, sweep):
        rc = lib().orc_fast_w_sweep(self.h, m.shape[0], _i32(c), _f64(m), _f64(mu), _f64(sigma), _f64(mu_self), _f64(sigma_self),
                                    _f64(np.ascontiguousarray(pi, dtype=np.float64)), _f64(niw), _f64(niw_self), int(special),
                                    C.c_uint64(seed), C.c_uint32(sweep))
        if rc:
            raise RuntimeError("oracle fast w sweep failed rc=%d" % rc)

    def close(self):
        if self.h:
            lib().orc_fast_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- latent-distance adjacency model (oracle_latent.cpp)
LD_PART_L, LD_PART_MU0, LD_PART_MUSELF, LD_PART_ALL = 1, 2, 4, 7


def ld_set_threads(n):
    """n > 1: the latent-distance oracle evaluates node-parallel on n threads (same terms, per-node partial sums added in node
    order: agrees with the plain loop to rounding); 1: the plain double loop of adjacency.py:244-275."""
    lib().orc_ld_set_threads(int(n))


def ld_logprob(A, L, mu0, mu_self, sigma, beta=1.0, grad=False):
    """log p(L, mu_0, mu_self, A) (adjacency.py:244-275), likelihood weighted by beta; with grad: (lp, dL, dmu0, dmuself)."""
    N, dim = L.shape
    A8, pA = _u8(A != 0)
    lib().orc_ld_logprob.restype = C.c_double
    L = np.ascontiguousarray(L, dtype=np.float64)
    if not grad:
        return lib().orc_ld_logprob(N, dim, pA, _f64(L), C.c_double(mu0), C.c_double(mu_self), C.c_double(sigma),
                                    C.c_double(beta), None, None, None)
    gL = np.zeros_like(L)
    g0, gs = C.c_double(0), C.c_double(0)
    lp = lib().orc_ld_logprob(N, dim, pA, _f64(L), C.c_double(mu0), C.c_double(mu_self), C.c_double(sigma),
                              C.c_double(beta), _f64(gL), C.byref(g0), C.byref(gs))
    return lp, gL, g0.value, gs.value


def ld_loglik(A, L, mu0, mu_self):
    N, dim = L.shape
    A8, pA = _u8(A != 0)
    lib().orc_ld_loglik.restype = C.c_double
    return lib().orc_ld_loglik(N, dim, pA, _f64(np.ascontiguousarray(L, dtype=np.float64)), C.c_double(mu0), C.c_double(mu_self))


def ld_resample(A, L, mu0, mu_self, sigma, step_sz, accept_rate, seed, sweep, parts=LD_PART_ALL, beta=1.0):
    """One resample (adjacency.py:360-372), L updated in place; returns (mu0, mu_self, step_sz, accept_rate, accepted mask)."""
    N, dim = L.shape
    A8, pA = _u8(A != 0)
    m0, ms, st, ar, acc = C.c_double(mu0), C.c_double(mu_self), C.c_double(step_sz), C.c_double(accept_rate), C.c_int(0)
    rc = lib().orc_ld_resample(N, dim, pA, _f64(L), C.byref(m0), C.byref(ms), C.c_double(sigma), C.byref(st), C.byref(ar),
                               C.c_uint64(seed), C.c_uint32(sweep), int(parts), C.c_double(beta), C.byref(acc))
    if rc:
        raise RuntimeError("oracle ld_resample failed rc=%d" % rc)
    return m0.value, ms.value, st.value, ar.value, acc.value


# ---------------------------------------------------------------- latent-distance Gaussian weights (oracle_latent.cpp)
LDW_PART_L, LDW_PART_B, LDW_PART_COV, LDW_PART_ETA, LDW_PART_ALL = 1, 2, 4, 8, 15


def ldw_logprob(A, q, shift, L, b, sigma, eta, beta=1.0, grad=False):
    """log p(L | b, A, W) of weights.py:703-733 (likelihood weighted by beta); with grad: (lp, dL)."""
    N, dim = L.shape
    A8, pA = _u8(A != 0)
    lib().orc_ldw_logprob.restype = C.c_double
    L = np.ascontiguousarray(L, dtype=np.float64)
    gL = np.zeros_like(L) if grad else None
    lp = lib().orc_ldw_logprob(N, dim, pA, _i32(q), int(shift), _f64(L), C.c_double(b), C.c_double(sigma), C.c_double(eta),
                               C.c_double(beta), _f64(gL) if grad else None)
    return (lp, gL) if grad else lp


def ldw_loglik(A, q, shift, L, b, sigma, mu_self, sigma_self):
    N, dim = L.shape
    A8, pA = _u8(A != 0)
    lib().orc_ldw_loglik.restype = C.c_double
    return lib().orc_ldw_loglik(N, dim, pA, _i32(q), int(shift), _f64(np.ascontiguousarray(L, dtype=np.float64)), C.c_double(b),
                                C.c_double(sigma), C.c_double(mu_self), C.c_double(sigma_self))


def ldw_resample(A, q, shift, L, b, eta, sigma, mu_self, sigma_self, prior, prior_self, seed, sweep, parts=LDW_PART_ALL, beta=1.0):
    """One resample (weights.py:767-773), L updated in place; returns (b, eta, sigma, mu_self, sigma_self, accepted)."""
    N, dim = L.shape
    A8, pA = _u8(A != 0)
    vb, ve, vs, vm, vss, acc = (C.c_double(b), C.c_double(eta), C.c_double(sigma), C.c_double(mu_self), C.c_double(sigma_self),
                                C.c_int(0))
    pr = np.ascontiguousarray(prior, dtype=np.float64)
    prs = np.ascontiguousarray(prior_self, dtype=np.float64)
    assert pr.shape == (2,) and prs.shape == (4,)
    rc = lib().orc_ldw_resample(N, dim, pA, _i32(q), int(shift), _f64(L), C.byref(vb), C.byref(ve), C.byref(vs), C.byref(vm),
                                C.byref(vss), _f64(pr), _f64(prs), C.c_uint64(seed), C.c_uint32(sweep), int(parts),
                                C.c_double(beta), C.byref(acc))
    if rc:
        raise RuntimeError("oracle ldw_resample failed rc=%d" % rc)
    return vb.value, ve.value, vs.value, vm.value, vss.value, acc.value
