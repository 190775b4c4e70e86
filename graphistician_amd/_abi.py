"""ctypes binding of libgraphgibbs.so (include/graphgibbs.h).

This is the only bridge between the Python classes and the HIP engine.  There is
no fallback: if the shared library is missing or no HIP device is present the
calls raise (EngineError) instead of silently computing on the CPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GG_LIB_PATH") or os.path.join(_HERE, "_build", "libgraphgibbs.so")   # (GG_LIB_PATH: kernel-tuning builds, tools/scripts/sweepq_variants.sh)

# enums of include/graphgibbs.h
SBM_ADJ, BB_ADJ, BERN_ADJ, SBM_GAUSS_W, NIW_W, FIXED_W, LOGISTIC_EIG, PROBIT_EIG, LATENT_DIST, SBM_GAUSS_SHCOV_W = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
LATENT_DIST_W = 11
SBM_JOINT = 12
(ST_C, ST_M, ST_PI, ST_P, ST_P_SELF, ST_MU, ST_SIGMA, ST_MU_SELF, ST_SIGMA_SELF, ST_NIW, ST_NIW_SELF, ST_SWEEP, ST_F,
 ST_LMBDA, ST_MU0, ST_MU_LMBDA) = range(1, 17)
ST_OMEGA_S = 17
ST_LD_AUX = 18
PART_PARAMS, PART_C, PART_M, PART_OMEGA, PART_F, PART_MU0, PART_LMBDA, PART_ALL = 1, 2, 4, 8, 16, 32, 64, 127
ERR_NO_DEVICE, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NUMERIC = -1, -2, -3, -4, -5


HOST_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64)
HOST_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int64)


class EngineError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("graphgibbs error %d: %s" % (code, text))
        self.code = code


class Config(C.Structure):
    # field for field include/graphgibbs.h `gg_config` (tests/test_abi_and_host.py parses the header and compares)
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("B", C.c_int32),
                ("D", C.c_int32), ("allow_self_connections", C.c_int32), ("special_case_self_conns", C.c_int32),
                ("device", C.c_int32), ("reserved0", C.c_int32), ("seed", C.c_uint64), ("tau1", C.c_double), ("tau0", C.c_double), ("sigma_mu0", C.c_double),
                ("sigma_F", C.c_double), ("sigma_lmbda", C.c_double), ("mu_mu_0", C.c_double),
                ("lmbda_given", C.c_int32), ("wide_weights", C.c_int32), ("shard_begin", C.c_int32),
                ("shard_end", C.c_int32), ("shard_block", C.c_int32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
                ("reserved1", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32),
                ("sweeps", C.c_uint64), ("label_changes", C.c_uint64), ("resolve_rounds", C.c_uint64),
                ("bulk_bytes", C.c_double), ("bulk_ms", C.c_double), ("bulk_launches", C.c_uint64),
                ("total_ms", C.c_double), ("sweeps_fused", C.c_uint64), ("sweeps_two_launch", C.c_uint64),
                ("sweeps_serial", C.c_uint64)]


EXPORTS = ["gg_device_count", "gg_last_error", "gg_version", "gg_data_create", "gg_data_destroy", "gg_data_update", "gg_data_info", "gg_data_quantisation",
           "gg_data_create_sbm", "gg_data_download", "gg_model_create", "gg_model_destroy", "gg_set_state",
           "gg_get_state", "gg_resample", "gg_resample_part", "gg_resample_part_many", "gg_resample_async", "gg_sync", "gg_sync_many", "gg_resample_many", "gg_log_likelihood", "gg_log_likelihood_many", "gg_node_lp", "gg_predictive_ll", "gg_last_stats", "gg_bench_bulk", "gg_data_create_shard", "gg_data_create_sbm_shard", "gg_comm_unique_id", "gg_comm_create_rccl",
           "gg_comm_create_host", "gg_comm_destroy", "gg_comm_info", "gg_shard_sweep", "gg_shard_log_likelihood",
           "gg_set_timing", "gg_bench_stream", "gg_selftest_math", "gg_selftest_draws", "gg_hash64", "gg_data_create_shard_blocks",
           "gg_data_create_sbm_shard_blocks", "gg_log_prior_many", "gg_data_create_wide"]


def build(force=False):
    """Compile the engine for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        cmd = ["make", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else [])
        subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """Load the engine.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(ERR_NO_DEVICE, "%s not found: build it with `python -c 'import __graft_entry__ as g; "
                                              "g.build()'` or `make -C graphistician_amd/csrc`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.gg_last_error.restype = C.c_char_p
        L.gg_version.restype = C.c_char_p
        vp = C.c_void_p
        L.gg_data_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, vp, vp, C.c_int32]
        L.gg_data_create_wide.argtypes = [C.POINTER(vp), C.c_int32, vp, vp, C.c_int32]
        L.gg_data_destroy.argtypes = [vp]
        L.gg_data_update.argtypes = [vp, vp, vp]
        L.gg_data_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int64)]
        L.gg_data_quantisation.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.gg_data_create_sbm.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, vp, vp, C.c_double, vp, vp, C.c_double,
                                         C.c_double, C.c_uint64, C.c_int32]
        L.gg_data_download.argtypes = [vp, vp, vp]
        L.gg_model_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
        L.gg_model_destroy.argtypes = [vp]
        L.gg_set_state.argtypes = [vp, C.c_int32, vp, C.c_size_t]
        L.gg_get_state.argtypes = [vp, C.c_int32, vp, C.c_size_t]
        L.gg_resample.argtypes = [vp, vp, C.c_int32, C.c_double]
        L.gg_resample_part.argtypes = [vp, vp, C.c_int32, C.c_double]
        L.gg_resample_part_many.argtypes = [vp, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_double]
        L.gg_resample_async.argtypes = [vp, vp, C.c_int32, C.c_double]
        L.gg_sync.argtypes = [vp]
        L.gg_sync_many.argtypes = [C.POINTER(vp), C.c_int32]
        L.gg_bench_stream.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        L.gg_resample_many.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int32, C.c_int32, C.c_double]
        L.gg_log_likelihood_many.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int32, C.POINTER(C.c_double)]
        L.gg_log_likelihood.argtypes = [vp, vp, C.POINTER(C.c_double)]
        L.gg_log_prior_many.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(C.c_double)]
        L.gg_node_lp.argtypes = [vp, vp, C.c_int32, vp]
        L.gg_predictive_ll.argtypes = [vp, vp, vp, vp, vp, vp]
        L.gg_last_stats.argtypes = [vp, C.POINTER(Stats)]
        L.gg_set_timing.argtypes = [vp, C.c_int32]
        L.gg_bench_bulk.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.gg_data_create_shard.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32]
        L.gg_data_create_sbm_shard.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, vp, vp, C.c_double, vp, vp, C.c_double,
                                               C.c_double, C.c_uint64, C.c_int32, C.c_int32, C.c_int32]
        L.gg_data_create_shard_blocks.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.gg_data_create_sbm_shard_blocks.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, vp, vp, C.c_double, vp, vp, C.c_double,
                                                      C.c_double, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.gg_hash64.argtypes = [vp, C.c_size_t, C.c_int32, C.POINTER(C.c_uint64)]
        L.gg_comm_unique_id.argtypes = [vp]
        L.gg_comm_create_rccl.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, vp, C.c_int32]
        L.gg_comm_create_host.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, HOST_ALLREDUCE, HOST_ALLGATHER, vp, C.c_int32]
        L.gg_comm_destroy.argtypes = [vp]
        L.gg_comm_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64)]
        L.gg_shard_sweep.argtypes = [vp, vp, vp, C.c_int32]
        L.gg_shard_log_likelihood.argtypes = [vp, vp, vp, C.POINTER(C.c_double)]
        L.gg_selftest_math.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, C.c_int64]
        L.gg_selftest_draws.argtypes = [C.c_int32, C.c_int32, C.c_uint64, vp, C.c_int32, vp, C.c_int64, C.c_int32]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise EngineError(rc, lib().gg_last_error().decode("utf-8", "replace"))


def hash64(a, threads=0):
    """64-bit content hash of a C-contiguous ndarray's buffer, computed on all host threads inside the library (gg_hash64; the
    GIL is released for the call).  134 MB of float64 W: 2-4 ms against 17 ms for a single-threaded hash."""
    out = C.c_uint64(0)
    check(lib().gg_hash64(a.ctypes.data_as(C.c_void_p), a.nbytes, int(threads), C.byref(out)))
    return out.value


def device_count():
    n = C.c_int(0)
    lib().gg_device_count(C.byref(n))
    return n.value


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


A_MISSING = 2     # byte value of an unobserved adjacency entry (GG_A_MISSING)


def _adjacency_u8(A, missing_ok=False):
    """(N, N) adjacency as uint8 {0, 1}.  NaN marks missing data in the reference's eigenmodel (eigenmodel.py:121-127): with
    missing_ok it becomes the byte A_MISSING, which the eigenmodel kernels leave out of every conditional and of the
    log-likelihood; otherwise it is refused - `NaN != 0` would silently upload a missing entry as an edge, and the block
    models define nothing for it (the reference's masked sums turn NaN)."""
    A = np.asarray(A)
    if A.dtype.kind == "f":
        nan = np.isnan(A)
        if nan.any():
            if not missing_ok:
                raise ValueError("the adjacency matrix contains NaN: missing entries are defined for the eigenmodels only")
            out = np.ascontiguousarray(np.where(nan, 0, A) != 0, dtype=np.uint8)
            out[nan] = A_MISSING
            return out
    return np.ascontiguousarray(A != 0, dtype=np.uint8)


class DeviceData(object):
    """Device-resident (A, W): replaces the ndarray pair of networks.py:28,69."""

    def __init__(self, A=None, W=None, device=0, handle=None, N=None, B=None, missing_ok=False, wide=False):
        """wide=True (B = 1): reference-precision weights, 48 bits in two limbs instead of 24 (gg_data_create_wide) - for a model
        created with wide_weights=1."""
        self._h = C.c_void_p()
        self.wide = bool(wide)
        if handle is not None:
            self._h = handle
        else:
            A = _adjacency_u8(A, missing_ok)
            assert A.ndim == 2 and A.shape[0] == A.shape[1]
            N = A.shape[0]
            if W is not None:
                W = np.ascontiguousarray(W, dtype=np.float64)
                if W.ndim == 2:
                    W = W[:, :, None].copy()
                assert W.shape[:2] == (N, N)
                B = W.shape[2]
                if wide:
                    if B != 1:
                        raise ValueError("wide weights are built for one-dimensional weights (B = 1)")
                    check(lib().gg_data_create_wide(C.byref(self._h), N, _ptr(A), _ptr(W), device))
                else:
                    check(lib().gg_data_create(C.byref(self._h), N, B, _ptr(A), _ptr(W), device))
            else:
                if wide:
                    raise ValueError("wide=True needs weights")
                B = 1
                check(lib().gg_data_create(C.byref(self._h), N, 1, _ptr(A), None, device))
        self.N, self.B, self.device = N, B, device
        self.has_w = W is not None or handle is not None
        if handle is None and W is not None:
            self._warn_if_coarse()

    LOW_BITS_WARN = 12

    def quantisation(self):
        """(w_max, w_low, low_bits, shift) of the stored weights (gg_data_quantisation): the fixed-point scale is set by the
        largest |w| on an edge; low_bits = the significant bits the median of |w| keeps (-1: not measured)."""
        wmax, wlow, bits, shift = C.c_double(0), C.c_double(0), C.c_int32(0), C.c_int32(0)
        check(lib().gg_data_quantisation(self._h, C.byref(wmax), C.byref(wlow), C.byref(bits), C.byref(shift)))
        return wmax.value, wlow.value, bits.value, shift.value

    def _warn_if_coarse(self):
        wmax, wlow, bits, shift = self.quantisation()
        if 0 <= bits < self.LOW_BITS_WARN:
            import warnings
            warnings.warn("graphistician_amd: the weights span a wide dynamic range (max |w| = %.3g, median %.3g): stored as "
                          "fixed point with one scale, a typical one keeps only %d significant bits (the reference computes in "
                          "float64). Rescale, clip or transform W if its small entries matter." % (wmax, wlow, bits),
                          RuntimeWarning, stacklevel=3)

    @classmethod
    def shard(cls, A, W, n0=None, n1=None, device=0, blocks=None):
        """Column shard of the network for a chain sharded over several GPUs: the contiguous node range [n0, n1), or with
        blocks=(block, rank, world) the blocks rank, rank + world, ... of `block` consecutive nodes (sharded.Ownership)."""
        A = _adjacency_u8(A)
        N = A.shape[0]
        h = C.c_void_p()
        if W is not None:
            W = np.ascontiguousarray(W, dtype=np.float64)
            if W.ndim == 2:
                W = W[:, :, None].copy()
        B = W.shape[2] if W is not None else 1
        pW = _ptr(W) if W is not None else None
        if blocks is not None:
            check(lib().gg_data_create_shard_blocks(C.byref(h), N, B, _ptr(A), pW, int(blocks[0]), int(blocks[1]), int(blocks[2]), device))
        else:
            check(lib().gg_data_create_shard(C.byref(h), N, B, _ptr(A), pW, n0, n1, device))
        obj = cls(handle=h, N=N, B=B, device=device)
        obj.has_w = W is not None
        return obj

    @classmethod
    def synthetic_sbm(cls, N, c, p, p_self, mu, sigma, mu_self, sigma_self, seed, device=0, shard=None, blocks=None):
        """Planted Gaussian SBM (B = 1) generated on the device; shard=(n0, n1) generates that column shard only, blocks=(block,
        rank, world) the block-interleaved shard of that rank."""
        c = np.ascontiguousarray(c, dtype=np.int32)
        p = np.ascontiguousarray(p, dtype=np.float64)
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        sigma = np.ascontiguousarray(sigma, dtype=np.float64)
        K = p.shape[0]
        h = C.c_void_p()
        if blocks is not None:
            check(lib().gg_data_create_sbm_shard_blocks(C.byref(h), N, K, _ptr(c), _ptr(p), float(p_self), _ptr(mu), _ptr(sigma),
                                                        float(mu_self), float(sigma_self), C.c_uint64(seed), int(blocks[0]),
                                                        int(blocks[1]), int(blocks[2]), device))
        elif shard is None:
            check(lib().gg_data_create_sbm(C.byref(h), N, K, _ptr(c), _ptr(p), float(p_self), _ptr(mu), _ptr(sigma),
                                           float(mu_self), float(sigma_self), C.c_uint64(seed), device))
        else:
            check(lib().gg_data_create_sbm_shard(C.byref(h), N, K, _ptr(c), _ptr(p), float(p_self), _ptr(mu), _ptr(sigma),
                                                 float(mu_self), float(sigma_self), C.c_uint64(seed), int(shard[0]),
                                                 int(shard[1]), device))
        return cls(handle=h, N=N, B=1, device=device)

    def update(self, A, W=None, missing_ok=False):
        """Replace the network in place with another (A, W) of the same shape (gg_data_update)."""
        A = _adjacency_u8(A, missing_ok)
        assert A.shape == (self.N, self.N)
        if W is not None:
            W = np.ascontiguousarray(W, dtype=np.float64)
            if W.ndim == 2:
                W = W[:, :, None].copy()
            assert W.shape == (self.N, self.N, self.B)
        check(lib().gg_data_update(self._h, _ptr(A), _ptr(W) if W is not None else None))
        if W is not None:
            self._warn_if_coarse()

    @property
    def handle(self):
        return self._h

    def info(self):
        N, B, sh, ne = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        check(lib().gg_data_info(self._h, C.byref(N), C.byref(B), C.byref(sh), C.byref(ne)))
        return dict(N=N.value, B=B.value, shift=sh.value, n_edges=ne.value)

    def download(self, with_weights=True):
        A = np.zeros((self.N, self.N), dtype=np.uint8)
        W = np.zeros((self.N, self.N, self.B)) if (with_weights and self.has_w) else None
        check(lib().gg_data_download(self._h, _ptr(A), _ptr(W) if W is not None else None))
        return A, W

    def close(self):
        if self._h:
            lib().gg_data_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model(object):
    """Thin owner of one gg_model handle."""

    def __init__(self, **cfg):
        c = Config()
        for k, v in cfg.items():
            setattr(c, k, v)
        c.struct_size = C.sizeof(Config)
        self.cfg = c
        self._h = C.c_void_p()
        check(lib().gg_model_create(C.byref(self._h), C.byref(c)))

    def set(self, sid, arr, dtype=np.float64):
        a = np.ascontiguousarray(arr, dtype=dtype)
        check(lib().gg_set_state(self._h, sid, _ptr(a), a.nbytes))

    def get(self, sid, shape, dtype=np.float64):
        a = np.zeros(shape, dtype=dtype)
        check(lib().gg_get_state(self._h, sid, _ptr(a), a.nbytes))
        return a

    def resample(self, data, n_sweeps=1, beta=1.0):
        check(lib().gg_resample(self._h, data.handle, n_sweeps, beta))

    def resample_async(self, data, n_sweeps=1, beta=1.0):
        check(lib().gg_resample_async(self._h, data.handle, n_sweeps, beta))

    def sync(self):
        check(lib().gg_sync(self._h))

    def resample_part(self, data, parts, beta=1.0):
        check(lib().gg_resample_part(self._h, data.handle if data is not None else None, parts, beta))

    def resample_part_many(self, datas, parts, beta=1.0):
        """The reference's list-taking forms (gg_resample_part_many): SBM adjacency resample_p(As), eigenmodel resample(networks)."""
        hs = (C.c_void_p * len(datas))(*[d.handle for d in datas])
        check(lib().gg_resample_part_many(self._h, hs, len(datas), parts, beta))

    def log_likelihood(self, data):
        out = C.c_double(0)
        check(lib().gg_log_likelihood(self._h, data.handle, C.byref(out)))
        return out.value

    def predictive_ll(self, a_row, a_col, w_row=None, w_col=None, K=1):
        """ll[k] of a held-out node's row and column for each block k it may belong to (gg_predictive_ll)."""
        ar = np.ascontiguousarray(np.asarray(a_row) != 0, dtype=np.uint8)
        ac = np.ascontiguousarray(np.asarray(a_col) != 0, dtype=np.uint8)
        wr = None if w_row is None else np.ascontiguousarray(w_row, dtype=np.float64)
        wc = None if w_col is None else np.ascontiguousarray(w_col, dtype=np.float64)
        out = np.zeros(K)
        check(lib().gg_predictive_ll(self._h, _ptr(ar), _ptr(ac), _ptr(wr) if wr is not None else None,
                                     _ptr(wc) if wc is not None else None, _ptr(out)))
        return out

    def node_lp(self, data, n, K):
        lp = np.zeros(K)
        check(lib().gg_node_lp(self._h, data.handle, n, _ptr(lp)))
        return lp

    # ---- one chain sharded over several GPUs (graphistician_amd/sharded.py) ----
    def shard_sweep(self, data, comm, n=1):
        check(lib().gg_shard_sweep(self._h, data.handle, comm.handle, int(n)))

    def shard_log_likelihood(self, data, comm):
        out = C.c_double(0)
        check(lib().gg_shard_log_likelihood(self._h, data.handle, comm.handle, C.byref(out)))
        return out.value

    def stats(self):
        s = Stats()
        s.struct_size = C.sizeof(Stats)
        check(lib().gg_last_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_ if f[0] not in ("struct_size", "reserved")}

    def bench_bulk(self, data, reps=20):
        ms, nb = C.c_double(0), C.c_double(0)
        check(lib().gg_bench_bulk(self._h, data.handle, reps, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def set_timing(self, every=1):
        """HIP-event timing of the bulk kernel on every `every`-th sweep (0/False = off)."""
        check(lib().gg_set_timing(self._h, int(every)))

    @property
    def sweep(self):
        return int(self.get(ST_SWEEP, (1,), np.uint32)[0])

    @sweep.setter
    def sweep(self, v):
        self.set(ST_SWEEP, np.array([v], dtype=np.uint32), np.uint32)

    def close(self):
        if self._h:
            lib().gg_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm(object):
    """Communicator of a sharded chain (include/graphgibbs.h): RCCL, or host callbacks for the protocol tests."""

    def __init__(self, handle, keep=()):
        self._h = handle
        self._keep = keep          # ctypes callbacks must outlive the communicator

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib().gg_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def rccl(cls, rank, world, unique_id, device=0):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib().gg_comm_create_rccl(C.byref(h), rank, world, buf, device))
        return cls(h)

    @classmethod
    def host(cls, rank, world, allreduce_sum, allgather, device=0):
        """allreduce_sum(int64 ndarray) -> ndarray (sum over ranks); allgather(int32 ndarray) -> (world, n) ndarray."""
        def ar(user, buf, n):
            try:
                a = np.ctypeslib.as_array(buf, shape=(n,))
                a[:] = allreduce_sum(a.copy())
                return 0
            except Exception:      # never raise through the C frame
                return 1

        def ag(user, send, recv, n):
            try:
                s = np.ctypeslib.as_array(send, shape=(n,)).copy()
                r = np.ctypeslib.as_array(recv, shape=(world * n,))
                r[:] = np.asarray(allgather(s), dtype=np.int32).reshape(-1)
                return 0
            except Exception:
                return 1

        car, cag = HOST_ALLREDUCE(ar), HOST_ALLGATHER(ag)
        h = C.c_void_p()
        check(lib().gg_comm_create_host(C.byref(h), rank, world, car, cag, None, device))
        return cls(h, keep=(car, cag))

    @property
    def handle(self):
        return self._h

    def info(self):
        r, w, k, n, nr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64(), C.c_uint64()
        check(lib().gg_comm_info(self._h, C.byref(r), C.byref(w), C.byref(k), C.byref(n), C.byref(nr)))
        return dict(rank=r.value, world=w.value, rccl=bool(k.value), collectives=n.value, rounds=nr.value)

    def close(self):
        if self._h:
            lib().gg_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sync_many(models):
    """gg_sync for several independent models at once (their restart rounds overlap on the device)."""
    models = [m for m in models if m is not None]
    arr = (C.c_void_p * len(models))(*[m._h for m in models])
    check(lib().gg_sync_many(arr, len(models)))


def bench_stream(nbytes, reps=20, mode=0, device=0):
    """Measured streaming rate of the device in GB/s (mode 0 read-only, 1 copy: read + write bytes counted)."""
    ms = C.c_double(0)
    check(lib().gg_bench_stream(device, int(nbytes), reps, mode, C.byref(ms)))
    return (2 if mode == 1 else 1) * float(nbytes) / ms.value / 1e6


_many_args = {}


def _handles(models, datas):
    """ctypes arrays of the models' and the networks' handles, kept per (models, networks) combination: the reference's loop
    calls `resample` and `log_probability` with the same objects thousands of times, and building the arrays is a few
    microseconds of a call that takes a few tens."""
    if not isinstance(datas, (list, tuple)):
        datas = [datas] * len(models)
    key = (tuple(m._h.value if hasattr(m._h, "value") else m._h for m in models), tuple(d.handle.value if hasattr(d.handle, "value") else d.handle for d in datas))
    hit = _many_args.get(key)
    if hit is None:
        if len(_many_args) > 64:
            _many_args.clear()
        hit = ((C.c_void_p * len(models))(*[m._h for m in models]), (C.c_void_p * len(models))(*[d.handle for d in datas]),
               (C.c_double * len(models))())
        _many_args[key] = hit
    return hit


def log_likelihood_many(models, datas):
    """log_likelihood of several independent models at once (enqueued together, then collected); returns the list of values."""
    marr, darr, out = _handles(models, datas)
    check(lib().gg_log_likelihood_many(marr, darr, len(models), out))
    return [float(v) for v in out]


_prior_args = {}


def log_prior_many(models):
    """log_prior of several SBM adjacency / weight models from their device-resident state (gg_log_prior_many): list of values."""
    key = tuple(m._h.value for m in models)
    hit = _prior_args.get(key)
    if hit is None:
        if len(_prior_args) > 64:
            _prior_args.clear()
        hit = ((C.c_void_p * len(models))(*[m._h for m in models]), (C.c_double * len(models))())
        _prior_args[key] = hit
    check(lib().gg_log_prior_many(hit[0], len(models), hit[1]))
    return [float(v) for v in hit[1]]


def resample_many(models, datas, n_sweeps=1, beta=1.0):
    """n_sweeps x resample for several independent models (one host thread each)."""
    marr, darr, _ = _handles(models, datas)
    check(lib().gg_resample_many(marr, darr, len(models), n_sweeps, beta))


def selftest_math(op, x, y=None, device=0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    yy = np.ascontiguousarray(y, dtype=np.float64) if y is not None else None
    check(lib().gg_selftest_math(device, op, _ptr(x), _ptr(yy) if yy is not None else None, _ptr(out), x.size))
    return out


def selftest_draws(what, seed, params, n, stride=1, device=0):
    params = np.ascontiguousarray(params, dtype=np.float64)
    out = np.empty((n, stride) if stride > 1 else (n,))
    check(lib().gg_selftest_draws(device, what, C.c_uint64(seed), _ptr(params), params.size, _ptr(out), n, stride))
    return out
