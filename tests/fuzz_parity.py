"""Randomised parity stress: small Gaussian-SBM networks of random shape (N, K, B, density, self-connection conventions),
a few sweeps each in uneven chunks, engine against the oracle, labels and parameters bit for bit.  Run it under the
scan-form switches as well (GG_SCAN=0 / 1, GG_SCAN_SPIN=2, GG_AHEAD=0).

    python tests/fuzz_parity.py [cases] [seed]        (FUZZ_VERBOSE=1 prints every configuration,
                                                       FUZZ_EXTENDED=1 adds tempering and shared-covariance weights)

Test infrastructure (it checks against oracle/): tests/test_gpu_edge_cases.py runs a short stream of it.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import planted_sbm, AdjState, WState  # noqa: E402
from graphistician_amd import _abi as abi  # noqa: E402


def one_case(N, K, B, special, seed, chunks, beta=1.0, shared_cov=False):
    """Adjacency and weight models of one random network, advanced in the given chunks (tempered by beta; the weights
    optionally with one covariance shared by all block pairs); None or a mismatch message."""
    A, W, _ = planted_sbm(N, K, B=B, seed=seed % 1000)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=seed % 977), WState(N, K, B, seed=seed % 991)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=seed, tau1=1.0, tau0=1.0, allow_self_connections=1,
                    special_case_self_conns=special)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    niw = sw.niw
    if shared_cov:
        sw.sigma[:] = sw.sigma[0, 0]
        niw = O.pack_niw(np.full(B, 0.2), np.eye(B) * 1.3, B + 2, 1.0)
    w = abi.Model(kind=abi.SBM_GAUSS_SHCOV_W if shared_cov else abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=seed + 1,
                  special_case_self_conns=special)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    w.set(abi.ST_MU_SELF, sw.mu_self); w.set(abi.ST_SIGMA_SELF, sw.sigma_self)
    ps, done = sa.p_self, 0
    for chunk in chunks:
        abi.resample_many([adj, w], d, chunk, beta)
        for s in range(done, done + chunk):
            ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, special, seed, s, beta=beta)
            if shared_cov:
                O.sbm_wsh_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, niw, sw.niw_self,
                                seed + 1, s, beta=beta, special=special)
            else:
                O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, niw, sw.niw_self,
                              special, seed + 1, s, beta=beta)
        done += chunk
        ok = (np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p) and
              np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c) and np.array_equal(w.get(abi.ST_MU, (K, K, B)), sw.mu) and
              np.array_equal(w.get(abi.ST_SIGMA, (K, K, B, B)), sw.sigma) and np.array_equal(w.get(abi.ST_M, (K,)), sw.m))
        if not ok:
            return "N=%d K=%d B=%d special=%d seed=%d beta=%g shared_cov=%d after %d sweeps" % (N, K, B, special, seed, beta,
                                                                                                  shared_cov, done)
        # log p(A | .) + log p(W | A, .) of the network (networks.py:28-31) between the chunks: after a quiet sweep the engine
        # takes it from the statistics counted ahead (float sums in another order than the oracle's: 1e-9)
        ll = float(np.sum(abi.log_likelihood_many([adj, w], d)))
        ref = (O.sbm_adj_loglik(A, sa.c, sa.p, ps, True, bool(special)) +
               O.sbm_w_loglik(A, q, shift, sw.c, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, bool(special)))
        if not (abs(ll - ref) <= 1e-9 * abs(ref) + 1e-7 or (np.isnan(ll) and np.isnan(ref)) or (np.isinf(ref) and ll == ref)):
            return "N=%d K=%d B=%d special=%d seed=%d beta=%g shared_cov=%d after %d sweeps: log-likelihood %r, oracle %r" % (
                N, K, B, special, seed, beta, shared_cov, done, ll, ref)
    return None


def run(cases, seed0, verbose=False, extended=False):
    """`cases` random configurations from the stream seeded with `seed0`; returns the list of mismatch messages.
    extended: tempered sweeps (beta < 1) and shared-covariance weights as well."""
    rng = np.random.RandomState(seed0)
    bad = []
    for case in range(cases):
        N = int(rng.choice([5, 17, 33, 64, 65, 100, 129, 200, 257, 300]))
        # (more than 64 blocks: every scan form since round 3, also under forced exchange timeouts)
        ks = [2, 3, 4, 7, 10, 16, 17, 27, 33, 37, 40, 50, 57, 64, 65, 100, 128, 200, 256]   # (10, 27: padding steps of gg_lp_mfma_small; 37, 50, 57: of gg_lp_mfma)
        K = min(int(rng.choice(ks)), N)
        B = int(rng.choice([1, 1, 1, 2, 3, 4]))
        special = int(rng.rand() < 0.7)
        seed = int(rng.randint(1, 1 << 30))
        chunks = (1, int(rng.randint(1, 4)), int(rng.randint(1, 12)))
        beta, shared = 1.0, False
        if extended:
            beta = float(rng.choice([1.0, 1.0, 0.5, 0.83, 0.0]))
            shared = bool(rng.rand() < 0.35)
        if verbose:
            print("case %d: N=%d K=%d B=%d special=%d seed=%d chunks=%s beta=%g shared_cov=%d"
                  % (case, N, K, B, special, seed, chunks, beta, shared), flush=True)
        msg = one_case(N, K, B, special, seed, chunks, beta, shared)
        if msg:
            bad.append("case %d: %s" % (case, msg))
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0, verbose=bool(os.environ.get("FUZZ_VERBOSE")),
              extended=bool(os.environ.get("FUZZ_EXTENDED")))
    for b in bad:
        print("MISMATCH", b)
    if bad:
        sys.exit(1)
    print("fuzz ok: %d cases" % n)
