"""Long run of the reference's loop (one sweep per call, a log-likelihood after each: the epilogue form) against the oracle:
labels and parameters bit for bit every 25 iterations, every log-likelihood to 1e-9.  Test infrastructure (checks against
oracle/); part of the release soak (tools/scripts/soak_r03.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from tests.helpers import AdjState, WState  # noqa: E402
from graphistician_amd import _abi as abi  # noqa: E402


def run(N, K, B, special, iters, seed):
    rng = np.random.RandomState(seed)
    Kt = min(K, 4)
    ctrue = rng.choice(Kt, size=N)
    ptrue = np.full((Kt, Kt), 0.08) + np.eye(Kt) * 0.8
    A = (rng.rand(N, N) < ptrue[np.ix_(ctrue, ctrue)]).astype(np.uint8)
    W = (rng.randn(Kt, Kt, B) * 3.0)[ctrue][:, ctrue] + 0.3 * rng.randn(N, N, B)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=seed + 1), WState(N, K, B, seed=seed + 2)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=seed + 3, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=special)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=seed + 4, special_case_self_conns=special)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    w.set(abi.ST_MU_SELF, sw.mu_self); w.set(abi.ST_SIGMA_SELF, sw.sigma_self)
    ps = sa.p_self
    for it in range(iters):
        abi.resample_many([adj, w], d, 1)
        ll = abi.log_likelihood_many([adj, w], d)
        ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, special, seed + 3, it)
        O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, special, seed + 4, it)
        ref = (O.sbm_adj_loglik(A, sa.c, sa.p, ps, True, bool(special)),
               O.sbm_w_loglik(A, q, shift, sw.c, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, bool(special)))
        if not (abs(ll[0] - ref[0]) <= 1e-9 * abs(ref[0]) + 1e-7 and abs(ll[1] - ref[1]) <= 1e-9 * abs(ref[1]) + 1e-7):
            return "log-likelihood differs at iteration %d: %r vs %r" % (it, ll, ref)
        if it % 25 == 24 or it == iters - 1:
            ok = (np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c) and
                  np.array_equal(adj.get(abi.ST_M, (K,)), sa.m) and np.array_equal(w.get(abi.ST_M, (K,)), sw.m) and
                  np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p) and np.array_equal(w.get(abi.ST_MU, (K, K, B)), sw.mu))
            if not ok:
                return "state differs at iteration %d" % it
    st = adj.stats()
    return "ok: N=%d K=%d B=%d special=%d, %d iterations (%d of the adjacency model's calls in the quiet two-launch form)" % (
        N, K, B, special, iters, st["sweeps_two_launch"])


if __name__ == "__main__":
    bad = 0
    for cfg in ((600, 6, 1, 1, 300, 11), (500, 40, 2, 0, 300, 12), (400, 100, 1, 1, 300, 13)):
        msg = run(*cfg)
        print("soak_calls", msg, flush=True)
        bad += not msg.startswith("ok")
    sys.exit(1 if bad else 0)
