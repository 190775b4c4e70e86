"""Long-run parity soak: a Gaussian-SBM chain on the engine against the (multi-threaded, bit-identical) CPU oracle for
many sweeps, compared every `every` sweeps: labels and parameters must stay bit-identical throughout."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from tests.helpers import planted_sbm, AdjState, WState
from graphistician_amd import _abi as abi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
total = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
every = int(sys.argv[4]) if len(sys.argv) > 4 else 100
A, W, _ = planted_sbm(N, K, seed=77)
q, shift = O.quantise(A, W)
fs = O.FastSweeper(A, q, shift)
sa, sw = AdjState(N, K, seed=1), WState(N, K, 1, seed=2)
d = abi.DeviceData(A, W)
adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=91, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=92, special_case_self_conns=1)
w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
ps, s = sa.p_self, 0
t0 = time.time()
while s < total:
    abi.resample_many([adj, w], d, every)
    for _ in range(every):
        ps = fs.adj_sweep(sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, 1, 91, s)
        fs.w_sweep(sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, 1, 92, s)
        s += 1
    ok = (np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p)
          and np.array_equal(adj.get(abi.ST_M, (K,)), sa.m) and np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c)
          and np.array_equal(w.get(abi.ST_MU, (K, K, 1)), sw.mu) and np.array_equal(w.get(abi.ST_SIGMA, (K, K, 1, 1)), sw.sigma)
          and np.array_equal(w.get(abi.ST_M, (K,)), sw.m))
    if not ok:
        print("MISMATCH after sweep", s); sys.exit(1)
print("soak ok: N=%d K=%d %d sweeps bit-identical (adj changes %d, w changes %d) in %.1f s" % (
    N, K, s, adj.stats()["label_changes"], w.stats()["label_changes"], time.time() - t0))
