"""Per-sweep trace of the Gaussian-SBM chain: time, label changes, resolve rounds (burn-in vs steady state)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 40
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
prev = [0, 0, 0, 0]
for s in range(S):
    t0 = time.perf_counter(); adj.resample(data, 1); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); w.resample(data, 1); tw = time.perf_counter() - t0
    sa, sw = adj.stats(), w.stats()
    cur = [sa["label_changes"], sa["resolve_rounds"], sw["label_changes"], sw["resolve_rounds"]]
    print("sweep %3d  adj %8.2f ms  chg %6d rounds %4d | w %8.2f ms  chg %6d rounds %4d" % (
        s, ta * 1e3, cur[0] - prev[0], cur[1] - prev[1], tw * 1e3, cur[2] - prev[2], cur[3] - prev[3]), flush=True)
    prev = cur
