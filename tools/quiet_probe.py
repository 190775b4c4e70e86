"""Quiet chain at the benchmark configuration, both models, batched: us per sweep (best of 5 x 400 sweeps): python tools/quiet_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N, K = 4096, 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
ms = {"adj": [adj], "w": [w], "both": [adj, w]}[which]
for m in ms: m.set_timing(0)
abi.resample_many(ms, data, 60)
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    abi.resample_many(ms, data, 400)
    best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
print("%s %s: %.2f us per sweep" % (which, {k: v for k, v in os.environ.items() if k.startswith("GG_")}, best))
