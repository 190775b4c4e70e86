"""Quiet chain, fused launches: timing marks inside the last k_sweep_q of each model (GG_SWQ_DBG=1): python tools/swq_probe.py [which]"""
import os, sys, time
os.environ.setdefault("GG_SWQ_DBG", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
which = sys.argv[1] if len(sys.argv) > 1 else "both"
N, K = 4096, 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
ms = {"adj": [adj], "w": [w], "both": [adj, w]}[which]
for m in ms: m.set_timing(0)
abi.resample_many(ms, data, 60)
for rep in range(4):
    t0 = time.perf_counter()
    abi.resample_many(ms, data, 200)
    el = time.perf_counter() - t0
    print("%s: %.1f us per sweep" % (which, el / 200 * 1e6), flush=True)
