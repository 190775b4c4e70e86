"""Quiet chain, both models: us per sweep for calls of n sweeps, timing off / on: python tools/quiet_probe2.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N, K = 4096, 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
ms = [adj, w]
abi.resample_many(ms, data, 60)
for timing in (0, 3):
    for n in (400, 20):
        for m in ms: m.set_timing(timing)
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            abi.resample_many(ms, data, n)
            best = min(best, (time.perf_counter() - t0) / n * 1e6)
        st = w.stats()
        print("timing %d, calls of %d sweeps: %.2f us per sweep; bulk launch %.1f us (%d sampled)" % (timing, n, best, 1e3 * st["bulk_ms"] / max(1, st["bulk_launches"]), st["bulk_launches"]), flush=True)
