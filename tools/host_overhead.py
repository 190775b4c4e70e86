"""Measures host enqueue time vs total time for the async sweep pipeline."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N, K = 4096, 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
for _ in range(10):
    adj.resample(data, 1); w.resample(data, 1)
for timing in (False, True):
    adj.set_timing(timing); w.set_timing(timing)
    for mode in ("both", "adj", "w", "batched"):
        steps = 200
        t0 = time.perf_counter()
        if mode == "batched":
            adj.resample_async(data, steps); w.resample_async(data, steps)
        else:
            for _ in range(steps):
                if mode in ("both", "adj"): adj.resample_async(data, 1)
                if mode in ("both", "w"): w.resample_async(data, 1)
        t1 = time.perf_counter()
        adj.sync(); w.sync()
        t2 = time.perf_counter()
        print("timing=%s mode=%-7s enqueue %.1f us/step, total %.1f us/step" % (timing, mode, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
