"""Aggregate throughput of R independent Gaussian-SBM chains sharing one GPU (and one resident copy of the data)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
for R in (1, 2, 4, 8):
    models = []
    for r in range(R):
        models += list(bench.build_models(abi, N, K, 1 + r, 0))
    for m in models:
        m.resample(data, 10)
    t0 = time.perf_counter()
    abi.resample_many(models, data, steps)
    el = time.perf_counter() - t0
    print("chains %d: %.0f sweeps/s aggregate (%.1f us per sweep per chain)" % (R, R * steps / el, el / steps * 1e6), flush=True)
    del models
