"""Sweeps/s of ONE half of the Gaussian-SBM chain alone on the device (adj | w | both): python tools/one_model.py which [N K steps chain_seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
which = sys.argv[1] if len(sys.argv) > 1 else "w"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 400
cseed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, cseed, 0)
ms = {"adj": [adj], "w": [w], "both": [adj, w]}[which]
abi.resample_many(ms, data, 30)
for m in ms:
    m.resample(data, 20)
t0 = time.perf_counter()
abi.resample_many(ms, data, steps)
el = time.perf_counter() - t0
print("%s alone: %.0f sweeps/s, %.1f us per sweep" % (which, steps / el, el / steps * 1e6))
