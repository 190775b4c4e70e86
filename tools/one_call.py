"""The reference's loop at the benchmark configuration, for a kernel trace: python tools/one_call.py [reps] [ll]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with_ll = len(sys.argv) > 2 and sys.argv[2] == "ll"
N, K = 4096, 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
adj.set_timing(0); w.set_timing(0)
abi.resample_many([adj, w], [data, data], 50)
t0 = time.perf_counter()
for _ in range(reps):
    abi.resample_many([adj, w], [data, data], 1)
    if with_ll:
        abi.log_likelihood_many([adj, w], [data, data])
print("%.1f us per iteration" % ((time.perf_counter() - t0) / reps * 1e6))
