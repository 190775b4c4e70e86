"""Per-call cost of the reference's driving loops at the benchmark configuration (N=4096, K=16, quiet chain):
   `resample` one sweep per call, `log_likelihood` per call, and the demo loop `resample; log_probability`
   (examples/demo_model_comparison.py:69-72 in the reference)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N, K = int(os.environ.get("CP_N", 4096)), 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
adj.set_timing(0); w.set_timing(0)
abi.resample_many([adj, w], [data, data], 50)
reps = 300
def timed(f):
    f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
print("batched            : %.1f us/sweep" % (timed(lambda: abi.resample_many([adj, w], [data, data], 100)) / 100))
print("one sweep per call : %.1f us" % timed(lambda: abi.resample_many([adj, w], [data, data], 1)))
print("log_likelihood     : %.1f us" % timed(lambda: abi.log_likelihood_many([adj, w], [data, data])))
def demo():
    abi.resample_many([adj, w], [data, data], 1)
    abi.log_likelihood_many([adj, w], [data, data])
print("resample; log_prob : %.1f us" % timed(demo))
print("adj LL alone       : %.1f us" % timed(lambda: adj.log_likelihood(data)))
print("w   LL alone       : %.1f us" % timed(lambda: w.log_likelihood(data)))
abi.resample_many([adj, w], [data, data], 200)
print("LL after a 200-sweep call : %.1f us" % timed(lambda: abi.log_likelihood_many([adj, w], [data, data])))
adj.set_timing(12); w.set_timing(12)
abi.resample_many([adj, w], [data, data], 200)
sa, sw = adj.stats(), w.stats()
adj.set_timing(0); w.set_timing(0)
print("LL after a timed 200-sweep call + stats : %.1f us" % timed(lambda: abi.log_likelihood_many([adj, w], [data, data])))
