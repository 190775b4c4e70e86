"""More than 64 blocks: sweeps/s and the forms the sweeps ran in: python tools/wide_probe.py [N K steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
for which, ms in (("adj", [adj]), ("w", [w]), ("both", [adj, w])):
    abi.resample_many(ms, data, 30)
    s0 = [m.stats() for m in ms]
    t0 = time.perf_counter()
    abi.resample_many(ms, data, steps)
    el = time.perf_counter() - t0
    s1 = [m.stats() for m in ms]
    print("%s: %.0f sweeps/s, %.1f us per sweep" % (which, steps / el, el / steps * 1e6))
    for a, b in zip(s0, s1):
        print("   ", {k: b[k] - a[k] for k in ("label_changes", "sweeps_fused", "sweeps_two_launch", "sweeps_serial", "resolve_rounds") if k in b})
