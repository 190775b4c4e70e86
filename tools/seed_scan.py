"""How the Gaussian-SBM chain behaves from different prior draws: label changes and time per sweep after burn-in."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
for seed in range(1, 7):
    adj, w = bench.build_models(abi, N, K, seed, 0)
    line = "seed %d:" % seed
    for burn in (10, 40, 150):
        adj.resample(data, burn); w.resample(data, burn)
        c0 = adj.stats()["label_changes"], w.stats()["label_changes"]
        t0 = time.perf_counter()
        abi.resample_many([adj, w], data, 50)
        el = time.perf_counter() - t0
        c1 = adj.stats()["label_changes"], w.stats()["label_changes"]
        line += "  after +%d: %.0f us/sweep, changes/sweep adj %.1f w %.1f |" % (burn, el / 50 * 1e6, (c1[0] - c0[0]) / 50.0, (c1[1] - c0[1]) / 50.0)
    nb_a = len(np.unique(adj.get(abi.ST_C, (N,), np.int32))); nb_w = len(np.unique(w.get(abi.ST_C, (N,), np.int32)))
    print(line, "blocks used adj %d w %d" % (nb_a, nb_w), flush=True)
