"""The reference's loop through the host classes on plain ndarrays: where the microseconds of one iteration go.
    python tools/host_loop.py [N] [K]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from graphistician_amd import _abi as abi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
r = bench.extra_host_classes(abi, 0, N=N, K=K, iters=200)
for k, v in r.items():
    print(k, v)
# profile of the read-only loop
from graphistician_amd.networks import FactorizedNetworkDistribution
from graphistician_amd.adjacency import SBMAdjacencyDistribution
from graphistician_amd.weights import SBMGaussianWeightDistribution
truth = bench.planted_truth(N, K, seed=1234)
dd = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], truth["mu_self"], truth["sigma_self"], seed=1234)
A, W = dd.download()
A.setflags(write=False); W.setflags(write=False)
np.random.seed(11)
model = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, dict(C=K, seed=31), SBMGaussianWeightDistribution, dict(B=1, C=K, seed=32))
for _ in range(80):
    model.resample((A, W)); model.log_probability((A, W))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    model.resample((A, W)); model.log_probability((A, W))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
