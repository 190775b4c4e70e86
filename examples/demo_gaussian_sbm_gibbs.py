"""Gaussian SBM Gibbs (BASELINE.json configs[1] shape at a size you choose): the sweep loop of
the reference's examples/demo_model_comparison.py:61-74 / demo_gaussian_sbm_gibbs.py on the engine.

    python examples/demo_gaussian_sbm_gibbs.py [N] [C] [sweeps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.networks import FactorizedNetworkDistribution  # noqa: E402
from graphistician_amd.adjacency import SBMAdjacencyDistribution  # noqa: E402
from graphistician_amd.weights import SBMGaussianWeightDistribution  # noqa: E402


def demo(N=512, C=4, N_samples=50, seed=1234):
    np.random.seed(seed)
    true_model = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, {"C": C},
                                               SBMGaussianWeightDistribution, {"C": C})
    A, W = true_model.rvs()
    # the network is fixed for the whole run: read-only arrays are found on the device by identity alone (a writeable array is
    # hashed on every call - it may have been edited in place since the last one; graphistician_amd/engine.py)
    A, W = np.ascontiguousarray(A), np.ascontiguousarray(W)
    A.setflags(write=False); W.setflags(write=False)
    test_model = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, {"C": C},
                                               SBMGaussianWeightDistribution, {"C": C})
    lps = [test_model.log_probability((A, W))]
    t0 = time.time()
    for it in range(N_samples):
        test_model.resample((A, W))
        lps.append(test_model.log_probability((A, W)))
        if it % 10 == 0:
            print("Iteration %d  LP: %.3f" % (it, lps[-1]))
    print("%.1f sweeps/s (with one log_probability per sweep)" % (N_samples / (time.time() - t0)))
    print("true LP = %.3f   final LP = %.3f" % (true_model.log_probability((A, W)), lps[-1]))
    # label agreement up to permutation: contingency table of true vs inferred blocks
    ct = np.zeros((C, C), dtype=int)
    for a, b in zip(true_model.adjacency.c, test_model.adjacency.c):
        ct[a, b] += 1
    print("adjacency blocks, true (rows) vs inferred (cols):\n", ct)
    return lps


if __name__ == "__main__":
    args = [int(x) for x in sys.argv[1:]]
    demo(*args)
