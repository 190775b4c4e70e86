"""BASELINE.json configs[0]: Gaussian Erdos-Renyi Gibbs, N=50 (plumbing case).

The harness of the reference's examples/demo_gaussian_er_gibbs.py:36-43 (LP, then
N_samples x (resample -> LP)) against the current API: Beta-Bernoulli adjacency + NIW
Gaussian weights (the live stand-in for the stale `GaussianErdosRenyiFixedSparsity`
import at :7, SURVEY.md §8d cfg 1).  Runs on the MI355X engine; needs a GPU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.networks import FactorizedNetworkDistribution  # noqa: E402
from graphistician_amd.adjacency import BetaBernoulliAdjacencyDistribution  # noqa: E402
from graphistician_amd.weights import NIWGaussianWeightDistribution  # noqa: E402


def demo(seed=1234, N=50, B=2, N_samples=100):
    np.random.seed(seed)
    true_model = FactorizedNetworkDistribution(N, BetaBernoulliAdjacencyDistribution, {},
                                               NIWGaussianWeightDistribution, {"B": B})
    A, W = true_model.rvs()
    # the network is fixed for the whole run: read-only arrays are found on the device by identity alone (a writeable array is
    # hashed on every call - it may have been edited in place since the last one; graphistician_amd/engine.py)
    A, W = np.ascontiguousarray(A), np.ascontiguousarray(W)
    A.setflags(write=False); W.setflags(write=False)
    test_model = FactorizedNetworkDistribution(N, BetaBernoulliAdjacencyDistribution, {},
                                               NIWGaussianWeightDistribution, {"B": B})
    lps = [test_model.log_probability((A, W))]
    for it in range(N_samples):
        test_model.resample((A, W))
        lps.append(test_model.log_probability((A, W)))
        if it % 10 == 0:
            print("Iteration %d  LP: %.3f" % (it, lps[-1]))
    print("true p = %.3f   inferred p = %.3f" % (true_model.adjacency.p, test_model.adjacency.p))
    print("true LP = %.3f  final LP = %.3f" % (true_model.log_probability((A, W)), lps[-1]))
    return lps


if __name__ == "__main__":
    demo(*[int(x) for x in sys.argv[1:]])
