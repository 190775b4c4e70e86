"""Model comparison by held-out predictive likelihood: the flow of the reference's examples/demo_model_comparison.py
(synthesise a network from a latent-distance model, fit every adjacency x weight combination by Gibbs sampling,
score the held-out node's row and column) on the engine, without the plots.

    python examples/demo_model_comparison.py [N] [sweeps]
"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.networks import FactorizedNetworkDistribution  # noqa: E402
from graphistician_amd.adjacency import (LatentDistanceAdjacencyDistribution, SBMAdjacencyDistribution,  # noqa: E402
                                         BetaBernoulliAdjacencyDistribution)
from graphistician_amd.weights import NIWGaussianWeightDistribution, SBMGaussianWeightDistribution  # noqa: E402


def demo(N=30, N_samples=200, seed=1234):
    np.random.seed(seed)
    N_test = 1
    true_model = FactorizedNetworkDistribution(N + N_test, LatentDistanceAdjacencyDistribution, {},
                                               NIWGaussianWeightDistribution, {})
    Afull, Wfull = true_model.rvs()
    Afull = Afull.astype(np.uint8)
    Atrain, Wtrain = np.ascontiguousarray(Afull[:N, :N]), np.ascontiguousarray(Wfull[:N, :N, :])
    # the training network is fixed for the whole comparison: read-only arrays are found on the device by identity alone (a
    # writeable array is hashed on every call - it may have been edited in place since the last one; graphistician_amd/engine.py)
    Atrain.setflags(write=False); Wtrain.setflags(write=False)
    Atest_row, Wtest_row = Afull[N:, :].ravel(), Wfull[N:, :, :].reshape((N + N_test, 1))
    Atest_col, Wtest_col = Afull[:, N:].ravel(), Wfull[:, N:, :].reshape((N + N_test, 1))
    results = []
    for adj_model, weight_model in itertools.product(
            [LatentDistanceAdjacencyDistribution, SBMAdjacencyDistribution, BetaBernoulliAdjacencyDistribution],
            [NIWGaussianWeightDistribution, SBMGaussianWeightDistribution]):
        test_model = FactorizedNetworkDistribution(N, adj_model, {}, weight_model, {})
        lps, plls = [], []
        for smpl in range(N_samples):
            test_model.resample((Atrain, Wtrain))
            lps.append(test_model.log_probability((Atrain, Wtrain)))
            plls.append(test_model.approx_predictive_ll(Atest_row, Atest_col, Wtest_row, Wtest_col))
        name = adj_model.__name__ + "-" + weight_model.__name__
        half = N_samples // 2
        results.append((name, float(np.mean(lps[half:])), float(np.mean(plls[half:]))))
        print("%-75s LP %10.2f   held-out PLL %8.2f" % results[-1])
    return results


if __name__ == "__main__":
    demo(*[int(x) for x in sys.argv[1:]])
