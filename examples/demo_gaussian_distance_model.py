"""Latent-distance Gaussian weights on a complete graph: the flow of the reference's
examples/demo_gaussian_distance_model.py (true locations on two circles, fit by Gibbs/HMC) on the engine, without the plots.

    python examples/demo_gaussian_distance_model.py [N] [sweeps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.networks import FactorizedNetworkDistribution  # noqa: E402
from graphistician_amd.adjacency import CompleteAdjacencyDistribution  # noqa: E402
from graphistician_amd.weights import LatentDistanceGaussianWeightDistribution  # noqa: E402


def procrustes_error(L, L_true):
    """RMS distance between the inferred and the true locations after the best rotation / reflection / shift."""
    a, b = L - L.mean(0), L_true - L_true.mean(0)
    u, _, vt = np.linalg.svd(a.T.dot(b))
    return float(np.sqrt(((a.dot(u.dot(vt)) - b) ** 2).sum(1).mean()))


def demo(N=30, N_samples=1000, seed=1234):
    np.random.seed(seed)
    D = 2
    true_model = FactorizedNetworkDistribution(N, CompleteAdjacencyDistribution, {},
                                               LatentDistanceGaussianWeightDistribution, dict(dim=D, b=-12.0))
    r = 1.5 + np.arange(N) // (N / 2.0)
    th = np.linspace(0, 4 * np.pi, N, endpoint=False)
    true_model.weights.L = np.hstack(((r * np.cos(th))[:, None], (r * np.sin(th))[:, None]))
    A, W = true_model.rvs()
    # the network is fixed for the whole run: read-only arrays are found on the device by identity alone (a writeable array is
    # hashed on every call - it may have been edited in place since the last one; graphistician_amd/engine.py)
    A, W = np.ascontiguousarray(A), np.ascontiguousarray(W)
    A.setflags(write=False); W.setflags(write=False)
    test_model = FactorizedNetworkDistribution(N, CompleteAdjacencyDistribution, {},
                                               LatentDistanceGaussianWeightDistribution, dict(dim=D, b=-12.0))
    lps = [test_model.log_probability((A, W))]
    for smpl in range(N_samples):
        test_model.resample((A, W))
        lps.append(test_model.log_probability((A, W)))
        if smpl % 100 == 0:
            w = test_model.weights
            print("Iteration %4d  LP: %10.2f  b: %6.2f  eta: %5.2f  sigma: %6.3f  location error: %.3f"
                  % (smpl, lps[-1], w.b, w.eta, w.cov.sigma[0, 0], procrustes_error(w.L, true_model.weights.L)))
    print("true LP = %.2f   final LP = %.2f" % (true_model.log_probability((A, W)), lps[-1]))
    return lps


if __name__ == "__main__":
    demo(*[int(x) for x in sys.argv[1:]])
