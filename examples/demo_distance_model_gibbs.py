"""Latent-distance adjacency model fitted by HMC-within-Gibbs: the flow of the reference's
examples/demo_distance_model_gibbs.py on the engine, without the plots.

    python examples/demo_distance_model_gibbs.py [N] [sweeps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.adjacency import LatentDistanceAdjacencyDistribution  # noqa: E402


def demo(N=30, N_samples=1000, seed=1234):
    np.random.seed(seed)
    D = 2
    true_model = LatentDistanceAdjacencyDistribution(N=N, dim=D)
    # true locations on a grid
    w, s = 5, 0.8
    true_model.L = np.hstack(((s * (np.arange(N) % w))[:, None], (s * (np.arange(N) // w))[:, None]))
    true_model.mu_0 = 1.0
    A = true_model.rvs().astype(np.uint8)
    # the network is fixed for the whole run: read-only arrays are found on the device by identity alone (a writeable array is
    # hashed on every call - it may have been edited in place since the last one; graphistician_amd/engine.py)
    A.setflags(write=False)
    test_model = LatentDistanceAdjacencyDistribution(N=N, dim=D)
    lps = [test_model.log_probability(A)]
    for smpl in range(N_samples):
        test_model.resample(A)
        lps.append(test_model.log_probability(A))
        if smpl % 100 == 0:
            print("Iteration %4d  LP: %9.2f  mu_0: %5.2f  mu_self: %5.2f  step: %.4f"
                  % (smpl, lps[-1], test_model.mu_0, test_model.mu_self, test_model._L_step_sz))
    print("true LP = %.2f   final LP = %.2f" % (true_model.log_probability(A), lps[-1]))
    return lps


if __name__ == "__main__":
    demo(*[int(x) for x in sys.argv[1:]])
