"""The legacy joint Gaussian SBM (one block assignment per node for adjacency AND weights,
reference graphistician/stochastic_block_model.py:202-309, 589-678) on the engine.

    python examples/demo_joint_sbm_gibbs.py [N] [C] [sweeps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.stochastic_block_model import GaussianStochasticBlockModel  # noqa: E402


def demo(N=400, C=4, N_samples=40, seed=1234):
    np.random.seed(seed)
    true_model = GaussianStochasticBlockModel(N, 1, C=C, pi=5.0)
    A, W = true_model.rvs()
    # the network is fixed for the whole run: read-only arrays are found on the device by identity alone (a writeable array is
    # hashed on every call - it may have been edited in place since the last one; graphistician_amd/engine.py)
    A, W = np.ascontiguousarray(A), np.ascontiguousarray(W)
    A.setflags(write=False); W.setflags(write=False)
    test_model = GaussianStochasticBlockModel(N, 1, C=C, pi=5.0)
    lls = [test_model.log_likelihood((A, W))]
    t0 = time.time()
    for it in range(N_samples):
        test_model.resample((A, W))
        lls.append(test_model.log_likelihood((A, W)))
        if it % 10 == 0:
            print("Iteration %d  LL: %.3f" % (it, lls[-1]))
    print("%.1f sweeps/s (with one log_likelihood per sweep)" % (N_samples / (time.time() - t0)))
    print("true LL = %.3f   final LL = %.3f" % (true_model.log_likelihood((A, W)), lls[-1]))
    ct = np.zeros((C, C), dtype=int)
    for a, b in zip(true_model.c, test_model.c):
        ct[a, b] += 1
    print("blocks, true (rows) vs inferred (cols):\n", ct)
    return lls


if __name__ == "__main__":
    demo(*[int(x) for x in sys.argv[1:]])
