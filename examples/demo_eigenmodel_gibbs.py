"""Eigenmodel fitted by Gibbs sampling: the flow of the reference's examples/demo_eigenmodel_gibbs.py (probit link)
and demo_logistic_eigenmodel_gibbs.py (logistic link, Polya-gamma augmentation) on the engine, without the plots.

    python examples/demo_eigenmodel_gibbs.py [probit|logistic] [N] [D] [sweeps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.eigenmodel import LogisticEigenmodel, ProbitEigenmodel  # noqa: E402


def demo(link="probit", N=30, D=2, N_samples=1000, seed=1234):
    np.random.seed(seed)
    cls = ProbitEigenmodel if link == "probit" else LogisticEigenmodel
    p, sigma_F = 0.01, 9.0            # sparse base rate, wide features: the reference demo's setting
    true_model = cls(N=N, D=D, p=p, sigma_F=sigma_F, lmbda=np.ones(D))
    A = true_model.rvs().astype(np.uint8)
    # the network is fixed for the whole run: read-only arrays are found on the device by identity alone (a writeable array is
    # hashed on every call - it may have been edited in place since the last one; graphistician_amd/engine.py)
    A.setflags(write=False)
    test_model = cls(N=N, D=D, p=p, sigma_F=sigma_F, lmbda=np.ones(D))
    lps = [test_model.log_probability(A)]
    for smpl in range(N_samples):
        test_model.resample(A)
        lps.append(test_model.log_probability(A))
        if smpl % 100 == 0:
            print("Iteration %4d  LP: %9.2f  mu_0: %6.2f  lambda: %s" % (smpl, lps[-1], test_model.mu_0, np.round(test_model.lmbda, 2)))
    print("true LP = %.2f   final LP = %.2f   edges: %d of %d" % (true_model.log_probability(A), lps[-1], A.sum(), N * N))
    return lps


if __name__ == "__main__":
    a = sys.argv[1:]
    demo(*([a[0]] + [int(x) for x in a[1:]] if a else []))
