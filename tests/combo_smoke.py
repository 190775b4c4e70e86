"""Every adjacency class x every weight class through FactorizedNetworkDistribution on the GPU: draw a network from the
model, resample a few sweeps, evaluate log_likelihood / log_prior / log_probability, sample predictive parameters.
Looks for host-layer breakage (shapes, engine plumbing, cached device data), not for parity.

    python tests/combo_smoke.py
"""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.networks import FactorizedNetworkDistribution  # noqa: E402
from graphistician_amd import adjacency as AD  # noqa: E402
from graphistician_amd import weights as WT  # noqa: E402
from graphistician_amd import eigenmodel as EM  # noqa: E402


def main():
    N = 48
    rng = np.random.RandomState(0)
    A0 = rng.rand(N, N) < 0.3
    adjs = [("Fixed", AD.FixedAdjacencyDistribution, {"A": A0}), ("Complete", AD.CompleteAdjacencyDistribution, {}),
            ("Empty", AD.EmptyAdjacencyDistribution, {}), ("Bernoulli", AD.BernoulliAdjacencyDistribution, {"p": 0.3}),
            ("BetaBernoulli", AD.BetaBernoulliAdjacencyDistribution, {}),
            ("LatentDistance", AD.LatentDistanceAdjacencyDistribution, {"dim": 2}),
            ("SBM", AD.SBMAdjacencyDistribution, {"C": 3}),
            ("LogisticEigen", EM.LogisticEigenmodel, {"D": 2}), ("ProbitEigen", EM.ProbitEigenmodel, {"D": 2})]
    bad = 0
    for B in (1, 2):
        wts = [("FixedGaussian", WT.FixedGaussianWeightDistribution, {"B": B, "mu": np.zeros(B), "sigma": np.eye(B)}),
               ("NIW", WT.NIWGaussianWeightDistribution, {"B": B}),
               ("SBMGaussian", WT.SBMGaussianWeightDistribution, {"B": B, "C": 3}),
               ("SharedCov", WT.SBMGaussianWeightSharedCov, {"B": B, "C": 3})]
        if B == 1:
            wts.append(("LatentDistanceGaussian", WT.LatentDistanceGaussianWeightDistribution, {"B": 1, "dim": 2}))
        for an, ac, ak in adjs:
            for wn, wc, wk in wts:
                tag = "%s x %s (B=%d)" % (an, wn, B)
                try:
                    np.random.seed(1)
                    model = FactorizedNetworkDistribution(N, ac, dict(ak), wc, dict(wk))
                    A, W = model.rvs()
                    assert A.shape == (N, N) and W.shape == (N, N, B), (A.shape, W.shape)
                    data = (A, W)
                    for _ in range(3):
                        model.resample(data)
                    ll, lp = model.log_likelihood(data), model.log_prior()
                    assert np.isfinite(ll) and np.isfinite(lp), (ll, lp)
                    assert np.isfinite(model.log_probability(data))
                    if an != "Fixed":       # (a fixed adjacency matrix has no parameters to sample: raises, as in the reference)
                        model.sample_predictive_parameters()
                except Exception:
                    bad += 1
                    print("FAIL", tag, flush=True)
                    traceback.print_exc(limit=4)
    print("combo smoke: %d failures" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
