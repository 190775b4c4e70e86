"""Annealed importance sampling estimate of the marginal likelihood of an adjacency model.

Restates `estimate_marginal_likelihood` of examples/demo_ais.py:10-62 for the engine-backed
models: temperatures `betas = linspace(0, 1, B)`, `steps_per_B` tempered Gibbs sweeps per
temperature (likelihood raised to beta, priors untouched: graph_models.py:50-64, 168-182,
261-297, 337-371), weight increment `logpr(beta_b) - logpr(beta_{b-1})
= (beta_b - beta_{b-1}) * log_lkhd` evaluated AFTER the transition (:49-52), and
`log Z = -log M + logsumexp(log_w)` (:61).

Deviations, stated: by default the weight increment of temperature b is evaluated at the
state BEFORE the transition at beta_b (Neal 2001, eq. 3: w = prod f_b(x_{b-1}) / f_{b-1}(x_{b-1})),
which makes exp(log_w) an unbiased estimate of Z; the reference evaluates it AFTER the
transition (examples/demo_ais.py:39-52), a right-endpoint rule that over-estimates log Z by
about (E_1[log L] - E_0[log L]) / (2 (B-1)) — `reference_order=True` reproduces that ordering.
Each AIS run starts from its own prior draw (the reference deep-copies
one `(f0, theta0)` for every run, examples/demo_ais.py:21-22, which biases the estimate);
the tempered conditionals are those of the current-generation classes
(`SBMAdjacencyDistribution`, `BetaBernoulliAdjacencyDistribution`), not of the legacy
function-style models whose self edges are counted twice (graph_models.py:276-293).
"""
import numpy as np
from scipy.special import logsumexp

from .adjacency import SBMAdjacencyDistribution, BetaBernoulliAdjacencyDistribution, LatentDistanceAdjacencyDistribution
from .eigenmodel import LogisticEigenmodel, ProbitEigenmodel


def prior_draw(model):
    """Put the model at an independent draw from its prior (parameters and latent labels)."""
    if isinstance(model, SBMAdjacencyDistribution):
        model.m = np.random.dirichlet(model.pi)
        model.c = np.random.choice(model.C, p=model.m, size=model.N)
        model.initialize_from_prior()
    elif isinstance(model, ProbitEigenmodel):
        raise NotImplementedError("AIS: the probit eigenmodel has no tempered conditionals")
    elif isinstance(model, (BetaBernoulliAdjacencyDistribution, LatentDistanceAdjacencyDistribution, LogisticEigenmodel)):
        model.initialize_from_prior()
    elif hasattr(model, "adjacency") and hasattr(model, "weights"):
        # a whole network (networks.py:9-75): both halves from their priors; the tempered path raises the JOINT
        # likelihood p(A | .) p(W | A, .) to beta (weights: build-defined, see weights.py)
        prior_draw(model.adjacency)
        getattr(model.weights, "prior_draw", model.weights.initialize_from_prior)()
    else:
        raise NotImplementedError("AIS is defined for the Beta-Bernoulli (Erdos-Renyi), SBM and latent-distance adjacency models and for whole networks")


def estimate_marginal_likelihood(model, A, N_samples=1000, B=100, steps_per_B=11, return_weights=False,
                                 reference_order=False):
    """log p(A) under `model` by AIS; all sweeps and likelihood evaluations run on the engine."""
    betas = np.linspace(0, 1, B)
    data = model._device_data(A) if hasattr(model, "_device_data") else model._data(A)
    log_weights = np.zeros(N_samples)
    for m in range(N_samples):
        prior_draw(model)
        ratios = np.zeros(B - 1)
        for b in range(1, B):
            if not reference_order:
                ratios[b - 1] = (betas[b] - betas[b - 1]) * model.log_likelihood(data)
                if b == B - 1:
                    break                    # the last transition does not enter the weight
            for _ in range(steps_per_B):
                model.resample(data, beta=betas[b])
            if reference_order:
                ratios[b - 1] = (betas[b] - betas[b - 1]) * model.log_likelihood(data)
        log_weights[m] = ratios.sum()
    log_Z = -np.log(N_samples) + logsumexp(log_weights)
    return (log_Z, log_weights) if return_weights else log_Z


def combine_log_weights(chunks):
    """Pool the log-weights of independent AIS chains (e.g. one chunk per GPU, config 5)."""
    w = np.concatenate([np.asarray(c, dtype=np.float64).ravel() for c in chunks])
    return -np.log(w.size) + logsumexp(w)
