"""Network models with the interface of graphistician/networks.py: a factorised
pair (adjacency distribution, weight distribution) driven by
`resample((A, W))` / `log_likelihood((A, W))`.  Both halves run on the MI355X
engine; the (A, W) pair is uploaded once and cached by identity (engine.py)."""
import numpy as np

from . import _abi
from . import adjacency
from . import weights
from .abstractions import NetworkDistribution, AdjacencyDistribution, WeightDistribution, GibbsSampling
from .engine import as_device_data


def _on_engine(part):
    return hasattr(part, "_engine")


class FactorizedNetworkDistribution(NetworkDistribution, GibbsSampling):
    """networks.py:9-75: p(A, W) = p(A | adjacency model) p(W | A, weight model).  The two halves share nothing but the
    data, so the engine runs them on separate streams and this class only pairs them up."""

    def __init__(self, N, adjacency_class, adjacency_kwargs, weight_class, weight_kwargs):
        super(FactorizedNetworkDistribution, self).__init__()
        assert issubclass(adjacency_class, AdjacencyDistribution) and issubclass(weight_class, WeightDistribution)
        self.N = N
        self._halves = (adjacency_class(N, **adjacency_kwargs), weight_class(N, **weight_kwargs))

    adjacency = property(lambda self: self._halves[0])
    weights = property(lambda self: self._halves[1])

    # -- engine plumbing ---------------------------------------------------------------------------------------
    def _device_data(self, data):
        """The (A, W) pair as a device handle (uploaded once, cached by identity); handles pass through."""
        if isinstance(data, _abi.DeviceData):
            return data
        A, W = data
        adj, wts = self._halves
        return as_device_data(A, W, device=getattr(wts, "_device", getattr(adj, "_device", 0)), wide=getattr(wts, "_wide", False))

    def _adjacency_arg(self, d, data):
        # host-only adjacency classes (Fixed / Complete / Empty) take the matrix itself
        return d if _on_engine(self.adjacency) else data[0]

    # -- densities (networks.py:28-37) ---------------------------------------------------------------------------
    def log_likelihood(self, data):
        d = self._device_data(data)
        adj, wts = self._halves
        if _on_engine(adj) and _on_engine(wts):
            assert d.N == self.N
            # both halves' passes are enqueued before either is waited for (gg_log_likelihood_many)
            return sum(_abi.log_likelihood_many([adj._engine(), wts._engine()], d))
        return adj.log_likelihood(self._adjacency_arg(d, data)) + wts.log_likelihood(d)

    def log_prior(self):
        halves = self._halves
        if all(getattr(h, "_device_prior", False) and getattr(h, "_model", None) is not None for h in halves):
            # both priors in one call: the two small launches are enqueued before either is waited for (gg_log_prior_many)
            return sum(_abi.log_prior_many([h._engine() for h in halves]))
        return sum(half.log_prior() for half in halves)

    # -- sampling (networks.py:39-58) ------------------------------------------------------------------------------
    def rvs(self, *args, **kwargs):
        return self.adjacency.rvs(), self.weights.rvs()

    def sample_predictive_parameters(self):
        # (the reference asks the adjacency model twice at :47-48; the weight parameters are what is meant)
        return tuple(half.sample_predictive_parameters() for half in self._halves)

    def sample_predictive_distribution(self):
        return tuple(self.adjacency.sample_predictive_distribution()) + tuple(self.weights.sample_predictive_distribution())

    def approx_predictive_ll(self, Arow, Acol, Wrow, Wcol, M=100):
        # networks.py:60-67
        return (self.adjacency.approx_predictive_ll(Arow, Acol, M=M) +
                self.weights.approx_predictive_ll(Arow, Acol, Wrow, Wcol, M=M))

    # -- one Gibbs sweep (networks.py:69-71) -----------------------------------------------------------------------
    def resample(self, data, beta=1.0):
        """Adjacency half, then weight half; beta < 1 tempers both likelihoods (AIS over whole networks)."""
        d = self._device_data(data)
        kw = {} if beta == 1.0 else {"beta": beta}
        self.adjacency.resample(self._adjacency_arg(d, data), **kw)
        self.weights.resample(d, **kw)
        # each half was enqueued stream-ordered on its own model: finishing them together lets the engine overlap
        # whatever either still has to do (gg_sync_many)
        models = [m for m in (getattr(half, "_model", None) for half in self._halves) if m is not None]
        if len(models) > 1:
            _abi.sync_many(models)

    def meanfieldupdate(self, expected_suff_stats):
        pass   # a stub in the reference as well (networks.py:73-75)


class GaussianBernoulliNetwork(FactorizedNetworkDistribution):
    """networks.py:78-88: Bernoulli(p) edges with N(mu, sigma) weights, nothing to learn."""

    def __init__(self, N, B, p=0.5, mu=None, sigma=None):
        gauss = {"B": B, "mu": np.zeros(B) if mu is None else mu, "sigma": np.eye(B) if sigma is None else sigma}
        super(GaussianBernoulliNetwork, self).__init__(N, adjacency.BernoulliAdjacencyDistribution, {"p": p},
                                                       weights.FixedGaussianWeightDistribution, gauss)
