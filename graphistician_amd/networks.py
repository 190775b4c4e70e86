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


class FactorizedNetworkDistribution(NetworkDistribution, GibbsSampling):
    """networks.py:9-75."""

    def __init__(self, N, adjacency_class, adjacency_kwargs, weight_class, weight_kwargs):
        super(FactorizedNetworkDistribution, self).__init__()
        self.N = N
        assert issubclass(adjacency_class, AdjacencyDistribution)
        self._adjacency = adjacency_class(N, **adjacency_kwargs)
        assert issubclass(weight_class, WeightDistribution)
        self._weights = weight_class(N, **weight_kwargs)

    @property
    def adjacency(self):
        return self._adjacency

    @property
    def weights(self):
        return self._weights

    def _device_data(self, data):
        if isinstance(data, _abi.DeviceData):   # already resident
            return data
        A, W = data
        dev = getattr(self._weights, "_device", getattr(self._adjacency, "_device", 0))
        return as_device_data(A, W, device=dev)

    def log_likelihood(self, data):
        # networks.py:28-31
        d = self._device_data(data)
        if all(hasattr(part, "_engine") for part in (self.adjacency, self.weights)):
            # both halves on the engine: evaluated concurrently (gg_log_likelihood_many)
            assert d.N == self.N
            return sum(_abi.log_likelihood_many([self.adjacency._engine(), self.weights._engine()], d))
        lls = self.adjacency.log_likelihood(d if hasattr(self.adjacency, "_engine") else data[0])
        lls += self.weights.log_likelihood(d)
        return lls

    def log_prior(self):
        # networks.py:33-37
        lp = 0
        lp += self.adjacency.log_prior()
        lp += self.weights.log_prior()
        return lp

    def rvs(self, *args, **kwargs):
        # networks.py:39-42
        A = self.adjacency.rvs()
        W = self.weights.rvs()
        return A, W

    def sample_predictive_parameters(self):
        # networks.py:44-50 (the reference asks the adjacency twice; the weight parameters are what is meant)
        Aprms = self.adjacency.sample_predictive_parameters()
        Wprms = self.weights.sample_predictive_parameters()
        return Aprms, Wprms

    def sample_predictive_distribution(self):
        # networks.py:52-58
        Arow, Acol = self.adjacency.sample_predictive_distribution()
        Wrow, Wcol = self.weights.sample_predictive_distribution()
        return Arow, Acol, Wrow, Wcol

    def approx_predictive_ll(self, Arow, Acol, Wrow, Wcol, M=100):
        # networks.py:60-67
        pll = 0
        pll += self.adjacency.approx_predictive_ll(Arow, Acol, M=M)
        pll += self.weights.approx_predictive_ll(Arow, Acol, Wrow, Wcol, M=M)
        return pll

    def resample(self, data, beta=1.0):
        # networks.py:69-71; beta < 1 tempers both halves' likelihoods (AIS over whole networks)
        d = self._device_data(data)
        if beta == 1.0:
            self.adjacency.resample(d if hasattr(self.adjacency, "_engine") else data[0])
            self.weights.resample(d)
        else:
            self.adjacency.resample(d if hasattr(self.adjacency, "_engine") else data[0], beta=beta)
            self.weights.resample(d, beta=beta)
        # both halves are enqueued stream-ordered on their own models; finishing them together lets the
        # engine overlap whatever it still has to do for either (gg_sync_many)
        engines = [getattr(part, "_model", None) for part in (self.adjacency, self.weights)]
        engines = [e for e in engines if e is not None]
        if len(engines) > 1:
            _abi.sync_many(engines)

    def meanfieldupdate(self, expected_suff_stats):
        # networks.py:73-75 (a stub in the reference as well)
        pass


class GaussianBernoulliNetwork(FactorizedNetworkDistribution):
    """networks.py:78-88."""

    def __init__(self, N, B, p=0.5, mu=None, sigma=None):
        if mu is None:
            mu = np.zeros(B)
        if sigma is None:
            sigma = np.eye(B)
        super(GaussianBernoulliNetwork, self).__init__(
            N,
            adjacency.BernoulliAdjacencyDistribution, {"p": p},
            weights.FixedGaussianWeightDistribution, {"B": B, "mu": mu, "sigma": sigma})
