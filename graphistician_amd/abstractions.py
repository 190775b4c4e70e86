"""Abstract bases with the reference's interface (graphistician/abstractions.py).

Same names, argument meaning and error behaviour (bare asserts for shapes,
NotImplementedError for abstract members).  Python-2 tuple parameters
`def f(self, (A,W))` become a single `data` argument unpacked inside.  The O(N^2)
likelihood sweeps are not implemented here: subclasses route them to the HIP
engine; what stays on the host is what the reference itself keeps in O(N) or
O(K^2) numpy (predictive helpers, priors, rvs).
"""
import abc

import numpy as np
from scipy.special import logsumexp


class Distribution(object):
    """pybasicbayes.abstractions.Distribution conventions:
    rvs(size=[]), log_likelihood(x)."""
    __metaclass__ = abc.ABCMeta

    def rvs(self, size=[]):
        raise NotImplementedError

    def log_likelihood(self, x):
        raise NotImplementedError


class GibbsSampling(object):
    """pybasicbayes.abstractions.GibbsSampling: resample(data=[]) mutates in place, returns None."""

    def resample(self, data=[]):
        raise NotImplementedError


class NetworkDistribution(Distribution):
    """abstractions.py:9-72."""

    @property
    def adjacency(self):
        raise NotImplementedError

    @property
    def weights(self):
        raise NotImplementedError

    def log_prior(self):
        return np.nan

    def log_probability(self, data):
        # abstractions.py:39-40
        return self.log_likelihood(data) + self.log_prior()

    def initialize_hypers(self, A, W):
        # abstractions.py:45-47
        self.adjacency.initialize_hypers(A)
        self.weights.initialize_hypers(W)

    def initialize_from_prior(self):
        # abstractions.py:49-51
        self.adjacency.initialize_from_prior()
        self.weights.initialize_from_prior()

    def sample_predictive_parameters(self):
        raise NotImplementedError

    def sample_predictive_distribution(self):
        raise NotImplementedError

    def approx_predictive_ll(self, Arow, Acol, Wrow, Wcol, M=100):
        raise NotImplementedError()


class AdjacencyDistribution(Distribution):
    """abstractions.py:75-182: a distribution over adjacency matrices exposing P."""

    def __init__(self, N):
        self.N = N

    @property
    def P(self):
        return np.nan

    @property
    def safe_P(self):
        # abstractions.py:92-94
        return np.clip(self.P, 1e-64, 1 - 1e-64)

    @property
    def is_deterministic(self):
        P = self.P
        return np.all(np.isclose(P, 0) | np.isclose(P, 1))

    def log_prior(self):
        return np.nan

    def log_likelihood(self, A):
        """abstractions.py:111-115.  Subclasses with structured P (SBM, constant p)
        evaluate this on the device from K x K count tables."""
        raise NotImplementedError

    def log_probability(self, A):
        return self.log_likelihood(A) + self.log_prior()

    def rvs(self, size=[]):
        # abstractions.py:120-123
        P = self.P
        return np.random.rand(*P.shape) < P

    def initialize_from_prior(self):
        raise NotImplementedError

    def initialize_hypers(self, A):
        pass

    def sample_predictive_parameters(self):
        raise NotImplementedError

    def sample_predictive_distribution(self):
        # abstractions.py:141-155
        N = self.N
        Prow, Pcol = self.sample_predictive_parameters()
        assert Prow[-1] == Pcol[-1]
        Arow = np.random.rand(N + 1) < Prow
        Acol = np.random.rand(N + 1) < Pcol
        Acol[-1] = Arow[-1]
        return Arow, Acol

    def approx_predictive_ll(self, Arow, Acol, M=100):
        # abstractions.py:157-182 (host: 2N terms x M draws)
        N = self.N
        assert Arow.shape == Acol.shape == (N + 1,)
        Acol = Acol[:-1]
        lps = np.zeros(M)
        for m in range(M):
            Prow, Pcol = self.sample_predictive_parameters()
            Prow = np.clip(Prow, 1e-64, 1 - 1e-64)
            Pcol = np.clip(Pcol, 1e-64, 1 - 1e-64)[:-1]
            lps[m] += (Arow * np.log(Prow) + (1 - Arow) * np.log(1 - Prow)).sum()
            lps[m] += (Acol * np.log(Pcol) + (1 - Acol) * np.log(1 - Pcol)).sum()
        return -np.log(M) + logsumexp(lps)


class WeightDistribution(Distribution):
    """abstractions.py:185-234."""

    def __init__(self, N):
        self.N = N

    def initialize_from_prior(self):
        raise NotImplementedError

    def initialize_hypers(self, W):
        pass

    def sample_predictive_parameters(self):
        raise NotImplementedError

    def sample_predictive_distribution(self):
        raise NotImplementedError

    def approx_predictive_ll(self, Arow, Acol, Wrow, Wcol, M=100):
        raise NotImplementedError


class GaussianWeightDistribution(WeightDistribution):
    """abstractions.py:237-357."""

    def __init__(self, N, B=1):
        super(GaussianWeightDistribution, self).__init__(N)
        self.B = B

    @property
    def Mu(self):
        return np.nan

    @property
    def Sigma(self):
        return np.nan

    def log_likelihood(self, data):
        """abstractions.py:257-274 (per-edge multivariate normal log-densities);
        subclasses evaluate it on the device from sufficient statistics."""
        raise NotImplementedError

    def rvs(self, size=[]):
        # abstractions.py:276-287, vectorised per distinct (Mu, Sigma) via Cholesky factors
        N, B = self.N, self.B
        Mu, Sig = self.Mu, self.Sigma
        L = np.linalg.cholesky(Sig.reshape(-1, B, B)).reshape(N, N, B, B)
        z = np.random.randn(N, N, B)
        return Mu + np.einsum("mnij,mnj->mni", L, z)

    def sample_predictive_distribution(self):
        # abstractions.py:289-311
        N = self.N
        Murow, Mucol, Lrow, Lcol = self.sample_predictive_parameters()
        assert np.all(Murow[-1] == Mucol[-1])
        assert np.all(Lrow[-1] == Lcol[-1])
        Wrow = np.zeros((N + 1, self.B))
        Wcol = np.zeros((N + 1, self.B))
        for n in range(N + 1):
            Wrow[n] = np.random.multivariate_normal(Murow[n], Lrow[n].dot(Lrow[n].T))
            if n < N:
                Wcol[n] = np.random.multivariate_normal(Mucol[n], Lcol[n].dot(Lcol[n].T))
            else:
                Wcol[n] = Wrow[n]
        return Wrow, Wcol

    def approx_predictive_ll(self, Arow, Acol, Wrow, Wcol, M=100):
        # abstractions.py:313-357 (host: 2N terms x M draws)
        import scipy.linalg
        N, B = self.N, self.B
        assert Arow.shape == Acol.shape == (N + 1,)
        lps = np.zeros(M)
        for m in range(M):
            Murow, Mucol, Lrow, Lcol = self.sample_predictive_parameters()
            for n in range(N + 1):
                if Arow[n]:
                    xs = scipy.linalg.solve_triangular(Lrow[n], Wrow[n] - Murow[n], lower=True)
                    lps[m] += -0.5 * xs.dot(xs) - B / 2. * np.log(2 * np.pi) - np.log(Lrow[n].diagonal()).sum()
                if n < N and Acol[n]:
                    xs = scipy.linalg.solve_triangular(Lcol[n], Wcol[n] - Mucol[n], lower=True)
                    lps[m] += -0.5 * xs.dot(xs) - B / 2. * np.log(2 * np.pi) - np.log(Lcol[n].diagonal()).sum()
        return -np.log(M) + logsumexp(lps)
