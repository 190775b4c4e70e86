"""Logistic eigenmodel (Hoff 2008) with Polya-gamma Gibbs sampling on the MI355X engine.

Algorithm source: graphistician/eigenmodel.py — `_EigenmodelBase` (:14-133),
`_LogisticEigenmodelBase` (:795-820), `_GibbsLogisticEigenmodel` (:822-965).  The
reference class is not an `AdjacencyDistribution` and its module does not import
(undefined base classes at :1319-1320); this adapter keeps its
constructor signature and attribute names (`F`, `lmbda`, `mu_0`, `mu_mu_0`, `Mu`,
`P`) and conforms to the adjacency contract so that it composes with
`FactorizedNetworkDistribution`.
"""
import numpy as np
from scipy.stats import norm

from . import _abi
from .abstractions import AdjacencyDistribution, GibbsSampling
from .engine import EngineBacked


class LogisticEigenmodel(AdjacencyDistribution, GibbsSampling, EngineBacked):
    """A_ij ~ Bern(logistic(mu_0 + f_i^T Lambda f_j)); f_i ~ N(0, sigma_F I), lambda ~ N(mu_lmbda, sigma_lmbda I),
    mu_0 ~ N(mu_mu_0, sigma_mu0)."""

    _kind = _abi.LOGISTIC_EIG

    _fields = {"F": (_abi.ST_F, np.float64, lambda s: (s.N, s.D)),
               "lmbda": (_abi.ST_LMBDA, np.float64, lambda s: (s.D,)),
               "_mu0": (_abi.ST_MU0, np.float64, lambda s: (1,)),
               "mu_lmbda": (_abi.ST_MU_LMBDA, np.float64, lambda s: (s.D,))}

    def __init__(self, N, D, p=0.5, sigma_mu0=1.0, sigma_F=1.0, lmbda=None, mu_lmbda=0, sigma_lmbda=1.0,
                 device=0, seed=None):
        super(LogisticEigenmodel, self).__init__(N)
        self.D = D
        # eigenmodel.py:53-57
        assert p >= 0 and p <= 1.0, "p must be a baseline probability of connection"
        self.p = p
        self.mu_mu_0 = norm(0, 1).ppf(p)
        assert sigma_mu0 >= 0 and sigma_F >= 0 and sigma_lmbda >= 0, "q,r,s must be positive variances"
        self.sigma_mu0 = sigma_mu0
        self.sigma_F = sigma_F
        self.sigma_lmbda = sigma_lmbda
        self.lmbda_given = lmbda is not None
        self._init_engine(device=device, seed=seed, kind=self._kind, N=N, D=D, K=1, B=1,
                          sigma_mu0=sigma_mu0, sigma_F=sigma_F, sigma_lmbda=sigma_lmbda, mu_mu_0=self.mu_mu_0,
                          lmbda_given=1 if self.lmbda_given else 0)
        if self.lmbda_given:
            assert lmbda.shape == (self.D,)
            self.mu_lmbda = np.zeros(D)
            self.lmbda = lmbda
        else:
            assert np.isscalar(mu_lmbda) or mu_lmbda.shape == (self.D,)
            self.mu_lmbda = mu_lmbda * np.ones(self.D)
            self.lmbda = self.mu_lmbda.copy()
        # eigenmodel.py:830-834: the Gibbs class starts from a prior draw (its data-free resample())
        self.initialize_from_prior()

    F = property(lambda s: s._hget("F"), lambda s, v: s._hset("F", v))
    lmbda = property(lambda s: s._hget("lmbda"), lambda s, v: s._hset("lmbda", v))
    mu_lmbda = property(lambda s: s._hget("mu_lmbda"), lambda s, v: s._hset("mu_lmbda", v))

    @property
    def mu_0(self):
        return float(self._hget("_mu0")[0])

    @mu_0.setter
    def mu_0(self, v):
        self._hset("_mu0", np.array([v], dtype=np.float64))

    def initialize_from_prior(self):
        self.F = np.sqrt(self.sigma_F) * np.random.randn(self.N, self.D)
        self.mu_0 = self.mu_mu_0 + np.sqrt(self.sigma_mu0) * np.random.randn()
        if not self.lmbda_given:
            self.lmbda = self.mu_lmbda + np.sqrt(self.sigma_lmbda) * np.random.randn(self.D)

    def initialize_hypers(self, A):
        pass

    @property
    def Mu(self):
        # eigenmodel.py:75-83
        Mu = self.mu_0 + (self.F * self.lmbda[None, :]).dot(self.F.T)
        np.fill_diagonal(Mu, self.mu_0)
        return Mu

    @property
    def P(self):
        # eigenmodel.py:803-811
        P = 1.0 / (1.0 + np.exp(-self.Mu))
        return np.clip(P, 1e-16, 1 - 1e-16)

    def kappa(self, A):
        # eigenmodel.py:813-820
        return A - 0.5

    def log_prior(self):
        # eigenmodel.py:92-108
        lp = 0
        lp += -0.5 * (self.F * self.F / self.sigma_F).sum()
        lp += -0.5 * (self.mu_0 - self.mu_mu_0) ** 2 / self.sigma_mu0
        if not self.lmbda_given:
            lp += -0.5 * (self.lmbda * self.lmbda / self.sigma_lmbda).sum()
        return lp

    def log_likelihood(self, A):
        """eigenmodel.py:110-129, evaluated tile by tile on the device (P is never materialised).  NaN entries of A are
        missing data: their terms are zero (:125-127)."""
        d = self._data(A, missing_ok=True)
        assert d.N == self.N
        return self._engine().log_likelihood(d)

    def rvs(self, size=[]):
        # eigenmodel.py:134-148
        return np.random.rand(self.N, self.N) < self.P

    def sample_predictive_parameters(self):
        f = np.sqrt(self.sigma_F) * np.random.randn(self.D)
        Fe = np.vstack((self.F, f[None, :]))
        Mu = self.mu_0 + (Fe * self.lmbda[None, :]).dot(Fe.T)
        np.fill_diagonal(Mu, self.mu_0)
        P = np.clip(1.0 / (1.0 + np.exp(-Mu)), 1e-16, 1 - 1e-16)
        return P[-1, :], P[:, -1]

    def _networks(self, A):
        """One adjacency matrix, or the reference's list of observed graphs (`resample(networks=[])`, eigenmodel.py:836-845:
        objects with an `.A`, or the matrices themselves).  NaN entries are unobserved."""
        if isinstance(A, list):
            return [self._data(getattr(a, "A", a), missing_ok=True) for a in A]
        return [self._data(getattr(A, "A", A), missing_ok=True)]

    def _run(self, A, parts, beta=1.0):
        ds = self._networks(A)
        assert len(ds) >= 1 and all(d.N == self.N for d in ds)
        eng = self._engine()
        if len(ds) > 1:
            eng.resample_part_many(ds, parts, beta)     # auxiliary variables per graph, conditionals from the summed statistics
        elif parts == _abi.PART_ALL:
            eng.resample(ds[0], 1, beta)
        else:
            eng.resample_part(ds[0], parts, beta)
        self._mark_stale("F", "lmbda", "_mu0")

    def resample(self, A, beta=1.0):
        """eigenmodel.py:836-857: Omega -> F -> mu_0 -> lambda, for one graph or a list of them.  beta < 1 tempers the
        likelihood for AIS: BUILD-DEFINED (the reference leaves it a TODO, graph_models.py:478) - omega ~ PG(beta, psi),
        kappa = beta (A - 1/2); the PG(beta, .) draw for a real shape is the truncated gamma-sum representation
        (csrc/spec/gg_pg.h), approximate by construction.  Entries of A that are NaN are missing data: the reference
        zeroes their log-likelihood terms (:121-127) and its Gibbs updates would turn NaN on them; here they are left out
        of every conditional (no auxiliary variable, no term in F, mu_0, lambda)."""
        self._run(A, _abi.PART_ALL, beta)

    def _resample_Omega(self, A, beta=1.0):
        self._run(A, _abi.PART_OMEGA, beta)

    def _resample_F(self, A, beta=1.0):
        self._run(A, _abi.PART_F, beta)

    def _resample_mu_0(self, A, beta=1.0):
        self._run(A, _abi.PART_MU0, beta)

    def _resample_lmbda(self, A, beta=1.0):
        self._run(A, _abi.PART_LMBDA, beta)


class ProbitEigenmodel(LogisticEigenmodel):
    """A_ij ~ Bern(Phi(mu_0 + f_i^T Lambda f_j)) with truncated-normal auxiliaries Z (Albert & Chib):
    graphistician/eigenmodel.py `_ProbitEigenmodelBase` (:223-237), `_GibbsProbitEigenmodel` (:239-379).
    Same state and constructor as the logistic model; the engine kind differs."""

    _kind = _abi.PROBIT_EIG

    @property
    def P(self):
        # eigenmodel.py:228-237
        from scipy.special import erf
        P = 0.5 * (1 + erf(self.Mu / np.sqrt(2.0)))
        return np.clip(P, 1e-16, 1 - 1e-16)

    def sample_predictive_parameters(self):
        from scipy.special import erf
        f = np.sqrt(self.sigma_F) * np.random.randn(self.D)
        Fe = np.vstack((self.F, f[None, :]))
        Mu = self.mu_0 + (Fe * self.lmbda[None, :]).dot(Fe.T)
        np.fill_diagonal(Mu, self.mu_0)
        P = np.clip(0.5 * (1 + erf(Mu / np.sqrt(2.0))), 1e-16, 1 - 1e-16)
        return P[-1, :], P[:, -1]

    def resample(self, A):
        """eigenmodel.py:246-272: Z -> F -> mu_0 -> lambda."""
        self._run(A, _abi.PART_ALL)

    def _resample_Z(self, A):
        self._run(A, _abi.PART_OMEGA)
