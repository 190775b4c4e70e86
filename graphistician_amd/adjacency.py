"""Adjacency distributions with the interface of graphistician/adjacency.py.

resample / log_likelihood of the Bernoulli, Beta-Bernoulli and SBM models run on
the MI355X engine (HIP kernels behind libgraphgibbs.so); construction, priors,
`P`, `rvs` and the predictive helpers are the same small host-side numpy the
reference uses.  Constructor signatures, attribute names and assertion
behaviour follow the reference (file:line cited per member).
"""
import numpy as np
from scipy.special import logsumexp
from scipy.stats import beta as _beta, dirichlet as _dirichlet

from . import _abi
from .abstractions import AdjacencyDistribution, GibbsSampling
from .engine import EngineBacked


def logistic(x):
    """internals/utils.py:5-6."""
    return 1.0 / (1.0 + np.exp(-x))


class FixedAdjacencyDistribution(AdjacencyDistribution, GibbsSampling):
    """adjacency.py:13-41: a fixed 0/1 matrix (host-only, no parameters)."""

    def __init__(self, N, A):
        super(FixedAdjacencyDistribution, self).__init__(N)
        assert A.shape == (N, N)
        assert np.all((A == 1) | (A == 0))
        self._P = A

    @property
    def P(self):
        return self._P

    def initialize_from_prior(self):
        pass

    def initialize_hypers(self, A):
        pass

    def log_prior(self):
        return 0

    def log_likelihood(self, A):
        # abstractions.py:111-115 with P in {0,1}: 0 when A equals P, -inf otherwise
        assert A.shape == (self.N, self.N)
        return 0.0 if np.array_equal(np.asarray(A) != 0, np.asarray(self._P) != 0) else -np.inf

    def resample(self, data=[]):
        pass

    def sample_predictive_parameters(self):
        raise Exception("Cannot sample parameters for FixedAdjacencyDistribution")


class CompleteAdjacencyDistribution(FixedAdjacencyDistribution):
    """adjacency.py:43-54."""

    def __init__(self, N):
        super(CompleteAdjacencyDistribution, self).__init__(N, np.ones((N, N)))

    def sample_predictive_parameters(self):
        return np.ones(self.N + 1), np.ones(self.N + 1)

    def log_likelihood(self, A):
        return 0 if np.all(A == 1) else -np.inf


class EmptyAdjacencyDistribution(FixedAdjacencyDistribution):
    """adjacency.py:57-62."""

    def __init__(self, N):
        super(EmptyAdjacencyDistribution, self).__init__(N, np.zeros((N, N)))

    def sample_predictive_parameters(self):
        return np.zeros(self.N + 1), np.zeros(self.N + 1)


def _scalar_field(name):
    def g(self):
        return float(self._hget(name).ravel()[0])

    def s(self, v):
        self._hset(name, np.array([v], dtype=np.float64))
    return property(g, s)


class BernoulliAdjacencyDistribution(AdjacencyDistribution, GibbsSampling, EngineBacked):
    """adjacency.py:65-107: Bernoulli edges with fixed probability p (p_self on the diagonal)."""

    _fields = {"p": (_abi.ST_P, np.float64, lambda s: (1, 1)),
               "p_self": (_abi.ST_P_SELF, np.float64, lambda s: (1,))}

    def __init__(self, N, p=None, P=None, p_self=None, device=0, seed=None):
        super(BernoulliAdjacencyDistribution, self).__init__(N)
        self._full_P = None
        if P is not None:
            assert P.shape == (N, N)
            assert np.all(P >= 0) and np.all(P <= 1)
            self._full_P = P.copy()
        else:
            assert p is not None
            assert p > 0 and p < 1
            self._init_engine(device=device, seed=seed, kind=_abi.BERN_ADJ, N=N, K=1, B=1,
                              allow_self_connections=1, special_case_self_conns=1)
            self.p = p
            self.p_self = p_self if p_self else p

    p = _scalar_field("p")
    p_self = _scalar_field("p_self")

    @property
    def P(self):
        if self._full_P is not None:
            return self._full_P
        P = self.p * np.ones((self.N, self.N))
        np.fill_diagonal(P, self.p_self)
        return P

    def log_prior(self):
        return 0

    def log_likelihood(self, A):
        if self._full_P is not None:
            raise NotImplementedError("dense per-edge P: the engine evaluates structured P only (scalar p, SBM)")
        d = self._data(A)
        assert (d.N, d.N) == (self.N, self.N)
        return self._engine().log_likelihood(d)

    def initialize_from_prior(self):
        pass

    def initialize_hypers(self, A):
        pass

    def resample(self, data=[]):
        pass

    def sample_predictive_parameters(self):
        p_row, p_col = self.p * np.ones(self.N + 1), self.p * np.ones(self.N + 1)
        p_row[-1] = self.p_self
        p_col[-1] = self.p_self
        return p_row, p_col


class BetaBernoulliAdjacencyDistribution(AdjacencyDistribution, GibbsSampling, EngineBacked):
    """adjacency.py:110-188."""

    _fields = {"p": (_abi.ST_P, np.float64, lambda s: (1, 1)),
               "p_self": (_abi.ST_P_SELF, np.float64, lambda s: (1,))}

    def __init__(self, N, tau1=1.0, tau0=1.0, tau1_self=None, tau0_self=None, device=0, seed=None):
        super(BetaBernoulliAdjacencyDistribution, self).__init__(N)
        assert tau1 > 0 and tau0 > 0
        self.tau1 = tau1
        self.tau0 = tau0
        self.self_connection = tau1_self is not None and tau0_self is not None
        self._init_engine(device=device, seed=seed, kind=_abi.BB_ADJ, N=N, K=1, B=1, tau1=tau1, tau0=tau0,
                          allow_self_connections=1, special_case_self_conns=1 if self.self_connection else 0)
        self.p = np.random.beta(tau1, tau0)
        if self.self_connection:
            self.tau1_self = tau1_self
            self.tau0_self = tau0_self
            self.p_self = np.random.beta(tau1_self, tau0_self)
        else:
            self._hset("p_self", np.array([self.p]))

    p = _scalar_field("p")
    p_self = _scalar_field("p_self")

    @property
    def P(self):
        # adjacency.py:128-133
        P = self.p * np.ones((self.N, self.N))
        if self.self_connection:
            np.fill_diagonal(P, self.p_self)
        return P

    def initialize_from_prior(self):
        self.p = np.random.beta(self.tau1, self.tau0)
        if self.self_connection:
            self.p_self = np.random.beta(self.tau1_self, self.tau0_self)

    def initialize_hypers(self, A):
        pass

    def rvs(self, size=[]):
        return np.random.rand(self.N, self.N) < self.P

    def log_prior(self):
        lp = _beta(self.tau1, self.tau0).logpdf(self.p)
        if self.self_connection:
            lp += _beta(self.tau1_self, self.tau0_self).logpdf(self.p_self)
        return lp

    def log_likelihood(self, A):
        d = self._data(A)
        assert d.N == self.N
        return self._engine().log_likelihood(d)

    def resample(self, A, beta=1.0):
        """adjacency.py:153-181 (the self-probability posterior uses tau1, tau0 as the
        reference does at :175).  beta < 1 tempers the likelihood (graph_models.py:178-182)."""
        d = self._data(A)
        assert d.N == self.N
        self._engine().resample_async(d, 1, beta)
        self._mark_stale("p", "p_self")

    def sample_predictive_parameters(self):
        p_row, p_col = self.p * np.ones(self.N + 1), self.p * np.ones(self.N + 1)
        if self.self_connection:
            p_row[-1] = p_col[-1] = self.p_self
        return p_row, p_col

    def approx_predictive_ll(self, Arow, Acol, M=100):
        """abstractions.py:157-182 on the engine: the predictive parameters of this model are not random, so the M-draw average
        is the one log-likelihood of the held-out row and column (gg_predictive_ll)."""
        assert Arow.shape == Acol.shape == (self.N + 1,)
        return float(self._engine().predictive_ll(Arow, Acol, K=1)[0])


class LatentDistanceAdjacencyDistribution(AdjacencyDistribution, GibbsSampling, EngineBacked):
    """adjacency.py:191-437.

    l_n ~ N(0, sigma I);  A_{n',n} ~ Bern(logistic(-||l_{n'} - l_n||^2 + mu_0 + mu_self [n' == n])).
    resample = HMC on L (10 leapfrog steps, adaptive step size), HMC on mu_0, HMC on mu_self (:360-421), all on the
    engine.  The reference differentiates with autograd and integrates with hips.inference.hmc; neither is available,
    so the gradient is hand-derived and the HMC transition restated (csrc/gg_latent.hip and the test oracle):
    parity unpinned, checked against the oracle's same definition.  beta < 1 tempers the likelihood (AIS)."""

    _fields = {"L": (_abi.ST_F, np.float64, lambda s: (s.N, s.dim)),
               "_mu0": (_abi.ST_MU0, np.float64, lambda s: (1,)),
               "_aux": (_abi.ST_LD_AUX, np.float64, lambda s: (3,))}

    def __init__(self, N, dim=2, sigma=2.0, mu0=0.0, mu_self=0.0, device=0, seed=None):
        super(LatentDistanceAdjacencyDistribution, self).__init__(N)
        self.dim = dim
        self.sigma = sigma
        self._init_engine(device=device, seed=seed, kind=_abi.LATENT_DIST, N=N, D=dim, K=1, B=1, sigma_F=sigma)
        self.mu_0 = mu0
        # adjacency.py:201-206: mu_self, HMC step size 0.01, running acceptance rate 0.9
        self._hset("_aux", np.array([mu_self, 0.01, 0.9]))
        self.L = np.sqrt(self.sigma) * np.random.randn(N, dim)

    L = property(lambda s: s._hget("L"), lambda s, v: s._hset("L", v))

    @property
    def mu_0(self):
        return float(self._hget("_mu0")[0])

    @mu_0.setter
    def mu_0(self, v):
        self._hset("_mu0", np.array([v], dtype=np.float64))

    @property
    def mu_self(self):
        return float(self._hget("_aux")[0])

    @mu_self.setter
    def mu_self(self, v):
        a = self._hget("_aux").copy()
        a[0] = v
        self._hset("_aux", a)

    _L_step_sz = property(lambda s: float(s._hget("_aux")[1]))
    _L_accept_rate = property(lambda s: float(s._hget("_aux")[2]))

    @property
    def D(self):
        # adjacency.py:208-214
        Mu = -((self.L[:, None, :] - self.L[None, :, :]) ** 2).sum(2)
        Mu += self.mu_0
        Mu += self.mu_self * np.eye(self.N)
        return Mu

    @property
    def P(self):
        return logistic(self.D)

    def initialize_from_prior(self):
        # adjacency.py:221-224
        self.mu_0 = np.random.randn()
        self.mu_self = np.random.randn()
        self.L = np.sqrt(self.sigma) * np.random.randn(self.N, self.dim)

    def initialize_hypers(self, A):
        pass

    def log_prior(self):
        # adjacency.py:229-242
        from scipy.stats import norm
        lp = norm.logpdf(self.L, 0, np.sqrt(self.sigma)).sum()
        lp += norm.logpdf(self.mu_0, 0, 1)
        lp += norm.logpdf(self.mu_self, 0, 1)
        return lp

    def log_likelihood(self, A):
        """abstractions.py:111-115 on the engine (stable logit form; P is never materialised)."""
        d = self._data(A)
        assert d.N == self.N
        return self._engine().log_likelihood(d)

    def rvs(self, size=[]):
        return np.random.rand(self.N, self.N) < self.P

    def sample_predictive_parameters(self):
        # adjacency.py:291-303
        Lext = np.vstack((self.L, np.sqrt(self.sigma) * np.random.randn(1, self.dim)))
        D = -((Lext[:, None, :] - Lext[None, :, :]) ** 2).sum(2)
        D += self.mu_0
        D += self.mu_self * np.eye(self.N + 1)
        P = logistic(D)
        return P[-1, :], P[:, -1]

    def _run(self, A, parts, beta=1.0):
        d = self._data(A)
        assert d.N == self.N
        eng = self._engine()
        if parts == _abi.PART_ALL:
            eng.resample(d, 1, beta)
        else:
            eng.resample_part(d, parts, beta)
        self._mark_stale("L", "_mu0", "_aux")

    def resample(self, A, beta=1.0):
        """adjacency.py:360-372: L, then mu_0, then mu_self."""
        self._run(A, _abi.PART_ALL, beta)

    def _resample_L(self, A, beta=1.0):
        self._run(A, _abi.PART_F, beta)

    def _resample_mu_0(self, A, beta=1.0):
        self._run(A, _abi.PART_MU0, beta)

    def _resample_mu_self(self, A, beta=1.0):
        self._run(A, _abi.PART_LMBDA, beta)


class SBMAdjacencyDistribution(AdjacencyDistribution, GibbsSampling, EngineBacked):
    """adjacency.py:440-702.

    C blocks; m[c] block probabilities; p[c,c'] connection probability from block c to c'.
    Hyper-parameters pi (Dirichlet over m), tau0/tau1 (Beta over p).
    """

    _fields = {"c": (_abi.ST_C, np.int32, lambda s: (s.N,)),
               "m": (_abi.ST_M, np.float64, lambda s: (s.C,)),
               "pi": (_abi.ST_PI, np.float64, lambda s: (s.C,)),
               "p": (_abi.ST_P, np.float64, lambda s: (s.C, s.C)),
               "p_self": (_abi.ST_P_SELF, np.float64, lambda s: (1,))}

    def __init__(self, N, C=2, c=None, m=None, pi=1.0, p=None, tau0=1.0, tau1=1.0,
                 allow_self_connections=True, special_case_self_conns=True, device=0, seed=None):
        super(SBMAdjacencyDistribution, self).__init__(N)
        assert isinstance(C, (int, np.integer)) and C >= 1, "C must be a positive integer number of blocks"
        self.C = int(C)
        if isinstance(pi, (int, float)):
            pi = pi * np.ones(C)
        else:
            assert isinstance(pi, np.ndarray) and pi.shape == (C,), "pi must be a sclar or a C-vector"
        self.tau0 = tau0
        self.tau1 = tau1
        self.allow_self_connections = allow_self_connections
        self.special_case_self_conns = special_case_self_conns
        self._init_engine(device=device, seed=seed, kind=_abi.SBM_ADJ, N=N, K=self.C, B=1, tau1=tau1, tau0=tau0,
                          allow_self_connections=1 if allow_self_connections else 0,
                          special_case_self_conns=1 if special_case_self_conns else 0)
        self.pi = pi
        if m is not None:
            assert isinstance(m, np.ndarray) and m.shape == (C,) and np.allclose(m.sum(), 1.0) and np.amin(m) >= 0.0, \
                "m must be a length C probability vector"
            self.m = m
        else:
            self.m = np.random.dirichlet(self.pi)
        if c is not None:
            assert isinstance(c, np.ndarray) and c.shape == (self.N,) and np.issubdtype(c.dtype, np.integer) \
                and np.amin(c) >= 0 and np.amax(c) <= self.C - 1, "c must be a length K-vector of block assignments"
            self.c = c.copy()
        else:
            self.c = np.random.choice(self.C, p=self.m, size=(self.N))
        if p is not None:
            if np.isscalar(p):
                assert p >= 0 and p <= 1, "p must be a probability"
                self.p = p * np.ones((C, C))
            else:
                assert isinstance(p, np.ndarray) and p.shape == (C, C) and np.amin(p) >= 0 and np.amax(p) <= 1.0, \
                    "p must be a CxC matrix of probabilities"
                self.p = p
        else:
            self.p = np.random.beta(self.tau1, self.tau0, size=(self.C, self.C))
        if special_case_self_conns:
            self.p_self = np.random.beta(self.tau1, self.tau0)
        else:
            self._hset("p_self", np.array([0.5]))

    c = property(lambda s: s._hget("c"), lambda s, v: s._hset("c", v))
    m = property(lambda s: s._hget("m"), lambda s, v: s._hset("m", v))
    pi = property(lambda s: s._hget("pi"), lambda s, v: s._hset("pi", v))
    p = property(lambda s: s._hget("p"), lambda s, v: s._hset("p", v))
    p_self = _scalar_field("p_self")

    @property
    def P(self):
        # adjacency.py:516-528 (host gather; small-N convenience, never used by the sweep)
        c = self.c
        P = self.p[np.ix_(c, c)]
        if self.special_case_self_conns:
            np.fill_diagonal(P, self.p_self)
        if not self.allow_self_connections:
            np.fill_diagonal(P, 0.0)
        return P

    def initialize_from_prior(self):
        # adjacency.py:530-533
        self.p = np.random.beta(self.tau1, self.tau0, size=(self.C, self.C))
        if self.special_case_self_conns:
            self.p_self = np.random.beta(self.tau1, self.tau0)

    def initialize_hypers(self, A):
        pass

    _device_prior = True      # gg_log_prior_many evaluates this class's log_prior from the engine's state

    def log_prior(self):
        """adjacency.py:538-555.  Once the model lives on the engine the value comes from there (one small launch instead of
        four state reads and three scipy calls: the reference's loops ask for log_probability once per sweep); a model that has
        not touched the engine yet uses the host formula below - the two agree to rounding (tests/test_gpu_edge_cases.py)."""
        if self._model is not None:
            return _abi.log_prior_many([self._engine()])[0]
        return self._log_prior_host()

    def _log_prior_host(self):
        # adjacency.py:538-555
        lp = 0
        lp += _dirichlet(self.pi).logpdf(self.m)
        lp += _beta(self.tau1 * np.ones((self.C, self.C)), self.tau0 * np.ones((self.C, self.C))).logpdf(self.p).sum()
        if self.special_case_self_conns:
            lp += _beta(self.tau1, self.tau0).logpdf(self.p_self)
        lp += (np.log(self.m)[self.c]).sum()
        return lp

    def log_likelihood(self, A):
        """abstractions.py:111-115 evaluated from the K x K edge-count tables on the device."""
        d = self._data(A)
        assert d.N == self.N
        return self._engine().log_likelihood(d)

    def rvs(self, size=[]):
        return np.random.rand(self.N, self.N) < self.P

    def sample_predictive_parameters(self):
        # adjacency.py:562-573
        cext = np.concatenate((self.c, [np.random.choice(self.C, p=self.m)]))
        P = self.p[np.ix_(cext, cext)]
        if not self.allow_self_connections:
            np.fill_diagonal(P, 0.0)
        return P[-1, :], P[:, -1]

    def approx_predictive_ll(self, Arow, Acol, M=100):
        """abstractions.py:157-182 on the engine.  A predictive draw (adjacency.py:562-573) is the held-out node's block
        c2 ~ Cat(m); the engine returns the log-likelihood of its row and column for every block (gg_predictive_ll: labels
        and parameters stay on the device), the M draws average over them."""
        assert Arow.shape == Acol.shape == (self.N + 1,)
        m = self.m
        ll_k = self._engine().predictive_ll(Arow, Acol, K=self.C)
        c2 = np.random.choice(self.C, p=m, size=M)
        return float(-np.log(M) + logsumexp(ll_k[c2]))

    def invariant_sbm_order(self):
        # adjacency.py:575-591
        blocksz = np.bincount(self.c, minlength=self.C)
        corder = np.argsort(blocksz)
        newc = np.zeros(self.N)
        for c in np.arange(self.C):
            newc[self.c == corder[c]] = c
        return np.argsort(-newc)

    # -- Gibbs (adjacency.py:615-702): on the device ------------------------------
    def _run(self, A, parts, beta=1.0):
        d = self._data(A) if A is not None else None
        assert d is None or d.N == self.N
        eng = self._engine()
        if parts == _abi.PART_ALL:
            eng.resample_async(d, 1, beta)   # stream-ordered: the next state read (or log_likelihood) waits for it
        else:
            eng.resample_part(d, parts, beta)   # (the engine advances its Philox sweep index: graphgibbs.h, gg_resample_part)
        self._mark_stale("c", "m", "p", "p_self")

    def resample(self, A, beta=1.0):
        """beta < 1: tempered sweep for AIS (likelihood^beta, graph_models.py:278-293,366-367)."""
        self._run(A, _abi.PART_ALL, beta)

    def resample_p(self, As, beta=1.0):
        """adjacency.py:620-648: the block-pair counts are summed over the graphs of the list, one Beta draw per pair."""
        assert len(As) >= 1
        if len(As) == 1:
            return self._run(As[0], _abi.PART_PARAMS, beta)
        ds = [self._data(A) for A in As]
        assert all(d.N == self.N for d in ds)
        self._engine().resample_part_many(ds, _abi.PART_PARAMS, beta)
        self._mark_stale("c", "m", "p", "p_self")

    def resample_c(self, A, beta=1.0):
        if self.C == 1:
            return
        self._run(A, _abi.PART_C, beta)

    def resample_m(self):
        self._run(None, _abi.PART_M)
