"""The legacy joint stochastic block model on the MI355X engine: ONE block assignment per node drives both the
adjacency matrix and the Gaussian weights.

Algorithm source: graphistician/stochastic_block_model.py — `_StochasticBlockModelBase` (:17-110), `_GibbsSBM`
(:202-309), `GaussianStochasticBlockModel` (:589-678).  That module does not import in the reference (it pulls in
`graphistician.old_abstractions`, which is not in the tree, :12); its Gibbs sweep is what the stale demos mean by
"Gaussian SBM" (SURVEY.md section 8f rank 4):

    resample: the weight model of every block pair from the pair's edges (:220-226)  ->  resample_p (:232-250)
              ->  resample_c with Bernoulli AND Gaussian terms, self connection under pair (c_n, c_n) (:252-302)
              ->  resample_m (:304-309)

The engine runs it as one component (kind GG_SBM_JOINT): the weight model's statistics already carry the edge counts,
the conditional gets one more coefficient per block pair (csrc/spec/gg_lp.h::gg_joint_table).  Constructor and
attribute names follow the reference class; `weight_hypers` takes the NIW prior of the pybasicbayes-style
`GaussianWeights(B, mu_0, Sigma_0, nu_0, kappa_0)` (internals/weights.py:62-86).
"""
import numpy as np

from . import _abi
from .abstractions import Distribution, GibbsSampling
from .engine import EngineBacked
from .weights import _pack_niw, _niw_prior_draw, _GaussianView


class GaussianStochasticBlockModel(Distribution, GibbsSampling, EngineBacked):
    _fields = {"c": (_abi.ST_C, np.int32, lambda s: (s.N,)),
               "m": (_abi.ST_M, np.float64, lambda s: (s.C,)),
               "pi": (_abi.ST_PI, np.float64, lambda s: (s.C,)),
               "p": (_abi.ST_P, np.float64, lambda s: (s.C, s.C)),
               "_mu": (_abi.ST_MU, np.float64, lambda s: (s.C, s.C, s.B)),
               "_sigma": (_abi.ST_SIGMA, np.float64, lambda s: (s.C, s.C, s.B, s.B)),
               "_niw": (_abi.ST_NIW, np.float64, lambda s: (s.B + s.B * s.B + 2,))}

    def __init__(self, N, B, C=1, c=None, m=None, pi=1.0, p=None, tau0=1.0, tau1=1.0, weight_hypers={},
                 allow_self_connections=True, device=0, seed=None):
        # stochastic_block_model.py:41-110
        self.N, self.B = N, B
        assert isinstance(C, (int, np.integer)) and C >= 1, "C must be a positive integer number of blocks"
        self.C = int(C)
        if isinstance(pi, (int, float)):
            pi = pi * np.ones(C)
        else:
            assert isinstance(pi, np.ndarray) and pi.shape == (C,), "pi must be a scalar or a C-vector"
        self.tau0, self.tau1 = tau0, tau1
        self.allow_self_connections = allow_self_connections
        self._init_engine(device=device, seed=seed, kind=_abi.SBM_JOINT, N=N, K=self.C, B=B, tau1=tau1, tau0=tau0,
                          allow_self_connections=1, special_case_self_conns=0)
        self.pi = pi
        if m is not None:
            assert isinstance(m, np.ndarray) and m.shape == (C,) and np.allclose(m.sum(), 1.0) and np.amin(m) >= 0.0, \
                "m must be a length C probability vector"
            self.m = m
        else:
            self.m = np.random.dirichlet(self.pi)
        if c is not None:
            assert isinstance(c, np.ndarray) and c.shape == (N,) and np.issubdtype(c.dtype, np.integer) \
                and np.amin(c) >= 0 and np.amax(c) <= C - 1, "c must be a length N vector of block assignments"
            self.c = c.copy()
        else:
            self.c = np.random.choice(self.C, p=self.m, size=N)
        if p is not None:
            if np.isscalar(p):
                assert 0 <= p <= 1, "p must be a probability"
                self.p = p * np.ones((C, C))
            else:
                assert isinstance(p, np.ndarray) and p.shape == (C, C) and np.amin(p) >= 0 and np.amax(p) <= 1.0, \
                    "p must be a CxC matrix of probabilities"
                self.p = p
        else:
            self.p = np.random.beta(tau1, tau0, size=(C, C))
        hyp = dict(mu_0=np.zeros(B), Sigma_0=np.eye(B), nu_0=B + 2, kappa_0=1.0)
        hyp.update(weight_hypers)
        self._hset("_niw", _pack_niw(hyp["mu_0"], hyp["Sigma_0"], hyp["nu_0"], hyp["kappa_0"]))
        mu, sig = np.zeros((C, C, B)), np.zeros((C, C, B, B))
        for a in range(C):
            for b in range(C):      # each weight model starts at a prior draw, as pybasicbayes' Gaussian does
                mu[a, b], sig[a, b] = _niw_prior_draw(hyp["mu_0"], hyp["Sigma_0"], hyp["nu_0"], hyp["kappa_0"])
        self._hset("_mu", mu)
        self._hset("_sigma", sig)
        self.weight_models = [[_GaussianView(self, "_mu", "_sigma", "_niw", (a, b)) for b in range(C)] for a in range(C)]

    c = property(lambda s: s._hget("c"), lambda s, v: s._hset("c", v))
    m = property(lambda s: s._hget("m"), lambda s, v: s._hset("m", v))
    pi = property(lambda s: s._hget("pi"), lambda s, v: s._hset("pi", v))
    p = property(lambda s: s._hget("p"), lambda s, v: s._hset("p", v))

    @property
    def P(self):
        # stochastic_block_model.py:112-122
        P = self.p[np.ix_(self.c, self.c)]
        if not self.allow_self_connections:
            np.fill_diagonal(P, 0.0)
        return P

    @property
    def Mu(self):
        # stochastic_block_model.py:594-610
        return self._hget("_mu")[np.ix_(self.c, self.c)]

    @property
    def Sigma(self):
        # stochastic_block_model.py:612-628
        return self._hget("_sigma")[np.ix_(self.c, self.c)]

    def rvs(self, size=[]):
        """(A, W): A_ij ~ Bern(p[c_i, c_j]), W_ij ~ N(mu[c_i, c_j], Sigma[c_i, c_j]) for every ordered pair."""
        N, B = self.N, self.B
        A = np.random.rand(N, N) < self.P
        L = np.linalg.cholesky(self.Sigma.reshape(-1, B, B)).reshape(N, N, B, B)
        W = self.Mu + np.einsum("mnij,mnj->mni", L, np.random.randn(N, N, B))
        return A, W

    def log_likelihood(self, data):
        """log p(A | c, p) + log p(W | A, c, mu, Sigma), from the block-pair tables on the device."""
        d = self._data(data)
        assert d.N == self.N and d.B == self.B
        return self._engine().log_likelihood(d)

    def _run(self, data, parts, beta=1.0):
        d = self._data(data) if data is not None else None
        assert d is None or (d.N == self.N and d.B == self.B)
        eng = self._engine()
        if parts == _abi.PART_ALL:
            eng.resample_async(d, 1, beta)
        else:
            eng.resample_part(d, parts, beta)
        self._mark_stale("c", "m", "p", "_mu", "_sigma")

    def resample(self, data, beta=1.0):
        """stochastic_block_model.py:206-230 (the reference takes a network object with .A and .W; here the (A, W) pair)."""
        self._run(data, _abi.PART_ALL, beta)

    def resample_p(self, data):
        """Weight models of every block pair and p (:220-250; one engine step - both are conjugate draws from the pair tables)."""
        self._run(data, _abi.PART_PARAMS)

    def resample_c(self, data):
        if self.C == 1:
            return
        self._run(data, _abi.PART_C)

    def resample_m(self):
        self._run(None, _abi.PART_M)
