"""Gaussian weight distributions with the interface of graphistician/weights.py.

resample / log_likelihood of the fixed, NIW and SBM Gaussian models run on the
MI355X engine; the reference delegates the NIW arithmetic to pybasicbayes
`Gaussian` objects (not vendored) — here `_gaussian`,
`_self_gaussian` and `_gaussians[c1][c2]` are light views onto the engine state
that expose the attributes the reference reads (`mu`, `sigma`, `sigma_chol`,
`mu_0`, `sigma_0`, `nu_0`, `kappa_0`).
"""
import numpy as np
from scipy.special import logsumexp
from scipy.stats import dirichlet as _dirichlet, invwishart as _invwishart, multivariate_normal as _mvn

from . import _abi
from .abstractions import GaussianWeightDistribution, GibbsSampling
from .engine import EngineBacked


def _pack_niw(mu_0, sigma_0, nu_0, kappa_0):
    return np.concatenate([np.asarray(mu_0, dtype=np.float64).ravel(), np.asarray(sigma_0, dtype=np.float64).ravel(),
                           [float(nu_0), float(kappa_0)]])


def _niw_prior_draw(mu_0, sigma_0, nu_0, kappa_0):
    """(mu, Sigma) ~ NIW: what pybasicbayes Gaussian.resample() with no data returns (weights.py:118-119)."""
    B = len(mu_0)
    sigma = np.atleast_2d(_invwishart.rvs(df=nu_0, scale=np.asarray(sigma_0).reshape(B, B)))
    mu = np.random.multivariate_normal(mu_0, sigma / kappa_0)
    return mu, sigma


def niw_log_prob(mu, sigma, mu_0, sigma_0, nu_0, kappa_0):
    """internals/utils.py:34-51 normal_inverse_wishart_log_prob."""
    lp = _invwishart.logpdf(sigma, nu_0, sigma_0)
    lp += _mvn.logpdf(mu, mu_0, sigma / kappa_0)
    return lp


def niw_log_prob_many(mu, sigma, mu_0, sigma_0, nu_0, kappa_0):
    """Sum of niw_log_prob over a batch of (mu, sigma) pairs that share one prior - mu (..., B), sigma (..., B, B) - in
    closed form (the K^2 scipy calls of the loop in weights.py:343-352 cost milliseconds per log_prior at K = 8):
      log IW(S; nu, Psi) = nu/2 log|Psi| - nu B/2 log 2 - log Gamma_B(nu/2) - (nu + B + 1)/2 log|S| - tr(Psi S^-1)/2
      log N(mu; mu_0, S/kappa) = -B/2 log(2 pi) - log|S/kappa|/2 - kappa/2 (mu - mu_0)' S^-1 (mu - mu_0)."""
    from scipy.special import multigammaln
    mu = np.asarray(mu, dtype=np.float64)
    sigma = np.asarray(sigma, dtype=np.float64)
    B = mu.shape[-1]
    mu = mu.reshape(-1, B)
    sigma = sigma.reshape(-1, B, B)
    n = mu.shape[0]
    psi = np.asarray(sigma_0, dtype=np.float64).reshape(B, B)
    _, logdet = np.linalg.slogdet(sigma)
    _, logdet_psi = np.linalg.slogdet(psi)
    sinv = np.linalg.inv(sigma)
    tr = np.einsum("ij,nji->n", psi, sinv)
    lp_iw = n * (0.5 * nu_0 * logdet_psi - 0.5 * nu_0 * B * np.log(2.0) - multigammaln(0.5 * nu_0, B)) \
        - 0.5 * (nu_0 + B + 1) * logdet.sum() - 0.5 * tr.sum()
    dmu = mu - np.asarray(mu_0, dtype=np.float64).reshape(1, B)
    quad = np.einsum("ni,nij,nj->n", dmu, sinv, dmu)
    lp_n = -0.5 * n * B * np.log(2.0 * np.pi) - 0.5 * (logdet - B * np.log(kappa_0)).sum() - 0.5 * kappa_0 * quad.sum()
    return float(lp_iw + lp_n)


class _GaussianView(object):
    """The slice of a pybasicbayes `Gaussian` the reference touches, backed by the owner's state arrays."""

    def __init__(self, owner, mu_name, sigma_name, niw_name, index=None):
        self._o, self._mu, self._sg, self._niw, self._ix = owner, mu_name, sigma_name, niw_name, index

    def _get(self, name):
        a = self._o._hget(name)
        return a[self._ix] if self._ix is not None else a

    @property
    def mu(self):
        return self._get(self._mu).copy()

    @mu.setter
    def mu(self, v):
        a = self._o._hget(self._mu).copy()
        if self._ix is not None:
            a[self._ix] = v
        else:
            a[...] = v
        self._o._hset(self._mu, a)

    @property
    def sigma(self):
        return self._get(self._sg).copy()

    @sigma.setter
    def sigma(self, v):
        a = self._o._hget(self._sg).copy()
        if self._ix is not None:
            a[self._ix] = v
        else:
            a[...] = v
        self._o._hset(self._sg, a)

    @property
    def sigma_chol(self):
        return np.linalg.cholesky(self.sigma)

    def _hyp(self):
        B = self._o.B
        h = self._o._hget(self._niw)
        return h[:B].copy(), h[B:B + B * B].reshape(B, B).copy(), h[B + B * B], h[B + B * B + 1]

    def _set_hyp(self, pos, v):
        B = self._o.B
        h = self._o._hget(self._niw).copy()
        if pos == 0:
            h[:B] = np.asarray(v).ravel()
        elif pos == 1:
            h[B:B + B * B] = np.asarray(v).ravel()
        else:
            h[B + B * B + (pos - 2)] = v
        self._o._hset(self._niw, h)

    mu_0 = property(lambda s: s._hyp()[0], lambda s, v: s._set_hyp(0, v))
    sigma_0 = property(lambda s: s._hyp()[1], lambda s, v: s._set_hyp(1, v))
    nu_0 = property(lambda s: s._hyp()[2], lambda s, v: s._set_hyp(2, v))
    kappa_0 = property(lambda s: s._hyp()[3], lambda s, v: s._set_hyp(3, v))

    def resample(self, data=[]):
        """Prior draw (pybasicbayes Gaussian.resample() without data), host numpy."""
        assert len(data) == 0, "posterior draws happen inside the engine sweep"
        mu, sigma = _niw_prior_draw(*self._hyp())
        self.mu, self.sigma = mu, sigma

    def rvs(self, size=None):
        return np.random.multivariate_normal(self.mu, self.sigma, size=size)

    def log_likelihood(self, x):
        x = np.atleast_2d(x)
        if x.shape[0] == 0:
            return np.zeros(0)
        return np.atleast_1d(_mvn.logpdf(x, self.mu, self.sigma))


class _WeightsBase(GaussianWeightDistribution, GibbsSampling, EngineBacked):
    def log_likelihood(self, data):
        d = self._data(data)
        assert d.N == self.N and d.B == self.B
        return self._engine().log_likelihood(d)


class FixedGaussianWeightDistribution(_WeightsBase):
    """weights.py:9-65."""

    _fields = {"_mu": (_abi.ST_MU, np.float64, lambda s: (1, 1, s.B)),
               "_sigma": (_abi.ST_SIGMA, np.float64, lambda s: (1, 1, s.B, s.B)),
               "_mu_self": (_abi.ST_MU_SELF, np.float64, lambda s: (s.B,)),
               "_sigma_self": (_abi.ST_SIGMA_SELF, np.float64, lambda s: (s.B, s.B)),
               "_niw": (_abi.ST_NIW, np.float64, lambda s: (s.B + s.B * s.B + 2,)),
               "_niw_self": (_abi.ST_NIW_SELF, np.float64, lambda s: (s.B + s.B * s.B + 2,))}

    def __init__(self, N, B, mu, sigma, mu_self=None, sigma_self=None, device=0, seed=None):
        super(FixedGaussianWeightDistribution, self).__init__(N)
        self.B = B
        assert mu.shape == (B,)
        self.mu = mu
        assert sigma.shape == (B, B)
        self.sigma = sigma
        self._init_engine(device=device, seed=seed, kind=_abi.FIXED_W, N=N, K=1, B=B, special_case_self_conns=1)
        self._hset("_mu", mu)
        self._hset("_sigma", sigma)
        if mu_self is not None and sigma_self is not None:
            self._hset("_mu_self", mu_self)
            self._hset("_sigma_self", sigma_self)
        else:
            self._hset("_mu_self", mu)
            self._hset("_sigma_self", sigma)
        hyp = _pack_niw(np.zeros(B), np.eye(B), B + 2, 1.0)
        self._hset("_niw", hyp)
        self._hset("_niw_self", hyp)
        self._gaussian = _GaussianView(self, "_mu", "_sigma", "_niw", (0, 0))
        self._self_gaussian = _GaussianView(self, "_mu_self", "_sigma_self", "_niw_self")

    @property
    def Mu(self):
        Mu = np.tile(self._gaussian.mu[None, None, :], (self.N, self.N, 1))
        Mu[np.arange(self.N), np.arange(self.N)] = self._self_gaussian.mu
        return Mu

    @property
    def Sigma(self):
        Sig = np.tile(self._gaussian.sigma[None, None, :, :], (self.N, self.N, 1, 1))
        Sig[np.arange(self.N), np.arange(self.N)] = self._self_gaussian.sigma
        return Sig

    def initialize_from_prior(self):
        pass

    def initialize_hypers(self, W=None):
        pass

    def log_prior(self):
        return 0

    def sample_predictive_parameters(self):
        Murow = Mucol = np.tile(self._gaussian.mu[None, :], (self.N + 1, 1))
        Lrow = Lcol = np.tile(self._gaussian.sigma_chol[None, :, :], (self.N + 1, 1, 1))
        return Murow, Mucol, Lrow, Lcol

    def resample(self, data=[]):
        pass


class NIWGaussianWeightDistribution(_WeightsBase):
    """weights.py:68-162: one Gaussian for off-diagonal edges, one for self edges, NIW priors."""

    _fields = FixedGaussianWeightDistribution._fields

    def __init__(self, N, B=1, mu_0=None, Sigma_0=None, nu_0=None, kappa_0=None, device=0, seed=None, wide_weights=False):
        """wide_weights=True (B = 1): reference-precision weight storage - 48 bits in two limbs under the one scale instead of
        24 (the reference computes on float64 W: abstractions.py:257-274) - for data with a wide dynamic range; the (A, W)
        arrays handed to the model are then uploaded in that form."""
        super(NIWGaussianWeightDistribution, self).__init__(N)
        self.B = B
        self._wide = bool(wide_weights)
        if mu_0 is None:
            mu_0 = np.zeros(B)
        if Sigma_0 is None:
            Sigma_0 = np.eye(B)
        if nu_0 is None:
            nu_0 = B + 2
        if kappa_0 is None:
            kappa_0 = 1.0
        self._init_engine(device=device, seed=seed, kind=_abi.NIW_W, N=N, K=1, B=B, special_case_self_conns=1,
                          wide_weights=1 if wide_weights else 0)
        hyp = _pack_niw(mu_0, Sigma_0, nu_0, kappa_0)
        self._hset("_niw", hyp)
        self._hset("_niw_self", hyp)
        self._hset("_mu", np.zeros((1, 1, B)))
        self._hset("_sigma", np.eye(B).reshape(1, 1, B, B))
        self._hset("_mu_self", np.zeros(B))
        self._hset("_sigma_self", np.eye(B))
        self._gaussian = _GaussianView(self, "_mu", "_sigma", "_niw", (0, 0))
        self._self_gaussian = _GaussianView(self, "_mu_self", "_sigma_self", "_niw_self")
        # pybasicbayes Gaussian(mu_0=..) initialises itself with a prior draw
        self._gaussian.resample()
        self._self_gaussian.resample()

    Mu = FixedGaussianWeightDistribution.Mu
    Sigma = FixedGaussianWeightDistribution.Sigma

    def initialize_from_prior(self):
        self._gaussian.resample()
        self._self_gaussian.resample()

    def initialize_hypers(self, W):
        # weights.py:121-134
        self._gaussian.mu_0 = W.mean(axis=(0, 1))
        self._gaussian.sigma_0 = np.diag(W.var(axis=(0, 1)))
        self._gaussian.resample()
        W_self = W[np.arange(self.N), np.arange(self.N)]
        self._self_gaussian.mu_0 = W_self.mean(axis=0)
        self._self_gaussian.sigma_0 = np.diag(W_self.var(axis=0))
        self._self_gaussian.resample()

    def log_prior(self):
        # weights.py:136-142
        lp = 0
        for g in (self._gaussian, self._self_gaussian):
            lp += niw_log_prob(g.mu, g.sigma, g.mu_0, g.sigma_0, g.nu_0, g.kappa_0)
        return lp

    def sample_predictive_parameters(self):
        # weights.py:144-152
        Murow = np.tile(self._gaussian.mu[None, :], (self.N + 1, 1))
        Mucol = Murow.copy()
        Lrow = np.tile(self._gaussian.sigma_chol[None, :, :], (self.N + 1, 1, 1))
        Lcol = Lrow.copy()
        Murow[-1, :] = Mucol[-1, :] = self._self_gaussian.mu
        Lrow[-1, :, :] = Lcol[-1, :, :] = self._self_gaussian.sigma_chol
        return Murow, Mucol, Lrow, Lcol

    def approx_predictive_ll(self, Arow, Acol, Wrow, Wcol, M=100):
        """abstractions.py:313-357 on the engine: one Gaussian for every edge of the held-out node, the self Gaussian for its
        self connection (weights.py:144-152) - nothing to average over."""
        assert Arow.shape == Acol.shape == (self.N + 1,)
        return float(self._engine().predictive_ll(Arow, Acol, Wrow, Wcol, K=1)[0])

    def resample(self, data, beta=1.0):
        """weights.py:154-162.  beta < 1: tempered posterior (build-defined for weights: the Gaussian likelihood
        to the power beta scales the sufficient statistics, the way graph_models.py:366-367 scales counts)."""
        d = self._data(data)
        assert d.N == self.N and d.B == self.B
        self._engine().resample_async(d, 1, beta)
        self._mark_stale("_mu", "_sigma", "_mu_self", "_sigma_self")


class LowRankGaussianWeightDistribution(GaussianWeightDistribution, GibbsSampling):
    """weights.py:165-170."""

    def __init__(self, N, dim):
        raise NotImplementedError


class SBMGaussianWeightDistribution(_WeightsBase):
    """weights.py:172-498: K x K Gaussians indexed by the blocks of the two endpoints."""

    _kind = _abi.SBM_GAUSS_W

    _fields = {"c": (_abi.ST_C, np.int32, lambda s: (s.N,)),
               "m": (_abi.ST_M, np.float64, lambda s: (s.C,)),
               "pi": (_abi.ST_PI, np.float64, lambda s: (s.C,)),
               "_mu": (_abi.ST_MU, np.float64, lambda s: (s.C, s.C, s.B)),
               "_sigma": (_abi.ST_SIGMA, np.float64, lambda s: (s.C, s.C, s.B, s.B)),
               "_mu_self": (_abi.ST_MU_SELF, np.float64, lambda s: (s.B,)),
               "_sigma_self": (_abi.ST_SIGMA_SELF, np.float64, lambda s: (s.B, s.B)),
               "_niw": (_abi.ST_NIW, np.float64, lambda s: (s.B + s.B * s.B + 2,)),
               "_niw_self": (_abi.ST_NIW_SELF, np.float64, lambda s: (s.B + s.B * s.B + 2,))}

    def __init__(self, N, B=1, C=2, pi=10.0, mu_0=None, Sigma_0=None, nu_0=None, kappa_0=None,
                 special_case_self_conns=True, device=0, seed=None, wide_weights=False):
        """wide_weights=True (B = 1, C <= 64): reference-precision weight storage, see NIWGaussianWeightDistribution."""
        super(SBMGaussianWeightDistribution, self).__init__(N)
        self.B = B
        self._wide = bool(wide_weights)
        assert isinstance(C, (int, np.integer)) and C >= 1, "C must be a positive integer number of blocks"
        self.C = int(C)
        if isinstance(pi, (int, float)):
            pi = pi * np.ones(C)
        else:
            assert isinstance(pi, np.ndarray) and pi.shape == (C,), "pi must be a sclar or a C-vector"
        self.special_case_self_conns = special_case_self_conns
        self._init_engine(device=device, seed=seed, kind=self._kind, N=N, K=self.C, B=B,
                          special_case_self_conns=1 if special_case_self_conns else 0, wide_weights=1 if wide_weights else 0)
        self.pi = pi
        self.m = np.random.dirichlet(self.pi)
        self.c = np.random.choice(self.C, p=self.m, size=(self.N))
        if mu_0 is None:
            mu_0 = np.zeros(B)
        if Sigma_0 is None:
            Sigma_0 = np.eye(B)
        if nu_0 is None:
            nu_0 = B + 2
        if kappa_0 is None:
            kappa_0 = 1.0
        hyp = _pack_niw(mu_0, Sigma_0, nu_0, kappa_0)
        self._hset("_niw", hyp)
        self._hset("_niw_self", hyp)
        self._hset("_mu", np.zeros((C, C, B)))
        self._hset("_sigma", np.tile(np.eye(B), (C, C, 1, 1)))
        self._hset("_mu_self", np.zeros(B))
        self._hset("_sigma_self", np.eye(B))
        self._gaussians = [[_GaussianView(self, "_mu", "_sigma", "_niw", (c1, c2)) for c2 in range(C)]
                           for c1 in range(C)]
        for row in self._gaussians:
            for g in row:
                g.resample()
        if special_case_self_conns:
            self._self_gaussian = _GaussianView(self, "_mu_self", "_sigma_self", "_niw_self")
            self._self_gaussian.resample()

    c = property(lambda s: s._hget("c"), lambda s, v: s._hset("c", v))
    m = property(lambda s: s._hget("m"), lambda s, v: s._hset("m", v))
    pi = property(lambda s: s._hget("pi"), lambda s, v: s._hset("pi", v))

    @property
    def _Mu(self):
        return self._hget("_mu").copy()

    @property
    def _Sigma(self):
        return self._hget("_sigma").copy()

    @property
    def Mu(self):
        # weights.py:243-256
        c = self.c
        Mu = self._Mu[np.ix_(c, c)]
        if self.special_case_self_conns:
            Mu[np.arange(self.N), np.arange(self.N)] = self._self_gaussian.mu
        return Mu

    @property
    def Sigma(self):
        # weights.py:258-271
        c = self.c
        Sigma = self._Sigma[np.ix_(c, c)]
        if self.special_case_self_conns:
            Sigma[np.arange(self.N), np.arange(self.N)] = self._self_gaussian.sigma
        return Sigma

    def initialize_from_prior(self):
        # weights.py:273-281
        self.m = np.random.dirichlet(self.pi)
        self.c = np.random.choice(self.C, p=self.m, size=(self.N))
        for row in self._gaussians:
            for g in row:
                g.resample()
        if self.special_case_self_conns:
            self._self_gaussian.resample()

    def initialize_hypers(self, W):
        # weights.py:283-306 (all block pairs share one prior here, as they do after this call in the reference)
        g00 = self._gaussians[0][0]
        nu_0 = g00.nu_0
        g00.mu_0 = W.mean(axis=(0, 1))
        g00.sigma_0 = np.diag(W.var(axis=(0, 1))) * (nu_0 - self.B - 1) / self.C
        for row in self._gaussians:
            for g in row:
                g.resample()
        if self.special_case_self_conns:
            W_self = W[np.arange(self.N), np.arange(self.N)]
            self._self_gaussian.mu_0 = W_self.mean(axis=0)
            self._self_gaussian.sigma_0 = np.diag(W_self.var(axis=0))
            self._self_gaussian.resample()
        from sklearn.cluster import KMeans
        features = np.hstack((W[:, :, 0], W[:, :, 0].T))
        km = KMeans(n_clusters=self.C, n_init=10)
        km.fit(features)
        self.c = km.labels_.astype(np.int32)

    _device_prior = True      # gg_log_prior_many evaluates this class's log_prior from the engine's state

    def log_prior(self):
        """weights.py:333-357; from the engine's state once the model lives there (see SBMAdjacencyDistribution.log_prior)."""
        if self._model is not None:
            return _abi.log_prior_many([self._engine()])[0]
        return self._log_prior_host()

    def _log_prior_host(self):
        # weights.py:333-357
        lp = 0
        lp += _dirichlet(self.pi).logpdf(self.m)
        g0 = self._gaussians[0][0]                      # (every pair shares the one NIW prior)
        lp += niw_log_prob_many(self._hget("_mu"), self._hget("_sigma"), g0.mu_0, g0.sigma_0, g0.nu_0, g0.kappa_0)
        if self.special_case_self_conns:
            g = self._self_gaussian
            lp += niw_log_prob(g.mu, g.sigma, g.mu_0, g.sigma_0, g.nu_0, g.kappa_0)
        lp += (np.log(self.m)[self.c]).sum()
        return lp

    def rvs(self, size=[]):
        # weights.py:360-373 (same law: W_ij ~ N(mu[c_i,c_j], Sigma[c_i,c_j]); self Gaussian on the diagonal)
        return GaussianWeightDistribution.rvs(self)

    def sample_predictive_parameters(self):
        # weights.py:375-389
        c2 = np.random.choice(self.C, p=self.m)
        cext = np.concatenate((self.c, [c2]))
        mu, sig = self._Mu, self._Sigma
        chol = np.linalg.cholesky(sig)
        Murow, Lrow = mu[c2, cext], chol[c2, cext]
        Mucol, Lcol = mu[cext, c2], chol[cext, c2]
        if self.special_case_self_conns:
            Murow[-1] = Mucol[-1] = self._self_gaussian.mu
            Lrow[-1] = Lcol[-1] = self._self_gaussian.sigma_chol
        return Murow, Mucol, Lrow, Lcol

    def approx_predictive_ll(self, Arow, Acol, Wrow, Wcol, M=100):
        """abstractions.py:313-357 on the engine: a predictive draw (weights.py:375-389) is the held-out node's block
        c2 ~ Cat(m); the engine evaluates the Gaussian log-densities of its row and column for every block
        (gg_predictive_ll), the M draws average over them."""
        assert Arow.shape == Acol.shape == (self.N + 1,)
        m = self.m
        ll_k = self._engine().predictive_ll(Arow, Acol, Wrow, Wcol, K=self.C)
        c2 = np.random.choice(self.C, p=m, size=M)
        return float(-np.log(M) + logsumexp(ll_k[c2]))

    # -- Gibbs (weights.py:394-498): on the device ---------------------------------
    def _run(self, data, parts, beta=1.0):
        d = self._data(data) if data is not None else None
        assert d is None or (d.N == self.N and d.B == self.B)
        eng = self._engine()
        if parts == _abi.PART_ALL:
            eng.resample_async(d, 1, beta)    # stream-ordered: the next state read (or log_likelihood) waits for it
        else:
            eng.resample_part(d, parts, beta)   # (the engine advances its Philox sweep index: graphgibbs.h, gg_resample_part)
        self._mark_stale("c", "m", "_mu", "_sigma", "_mu_self", "_sigma_self")

    def resample(self, data, beta=1.0):
        self._run(data, _abi.PART_ALL, beta)

    def resample_mu_and_Sig(self, A, W, beta=1.0):
        self._run((A, W), _abi.PART_PARAMS, beta)

    def resample_c(self, A, W, beta=1.0):
        if self.C == 1:
            return
        self._run((A, W), _abi.PART_C, beta)

    def resample_m(self):
        self._run(None, _abi.PART_M)


class _SharedCovView(object):
    """Stand-in for the reference's `_cov_model` (pybasicbayes GaussianFixedMean): sigma, nu_0, lmbda_0."""

    def __init__(self, owner):
        self._o = owner

    @property
    def sigma(self):
        return self._o._hget("_sigma")[0, 0].copy()

    @property
    def nu_0(self):
        B = self._o.B
        return float(self._o._hget("_niw")[B + B * B])

    @property
    def lmbda_0(self):
        B = self._o.B
        return self._o._hget("_niw")[B:B + B * B].reshape(B, B).copy()


class SBMGaussianWeightSharedCov(SBMGaussianWeightDistribution):
    """weights.py:501-610: block-pair means with ONE covariance shared by all pairs.  resample_mu_and_Sig draws every
    pair's mean given the covariance (GaussianFixedCov, prior N(mu_0, I)), the self Gaussian, then the covariance from
    the centred off-diagonal edges (GaussianFixedMean / inverse Wishart(lmbda_0 = Sigma_0, nu_0)); the engine kind differs,
    the scan and the likelihood are the parent's.  Without special_case_self_conns the diagonal belongs to pair (c_i, c_i) for
    the means and stays out of the covariance's centred scatter, as in the reference (:545-548)."""

    _kind = _abi.SBM_GAUSS_SHCOV_W

    def __init__(self, N, B=1, C=3, pi=10.0, mu_0=None, Sigma_0=None, nu_0=None, special_case_self_conns=True,
                 device=0, seed=None):
        super(SBMGaussianWeightSharedCov, self).__init__(N, B=B, C=C, pi=pi, mu_0=mu_0, Sigma_0=Sigma_0, nu_0=nu_0,
                                                         special_case_self_conns=special_case_self_conns, device=device,
                                                         seed=seed)
        # one covariance for every pair: start from the (0, 0) draw of the parent's constructor
        sig = self._hget("_sigma").copy()
        sig[:] = sig[0, 0]
        self._hset("_sigma", sig)
        self._cov_model = _SharedCovView(self)

    def _log_prior_host(self):
        # weights.py:554-577 (the Gaussian terms are commented out in the reference too)
        lp = _dirichlet(self.pi).logpdf(self.m)
        lp += (np.log(self.m)[self.c]).sum()
        return lp


class _FixedMeanCovView(object):
    """Stand-in for `cov` of the latent-distance weights (pybasicbayes GaussianFixedMean): mu, sigma, lmbda_0, nu_0."""

    def __init__(self, owner):
        self._o = owner

    mu = property(lambda s: np.zeros(1))
    sigma = property(lambda s: np.array([[s._o._aux_get(1)]]),
                     lambda s, v: s._o._aux_set(1, float(np.asarray(v).ravel()[0])))
    lmbda_0 = property(lambda s: np.array([[s._o._hget("_niw")[1]]]))
    nu_0 = property(lambda s: float(s._o._hget("_niw")[2]))

    def resample(self, data=[]):
        assert len(data) == 0, "posterior draws happen inside the engine sweep"
        self.sigma = np.atleast_2d(_invwishart.rvs(df=self.nu_0, scale=self.lmbda_0))


class _AuxGaussianView(_GaussianView):
    """The self Gaussian of the latent-distance weights: mu, sigma live in slots 2, 3 of the auxiliary state."""

    def __init__(self, owner):
        super(_AuxGaussianView, self).__init__(owner, None, None, "_niw_self")

    mu = property(lambda s: np.array([s._o._aux_get(2)]), lambda s, v: s._o._aux_set(2, float(np.asarray(v).ravel()[0])))
    sigma = property(lambda s: np.array([[s._o._aux_get(3)]]), lambda s, v: s._o._aux_set(3, float(np.asarray(v).ravel()[0])))


class LatentDistanceGaussianWeightDistribution(_WeightsBase):
    """weights.py:613-903.

    l_n ~ N(0, eta I);  W_{n',n} ~ N(-||l_{n'} - l_n||^2 + b, sigma) for n' != n, self weights ~ N(mu_self, sigma_self).
    resample = HMC on L (step 0.005, 10 leapfrog steps), b from its Gaussian conditional, sigma from the inverse-Wishart
    conditional of the residuals, the self Gaussian from its NIW posterior, eta from its inverse-gamma conditional
    (:767-886), all on the engine.  B = 1 as in the reference (:711).  Like the adjacency counterpart the gradient is
    hand-derived and the HMC transition restated (autograd / hips.inference.hmc are not available): parity unpinned,
    checked against the oracle's same definition.  beta < 1 tempers the likelihood (AIS)."""

    _fields = {"L": (_abi.ST_F, np.float64, lambda s: (s.N, s.dim)),
               "_b": (_abi.ST_MU0, np.float64, lambda s: (1,)),
               "_aux": (_abi.ST_LD_AUX, np.float64, lambda s: (4,)),      # eta, sigma, mu_self, sigma_self
               "_niw": (_abi.ST_NIW, np.float64, lambda s: (4,)),
               "_niw_self": (_abi.ST_NIW_SELF, np.float64, lambda s: (4,))}

    def __init__(self, N, B=1, dim=2, b=0.5, sigma=None, Sigma_0=None, nu_0=None, mu_self=0.0, eta=0.01,
                 device=0, seed=None):
        super(LatentDistanceGaussianWeightDistribution, self).__init__(N)
        if B != 1:
            raise NotImplementedError("LatentDistanceGaussianWeightDistribution: B = 1 only (weights.py:711 asserts it too)")
        self.B = B
        self.dim = dim
        if Sigma_0 is None:
            Sigma_0 = np.eye(B)
        if nu_0 is None:
            nu_0 = B + 2
        self._init_engine(device=device, seed=seed, kind=_abi.LATENT_DIST_W, N=N, D=dim, K=1, B=1)
        s0 = float(np.asarray(Sigma_0).ravel()[0])
        self._hset("_niw", np.array([0.0, s0, float(nu_0), 0.0]))
        self._hset("_niw_self", np.array([float(mu_self), s0, float(nu_0), 1.0]))
        self._hset("_b", np.array([b]))
        self._hset("_aux", np.array([eta, 1.0, 0.0, 1.0]))
        self.L = np.sqrt(eta) * np.random.randn(N, dim)
        self.cov = _FixedMeanCovView(self)
        self._self_gaussian = _AuxGaussianView(self)
        # pybasicbayes: GaussianFixedMean(sigma=None) and Gaussian(mu_0=...) start from prior draws
        if sigma is None:
            self.cov.resample()
        else:
            self.cov.sigma = sigma
        self._self_gaussian.resample()

    def _aux_get(self, k):
        return float(self._hget("_aux")[k])

    def _aux_set(self, k, v):
        a = self._hget("_aux").copy()
        a[k] = v
        self._hset("_aux", a)

    L = property(lambda s: s._hget("L"), lambda s, v: s._hset("L", v))
    b = property(lambda s: float(s._hget("_b")[0]), lambda s, v: s._hset("_b", np.array([v], dtype=np.float64)))
    eta = property(lambda s: s._aux_get(0), lambda s, v: s._aux_set(0, v))

    @property
    def D(self):
        return ((self.L[:, None, :] - self.L[None, :, :]) ** 2).sum(2)

    @property
    def Mu(self):
        # weights.py:653-660
        Mu = np.tile((-self.D + self.b)[:, :, None], (1, 1, self.B))
        Mu[np.arange(self.N), np.arange(self.N), :] = self._self_gaussian.mu
        return Mu

    @property
    def Sigma(self):
        # weights.py:662-670
        Sig = np.tile(self.cov.sigma[None, None, :, :], (self.N, self.N, 1, 1))
        Sig[np.arange(self.N), np.arange(self.N)] = self._self_gaussian.sigma
        return Sig

    def initialize_from_prior(self):
        # weights.py:672-674
        self.L = np.sqrt(self.eta) * np.random.randn(self.N, self.dim)
        self.cov.resample()

    def prior_draw(self):
        """Every parameter from its prior (weights.py:680-701), for AIS: initialize_from_prior leaves b, eta and the
        self Gaussian where they are."""
        from scipy.stats import invgamma
        self.eta = float(invgamma.rvs(1.0, scale=1.0))
        self.b = float(np.random.randn())
        self.initialize_from_prior()
        self._self_gaussian.resample()

    def _location_objective(self, Lflat, A, W):
        """-log p(W, L | b, sigma, eta) as a function of the locations and its gradient (weights.py:701-733 hands the same
        function to autograd; the gradient here is written out: d/dl_i = (2/sigma) sum_j (r_ij a_ij + r_ji a_ji)(l_i - l_j)
        + l_i / eta for the NEGATIVE log probability, r = W - (-|l_i - l_j|^2 + b), a = off-diagonal edge mask)."""
        N = self.N
        L = Lflat.reshape(N, self.dim)
        diff = L[:, None, :] - L[None, :, :]
        Mu = -(diff ** 2).sum(2) + self.b
        Aoff = np.asarray(A, dtype=np.float64) * (1.0 - np.eye(N))
        R = (np.asarray(W, dtype=np.float64).reshape(N, N) - Mu) * Aoff
        lmb = 1.0 / float(np.asarray(self.cov.sigma).ravel()[0])
        f = 0.5 * lmb * (R ** 2).sum() + 0.5 * (L * L).sum() / self.eta + 0.5 * self.b ** 2
        S = R + R.T
        g = 2.0 * lmb * (S.sum(1)[:, None] * L - S.dot(L)) + L / self.eta
        return f, g.ravel()

    def initialize_hypers(self, W):
        # weights.py:676-678, 792-808: the initial locations by BFGS on -log p with every pair counted as an edge (host
        # code in the reference as well: initialisation, not part of the sampler)
        from scipy.optimize import minimize
        A = np.ones((self.N, self.N))
        res = minimize(self._location_objective, np.ravel(self.L), args=(A, W), jac=True, method="BFGS")
        self.L = res.x.reshape(self.N, self.dim)

    def log_prior(self):
        # weights.py:680-701
        from scipy.stats import norm, invgamma
        lp = invgamma.logpdf(self.eta, 1, scale=1)
        lp += norm.logpdf(self.b, 0, 1)
        lp += norm.logpdf(self.L, 0, 1).sum()
        lp += _invwishart.logpdf(self.cov.sigma, self.cov.nu_0, self.cov.lmbda_0)
        g = self._self_gaussian
        lp += niw_log_prob(g.mu, g.sigma, g.mu_0, g.sigma_0, g.nu_0, g.kappa_0)
        return lp

    def sample_predictive_parameters(self):
        # weights.py:735-764
        Lext = np.vstack((self.L, np.sqrt(self.eta) * np.random.randn(1, self.dim)))
        D = ((Lext[:, None, :] - Lext[None, :, :]) ** 2).sum(2)
        Mu = -D + self.b
        Mu_row = np.tile(Mu[-1, :][:, None], (1, self.B))
        Mu_row[-1] = self._self_gaussian.mu
        Mu_col = Mu_row.copy()
        Lc = np.linalg.cholesky(self.cov.sigma)
        L_row = np.tile(Lc[None, :, :], (self.N + 1, 1, 1))
        L_row[-1] = np.linalg.cholesky(self._self_gaussian.sigma)
        L_col = L_row.copy()
        return Mu_row, Mu_col, L_row, L_col

    def _run(self, data, parts, beta=1.0):
        d = self._data(data)
        assert d.N == self.N and d.B == self.B
        eng = self._engine()
        if parts == _abi.PART_ALL:
            eng.resample(d, 1, beta)
        else:
            eng.resample_part(d, parts, beta)
        self._mark_stale("L", "_b", "_aux")

    def resample(self, data, beta=1.0):
        """weights.py:767-773: L, b, cov, self Gaussian, eta."""
        self._run(data, _abi.PART_ALL, beta)

    def _resample_L(self, A, W, beta=1.0):
        self._run((A, W), _abi.PART_F, beta)

    def _resample_b(self, A, W, beta=1.0):
        self._run((A, W), _abi.PART_MU0, beta)

    def _resample_cov(self, A, W, beta=1.0):
        """sigma and (same engine part) the self Gaussian: weights.py:859-868."""
        self._run((A, W), _abi.PART_PARAMS, beta)

    _resample_self_gaussian = _resample_cov

    def _resample_eta(self, A=None, W=None):
        # eta | L does not involve the data (weights.py:870-886); the engine part takes a handle all the same
        if A is None:
            raise ValueError("pass the (A, W) the model is being fitted to: the engine part needs a data handle")
        self._run((A, W), _abi.PART_LMBDA, 1.0)
