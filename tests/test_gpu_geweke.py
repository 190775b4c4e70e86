"""Statistical tests ON THE HIP PATH (the oracle shares the engine's arithmetic spec, so a wrong
sampler in `csrc/spec/` would pass every engine-vs-oracle comparison; these tests share nothing with the oracle).

* Geweke "getting it right" joint-distribution tests with assertions — the methodology of the reference's
  test/test_logistic_eigenmodel_geweke.py:35-52, test/test_probit_eigenmodel_geweke.py:35-52 and
  examples/demo.py:170-275 (which only plot): alternate data ~ p(data | theta) (host `rvs`) and
  theta ~ resample(data) (the engine, through the host classes); every marginal of theta must then be its PRIOR.
  Run for the SBM adjacency model (p, p_self, m, c vs Beta / Dirichlet priors), NIW and SBM Gaussian weights
  (mu, Sigma vs Student-t / inverse-gamma marginals of the NIW prior), the logistic and the probit eigenmodel
  (F, mu_0, lambda vs their normal priors).
* The reference's one asserted test, test/test_distributions.py:14-36 — moments of the truncated normal against
  scipy.stats.truncnorm — against `spec/gg_probit.h` executed on the device, plus the two-stage identity of its
  third test (:39-56: sign, then truncated normal, is the untruncated normal).
* Laws of the device's primitive draws against scipy (gamma, beta, Dirichlet, NIW, PG(1,z), categorical).

Kolmogorov-Smirnov p-values are asserted > 1e-3 on thinned chains; means by batch-means z-scores < 4.5.
"""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def abi():
    from graphistician_amd import _abi
    if _abi.device_count() < 1:
        pytest.fail("no HIP device: the engine has no CPU fallback")
    return _abi


def batch_z(x, mean, nb=25):
    """z-score of the sample mean against `mean`, the standard error estimated from batch means (autocorrelated chain)."""
    x = np.asarray(x, dtype=np.float64)
    n = (len(x) // nb) * nb
    bm = x[:n].reshape(nb, -1).mean(1)
    return (bm.mean() - mean) / (bm.std(ddof=1) / np.sqrt(nb))


def check_law(x, dist, args=(), thin=8, burn=200, mean=None):
    """The data sets are kept tiny on purpose: the more data, the tighter the posterior and the slower a Geweke chain
    forgets (its autocorrelation is about 1 - prior precision / posterior precision), and a KS test on correlated
    samples rejects correct samplers."""
    x = np.asarray(x)[burn:]
    assert stats.kstest(x[::thin], dist, args=args).pvalue > 1e-3, (dist, args)
    if mean is not None:
        assert abs(batch_z(x, mean)) < 4.5, (dist, batch_z(x, mean))


# ------------------------------------------------------------------ Geweke: SBM adjacency (adjacency.py:615-702)
def test_geweke_sbm_adjacency_on_the_engine(abi):
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    np.random.seed(11)
    N, C, iters = 8, 2, 4000
    m = SBMAdjacencyDistribution(N, C=C, pi=2.0, tau0=1.5, tau1=0.8, seed=2024)
    tr = dict(p01=[], p11=[], ps=[], m0=[], c0=[])
    for it in range(iters):
        A = m.rvs()
        m.resample(A)
        tr["p01"].append(m.p[0, 1]); tr["p11"].append(m.p[1, 1]); tr["ps"].append(m.p_self)
        tr["m0"].append(m.m[0]); tr["c0"].append(m.c[0])
    check_law(tr["p01"], "beta", (0.8, 1.5), mean=0.8 / 2.3)
    check_law(tr["p11"], "beta", (0.8, 1.5), mean=0.8 / 2.3)
    check_law(tr["ps"], "beta", (0.8, 1.5), mean=0.8 / 2.3)
    check_law(tr["m0"], "beta", (2.0, 2.0), mean=0.5)
    assert abs(batch_z(np.array(tr["c0"][200:], dtype=float), 0.5)) < 4.5     # P(c_0 = 1) = E m_1 = 1/2


def test_geweke_sbm_adjacency_without_the_self_connection_special_case(abi):
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    np.random.seed(12)
    N, C, iters = 7, 3, 4000
    m = SBMAdjacencyDistribution(N, C=C, pi=1.5, special_case_self_conns=False, seed=77)
    p00, m2 = [], []
    for it in range(iters):
        m.resample(m.rvs())
        p00.append(m.p[0, 0]); m2.append(m.m[2])
    check_law(p00, "uniform", mean=0.5)
    check_law(m2, "beta", (1.5, 3.0), mean=1.0 / 3.0)


# ------------------------------------------------------------------ Geweke: Gaussian weights (weights.py:154-162, 394-498)
def _niw_marginals_1d(mu_0, s0, nu0, k0):
    """B = 1: sigma^2 ~ InvGamma(nu0/2, s0/2); mu ~ Student-t_{nu0}(mu_0, sqrt(s0 / (nu0 k0)))."""
    return ("invgamma", (nu0 / 2.0, 0.0, s0 / 2.0)), ("t", (nu0, mu_0, np.sqrt(s0 / (nu0 * k0))))


def test_geweke_niw_weights_on_the_engine(abi):
    from graphistician_amd.weights import NIWGaussianWeightDistribution
    np.random.seed(13)
    N, iters = 4, 6000
    A = np.zeros((N, N), dtype=np.uint8)
    A[0, 1] = A[1, 0] = A[2, 3] = A[3, 1] = A[0, 0] = A[2, 2] = 1          # 4 edges + 2 self edges
    w = NIWGaussianWeightDistribution(N, B=1, mu_0=np.array([0.3]), Sigma_0=np.array([[1.4]]), nu_0=5.0, kappa_0=0.7, seed=5)
    mus, sig, mus_s, sig_s = [], [], [], []
    for it in range(iters):
        W = w.rvs()
        w.resample((A, W))
        mus.append(w._gaussian.mu[0]); sig.append(w._gaussian.sigma[0, 0])
        mus_s.append(w._self_gaussian.mu[0]); sig_s.append(w._self_gaussian.sigma[0, 0])
    (ds, as_), (dm, am) = _niw_marginals_1d(0.3, 1.4, 5.0, 0.7)
    check_law(sig, ds, as_)
    check_law(mus, dm, am, mean=0.3)
    check_law(sig_s, ds, as_)
    check_law(mus_s, dm, am, mean=0.3)


def test_geweke_niw_weights_two_dimensional(abi):
    """B = 2 (the Bartlett draw): marginals of the NIW prior — Sigma_11 ~ InvGamma((nu-1)/2, S_11/2), mu_1 Student-t."""
    from graphistician_amd.weights import NIWGaussianWeightDistribution
    np.random.seed(14)
    N, iters = 4, 6000
    A = np.zeros((N, N), dtype=np.uint8)
    A[0, 1] = A[1, 2] = A[2, 0] = A[3, 0] = A[1, 3] = 1                     # 5 edges
    S0 = np.array([[1.2, 0.4], [0.4, 0.9]])
    w = NIWGaussianWeightDistribution(N, B=2, mu_0=np.zeros(2), Sigma_0=S0, nu_0=6.0, kappa_0=1.0, seed=6)
    s11, mu1, rho = [], [], []
    for it in range(iters):
        w.resample((A, w.rvs()))
        S = w._gaussian.sigma
        s11.append(S[1, 1]); mu1.append(w._gaussian.mu[1]); rho.append(S[0, 1] / np.sqrt(S[0, 0] * S[1, 1]))
    check_law(s11, "invgamma", ((6.0 - 1.0) / 2.0, 0.0, 0.9 / 2.0))
    check_law(mu1, "t", (6.0 - 1.0, 0.0, np.sqrt(0.9 / (5.0 * 1.0))), mean=0.0)
    # the correlation of an inverse-Wishart draw is centred on that of its scale matrix
    assert abs(np.median(rho[200:]) - 0.4 / np.sqrt(1.2 * 0.9)) < 0.05


def test_geweke_sbm_gaussian_weights_on_the_engine(abi):
    from graphistician_amd.weights import SBMGaussianWeightDistribution
    np.random.seed(15)
    N, C, iters = 6, 2, 6000
    A = (np.random.rand(N, N) < 0.3).astype(np.uint8)
    w = SBMGaussianWeightDistribution(N, B=1, C=C, pi=3.0, mu_0=np.array([0.0]), Sigma_0=np.array([[1.0]]), nu_0=4.0,
                                      kappa_0=0.5, seed=8)
    mu01, s10, m0, c3 = [], [], [], []
    for it in range(iters):
        w.resample((A, w.rvs()))
        mu01.append(w._gaussians[0][1].mu[0]); s10.append(w._gaussians[1][0].sigma[0, 0]); m0.append(w.m[0]); c3.append(w.c[3])
    (ds, as_), (dm, am) = _niw_marginals_1d(0.0, 1.0, 4.0, 0.5)
    check_law(s10, ds, as_)
    check_law(mu01, dm, am, mean=0.0)
    check_law(m0, "beta", (3.0, 3.0), mean=0.5)
    assert abs(batch_z(np.array(c3[200:], dtype=float), 0.5)) < 4.5


# ------------------------------------------------------------------ Geweke: eigenmodels (eigenmodel.py:836-965, 246-379)
@pytest.mark.parametrize("cls_name", ["LogisticEigenmodel", "ProbitEigenmodel"])
def test_geweke_eigenmodel_on_the_engine(abi, cls_name):
    from graphistician_amd import eigenmodel as E
    np.random.seed(16)
    N, D, iters = 5, 2, 8000
    m = getattr(E, cls_name)(N, D, p=0.3, sigma_mu0=0.8, sigma_F=1.3, sigma_lmbda=0.9, seed=99)
    mu0, lam, f00, f31 = [], [], [], []
    for it in range(iters):
        m.resample(m.rvs())
        mu0.append(m.mu_0); lam.append(m.lmbda[1]); f00.append(m.F[0, 0]); f31.append(m.F[3, 1])
    check_law(mu0, "norm", (m.mu_mu_0, np.sqrt(0.8)), thin=12, mean=m.mu_mu_0)
    check_law(lam, "norm", (0.0, np.sqrt(0.9)), thin=12, mean=0.0)
    check_law(f00, "norm", (0.0, np.sqrt(1.3)), thin=12, mean=0.0)
    check_law(f31, "norm", (0.0, np.sqrt(1.3)), thin=12, mean=0.0)


# ------------------------------------------------------------------ the reference's asserted test, on the device
def test_truncated_normal_moments_match_scipy_truncnorm(abi):
    """test/test_distributions.py:8-36: E and Var of TN(mu, 1; lb, ub) against scipy.stats.truncnorm — here for the
    sampler the probit eigenmodel uses (spec/gg_probit.h, internals/utils.py:53-76), drawn on the device: lb = 0
    (A = 1) and ub = 0 (A = 0); the reference's ub = 10 at mu = 3 is 7 sigma out, i.e. the half line."""
    n = 400000
    for mu, a in [(0.0, 1), (3.0, 1), (3.0, 0), (-1.5, 1), (0.7, 0)]:
        x = abi.selftest_draws(5, 31, [mu, a], n)
        ref = stats.truncnorm(0.0 - mu, np.inf, loc=mu) if a else stats.truncnorm(-np.inf, 0.0 - mu, loc=mu)
        assert np.all(x > 0) if a else np.all(x < 0)
        se = np.sqrt(ref.var() / n)
        assert abs(x.mean() - ref.mean()) < 5 * se, (mu, a, x.mean(), ref.mean())
        assert abs(x.var() - ref.var()) < 5 * ref.var() * np.sqrt(2.0 / n) * 2.0, (mu, a, x.var(), ref.var())
        assert stats.kstest(x[::40], ref.cdf).pvalue > 1e-3


def test_sign_then_truncated_normal_is_the_normal(abi):
    """test/test_distributions.py:39-56: A ~ Bern(1 - Phi(-mu)), Z | A ~ TN(mu, 1) on the matching half line => Z ~ N(mu, 1)."""
    for mu in (1.0, -0.4):
        z = abi.selftest_draws(5, 37, [mu, -1], 200000)
        assert stats.kstest(z[::20], "norm", args=(mu, 1.0)).pvalue > 1e-3
        assert abs(z.mean() - mu) < 5 / np.sqrt(z.size) and abs(z.var() - 1.0) < 0.02


# ------------------------------------------------------------------ primitive draws on the device vs scipy
def test_device_draw_laws(abi):
    n = 40000
    for a in (0.3, 1.0, 4.5, 60.0):
        assert stats.kstest(abi.selftest_draws(1, 3, [a], n), "gamma", args=(a,)).pvalue > 1e-3
    assert stats.kstest(abi.selftest_draws(2, 4, [2.5, 0.7], n), "beta", args=(2.5, 0.7)).pvalue > 1e-3
    alpha = np.array([0.5, 2.0, 7.0])
    d = abi.selftest_draws(3, 5, alpha, n, stride=3)
    assert np.allclose(d.sum(1), 1.0)
    for k in range(3):
        assert stats.kstest(d[:, k], "beta", args=(alpha[k], alpha.sum() - alpha[k])).pvalue > 1e-3
    assert stats.kstest(abi.selftest_draws(0, 6, [], n), "norm").pvalue > 1e-3


def test_device_niw_posterior_law(abi):
    """NIW posterior draw on the device: Sigma ~ IW(S_n, nu_n), mu | Sigma ~ N(mu_n, Sigma / kappa_n) (the conjugate update
    behind weights.py:161-162,408,413), checked through E[Sigma] = S_n / (nu_n - B - 1) and the Student-t law of mu."""
    B, n_obs = 2, 30
    rng = np.random.RandomState(2)
    X = rng.randn(n_obs, B) * [1.0, 2.0] + [0.5, -1.0]
    mu_0, S_0, nu_0, k_0 = np.array([0.2, 0.1]), np.array([[1.0, 0.2], [0.2, 2.0]]), 5.0, 0.8
    prm = np.concatenate([[B], mu_0, S_0.ravel(), [nu_0, k_0], [float(n_obs)], X.sum(0), X.T.dot(X).ravel()])
    out = abi.selftest_draws(4, 17, prm, 60000, stride=B + B * B)
    mu, Sig = out[:, :B], out[:, B:].reshape(-1, B, B)
    xbar = X.mean(0)
    k_n, nu_n = k_0 + n_obs, nu_0 + n_obs
    mu_n = (k_0 * mu_0 + n_obs * xbar) / k_n
    S_n = S_0 + (X - xbar).T.dot(X - xbar) + k_0 * n_obs / k_n * np.outer(xbar - mu_0, xbar - mu_0)
    assert np.allclose(Sig.mean(0), S_n / (nu_n - B - 1), rtol=0.02)
    assert np.allclose(mu.mean(0), mu_n, atol=4 * np.sqrt(S_n.diagonal() / (nu_n - B - 1) / k_n / 60000) + 1e-3)
    df = nu_n - B + 1
    for j in range(B):
        scale = np.sqrt(S_n[j, j] / (k_n * df))
        assert stats.kstest(mu[::3, j], "t", args=(df, mu_n[j], scale)).pvalue > 1e-3


def test_device_pg1_moments(abi):
    """PG(1, z) drawn on the device: E = tanh(z/2)/(2z) (the reference's own anchor, eigenmodel.py:1062-1063) and the variance."""
    n = 200000
    for z in (0.0, 0.8, -3.0, 12.0):
        x = abi.selftest_draws(6, 23, [z], n)
        za = abs(z)
        mean = 0.25 if za == 0 else np.tanh(za / 2) / (2 * za)
        var = 1 / 24.0 if za == 0 else (np.sinh(za) - za) / (4 * za ** 3 * np.cosh(za / 2) ** 2)
        assert np.all(x > 0)
        assert abs(x.mean() - mean) < 5 * np.sqrt(var / n)
        assert abs(x.var() - var) < 0.03 * var


def test_device_categorical_frequencies(abi):
    """The categorical draw of adjacency.py:695 / weights.py:488 (pybasicbayes sample_discrete_from_log semantics): the
    frequencies of the device's draws must be softmax(lp) (chi-square test), including a -inf entry and K not a power of two."""
    n = 300000
    for lp in ([0.0, -1.0, 2.0], [-700.0, -701.5, -699.0, -705.0, -700.2], [1.0, -np.inf, 0.0, 0.5, 0.5, -3.0, 2.0]):
        lp = np.array(lp)
        idx = abi.selftest_draws(7, 41, lp, n).astype(int)
        pr = np.exp(lp - lp.max()); pr /= pr.sum()
        cnt = np.bincount(idx, minlength=len(lp))
        assert cnt[pr == 0].sum() == 0
        keep = pr > 0
        chi2 = ((cnt[keep] - n * pr[keep]) ** 2 / (n * pr[keep])).sum()
        assert stats.chi2.sf(chi2, keep.sum() - 1) > 1e-4, (lp, cnt, n * pr)


# ------------------------------------------------------------------ Geweke: joint SBM (stochastic_block_model.py:206-309)
def test_geweke_joint_sbm_on_the_engine(abi):
    """GaussianStochasticBlockModel (one label vector for adjacency and weights): alternate (A, W) ~ p(A, W | theta) and the
    engine's joint sweep; p, m, mu and Sigma must keep their priors."""
    from graphistician_amd.stochastic_block_model import GaussianStochasticBlockModel
    np.random.seed(21)
    N, C, iters = 6, 2, 6000
    m = GaussianStochasticBlockModel(N, 1, C=C, pi=3.0, tau0=1.2, tau1=0.9, seed=77,
                                     weight_hypers=dict(mu_0=np.zeros(1), Sigma_0=np.eye(1), nu_0=4.0, kappa_0=0.5))
    p01, m0, mu10, s00 = [], [], [], []
    for it in range(iters):
        m.resample(m.rvs())
        p01.append(m.p[0, 1]); m0.append(m.m[0]); mu10.append(m.weight_models[1][0].mu[0]); s00.append(m.weight_models[0][0].sigma[0, 0])
    (ds, as_), (dm, am) = _niw_marginals_1d(0.0, 1.0, 4.0, 0.5)
    check_law(p01, "beta", (0.9, 1.2), mean=0.9 / 2.1)
    check_law(m0, "beta", (3.0, 3.0), mean=0.5)
    check_law(mu10, dm, am, mean=0.0)
    check_law(s00, ds, as_)
