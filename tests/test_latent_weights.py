"""Latent-distance Gaussian weight model (graphistician/weights.py:613-903, the weight-side half of SURVEY.md §8f
rank 3): CPU tests of the oracle (log probability and log-likelihood vs numpy / scipy, hand-derived gradient vs finite
differences, the conjugate draws against their closed-form conditionals) and GPU parity of the HIP engine vs the
oracle (float tolerance: sums are ordered differently)."""
import numpy as np
import pytest
from scipy import stats

from oracle import oracle as O

PRIOR = np.array([1.0, 3.0])                 # lmbda_0, nu_0        (weights.py:632-638, B = 1)
PRIOR_SELF = np.array([0.0, 1.0, 3.0, 1.0])  # mu_0, sigma_0, nu_0, kappa_0 (weights.py:641-644)


def make_case(N, dim, seed, density=0.5):
    rng = np.random.RandomState(seed)
    Lt = rng.randn(N, dim) * 0.6
    A = (rng.rand(N, N) < density).astype(np.uint8)
    D = ((Lt[:, None, :] - Lt[None, :, :]) ** 2).sum(2)
    W = -D + 0.5 + 0.3 * rng.randn(N, N)
    W[np.arange(N), np.arange(N)] = 1.5 + 0.2 * rng.randn(N)
    q, shift = O.quantise(A, W[:, :, None])
    return A, np.ascontiguousarray(W[:, :, None]), q, shift, np.ascontiguousarray(rng.randn(N, dim) * 0.3)


def lp_numpy(A, Wq, L, b, sigma, eta, beta=1.0):
    """weights.py:703-733 written out with numpy (autograd.numpy in the reference)."""
    N = L.shape[0]
    Mu = -((L[:, None, :] - L[None, :, :]) ** 2).sum(2) + b
    X = (Wq - Mu) * (A * (1 - np.eye(N)))
    return beta * (-0.5 * X ** 2 / sigma).sum() - 0.5 * (L * L / eta).sum() - 0.5 * b ** 2


def test_oracle_logprob_gradient_and_loglik():
    N, dim = 12, 2
    A, W, q, shift, L = make_case(N, dim, 0)
    Wq = O.dequantise(q, shift)[:, :, 0]
    b, sigma, eta, beta = 0.4, 0.7, 0.2, 0.8
    lp, g = O.ldw_logprob(A, q, shift, L, b, sigma, eta, beta, grad=True)
    assert abs(lp - lp_numpy(A, Wq, L, b, sigma, eta, beta)) < 1e-10
    eps = 1e-6
    for i in range(N):
        for d in range(dim):
            Lp, Lm = L.copy(), L.copy()
            Lp[i, d] += eps; Lm[i, d] -= eps
            fd = (lp_numpy(A, Wq, Lp, b, sigma, eta, beta) - lp_numpy(A, Wq, Lm, b, sigma, eta, beta)) / (2 * eps)
            assert abs(fd - g[i, d]) < 1e-6
    # abstractions.py:257-274 with Mu, Sigma of weights.py:649-670
    mus, sgs = 1.4, 0.3
    Mu = -((L[:, None, :] - L[None, :, :]) ** 2).sum(2) + b
    ref = 0.0
    for i in range(N):
        for j in range(N):
            if A[i, j]:
                ref += stats.norm.logpdf(Wq[i, j], mus if i == j else Mu[i, j], np.sqrt(sgs if i == j else sigma))
    assert abs(O.ldw_loglik(A, q, shift, L, b, sigma, mus, sgs) - ref) < 1e-9


def test_oracle_conjugate_draws_follow_their_conditionals():
    N, dim = 14, 2
    A, W, q, shift, L = make_case(N, dim, 1)
    Wq = O.dequantise(q, shift)[:, :, 0]
    off = (A != 0) & ~np.eye(N, dtype=bool)
    D = ((L[:, None, :] - L[None, :, :]) ** 2).sum(2)
    sigma, beta = 0.6, 0.7
    X = (Wq + D)[off]
    sp = 1.0 / (1.0 + beta * X.size / sigma)
    mp = sp * (beta * X.sum() / sigma)
    bs, sgs, etas, mss = [], [], [], []
    for sw in range(3000):
        Lc = L.copy()
        b, eta, sg, ms, ss, _ = O.ldw_resample(A, q, shift, Lc, 0.1, 0.05, sigma, 0.0, 1.0, PRIOR, PRIOR_SELF, 11, sw,
                                               parts=O.LDW_PART_B | O.LDW_PART_COV | O.LDW_PART_ETA, beta=beta)
        assert np.array_equal(Lc, L)
        bs.append(b); sgs.append(sg); etas.append(eta); mss.append(ms)
    # b | X, sigma ~ N(mp, sp)  (weights.py:833-856)
    assert stats.kstest((np.array(bs) - mp) / np.sqrt(sp), "norm").pvalue > 1e-3
    # eta | L ~ IG(1 + N dim / 2, 1 + sum L^2 / 2)  (weights.py:870-886)
    assert stats.kstest(etas, stats.invgamma(1 + 0.5 * L.size, scale=1 + 0.5 * (L ** 2).sum()).cdf).pvalue > 1e-3
    # sigma | residuals(b) ~ IG((nu_0 + beta n) / 2, (lmbda_0 + beta sum r^2) / 2): b varies from draw to draw, so test
    # the probability integral transform with each draw's own b
    u = []
    for b, sg in zip(bs, sgs):
        r = X - b
        u.append(stats.invgamma(0.5 * (PRIOR[1] + beta * r.size), scale=0.5 * (PRIOR[0] + beta * (r ** 2).sum())).cdf(sg))
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    # the self Gaussian's mean is centred on the NIW posterior mean (kappa_0 mu_0 + beta s1) / (kappa_0 + beta n)
    ws = Wq[np.arange(N), np.arange(N)][A[np.arange(N), np.arange(N)] != 0]
    mu_n = (PRIOR_SELF[3] * PRIOR_SELF[0] + beta * ws.sum()) / (PRIOR_SELF[3] + beta * ws.size)
    assert abs(np.mean(mss) - mu_n) < 0.05


def test_oracle_hmc_leaves_the_prior_invariant():
    """beta = 0: the L update targets N(0, eta I) whatever the data."""
    N, dim, eta = 4, 2, 0.05
    A, W, q, shift, L = make_case(N, dim, 2)
    L *= 0.5
    xs, accs = [], 0
    for sw in range(4000):
        _, _, _, _, _, acc = O.ldw_resample(A, q, shift, L, 0.5, eta, 1.0, 0.0, 1.0, PRIOR, PRIOR_SELF, 3, sw,
                                            parts=O.LDW_PART_L, beta=0.0)
        accs += acc
        xs.append(L[0, 0])
    assert accs > 3600            # step 0.005 on a scale sqrt(eta) = 0.22: nearly always accepted
    x = np.array(xs[500:])
    assert abs(x.mean()) < 0.08 and 0.5 * eta < x.var() < 1.6 * eta


@pytest.mark.gpu
@pytest.mark.parametrize("N,dim,beta,density", [(40, 2, 1.0, 0.5), (130, 3, 1.0, 0.2), (97, 1, 0.4, 0.7), (260, 4, 1.0, 0.05), (90, 6, 1.0, 0.3), (120, 8, 0.7, 0.2)])
def test_engine_parts_match_oracle(N, dim, beta, density):
    from graphistician_amd import _abi as abi
    A, W, q, shift, L = make_case(N, dim, N, density)
    d = abi.DeviceData(A, W)
    m = abi.Model(kind=abi.LATENT_DIST_W, N=N, D=dim, K=1, B=1, seed=77)
    b, eta, sg, ms, ss = 0.3, 0.2, 0.6, 1.2, 0.4
    m.set(abi.ST_F, L); m.set(abi.ST_MU0, np.array([b])); m.set(abi.ST_LD_AUX, np.array([eta, sg, ms, ss]))
    m.set(abi.ST_NIW, np.array([0.0, PRIOR[0], PRIOR[1], 0.0])); m.set(abi.ST_NIW_SELF, PRIOR_SELF)
    ll = m.log_likelihood(d)
    ref = O.ldw_loglik(A, q, shift, L, b, sg, ms, ss)
    assert abs(ll - ref) <= 1e-10 * abs(ref)
    Lo = L.copy()
    # L (HMC)
    m.resample_part(d, abi.PART_F, beta)
    *_, acc = O.ldw_resample(A, q, shift, Lo, b, eta, sg, ms, ss, PRIOR, PRIOR_SELF, 77, 0, parts=O.LDW_PART_L, beta=beta)
    assert np.allclose(m.get(abi.ST_F, (N, dim)), Lo, rtol=1e-9, atol=1e-11)
    # b
    m.resample_part(d, abi.PART_MU0, beta)
    bo, *_ = O.ldw_resample(A, q, shift, Lo, b, eta, sg, ms, ss, PRIOR, PRIOR_SELF, 77, 0, parts=O.LDW_PART_B, beta=beta)
    assert abs(m.get(abi.ST_MU0, (1,))[0] - bo) < 1e-10
    # sigma + self Gaussian
    m.resample_part(d, abi.PART_PARAMS, beta)
    _, _, sgo, mso, sso, _ = O.ldw_resample(A, q, shift, Lo, bo, eta, sg, ms, ss, PRIOR, PRIOR_SELF, 77, 0,
                                            parts=O.LDW_PART_COV, beta=beta)
    aux = m.get(abi.ST_LD_AUX, (4,))
    assert np.allclose(aux[1:], [sgo, mso, sso], rtol=1e-9)
    # eta
    m.resample_part(d, abi.PART_LMBDA, beta)
    _, etao, *_ = O.ldw_resample(A, q, shift, Lo, bo, eta, sgo, mso, sso, PRIOR, PRIOR_SELF, 77, 0, parts=O.LDW_PART_ETA, beta=beta)
    assert abs(m.get(abi.ST_LD_AUX, (4,))[0] - etao) <= 1e-10 * etao


@pytest.mark.gpu
def test_engine_chain_matches_oracle_and_fits():
    """Full resamples through the reference-shaped class; the chain must also do its job (the fit improves)."""
    from graphistician_amd.weights import LatentDistanceGaussianWeightDistribution
    N, dim = 80, 2
    A, W, q, shift, L = make_case(N, dim, 5)
    np.random.seed(0)
    m = LatentDistanceGaussianWeightDistribution(N, dim=dim, b=0.2, sigma=np.array([[0.8]]), eta=0.3, seed=999)
    m.L = L
    ms, ss = float(m._self_gaussian.mu[0]), float(m._self_gaussian.sigma[0, 0])
    ll0 = m.log_likelihood((A, W))
    ref0 = O.ldw_loglik(A, q, shift, L, 0.2, 0.8, ms, ss)
    assert abs(ll0 - ref0) <= 1e-10 * abs(ref0)
    Lo, b, eta, sg = L.copy(), 0.2, 0.3, 0.8
    for s in range(5):
        m.resample((A, W))
        b, eta, sg, ms, ss, _ = O.ldw_resample(A, q, shift, Lo, b, eta, sg, ms, ss, PRIOR, PRIOR_SELF, 999, s)
    assert np.allclose(m.L, Lo, rtol=1e-7, atol=1e-9)
    assert abs(m.b - b) < 1e-8 and abs(m.eta - eta) < 1e-8 * eta and abs(m.cov.sigma[0, 0] - sg) < 1e-8 * sg
    assert abs(m._self_gaussian.mu[0] - ms) < 1e-8 and abs(m._self_gaussian.sigma[0, 0] - ss) < 1e-8 * ss
    for s in range(300):
        m.resample((A, W))
    assert m.log_likelihood((A, W)) > ll0 + 100.0
    assert np.isfinite(m.log_prior())
    assert m.Mu.shape == (N, N, 1) and m.Sigma.shape == (N, N, 1, 1)
    # the data were generated with sigma = 0.09 around -D + 0.5: the residual variance must have come down towards it
    assert m.cov.sigma[0, 0] < 0.5


@pytest.mark.gpu
def test_initialize_hypers_optimises_the_locations():
    """weights.py:676-678, 792-808: initialize_hypers = BFGS on -log p(W, L | b, sigma, eta) with every pair counted as an edge.
    The written-out gradient against central differences of the objective (the reference differentiates the same function
    with autograd), and the optimum: a complete weighted graph generated from planted locations gives their pairwise
    distances back."""
    from graphistician_amd.weights import LatentDistanceGaussianWeightDistribution
    rng = np.random.RandomState(3)
    N, dim, b = 36, 2, 0.5
    Lt = 0.7 * rng.randn(N, dim)
    Dt = ((Lt[:, None, :] - Lt[None, :, :]) ** 2).sum(2)
    W = (-Dt + b + 0.05 * rng.randn(N, N))[:, :, None]
    np.random.seed(1)
    m = LatentDistanceGaussianWeightDistribution(N, dim=dim, b=b, sigma=np.array([[0.05]]), eta=1.0, seed=5)
    A = np.ones((N, N))
    x = rng.randn(N * dim) * 0.5
    f0, g0 = m._location_objective(x, A, W)
    for i in rng.choice(N * dim, 12, replace=False):
        e = np.zeros(N * dim); e[i] = 1e-5
        fd = (m._location_objective(x + e, A, W)[0] - m._location_objective(x - e, A, W)[0]) / 2e-5
        assert abs(fd - g0[i]) <= 1e-5 * max(1.0, abs(fd)), (i, fd, g0[i])
    f_start = m._location_objective(np.ravel(m.L), A, W)[0]
    m.initialize_hypers(W)
    f_end = m._location_objective(np.ravel(m.L), A, W)[0]
    assert f_end < 0.05 * f_start
    off = ~np.eye(N, dtype=bool)
    assert np.corrcoef(m.D[off], Dt[off])[0, 1] > 0.98
