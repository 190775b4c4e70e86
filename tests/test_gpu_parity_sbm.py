"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bit-exact for labels, counts and — because statistics are
exact integers and the arithmetic spec is shared — also for the drawn parameters."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import planted_sbm, er_network, AdjState, WState

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def abi():
    from graphistician_amd import _abi
    if _abi.device_count() < 1:
        pytest.fail("no HIP device: the engine has no CPU fallback")
    return _abi


def make_adj(abi, st, seed, special=1, allow=1):
    m = abi.Model(kind=abi.SBM_ADJ, N=st.N, K=st.K, B=1, seed=seed, tau1=st.tau1, tau0=st.tau0,
                  allow_self_connections=allow, special_case_self_conns=special)
    m.set(abi.ST_C, st.c, np.int32)
    m.set(abi.ST_M, st.m)
    m.set(abi.ST_PI, st.pi)
    m.set(abi.ST_P, st.p)
    m.set(abi.ST_P_SELF, np.array([st.p_self]))
    return m


def make_w(abi, st, seed, special=1):
    m = abi.Model(kind=abi.SBM_GAUSS_W, N=st.N, K=st.K, B=st.B, seed=seed, special_case_self_conns=special)
    m.set(abi.ST_C, st.c, np.int32)
    m.set(abi.ST_M, st.m)
    m.set(abi.ST_PI, st.pi)
    m.set(abi.ST_MU, st.mu)
    m.set(abi.ST_SIGMA, st.sigma)
    m.set(abi.ST_MU_SELF, st.mu_self)
    m.set(abi.ST_SIGMA_SELF, st.sigma_self)
    m.set(abi.ST_NIW, st.niw)
    m.set(abi.ST_NIW_SELF, st.niw_self)
    return m


# ------------------------------------------------------------------ arithmetic spec on the device
def test_device_math_bitexact(abi):
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.uniform(-745, 709, 200000), rng.uniform(-2, 2, 200000), rng.uniform(-60, 0, 100000)])
    assert np.array_equal(abi.selftest_math(0, x), O.exp_vec(x))
    y = np.exp(rng.uniform(-700, 700, 300000))
    y = np.concatenate([y, rng.uniform(0.4, 2.5, 200000)])
    assert np.array_equal(abi.selftest_math(1, y), O.log_vec(y))
    assert np.array_equal(abi.selftest_math(2, y), np.sqrt(y))
    a = rng.randn(300000) * 10 ** rng.uniform(-5, 5, 300000)
    b = rng.randn(300000) * 10 ** rng.uniform(-5, 5, 300000)
    assert np.array_equal(abi.selftest_math(3, a, b), a / b)
    big = np.round(rng.uniform(-2 ** 62, 2 ** 62, 100000))
    assert np.array_equal(abi.selftest_math(5, big), big.astype(np.int64).astype(np.float64))


def test_device_draws_bitexact(abi):
    for a in [0.3, 1.0, 4.5, 2000.0]:
        assert np.array_equal(abi.selftest_draws(1, 7, [a], 20000), O.gamma_vec(7, a, 20000))
    assert np.array_equal(abi.selftest_draws(2, 9, [3.0, 40.5], 20000), O.beta_vec(9, 3.0, 40.5, 20000))
    alpha = np.array([0.5, 1.0, 7.0, 100.0, 3.0])
    assert np.array_equal(abi.selftest_draws(3, 11, alpha, 5000, stride=5), O.dirichlet_vec(11, alpha, 5000))
    for B in [1, 2, 3, 4]:
        rng = np.random.RandomState(B)
        X = rng.randn(50, B) + 1.0
        niw = O.pack_niw(rng.randn(B), np.eye(B) * 1.5, B + 2, 0.7)
        s1, s2 = X.sum(0), X.T.dot(X)
        prm = np.concatenate([[B], niw, [50.0], s1, s2.ravel()])
        out = abi.selftest_draws(4, 13, prm, 4000, stride=B + B * B)
        mu, sig = O.niw_vec(13, niw, B, 50.0, s1, s2, 4000)
        assert np.array_equal(out[:, :B], mu)
        assert np.array_equal(out[:, B:].reshape(-1, B, B), sig)


# ------------------------------------------------------------------ SBM adjacency
@pytest.mark.parametrize("N,K,special", [(50, 3, 1), (193, 7, 1), (128, 16, 0), (300, 64, 1), (300, 100, 1), (260, 256, 0)])
def test_adj_node_lp_bitexact(abi, N, K, special):
    A, W, _ = planted_sbm(N, K, seed=N)
    st = AdjState(N, K, seed=K)
    d = abi.DeviceData(A, None)
    m = make_adj(abi, st, seed=5, special=special)
    for n in [0, 1, N // 2, N - 1]:
        lp_g = m.node_lp(d, n, K)
        lp_o = O.sbm_adj_node_lp(A, st.c, st.m, st.p, special, n)
        assert np.array_equal(lp_g, lp_o), (n, lp_g - lp_o)


@pytest.mark.parametrize("N,K,special", [(50, 3, 1), (193, 7, 0), (257, 16, 1)])
def test_adj_sweep_parts_bitexact(abi, N, K, special):
    A, W, _ = planted_sbm(N, K, seed=3 * N)
    st = AdjState(N, K, seed=K + 1)
    d = abi.DeviceData(A, None)
    m = make_adj(abi, st, seed=77, special=special)
    # p
    m.resample_part(d, abi.PART_PARAMS)
    ps = O.sbm_adj_sweep(A, st.c, st.m, st.p, st.p_self, st.pi, st.tau1, st.tau0, special, 77, 0, O.PART_PARAMS)
    assert np.array_equal(m.get(abi.ST_P, (K, K)), st.p)
    if special:
        assert m.get(abi.ST_P_SELF, (1,))[0] == ps
    # c
    m.resample_part(d, abi.PART_C)
    O.sbm_adj_sweep(A, st.c, st.m, st.p, ps, st.pi, st.tau1, st.tau0, special, 77, 0, O.PART_C)
    assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c)
    # m
    m.resample_part(d, abi.PART_M)
    O.sbm_adj_sweep(A, st.c, st.m, st.p, ps, st.pi, st.tau1, st.tau0, special, 77, 0, O.PART_M)
    assert np.array_equal(m.get(abi.ST_M, (K,)), st.m)


@pytest.mark.parametrize("N,K", [(120, 4), (400, 16), (300, 40), (520, 64)])
def test_adj_chain_bitexact(abi, N, K):
    """Several full sweeps from a random start: burn-in (many label changes per scan) then steady state."""
    A, W, _ = planted_sbm(N, K, seed=11)
    st = AdjState(N, K, seed=2)
    d = abi.DeviceData(A, None)
    m = make_adj(abi, st, seed=1234)
    ps = st.p_self
    nsweeps = 6
    m.resample(d, nsweeps)
    for s in range(nsweeps):
        ps = O.sbm_adj_sweep(A, st.c, st.m, st.p, ps, st.pi, st.tau1, st.tau0, 1, 1234, s)
    assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c)
    assert np.array_equal(m.get(abi.ST_P, (K, K)), st.p)
    assert np.array_equal(m.get(abi.ST_M, (K,)), st.m)
    assert m.get(abi.ST_P_SELF, (1,))[0] == ps
    assert m.stats()["label_changes"] > 0
    ll_g = m.log_likelihood(d)
    ll_o = O.sbm_adj_loglik(A, st.c, st.p, ps)
    assert abs(ll_g - ll_o) <= 1e-10 * abs(ll_o)


# ------------------------------------------------------------------ SBM Gaussian weights
@pytest.mark.parametrize("N,K,B,special", [(50, 3, 1, 1), (130, 5, 2, 1), (96, 16, 1, 0), (70, 4, 4, 1), (200, 64, 1, 1), (200, 70, 2, 1), (200, 200, 1, 0)])
def test_w_node_lp_bitexact(abi, N, K, B, special):
    A, W, _ = planted_sbm(N, K, B=B, seed=N + B)
    q, shift = O.quantise(A, W)
    st = WState(N, K, B, seed=K)
    d = abi.DeviceData(A, W)
    assert d.info()["shift"] == shift
    m = make_w(abi, st, seed=5, special=special)
    for n in [0, N // 3, N - 1]:
        lp_g = m.node_lp(d, n, K)
        lp_o = O.sbm_w_node_lp(A, q, shift, st.c, st.m, st.mu, st.sigma, special, n)
        assert np.array_equal(lp_g, lp_o), (n, lp_g - lp_o)


@pytest.mark.parametrize("N,K,B,special", [(60, 3, 1, 1), (150, 6, 2, 0), (260, 16, 1, 1), (300, 48, 1, 1), (260, 40, 2, 0),
                                           (330, 24, 2, 1), (170, 5, 3, 1), (301, 20, 3, 0), (140, 40, 3, 1)])
def test_w_chain_bitexact(abi, N, K, B, special):
    A, W, _ = planted_sbm(N, K, B=B, seed=5 * N)
    q, shift = O.quantise(A, W)
    st = WState(N, K, B, seed=K + 3)
    d = abi.DeviceData(A, W)
    m = make_w(abi, st, seed=4321, special=special)
    nsweeps = 5
    m.resample(d, nsweeps)
    for s in range(nsweeps):
        O.sbm_w_sweep(A, q, shift, st.c, st.m, st.mu, st.sigma, st.mu_self, st.sigma_self, st.pi, st.niw, st.niw_self,
                      special, 4321, s)
    assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c)
    assert np.array_equal(m.get(abi.ST_MU, (K, K, B)), st.mu)
    assert np.array_equal(m.get(abi.ST_SIGMA, (K, K, B, B)), st.sigma)
    assert np.array_equal(m.get(abi.ST_M, (K,)), st.m)
    if special:
        assert np.array_equal(m.get(abi.ST_MU_SELF, (B,)), st.mu_self)
    ll_g = m.log_likelihood(d)
    ll_o = O.sbm_w_loglik(A, q, shift, st.c, st.mu, st.sigma, st.mu_self, st.sigma_self, special)
    assert abs(ll_g - ll_o) <= 1e-9 * abs(ll_o)


def test_bb_and_niw_config1(abi):
    """Config 1 shape: Beta-Bernoulli adjacency + NIW weights, N=50, B=2."""
    A, W = er_network(50, B=2, seed=1)
    q, shift = O.quantise(A, W)
    d = abi.DeviceData(A, W)
    for selfc in [0, 1]:
        m = abi.Model(kind=abi.BB_ADJ, N=50, K=1, B=1, seed=99, tau1=1.0, tau0=1.0, special_case_self_conns=selfc,
                      allow_self_connections=1)
        for s in range(3):
            m.resample(d, 1)
            p, ps = O.bb_adj_sweep(A, 1.0, 1.0, selfc, 99, s)
            assert m.get(abi.ST_P, (1, 1))[0, 0] == p
            if selfc:
                assert m.get(abi.ST_P_SELF, (1,))[0] == ps
        ll = m.log_likelihood(d)
        assert abs(ll - O.const_adj_loglik(A, p, ps if selfc else p)) <= 1e-10 * abs(ll)
    B = 2
    w = abi.Model(kind=abi.NIW_W, N=50, K=1, B=B, seed=5)
    niw = O.pack_niw(np.zeros(B), np.eye(B), B + 2, 1.0)
    w.set(abi.ST_NIW, niw)
    w.set(abi.ST_NIW_SELF, niw)
    mu, sig, mus, sigs = np.zeros((1, 1, B)), np.zeros((1, 1, B, B)), np.zeros(B), np.zeros((B, B))
    for s in range(3):
        w.resample(d, 1)
        O.niw_w_sweep(A, q, shift, mu, sig, mus, sigs, niw, niw, 5, s)
        assert np.array_equal(w.get(abi.ST_MU, (1, 1, B)), mu)
        assert np.array_equal(w.get(abi.ST_SIGMA, (1, 1, B, B)), sig)
        assert np.array_equal(w.get(abi.ST_MU_SELF, (B,)), mus)
        assert np.array_equal(w.get(abi.ST_SIGMA_SELF, (B, B)), sigs)
    ll = w.log_likelihood(d)
    ll_o = O.sbm_w_loglik(A, q, shift, np.zeros(50, np.int32), mu, sig, mus, sigs, 1)
    assert abs(ll - ll_o) <= 1e-9 * abs(ll_o)


# ------------------------------------------------------------------ joint SBM (stochastic_block_model.py:202-309)
@pytest.mark.parametrize("N,K,B", [(60, 3, 1), (200, 8, 1), (260, 16, 2), (300, 40, 1), (150, 33, 2), (400, 70, 1), (520, 100, 2),
                                   (700, 200, 1)])
def test_joint_sbm_chain_bitexact(abi, N, K, B):
    """GG_SBM_JOINT: one label vector for adjacency and weights.  Conditionals, labels, p, (mu, Sigma), m bit-exact against
    the oracle's restatement over several sweeps from a random start; step-wise parts; log-likelihood vs the per-pair sum."""
    A, W, _ = planted_sbm(N, K, B=B, seed=7 * N)
    np.fill_diagonal(A, (np.arange(N) % 2).astype(np.uint8))
    q, shift = O.quantise(A, W)
    st = WState(N, K, B, seed=K + 9)
    rng = np.random.RandomState(N)
    p = np.ascontiguousarray(rng.beta(1.0, 1.0, size=(K, K)))
    d = abi.DeviceData(A, W)
    m = abi.Model(kind=abi.SBM_JOINT, N=N, K=K, B=B, seed=606, tau1=1.3, tau0=0.9)
    m.set(abi.ST_C, st.c, np.int32); m.set(abi.ST_M, st.m); m.set(abi.ST_PI, st.pi); m.set(abi.ST_P, p)
    m.set(abi.ST_MU, st.mu); m.set(abi.ST_SIGMA, st.sigma); m.set(abi.ST_NIW, st.niw)
    for n1 in (0, N // 2, N - 1):
        assert np.array_equal(m.node_lp(d, n1, K), O.sbm_joint_node_lp(A, q, shift, st.c, st.m, p, st.mu, st.sigma, n1)), n1
    nsw = 5
    m.resample(d, nsw)
    for s in range(nsw):
        O.sbm_joint_sweep(A, q, shift, st.c, st.m, p, st.mu, st.sigma, st.pi, st.niw, 1.3, 0.9, 606, s)
    assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c)
    assert np.array_equal(m.get(abi.ST_P, (K, K)), p)
    assert np.array_equal(m.get(abi.ST_MU, (K, K, B)), st.mu) and np.array_equal(m.get(abi.ST_SIGMA, (K, K, B, B)), st.sigma)
    assert np.array_equal(m.get(abi.ST_M, (K,)), st.m)
    assert m.stats()["label_changes"] > 0
    ll = m.log_likelihood(d)
    ref = O.sbm_joint_loglik(A, q, shift, st.c, p, st.mu, st.sigma)
    assert abs(ll - ref) <= 1e-9 * abs(ref)
    # step-wise parts of the next sweep
    m.resample_part(d, abi.PART_PARAMS)
    O.sbm_joint_sweep(A, q, shift, st.c, st.m, p, st.mu, st.sigma, st.pi, st.niw, 1.3, 0.9, 606, nsw, parts=O.PART_PARAMS)
    assert np.array_equal(m.get(abi.ST_P, (K, K)), p) and np.array_equal(m.get(abi.ST_MU, (K, K, B)), st.mu)
    m.resample_part(d, abi.PART_C)
    O.sbm_joint_sweep(A, q, shift, st.c, st.m, p, st.mu, st.sigma, st.pi, st.niw, 1.3, 0.9, 606, nsw, parts=O.PART_C)
    assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c)
    m.resample_part(d, abi.PART_M)
    O.sbm_joint_sweep(A, q, shift, st.c, st.m, p, st.mu, st.sigma, st.pi, st.niw, 1.3, 0.9, 606, nsw, parts=O.PART_M)
    assert np.array_equal(m.get(abi.ST_M, (K,)), st.m)
    with pytest.raises(abi.EngineError):
        abi.Model(kind=abi.SBM_JOINT, N=N, K=K, B=3, seed=1)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,B,special,beta", [(300, 100, 1, 1, 1.0), (520, 128, 2, 0, 1.0), (257, 65, 1, 1, 0.6), (700, 97, 3, 1, 1.0),
                                                 (600, 256, 1, 1, 1.0), (450, 193, 2, 0, 0.8), (1300, 200, 1, 1, 1.0)])
def test_more_than_64_blocks_chain_bitexact(abi, N, K, B, special, beta):
    """65..256 blocks: a node's (node, block) lanes span two or four wavefronts - the wide instantiation of the resident scan (categorical
    draw as a Kogge-Stone scan through LDS in the oracle's order, 8-bit labels in the exchange payloads) is the scan form.
    Adjacency and weight chains against the oracle, bit for bit, from a random start (hundreds of changes per scan), in
    uneven chunks, tempered and not; log-likelihood and step-wise parts as well."""
    A, W, _ = planted_sbm(N, 5, B=B, seed=123)        # (few planted blocks, many model blocks: most stay sparsely populated)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=11), WState(N, K, B, seed=12)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=501, tau1=1.0, tau0=1.0, allow_self_connections=1,
                    special_case_self_conns=special)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=502, special_case_self_conns=special)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    w.set(abi.ST_MU_SELF, sw.mu_self); w.set(abi.ST_SIGMA_SELF, sw.sigma_self)
    ps, done = sa.p_self, 0
    for chunk in (1, 2, 4):
        abi.resample_many([adj, w], d, chunk, beta)
        for s in range(done, done + chunk):
            ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, special, 501, s, beta=beta)
            O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self,
                          special, 502, s, beta=beta)
        done += chunk
        assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c), "adjacency labels after %d sweeps" % done
        assert np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p) and np.array_equal(adj.get(abi.ST_M, (K,)), sa.m)
        assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c), "weight labels after %d sweeps" % done
        assert np.array_equal(w.get(abi.ST_MU, (K, K, B)), sw.mu) and np.array_equal(w.get(abi.ST_SIGMA, (K, K, B, B)), sw.sigma)
        assert np.array_equal(w.get(abi.ST_M, (K,)), sw.m)
        ll = float(np.sum(abi.log_likelihood_many([adj, w], d)))
        ref = (O.sbm_adj_loglik(A, sa.c, sa.p, ps, True, bool(special)) +
               O.sbm_w_loglik(A, q, shift, sw.c, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, bool(special)))
        assert abs(ll - ref) <= 1e-9 * abs(ref) + 1e-7, (ll, ref)
    assert adj.stats()["label_changes"] > 100
    # the reference's step-wise calls resample_p / resample_c / resample_m (adjacency.py:615-618), one sweep index
    for part in (abi.PART_PARAMS, abi.PART_C, abi.PART_M):
        adj.resample_part(d, part, beta)
        w.resample_part(d, part, beta)
    ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, special, 501, done, beta=beta)
    O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, special, 502, done,
                  beta=beta)
    assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(adj.get(abi.ST_M, (K,)), sa.m)
    assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c) and np.array_equal(w.get(abi.ST_M, (K,)), sw.m)
    assert np.array_equal(w.get(abi.ST_MU, (K, K, B)), sw.mu)
    assert np.array_equal(adj.node_lp(d, 0, K), O.sbm_adj_node_lp(A, sa.c, sa.m, sa.p, special, 0))     # (round 3: also for this shape)
    with pytest.raises(abi.EngineError):
        abi.Model(kind=abi.SBM_ADJ, N=N, K=257, B=1, seed=1)


@pytest.mark.gpu
def test_more_than_64_blocks_through_the_host_classes(abi):
    """The reference-shaped classes with C = 70: sweeps, log-likelihood, and the device predictive log-likelihood against the
    reference's generic formula (graphistician_amd/abstractions.py) with m concentrated on single blocks."""
    from graphistician_amd.abstractions import AdjacencyDistribution, GaussianWeightDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution
    np.random.seed(8)
    N, C, B = 90, 70, 1
    A, W, _ = planted_sbm(N, 4, B=B, seed=9)
    a = SBMAdjacencyDistribution(N, C=C)
    w = SBMGaussianWeightDistribution(N, B=B, C=C)
    for _ in range(3):
        a.resample(A)
        w.resample((A, W))
    assert np.isfinite(a.log_likelihood(A)) and np.isfinite(w.log_likelihood((A, W)))
    assert a.c.min() >= 0 and a.c.max() < C and w.c.max() < C
    Arow, Acol = np.random.rand(N + 1) < 0.4, np.random.rand(N + 1) < 0.6
    Acol[-1] = Arow[-1] = True
    Wrow, Wcol = np.random.randn(N + 1, B), np.random.randn(N + 1, B)
    Wcol[-1] = Wrow[-1]
    for k in (0, 33, 69):
        a.m = np.eye(C)[k]; w.m = np.eye(C)[k]
        ref_a = AdjacencyDistribution.approx_predictive_ll(a, Arow, Acol, M=2)
        assert abs(a.approx_predictive_ll(Arow, Acol, M=2) - ref_a) <= 1e-10 * abs(ref_a), k
        ref_w = GaussianWeightDistribution.approx_predictive_ll(w, Arow, Acol, Wrow, Wcol, M=2)
        assert abs(w.approx_predictive_ll(Arow, Acol, Wrow, Wcol, M=2) - ref_w) <= 1e-10 * abs(ref_w), k


@pytest.mark.gpu
@pytest.mark.parametrize("K,special,beta", [(4, 1, 1.0), (7, 0, 1.0), (3, 1, 0.4)])
def test_resample_p_over_a_list_of_graphs(K, special, beta):
    """SBMAdjacencyDistribution.resample_p(As) (adjacency.py:620-648): the block-pair counts - and the diagonal's, for
    p_self - are summed over the graphs of the list, every graph offering every pair, then one Beta draw per pair: bit for
    bit against the oracle's restatement, through the C ABI (gg_resample_part_many) and through the host class."""
    from graphistician_amd import _abi as abi
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from oracle import oracle as O
    from tests.helpers import planted_sbm, AdjState
    N = 120
    As = [planted_sbm(N, K, seed=40 + g)[0] for g in range(3)]
    st = AdjState(N, K, seed=9)
    m = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=55, tau1=1.5, tau0=0.7, allow_self_connections=1, special_case_self_conns=special)
    m.set(abi.ST_C, st.c, np.int32); m.set(abi.ST_M, st.m); m.set(abi.ST_PI, st.pi); m.set(abi.ST_P, st.p)
    m.set(abi.ST_P_SELF, np.array([st.p_self]))
    ds = [abi.DeviceData(A, None) for A in As]
    m.resample_part_many(ds, abi.PART_PARAMS, beta)
    p = st.p.copy()
    ps = O.sbm_adj_resample_p_multi(As, st.c, p, st.p_self, 1.5, 0.7, special, 55, 0, beta)
    assert np.array_equal(m.get(abi.ST_P, (K, K)), p)
    if special:
        assert m.get(abi.ST_P_SELF, (1,))[0] == ps
    # asked for again: a new sweep index (fresh variates), still the summed counts
    m.resample_part_many(ds, abi.PART_PARAMS, beta)
    p2 = st.p.copy()
    O.sbm_adj_resample_p_multi(As, st.c, p2, st.p_self, 1.5, 0.7, special, 55, 1, beta)
    assert np.array_equal(m.get(abi.ST_P, (K, K)), p2) and not np.array_equal(p, p2)
    # the host class
    np.random.seed(1)
    h = SBMAdjacencyDistribution(N, C=K, tau0=0.7, tau1=1.5, special_case_self_conns=bool(special), seed=66)
    c0 = h.c.copy()
    h.resample_p(As, beta)
    p3 = np.ones((K, K))
    O.sbm_adj_resample_p_multi(As, c0, p3, 0.5, 1.5, 0.7, special, 66, 0, beta)
    assert np.array_equal(h.p, p3)
    h.resample(As[0])          # and a plain sweep afterwards still runs


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,B,special", [(400, 100, 1, 1), (300, 70, 2, 0), (500, 200, 1, 1)])
def test_more_than_64_blocks_quiet_form_is_the_same_chain(abi, N, K, B, special):
    """Round 3: 65..256 blocks run the adaptive form too - the resident scan while labels keep changing, and once the chain is
    quiet the speculative one-launch scan (k_spec_resolve with the cross-wave categorical and 8-bit label payloads) in the
    two-launch ahead form (k_params + the whole next pass as riders, then the scan).  Strongly planted network, 80 sweeps in
    uneven chunks: both switches are crossed; labels and parameters bit for bit against the oracle at every checkpoint."""
    rng = np.random.RandomState(N)
    Kp_true = 4
    ctrue = rng.choice(Kp_true, size=N)
    ptrue = np.full((Kp_true, Kp_true), 0.08) + np.eye(Kp_true) * 0.8
    A = (rng.rand(N, N) < ptrue[np.ix_(ctrue, ctrue)]).astype(np.uint8)
    W = (rng.randn(Kp_true, Kp_true, B) * 3.0)[ctrue][:, ctrue] + 0.3 * rng.randn(N, N, B)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=21), WState(N, K, B, seed=22)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=601, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=special)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=602, special_case_self_conns=special)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    w.set(abi.ST_MU_SELF, sw.mu_self); w.set(abi.ST_SIGMA_SELF, sw.sigma_self)
    ps, done = sa.p_self, 0
    for chunk in (3, 1, 12, 2, 25, 1, 36):
        abi.resample_many([adj, w], d, chunk)
        for s in range(done, done + chunk):
            ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, special, 601, s)
            O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, special, 602, s)
        done += chunk
        assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c), "adjacency labels after %d sweeps" % done
        assert np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p) and np.array_equal(adj.get(abi.ST_M, (K,)), sa.m)
        assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c), "weight labels after %d sweeps" % done
        assert np.array_equal(w.get(abi.ST_MU, (K, K, B)), sw.mu) and np.array_equal(w.get(abi.ST_M, (K,)), sw.m)
    ll = float(np.sum(abi.log_likelihood_many([adj, w], d)))
    ref = (O.sbm_adj_loglik(A, sa.c, sa.p, ps, True, bool(special)) +
           O.sbm_w_loglik(A, q, shift, sw.c, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, bool(special)))
    assert abs(ll - ref) <= 1e-9 * abs(ref) + 1e-7
    sta, stw = adj.stats(), w.stats()
    # (which forms a chain went through depends on when it went quiet: at least one of the two must have used the new one)
    assert sta["sweeps_two_launch"] + stw["sweeps_two_launch"] > 0, (sta, stw)
    assert sta["sweeps_serial"] + stw["sweeps_serial"] > 0


# ------------------------------------------------------------------ wide weights: two 24-bit limbs
def heavy_tailed_network(N, K, seed, outlier=1e3):
    rng = np.random.RandomState(seed)
    c = rng.choice(K, size=N).astype(np.int32)
    A = (rng.rand(N, N) < 0.6).astype(np.uint8)
    mu = rng.randn(K, K) * 0.05
    W = (mu[np.ix_(c, c)] + 0.01 * rng.randn(N, N))[:, :, None]
    i, j = np.argwhere(A)[7]
    W[i, j, 0] = outlier
    return A, np.ascontiguousarray(W), c


@pytest.mark.parametrize("N,K,special", [(70, 3, 1), (200, 8, 0), (300, 16, 1), (260, 40, 1), (150, 64, 0)])
def test_wide_weights_chain_bitexact(abi, N, K, special):
    """GG_SBM_GAUSS_W with wide_weights = 1 on data from gg_data_create_wide - reference-precision storage, 48 bits in two limbs
    (the reference computes on float64 W: abstractions.py:257-274, weights.py:316-331): labels, means, covariances and m bit for
    bit against the oracle fed the same two-limb data, over sweeps that change labels; the conditionals (gg_node_lp) to the bit;
    the log-likelihood against the oracle's per-edge sum on the recombined weights at 1e-9 and against scipy on the FLOAT64
    weights at 1e-10 - on a heavy-tailed network (weights ~ 0.01, one of 1e3) whose default storage keeps 6 bits and warns."""
    import warnings
    from oracle import literal as R
    A, W, _ = heavy_tailed_network(N, K, 11 * N)
    q2, sh = O.quantise_wide(A, W)
    st = WState(N, K, 1, seed=K + 5)
    st.sigma[:] = 1e-3
    with warnings.catch_warnings():
        warnings.simplefilter("error")                 # (the wide upload keeps 30 bits of a typical weight: no warning)
        d = abi.DeviceData(A, W, wide=True)
    assert d.info()["B"] == 1 and d.info()["shift"] == sh and d.quantisation()[2] >= 24
    Aback, Wback = d.download()
    assert np.array_equal(Aback, A) and np.all(np.abs(Wback - W)[A != 0] <= 2.0 ** (-sh - 25) * (1 + 1e-12))
    with pytest.warns(RuntimeWarning):
        abi.DeviceData(A, W)                           # the default storage reports its coarse scale
    m = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=4321, special_case_self_conns=special, wide_weights=1)
    for sid, v, dt in ((abi.ST_C, st.c, np.int32), (abi.ST_M, st.m, np.float64), (abi.ST_PI, st.pi, np.float64), (abi.ST_MU, st.mu, np.float64),
                       (abi.ST_SIGMA, st.sigma, np.float64), (abi.ST_MU_SELF, st.mu_self, np.float64), (abi.ST_SIGMA_SELF, st.sigma_self, np.float64),
                       (abi.ST_NIW, st.niw, np.float64), (abi.ST_NIW_SELF, st.niw_self, np.float64)):
        m.set(sid, v, dt)
    for n1 in (0, N // 2, N - 1):
        assert np.array_equal(m.node_lp(d, n1, K), O.sbm_w_node_lp(A, q2, sh, st.c, st.m, st.mu, st.sigma, special, n1, wide=True)), n1
    nsweeps = 5
    m.resample(d, 2)
    m.resample(d, nsweeps - 2)
    for s in range(nsweeps):
        O.sbm_w_sweep(A, q2, sh, st.c, st.m, st.mu, st.sigma, st.mu_self, st.sigma_self, st.pi, st.niw, st.niw_self, special, 4321, s, wide=True)
    assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c)
    assert np.array_equal(m.get(abi.ST_MU, (K, K, 1)), st.mu) and np.array_equal(m.get(abi.ST_SIGMA, (K, K, 1, 1)), st.sigma)
    assert np.array_equal(m.get(abi.ST_M, (K,)), st.m)
    if special:
        assert np.array_equal(m.get(abi.ST_MU_SELF, (1,)), st.mu_self) and np.array_equal(m.get(abi.ST_SIGMA_SELF, (1, 1)), st.sigma_self)
    assert m.stats()["label_changes"] > 0
    ll = m.log_likelihood(d)
    ll_o = O.sbm_w_loglik(A, q2, sh, st.c, st.mu, st.sigma, st.mu_self, st.sigma_self, special, wide=True)
    assert abs(ll - ll_o) <= 1e-9 * abs(ll_o)
    Mu, Sig = st.mu[np.ix_(st.c, st.c)], st.sigma[np.ix_(st.c, st.c)]
    if special:
        Mu[np.arange(N), np.arange(N)] = st.mu_self
        Sig[np.arange(N), np.arange(N)] = st.sigma_self
    ref = R.w_loglik(A, W, Mu, Sig)                    # scipy, per edge, on the float64 weights themselves
    assert abs(ll - ref) <= 1e-10 * abs(ref), (ll, ref)
    # mismatched pairs are refused, and the shapes the wide mode is not built for
    with pytest.raises(abi.EngineError):
        m.resample(abi.DeviceData(A, W[:, :, :1] * 0 + 0.5), 1)
    plain = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=1)
    with pytest.raises(abi.EngineError):
        plain.resample(d, 1)
    for bad in (dict(kind=abi.SBM_GAUSS_W, B=2, K=K), dict(kind=abi.SBM_GAUSS_W, B=1, K=70), dict(kind=abi.SBM_JOINT, B=1, K=K)):
        with pytest.raises(abi.EngineError):
            abi.Model(N=N, seed=1, wide_weights=1, **bad)


def test_wide_weights_through_the_host_classes(abi):
    """FactorizedNetworkDistribution with SBMGaussianWeightDistribution(wide_weights=True): the (A, W) arrays are uploaded in the
    wide form, the adjacency half shares the handle, resample / log_probability / log_prior run, the NIW kind too."""
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution, NIWGaussianWeightDistribution
    N, K = 120, 4
    A, W, _ = heavy_tailed_network(N, K, 9)
    A.setflags(write=False); W.setflags(write=False)
    np.random.seed(2)
    for wcls, kw in ((SBMGaussianWeightDistribution, dict(B=1, C=K, seed=5, wide_weights=True)),
                     (NIWGaussianWeightDistribution, dict(B=1, seed=6, wide_weights=True))):
        model = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, dict(C=K, seed=4), wcls, kw)
        lps = []
        for _ in range(6):
            model.resample((A, W))
            lps.append(model.log_probability((A, W)))
        assert np.all(np.isfinite(lps)) and lps[-1] > lps[0] - 1e-6 * abs(lps[0])
        assert model._device_data((A, W)).wide
        host = model.weights.log_prior() if not hasattr(model.weights, "_log_prior_host") else model.weights._log_prior_host()
        assert abs(model.weights.log_prior() - host) <= 1e-10 * abs(host)
