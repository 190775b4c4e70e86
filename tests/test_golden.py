"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py):
the CPU oracle must reproduce them (no GPU), and the HIP engine must reproduce
them through the Python surface (gpu)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SBM_CASES = ["sbm_n64_k4_b1.npz", "sbm_n48_k3_b2_nospecial.npz"]


def load(name):
    z = np.load(os.path.join(G, name))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", SBM_CASES)
def test_oracle_reproduces_sbm_golden(name):
    g = load(name)
    A, W = g["A"], g["W"]
    special = int(g["special"])
    q, shift = O.quantise(A, W)
    assert shift == int(g["shift"])
    c, m, p, ps = g["init_a_c"].copy(), g["init_a_m"].copy(), g["init_a_p"].copy(), float(g["init_a_p_self"])
    wc, wm, mu, sig = g["init_w_c"].copy(), g["init_w_m"].copy(), g["init_w_mu"].copy(), g["init_w_sigma"].copy()
    mus, sigs = g["init_w_mu_self"].copy(), g["init_w_sigma_self"].copy()
    for s in range(int(g["nsweeps"])):
        ps = O.sbm_adj_sweep(A, c, m, p, ps, g["init_a_pi"], 1.0, 1.0, special, int(g["seed_a"]), s)
        O.sbm_w_sweep(A, q, shift, wc, wm, mu, sig, mus, sigs, g["init_w_pi"], g["init_w_niw"], g["init_w_niw_self"],
                      special, int(g["seed_w"]), s)
        assert np.array_equal(np.concatenate([c, wc]), g["trace_c"][s])
    assert np.array_equal(p, g["out_a_p"]) and np.array_equal(m, g["out_a_m"])
    assert np.array_equal(mu, g["out_w_mu"]) and np.array_equal(sig, g["out_w_sigma"])
    assert O.sbm_adj_loglik(A, c, p, ps, 1, special) == float(g["ll_a"])


def test_oracle_reproduces_er_golden():
    g = load("er_n50_b2.npz")
    A, W, B = g["A"], g["W"], int(g["B"])
    q, shift = O.quantise(A, W)
    mu, sig, mus, sigs = np.zeros((1, 1, B)), np.zeros((1, 1, B, B)), np.zeros(B), np.zeros((B, B))
    for s in range(int(g["nsweeps"])):
        p, ps = O.bb_adj_sweep(A, 1.0, 1.0, 1, int(g["seed_a"]), s)
        O.niw_w_sweep(A, q, shift, mu, sig, mus, sigs, g["niw"], g["niw"], int(g["seed_w"]), s)
        assert [p, ps] == list(g["p_trace"][s])
    assert np.array_equal(mu, g["out_mu"]) and np.array_equal(sigs, g["out_sigma_self"])


# ------------------------------------------------------------------ engine, through the reference-shaped classes
@pytest.mark.gpu
@pytest.mark.parametrize("name", SBM_CASES)
def test_engine_reproduces_sbm_golden(name):
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution
    g = load(name)
    N, K, B, special = int(g["N"]), int(g["K"]), int(g["B"]), bool(g["special"])
    net = FactorizedNetworkDistribution(
        N, SBMAdjacencyDistribution, dict(C=K, special_case_self_conns=special, seed=int(g["seed_a"])),
        SBMGaussianWeightDistribution, dict(C=K, B=B, special_case_self_conns=special, seed=int(g["seed_w"])))
    a, w = net.adjacency, net.weights
    a.c, a.m, a.p, a.pi = g["init_a_c"], g["init_a_m"], g["init_a_p"], g["init_a_pi"]
    a.p_self = float(g["init_a_p_self"])
    w.c, w.m, w.pi = g["init_w_c"], g["init_w_m"], g["init_w_pi"]
    w._hset("_mu", g["init_w_mu"]); w._hset("_sigma", g["init_w_sigma"])
    w._hset("_mu_self", g["init_w_mu_self"]); w._hset("_sigma_self", g["init_w_sigma_self"])
    w._hset("_niw", g["init_w_niw"]); w._hset("_niw_self", g["init_w_niw_self"])
    data = (g["A"], g["W"])
    for s in range(int(g["nsweeps"])):
        net.resample(data)
        assert np.array_equal(np.concatenate([a.c, w.c]), g["trace_c"][s]), "sweep %d" % s
    assert np.array_equal(a.p, g["out_a_p"]) and np.array_equal(a.m, g["out_a_m"])
    assert np.array_equal(w._Mu, g["out_w_mu"]) and np.array_equal(w._Sigma, g["out_w_sigma"])
    assert np.array_equal(w.m, g["out_w_m"])
    ll = net.log_likelihood(data)
    ref = float(g["ll_a"]) + float(g["ll_w"])
    assert abs(ll - ref) <= 1e-9 * abs(ref)
    assert np.isfinite(net.log_probability(data))


@pytest.mark.gpu
def test_engine_reproduces_er_golden():
    """Config 1 (BASELINE.json configs[0] shape): Beta-Bernoulli + NIW, N=50, B=2."""
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import BetaBernoulliAdjacencyDistribution
    from graphistician_amd.weights import NIWGaussianWeightDistribution
    g = load("er_n50_b2.npz")
    N, B = int(g["N"]), int(g["B"])
    net = FactorizedNetworkDistribution(
        N, BetaBernoulliAdjacencyDistribution, dict(tau1_self=1.0, tau0_self=1.0, seed=int(g["seed_a"])),
        NIWGaussianWeightDistribution, dict(B=B, seed=int(g["seed_w"])))
    data = (g["A"], g["W"])
    for s in range(int(g["nsweeps"])):
        net.resample(data)
        assert [net.adjacency.p, net.adjacency.p_self] == list(g["p_trace"][s])
    assert np.array_equal(net.weights._gaussian.mu, g["out_mu"][0, 0])
    assert np.array_equal(net.weights._gaussian.sigma, g["out_sigma"][0, 0])
    assert np.array_equal(net.weights._self_gaussian.sigma, g["out_sigma_self"])
    ll = net.log_likelihood(data)
    ref = float(g["ll_a"]) + float(g["ll_w"])
    assert abs(ll - ref) <= 1e-9 * abs(ref)


# ---------------------------------------------------------------- eigenmodels and latent-distance model
EIG_FIXTURES = ["eigen_logistic_n40_d3.npz", "eigen_probit_n40_d3.npz"]


def _run_eigen_oracle(g):
    F, lmbda, mu0 = g["init_F"].copy(), g["init_lmbda"].copy(), float(g["init_mu0"])
    probit = int(g["probit"]) == 1
    sweep = O.probit_sweep if probit else O.eig_sweep
    for s in range(int(g["nsweeps"])):
        mu0, _ = sweep(g["A"], F, lmbda, mu0, np.zeros(int(g["D"])), -0.3, 1.0, 1.0, 1.0, 0, int(g["seed"]), s)
    return F, lmbda, mu0


@pytest.mark.parametrize("name", EIG_FIXTURES)
def test_oracle_reproduces_eigen_golden(name):
    g = np.load(os.path.join(G, name))
    F, lmbda, mu0 = _run_eigen_oracle(g)
    assert np.allclose(F, g["out_F"], rtol=1e-12, atol=1e-13) and np.allclose(lmbda, g["out_lmbda"], rtol=1e-12, atol=1e-13)
    assert abs(mu0 - float(g["out_mu0"])) < 1e-12
    ll = (O.probit_loglik if int(g["probit"]) else O.eig_loglik)(g["A"], F, lmbda, mu0)
    assert abs(ll - float(g["ll"])) <= 1e-10 * abs(ll)


def test_oracle_reproduces_latent_golden():
    g = np.load(os.path.join(G, "latent_n40_dim2.npz"))
    L, mu0, ms, st, ar = g["init_L"].copy(), float(g["init_mu0"]), float(g["init_mu_self"]), 0.01, 0.9
    for s in range(int(g["nsweeps"])):
        mu0, ms, st, ar, _ = O.ld_resample(g["A"], L, mu0, ms, float(g["sigma"]), st, ar, int(g["seed"]), s)
    assert np.allclose(L, g["out_L"], rtol=1e-12, atol=1e-13)
    assert abs(mu0 - float(g["out_mu0"])) < 1e-12 and abs(ms - float(g["out_mu_self"])) < 1e-12
    assert abs(st - float(g["out_step"])) < 1e-15 and abs(ar - float(g["out_rate"])) < 1e-15


def _ldw_state(g, pre):
    return [float(g[pre + k]) for k in ("b", "eta", "sigma", "mu_self", "sigma_self")]


def test_oracle_reproduces_latent_weights_golden():
    g = np.load(os.path.join(G, "latentw_n40_dim2.npz"))
    q, shift = O.quantise(g["A"], g["W"])
    L = g["init_L"].copy()
    b, eta, sg, ms, ss = _ldw_state(g, "init_")
    for s in range(int(g["nsweeps"])):
        b, eta, sg, ms, ss, _ = O.ldw_resample(g["A"], q, shift, L, b, eta, sg, ms, ss, g["prior"], g["prior_self"], int(g["seed"]), s)
    assert np.allclose(L, g["out_L"], rtol=1e-12, atol=1e-13)
    assert np.allclose([b, eta, sg, ms, ss], _ldw_state(g, "out_"), rtol=1e-12)
    assert abs(O.ldw_loglik(g["A"], q, shift, L, b, sg, ms, ss) - float(g["ll"])) <= 1e-10 * abs(float(g["ll"]))


@pytest.mark.gpu
def test_engine_reproduces_latent_weights_golden():
    from graphistician_amd import _abi as abi
    g = np.load(os.path.join(G, "latentw_n40_dim2.npz"))
    N, dim = int(g["N"]), int(g["dim"])
    d = abi.DeviceData(g["A"], g["W"])
    m = abi.Model(kind=abi.LATENT_DIST_W, N=N, D=dim, K=1, B=1, seed=int(g["seed"]))
    b, eta, sg, ms, ss = _ldw_state(g, "init_")
    m.set(abi.ST_F, g["init_L"]); m.set(abi.ST_MU0, np.array([b])); m.set(abi.ST_LD_AUX, np.array([eta, sg, ms, ss]))
    m.set(abi.ST_NIW, np.array([0.0, g["prior"][0], g["prior"][1], 0.0])); m.set(abi.ST_NIW_SELF, g["prior_self"])
    m.resample(d, int(g["nsweeps"]))
    assert np.allclose(m.get(abi.ST_F, (N, dim)), g["out_L"], rtol=1e-7, atol=1e-9)
    out = _ldw_state(g, "out_")
    assert abs(m.get(abi.ST_MU0, (1,))[0] - out[0]) < 1e-8
    assert np.allclose(m.get(abi.ST_LD_AUX, (4,)), out[1:], rtol=1e-8)
    assert abs(m.log_likelihood(d) - float(g["ll"])) <= 1e-8 * abs(float(g["ll"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", EIG_FIXTURES)
def test_engine_reproduces_eigen_golden(name):
    from graphistician_amd import _abi as abi
    g = np.load(os.path.join(G, name))
    N, D = int(g["N"]), int(g["D"])
    kind = abi.PROBIT_EIG if int(g["probit"]) else abi.LOGISTIC_EIG
    d = abi.DeviceData(g["A"], None)
    m = abi.Model(kind=kind, N=N, D=D, K=1, B=1, seed=int(g["seed"]), sigma_mu0=1.0, sigma_F=1.0, sigma_lmbda=1.0, mu_mu_0=-0.3,
                  lmbda_given=0)
    m.set(abi.ST_F, g["init_F"]); m.set(abi.ST_LMBDA, g["init_lmbda"]); m.set(abi.ST_MU0, np.array([float(g["init_mu0"])]))
    m.set(abi.ST_MU_LMBDA, np.zeros(D))
    m.resample(d, int(g["nsweeps"]))
    assert np.allclose(m.get(abi.ST_F, (N, D)), g["out_F"], rtol=1e-6, atol=1e-8)
    assert np.allclose(m.get(abi.ST_LMBDA, (D,)), g["out_lmbda"], rtol=1e-6, atol=1e-8)
    assert abs(m.get(abi.ST_MU0, (1,))[0] - float(g["out_mu0"])) < 1e-7
    assert abs(m.log_likelihood(d) - float(g["ll"])) <= 1e-7 * abs(float(g["ll"]))


@pytest.mark.gpu
def test_engine_reproduces_latent_golden():
    from graphistician_amd import _abi as abi
    g = np.load(os.path.join(G, "latent_n40_dim2.npz"))
    N, dim = int(g["N"]), int(g["dim"])
    d = abi.DeviceData(g["A"], None)
    m = abi.Model(kind=abi.LATENT_DIST, N=N, D=dim, K=1, B=1, seed=int(g["seed"]), sigma_F=float(g["sigma"]))
    m.set(abi.ST_F, g["init_L"]); m.set(abi.ST_MU0, np.array([float(g["init_mu0"])]))
    m.set(abi.ST_LD_AUX, np.array([float(g["init_mu_self"]), 0.01, 0.9]))
    m.resample(d, int(g["nsweeps"]))
    assert np.allclose(m.get(abi.ST_F, (N, dim)), g["out_L"], rtol=1e-7, atol=1e-9)
    aux = m.get(abi.ST_LD_AUX, (3,))
    assert abs(m.get(abi.ST_MU0, (1,))[0] - float(g["out_mu0"])) < 1e-8 and abs(aux[0] - float(g["out_mu_self"])) < 1e-8
    assert abs(aux[1] - float(g["out_step"])) < 1e-12
    assert abs(m.log_likelihood(d) - float(g["ll"])) <= 1e-8 * abs(float(g["ll"]))
