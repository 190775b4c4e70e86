"""CPU tests: the C-ABI library loads and exports every symbol declared in
include/graphgibbs.h, fails loudly without a device, and the host-side logic of the
reference-shaped classes (construction, P, rvs, priors, data cache)."""
import os
import re

import numpy as np
import pytest
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from graphistician_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "graphgibbs.h")).read()
    declared = sorted(set(re.findall(r"\b(gg_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 18
    L = _abi.lib()
    for name in declared:
        assert hasattr(L, name), "libgraphgibbs.so does not export %s" % name
    assert sorted(_abi.EXPORTS) == declared
    assert b"gfx950" in L.gg_version()


_CTYPES_OF = {"int32_t": "c_int", "uint32_t": "c_uint", "uint64_t": "c_ulong", "int64_t": "c_long", "double": "c_double"}


def _header_struct(hdr, name):
    """[(c type, field name), ...] of `typedef struct <name> { ... } <name>;` in include/graphgibbs.h."""
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        for n in names.split(","):
            out.append((ctype, n.strip()))
    return out


def _stub_struct(md, cls):
    """The ctypes `_fields_` of class `cls` in the stub INTEGRATION.md shows a maintainer."""
    block = re.search(r"class\s+%s\(C\.Structure\):.*?_fields_\s*=\s*\[(.*?)\]\n" % cls, md, re.S).group(1)
    return re.findall(r'\("(\w+)",\s*C\.(\w+)\)', block)


def test_ctypes_structs_match_the_header_field_for_field():
    """gg_config / gg_stats as declared in include/graphgibbs.h, as bound in graphistician_amd/_abi.py and as printed in
    INTEGRATION.md must agree in names, types and order (round 1 shipped a stub that ended two fields early), and
    gg_config has no hidden padding, so sizeof is the sum of its fields."""
    import ctypes as C
    from graphistician_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "graphgibbs.h")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for cname, cls in (("gg_config", _abi.Config), ("gg_stats", _abi.Stats)):
        want = _header_struct(hdr, cname)
        got = [(f[1].__name__, f[0]) for f in cls._fields_]
        assert [(_CTYPES_OF[t], n) for t, n in want] == got, cname
    cfg = _header_struct(hdr, "gg_config")
    assert cfg[0] == ("uint32_t", "struct_size")
    size = {"int32_t": 4, "uint32_t": 4, "uint64_t": 8, "int64_t": 8, "double": 8}
    assert C.sizeof(_abi.Config) == sum(size[t] for t, _ in cfg), "hidden padding in gg_config"
    stub = _stub_struct(md, "Config")
    assert [n for n, _ in stub] == [n for _, n in cfg], "INTEGRATION.md's Config stub is out of step with gg_config"
    assert [getattr(C, t).__name__ for _, t in stub] == [_CTYPES_OF[t] for t, _ in cfg]
    assert "struct_size=C.sizeof(Config)" in md


def test_no_cpu_fallback():
    from graphistician_amd import _abi
    if _abi.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_abi.EngineError) as e:
        _abi.Model(kind=_abi.SBM_ADJ, N=8, K=2, B=1)
    assert e.value.code == _abi.ERR_NO_DEVICE
    with pytest.raises(_abi.EngineError):
        _abi.DeviceData(np.zeros((4, 4), np.uint8), None)
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    m = SBMAdjacencyDistribution(8, C=2)
    with pytest.raises(_abi.EngineError):
        m.resample(np.zeros((8, 8)))
    with pytest.raises(_abi.EngineError):
        m.log_likelihood(np.zeros((8, 8)))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "graphistician_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"import\s+oracle", r"from\s+oracle", r"oracle/", r"liboracle", r"oracle\.py", r"orc_[a-z_]+\("):
                    assert not re.search(pat, src), "%s references the oracle (%s)" % (f, pat)


def test_sbm_adjacency_host_logic():
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    np.random.seed(0)
    m = SBMAdjacencyDistribution(20, C=3, pi=2.0, tau0=2.0, tau1=3.0)
    assert m.c.shape == (20,) and m.c.min() >= 0 and m.c.max() <= 2
    assert m.p.shape == (3, 3) and abs(m.m.sum() - 1) < 1e-12
    P = m.P
    assert P.shape == (20, 20) and np.all(np.diag(P) == m.p_self) and P[0, 1] == m.p[m.c[0], m.c[1]]
    lp = stats.dirichlet(m.pi).logpdf(m.m) + stats.beta(3.0, 2.0).logpdf(m.p).sum() + stats.beta(3.0, 2.0).logpdf(m.p_self) \
        + np.log(m.m)[m.c].sum()
    assert abs(m.log_prior() - lp) < 1e-10
    A = m.rvs()
    assert A.shape == (20, 20) and A.dtype == bool
    m2 = SBMAdjacencyDistribution(20, C=3, allow_self_connections=False)
    assert np.all(np.diag(m2.P) == 0)
    with pytest.raises(AssertionError):
        SBMAdjacencyDistribution(20, C=0)
    with pytest.raises(AssertionError):
        SBMAdjacencyDistribution(20, C=2, m=np.array([0.9, 0.3]))
    with pytest.raises(AssertionError):
        SBMAdjacencyDistribution(20, C=2, c=np.zeros(19, dtype=int))
    m.c = np.zeros(20, dtype=np.int64)
    assert m.c.dtype == np.int32 and m.P[0, 1] == m.p[0, 0]
    prow, pcol = m.sample_predictive_parameters()
    assert prow.shape == (21,) and pcol.shape == (21,)
    assert sorted(m.invariant_sbm_order()) == list(range(20))


def test_weights_host_logic():
    from graphistician_amd.weights import SBMGaussianWeightDistribution, NIWGaussianWeightDistribution, \
        FixedGaussianWeightDistribution, LowRankGaussianWeightDistribution, niw_log_prob
    np.random.seed(1)
    w = SBMGaussianWeightDistribution(15, B=2, C=3)
    assert w.Mu.shape == (15, 15, 2) and w.Sigma.shape == (15, 15, 2, 2)
    g = w._gaussians[1][2]
    assert np.array_equal(w.Mu[np.where(w.c == 1)[0][:1], np.where(w.c == 2)[0][:1]].ravel(), g.mu) or \
        not ((w.c == 1).any() and (w.c == 2).any())
    assert np.allclose(g.sigma_chol.dot(g.sigma_chol.T), g.sigma)
    assert g.nu_0 == 4 and g.kappa_0 == 1.0 and np.array_equal(g.mu_0, np.zeros(2))
    lp = stats.dirichlet(w.pi).logpdf(w.m) + np.log(w.m)[w.c].sum()
    for r in w._gaussians:
        for gg in r:
            lp += niw_log_prob(gg.mu, gg.sigma, gg.mu_0, gg.sigma_0, gg.nu_0, gg.kappa_0)
    sg = w._self_gaussian
    lp += niw_log_prob(sg.mu, sg.sigma, sg.mu_0, sg.sigma_0, sg.nu_0, sg.kappa_0)
    assert abs(w.log_prior() - lp) < 1e-9
    W = w.rvs()
    assert W.shape == (15, 15, 2)
    n = NIWGaussianWeightDistribution(10, B=1)
    assert n.Mu.shape == (10, 10, 1) and np.isfinite(n.log_prior())
    Mr, Mc, Lr, Lc = n.sample_predictive_parameters()
    assert Mr.shape == (11, 1) and Lc.shape == (11, 1, 1)
    f = FixedGaussianWeightDistribution(6, 2, np.zeros(2), np.eye(2))
    assert f.log_prior() == 0 and f.Sigma.shape == (6, 6, 2, 2)
    with pytest.raises(NotImplementedError):
        LowRankGaussianWeightDistribution(5, 2)


def test_rvs_statistics_follow_the_model():
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution
    np.random.seed(3)
    net = FactorizedNetworkDistribution(200, SBMAdjacencyDistribution, {"C": 2}, SBMGaussianWeightDistribution, {"C": 2})
    A, W = net.rvs()
    a, w = net.adjacency, net.weights
    for k1 in range(2):
        for k2 in range(2):
            blk = (a.c == k1)[:, None] & (a.c == k2)[None, :] & ~np.eye(200, dtype=bool)
            if blk.sum() > 500:
                assert abs(A[blk].mean() - a.p[k1, k2]) < 5 * np.sqrt(0.25 / blk.sum())
            blk = (w.c == k1)[:, None] & (w.c == k2)[None, :] & ~np.eye(200, dtype=bool)
            if blk.sum() > 500:
                sd = np.sqrt(w._Sigma[k1, k2, 0, 0])
                assert abs(W[blk, 0].mean() - w._Mu[k1, k2, 0]) < 5 * sd / np.sqrt(blk.sum())


def test_network_composition_and_errors():
    from graphistician_amd.networks import FactorizedNetworkDistribution, GaussianBernoulliNetwork
    from graphistician_amd.adjacency import BetaBernoulliAdjacencyDistribution, LatentDistanceAdjacencyDistribution, \
        CompleteAdjacencyDistribution, EmptyAdjacencyDistribution
    from graphistician_amd.weights import NIWGaussianWeightDistribution
    np.random.seed(2)
    with pytest.raises(AssertionError):
        FactorizedNetworkDistribution(5, NIWGaussianWeightDistribution, {}, NIWGaussianWeightDistribution, {})
    net = FactorizedNetworkDistribution(9, BetaBernoulliAdjacencyDistribution, {}, NIWGaussianWeightDistribution, {"B": 2})
    assert net.adjacency.N == 9 and net.weights.B == 2
    assert abs(net.log_prior() - (net.adjacency.log_prior() + net.weights.log_prior())) < 1e-12
    g = GaussianBernoulliNetwork(7, 3, p=0.25)
    assert g.adjacency.p == 0.25 and g.adjacency.P[0, 0] == 0.25 and g.weights.Mu.shape == (7, 7, 3)
    ld = LatentDistanceAdjacencyDistribution(5, dim=2, sigma=2.0, mu0=0.3, mu_self=-0.2)   # host side only: no engine call
    assert ld.L.shape == (5, 2) and ld.mu_0 == 0.3 and ld.mu_self == -0.2 and ld._L_step_sz == 0.01
    assert np.allclose(np.diag(ld.D), 0.3 - 0.2) and ld.P.shape == (5, 5) and np.isfinite(ld.log_prior())
    assert CompleteAdjacencyDistribution(4).log_likelihood(np.ones((4, 4))) == 0
    assert EmptyAdjacencyDistribution(4).log_likelihood(np.ones((4, 4))) == -np.inf
    Arow, Acol, Wrow, Wcol = net.sample_predictive_distribution()
    from graphistician_amd import _abi
    if _abi.device_count() > 0:
        assert np.isfinite(net.approx_predictive_ll(Arow, Acol, Wrow, Wcol, M=4))
    else:      # the held-out evaluation of the block models runs on the engine (gg_predictive_ll): no CPU fallback
        with pytest.raises(_abi.EngineError):
            net.approx_predictive_ll(Arow, Acol, Wrow, Wcol, M=4)


def test_vectorised_niw_log_prior_matches_the_scipy_loop():
    """SBMGaussianWeightDistribution.log_prior sums K^2 NIW log densities (weights.py:343-352): the closed form over the
    whole batch equals the reference's loop of scipy calls."""
    from graphistician_amd.weights import niw_log_prob, niw_log_prob_many
    rng = np.random.RandomState(3)
    for B in (1, 2, 3):
        K = 4
        mu = rng.randn(K, K, B)
        a = rng.randn(K, K, B, B + 2)
        sigma = np.einsum("...ik,...jk->...ij", a, a) + 0.1 * np.eye(B)
        mu_0, nu_0, kappa_0 = rng.randn(B), B + 2.5, 0.7
        s0 = rng.randn(B, B + 1)
        sigma_0 = s0.dot(s0.T) + 0.2 * np.eye(B)
        ref = sum(niw_log_prob(mu[i, j], sigma[i, j], mu_0, sigma_0, nu_0, kappa_0) for i in range(K) for j in range(K))
        got = niw_log_prob_many(mu, sigma, mu_0, sigma_0, nu_0, kappa_0)
        assert abs(got - ref) <= 1e-9 * abs(ref)


def test_stale_config_struct_is_refused_before_anything_else():
    """gg_model_create checks gg_config.struct_size first (no device needed to see it)."""
    import ctypes as C
    from graphistician_amd import _abi
    cfg = _abi.Config()
    cfg.kind, cfg.N, cfg.K, cfg.B = _abi.SBM_ADJ, 8, 2, 1
    h = C.c_void_p()
    for bad in (0, C.sizeof(_abi.Config) - 8):
        cfg.struct_size = bad
        assert _abi.lib().gg_model_create(C.byref(h), C.byref(cfg)) == _abi.ERR_INVALID
        assert b"struct_size" in _abi.lib().gg_last_error()


def test_data_cache_key_sees_in_place_edits():
    """engine.as_device_data keys its cache on a hash of the WHOLE buffer as well as on identity: refilling a
    buffer in place, masking a held-out row / column or flipping ONE entry changes the key - the reference reads its arrays
    on every call (networks.py:28,69).  Arrays declared immutable are keyed by identity alone.  NaN in A is refused at upload."""
    from graphistician_amd import engine, _abi
    rng = np.random.RandomState(0)
    A = (rng.rand(512, 512) < 0.5).astype(np.float64)
    W = rng.randn(512, 512, 2)
    k0 = (engine._fingerprint(A), engine._fingerprint(W))
    assert k0 == (engine._fingerprint(A), engine._fingerprint(W))
    A[200, 301] = 1.0 - A[200, 301]                    # one entry
    k1 = engine._fingerprint(A)
    assert k1 != k0[0]
    W[17, 400, 1] += 1e-9                              # one weight, the smallest of edits
    assert engine._fingerprint(W) != k0[1]
    A2 = A.copy()
    A2[37, :] = 0; A2[:, 37] = 0                       # hold one node out
    A[:] = A2
    assert engine._fingerprint(A) not in (k0[0], k1)
    At = np.asfortranarray(A)                          # any memory layout is hashed through its C-order bytes
    kt = engine._fingerprint(At)
    At[5, 6] = 1.0 - At[5, 6]
    assert engine._fingerprint(At) != kt
    engine.declare_immutable(A)
    ki = engine._fingerprint(A)
    A[1, 2] = 1.0 - A[1, 2]
    assert engine._fingerprint(A) == ki                # promised read-only: identity is the key
    engine.invalidate(A)
    assert engine._fingerprint(A) != ki
    A[3, 4] = np.nan
    with pytest.raises(ValueError):
        _abi._adjacency_u8(A)
