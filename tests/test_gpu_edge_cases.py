"""GPU edge cases: ragged sizes, empty / complete graphs, empty blocks, single block, tiny N,
self-connection conventions, B=4, model reuse across data sets, the Python-level data cache."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import planted_sbm, AdjState, WState

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def abi():
    from graphistician_amd import _abi
    if _abi.device_count() < 1:
        pytest.fail("no HIP device: the engine has no CPU fallback")
    return _abi


def run_adj(abi, A, st, seed, nsweeps, special=1, allow=1):
    N, K = st.N, st.K
    d = abi.DeviceData(A, None)
    m = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=seed, tau1=st.tau1, tau0=st.tau0, allow_self_connections=allow,
                  special_case_self_conns=special)
    m.set(abi.ST_C, st.c, np.int32); m.set(abi.ST_M, st.m); m.set(abi.ST_PI, st.pi); m.set(abi.ST_P, st.p)
    m.set(abi.ST_P_SELF, np.array([st.p_self]))
    ps = st.p_self
    for s in range(nsweeps):
        m.resample(d, 1)
        ps = O.sbm_adj_sweep(A, st.c, st.m, st.p, ps, st.pi, st.tau1, st.tau0, special, seed, s)
        assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c), "sweep %d" % s
    assert np.array_equal(m.get(abi.ST_P, (K, K)), st.p) and np.array_equal(m.get(abi.ST_M, (K,)), st.m)
    return m, d, ps


@pytest.mark.parametrize("N,K", [(1, 2), (2, 2), (3, 3), (63, 2), (65, 5), (129, 33), (1000, 16)])
def test_ragged_and_tiny_sizes(abi, N, K):
    rng = np.random.RandomState(N)
    A = (rng.rand(N, N) < 0.4).astype(np.uint8)
    m, d, ps = run_adj(abi, A, AdjState(N, K, seed=N + 1), seed=3, nsweeps=3)
    ll = m.log_likelihood(d)
    st_ll = O.sbm_adj_loglik(A, m.get(abi.ST_C, (N,), np.int32), m.get(abi.ST_P, (K, K)), ps)
    assert abs(ll - st_ll) <= 1e-10 * max(1.0, abs(st_ll))


@pytest.mark.parametrize("fill", [0, 1])
def test_empty_and_complete_graphs(abi, fill):
    N, K = 70, 4
    A = np.full((N, N), fill, dtype=np.uint8)
    m, d, ps = run_adj(abi, A, AdjState(N, K, seed=5), seed=8, nsweeps=3)
    assert np.isfinite(m.log_likelihood(d))
    # weights on a graph without edges: every Gaussian is a prior draw, labels follow m only
    W = np.random.RandomState(1).randn(N, N, 1)
    st = WState(N, K, 1, seed=2)
    q, shift = O.quantise(A, W)
    dw = abi.DeviceData(A, W)
    assert dw.info()["shift"] == shift
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=21, special_case_self_conns=1)
    for sid, v, dt in ((abi.ST_C, st.c, np.int32), (abi.ST_M, st.m, np.float64), (abi.ST_PI, st.pi, np.float64),
                       (abi.ST_MU, st.mu, np.float64), (abi.ST_SIGMA, st.sigma, np.float64),
                       (abi.ST_NIW, st.niw, np.float64), (abi.ST_NIW_SELF, st.niw_self, np.float64)):
        w.set(sid, v, dt)
    for s in range(2):
        w.resample(dw, 1)
        O.sbm_w_sweep(A, q, shift, st.c, st.m, st.mu, st.sigma, st.mu_self, st.sigma_self, st.pi, st.niw, st.niw_self, 1, 21, s)
    assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), st.c)
    assert np.array_equal(w.get(abi.ST_MU, (K, K, 1)), st.mu)


def test_single_block_and_empty_blocks(abi):
    N = 90
    A, W, _ = planted_sbm(N, 3, seed=3)
    # C = 1: resample_c returns early (adjacency.py:657-658), p and m are still drawn
    run_adj(abi, A, AdjState(N, 1, seed=1), seed=4, nsweeps=2)
    # labels that leave most of the 40 blocks empty
    st = AdjState(N, 40, seed=2)
    st.c[:] = np.random.RandomState(0).choice(3, size=N)
    run_adj(abi, A, st, seed=6, nsweeps=3)


def test_self_connection_conventions(abi):
    N, K = 80, 3
    A, _, _ = planted_sbm(N, K, seed=12)
    for special, allow in [(1, 1), (0, 1), (1, 0), (0, 0)]:
        m, d, ps = run_adj(abi, A.copy(), AdjState(N, K, seed=7), seed=9, nsweeps=2, special=special, allow=allow)
        ll = m.log_likelihood(d)
        ref = O.sbm_adj_loglik(A, m.get(abi.ST_C, (N,), np.int32), m.get(abi.ST_P, (K, K)), ps, allow, special)
        if np.isfinite(ref):
            assert abs(ll - ref) <= 1e-10 * abs(ref)
        else:
            assert ll == ref          # -inf when a self edge exists but P_ii = 0 (adjacency.py:525-526)


def test_weights_b4_chain(abi):
    N, K, B = 60, 3, 4
    A, W, _ = planted_sbm(N, K, B=B, seed=31)
    q, shift = O.quantise(A, W)
    st = WState(N, K, B, seed=32)
    d = abi.DeviceData(A, W)
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=77, special_case_self_conns=1)
    for sid, v, dt in ((abi.ST_C, st.c, np.int32), (abi.ST_M, st.m, np.float64), (abi.ST_PI, st.pi, np.float64),
                       (abi.ST_MU, st.mu, np.float64), (abi.ST_SIGMA, st.sigma, np.float64),
                       (abi.ST_MU_SELF, st.mu_self, np.float64), (abi.ST_SIGMA_SELF, st.sigma_self, np.float64),
                       (abi.ST_NIW, st.niw, np.float64), (abi.ST_NIW_SELF, st.niw_self, np.float64)):
        w.set(sid, v, dt)
    for s in range(3):
        w.resample(d, 1)
        O.sbm_w_sweep(A, q, shift, st.c, st.m, st.mu, st.sigma, st.mu_self, st.sigma_self, st.pi, st.niw, st.niw_self, 1, 77, s)
        assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), st.c)
    assert np.array_equal(w.get(abi.ST_SIGMA, (K, K, B, B)), st.sigma)
    with pytest.raises(abi.EngineError):
        abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=5, seed=1)       # B <= 4 (GG_MAXB)


def test_model_reuse_across_data_and_async_batches(abi):
    """One model, two data sets, sweeps issued in async batches: same chain as the oracle."""
    N, K = 140, 6
    A1, _, _ = planted_sbm(N, K, seed=1)
    A2, _, _ = planted_sbm(N, K, seed=2)
    st = AdjState(N, K, seed=3)
    d1, d2 = abi.DeviceData(A1, None), abi.DeviceData(A2, None)
    m = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=55, tau1=1.0, tau0=1.0, allow_self_connections=1,
                  special_case_self_conns=1)
    m.set(abi.ST_C, st.c, np.int32); m.set(abi.ST_M, st.m); m.set(abi.ST_PI, st.pi); m.set(abi.ST_P, st.p)
    m.set(abi.ST_P_SELF, np.array([st.p_self]))
    m.resample_async(d1, 3)
    m.resample_async(d1, 2)
    m.resample_async(d2, 4)          # switching data forces a sync in between
    m.sync()
    ps, s = st.p_self, 0
    for A, n in ((A1, 5), (A2, 4)):
        for _ in range(n):
            ps = O.sbm_adj_sweep(A, st.c, st.m, st.p, ps, st.pi, 1.0, 1.0, 1, 55, s)
            s += 1
    assert np.array_equal(m.get(abi.ST_C, (N,), np.int32), st.c)
    assert np.array_equal(m.get(abi.ST_M, (K,)), st.m) and m.sweep == 9
    with pytest.raises(abi.EngineError):
        m.resample(abi.DeviceData(A1[:50, :50], None), 1)           # N mismatch is an error, not a crash


def test_python_data_cache_and_state_roundtrip(abi):
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd import engine
    engine.clear_data_cache()
    N = 64
    A, _, _ = planted_sbm(N, 3, seed=9)
    np.random.seed(1)
    m = SBMAdjacencyDistribution(N, C=3, seed=5)
    m.resample(A); m.resample(A)
    assert len(engine._DATA_CACHE) == 1                  # same ndarray -> one upload
    c1 = m.c.copy()
    m.c = np.zeros(N, dtype=int)                         # host edit is pushed before the next sweep
    m.resample(A)
    assert m.c.shape == (N,) and m.engine_stats["sweeps"] == 3
    m.c = c1
    assert np.array_equal(m.c, c1)


def test_one_flipped_entry_is_new_data(abi):
    """the host classes find their device copy by a hash of the whole array - flip ONE entry of A (or edit one
    weight) in place and the next log_likelihood is that of the edited network, as in the reference, which reads its arrays on
    every call (networks.py:28,69).  A handle held by a caller is never recycled for other data."""
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution
    from graphistician_amd import engine
    engine.clear_data_cache()
    N = 96
    A, W, _ = planted_sbm(N, 3, seed=4, B=1)
    A = A.astype(bool)
    np.random.seed(2)
    net = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, dict(C=3), SBMGaussianWeightDistribution, dict(C=3, B=1))
    ll0 = net.log_likelihood((A, W))
    assert net.log_likelihood((A, W)) == ll0 and len(engine._DATA_CACHE) == 1
    i, j = [(i, j) for i in range(N) for j in range(N) if i != j and not A[i, j]][0]
    A[i, j] = True                                        # one new edge, in place
    ll1 = net.log_likelihood((A, W))
    assert ll1 != ll0
    ref = net.adjacency.log_likelihood(A.copy()) + net.weights.log_likelihood((A.copy(), W.copy()))
    assert abs(ll1 - ref) <= 1e-9 * abs(ref)
    W[i, j, 0] += 0.25                                    # and one weight
    ll2 = net.log_likelihood((A, W))
    assert ll2 != ll1
    # a handle a caller holds keeps its network while the cache turns over
    held = engine.as_device_data(A, W)
    ll_held = net.adjacency.log_likelihood(held)
    for s in range(engine._DATA_CACHE_SIZE + 2):
        A2, W2, _ = planted_sbm(N, 3, seed=50 + s, B=1)
        net.log_likelihood((A2, W2))
    assert net.adjacency.log_likelihood(held) == ll_held


@pytest.mark.gpu
@pytest.mark.parametrize("ahead,split", [("0", None), (None, None), (None, "0"), (None, "1"), (None, "0.27")])
def test_ahead_form_is_the_same_chain(abi, monkeypatch, ahead, split):
    """The ahead form (default while a chain is quiet): the bulk pass of sweep s+1 rides in the k_params and k_spec_resolve
    launches of sweep s and is dropped when the scan changes a label.  Same labels and parameters bit for bit as the
    oracle - and as the serial form (GG_AHEAD=0) - for any split of the riders between the two launches, through burn-in
    (resident scan), the quiet phase, calls that use or invalidate the statistics counted ahead (log_likelihood, node_lp,
    step-wise parts), labels replaced from outside, and another network on the same models."""
    if ahead is not None:
        monkeypatch.setenv("GG_AHEAD", ahead)
    if split is not None:
        monkeypatch.setenv("GG_RIDE_SPLIT", split)
    N, K = 260, 4
    A, W, _ = planted_sbm(N, K, seed=77)
    A2, W2, _ = planted_sbm(N, K, seed=78)
    q, shift = O.quantise(A, W)
    q2, shift2 = O.quantise(A2, W2)
    sa, sw = AdjState(N, K, seed=15), WState(N, K, 1, seed=16)
    d, d2 = abi.DeviceData(A, W), abi.DeviceData(A2, W2)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=51, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=52, special_case_self_conns=1)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    st = dict(ps=sa.p_self, sweep=0)

    def run(n, dd, AA, qq, sh):
        adj.resample_async(dd, n); w.resample_async(dd, n)
        abi.sync_many([adj, w])
        for _ in range(n):
            st["ps"] = O.sbm_adj_sweep(AA, sa.c, sa.m, sa.p, st["ps"], sa.pi, 1.0, 1.0, 1, 51, st["sweep"])
            O.sbm_w_sweep(AA, qq, sh, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, 1, 52, st["sweep"])
            st["sweep"] += 1
        assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c), st["sweep"]
        assert np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p) and np.array_equal(adj.get(abi.ST_M, (K,)), sa.m)
        assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c), st["sweep"]
        assert np.array_equal(w.get(abi.ST_MU, (K, K, 1)), sw.mu) and np.array_equal(w.get(abi.ST_SIGMA, (K, K, 1, 1)), sw.sigma)
        assert np.array_equal(w.get(abi.ST_M, (K,)), sw.m)

    def ll_check(dd, AA, qq, sh):
        ll = sum(abi.log_likelihood_many([adj, w], dd))
        ref = O.sbm_adj_loglik(AA, sa.c, sa.p, st["ps"]) + O.sbm_w_loglik(AA, qq, sh, sw.c, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, 1)
        assert abs(ll - ref) <= 1e-9 * abs(ref)

    run(6, d, A, q, shift)            # burn-in: resident scan
    run(40, d, A, q, shift)           # settles; the adaptive form goes back to the speculative launches (ahead form)
    assert adj.stats()["label_changes"] > 0
    ll_check(d, A, q, shift)          # log-likelihood from the statistics counted ahead
    run(3, d, A, q, shift)
    assert np.array_equal(adj.node_lp(d, 7, K), O.sbm_adj_node_lp(A, sa.c, sa.m, sa.p, 1, 7))
    run(1, d, A, q, shift); run(2, d, A, q, shift)
    # step-wise parts between full sweeps (they share the statistics counted ahead, then invalidate them)
    adj.resample_part(d, abi.PART_PARAMS)
    st["ps"] = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, st["ps"], sa.pi, 1.0, 1.0, 1, 51, st["sweep"], O.PART_PARAMS)
    adj.resample_part(d, abi.PART_C)
    O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, st["ps"], sa.pi, 1.0, 1.0, 1, 51, st["sweep"], O.PART_C)
    adj.resample_part(d, abi.PART_M)
    O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, st["ps"], sa.pi, 1.0, 1.0, 1, 51, st["sweep"], O.PART_M)
    assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p)
    w.resample(d, 1)
    O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, 1, 52, st["sweep"])
    st["sweep"] += 1
    run(5, d, A, q, shift)
    # another network on the same models, and back
    run(4, d2, A2, q2, shift2)
    ll_check(d2, A2, q2, shift2)
    run(7, d, A, q, shift)
    # labels replaced from outside
    rng = np.random.RandomState(9)
    sa.c[:] = rng.choice(K, size=N); sw.c[:] = rng.choice(K, size=N)
    adj.set(abi.ST_C, sa.c, np.int32); w.set(abi.ST_C, sw.c, np.int32)
    run(30, d, A, q, shift)
    ll_check(d, A, q, shift)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,lcap,ride", [(300, 7, 0, 1), (1100, 64, 0, 1), (900, 32, 0, 1), (700, 40, 0, 1), (800, 50, 0, 1), (700, 37, 4, 1),
                                            (300, 7, 3, 0), (1100, 64, 5, 0),
                                            (700, 16, 2, 0), (300, 7, 0, 2), (1100, 64, 0, 2), (900, 32, 0, 2), (300, 7, 3, 2), (1100, 64, 5, 2),
                                            (700, 16, 2, 2), (2500, 16, 0, 2), (2500, 16, 40, 2)])
def test_resident_grid_scan_is_the_same_chain(abi, N, K, lcap, ride, monkeypatch):
    """GG_SCAN=1: the scan as one resident-grid launch (k_scan: one grid-wide exchange per round, every round accepting the
    hypothesised changes up to the earliest surprise).  Same chain bit for bit, burn-in included (random start: hundreds
    of changes per scan, several changers per workgroup); N=1100, K=64 needs two bands of node groups (lazy change list
    at band load); GG_SCAN_LCAP: a change list of 2-5 entries, so that acceptances are cut short and the chip-wide
    flush of the list runs many times per scan.  ride = 2 (GG_SCAN_RIDE=2): from the second sweep on the NEXT sweep's bulk
    pass rides in the scan's launch, counting under the labels the scan starts with, and the scan's change list - flushed
    into that buffer too whenever it overflows - moves those statistics to the labels it ends with (the default, 1, does
    so only for chains that change few labels per sweep); 0: never.  K = 50, 37: block counts of the matrix-core conditional
    (33..64 blocks, gg_lp_mfma) that are not multiples of four / whose steps do not fill the last ring of four."""
    _resident_scan_case(abi, N, K, lcap, ride, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(700, 16), (900, 32), (500, 10), (600, 27)])
def test_small_block_conditional_on_the_matrix_cores(abi, N, K, monkeypatch):
    """GG_MFMA_SMALL=1 (off by default: measured no faster at K = 16): the resident scan's conditional for 9..32 blocks as f64
    matrix instructions (gg_lp_mfma_small) - the chains of gg_block_lp bit for bit, block counts that are not multiples of four
    (padding steps) included."""
    monkeypatch.setenv("GG_MFMA_SMALL", "1")
    _resident_scan_case(abi, N, K, 0, 0, monkeypatch)


def _resident_scan_case(abi, N, K, lcap, ride, monkeypatch):
    monkeypatch.setenv("GG_SCAN", "1")
    monkeypatch.setenv("GG_SCAN_RIDE", str(ride))
    if lcap:
        monkeypatch.setenv("GG_SCAN_LCAP", str(lcap))
    A, W, _ = planted_sbm(N, K, seed=31)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=5), WState(N, K, 1, seed=6)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=71, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=72, special_case_self_conns=1)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    ps = sa.p_self
    nsw = 4 if ride < 2 else 7
    abi.resample_many([adj, w], d, nsw - 2)        # both grids resident at the same time
    adj.resample(d, 1); w.resample(d, 1)           # (one call per sweep as well: what was counted ahead survives the call)
    ll_mid = adj.log_likelihood(d)                 # (and serves a log-likelihood in between)
    abi.resample_many([adj, w], d, 1)
    assert np.isfinite(ll_mid)
    for s in range(nsw):
        ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, 1, 71, s)
        O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, 1, 72, s)
    assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p)
    assert np.array_equal(adj.get(abi.ST_M, (K,)), sa.m)
    assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c) and np.array_equal(w.get(abi.ST_MU, (K, K, 1)), sw.mu)
    assert adj.stats()["label_changes"] > 0
    for mdl in (adj, w):
        st = mdl.stats()
        # hypothesis rounds: one grid-wide exchange accepts every hypothesised change up to the earliest surprise, so a
        # scan takes far fewer exchanges than it changes labels (one per change before)
        if st["label_changes"] > 200 and not lcap:
            assert st["resolve_rounds"] * 3 < st["label_changes"], st


@pytest.mark.gpu
@pytest.mark.parametrize("spin,retry,nsw", [(1, 0, 5), (3, 0, 5), (9, 0, 5), (1, 2, 14)])
def test_resident_scan_timeout_falls_back_exactly(abi, monkeypatch, capfd, spin, retry, nsw):
    """A k_scan exchange that times out (GG_SCAN_SPIN: a handful of polls instead of 4 million - what a device shared
    with other processes would do to the resident grid) leaves the scan as a consistent stalled one; the engine carries
    on with restart rounds and the chain stays the oracle's bit for bit.  The resident grid is tried again after a
    back-off (1024 sweeps, doubling; GG_SCAN_RETRY=2 here so that 14 sweeps see several suspensions and retries)."""
    monkeypatch.setenv("GG_SCAN", "1")
    monkeypatch.setenv("GG_SCAN_SPIN", str(spin))
    if retry:
        monkeypatch.setenv("GG_SCAN_RETRY", str(retry))
    N, K = 400, 16
    A, W, _ = planted_sbm(N, K, seed=19)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=25), WState(N, K, 1, seed=26)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=81, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=82, special_case_self_conns=1)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    ps = sa.p_self
    abi.resample_many([adj, w], d, nsw)
    for s in range(nsw):
        ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, 1, 81, s)
        O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, 1, 82, s)
    assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p)
    assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c) and np.array_equal(w.get(abi.ST_MU, (K, K, 1)), sw.mu)
    # (with a few more polls the 25 workgroups sometimes do meet in time: then nothing is printed)
    if spin == 1:
        err = capfd.readouterr().err
        assert "timed out" in err
        if retry:      # suspended, resumed, suspended again: more than one message per model
            assert err.count("timed out") > 2


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"GG_SCAN_SPIN": "1"}, {"GG_SCAN": "0"}, {"GG_SCAN": "1", "GG_SCAN_SPIN": "2", "GG_SCAN_RETRY": "2"},
                                 {"GG_SCAN": "0", "GG_AHEAD": "0"}])
def test_more_than_64_blocks_fall_back_to_restart_rounds(abi, monkeypatch, capfd, env):
    """Round 3: 65..256 blocks have the restart rounds too (k_round with the cross-wave categorical
    and 8-bit label payloads; k_spec_resolve stands in for the two-kernel form) - a resident-grid exchange that times out
    (GG_SCAN_SPIN) is no longer an error for them, and GG_SCAN=0 runs them without the resident grid at all.  Same chain as the
    oracle's, bit for bit, through burn-in (hundreds of restart rounds) and the step-wise parts."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from tests.fuzz_parity import one_case
    assert one_case(150, 70, 1, 1, 4242, (1, 2, 6)) is None
    assert one_case(130, 129, 2, 0, 4243, (2, 5)) is None
    N, K = 140, 100
    A, W, _ = planted_sbm(N, 3, seed=23)
    sa = AdjState(N, K, seed=27)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=83, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    ps = sa.p_self
    for s in range(3):      # the reference's step-wise calls (adjacency.py:615-618)
        adj.resample_part(d, abi.PART_PARAMS); adj.resample_part(d, abi.PART_C); adj.resample_part(d, abi.PART_M)
        ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, 1, 83, s)
        assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c), s
        assert np.array_equal(adj.get(abi.ST_M, (K,)), sa.m) and np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p), s
    if env.get("GG_SCAN_SPIN") == "1":
        assert "timed out" in capfd.readouterr().err


@pytest.mark.gpu
def test_timeout_behind_a_flush_recounts_and_continues(abi, monkeypatch, capfd):
    """A resident scan whose change list overflowed (flushed into the statistics chip-wide) and whose grid then cannot assemble:
    the band that was in LDS at the flush is not in G any more, so the stalled scan's statistics are not usable.  Rounds 2-3
    reported that as an error (GG_ERR_HIP, "flush"); round 4 (two processes sharing one device ran into it during burn-in at
    N=65536): labels, block sizes and position are a valid prefix of the scan, so the statistics are COUNTED AGAIN under the labels
    as they stand and the restart rounds go on from the position - the chain stays the oracle's.  Forced here with a 2-entry list
    and exchanges that give up after a few polls, from "never meets" to "mostly meets": timeouts at any point of a scan."""
    monkeypatch.setenv("GG_SCAN", "1")
    monkeypatch.setenv("GG_SCAN_LCAP", "2")
    N, K = 600, 8
    A, W, _ = planted_sbm(N, K, seed=29)
    d = abi.DeviceData(A, W)
    outcomes = []
    for trial, spin in enumerate((1, 2, 3, 4, 5, 6, 8, 10, 13, 17, 21, 34)):      # (from "never meets" to "mostly meets": timeouts at any point of a scan)
        monkeypatch.setenv("GG_SCAN_SPIN", str(spin))
        sa = AdjState(N, K, seed=40 + trial)
        adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=90 + trial, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
        adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
        adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
        adj.resample(d, 2)       # (no error any more)
        ps = sa.p_self
        for s in range(2):
            ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, 1, 90 + trial, s)
        assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c)
        outcomes.append("exact")
    # the library is usable afterwards: the plain forms on the same data
    print("outcomes by spin limit:", outcomes)
    for k in ("GG_SCAN", "GG_SCAN_LCAP", "GG_SCAN_SPIN"):
        monkeypatch.delenv(k)
    sa = AdjState(N, K, seed=44)
    run_adj(abi, A, sa, 95, 2)      # (asserts the chain against the oracle itself)
    assert outcomes and all(o == "exact" for o in outcomes)
    err = capfd.readouterr().err
    print("recounts:", err.count("counted again"), "plain timeouts:", err.count("timed out") - err.count("counted again"))
    # (which trials give up behind a flush is a matter of timing - one of the eight in the round's runs; whichever path a trial
    # takes, it must end in the oracle's chain, which is what is asserted above)


@pytest.mark.gpu
def test_adaptive_scan_form_is_the_same_chain(abi, monkeypatch):
    """Default (GG_SCAN unset): the engine runs the scan as k_scan while the chain keeps changing labels and as
    k_speculate + k_resolve after a run of quiet sweeps (4 at first), deciding batch by batch.  A strongly planted network burns in within a
    few sweeps and then stays quiet, so 240 sweeps in uneven chunks cross both switches: same chain as the oracle bit for
    bit at every checkpoint."""
    monkeypatch.delenv("GG_SCAN", raising=False)
    N, K = 260, 4
    A, W, _ = planted_sbm(N, K, seed=77)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=15), WState(N, K, 1, seed=16)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=91, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=92, special_case_self_conns=1)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    fs = O.FastSweeper(A, q, shift, threads=4)
    ps, done = sa.p_self, 0
    changes = []
    for chunk in (1, 2, 5, 40, 100, 3, 89):
        abi.resample_many([adj, w], d, chunk)
        for s in range(done, done + chunk):
            ps = fs.adj_sweep(sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, 1, 91, s)
            fs.w_sweep(sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, 1, 92, s)
        done += chunk
        assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c), done
        assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c), done
        assert np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p) and np.array_equal(w.get(abi.ST_MU, (K, K, 1)), sw.mu), done
        changes.append(adj.stats()["label_changes"] + w.stats()["label_changes"])
    fs.close()
    assert changes[0] > 0                       # burn-in changed labels (k_scan from the second sweep on)
    assert changes[-1] == changes[-3]           # ... and the last 92 sweeps none: back on speculate + resolve by then


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,B,beta,special", [(140, 4, 1, 1.0, 1), (180, 5, 2, 1.0, 1), (120, 3, 2, 0.5, 1),
                                                (150, 4, 1, 1.0, 0), (130, 3, 2, 0.6, 0)])
def test_shared_covariance_weights_bitexact(abi, N, K, B, beta, special):
    """SBMGaussianWeightSharedCov (weights.py:501-552): means given the shared covariance, then the covariance from the
    centred edges, then the parent's scan: bit-exact against the oracle over several sweeps."""
    A, W, _ = planted_sbm(N, K, B=B, seed=41)
    assert np.diag(A).sum() > 3        # self edges present: the two settings differ
    q, shift = O.quantise(A, W)
    st = WState(N, K, B, seed=8)
    st.sigma[:] = st.sigma[0, 0]
    prior = O.pack_niw(np.full(B, 0.2), np.eye(B) * 1.3, B + 2, 1.0)
    d = abi.DeviceData(A, W)
    w = abi.Model(kind=abi.SBM_GAUSS_SHCOV_W, N=N, K=K, B=B, seed=83, special_case_self_conns=special)
    w.set(abi.ST_C, st.c, np.int32); w.set(abi.ST_M, st.m); w.set(abi.ST_PI, st.pi)
    w.set(abi.ST_MU, st.mu); w.set(abi.ST_SIGMA, st.sigma); w.set(abi.ST_NIW, prior); w.set(abi.ST_NIW_SELF, st.niw_self)
    w.set(abi.ST_MU_SELF, st.mu_self); w.set(abi.ST_SIGMA_SELF, st.sigma_self)   # (left alone without the special case)
    for s in range(4):
        w.resample(d, 1, beta)
        O.sbm_wsh_sweep(A, q, shift, st.c, st.m, st.mu, st.sigma, st.mu_self, st.sigma_self, st.pi, prior, st.niw_self, 83, s, beta=beta,
                        special=special)
        assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), st.c), s
        assert np.array_equal(w.get(abi.ST_MU, (K, K, B)), st.mu)
        sig = w.get(abi.ST_SIGMA, (K, K, B, B))
        assert np.array_equal(sig, st.sigma) and np.array_equal(sig, np.broadcast_to(sig[0, 0], sig.shape))
        assert np.array_equal(w.get(abi.ST_MU_SELF, (B,)), st.mu_self) and np.array_equal(w.get(abi.ST_M, (K,)), st.m)
    ll = w.log_likelihood(d)
    ref = O.sbm_w_loglik(A, q, shift, st.c, st.mu, st.sigma, st.mu_self, st.sigma_self, special)
    assert abs(ll - ref) <= 1e-9 * abs(ref)


@pytest.mark.gpu
def test_shared_covariance_host_class(abi):
    from graphistician_amd.weights import SBMGaussianWeightSharedCov
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    np.random.seed(6)
    N, C, B = 60, 3, 2
    A, W, _ = planted_sbm(N, C, B=B, seed=2)
    net = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, {"C": C, "seed": 1},
                                        SBMGaussianWeightSharedCov, {"C": C, "B": B, "seed": 2})
    lp0 = net.log_likelihood((A, W))
    for _ in range(30):
        net.resample((A, W))
    sig = net.weights._Sigma
    assert np.array_equal(sig, np.broadcast_to(sig[0, 0], sig.shape))                    # one covariance for all pairs
    assert np.allclose(net.weights._cov_model.sigma, sig[0, 0]) and net.weights._cov_model.nu_0 == B + 2
    assert net.log_likelihood((A, W)) > lp0 and np.isfinite(net.log_prior())


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["auto", "0", "1"])
@pytest.mark.parametrize("K,B", [(64, 4), (40, 4), (33, 4), (64, 2)])
def test_weights_with_large_tables_fit_in_lds(abi, monkeypatch, K, B, form):
    """K x K tables of NT = 6 / 15 doubles do not fit in LDS next to the statistics: the kernels read them from global
    memory and (K > 32) split the conditional over four chains.  Every scan form has to fit its LDS carve for these
    shapes (k_round shares one region between the dead integer statistics and the chains' meeting place; a resident
    scan that does not fit is switched off at creation) and give the oracle's chain."""
    if form != "auto":
        monkeypatch.setenv("GG_SCAN", form)
    N = 160
    A, W, _ = planted_sbm(N, K, B=B, seed=3)
    q, shift = O.quantise(A, W)
    st = WState(N, K, B, seed=5)
    d = abi.DeviceData(A, W)
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=9, special_case_self_conns=1)
    w.set(abi.ST_C, st.c, np.int32); w.set(abi.ST_M, st.m); w.set(abi.ST_PI, st.pi)
    w.set(abi.ST_MU, st.mu); w.set(abi.ST_SIGMA, st.sigma); w.set(abi.ST_NIW, st.niw); w.set(abi.ST_NIW_SELF, st.niw_self)
    w.set(abi.ST_MU_SELF, st.mu_self); w.set(abi.ST_SIGMA_SELF, st.sigma_self)
    for s in range(3):
        w.resample(d, 1)
        O.sbm_w_sweep(A, q, shift, st.c, st.m, st.mu, st.sigma, st.mu_self, st.sigma_self, st.pi, st.niw, st.niw_self, 1, 9, s)
        assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), st.c), "sweep %d" % s
    assert np.array_equal(w.get(abi.ST_MU, (K, K, B)), st.mu) and np.array_equal(w.get(abi.ST_SIGMA, (K, K, B, B)), st.sigma)
    # a refused call must not poison the ones after it (sticky HIP error): the handle is still usable
    with pytest.raises(abi.EngineError):
        abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=5, seed=1)
    w.resample(d, 1)
    O.sbm_w_sweep(A, q, shift, st.c, st.m, st.mu, st.sigma, st.mu_self, st.sigma_self, st.pi, st.niw, st.niw_self, 1, 9, 3)
    assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), st.c)


@pytest.mark.gpu
def test_timeout_right_after_the_last_node_changed(abi, monkeypatch):
    """Found by tests/fuzz_parity.py: N = K = 17 moves the last node often; a resident scan that gives up on the exchange
    confirming that change used to leave the scan stalled at pos == N, where the restart rounds have nothing to do."""
    from tests.fuzz_parity import one_case
    monkeypatch.setenv("GG_SCAN_SPIN", "2")
    for seed in (702605259, 702605260, 702605261):
        assert one_case(17, 17, 1, 1, seed, (1, 3, 10)) is None


@pytest.mark.gpu
def test_shared_covariance_kind_in_the_ahead_form(abi, monkeypatch):
    """Found by the extended fuzz of the release soak: the quiet chain's fused launch ran the first half of the
    shared-covariance parameter step (means, centred scatter) and left out the second (the covariance draw and the tables,
    `k_shared_cov`, weights.py:545-552) - visible as soon as such a chain was quiet for two sweeps in a row.  The two cases
    it was found with, and a larger network that settles within the call; with and without the ahead form."""
    from tests.fuzz_parity import one_case
    for ahead in ("1", "0"):
        monkeypatch.setenv("GG_AHEAD", ahead)
        assert one_case(5, 5, 4, 1, 154869791, (1, 2, 9), 0.5, True) is None
        assert one_case(17, 17, 4, 1, 567335410, (3, 12), 1.0, True) is None
        assert one_case(200, 4, 2, 1, 99, (1, 3, 40), 1.0, True) is None
        assert one_case(129, 3, 1, 0, 7, (2, 30), 0.83, True) is None


@pytest.mark.gpu
def test_short_parity_fuzz(abi):
    """Thirty random shapes / conventions / chunkings, engine against the oracle bit for bit (tests/fuzz_parity.py; longer
    streams under every scan-form switch are part of the release soak, profiles/README.md)."""
    from tests.fuzz_parity import run
    assert run(30, 20261003) == []
    assert run(40, 20261004, extended=True) == []     # tempered sweeps and shared-covariance weights as well


@pytest.mark.gpu
def test_invalid_arguments_are_errors_not_hangs(abi):
    """Values the samplers cannot digest (a NaN or a non-positive shape never leaves a rejection loop: a hung GPU) and
    mismatched shapes are refused at the C ABI with GG_ERR_INVALID / UNSUPPORTED; the library stays usable."""
    N, K = 50, 3
    A, W, _ = planted_sbm(N, K, seed=1)
    d, dA = abi.DeviceData(A, W), abi.DeviceData(A, None)
    A2, W2, _ = planted_sbm(60, K, seed=1)
    d2 = abi.DeviceData(A2, W2)

    def adj(**kw):
        cfg = dict(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=1, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=1)
        cfg.update(kw)
        return abi.Model(**cfg)

    m = adj()
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=2, seed=2, special_case_self_conns=1)
    eye = np.eye(2)
    bad_calls = [
        lambda: m.resample(d2, 1),                                          # data of another size
        lambda: m.log_likelihood(d2),
        lambda: m.resample(d, 1, -0.5), lambda: m.resample(d, 1, 1.5), lambda: m.resample(d, 1, float("nan")),
        lambda: m.set(abi.ST_C, np.full(N, K), np.int32), lambda: m.set(abi.ST_C, np.full(N, -1), np.int32),
        lambda: m.set(abi.ST_P, np.full((K, K), 1.5)), lambda: m.set(abi.ST_P, np.full((K, K), np.nan)),
        lambda: m.set(abi.ST_PI, np.array([-1.0, 1.0, 1.0])), lambda: m.set(abi.ST_PI, np.array([0.0, 1.0, 1.0])),
        lambda: m.set(abi.ST_M, np.array([np.inf, 0.0, 0.0])),
        lambda: m.set(abi.ST_P, np.zeros(2)), lambda: m.set(999, np.zeros(2)), lambda: m.get(999, (2,)),
        lambda: adj(tau1=0.0), lambda: adj(tau0=-1.0), lambda: adj(tau1=float("nan")),
        lambda: adj(K=0), lambda: adj(K=257), lambda: adj(N=0), lambda: adj(device=99),
        lambda: w.resample(dA, 1),                                          # weight model, data without W
        lambda: w.set(abi.ST_SIGMA, -np.tile(eye, (K, K, 1, 1))),
        lambda: w.set(abi.ST_SIGMA, np.tile(np.array([[1.0, 2.0], [2.0, 1.0]]), (K, K, 1, 1))),   # indefinite
        lambda: w.set(abi.ST_MU, np.full((K, K, 2), np.nan)),
        lambda: w.set(abi.ST_NIW, O.pack_niw(np.zeros(2), eye, 0.5, 1.0)),  # nu_0 <= B - 1
        lambda: w.set(abi.ST_NIW, O.pack_niw(np.zeros(2), eye, 4.0, 0.0)),  # kappa_0 = 0
        lambda: w.set(abi.ST_NIW, O.pack_niw(np.zeros(2), -eye, 4.0, 1.0)),
        lambda: abi.Model(kind=abi.LOGISTIC_EIG, N=N, D=2, K=1, B=1, seed=1, sigma_mu0=1.0, sigma_F=0.0, sigma_lmbda=1.0, mu_mu_0=0.0),
        lambda: abi.Model(kind=abi.LOGISTIC_EIG, N=N, D=9, K=1, B=1, seed=1, sigma_mu0=1.0, sigma_F=1.0, sigma_lmbda=1.0, mu_mu_0=0.0),
        lambda: abi.Model(kind=abi.LATENT_DIST, N=N, D=2, K=1, B=1, seed=1, sigma_F=-1.0),
    ]
    for i, call in enumerate(bad_calls):
        with pytest.raises(abi.EngineError):
            call()
            pytest.fail("bad call %d was accepted" % i)
    e = abi.Model(kind=abi.LOGISTIC_EIG, N=N, D=2, K=1, B=1, seed=1, sigma_mu0=1.0, sigma_F=1.0, sigma_lmbda=1.0, mu_mu_0=0.0)
    with pytest.raises(abi.EngineError):
        e.set(abi.ST_F, np.full((N, 2), np.nan))
    Wn, An = W.copy(), A.copy()
    An[0, 1] = 1; Wn[0, 1] = np.nan
    with pytest.raises(abi.EngineError):
        abi.DeviceData(An, Wn)
    # zero sweeps, a sync without work and a zero entry in m are fine; the handles are still usable after all that
    m.resample(d, 0); m.sync()
    m.set(abi.ST_M, np.array([0.5, 0.5, 0.0]))
    m.resample(d, 2)
    assert m.get(abi.ST_C, (N,), np.int32).max() < K


@pytest.mark.gpu
def test_every_adjacency_and_weight_class_combination_runs(abi):
    """9 adjacency classes x 4-5 weight classes x B in {1, 2} through FactorizedNetworkDistribution: rvs, three sweeps,
    log_likelihood / log_prior / log_probability finite, predictive parameters (tests/combo_smoke.py)."""
    from tests.combo_smoke import main
    assert main() == 0


@pytest.mark.gpu
def test_predictive_ll_on_the_device_matches_the_reference_formula(abi):
    """approx_predictive_ll (abstractions.py:157-182, 313-357) evaluated by gg_predictive_ll: with m concentrated on one block
    every predictive draw puts the held-out node there, so the generic host code (the reference's double loop over draws and
    nodes, graphistician_amd/abstractions.py) and the engine must return the same number, for every block, for both
    self-connection conventions, for the K = 1 kinds, and the M-draw average must be the log-mean-exp over sampled blocks."""
    from scipy.special import logsumexp
    from graphistician_amd.abstractions import AdjacencyDistribution, GaussianWeightDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution, BetaBernoulliAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution, NIWGaussianWeightDistribution
    np.random.seed(5)
    N, C, B = 37, 4, 2
    Arow, Acol = np.random.rand(N + 1) < 0.4, np.random.rand(N + 1) < 0.6
    Acol[-1] = Arow[-1] = True
    Wrow, Wcol = np.random.randn(N + 1, B), np.random.randn(N + 1, B)
    Wcol[-1] = Wrow[-1]
    for special, allow in ((True, True), (False, True), (False, False)):
        a = SBMAdjacencyDistribution(N, C=C, special_case_self_conns=special, allow_self_connections=allow)
        w = SBMGaussianWeightDistribution(N, B=B, C=C, special_case_self_conns=special)
        for k in range(C):
            a.m = np.eye(C)[k]; w.m = np.eye(C)[k]
            ref_a = AdjacencyDistribution.approx_predictive_ll(a, Arow, Acol, M=3)
            assert abs(a.approx_predictive_ll(Arow, Acol, M=3) - ref_a) <= 1e-10 * abs(ref_a), (special, allow, k)
            ref_w = GaussianWeightDistribution.approx_predictive_ll(w, Arow, Acol, Wrow, Wcol, M=3)
            assert abs(w.approx_predictive_ll(Arow, Acol, Wrow, Wcol, M=3) - ref_w) <= 1e-10 * abs(ref_w), (special, k)
        # a proper m: the estimate is the log-mean-exp of the per-block values over the sampled blocks
        a.m = np.array([0.1, 0.2, 0.3, 0.4])
        ll_k = a._engine().predictive_ll(Arow, Acol, K=C)
        np.random.seed(11)
        got = a.approx_predictive_ll(Arow, Acol, M=50)
        np.random.seed(11)
        c2 = np.random.choice(C, p=a.m, size=50)
        assert abs(got - (-np.log(50) + logsumexp(ll_k[c2]))) < 1e-12
    bb = BetaBernoulliAdjacencyDistribution(N)
    ref = AdjacencyDistribution.approx_predictive_ll(bb, Arow, Acol, M=2)
    assert abs(bb.approx_predictive_ll(Arow, Acol) - ref) <= 1e-10 * abs(ref)
    niw = NIWGaussianWeightDistribution(N, B=B)
    ref = GaussianWeightDistribution.approx_predictive_ll(niw, Arow, Acol, Wrow, Wcol, M=2)
    assert abs(niw.approx_predictive_ll(Arow, Acol, Wrow, Wcol) - ref) <= 1e-10 * abs(ref)


def test_quantisation_guard_reports_a_coarse_scale(abi):
    """the weights are stored as 24-bit fixed point with ONE shift, fixed by the largest |w| on an edge - 24 bits
    absolute where the reference's float64 keeps 53 relative (abstractions.py:261).  One outlier edge (|w| = 1e3 among
    N(0, 0.01)) leaves the bulk of the data a handful of bits: gg_data_quantisation reports it and the binding warns; well-scaled
    weights keep >= 20 bits and are silent.  The log-likelihood then shows the loss the guard is there to announce."""
    import warnings
    from oracle import oracle as O
    N, K = 96, 3
    A, W, c = planted_sbm(N, K, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        d = abi.DeviceData(A, W)                      # N(mu, 0.7^2): silent
    wmax, wlow, bits, shift = d.quantisation()
    assert bits >= 18 and wmax > wlow > 0
    W2 = W * 0.01
    i, j = np.argwhere(A == 1)[5]
    W2[i, j, 0] = 1.0e3                               # one outlier sets the scale
    with pytest.warns(RuntimeWarning, match="significant bits"):
        d2 = abi.DeviceData(A, W2)
    wmax2, wlow2, bits2, shift2 = d2.quantisation()
    assert wmax2 == 1.0e3 and bits2 < 12 and shift2 < shift
    # what the stored network is worth: round-trip error of a typical weight, relative
    A_back, W_back = d2.download()
    typical = (A == 1) & (np.abs(W2[:, :, 0]) < 1.0)
    rel = np.abs(W_back[:, :, 0][typical] - W2[:, :, 0][typical]) / np.abs(W2[:, :, 0][typical]).clip(1e-9)
    assert np.median(rel) > 1e-4                      # (24 bits relative would be 6e-8)
    assert abi.DeviceData(A, None).quantisation()[2] == -1


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,B,special", [(300, 4, 1, 1), (257, 5, 2, 0), (400, 70, 1, 1)])
def test_call_epilogue_is_the_plain_chain_and_the_plain_log_likelihood(abi, N, K, B, special):
    """The reference's loop `model.resample(data); lps.append(model.log_probability(data))` (examples/
    demo_model_comparison.py:69-72), one sweep per call: a quiet call's last launch carries the call's epilogue - resample_m,
    the control block for the host and, once the caller has asked for a log-likelihood right after a resample, the
    log-likelihood of the state the call leaves (k_spec_resolve_epi).  Labels, m and parameters stay the oracle's chain bit
    for bit through burn-in and the quiet phase; the value the epilogue leaves is, to the last bit, the one the plain
    evaluation (fresh pass, k_loglik) gives for the same state; a state change in between drops it."""
    rng = np.random.RandomState(N + K)
    Kt = 3
    ctrue = rng.choice(Kt, size=N)
    ptrue = np.full((Kt, Kt), 0.08) + np.eye(Kt) * 0.8
    A = (rng.rand(N, N) < ptrue[np.ix_(ctrue, ctrue)]).astype(np.uint8)
    W = (rng.randn(Kt, Kt, B) * 3.0)[ctrue][:, ctrue] + 0.3 * rng.randn(N, N, B)
    q, shift = O.quantise(A, W)
    sa, sw = AdjState(N, K, seed=31), WState(N, K, B, seed=32)
    d = abi.DeviceData(A, W)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=701, tau1=1.0, tau0=1.0, allow_self_connections=1, special_case_self_conns=special)
    adj.set(abi.ST_C, sa.c, np.int32); adj.set(abi.ST_M, sa.m); adj.set(abi.ST_PI, sa.pi); adj.set(abi.ST_P, sa.p)
    adj.set(abi.ST_P_SELF, np.array([sa.p_self]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=702, special_case_self_conns=special)
    w.set(abi.ST_C, sw.c, np.int32); w.set(abi.ST_M, sw.m); w.set(abi.ST_PI, sw.pi)
    w.set(abi.ST_MU, sw.mu); w.set(abi.ST_SIGMA, sw.sigma); w.set(abi.ST_NIW, sw.niw); w.set(abi.ST_NIW_SELF, sw.niw_self)
    w.set(abi.ST_MU_SELF, sw.mu_self); w.set(abi.ST_SIGMA_SELF, sw.sigma_self)
    ps = sa.p_self
    lls = []
    for it in range(45):
        abi.resample_many([adj, w], d, 1)
        ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, special, 701, it)
        O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, special, 702, it)
        ll = abi.log_likelihood_many([adj, w], d)
        lls.append(ll)
        ref = (O.sbm_adj_loglik(A, sa.c, sa.p, ps, True, bool(special)),
               O.sbm_w_loglik(A, q, shift, sw.c, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, bool(special)))
        assert abs(ll[0] - ref[0]) <= 1e-9 * abs(ref[0]) + 1e-7 and abs(ll[1] - ref[1]) <= 1e-9 * abs(ref[1]) + 1e-7, (it, ll, ref)
        if it % 7 == 6 or it >= 40:
            assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c), it
            assert np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c), it
            assert np.array_equal(adj.get(abi.ST_M, (K,)), sa.m) and np.array_equal(w.get(abi.ST_M, (K,)), sw.m), it
            assert np.array_equal(adj.get(abi.ST_P, (K, K)), sa.p) and np.array_equal(w.get(abi.ST_MU, (K, K, B)), sw.mu), it
    # a request every SECOND call (the AIS loop's pattern, graphistician_amd/ais.py): the epilogue follows the period
    it = 45
    for rep in range(5):
        for _ in range(2):
            abi.resample_many([adj, w], d, 1)
            ps = O.sbm_adj_sweep(A, sa.c, sa.m, sa.p, ps, sa.pi, 1.0, 1.0, special, 701, it)
            O.sbm_w_sweep(A, q, shift, sw.c, sw.m, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, sw.pi, sw.niw, sw.niw_self, special, 702, it)
            it += 1
        ll = abi.log_likelihood_many([adj, w], d)
        lls.append(ll)
        ref = (O.sbm_adj_loglik(A, sa.c, sa.p, ps, True, bool(special)),
               O.sbm_w_loglik(A, q, shift, sw.c, sw.mu, sw.sigma, sw.mu_self, sw.sigma_self, bool(special)))
        assert abs(ll[0] - ref[0]) <= 1e-9 * abs(ref[0]) + 1e-7 and abs(ll[1] - ref[1]) <= 1e-9 * abs(ref[1]) + 1e-7, (rep, ll, ref)
    assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), sa.c) and np.array_equal(w.get(abi.ST_C, (N,), np.int32), sw.c)
    # the same state evaluated the plain way: setting the labels again (to what they are) drops everything counted ahead
    # and the epilogue's value with it
    adj.set(abi.ST_C, sa.c, np.int32); w.set(abi.ST_C, sw.c, np.int32)
    plain = abi.log_likelihood_many([adj, w], d)
    assert plain[0] == lls[-1][0] and plain[1] == lls[-1][1], (plain, lls[-1])
    # ... and a changed state is not answered from the record: move one node, compare with the oracle
    c2 = sa.c.copy(); c2[0] = (c2[0] + 1) % K
    adj.set(abi.ST_C, c2, np.int32)
    ll2 = adj.log_likelihood(d)
    ref2 = O.sbm_adj_loglik(A, c2, sa.p, ps, True, bool(special))
    assert abs(ll2 - ref2) <= 1e-9 * abs(ref2) + 1e-7 and ll2 != plain[0]
    st = adj.stats()
    assert st["sweeps_two_launch"] > 0     # the quiet phase was reached: calls ended in the epilogue form


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,B,special,shared", [(120, 3, 1, True, False), (90, 5, 2, False, False), (200, 16, 1, True, False),
                                                  (150, 70, 1, True, False), (100, 4, 2, True, True), (80, 1, 3, True, False)])
def test_log_prior_on_the_device_is_the_host_formula(abi, N, K, B, special, shared):
    """log_prior of the SBM classes (adjacency.py:538-555, weights.py:333-357, :554-577) evaluated by the engine from the state
    where it lives (gg_log_prior_many: what `log_probability` adds once per sweep in the reference's loops) against the scipy /
    numpy formula of the host classes, after a few sweeps, for both halves at once and one at a time; also with a label that
    no node carries (its log m enters the Dirichlet term only) and after host-side edits of the state."""
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution, SBMGaussianWeightSharedCov
    A, W, _ = planted_sbm(N, max(K, 2) if K < 64 else 8, B=B, seed=N)
    np.random.seed(N + 1)
    wcls = SBMGaussianWeightSharedCov if shared else SBMGaussianWeightDistribution
    model = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, dict(C=K, tau1=1.3, tau0=0.8, pi=2.0, special_case_self_conns=special, seed=3),
                                          wcls, dict(B=B, C=K, special_case_self_conns=special, seed=4))
    host0 = model.adjacency._log_prior_host() + model.weights._log_prior_host()
    assert abs(model.log_prior() - host0) <= 1e-12 * abs(host0)          # (nothing on the engine yet: the host formula itself)
    for it in range(4):
        model.resample((A, W))
        dev = model.log_prior()
        host = model.adjacency._log_prior_host() + model.weights._log_prior_host()
        assert np.isfinite(dev) and abs(dev - host) <= 1e-10 * abs(host), (it, dev, host)
        assert abs(model.adjacency.log_prior() - model.adjacency._log_prior_host()) <= 1e-10 * abs(host)
        assert abs(model.weights.log_prior() - model.weights._log_prior_host()) <= 1e-10 * abs(host)
        lp = model.log_probability((A, W))
        assert abs(lp - (model.log_likelihood((A, W)) + host)) <= 1e-9 * abs(lp)
    if K > 1:
        c = model.adjacency.c.copy()
        c[c == 0] = 1                                   # block 0 is empty now
        model.adjacency.c = c
        m = np.random.dirichlet(np.ones(K) * 3.0)
        model.weights.m = m
        dev, host = model.log_prior(), model.adjacency._log_prior_host() + model.weights._log_prior_host()
        assert abs(dev - host) <= 1e-10 * abs(host), (dev, host)
