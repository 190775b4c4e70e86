#!/usr/bin/env python3
"""bench.py — full Gibbs sweeps/sec on the Gaussian-SBM configuration of BASELINE.json.

One *step* = one full sweep = FactorizedNetworkDistribution.resample((A,W))
(reference networks.py:69-71): the SBM adjacency half (adjacency.py:615-618)
followed by the SBM Gaussian-weight half (weights.py:394-397), on a synthetic
planted Gaussian SBM resident in HBM.  N=1 runs BASELINE.json configs[1]
(N=4096, K=16, B=1, dense) and reports a MIX of chains: how many labels a chain
keeps changing per sweep depends on its prior draw, so `value` = all timed sweeps /
all timed seconds over chain seeds 1..6 (K sweeps each), the quiet chain alone in
`quiet_chain`.  With --gpus G > 1 (one process per GPU, launched by
torch.distributed.run) the workload is BASELINE.json configs[3] (N=65536, K=64):
every rank runs an independent replica of the chain on its own copy of the network
(north_star: "otherwise independent chains map one-per-GPU"), no data-path
collective, value = total sweeps / max-over-ranks time (weak scaling) - and the
same line carries, under `sharded_chain`, ONE chain of that network sharded over
the G GPUs (exact; RCCL inside the library; strong scaling).  --shard prints the
sharded chain as the line itself.

Prints ONE JSON line (rank 0).  torch is used only for the multi-process
rendezvous / barrier; the engine itself is ctypes -> libgraphgibbs.so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_JSON_OUT = None


def _claim_stdout():
    """stdout carries exactly ONE line, the JSON result: the descriptor is kept aside for it and everything else that
    writes to fd 1 from here on (RCCL greets with a version banner when its first communicator comes up, for one) lands on
    stderr."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def _emit(line):
    out = _JSON_OUT if _JSON_OUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def _progress(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable


def planted_truth(N, K, seed):
    """Planted Gaussian SBM parameters (SURVEY.md §8d cfg 2 recipe)."""
    rng = np.random.RandomState(seed)
    m = rng.dirichlet(np.ones(K) * 10.0)
    c = rng.choice(K, p=m, size=N).astype(np.int32)
    p = rng.beta(1.0, 1.0, size=(K, K))
    p_self = float(rng.beta(1.0, 1.0))
    sigma = 1.0 / rng.gamma(1.5, 1.0 / 0.5, size=(K, K))   # IW(1, nu=3) in one dimension: 1/Gamma(nu/2, rate 1/2)
    mu = rng.randn(K, K) * np.sqrt(sigma)
    return dict(m=m, c=c, p=p, p_self=p_self, mu=mu, sigma=sigma, mu_self=0.0, sigma_self=1.0)


def build_models(abi, N, K, chain_seed, device):
    rng = np.random.RandomState(chain_seed)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=chain_seed * 2 + 1, tau1=1.0, tau0=1.0,
                    allow_self_connections=1, special_case_self_conns=1, device=device)
    pi_a = np.ones(K)
    m_a = rng.dirichlet(pi_a)
    adj.set(abi.ST_PI, pi_a)
    adj.set(abi.ST_M, m_a)
    adj.set(abi.ST_C, rng.choice(K, p=m_a, size=N).astype(np.int32), np.int32)
    adj.set(abi.ST_P, rng.beta(1.0, 1.0, size=(K, K)))
    adj.set(abi.ST_P_SELF, np.array([rng.beta(1.0, 1.0)]))
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=chain_seed * 2 + 2, special_case_self_conns=1, device=device)
    pi_w = np.ones(K) * 10.0
    m_w = rng.dirichlet(pi_w)
    w.set(abi.ST_PI, pi_w)
    w.set(abi.ST_M, m_w)
    w.set(abi.ST_C, rng.choice(K, p=m_w, size=N).astype(np.int32), np.int32)
    w.set(abi.ST_MU, rng.randn(K, K, 1))
    w.set(abi.ST_SIGMA, np.ones((K, K, 1, 1)))
    niw = np.array([0.0, 1.0, 3.0, 1.0])
    w.set(abi.ST_NIW, niw)
    w.set(abi.ST_NIW_SELF, niw)
    return adj, w


def cpu_baseline(abi, data, N, K, budget_s):
    """The CPU oracle on the same workload, both compiled flavours: `value` is the optimised one
    (oracle_fast.cpp: OpenMP statistics pass + O(K^2)-per-node scan, all host threads), `scalar_port` the plain
    single-threaded restatement that recounts every node's row and column like the reference does."""
    from oracle import oracle as O
    A, W = data.download()
    q, shift = O.quantise(A, W)
    niw = O.pack_niw([0.0], [[1.0]], 3.0, 1.0)

    def fresh():
        rng = np.random.RandomState(7)
        return dict(c_a=rng.choice(K, size=N).astype(np.int32), m_a=np.ones(K) / K,
                    p=np.ascontiguousarray(rng.beta(1, 1, size=(K, K))), ps=0.5,
                    c_w=rng.choice(K, size=N).astype(np.int32), m_w=np.ones(K) / K,
                    mu=np.zeros((K, K, 1)), sig=np.ones((K, K, 1, 1)), mus=np.zeros(1), sigs=np.ones((1, 1)))

    # optimised flavour: a few untimed sweeps first (the GPU number is a steady-state number too)
    st = fresh()
    fs = O.FastSweeper(A, q, shift)
    _progress("cpu baseline: optimised flavour on %d threads" % fs.threads)

    def fast_sweep(i):
        st["ps"] = fs.adj_sweep(st["c_a"], st["m_a"], st["p"], st["ps"], np.ones(K), 1.0, 1.0, 1, 11, i)
        fs.w_sweep(st["c_w"], st["m_w"], st["mu"], st["sig"], st["mus"], st["sigs"], np.ones(K) * 10, niw, niw, 1, 12, i)

    n_warm, t0 = 0, time.perf_counter()
    while n_warm < 10 and time.perf_counter() - t0 < 0.3 * budget_s:
        fast_sweep(n_warm); n_warm += 1
    n, t0 = 0, time.perf_counter()
    while True:
        fast_sweep(n_warm + n); n += 1
        el = time.perf_counter() - t0
        if el > 0.35 * budget_s or n >= 200:
            break
    out = dict(value=n / el, unit="sweeps/s", cores=int(fs.threads), kind="port",
               sample="%d full sweeps (adjacency + weights) after %d untimed ones on the same N=%d K=%d network, oracle/oracle_fast.cpp "
                      "(C++ -O2, OpenMP, %d threads)" % (n, n_warm, N, K, fs.threads))
    fs.close()
    # parity, checked in the same run: the engine runs the very chain the CPU just ran (same start, same Philox keys, same
    # number of sweeps) and must land on the same labels and parameters bit for bit
    _progress("parity check: the engine on the CPU baseline's chain (%d sweeps)" % (n_warm + n))
    f0 = fresh()
    ea = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=11, tau1=1.0, tau0=1.0, allow_self_connections=1,
                   special_case_self_conns=1, device=data.device)
    ea.set(abi.ST_C, f0["c_a"], np.int32); ea.set(abi.ST_M, f0["m_a"]); ea.set(abi.ST_PI, np.ones(K)); ea.set(abi.ST_P, f0["p"])
    ea.set(abi.ST_P_SELF, np.array([f0["ps"]]))
    ew = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=12, special_case_self_conns=1, device=data.device)
    ew.set(abi.ST_C, f0["c_w"], np.int32); ew.set(abi.ST_M, f0["m_w"]); ew.set(abi.ST_PI, np.ones(K) * 10)
    ew.set(abi.ST_MU, f0["mu"]); ew.set(abi.ST_SIGMA, f0["sig"]); ew.set(abi.ST_NIW, niw); ew.set(abi.ST_NIW_SELF, niw)
    abi.resample_many([ea, ew], data, n_warm + n)
    same = dict(adjacency_labels=bool(np.array_equal(ea.get(abi.ST_C, (N,), np.int32), st["c_a"])),
                weight_labels=bool(np.array_equal(ew.get(abi.ST_C, (N,), np.int32), st["c_w"])),
                p=bool(np.array_equal(ea.get(abi.ST_P, (K, K)), st["p"])),
                mu=bool(np.array_equal(ew.get(abi.ST_MU, (K, K, 1)), st["mu"])),
                sigma=bool(np.array_equal(ew.get(abi.ST_SIGMA, (K, K, 1, 1)), st["sig"])))
    out["parity"] = dict(checked=True, bit_exact=all(same.values()), sweeps=int(n_warm + n), fields=same,
                         label_changes=int(ea.stats()["label_changes"] + ew.stats()["label_changes"]),
                         labels_digest=int(np.bitwise_xor.reduce(st["c_a"].astype(np.int64) * np.arange(1, N + 1))))
    del ea, ew
    _progress("cpu baseline: scalar flavour")
    # plain scalar restatement
    st = fresh()
    n, t0 = 0, time.perf_counter()
    while True:
        st["ps"] = O.sbm_adj_sweep(A, st["c_a"], st["m_a"], st["p"], st["ps"], np.ones(K), 1.0, 1.0, 1, 11, n)
        O.sbm_w_sweep(A, q, shift, st["c_w"], st["m_w"], st["mu"], st["sig"], st["mus"], st["sigs"], np.ones(K) * 10, niw, niw, 1, 12, n)
        n += 1
        el = time.perf_counter() - t0
        if el > 0.35 * budget_s or n >= 50:
            break
    out["scalar_port"] = dict(value=n / el, unit="sweeps/s", cores=1, kind="port",
                              sample="%d full sweeps, oracle/oracle_sbm.cpp (C++ -O2, 1 thread, per-node recount as in the reference)" % n)
    return out


def cpu_literal_extrapolated(N, K, seed=3):
    """Cost of the reference ALGORITHM as written (SURVEY.md §8d baseline (i)): the loop-for-loop numpy/scipy
    transliteration (oracle/literal.py) timed on a small network and scaled by the known operation counts:
    adjacency scan N * (K N) loop iterations (adjacency.py:661-693), block-pair counts K^2 masks of N^2
    (adjacency.py:624-634), weight scan N * K^2 masked gathers of length N (weights.py:451-470)."""
    from oracle import literal as R
    K0 = 4
    rng = np.random.RandomState(seed)

    def probe(N0):
        c = rng.choice(K0, size=N0).astype(np.int32)
        A = (rng.rand(N0, N0) < 0.5).astype(np.uint8)
        W = rng.randn(N0, N0, 1)
        m = np.ones(K0) / K0
        p = rng.beta(1, 1, size=(K0, K0))
        mu, sig = rng.randn(K0, K0, 1), np.ones((K0, K0, 1, 1))
        t0 = time.perf_counter(); [R.sbm_adj_node_lp(A, c, m, p, n) for n in (0, 1)]; ta = (time.perf_counter() - t0) / 2
        t0 = time.perf_counter(); R.sbm_adj_counts(A, c, K0); tc = time.perf_counter() - t0
        t0 = time.perf_counter(); [R.sbm_w_node_lp(A, W, c, m, mu, sig, n) for n in (0, 1, 2, 3)]; tw = (time.perf_counter() - t0) / 4
        return ta, tc, tw

    Na, Nb = 96, 192
    ta_a, tc_a, tw_a = probe(Na)
    ta_b, tc_b, tw_b = probe(Nb)
    fK = float(K) / K0
    # adjacency node: K N loop iterations; counts: K^2 masks of N^2; weight node: K^2 (fixed scipy cost + masks of N)
    t_adj_node = ta_b * (float(N) / Nb) * fK
    t_counts = tc_b * (float(N) / Nb) ** 2 * fK * fK
    slope = max((tw_b - tw_a) / (Nb - Na), 0.0)
    t_w_node = (tw_b + slope * (N - Nb)) * fK * fK
    t_adj = N * t_adj_node + t_counts
    t_w = N * t_w_node + t_counts
    return dict(value=1.0 / (t_adj + t_w), unit="sweeps/s", cores=1, kind="port",
                sample="oracle/literal.py (numpy/scipy loops of the reference) timed at N=%d and %d, K=%d, extrapolated to N=%d K=%d "
                       "by operation count: %.0f s adjacency half + %.0f s weight half per sweep" % (Na, Nb, K0, N, K, t_adj, t_w))


def pmc_traffic(N, K, forms):
    """roofline.traffic: HBM bytes per launch of the dominant kernel, FROM THE COMMITTED PROFILE of that launch shape
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 correction:
    tools/scripts/pmc_traffic_r03.sh, tools/pmc_traffic.py) - counters cannot be collected inside this process.  Only when the
    run is the profiled shape (N=4096, K=16, the fused launch for most sweeps); null otherwise."""
    try:
        fused = forms.get("sweeps_fused", 0)
        if (N, K) == (4096, 16) and fused > forms.get("sweeps_serial", 0) + forms.get("sweeps_two_launch", 0):
            e = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic_n4096.json")))["k_sweep_q<3"]
            return {"traffic": e["traffic_bytes"], "traffic_source": "from_profile: profiles/r04_pmc_traffic_n4096.json (round 4: tools/scripts/pmc_traffic_r04.sh, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes)",
                    "note": "k_sweep_q<3,false,0,0> at N=4096: %.1f MB read (FETCH_SIZE x 2) + %.1f MB written per launch against %.1f MB "
                            "algorithmic (ratio %.3f); the 138 MB working set stays in the 256 MiB Infinity Cache, so these memory-side "
                            "counters move with what else is resident (0.82 in round 3's passes)" % (e["hbm_read_bytes_corrected"] / 1e6, e["hbm_write_bytes"] / 1e6,
                                                              e["algorithmic_bytes_per_launch"] / 1e6, e["traffic_over_algorithmic"])}
        return {"traffic": None, "traffic_source": None,
                "note": "no PMC profile of this launch shape on file (N=%d, K=%d, forms %s); profiles/r04_pmc_traffic_n32768.json holds "
                        "k_bulk_w<1,4,true> at N=32768: 8.74 GB per launch against 8.59 GB algorithmic" % (N, K, forms)}
    except Exception as exc:
        return {"traffic": None, "traffic_source": None, "note": "no PMC summary on file (%s)" % exc}


def run_chain(abi, data, N, K, chain_seed, device, burnin, warmup, steps, every, barrier=lambda: None, stepwise=False, repeats=1,
              max_over_ranks=lambda x: x):
    """One Gaussian-SBM chain (adjacency + weight models) on `data`: set-up burn-in, `warmup` untimed sweeps, then `repeats`
    timed BLOCKS of exactly `steps` sweeps each (with repeats > 1 one more block of `steps` runs untimed in front of them).
    One step = one FactorizedNetworkDistribution.resample((A,W)): the two halves are independent models (networks.py:69-71,
    SURVEY.md 3.1) enqueued stream-ordered on their own HIP streams; a timed block is bracketed by the barrier on both sides
    and ends when both streams have drained (gg_sync).  A 20-sweep block lasts about a millisecond, so a single one is
    start / stop sensitive: the caller takes the MEDIAN block and reports the spread."""
    adj, w = build_models(abi, N, K, chain_seed=chain_seed, device=device)
    if burnin > 0:          # set-up: out of the prior draw's transient (SBM config: --burnin)
        abi.resample_many([adj, w], data, burnin)
    for _ in range(warmup):
        adj.resample(data, 1)
        w.resample(data, 1)

    def block():
        if stepwise:
            for _ in range(steps):
                adj.resample_async(data, 1)
                w.resample_async(data, 1)
            abi.sync_many([adj, w])
        else:
            abi.resample_many([adj, w], data, steps)

    if repeats > 1:
        block()             # one warm block: the engine's run-ahead depth and scan form have settled for this call length
    st0a, st0w = adj.stats(), w.stats()
    chg0 = st0a["label_changes"] + st0w["label_changes"]
    rnd0 = st0a["resolve_rounds"] + st0w["resolve_rounds"]
    blocks = []
    bulk = dict(adj_ms=0.0, adj_n=0, adj_bytes=0.0, w_ms=0.0, w_n=0, w_bytes=0.0)
    for b in range(repeats):
        # HIP events around the sampled launches (each record drains the engine's queue: the instrumentation costs the quiet
        # chains a few microseconds per sweep) in every third block only - the roofline is still measured live, inside timed
        # blocks of this very run, and the median block is one the instrumentation did not touch
        on = every if (repeats < 3 or b % 3 == 0) else 0
        adj.set_timing(on)
        w.set_timing(on)
        barrier()
        t0 = time.perf_counter()
        block()
        barrier()
        blocks.append(max_over_ranks(time.perf_counter() - t0))
        ba, bw = adj.stats(), w.stats()       # (bulk_* describe the last call)
        for k, v in (("adj_ms", ba["bulk_ms"]), ("adj_n", ba["bulk_launches"]), ("adj_bytes", ba["bulk_bytes"]),
                     ("w_ms", bw["bulk_ms"]), ("w_n", bw["bulk_launches"]), ("w_bytes", bw["bulk_bytes"])):
            bulk[k] += v
    sa, sw = adj.stats(), w.stats()
    adj.set_timing(0); w.set_timing(0)
    nsw = float(steps * repeats)
    return dict(adj=adj, w=w, elapsed=float(np.median(blocks)), blocks=blocks, bulk=bulk,
                changes=(sa["label_changes"] + sw["label_changes"] - chg0) / float(repeats),
                rounds=(sa["resolve_rounds"] + sw["resolve_rounds"] - rnd0) / float(repeats),
                changes_adj=sa["label_changes"], changes_w=sw["label_changes"],
                # how the engine ran the weight model's sweeps of the timed blocks (gg_stats: enqueued sweeps by form)
                forms={k: int(sw[k] - st0w[k]) for k in ("sweeps_fused", "sweeps_two_launch", "sweeps_serial")},
                forms_adj={k: int(sa[k] - st0a[k]) for k in ("sweeps_fused", "sweeps_two_launch", "sweeps_serial")},
                ahead=bool(sw["sweeps_fused"] - st0w["sweeps_fused"] > (sw["sweeps_serial"] - st0w["sweeps_serial"])),
                total_timed_s=float(sum(blocks)), sweeps_run=nsw)


def extra_eigen(abi, device, N=4096, D=8, warmup=5, steps=20):
    """BASELINE.json configs[2] (logistic eigenmodel Gibbs, eigenmodel.py:836-857) inside the default run."""
    rng = np.random.RandomState(1234)
    F = rng.randn(N, D); lam = rng.randn(D); mm = -1.2815515655446004
    mu0 = mm + rng.randn()
    Mu = mu0 + (F * lam).dot(F.T)
    np.fill_diagonal(Mu, mu0)
    A = (rng.rand(N, N) < 1.0 / (1.0 + np.exp(-Mu))).astype(np.uint8)
    del Mu
    data = abi.DeviceData(A, None, device=device)
    m = abi.Model(kind=abi.LOGISTIC_EIG, N=N, D=D, K=1, B=1, seed=1, sigma_mu0=1.0, sigma_F=1.0, sigma_lmbda=1.0, mu_mu_0=mm,
                  device=device)
    crng = np.random.RandomState(1)
    m.set(abi.ST_F, crng.randn(N, D)); m.set(abi.ST_LMBDA, crng.randn(D)); m.set(abi.ST_MU0, np.array([mm + crng.randn()]))
    m.set(abi.ST_MU_LMBDA, np.zeros(D))
    m.resample(data, warmup)
    t0 = time.perf_counter()
    m.resample(data, steps)
    m.get(abi.ST_MU0, (1,))
    el = time.perf_counter() - t0
    sweep_bytes = float(N) * N * (3 * 8 + 2 * 2)
    out = {"workload": "Logistic eigenmodel Gibbs (Polya-gamma), N=%d, D=%d (BASELINE.json configs[2])" % (N, D),
           "value": steps / el, "unit": "sweeps/s", "ms_per_step": el / steps * 1e3, "steps": steps,
           "sweep_algorithmic_bytes": sweep_bytes, "effective_GBs": sweep_bytes * steps / el / 1e9,
           "bound": "latency of the sequential latent-feature scan (N dependent D x D solves) + PG(1,z) draws, not HBM (DESIGN.md 7)"}
    out.update(eigen_phases(abi, m, data, N))
    return out


def eigen_phases(abi, m, data, N, reps=5):
    """the eigenmodel's sweep against the bound it is ON - a latency chain, not HBM.  The sweep's parts timed one
    at a time (gg_resample_part, host clock, a synchronisation each): the auxiliary draws use the whole chip; the feature scan is
    ONE workgroup's chain of dependent steps, node after node - its time over N nodes, in shader clocks per node, is the
    number to read (1 155 clocks per scanned node inside the kernel + the tiles' prologues and launch boundaries)."""
    t = {"omega": 0.0, "F": 0.0, "mu0_lambda": 0.0}
    for _ in range(reps):
        for key, parts in (("omega", abi.PART_OMEGA), ("F", abi.PART_F), ("mu0_lambda", abi.PART_MU0 | abi.PART_LMBDA)):
            t0 = time.perf_counter()
            m.resample_part(data, parts)
            t[key] += time.perf_counter() - t0
    ms = {k: v / reps * 1e3 for k, v in t.items()}
    clock_ghz = 2.4          # MI355X_MICROARCH.md: peak engine clock (tools/probe/clock_rate.hip: clock64 ticks at 2.4 GHz)
    return {"phases_ms": ms,
            "latency_bound": {"what": "the feature scan: one workgroup, N dependent steps (perturbation draw + Sherman-Morrison fold per node)",
                              "f_scan_us_per_node": ms["F"] * 1e3 / N, "f_scan_clocks_per_node_at_2p4GHz": ms["F"] * 1e-3 * clock_ghz * 1e9 / N,
                              "share_of_sweep": ms["F"] / max(sum(ms.values()), 1e-9),
                              "note": "in-kernel step 1 155 clocks per scanned node (GG_EIG_DBG); the rest of the per-node figure is the 64 tile "
                                      "prologues and launch boundaries; the auxiliary draws (phases_ms.omega) are divergence-bound on the whole chip"}}


def extra_big(abi, device, N=65536, K=64, burnin=12, warmup=2, steps=20):
    """BASELINE.json configs[3] on ONE GPU (43 GB resident): every bulk launch of the timed region sampled."""
    truth = planted_truth(N, K, seed=1234)
    data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"],
                                        truth["mu_self"], truth["sigma_self"], seed=1234, device=device)
    r = run_chain(abi, data, N, K, 1, device, burnin, warmup, steps, 1)
    per_ms = r["bulk"]["w_ms"] / max(r["bulk"]["w_n"], 1)
    alg = r["bulk"]["w_bytes"] / max(r["bulk"]["w_n"], 1)
    ach = r["bulk"]["w_bytes"] / (r["bulk"]["w_ms"] * 1e-3) / 1e9 if r["bulk"]["w_ms"] > 0 else 0.0
    sweep_bytes = 2.0 * N * N * (0.125 + 4.0)
    out = {"workload": "Gaussian SBM Gibbs, N=%d, K=%d, B=1 (BASELINE.json configs[3] on one GPU)" % (N, K),
           "value": steps / r["elapsed"], "unit": "sweeps/s", "ms_per_step": r["elapsed"] / steps * 1e3, "steps": steps,
           "burn_in_sweeps": burnin, "label_changes_per_sweep": r["changes"] / float(steps),
           "sweep_algorithmic_bytes": sweep_bytes, "sweep_effective_GBs": sweep_bytes * steps / r["elapsed"] / 1e9,
           "roofline": {"bound": "hbm", "kernel": "k_bulk_w<1> (non-temporal loads)", "achieved": ach, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "launch_ms": per_ms, "launch_bytes": alg,
                        "sampled_launches": int(r["bulk"]["w_n"]),
                        "adj_kernel": {"launch_ms": r["bulk"]["adj_ms"] / max(r["bulk"]["adj_n"], 1),
                                       "launch_bytes": r["bulk"]["adj_bytes"] / max(r["bulk"]["adj_n"], 1)}}}
    del r, data
    return out


def extra_ais(abi, device, N=8192, R=3, B=100, steps_per_B=2):
    """BASELINE.json configs[4], one of its chains: AIS evidence of the SBM (R=3, b0=b1=1, alpha0=10) on a planted-SBM graph
    (examples/demo_ais.py:10-62,70-80): B temperatures, steps_per_B tempered sweeps each plus one log-likelihood
    evaluation per temperature, all through the host classes."""
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd import ais as AIS
    truth = planted_truth(N, R, seed=99)
    data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"],
                                        truth["mu_self"], truth["sigma_self"], seed=99, device=device)
    np.random.seed(5)
    sbm = SBMAdjacencyDistribution(N, C=R, pi=10.0, tau0=1.0, tau1=1.0, device=device, seed=20)
    AIS.estimate_marginal_likelihood(sbm, data, N_samples=1, B=6, steps_per_B=steps_per_B)      # warm-up
    t0 = time.perf_counter()
    logZ = AIS.estimate_marginal_likelihood(sbm, data, N_samples=1, B=B, steps_per_B=steps_per_B)
    el = time.perf_counter() - t0
    nsweeps = (B - 2) * steps_per_B
    return {"workload": "AIS, SBM adjacency R=%d, N=%d, one chain: %d temperatures x %d tempered sweeps + 1 log-likelihood each "
                        "(BASELINE.json configs[4]; its 8 chains are independent, one per GPU)" % (R, N, B, steps_per_B),
            "value": nsweeps / el, "unit": "sweeps/s (including the per-temperature log-likelihood)", "seconds_per_chain": el,
            "tempered_sweeps": nsweeps, "log_likelihood_evaluations": B - 1, "log_Z_one_chain": float(logZ),
            "sweep_algorithmic_bytes": 2.0 * N * N / 8.0, "log_likelihood_algorithmic_bytes": 1.0 * N * N / 8.0}


def extra_ais_eigen(abi, device, N=8192, D=8, B=12, steps_per_B=2):
    """BASELINE.json configs[4], the eigenmodel's chain: AIS with the tempered logistic eigenmodel (D = 8) at N = 8192 through
    the host class (examples/demo_ais.py:10-62; tempering build-defined: omega ~ PG(beta, psi)).  Few
    temperatures: a tempered sweep draws 67 M PG(beta, psi) variates by the 16-term gamma series (spec/gg_pg.h)."""
    from graphistician_amd.eigenmodel import LogisticEigenmodel
    from graphistician_amd import ais as AIS
    rng = np.random.RandomState(4321)
    Ft, lt, mu_t = rng.randn(N, D) * 0.6, rng.randn(D), -1.3
    A = np.empty((N, N), dtype=np.uint8)
    for i0 in range(0, N, 1024):
        Mu = mu_t + (Ft[i0:i0 + 1024] * lt).dot(Ft.T)
        A[i0:i0 + 1024] = rng.rand(*Mu.shape) < 1.0 / (1.0 + np.exp(-Mu))
    data = abi.DeviceData(A, None, device=device)
    del A
    np.random.seed(5)
    m = LogisticEigenmodel(N, D, p=0.1, device=device, seed=21)
    AIS.estimate_marginal_likelihood(m, data, N_samples=1, B=3, steps_per_B=1)      # warm-up
    t0 = time.perf_counter()
    logZ = AIS.estimate_marginal_likelihood(m, data, N_samples=1, B=B, steps_per_B=steps_per_B)
    el = time.perf_counter() - t0
    nsweeps = (B - 2) * steps_per_B
    t0 = time.perf_counter()
    for _ in range(3):
        m.resample(data)
    m.mu_0
    plain = (time.perf_counter() - t0) / 3
    return {"workload": "AIS, logistic eigenmodel D=%d, N=%d, one chain: %d temperatures x %d tempered sweeps + 1 log-likelihood each "
                        "(BASELINE.json configs[4])" % (D, N, B, steps_per_B),
            "value": nsweeps / el, "unit": "sweeps/s (including the per-temperature log-likelihood)", "seconds_per_chain": el,
            "tempered_sweeps": nsweeps, "log_likelihood_evaluations": B - 1, "log_Z_one_chain": float(logZ),
            "ms_per_plain_sweep_beta_1": plain * 1e3,
            "bound": "the PG(beta, psi) draws of a tempered sweep (16 gamma variates per pair; 64 until the end of round 4); at beta = 1 the exact PG(1, psi) sampler "
                     "and the sequential feature scan (DESIGN.md 7)"}


def extra_ais_latent(abi, device, N=8192, dim=2, B=30, steps_per_B=2):
    """BASELINE.json configs[4], the latent-distance model's chain: AIS with tempered HMC resamples (dim = 2) at N = 8192 through
    the host class (examples/demo_ais.py:10-62; adjacency.py:360-421)."""
    from graphistician_amd.adjacency import LatentDistanceAdjacencyDistribution
    from graphistician_amd import ais as AIS
    rng = np.random.RandomState(7)
    Lt = rng.randn(N, dim) * 1.3
    A = np.empty((N, N), dtype=np.uint8)
    for i0 in range(0, N, 1024):
        X = -((Lt[i0:i0 + 1024, None, :] - Lt[None, :, :]) ** 2).sum(2) + 1.0
        A[i0:i0 + 1024] = rng.rand(*X.shape) < 1.0 / (1.0 + np.exp(-X))
    data = abi.DeviceData(A, None, device=device)
    del A
    np.random.seed(6)
    m = LatentDistanceAdjacencyDistribution(N, dim=dim, sigma=2.0, device=device, seed=22)
    AIS.estimate_marginal_likelihood(m, data, N_samples=1, B=3, steps_per_B=1)      # warm-up
    t0 = time.perf_counter()
    logZ = AIS.estimate_marginal_likelihood(m, data, N_samples=1, B=B, steps_per_B=steps_per_B)
    el = time.perf_counter() - t0
    nres = (B - 2) * steps_per_B
    passes = 22
    res_bytes = (11 * 2.0 + 11 * 1.0) * N * N
    return {"workload": "AIS, latent-distance adjacency model dim=%d (HMC), N=%d, one chain: %d temperatures x %d tempered resamples + 1 "
                        "log-likelihood each (BASELINE.json configs[4])" % (dim, N, B, steps_per_B),
            "value": nres / el, "unit": "resamples/s (including the per-temperature log-likelihood)", "seconds_per_chain": el,
            "tempered_resamples": nres, "log_likelihood_evaluations": B - 1, "log_Z_one_chain": float(logZ),
            "resample_algorithmic_bytes": res_bytes, "effective_GBs": res_bytes * nres / el / 1e9,
            "bound": "%d passes per resample, each N^2 fp64 sigmoids (exp), softplus (log) at the trajectory ends (DESIGN.md 7d)" % passes}


def extra_host_classes(abi, device, N=4096, K=16, iters=40):
    """the reference's OWN loop on PLAIN NDARRAYS through the host classes -
    `model.resample((A, W)); lps.append(model.log_probability((A, W)))` (examples/demo_model_comparison.py:69-72, networks.py:28,69)
    - at N=4096, K=16.  The engine keeps a device copy per (A, W) pair; a writeable array may have been edited in place since the
    last call, so its whole buffer is hashed on EVERY call (exactness: tests/test_abi_and_host.py::test_one_flipped_entry_is_new_data);
    an array that cannot be written (`A.setflags(write=False)` and no writeable base) or was declared immutable is found by
    identity alone."""
    from graphistician_amd.networks import FactorizedNetworkDistribution
    from graphistician_amd.adjacency import SBMAdjacencyDistribution
    from graphistician_amd.weights import SBMGaussianWeightDistribution
    from graphistician_amd import engine as E
    truth = planted_truth(N, K, seed=1234)
    dd = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"],
                                      truth["mu_self"], truth["sigma_self"], seed=1234, device=device)
    A, W = dd.download()
    del dd
    np.random.seed(11)
    model = FactorizedNetworkDistribution(N, SBMAdjacencyDistribution, dict(C=K, device=device, seed=31),
                                          SBMGaussianWeightDistribution, dict(B=1, C=K, device=device, seed=32))
    out = {}

    def loop(data, n):
        lps = []
        t0 = time.perf_counter()
        for _ in range(n):
            model.resample(data)
            lps.append(model.log_probability(data))
        return (time.perf_counter() - t0) / n, lps

    E.clear_data_cache()
    for _ in range(60):                      # out of the prior draw's transient (the chain must be the same in all variants' timing)
        model.resample((A, W))
    t_w, _ = loop((A, W), max(5, iters // 4))
    Ar, Wr = A.copy(), W.copy()
    Ar.setflags(write=False); Wr.setflags(write=False)
    loop((Ar, Wr), 5)
    t_r, lps = loop((Ar, Wr), iters)
    t0 = time.perf_counter()
    for _ in range(iters):
        model.resample((Ar, Wr))
    t_res = (time.perf_counter() - t0) / iters
    t0 = time.perf_counter()
    for _ in range(iters):
        model.log_prior()
    t_prior = (time.perf_counter() - t0) / iters
    t0 = time.perf_counter()
    for _ in range(iters):
        model.adjacency._log_prior_host() + model.weights._log_prior_host()
    t_prior_host = (time.perf_counter() - t0) / iters
    E.clear_data_cache()
    E.declare_immutable(A, W)
    loop((A, W), 5)
    t_i, _ = loop((A, W), iters)
    E.clear_data_cache()
    return {"workload": "FactorizedNetworkDistribution(SBMAdjacencyDistribution, SBMGaussianWeightDistribution).resample((A, W)); "
                        "log_probability((A, W)) on numpy arrays, N=%d, K=%d (examples/demo_model_comparison.py:69-72)" % (N, K),
            "us_per_iteration_read_only_arrays": t_r * 1e6, "us_per_iteration_declare_immutable": t_i * 1e6,
            "us_per_iteration_writeable_arrays": t_w * 1e6,
            "us_per_resample_read_only_arrays": t_res * 1e6, "us_per_log_prior": t_prior * 1e6,
            "us_per_log_prior_host_formula": t_prior_host * 1e6,
            "last_log_probability": float(lps[-1]),
            "note": "writeable arrays: the buffers of A (uint8, %.0f MB) and W (float64, %.0f MB) are hashed on every call (gg_hash64, "
                    "all host threads) so that an in-place edit is new data; read-only / declared arrays: identity only"
                    % (A.nbytes / 1e6, W.nbytes / 1e6)}


def extra_wide(abi, device, N=4096, K=16, burnin=30, warmup=5, steps=20, repeats=5):
    """the configs[1] network with REFERENCE-PRECISION weight storage (two 24-bit limbs, sW = 8 bytes per stored
    weight): the weight model with wide_weights = 1 on gg_data_create_wide data, the adjacency model on the same handle.  The
    same planted network as the headline (downloaded and uploaded again in the wide form), chain seed 1."""
    truth = planted_truth(N, K, seed=1234)
    dd = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"],
                                      truth["mu_self"], truth["sigma_self"], seed=1234, device=device)
    A, W = dd.download()
    del dd
    data = abi.DeviceData(A, W, device=device, wide=True)
    del A, W
    adj, _w = build_models(abi, N, K, chain_seed=1, device=device)
    del _w
    rng = np.random.RandomState(1)
    rng.dirichlet(np.ones(K)); rng.choice(K, size=N); rng.beta(1.0, 1.0, size=(K, K)); rng.beta(1.0, 1.0)      # (build_models' stream up to the weight model)
    w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=4, special_case_self_conns=1, device=device, wide_weights=1)
    pi_w = np.ones(K) * 10.0
    m_w = rng.dirichlet(pi_w)
    w.set(abi.ST_PI, pi_w); w.set(abi.ST_M, m_w); w.set(abi.ST_C, rng.choice(K, p=m_w, size=N).astype(np.int32), np.int32)
    w.set(abi.ST_MU, rng.randn(K, K, 1)); w.set(abi.ST_SIGMA, np.ones((K, K, 1, 1)))
    niw = np.array([0.0, 1.0, 3.0, 1.0])
    w.set(abi.ST_NIW, niw); w.set(abi.ST_NIW_SELF, niw)
    abi.resample_many([adj, w], data, burnin)
    abi.resample_many([adj, w], data, warmup)
    chg0 = adj.stats()["label_changes"] + w.stats()["label_changes"]
    blocks = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        abi.resample_many([adj, w], data, steps)
        blocks.append(time.perf_counter() - t0)
    el = float(np.median(blocks))
    chg = adj.stats()["label_changes"] + w.stats()["label_changes"] - chg0
    ms, nb = w.bench_bulk(data, 20)
    sweep_bytes = 2.0 * N * N * (0.125 + 8.0)
    return {"workload": "Gaussian SBM Gibbs, N=%d, K=%d, B=1, WIDE weight storage (two 24-bit limbs per weight, 48 bits under the one scale)" % (N, K),
            "value": steps / el, "unit": "sweeps/s", "ms_per_step": el / steps * 1e3, "steps": steps, "timed_blocks": repeats,
            "label_changes_per_sweep": chg / float(steps * repeats),
            "sweep_algorithmic_bytes": sweep_bytes, "sweep_effective_GBs": sweep_bytes * steps / el / 1e9,
            "weight_pass_standalone_GBs": nb / ms / 1e6,
            "note": "the integer statistics of the two limbs are those of a B = 2 weight (k_bulk_w<2>, the B = 2 scans); the parameter step "
                    "collapses them to (n, sum w, sum w^2), the coefficient tables spread (g, h, Q) over them (spec/gg_lp.h)"}


def bench_eigen(args, abi, device, rank, world, barrier, max_over_ranks):
    """BASELINE.json configs[2]: logistic eigenmodel Gibbs, N=4096, D=8 (eigenmodel.py:836-857)."""
    N, D = args.nodes, args.latent
    rng = np.random.RandomState(1234)
    F = rng.randn(N, D); lam = rng.randn(D); mu_mu0 = -1.2815515655446004     # Phi^{-1}(0.1), eigenmodel.py:56
    mu0 = mu_mu0 + rng.randn()
    Mu = mu0 + (F * lam).dot(F.T)
    np.fill_diagonal(Mu, mu0)
    probit = args.config == "probit"
    if probit:
        from scipy.special import ndtr
        A = (rng.rand(N, N) < ndtr(Mu)).astype(np.uint8)
    else:
        A = (rng.rand(N, N) < 1.0 / (1.0 + np.exp(-Mu))).astype(np.uint8)
    data = abi.DeviceData(A, None, device=device)
    m = abi.Model(kind=abi.PROBIT_EIG if probit else abi.LOGISTIC_EIG, N=N, D=D, K=1, B=1, seed=1 + rank, sigma_mu0=1.0, sigma_F=1.0, sigma_lmbda=1.0,
                  mu_mu_0=mu_mu0, device=device)
    crng = np.random.RandomState(1 + rank)
    F0, l0, m0 = crng.randn(N, D), crng.randn(D), mu_mu0 + crng.randn()
    m.set(abi.ST_F, F0); m.set(abi.ST_LMBDA, l0); m.set(abi.ST_MU0, np.array([m0])); m.set(abi.ST_MU_LMBDA, np.zeros(D))
    m.resample(data, args.warmup)
    barrier()
    t0 = time.perf_counter()
    m.resample(data, args.steps)
    barrier()
    el = max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return
    sweep_bytes = float(N) * N * (3 * 8 + 2 * 2)      # write S once, read it twice; A and A^T read in both passes
    out = {"metric": "full Gibbs sweeps/sec on N-node graph; achieved HBM GB/s vs peak", "value": world * args.steps / el,
           "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": ("Probit eigenmodel Gibbs (truncated-normal auxiliaries), N=%d, D=%d (SURVEY.md 8f rank 2)" % (N, D)) if probit else
                                  ("Logistic eigenmodel Gibbs (Polya-gamma), N=%d, D=%d (BASELINE.json configs[2])" % (N, D)),
                      "sweep_algorithmic_bytes": sweep_bytes},
           "roofline": {"bound": "hbm", "achieved": sweep_bytes * args.steps / el / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": sweep_bytes * args.steps / el / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "whole-sweep effective rate; the sweep is bound by the sequential latent-feature scan "
                                "(N dependent D x D solves) and the PG(1,z) draws, not by HBM (DESIGN.md section 7)"}}
    if not args.no_cpu and world == 1:
        from oracle import oracle as O
        Fo, lo = F0.copy(), l0.copy()
        t0 = time.perf_counter()
        (O.probit_sweep if probit else O.eig_sweep)(A, Fo, lo, m0, np.zeros(D), mu_mu0, 1.0, 1.0, 1.0, 0, 1, 0)
        el1 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / el1, "unit": "sweeps/s", "cores": 1, "kind": "port",
                               "sample": "1 full sweep of the same N=%d D=%d network, oracle/ C++ -O2, 1 thread" % (N, D)}
    _emit(json.dumps(out))


def bench_latent(args, abi, device, rank, world, barrier, max_over_ranks):
    """Latent-distance adjacency model (adjacency.py:191-437): one step = resample = HMC on L
    (11 passes over A and A^T), HMC on mu_0 (11 passes), HMC on mu_self (diagonal only)."""
    N, dim = args.nodes, 2
    rng = np.random.RandomState(1234)
    Lt = rng.randn(N, dim) * np.sqrt(2.0)
    X = -((Lt[:, None, :] - Lt[None, :, :]) ** 2).sum(2) + 0.5
    A = (rng.rand(N, N) < 1.0 / (1.0 + np.exp(-X))).astype(np.uint8)
    del X
    data = abi.DeviceData(A, None, device=device)
    m = abi.Model(kind=abi.LATENT_DIST, N=N, D=dim, K=1, B=1, seed=1 + rank, sigma_F=2.0, device=device)
    crng = np.random.RandomState(1 + rank)
    L0 = crng.randn(N, dim) * np.sqrt(2.0)
    m.set(abi.ST_F, L0); m.set(abi.ST_MU0, np.array([0.0])); m.set(abi.ST_LD_AUX, np.array([0.0, 0.01, 0.9]))
    m.resample(data, args.warmup)
    barrier()
    t0 = time.perf_counter()
    m.resample(data, args.steps)
    barrier()
    el = max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return
    passes = 22
    sweep_bytes = (11 * 2.0 + 11 * 1.0) * N * N      # the L update reads A and A^T, the mu_0 update only A
    out = {"metric": "full Gibbs sweeps/sec on N-node graph; achieved HBM GB/s vs peak", "value": world * args.steps / el,
           "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "Latent-distance adjacency model, HMC (10 leapfrog steps), N=%d, dim=%d (SURVEY.md 8f rank 3)" % (N, dim),
                      "sweep_algorithmic_bytes": sweep_bytes, "accepted_L_proposals": m.stats()["label_changes"]},
           "roofline": {"bound": "hbm", "kernel": "k_ld_pass", "achieved": sweep_bytes * args.steps / el / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": sweep_bytes * args.steps / el / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "whole-resample effective rate over %d passes (11 of A and A^T, 11 of A; uint8); each pass also evaluates N^2 "
                                "fp64 exp (sigmoid) and, at the two ends of a trajectory, log (softplus), which is what bounds it "
                                "at this size" % passes}}
    if not args.no_cpu and world == 1:
        from oracle import oracle as O
        Lo = L0.copy()
        t0 = time.perf_counter()
        O.ld_resample(A, Lo, 0.0, 0.0, 2.0, 0.01, 0.9, 1, 0)
        el1 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / el1, "unit": "sweeps/s", "cores": 1, "kind": "port",
                               "sample": "1 resample of the same N=%d network, oracle/ C++ -O2, 1 thread" % N}
    _emit(json.dumps(out))


def bench_latent_w(args, abi, device, rank, world, barrier, max_over_ranks):
    """Latent-distance Gaussian weight model (weights.py:613-903; SURVEY.md 8f rank 3, weight side): one step = resample =
    HMC on L (11 passes over W and W^T), then one more pass for the conjugate draws of b, sigma, self Gaussian, eta."""
    N, dim = args.nodes, 2
    rng = np.random.RandomState(1234)
    Lt = rng.randn(N, dim) * 0.6
    A = (rng.rand(N, N) < 0.25).astype(np.uint8)
    W = (-((Lt[:, None, :] - Lt[None, :, :]) ** 2).sum(2) + 0.5 + 0.3 * rng.randn(N, N))[:, :, None]
    data = abi.DeviceData(A, W, device=device)
    m = abi.Model(kind=abi.LATENT_DIST_W, N=N, D=dim, K=1, B=1, seed=1 + rank, device=device)
    crng = np.random.RandomState(1 + rank)
    L0 = crng.randn(N, dim) * 0.1
    m.set(abi.ST_F, L0); m.set(abi.ST_MU0, np.array([0.5])); m.set(abi.ST_LD_AUX, np.array([0.01, 1.0, 0.0, 1.0]))
    m.resample(data, args.warmup)
    barrier()
    t0 = time.perf_counter()
    m.resample(data, args.steps)
    barrier()
    el = max_over_ranks(time.perf_counter() - t0)
    if rank != 0:
        return
    passes = 12
    sweep_bytes = passes * 8.0 * N * N
    out = {"metric": "full Gibbs sweeps/sec on N-node graph; achieved HBM GB/s vs peak", "value": world * args.steps / el,
           "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "Latent-distance Gaussian weight model, HMC (10 leapfrog steps) + conjugate draws, N=%d, dim=%d, "
                                  "edge density 0.25 (SURVEY.md 8f rank 3)" % (N, dim),
                      "sweep_algorithmic_bytes": sweep_bytes, "accepted_L_proposals": m.stats()["label_changes"],
                      "last_resample_device_ms": m.stats()["total_ms"]},
           "roofline": {"bound": "hbm", "kernel": "k_ldw_pass", "achieved": sweep_bytes * args.steps / el / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": sweep_bytes * args.steps / el / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "whole-resample effective rate over %d passes of W and W^T (int32 fixed point, 8 N^2 bytes a pass)" % passes}}
    if not args.no_cpu and world == 1:
        from oracle import oracle as O
        q, shift = O.quantise(A, W)
        Lo = L0.copy()
        t0 = time.perf_counter()
        O.ldw_resample(A, q, shift, Lo, 0.5, 0.01, 1.0, 0.0, 1.0, [1.0, 3.0], [0.0, 1.0, 3.0, 1.0], 1, 0)
        el1 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / el1, "unit": "sweeps/s", "cores": 1, "kind": "port",
                               "sample": "1 resample of the same N=%d network, oracle/ C++ -O2, 1 thread" % N}
    _emit(json.dumps(out))


def bench_sharded(args, abi, device, rank, world, barrier, max_over_ranks):
    line = run_sharded(args, abi, device, rank, world, barrier, max_over_ranks, args.nodes, args.blocks)
    if rank == 0:
        _emit(json.dumps(line))


def run_sharded(args, abi, device, rank, world, barrier, max_over_ranks, N, K, all_ok=None, steps=None, warmup=None):
    """One Gaussian-SBM chain sharded over the ranks (strong scaling; BASELINE.json configs[3] at N=65536, K=64):
    exact - the same chain for any number of ranks (graphistician_amd/sharded.py).  Returns the result line (rank 0).
    all_ok(flag) -> bool: agreement of all ranks: set-up (communicator, the rank's shard of the network, the
    models) may fail on ONE rank - out of memory beside the replica's network, an RCCL initialisation error - and the
    others would then wait in the library's collectives for ever; so every rank reports how its set-up went and the
    chain runs only if all of them are ready."""
    from graphistician_amd import sharded as S
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    err = None
    try:
        # collectives: the library's own RCCL calls (one process per GPU); GG_BENCH_BACKEND=gloo (CPU-side tests, several ranks on
        # one GPU) routes them through host callbacks instead.  torch.distributed is the launcher / rendezvous only.
        # (RCCL greets with a version banner on stdout when its first communicator comes up: that goes to stderr here, stdout
        # carries the one JSON line)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if world > 1:
                import torch.distributed as tdist
                comm = S.rccl_comm(tdist, device) if os.environ.get("GG_BENCH_BACKEND", "nccl") == "nccl" else S.host_comm(tdist, device)
            else:
                comm = S.local_comm(device)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        # ownership: blocks of 256 nodes dealt out round-robin (every band of the scan spreads over all ranks);
        # GG_BENCH_SHARD_RANGE=1: contiguous node ranges, the round-3 form
        truth = planted_truth(N, K, seed=1234)
        if os.environ.get("GG_BENCH_SHARD_RANGE"):
            n0, n1 = S.shard_range(N, rank, world)
            own, own_txt = dict(shard_begin=n0, shard_end=n1), "contiguous node ranges"
            data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"],
                                                truth["mu_self"], truth["sigma_self"], seed=1234, device=device, shard=(n0, n1))
        else:
            o = S.Ownership(N, rank, world)
            own, own_txt = o.config, "blocks of %d nodes dealt out round-robin" % o.block
            data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"],
                                                truth["mu_self"], truth["sigma_self"], seed=1234, device=device, blocks=o.blocks)
        rng = np.random.RandomState(1)                      # the same chain on every rank
        adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=3, tau1=1.0, tau0=1.0, allow_self_connections=1,
                        special_case_self_conns=1, device=device, **own)
        m_a = rng.dirichlet(np.ones(K))
        adj.set(abi.ST_PI, np.ones(K)); adj.set(abi.ST_M, m_a)
        adj.set(abi.ST_C, rng.choice(K, p=m_a, size=N).astype(np.int32), np.int32)
        adj.set(abi.ST_P, rng.beta(1.0, 1.0, size=(K, K))); adj.set(abi.ST_P_SELF, np.array([rng.beta(1.0, 1.0)]))
        w = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=1, seed=4, special_case_self_conns=1, device=device, **own)
        m_w = rng.dirichlet(np.ones(K) * 10.0)
        w.set(abi.ST_PI, np.ones(K) * 10.0); w.set(abi.ST_M, m_w)
        w.set(abi.ST_C, rng.choice(K, p=m_w, size=N).astype(np.int32), np.int32)
        w.set(abi.ST_MU, rng.randn(K, K, 1)); w.set(abi.ST_SIGMA, np.ones((K, K, 1, 1)))
        niw = np.array([0.0, 1.0, 3.0, 1.0])
        w.set(abi.ST_NIW, niw); w.set(abi.ST_NIW_SELF, niw)
    except Exception as exc:
        err = "rank %d: set-up failed: %s" % (rank, exc)
    if all_ok is not None and not all_ok(err is None):
        return {"error": err or "set-up failed on another rank: the sharded chain was skipped on every rank"} if rank == 0 else None
    if err is not None:
        raise RuntimeError(err)
    for _ in range(warmup):
        S.half_sweep(adj, data, comm); S.half_sweep(w, data, comm)
    chg0 = adj.stats()["label_changes"] + w.stats()["label_changes"]
    rnd0 = adj.stats()["resolve_rounds"] + w.stats()["resolve_rounds"]
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        S.half_sweep(adj, data, comm); S.half_sweep(w, data, comm)
    barrier()
    el = max_over_ranks(time.perf_counter() - t0)
    chg = adj.stats()["label_changes"] + w.stats()["label_changes"] - chg0
    rnd = adj.stats()["resolve_rounds"] + w.stats()["resolve_rounds"] - rnd0
    digest = int(np.bitwise_xor.reduce(adj.get(abi.ST_C, (N,), np.int32) * np.arange(1, N + 1, dtype=np.int32)))
    cinfo = comm.info()
    del adj, w, data
    if rank == 0:
        sweep_bytes = 2.0 * N * N * (0.125 + 4.0)
        return ({
            "metric": "full Gibbs sweeps/sec on N-node graph; achieved HBM GB/s vs peak", "value": steps / el,
            "unit": "sweeps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int64 statistics / f64 conditionals (bit-packed adjacency, 24-bit fixed-point weights in int32)",
            "data": "synthetic",
            "config": {"workload": "ONE Gaussian SBM chain, N=%d, K=%d, B=1, node-sharded over %d rank(s)" % (N, K, world),
                       "parallelism": "exact node sharding (%s): all-reduce of K x K integer tables + one all-gather of surprises per scan "
                                      "round (%s, %d ranks seen by the communicator)" % (own_txt, "RCCL, device buffers on the engine stream"
                                                                                         if cinfo["rccl"] else "host callbacks (gloo)", cinfo["world"]),
                       "ranks": int(cinfo["world"]),
                       "collectives_on_the_wire_in_run": int(cinfo["collectives"]), "scan_rounds_in_run": int(cinfo["rounds"]),
                       "note_one_rank": None if world > 1 else "one rank: the all-reduce and the all-gathers are bypassed, nothing travels",
                       "us_per_label_change": (el / max(chg, 1)) * 1e6 if chg else None,
                       "scan_rounds_in_timed_region": int(rnd), "scan_rounds_per_sweep": rnd / float(steps),
                       "us_per_scan_round": (el / max(rnd, 1)) * 1e6 if rnd else None,
                       "label_changes_in_timed_region": int(chg), "labels_digest": digest,
                       "sweep_algorithmic_bytes": sweep_bytes, "sweep_effective_GBs": sweep_bytes * steps / el / 1e9},
            "roofline": {"bound": "hbm", "achieved": sweep_bytes * steps / el / 1e9, "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": sweep_bytes * steps / el / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                         "note": "whole-sweep effective rate over the aggregate HBM peak of the ranks"}})
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shard", action="store_true",
                    help="one chain sharded over the ranks (strong scaling) instead of an independent chain per rank")
    ap.add_argument("--config", default="sbm", choices=["sbm", "eigen", "probit", "latent", "latentw"],
                    help="sbm = BASELINE.json configs[1] (default, the headline); eigen = configs[2]")
    ap.add_argument("--latent", type=int, default=8)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes", type=int, default=None, help="default: 4096 (configs[1]); with --gpus G > 1 the legs of configs[3] run at 65536")
    ap.add_argument("--blocks", type=int, default=None, help="default: 16 below 32768 nodes, 64 from there")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--burnin", type=int, default=None,
                    help="sweeps run as part of chain set-up, before the W warm-up steps: the chain starts from a prior draw "
                         "and relabels thousands of nodes in its first few sweeps; the metric is the sampler's steady state")
    ap.add_argument("--repeats", type=int, default=7,
                    help="timed blocks of `--steps` sweeps per chain; the median block is the chain's time (min / max reported)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs of the default run (mixing chains, configs[2], configs[3] on one GPU, one AIS chain)")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise the multi-process protocol (rendezvous, barrier, max-over-ranks) without the engine")
    args = ap.parse_args()
    _claim_stdout()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("GG_BENCH_BACKEND", "nccl")   # "gloo" for the CPU test of the N>1 path
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.dry_run:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.02 * (1 + rank))           # rank r "works" 20(r+1) ms: the slowest rank sets the time
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            _emit(json.dumps({"metric": "dry-run", "value": world * args.steps / el, "unit": "sweeps/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
                              "elapsed_s": el, "scaling": "weak"}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from graphistician_amd import _abi as abi
    # The headline workload is the SAME for every G - configs[1], the six-chain mix, one replica per GPU - so that the per-G
    # values of the driver's 1 / 2 / 4 / 8 run are values of one workload (it computes the scaling efficiency from them).
    # BASELINE.json's N=65536, K=64 at G > 1 comes in the same line as two legs: one replica of that chain per GPU
    # (other_configs.sbm_n65536_k64_replicas) and ONE chain sharded over the G ranks (sharded_chain); at G = 1 it is
    # other_configs.sbm_n65536_k64_one_gpu.  An explicit --nodes runs everything at that size (tests).
    explicit_size = args.nodes is not None
    if args.nodes is None:
        args.nodes = 4096
    if args.blocks is None:
        args.blocks = 16 if args.nodes < 32768 else 64
    if args.burnin is None:
        args.burnin = 30 if args.nodes < 32768 else 12
    N, K = args.nodes, args.blocks
    device = local_rank
    if os.environ.get("GG_BENCH_SAME_DEVICE"):      # several ranks on one GPU (tests on a 1-GPU box)
        device = 0
    if args.shard:
        bench_sharded(args, abi, device, rank, world, barrier, max_over_ranks)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.config in ("eigen", "probit", "latent", "latentw"):
        fn = {"latent": bench_latent, "latentw": bench_latent_w}.get(args.config, bench_eigen)
        fn(args, abi, device, rank, world, barrier, max_over_ranks)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    truth = planted_truth(N, K, seed=1234)
    data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"],
                                        truth["mu_self"], truth["sigma_self"], seed=1234, device=device)
    # How many labels a chain keeps changing per sweep - and so how much of the scan's sequential machinery runs in the timed
    # region - depends on its prior draw: of six chain seeds at N=4096 two settle into zero changes, two change a few labels now
    # and then, two keep mixing.  The headline is therefore the MIX: every seed runs its own burn-in,
    # warm-up and K timed sweeps (each timed segment bracketed by the barrier), value = all timed sweeps / all timed seconds.
    # Every rank runs the same seeds (weak scaling needs the same work on every GPU).  GG_BENCH_CHAIN_SEED=s: that chain only.
    if os.environ.get("GG_BENCH_CHAIN_SEED"):
        seeds = [int(os.environ["GG_BENCH_CHAIN_SEED"])]
    elif os.environ.get("GG_BENCH_CHAIN_SEEDS"):
        seeds = [int(x) for x in os.environ["GG_BENCH_CHAIN_SEEDS"].split(",")]
    else:
        seeds = [1, 2, 3, 4, 5, 6] if N < 32768 else [1]
    # HIP events (on the engine streams) around the launches that stream the data: about 18 sampled launches over the whole
    # mix whatever the run length at small N - 3 per chain of a six-chain mix, 8 for a single chain (each record costs the
    # stream a barrier packet that drains the queue: the instrumentation, not the engine - with 7 of a chain's 20 sweeps
    # bracketed the quiet chains lost 4-9 us per sweep to it), every launch for N > 8192
    per_chain = 8 if len(seeds) < 4 else 3
    every = 1 if N > 8192 else max(1, -(-args.steps // per_chain))
    chains, total_el = [], 0.0
    bulk = dict(w_ms=0.0, w_n=0, w_bytes=0.0, adj_ms=0.0, adj_n=0, adj_bytes=0.0)
    forms = dict(sweeps_fused=0, sweeps_two_launch=0, sweeps_serial=0)
    chg = 0
    r = None
    # a chain's `steps` timed sweeps are run as R blocks of exactly `steps` sweeps (after one untimed block), each
    # bracketed by barrier + synchronisation with the MAX over ranks taken per block; the chain's time is the MEDIAN block.
    # `value` = (chains x steps) / (sum of the chains' median block times); min / max blocks are reported beside it.
    R = max(1, args.repeats)
    total_min = total_max = total_timed = 0.0
    for seed in seeds:
        del r
        r = run_chain(abi, data, N, K, seed, device, args.burnin, args.warmup, args.steps, every, barrier,
                      stepwise=bool(os.environ.get("GG_BENCH_STEPWISE")), repeats=R, max_over_ranks=max_over_ranks)
        el_s = r["elapsed"]
        total_el += el_s
        total_min += min(r["blocks"]); total_max += max(r["blocks"]); total_timed += r["total_timed_s"]
        chg += r["changes"]
        for k in bulk:
            bulk[k] += r["bulk"][k]
        for k in forms:
            forms[k] += r["forms"][k]
        chains.append({"chain_seed": seed, "value": world * args.steps / el_s, "ms_per_step": el_s / args.steps * 1e3,
                       "ms_per_step_min_max_over_blocks": [min(r["blocks"]) / args.steps * 1e3, max(r["blocks"]) / args.steps * 1e3],
                       "label_changes_per_sweep": r["changes"] / float(args.steps),
                       "scan_rounds_per_sweep": r["rounds"] / float(args.steps), "weight_model_sweep_forms": r["forms"],
                       "adjacency_model_sweep_forms": r["forms_adj"],
                       "us_per_step_by_block": [round(b / args.steps * 1e6, 1) for b in r["blocks"]],
                       "bulk_launch_ms": r["bulk"]["w_ms"] / max(r["bulk"]["w_n"], 1),
                       "bulk_launch_GBs": r["bulk"]["w_bytes"] / (r["bulk"]["w_ms"] * 1e-3) / 1e9 if r["bulk"]["w_ms"] > 0 else None})
        if rank == 0:
            _progress("chain seed %d: %d sweeps in %.4f s (median of %d blocks, %.4f .. %.4f; %.1f label changes per sweep)" % (
                seed, args.steps, el_s, R, min(r["blocks"]), max(r["blocks"]), r["changes"] / float(args.steps)))
    adj, w = r["adj"], r["w"]
    el, nsw = total_el, len(seeds) * args.steps

    out = None
    if rank == 0:
        # Dominant kernel: the weight model's streaming pass over the N x N int32 weight matrix AND its stored transpose
        # (out- and in-statistics of every node): 2 * N^2 * 4 algorithmic bytes per sweep.  While a chain is quiet the pass of
        # sweep s+2 rides in the fused launch of sweep s (k_sweep_q), otherwise it is k_bulk_w on its own: the
        # events bracket exactly the launches that stream the data, achieved = the bytes they stream / their duration, over
        # the sampled launches of ALL chains of the mix.
        per_launch_ms = bulk["w_ms"] / max(bulk["w_n"], 1)
        alg_bytes = bulk["w_bytes"] / max(bulk["w_n"], 1)
        achieved = bulk["w_bytes"] / (bulk["w_ms"] * 1e-3) / 1e9 if bulk["w_ms"] > 0 else 0.0
        sweep_bytes = 2.0 * N * N * (0.125 + 4.0)
        pmc = pmc_traffic(N, K, forms)
        sa_ms, sa_b = adj.bench_bulk(data, 20)
        sw_ms, sw_b = w.bench_bulk(data, 20)
        quiet = [c for c in chains if c["label_changes_per_sweep"] == 0.0]
        out = {
            "metric": "full Gibbs sweeps/sec on N-node graph; achieved HBM GB/s vs peak",
            "value": world * nsw / el,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / nsw * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64 statistics / f64 conditionals (bit-packed adjacency, 24-bit fixed-point weights in int32)",
            "data": "synthetic",
            "config": {
                "workload": "Gaussian SBM Gibbs, N=%d, K=%d, B=1, dense adjacency (%s)" % (
                    N, K, "BASELINE.json configs[1]" if (N, K) == (4096, 16) else
                    ("BASELINE.json configs[3], one replica of the chain per GPU" if (N, K) == (65536, 64) else "same recipe, other size")),
                "parallelism": "one replica of the chain per GPU, no collective" if world > 1 else "single GPU",
                "chain_seeds": seeds, "timed_sweeps": nsw, "steps_per_chain": args.steps,
                "value_is": "timed sweeps of all chains / their timed seconds; each chain: burn-in, warm-up, one untimed block, then R "
                            "timed blocks of exactly `steps` sweeps (barrier + synchronisation on both sides, max over ranks per block) "
                            "of which the MEDIAN is the chain's time: value = chains x steps / sum of the medians",
                "timed_blocks_per_chain": R, "timed_sweeps_run": nsw * R, "total_timed_seconds": total_timed,
                "event_sampled_blocks": "every third block of a chain (blocks 0, 3, 6): HIP events around %d launches each" % per_chain,
                "value_min_max_over_blocks": [world * nsw / total_max, world * nsw / total_min],
                "burn_in_sweeps": args.burnin,
                "label_changes_in_timed_region": int(chg), "label_changes_per_sweep": chg / float(nsw),
                "labels_digest": int(np.bitwise_xor.reduce(adj.get(abi.ST_C, (N,), np.int32) * np.arange(1, N + 1, dtype=np.int32))),
                "labels_digest_of": "adjacency labels of chain seed %d after its last timed block" % seeds[-1],
                "chains": chains,
                "sweep_algorithmic_bytes": sweep_bytes,
                "sweep_effective_GBs": sweep_bytes * nsw / el / 1e9,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "the launches that stream W and W^T: k_sweep_q<3,false,0,0> (scan of sweep s | gated parameter step of sweep s+1 | "
                          "riders = the weight pass for sweep s+2) for %d of the weight model's timed sweeps, its two-launch form "
                          "(k_params + riders, k_spec_resolve + riders) for %d, k_bulk_w<1> on its own for %d (gg_stats)"
                          % (forms["sweeps_fused"], forms["sweeps_two_launch"], forms["sweeps_serial"]),
                "weight_model_sweep_forms": forms,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc.get("traffic"),
                "traffic_source": pmc.get("traffic_source"),
                "traffic_note": pmc.get("note"),
                "launch_ms": per_launch_ms,
                "launch_bytes": alg_bytes,
                "sampled_launches": int(bulk["w_n"]),
                "standalone_GBs": sw_b / sw_ms / 1e6,
                "note": "achieved = in situ (HIP events on the engine stream around the sampled launches inside the timed "
                        "regions, with the adjacency model's kernels running concurrently on another stream); "
                        "standalone_GBs = the plain bulk kernel k_bulk_w<1> launched back to back alone",
                "adj_kernel": {"kernel": "adjacency pass (k_bulk_adj, or its riders)", "launch_ms": bulk["adj_ms"] / max(bulk["adj_n"], 1),
                               "launch_bytes": bulk["adj_bytes"] / max(bulk["adj_n"], 1), "standalone_GBs": sa_b / sa_ms / 1e6},
            },
        }
        if quiet:
            out["quiet_chain"] = {"chain_seed": quiet[0]["chain_seed"], "value": quiet[0]["value"], "unit": "sweeps/s",
                                  "ms_per_step": quiet[0]["ms_per_step"],
                                  "note": "a chain of the mix that changed no label in its timed sweeps (the round-2 headline was this chain alone)"}
        # measured ceiling beside the nominal peak: a pure read stream of the same number of bytes as one
        # k_bulk_w launch (cache-resident at N=4096, like the kernel's own data) and of 4 GiB (HBM)
        if world == 1:
            # the log-likelihood sweep the reference's demos add after every resample, and the loop they run
            # (`model.resample(data); lps.append(model.log_probability(data))`, examples/demo_model_comparison.py:69-72):
            # one sweep per call, host clock, both halves of the network model
            _progress("log-likelihood sweeps / reference call pattern")
            try:
                nll = 50

                def ll_own_pass():
                    # (a label vector set again - to what it is - drops the statistics counted ahead: every evaluation then
                    # streams A and W once, the out-direction pass)
                    t = 0.0
                    ca, cw = adj.get(abi.ST_C, (N,), np.int32), w.get(abi.ST_C, (N,), np.int32)
                    for _ in range(10):
                        adj.set(abi.ST_C, ca, np.int32); w.set(abi.ST_C, cw, np.int32)
                        t0l = time.perf_counter()
                        abi.log_likelihood_many([adj, w], data)
                        t += time.perf_counter() - t0l
                    return t / 10

                def loop(with_ll):
                    for _ in range(5):
                        abi.resample_many([adj, w], data, 1)
                        if with_ll:
                            abi.log_likelihood_many([adj, w], data)
                    t0l = time.perf_counter()
                    for _ in range(nll):
                        abi.resample_many([adj, w], data, 1)
                        if with_ll:
                            abi.log_likelihood_many([adj, w], data)
                    return (time.perf_counter() - t0l) / nll

                ell_pass = ll_own_pass()
                t_call, t_iter = loop(False), loop(True)
                ll_bytes = float(N) * N * (0.125 + 4)
                out["log_likelihood_sweep"] = {
                    "ms_with_its_own_pass": ell_pass * 1e3, "GBs_with_its_own_pass": ll_bytes / ell_pass / 1e9, "algorithmic_bytes": ll_bytes,
                    "note": "log p(A | .) + log p(W | A, .) of the whole network (networks.py:28-31), both halves at once "
                            "(gg_log_likelihood_many), when the statistics of the current labels are not resident: one out-direction "
                            "pass over A and W, the K x K tables and the closed form on the device.  Right after a sweep of a quiet "
                            "chain it costs nothing: see reference_call_pattern.  Not part of `value`"}
                out["reference_call_pattern"] = {
                    "us_per_resample_call": t_call * 1e6, "us_per_iteration_resample_plus_log_probability": t_iter * 1e6,
                    "iterations_per_s": 1.0 / t_iter, "chain_seed": seeds[-1],
                    "note": "the reference's loop, one sweep per call (host clock, %d iterations, a chain of the mix as it stands after "
                            "its timed sweeps).  A quiet call is two launches per model - k_params with the whole next pass as "
                            "riders, then k_spec_resolve_epi: the scan, and gated on its verdict the pair terms of the "
                            "log-likelihood of the state the call leaves (from the statistics the riders just counted), resample_m "
                            "and the control block written to pinned memory, which the host watches instead of synchronising the "
                            "stream; the log_probability that follows is answered from that record.  Every call streams A and W "
                            "once.  Round 2: 94 / 130 us" % nll}
            except Exception as exc:
                out["log_likelihood_sweep"] = {"error": str(exc)}
        _progress("stream probe")
        try:
            out["roofline"]["measured_ceiling"] = {
                "read_GBs_same_bytes": abi.bench_stream(alg_bytes, 20, 0, device),
                "read_GBs_4GiB": abi.bench_stream(4 << 30, 5, 0, device),
                "copy_GBs_4GiB": abi.bench_stream(2 << 30, 5, 1, device),
                "note": "gg_bench_stream: 16-byte loads, grid-strided, nothing else; copy counts read + written bytes"}
        except Exception as exc:   # the probe must never cost the bench line
            out["roofline"]["measured_ceiling"] = {"error": str(exc)}
        if not args.no_cpu and world == 1:   # timed on rank 0 at N=1 only
            _progress("cpu baseline (bounded: about %.0f s)" % args.cpu_seconds)
            out["cpu_baseline"] = cpu_baseline(abi, data, N, K, args.cpu_seconds)
            out["parity_checked"] = bool(out["cpu_baseline"].get("parity", {}).get("bit_exact", False))
            _progress("literal baseline")
            out["cpu_baseline"]["literal_extrapolated"] = cpu_literal_extrapolated(N, K)
        else:
            out["parity_checked"] = False
        if world == 1 and not args.no_extras:
            # The other configurations of BASELINE.json and the chains that keep mixing, in the same driver-run line
            # (extra keys; `value` stays the configs[1] headline).  Each leg is bounded and must never cost the line.
            del adj, w, r, data
            legs = {}
            for name, fn in (("reference_call_pattern_host_classes", extra_host_classes), ("eigen_n4096_d8", extra_eigen),
                             ("ais_sbm_n8192", extra_ais), ("ais_eigen_n8192", extra_ais_eigen), ("ais_latent_n8192", extra_ais_latent),
                             ("sbm_n4096_k16_wide", extra_wide), ("sbm_n65536_k64_one_gpu", extra_big)):
                _progress("extra config: %s" % name)
                try:
                    legs[name] = fn(abi, device)
                except Exception as exc:
                    legs[name] = {"error": str(exc)}
            out["other_configs"] = legs
    if world > 1 and not os.environ.get("GG_BENCH_NO_SHARD_LEG"):
        # BASELINE.json configs[3] as written: ONE chain of the same network sharded over the G GPUs (every rank takes part).
        # A failing leg must not cost the replica line: (i) whether the leg runs is decided by ALL ranks - each
        # reports how its set-up went, one failure skips the leg everywhere; (ii) a rank that fails or hangs INSIDE the
        # chain leaves the others in the library's collectives, so the leg runs under a deadline: when it passes, rank 0
        # prints the line it has (the replica result, `sharded_chain` = the timeout) and every rank leaves the process.
        # GG_BENCH_NO_SHARD_LEG=1 switches the leg off, GG_BENCH_SHARD_LEG_SECONDS sets the deadline of the two configs[3] legs together (default 360).
        adj = w = r = data = None
        import gc
        import threading
        gc.collect()
        # (GG_BENCH_BIG_NODES / GG_BENCH_BIG_BLOCKS: the legs at another size - the two-ranks-on-one-GPU test)
        bigN, bigK = (N, K) if (explicit_size or args.config != "sbm") else (
            int(os.environ.get("GG_BENCH_BIG_NODES", "65536")), int(os.environ.get("GG_BENCH_BIG_BLOCKS", "64")))

        def all_ok(flag):
            import torch
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        deadline = float(os.environ.get("GG_BENCH_SHARD_LEG_SECONDS", "360"))   # both configs[3] legs
        done = threading.Event()

        def watchdog():
            if done.wait(deadline):
                return
            if rank == 0:
                out["sharded_chain"] = {"error": "the configs[3] legs did not finish within %.0f s (a rank failed or hung inside them); "
                                                 "the replica line above is complete" % deadline}
                _emit(json.dumps(out))
            sys.stderr.write("[bench] rank %d: sharded leg timed out, leaving\n" % rank)
            sys.stderr.flush()
            os._exit(0)

        wd = threading.Thread(target=watchdog, daemon=True)
        wd.start()
        if (bigN, bigK) != (N, K):
            # configs[3], one replica of the chain per GPU (weak scaling, no collective): the same leg as
            # other_configs.sbm_n65536_k64_one_gpu of the G = 1 line, every timed block bracketed by the barrier, max over ranks.
            # Collective discipline as for the sharded leg: all ranks agree that the network is resident before any of them
            # enters a barrier-bracketed block.
            if rank == 0:
                _progress("configs[3] replicas: N=%d, K=%d on each of %d ranks" % (bigN, bigK, world))
            bdata, err = None, None
            try:
                bt = planted_truth(bigN, bigK, seed=1234)
                bdata = abi.DeviceData.synthetic_sbm(bigN, bt["c"], bt["p"], bt["p_self"], bt["mu"], bt["sigma"],
                                                     bt["mu_self"], bt["sigma_self"], seed=1234, device=device)
            except Exception as exc:
                err = str(exc)
            if all_ok(err is None):
                bsteps = min(args.steps, 20)
                try:
                    br = run_chain(abi, bdata, bigN, bigK, 1, device, 12, min(args.warmup, 2), bsteps, 1, barrier,
                                   repeats=3, max_over_ranks=max_over_ranks)
                    bb = 2.0 * bigN * bigN * (0.125 + 4.0)
                    big = {"workload": "Gaussian SBM Gibbs, N=%d, K=%d, B=1 (BASELINE.json configs[3]), one replica of the chain "
                                       "per GPU, no collective" % (bigN, bigK),
                           "value": world * bsteps / br["elapsed"], "unit": "sweeps/s (all replicas)", "n_gpus": world,
                           "ms_per_step": br["elapsed"] / bsteps * 1e3, "steps": bsteps, "timed_blocks": 3,
                           "ms_per_step_min_max_over_blocks": [min(br["blocks"]) / bsteps * 1e3, max(br["blocks"]) / bsteps * 1e3],
                           "label_changes_per_sweep": br["changes"] / float(bsteps), "sweep_algorithmic_bytes": bb,
                           "sweep_effective_GBs_per_gpu": bb * bsteps / br["elapsed"] / 1e9,
                           "frac_of_hbm_peak_over_the_sweep": bb * bsteps / br["elapsed"] / 1e9 / HBM_PEAK_GBS,
                           "compare_with": "other_configs.sbm_n65536_k64_one_gpu of the --gpus 1 line"}
                    del br
                except Exception as exc:     # (a rank failing INSIDE the chain leaves the others at the barrier: the deadline)
                    big = {"error": str(exc)}
            else:
                big = {"skipped": "set-up failed on a rank" + ("" if err is None else ": " + err)}
            bdata = None
            gc.collect()
            if rank == 0:
                out.setdefault("other_configs", {})["sbm_n%d_k%d_replicas" % (bigN, bigK)] = big
        if rank == 0:
            _progress("sharded chain over %d ranks" % world)
        try:
            line = run_sharded(args, abi, device, rank, world, barrier, max_over_ranks, bigN, bigK, all_ok=all_ok,
                               steps=min(args.steps, 20), warmup=min(args.warmup, 3))
        except Exception as exc:
            line = {"error": str(exc)}
        done.set()
        if rank == 0:
            out["sharded_chain"] = line
    if rank == 0:
        _emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
