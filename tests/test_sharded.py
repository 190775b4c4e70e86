"""Exact sharded chain: 1, 2 and 3 ranks must all reproduce the oracle's single sequential chain bit
for bit.  One rank runs the library's RCCL communicator (ncclAllReduce / ncclAllGather on the engine stream); two and three
ranks are processes sharing the one GPU of the test box, so the same device-driven protocol runs over the host-callback
communicator (gloo) — RCCL refuses two ranks on one device."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (world, N, K, B, special, ownership form, forced band width or 0)
CASES = [(1, 150, 4, 1, 1, "blocks", 0), (2, 200, 5, 1, 1, "blocks", 0), (3, 330, 7, 1, 1, "blocks", 0), (2, 257, 33, 2, 0, "blocks8", 0),
         (3, 160, 64, 4, 1, "blocks16", 0), (2, 129, 17, 1, 0, "range", 0), (2, 4500, 6, 1, 1, "blocks", 0), (1, 8192, 16, 1, 1, "blocks", 0),
         (2, 210, 100, 1, 1, "blocks32", 0), (1, 300, 70, 2, 0, "range", 0), (3, 270, 256, 1, 1, "blocks", 0),
         # the band's width is free: narrow bands (many band starts, a band boundary inside an ownership block), one band for
         # the whole scan, and contiguous ranges through the same kernels
         (2, 4500, 6, 1, 1, "range", 0), (3, 1000, 9, 1, 1, "blocks64", 256), (2, 4500, 16, 1, 1, "blocks", 1024),
         (3, 9000, 12, 1, 1, "blocks", 65536), (2, 600, 40, 2, 1, "blocks128", 512)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,K,B,special,form,band", CASES)
def test_sharded_chain_equals_single_chain(world, N, K, B, special, form, band):
    """One chain, its nodes sharded over `world` ranks, against the oracle's sequential chain, bit for bit, from a random
    start (thousands of label changes in the first scans: every rank reports its first 32 surprises per round, the hypotheses
    are amended and accepted in bulk; N = 4500 crosses a band boundary of the 4096-node scan window; N = 8192, K = 16 - the
    network size of configs[4] - runs through the RCCL communicator; 100 / 70 / 256 blocks: a node's lanes spanning two or four
    wavefronts).  Round 4: the nodes are dealt out in BLOCKS (round-robin; 8 .. 256 nodes per block, also blocks that do not
    divide N and ranks that own one block more than others), contiguous ranges remain, and the band of the scan is as wide as
    the change count allows or as GG_SHARD_BAND forces (256 .. 65536 nodes) - the chain is the same whatever the width."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(29600 + CASES.index((world, N, K, B, special, form, band))),      # (a port per case)
           os.path.join(ROOT, "tests", "shard_worker.py"), str(N), str(K), "3", str(B), str(special), form]
    env = dict(os.environ)
    if band:
        env["GG_SHARD_BAND"] = str(band)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:      # keep the workers' whole output where a GPU-box run leaves its files
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "shard_case_%d_%d_%d_%s_%d.log" % (world, N, K, form, band)), "w") as f:
            f.write(out.stdout + "\n==== stderr ====\n" + out.stderr)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "SHARD-OK world=%d" % world in out.stdout, out.stdout[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(700, 6), (8192, 16)])
def test_sharded_bench_is_rank_count_invariant(N, K):
    """bench.py --shard on device-generated data: 1 rank and 2 ranks (sharing the GPU, gloo) run the same chain - also at
    N=8192 (two bands of the scan window per half-sweep, thousands of label changes in the first sweeps from the random
    start: every rank reports 32 surprises per round)."""
    import json
    digests = []
    for world in (1, 2):
        env = dict(os.environ, GG_BENCH_BACKEND="gloo", GG_BENCH_SAME_DEVICE="1")
        base = [os.path.join(ROOT, "bench.py"), "--shard", "--nodes", str(N), "--blocks", str(K), "--steps", "3", "--warmup", "2",
                "--gpus", str(world)]
        if world == 1:
            cmd = [sys.executable] + base
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                   "--master-addr", "127.0.0.1", "--master-port", str(29561 + (N > 1000))] + base
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert r["n_gpus"] == world and r["scaling"] == "strong"
        digests.append((r["config"]["labels_digest"], r["config"]["label_changes_in_timed_region"]))
    assert digests[0] == digests[1], digests


def test_block_ownership_covers_every_node_once():
    """sharded.Ownership (the host's statement of csrc/gg_common.h::gg_own): every node has exactly one owner, local order is
    global order, and the default block shrinks until every rank owns something."""
    import numpy as np
    from graphistician_amd.sharded import Ownership
    for N, world, block in ((1000, 3, 64), (65536, 8, None), (129, 2, 8), (600, 2, 128), (270, 3, None), (4500, 2, None), (100, 8, None)):
        owned = [Ownership(N, r, world, block).nodes() for r in range(world)]
        allv = np.concatenate(owned)
        assert np.array_equal(np.sort(allv), np.arange(N))
        assert all(np.all(np.diff(o) > 0) and o.size > 0 for o in owned)
        b = Ownership(N, 0, world, block).block
        assert all(np.all((o // b) % world == r) for r, o in enumerate(owned))
    with pytest.raises(ValueError):
        Ownership(100, 0, 2, block=12)


def test_shard_ranges_cover_and_are_never_empty():
    from graphistician_amd.sharded import shard_range
    for N, world in ((100, 3), (64, 8), (65536, 8), (7, 7), (129, 2), (1000, 6)):
        edges = [shard_range(N, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == N
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        assert all(n1 > n0 for n0, n1 in edges)
    with pytest.raises(ValueError):
        shard_range(3, 0, 4)
