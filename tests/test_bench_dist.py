"""World-size-2 gloo test (CPU) of bench.py's N>1 protocol: one process per rank launched by
torch.distributed.run, rendezvous on 127.0.0.1, barrier on both sides of the timed region,
MAX over ranks of the elapsed time, whole-job value on rank 0 only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_gloo():
    env = dict(os.environ, GG_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--dry-run"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0: %r" % out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["scaling"] == "weak"
    assert r["elapsed_s"] >= 0.04            # rank 1 sleeps 40 ms: the max over ranks, not rank 0's 20 ms
    assert abs(r["value"] - 2 * 4 / r["elapsed_s"]) < 1e-6 * r["value"]


def test_bench_single_process_dry_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "3"], cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["steps"] == 3


@pytest.mark.gpu
def test_combined_replica_and_sharded_line_with_two_ranks():
    """The form the driver's scaling run executes, `bench.py --gpus G` with G > 1: every rank runs a replica of
    the chain AND, in the same invocation, all ranks run ONE chain of that network sharded over them; rank 0 prints ONE JSON
    line that carries both.  Two ranks sharing this box's one GPU (gloo for the rendezvous and for the library's collectives -
    RCCL refuses two ranks on one device), at N=4096 so that the box holds both."""
    import json
    env = dict(os.environ, GG_BENCH_BACKEND="gloo", GG_BENCH_SAME_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nodes", "4096", "--steps", "6", "--warmup", "2",
           "--repeats", "2", "--no-cpu", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0 and "labels_digest" in r["config"]
    assert r["config"]["timed_blocks_per_chain"] == 2 and len(r["config"]["chains"]) == 6
    sh = r["sharded_chain"]
    assert "error" not in sh, sh
    assert sh["n_gpus"] == 2 and sh["scaling"] == "strong" and sh["value"] > 0
    assert sh["config"]["ranks"] == 2 and "labels_digest" in sh["config"] and sh["config"]["collectives_on_the_wire_in_run"] > 0


@pytest.mark.gpu
def test_default_two_rank_line_keeps_the_headline_workload_and_carries_the_configs3_legs():
    """`bench.py --gpus 2` as the driver calls it (no --nodes): the headline stays configs[1] - the same workload as the
    --gpus 1 line, so the per-G values the driver divides are values of one workload - and configs[3] comes as two legs of the
    same line: one replica of the chain per GPU and ONE chain sharded over the ranks.  Two ranks on this box's one GPU (gloo),
    the legs at N=8192, K=64 through GG_BENCH_BIG_NODES so that the test stays short."""
    import json
    env = dict(os.environ, GG_BENCH_BACKEND="gloo", GG_BENCH_SAME_DEVICE="1", GG_BENCH_BIG_NODES="8192", GG_BENCH_BIG_BLOCKS="64")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--repeats", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert "N=4096, K=16" in r["config"]["workload"] and len(r["config"]["chains"]) == 6
    big = r["other_configs"]["sbm_n8192_k64_replicas"]
    assert "error" not in big and "skipped" not in big, big
    assert big["n_gpus"] == 2 and big["value"] > 0 and big["steps"] == 6
    sh = r["sharded_chain"]
    assert "error" not in sh, sh
    assert sh["n_gpus"] == 2 and sh["scaling"] == "strong" and sh["value"] > 0 and sh["config"]["ranks"] == 2
    assert "N=8192" in sh["config"]["workload"] or sh["config"].get("nodes") == 8192, sh["config"]
