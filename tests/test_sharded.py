"""Exact sharded chain: 1, 2 and 3 ranks — processes sharing the one GPU of the test
box, collectives over gloo — must all reproduce the oracle's single sequential chain bit for bit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,K", [(1, 150, 4), (2, 200, 5), (3, 330, 7)])
def test_sharded_chain_equals_single_chain(world, N, K):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(29540 + world), os.path.join(ROOT, "tests", "shard_worker.py"), str(N), str(K), "3"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "SHARD-OK world=%d" % world in out.stdout, out.stdout[-1500:]
