"""One Gibbs chain sharded over several GPUs — exactly the single-GPU chain.

Each rank owns a contiguous range of nodes and stores the matching column shards of the data and
of its transpose (`DeviceData.shard`), so the block histograms of its own nodes are complete
locally.  What travels, per half-sweep:

  * one all-reduce (SUM, int64) of the K x K sufficient-statistic tables and the diagonal statistics
    -> every rank draws the same parameters from the same Philox stream (adjacency.py:620-648,
    weights.py:399-413);
  * the sequential scan (adjacency.py:650-695, weights.py:415-488): ranks speculate their own nodes
    window by window; an all-gather of (first node whose draw differs from its label, old, new)
    triples picks the earliest change, every rank applies it (labels are replicated, the statistics
    of owned nodes are corrected from the local row of X / X^T) and the scan resumes after it.

All traffic is integer, so the result does not depend on the number of ranks: labels and drawn
parameters are bit-identical to `SBMAdjacencyDistribution.resample` on one GPU (tests/test_sharded.py).
This replaces the approximate "shard-synchronous" schedule sketched in SURVEY.md section 8e.

`comm` needs `allreduce_sum(int64 ndarray) -> ndarray` and `allgather(int32 ndarray) -> (world, n) ndarray`.
"""
import numpy as np

from . import _abi

NONE = 0x7fffffff


class LocalComm(object):
    """Single rank."""
    rank, world = 0, 1

    def allreduce_sum(self, a):
        return a

    def allgather(self, a):
        return np.asarray(a)[None, :]


class TorchComm(object):
    """torch.distributed (backend "nccl" = RCCL over xGMI on GPU tensors, or "gloo" on host tensors)."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def allreduce_sum(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return self.torch.stack(out).cpu().numpy()


def shard_range(N, rank, world, align=64):
    """Contiguous, 64-aligned node ranges that cover [0, N)."""
    per = -(-N // world)
    per = -(-per // align) * align
    n0 = min(rank * per, N)
    return n0, min(n0 + per, N)


def half_sweep(model, data, comm):
    """One resample(...) of a sharded SBM component (adjacency.py:615-618 / weights.py:394-397)."""
    t, dg = model.shard_tables(data)
    both = comm.allreduce_sum(np.concatenate([t, dg]))
    model.shard_params(data, both[:t.size], both[t.size:])
    while True:
        out = model.shard_next(data)                       # (node, old, new, window end)
        allg = comm.allgather(out[:3])
        w = int(np.argmin(allg[:, 0]))
        if allg[w, 0] == NONE:
            model.shard_apply(data, -1, int(out[3]), 0)    # quiet window: move on
            if out[3] >= data.N:
                break
        else:
            model.shard_apply(data, int(allg[w, 0]), int(allg[w, 1]), int(allg[w, 2]))
    model.shard_finish()


def log_likelihood(model, data, comm):
    t, dg = model.shard_tables(data)
    both = comm.allreduce_sum(np.concatenate([t, dg]))
    return model.shard_log_likelihood(data, both[:t.size], both[t.size:])
