"""One Gibbs chain sharded over several GPUs — exactly the single-GPU chain.

Each rank owns a share of the nodes - by default blocks of 256 consecutive nodes dealt out round-robin
(`Ownership`), so that every stretch of the sequential scan spreads over all ranks; a contiguous range
(`shard_range`) is the other form - and stores the matching column shards of the data and of its transpose
(`DeviceData.shard`), so the block histograms of its own nodes are complete locally.
What travels, per half-sweep (all inside `gg_shard_sweep`, on the engine stream, device buffers):

  * one all-reduce (SUM, int64) of the K x K sufficient-statistic tables
    -> every rank draws the same parameters from the same Philox stream (adjacency.py:620-648,
    weights.py:399-413);
  * the sequential scan (adjacency.py:650-695, weights.py:415-488) in rounds over bands of 4096 .. 65536 nodes
    (as wide as the previous scan's change count lets a round's reports describe): every
    rank evaluates its own nodes under the changes hypothesised so far and reports the first 64 whose
    decision differs from the hypothesis, as (node, labels) pairs; ONE all-gather per round; every
    rank then accepts the hypotheses up to the earliest report (labels are replicated, the statistics
    of owned nodes are corrected from the local rows of X / X^T) and amends the rest.  A round
    without a report closes the band.

All traffic is integer, so the result does not depend on the number of ranks: labels and drawn
parameters are bit-identical to `SBMAdjacencyDistribution.resample` on one GPU (tests/test_sharded.py).
This replaces the approximate "shard-synchronous" schedule sketched in SURVEY.md section 8e.

The collectives are RCCL calls made by `libgraphgibbs.so` itself (`rccl_comm`: one process per GPU; the
128-byte unique id travels once through the launcher's store).  `host_comm` routes them through host
callbacks instead - gloo in the CPU-side protocol tests, where several ranks share one GPU.
torch.distributed is only the launcher / rendezvous here, never on the data path.
"""
import numpy as np

from . import _abi


def shard_range(N, rank, world):
    """Contiguous node ranges that cover [0, N), every rank non-empty (sizes differ by at most one)."""
    if N < world:
        raise ValueError("cannot shard %d nodes over %d ranks: every rank needs at least one node" % (N, world))
    base, extra = divmod(N, world)
    n0 = rank * base + min(rank, extra)
    return n0, n0 + base + (1 if rank < extra else 0)


class Ownership(object):
    """Block-interleaved ownership of N nodes: blocks of `block` consecutive nodes (a power of two >= 8), block b to rank
    b % world.  Every band of the scan then holds nodes of every rank, so a round's evaluation shrinks with the number of
    ranks (contiguous ranges leave it with the one rank that owns that stretch).  `config` goes into the model's constructor,
    `blocks` into `DeviceData.shard` / `DeviceData.synthetic_sbm`."""

    def __init__(self, N, rank, world, block=None):
        if block is None:
            block = 256
            while block > 8 and block * world > N:
                block //= 2
        if block < 8 or block & (block - 1):
            raise ValueError("block must be a power of two >= 8")
        if rank * block >= N:
            raise ValueError("cannot deal %d nodes in blocks of %d to %d ranks: rank %d would own none" % (N, block, world, rank))
        self.N, self.rank, self.world, self.block = N, rank, world, block

    @property
    def config(self):
        return dict(shard_block=self.block, shard_rank=self.rank, shard_world=self.world)

    @property
    def blocks(self):
        return (self.block, self.rank, self.world)

    def nodes(self):
        """Global indices of the owned nodes, in local order."""
        n = np.arange(self.N)
        return n[(n // self.block) % self.world == self.rank]


def local_comm(device=0):
    """Single rank (no collective ever leaves the device)."""
    return _abi.Comm.rccl(0, 1, _abi.Comm.unique_id(), device=device)


def rccl_comm(dist, device):
    """RCCL communicator over the ranks of an initialised torch.distributed group (used as rendezvous only: rank 0's
    unique id is broadcast as a Python object)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [_abi.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return _abi.Comm.rccl(rank, world, box[0], device=device)


def host_comm(dist, device=0):
    """The same protocol with the collectives done by torch.distributed on host tensors (gloo): the test harness for
    more ranks than there are GPUs."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()

    def allreduce_sum(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def allgather(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.stack(out).numpy()

    return _abi.Comm.host(rank, world, allreduce_sum, allgather, device=device)


def half_sweep(model, data, comm, n=1):
    """n x resample(...) of a sharded SBM component (adjacency.py:615-618 / weights.py:394-397)."""
    model.shard_sweep(data, comm, n)


def log_likelihood(model, data, comm):
    return model.shard_log_likelihood(data, comm)
