"""AIS model comparison (BASELINE.json configs[4]): Erdos-Renyi vs SBM vs latent-distance vs logistic-eigenmodel evidence of a
planted-SBM graph,
independent tempered chains, one per GPU when launched with torch.distributed.run.

Mirrors the reference's examples/demo_ais.py:64-110 (models: ER x=(1,1); SBM R=3, b0=b1=1, alpha0=10).

    python examples/demo_ais.py [N] [chains] [B]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/demo_ais.py 8192 8 100
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd.adjacency import (SBMAdjacencyDistribution, BetaBernoulliAdjacencyDistribution,  # noqa: E402
                                         LatentDistanceAdjacencyDistribution)
from graphistician_amd.eigenmodel import LogisticEigenmodel  # noqa: E402
from graphistician_amd.ais import estimate_marginal_likelihood, combine_log_weights  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    chains = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend=os.environ.get("GG_BENCH_BACKEND", "gloo"))   # only log-weights travel
    R = 3
    rng = np.random.RandomState(0)                       # every rank builds the same graph
    c = rng.choice(R, size=N)
    Bm = rng.beta(1, 1, size=(R, R))
    A = (rng.rand(N, N) < Bm[np.ix_(c, c)]).astype(np.uint8)
    np.random.seed(1000 + rank)
    mine = [g for g in range(chains) if g % world == rank]
    er = BetaBernoulliAdjacencyDistribution(N, tau1=1.0, tau0=1.0, device=device, seed=10 + rank)
    sbm = SBMAdjacencyDistribution(N, C=R, pi=10.0, tau0=1.0, tau1=1.0, device=device, seed=20 + rank)
    _, w_er = estimate_marginal_likelihood(er, A, N_samples=len(mine), B=B, steps_per_B=1, return_weights=True)
    _, w_sbm = estimate_marginal_likelihood(sbm, A, N_samples=len(mine), B=B, steps_per_B=2, return_weights=True)
    ld = LatentDistanceAdjacencyDistribution(N, dim=2, sigma=2.0, device=device, seed=30 + rank)
    _, w_ld = estimate_marginal_likelihood(ld, A, N_samples=len(mine), B=B, steps_per_B=2, return_weights=True)
    # tempered logistic eigenmodel: build-defined (the reference leaves it a TODO, graph_models.py:478)
    eig = LogisticEigenmodel(N, 2, p=0.5, device=device, seed=40 + rank)
    _, w_eig = estimate_marginal_likelihood(eig, A, N_samples=len(mine), B=B, steps_per_B=2, return_weights=True)
    if dist is not None:
        parts = [None] * world
        dist.all_gather_object(parts, (w_er, w_sbm, w_ld, w_eig))
        w_er = np.concatenate([p[0] for p in parts]); w_sbm = np.concatenate([p[1] for p in parts])
        w_ld = np.concatenate([p[2] for p in parts]); w_eig = np.concatenate([p[3] for p in parts])
    if rank == 0:
        print("uniform      log p(A) = %.2f" % (N * N * np.log(0.5)))
        print("Erdos-Renyi  log p(A) ~ %.2f   (%d chains)" % (combine_log_weights([w_er]), w_er.size))
        print("SBM (R=%d)    log p(A) ~ %.2f   (%d chains)" % (R, combine_log_weights([w_sbm]), w_sbm.size))
        print("latent dist. log p(A) ~ %.2f   (%d chains)" % (combine_log_weights([w_ld]), w_ld.size))
        print("eigenmodel   log p(A) ~ %.2f   (%d chains)" % (combine_log_weights([w_eig]), w_eig.size))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
