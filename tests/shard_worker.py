"""Worker of tests/test_sharded.py: every rank drives its shard of ONE Gaussian-SBM chain; rank 0 checks
labels and parameters against the CPU oracle's plain sequential chain (bit for bit)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    from graphistician_amd import _abi as abi
    from graphistician_amd import sharded as S
    from oracle import oracle as O
    from tests.helpers import planted_sbm, AdjState, WState
    rank, world = dist.get_rank(), dist.get_world_size()
    # one rank: the RCCL communicator (the library's own ncclAllReduce / ncclAllGather calls run, over one rank);
    # several ranks sharing the test box's single GPU: the same protocol over host callbacks (gloo)
    use_rccl = os.environ.get("GG_SHARD_COMM", "rccl" if world == 1 else "host") == "rccl"
    comm = S.rccl_comm(dist, 0) if use_rccl else S.host_comm(dist, 0)
    N, K, nsweeps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    special = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    A, W, _ = planted_sbm(N, K, B=B, seed=77)
    # ownership: "blocks" (default: blocks of `block` nodes dealt out round-robin) or "range" (contiguous node ranges)
    form = sys.argv[6] if len(sys.argv) > 6 else "blocks"
    if form == "range":
        n0, n1 = S.shard_range(N, rank, world)
        data = abi.DeviceData.shard(A, W, n0, n1, device=0)
        own = dict(shard_begin=n0, shard_end=n1)
    else:
        o = S.Ownership(N, rank, world, block=int(form[6:]) if form.startswith("blocks") and len(form) > 6 else None)
        data = abi.DeviceData.shard(A, W, device=0, blocks=o.blocks)
        own = o.config
    a, w = AdjState(N, K, seed=5), WState(N, K, B, seed=6)
    adj = abi.Model(kind=abi.SBM_ADJ, N=N, K=K, B=1, seed=901, tau1=1.0, tau0=1.0, allow_self_connections=1,
                    special_case_self_conns=special, **own)
    adj.set(abi.ST_C, a.c, np.int32); adj.set(abi.ST_M, a.m); adj.set(abi.ST_PI, a.pi); adj.set(abi.ST_P, a.p)
    adj.set(abi.ST_P_SELF, np.array([a.p_self]))
    wm = abi.Model(kind=abi.SBM_GAUSS_W, N=N, K=K, B=B, seed=902, special_case_self_conns=special, **own)
    for sid, v, dt in ((abi.ST_C, w.c, np.int32), (abi.ST_M, w.m, np.float64), (abi.ST_PI, w.pi, np.float64),
                       (abi.ST_MU, w.mu, np.float64), (abi.ST_SIGMA, w.sigma, np.float64),
                       (abi.ST_MU_SELF, w.mu_self, np.float64), (abi.ST_SIGMA_SELF, w.sigma_self, np.float64),
                       (abi.ST_NIW, w.niw, np.float64), (abi.ST_NIW_SELF, w.niw_self, np.float64)):
        wm.set(sid, v, dt)
    if os.environ.get("GG_SHARD_DIAG"):
        # diagnosis: the oracle's chain sweep by sweep beside the engine's, the first difference printed (rank 0)
        q, shift = O.quantise(A, W)
        ps = a.p_self
        for s in range(nsweeps):
            S.half_sweep(adj, data, comm)
            S.half_sweep(wm, data, comm)
            ps = O.sbm_adj_sweep(A, a.c, a.m, a.p, ps, a.pi, 1.0, 1.0, special, 901, s)
            O.sbm_w_sweep(A, q, shift, w.c, w.m, w.mu, w.sigma, w.mu_self, w.sigma_self, w.pi, w.niw, w.niw_self, special, 902, s)
            ca, cw = adj.get(abi.ST_C, (N,), np.int32), wm.get(abi.ST_C, (N,), np.int32)
            if rank == 0:
                da, dw = np.nonzero(ca != a.c)[0], np.nonzero(cw != w.c)[0]
                print("DIAG sweep %d: adj labels differ at %s, p %s, m %s | w labels differ at %s, mu %s, m %s | changes %d %d" % (
                    s, da[:8].tolist(), np.array_equal(adj.get(abi.ST_P, (K, K)), a.p), np.array_equal(adj.get(abi.ST_M, (K,)), a.m),
                    dw[:8].tolist(), np.array_equal(wm.get(abi.ST_MU, (K, K, B)), w.mu), np.array_equal(wm.get(abi.ST_M, (K,)), w.m),
                    adj.stats()["label_changes"], wm.stats()["label_changes"]), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    for s in range(nsweeps - 1):
        S.half_sweep(adj, data, comm)
        S.half_sweep(wm, data, comm)
    if nsweeps > 1:                      # (two half-sweeps in one call as well)
        S.half_sweep(adj, data, comm, 1)
        S.half_sweep(wm, data, comm, 1)
    ll = S.log_likelihood(adj, data, comm) + S.log_likelihood(wm, data, comm)
    # every rank holds every label and draws every parameter: the replicas must agree to the bit (rank 0 alone is compared with
    # the oracle below; a rank that lost an accepted change shows up here even while rank 0 is still right)
    import hashlib
    mine = hashlib.sha1(b"".join(np.ascontiguousarray(x).tobytes() for x in (
        adj.get(abi.ST_C, (N,), np.int32), adj.get(abi.ST_P, (K, K)), adj.get(abi.ST_M, (K,)), wm.get(abi.ST_C, (N,), np.int32),
        wm.get(abi.ST_MU, (K, K, B)), wm.get(abi.ST_SIGMA, (K, K, B, B)), wm.get(abi.ST_M, (K,))))).hexdigest()
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    assert len(set(everyone)) == 1, "the ranks' replicas of the chain differ: %s" % everyone
    if rank == 0:
        q, shift = O.quantise(A, W)
        ps = a.p_self
        for s in range(nsweeps):
            ps = O.sbm_adj_sweep(A, a.c, a.m, a.p, ps, a.pi, 1.0, 1.0, special, 901, s)
            O.sbm_w_sweep(A, q, shift, w.c, w.m, w.mu, w.sigma, w.mu_self, w.sigma_self, w.pi, w.niw, w.niw_self, special, 902, s)
        assert np.array_equal(adj.get(abi.ST_C, (N,), np.int32), a.c), "adjacency labels differ"
        assert np.array_equal(adj.get(abi.ST_P, (K, K)), a.p) and np.array_equal(adj.get(abi.ST_M, (K,)), a.m)
        assert adj.get(abi.ST_P_SELF, (1,))[0] == ps
        assert np.array_equal(wm.get(abi.ST_C, (N,), np.int32), w.c), "weight labels differ"
        assert np.array_equal(wm.get(abi.ST_MU, (K, K, B)), w.mu) and np.array_equal(wm.get(abi.ST_SIGMA, (K, K, B, B)), w.sigma)
        assert np.array_equal(wm.get(abi.ST_M, (K,)), w.m)
        ref = (O.sbm_adj_loglik(A, a.c, a.p, ps, True, bool(special)) +
               O.sbm_w_loglik(A, q, shift, w.c, w.mu, w.sigma, w.mu_self, w.sigma_self, bool(special)))
        assert abs(ll - ref) <= 1e-9 * abs(ref), (ll, ref)
        info = comm.info()
        assert info["world"] == world and info["rccl"] == use_rccl and info["rounds"] >= 2 * nsweeps
        assert (info["collectives"] > 2 * nsweeps) if world > 1 else info["collectives"] == 0      # (one rank: nothing travels)
        print("SHARD-OK world=%d N=%d K=%d B=%d special=%d sweeps=%d changes=%d comm=%s own=%s rounds=%d collectives=%d" % (
            world, N, K, B, special, nsweeps, adj.stats()["label_changes"] + wm.stats()["label_changes"],
            "rccl" if info["rccl"] else "host", form, info["rounds"], info["collectives"]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
