"""Generates the golden fixtures in this directory from the CPU oracle (fixed Philox
seeds).  The reference holds no fixtures for this path and cannot be run
(SURVEY.md §8c), so these are REGRESSION vectors: they pin the oracle (and through
it the HIP engine) across rounds; evidence of reference parity is in
tests/test_spec_and_oracle.py.   Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402
from tests.helpers import planted_sbm, er_network, AdjState, WState  # noqa: E402


def sbm_case(N, K, B, seed, nsweeps, special):
    A, W, _ = planted_sbm(N, K, B=B, seed=seed)
    q, shift = O.quantise(A, W)
    a, w = AdjState(N, K, seed=seed + 1), WState(N, K, B, seed=seed + 2)
    init = dict(a_c=a.c.copy(), a_m=a.m.copy(), a_p=a.p.copy(), a_p_self=a.p_self, a_pi=a.pi,
                w_c=w.c.copy(), w_m=w.m.copy(), w_mu=w.mu.copy(), w_sigma=w.sigma.copy(), w_mu_self=w.mu_self.copy(),
                w_sigma_self=w.sigma_self.copy(), w_pi=w.pi, w_niw=w.niw, w_niw_self=w.niw_self)
    ps = a.p_self
    trace_c = []
    for s in range(nsweeps):
        ps = O.sbm_adj_sweep(A, a.c, a.m, a.p, ps, a.pi, 1.0, 1.0, special, 1000 + seed, s)
        O.sbm_w_sweep(A, q, shift, w.c, w.m, w.mu, w.sigma, w.mu_self, w.sigma_self, w.pi, w.niw, w.niw_self, special,
                      2000 + seed, s)
        trace_c.append(np.concatenate([a.c, w.c]))
    ll_a = O.sbm_adj_loglik(A, a.c, a.p, ps, 1, special)
    ll_w = O.sbm_w_loglik(A, q, shift, w.c, w.mu, w.sigma, w.mu_self, w.sigma_self, special)
    out = dict(N=N, K=K, B=B, nsweeps=nsweeps, special=special, seed_a=1000 + seed, seed_w=2000 + seed, A=A, W=W,
               shift=shift, out_a_c=a.c, out_a_m=a.m, out_a_p=a.p, out_a_p_self=ps, out_w_c=w.c, out_w_m=w.m,
               out_w_mu=w.mu, out_w_sigma=w.sigma, out_w_mu_self=w.mu_self, out_w_sigma_self=w.sigma_self,
               trace_c=np.array(trace_c), ll_a=ll_a, ll_w=ll_w)
    out.update({"init_" + k: v for k, v in init.items()})
    return out


def er_case(N, B, seed, nsweeps):
    """Config 1: Beta-Bernoulli adjacency + NIW weights (plumbing case, N=50)."""
    A, W = er_network(N, B=B, seed=seed)
    q, shift = O.quantise(A, W)
    niw = O.pack_niw(np.zeros(B), np.eye(B), B + 2, 1.0)
    mu, sig, mus, sigs = np.zeros((1, 1, B)), np.zeros((1, 1, B, B)), np.zeros(B), np.zeros((B, B))
    ps_trace = []
    for s in range(nsweeps):
        p, ps = O.bb_adj_sweep(A, 1.0, 1.0, 1, 3000 + seed, s)
        O.niw_w_sweep(A, q, shift, mu, sig, mus, sigs, niw, niw, 4000 + seed, s)
        ps_trace.append([p, ps])
    ll_a = O.const_adj_loglik(A, p, ps)
    ll_w = O.sbm_w_loglik(A, q, shift, np.zeros(N, np.int32), mu, sig, mus, sigs, 1)
    return dict(N=N, B=B, nsweeps=nsweeps, seed_a=3000 + seed, seed_w=4000 + seed, A=A, W=W, shift=shift, niw=niw,
                out_p=p, out_p_self=ps, p_trace=np.array(ps_trace), out_mu=mu, out_sigma=sig, out_mu_self=mus,
                out_sigma_self=sigs, ll_a=ll_a, ll_w=ll_w)


def eigen_case(N, D, seed, nsweeps, probit):
    """Config 3 (logistic) / SURVEY 8f rank 2 (probit): eigenmodel sweeps from a fixed start."""
    rng = np.random.RandomState(seed)
    Ft, lt, m0t = rng.randn(N, D), rng.randn(D), -0.5
    Mu = m0t + (Ft * lt).dot(Ft.T)
    np.fill_diagonal(Mu, m0t)
    if probit:
        from scipy.special import ndtr
        A = (rng.rand(N, N) < ndtr(Mu)).astype(np.uint8)
    else:
        A = (rng.rand(N, N) < 1.0 / (1.0 + np.exp(-Mu))).astype(np.uint8)
    F = np.ascontiguousarray(rng.randn(N, D) * 0.7)
    lmbda = np.ascontiguousarray(rng.randn(D))
    mu0 = 0.1
    init = dict(init_F=F.copy(), init_lmbda=lmbda.copy(), init_mu0=mu0)
    sweep = O.probit_sweep if probit else O.eig_sweep
    for s in range(nsweeps):
        mu0, _ = sweep(A, F, lmbda, mu0, np.zeros(D), -0.3, 1.0, 1.0, 1.0, 0, 5000 + seed, s)
    ll = (O.probit_loglik if probit else O.eig_loglik)(A, F, lmbda, mu0)
    out = dict(N=N, D=D, nsweeps=nsweeps, probit=int(probit), seed=5000 + seed, A=A, out_F=F, out_lmbda=lmbda, out_mu0=mu0, ll=ll)
    out.update(init)
    return out


def latent_case(N, dim, seed, nsweeps):
    """latent-distance adjacency model, HMC resamples from a fixed start."""
    rng = np.random.RandomState(seed)
    Lt = rng.randn(N, dim)
    X = -((Lt[:, None, :] - Lt[None, :, :]) ** 2).sum(2) + 0.4
    A = (rng.rand(N, N) < 1.0 / (1.0 + np.exp(-X))).astype(np.uint8)
    L = np.ascontiguousarray(rng.randn(N, dim))
    init = dict(init_L=L.copy(), init_mu0=0.1, init_mu_self=-0.1)
    mu0, ms, st, ar = 0.1, -0.1, 0.01, 0.9
    for s in range(nsweeps):
        mu0, ms, st, ar, _ = O.ld_resample(A, L, mu0, ms, 2.0, st, ar, 6000 + seed, s)
    out = dict(N=N, dim=dim, nsweeps=nsweeps, seed=6000 + seed, sigma=2.0, A=A, out_L=L, out_mu0=mu0, out_mu_self=ms,
               out_step=st, out_rate=ar, ll=O.ld_loglik(A, L, mu0, ms))
    out.update(init)
    return out


def latent_w_case(N, dim, seed, nsweeps):
    """SURVEY 8f rank 3, weight side: latent-distance Gaussian weights (weights.py:613-903) from a fixed start."""
    rng = np.random.RandomState(seed)
    Lt = rng.randn(N, dim) * 0.6
    A = (rng.rand(N, N) < 0.5).astype(np.uint8)
    W = -((Lt[:, None, :] - Lt[None, :, :]) ** 2).sum(2) + 0.5 + 0.3 * rng.randn(N, N)
    W[np.arange(N), np.arange(N)] = 1.5 + 0.2 * rng.randn(N)
    W = np.ascontiguousarray(W[:, :, None])
    q, shift = O.quantise(A, W)
    L = np.ascontiguousarray(rng.randn(N, dim) * 0.3)
    prior, prior_self = np.array([1.0, 3.0]), np.array([0.0, 1.0, 3.0, 1.0])
    state = dict(b=0.2, eta=0.3, sigma=0.8, mu_self=0.1, sigma_self=1.1)
    out = dict(N=N, dim=dim, nsweeps=nsweeps, seed=7000 + seed, A=A, W=W, init_L=L.copy(), prior=prior, prior_self=prior_self)
    out.update({"init_" + k: v for k, v in state.items()})
    b, eta, sg, ms, ss = (state[k] for k in ("b", "eta", "sigma", "mu_self", "sigma_self"))
    for s in range(nsweeps):
        b, eta, sg, ms, ss, _ = O.ldw_resample(A, q, shift, L, b, eta, sg, ms, ss, prior, prior_self, 7000 + seed, s)
    out.update(out_L=L, out_b=b, out_eta=eta, out_sigma=sg, out_mu_self=ms, out_sigma_self=ss,
               ll=O.ldw_loglik(A, q, shift, L, b, sg, ms, ss))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "sbm_n64_k4_b1.npz"), **sbm_case(64, 4, 1, 7, 6, 1))
    np.savez_compressed(os.path.join(HERE, "sbm_n48_k3_b2_nospecial.npz"), **sbm_case(48, 3, 2, 9, 4, 0))
    np.savez_compressed(os.path.join(HERE, "er_n50_b2.npz"), **er_case(50, 2, 1234, 5))
    np.savez_compressed(os.path.join(HERE, "eigen_logistic_n40_d3.npz"), **eigen_case(40, 3, 11, 3, False))
    np.savez_compressed(os.path.join(HERE, "eigen_probit_n40_d3.npz"), **eigen_case(40, 3, 12, 3, True))
    np.savez_compressed(os.path.join(HERE, "latent_n40_dim2.npz"), **latent_case(40, 2, 13, 4))
    np.savez_compressed(os.path.join(HERE, "latentw_n40_dim2.npz"), **latent_w_case(40, 2, 14, 4))
    print("golden fixtures written to", HERE)
