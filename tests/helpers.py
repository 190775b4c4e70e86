"""Shared helpers for the parity tests: synthetic networks and model-state bundles."""
import numpy as np

from oracle import oracle as O


def planted_sbm(N, K, B=1, seed=0, sparse=False):
    """Planted Gaussian SBM data (the recipe of SURVEY.md §8d cfg 2), generated with numpy."""
    rng = np.random.RandomState(seed)
    m = rng.dirichlet(np.ones(K) * 5.0)
    c = rng.choice(K, p=m, size=N).astype(np.int32)
    p = rng.beta(1, 1, size=(K, K)) * (0.2 if sparse else 1.0)
    p_self = rng.beta(1, 1)
    P = p[np.ix_(c, c)]
    np.fill_diagonal(P, p_self)
    A = (rng.rand(N, N) < P).astype(np.uint8)
    mu = rng.randn(K, K, B) * 2.0
    W = mu[np.ix_(c, c)] + rng.randn(N, N, B) * 0.7
    W[np.arange(N), np.arange(N)] = rng.randn(N, B) * 0.5 + 1.0
    return A, W, c


def er_network(N, B=2, p=0.4, seed=0):
    rng = np.random.RandomState(seed)
    A = (rng.rand(N, N) < p).astype(np.uint8)
    W = rng.randn(N, N, B) * 1.3 + 0.5
    return A, W


class AdjState(object):
    """Host copy of an SBM adjacency model state (for the oracle)."""

    def __init__(self, N, K, seed=0, pi=1.0, tau1=1.0, tau0=1.0):
        rng = np.random.RandomState(seed)
        self.N, self.K = N, K
        self.pi = np.ones(K) * pi
        self.tau1, self.tau0 = tau1, tau0
        self.m = rng.dirichlet(self.pi)
        self.c = rng.choice(K, size=N).astype(np.int32)
        self.p = np.ascontiguousarray(rng.beta(tau1, tau0, size=(K, K)))
        self.p_self = float(rng.beta(tau1, tau0))


class WState(object):
    def __init__(self, N, K, B, seed=0, pi=10.0):
        rng = np.random.RandomState(seed)
        self.N, self.K, self.B = N, K, B
        self.pi = np.ones(K) * pi
        self.m = rng.dirichlet(self.pi)
        self.c = rng.choice(K, size=N).astype(np.int32)
        self.mu = np.ascontiguousarray(rng.randn(K, K, B))
        S = np.zeros((K, K, B, B))
        for a in range(K):
            for b in range(K):
                X = rng.randn(B, B + 2)
                S[a, b] = X.dot(X.T) / (B + 2) + 0.3 * np.eye(B)
        self.sigma = np.ascontiguousarray(S)
        self.mu_self = rng.randn(B)
        self.sigma_self = np.eye(B) * 0.8
        self.niw = O.pack_niw(np.zeros(B), np.eye(B), B + 2, 1.0)
        self.niw_self = O.pack_niw(np.zeros(B), np.eye(B), B + 2, 1.0)
