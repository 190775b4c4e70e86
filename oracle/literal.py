"""Pure numpy/scipy statements of the reference's formulas for TINY cases.

TEST INFRASTRUCTURE ONLY.  These functions are written the way the reference
writes the same quantities (dense masks, per-edge scipy log-densities, Python
loops over nodes) and share nothing with the C++ oracle or the arithmetic spec:
no Philox, no custom exp/log, no sufficient statistics.  tests/ uses them to
check that the C++ oracle restates the right algorithm (tolerance 1e-10; libm vs
the spec's exp/log differ in the last ulp).
"""
import numpy as np
from scipy.stats import multivariate_normal


def adj_loglik(A, P):
    """abstractions.py:111-115 with safe_P (:92-94).  Entries with A == 0 contribute only the
    (1-A) term (the reference's 0 * log 0 = NaN corner is not reproduced)."""
    A = np.asarray(A, dtype=float)
    safe = np.clip(P, 1e-64, 1 - 1e-64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = np.where(A > 0, np.log(P), 0.0)
    return float(np.sum(t1 + (1 - A) * np.log(1 - safe)))


def sbm_P(c, p, p_self, allow_self=True, special=True):
    """adjacency.py:516-528."""
    P = p[np.ix_(c, c)].copy()
    if special:
        np.fill_diagonal(P, p_self)
    if not allow_self:
        np.fill_diagonal(P, 0.0)
    return P


def sbm_adj_counts(A, c, C, special=True):
    """adjacency.py:624-634: per block pair, masked edge / non-edge counts."""
    N = A.shape[0]
    A = np.asarray(A, dtype=int)
    n1 = np.zeros((C, C), dtype=int)
    n0 = np.zeros((C, C), dtype=int)
    for c1 in range(C):
        for c2 in range(C):
            mask = (c == c1)[:, None] & (c == c2)[None, :]
            if special:
                mask &= ~np.eye(N, dtype=bool)
            n1[c1, c2] = A[mask].sum()
            n0[c1, c2] = (1 - A[mask]).sum()
    return n1, n0


def sbm_adj_node_lp(A, c, m, p, n1, special=True):
    """adjacency.py:661-693: the conditional of node n1, every term written out."""
    N, C = A.shape[0], len(m)
    A = np.asarray(A, dtype=float)
    lp = np.zeros(C)
    lp += np.log(m)
    for k in range(C):
        for n2 in range(N):
            k2 = c[n2]
            if n2 == n1:
                if special:
                    continue
                lp[k] += A[n1, n1] * np.log(p[k, k]) + (1 - A[n1, n1]) * np.log(1 - p[k, k])
            else:
                lp[k] += A[n1, n2] * np.log(p[k, k2]) + (1 - A[n1, n2]) * np.log(1 - p[k, k2])
                lp[k] += A[n2, n1] * np.log(p[k2, k]) + (1 - A[n2, n1]) * np.log(1 - p[k2, k])
    return lp


def sample_discrete_from_log(lp, u):
    """pybasicbayes.util.stats.sample_discrete_from_log as recollected in SURVEY.md §2.2:
    inverse CDF on the cumulative sum of exp(lp - max)."""
    cum = np.cumsum(np.exp(lp - np.max(lp)))
    return int(np.sum(u * cum[-1] > cum))


def w_loglik(A, W, Mu, Sigma):
    """abstractions.py:257-274: one scipy multivariate normal per edge."""
    N = A.shape[0]
    ll = 0.0
    for i in range(N):
        for j in range(N):
            if A[i, j]:
                ll += multivariate_normal(Mu[i, j], Sigma[i, j]).logpdf(W[i, j])
    return float(ll)


def sbm_w_node_lp(A, W, c, m, mu, sigma, n1, special=True):
    """weights.py:451-470 (_evaluate_lkhd): masks per destination block, scipy log-densities."""
    N, C = A.shape[0], len(m)
    Ab = np.asarray(A).astype(bool)
    lp = np.log(m).copy()
    for k in range(C):
        chat = c.copy()
        chat[n1] = k
        ll = 0.0
        for c2 in range(C):
            out_mask = (chat == c2) & Ab[n1, :]
            if special:
                out_mask[n1] = False
            if out_mask.any():
                ll += np.sum(multivariate_normal(mu[k, c2], sigma[k, c2]).logpdf(W[n1, out_mask]))
            in_mask = (chat == c2) & Ab[:, n1]
            in_mask[n1] = False
            if in_mask.any():
                ll += np.sum(multivariate_normal(mu[c2, k], sigma[c2, k]).logpdf(W[in_mask, n1]))
        lp[k] += ll
    return lp


def niw_posterior(X, mu_0, sigma_0, nu_0, kappa_0):
    """NIW posterior hyper-parameters given rows of X (pybasicbayes Gaussian._posterior_hypparams)."""
    X = np.atleast_2d(X)
    n = X.shape[0]
    if n == 0:
        return mu_0, sigma_0, nu_0, kappa_0
    xbar = X.mean(0)
    S = (X - xbar).T.dot(X - xbar)
    kappa_n = kappa_0 + n
    mu_n = (kappa_0 * mu_0 + n * xbar) / kappa_n
    sigma_n = sigma_0 + S + kappa_0 * n / kappa_n * np.outer(xbar - mu_0, xbar - mu_0)
    return mu_n, sigma_n, nu_0 + n, kappa_n


def eigen_Mu(F, lmbda, mu_0):
    """eigenmodel.py:75-83."""
    Mu = mu_0 + (F * lmbda[None, :]).dot(F.T)
    np.fill_diagonal(Mu, mu_0)
    return Mu


def eigen_loglik(A, F, lmbda, mu_0):
    """eigenmodel.py:110-129 with P of :803-811."""
    P = np.clip(1.0 / (1.0 + np.exp(-eigen_Mu(F, lmbda, mu_0))), 1e-16, 1 - 1e-16)
    A = np.asarray(A, dtype=float)
    return float(np.sum(A * np.log(P) + (1 - A) * np.log(1 - P)))


def eigen_F_conditional(A, Omega, F, lmbda, mu_0, sigma_F, n):
    """eigenmodel.py:884-906: posterior precision and mean-dot-precision of F[n]."""
    N, D = F.shape
    A = np.asarray(A, dtype=float)
    llT = np.outer(lmbda, lmbda)
    prec = np.eye(D) / sigma_F
    h = np.zeros(D)
    zcent = (A - 0.5) / Omega - mu_0
    for nn in range(N):
        if nn == n:
            continue
        prec += np.outer(F[nn], F[nn]) * llT * Omega[n, nn]
        prec += np.outer(F[nn], F[nn]) * llT * Omega[nn, n]
        h += Omega[n, nn] * zcent[n, nn] * F[nn] * lmbda
        h += Omega[nn, n] * zcent[nn, n] * F[nn] * lmbda
    return prec, h


def eigen_mu0_conditional(A, Omega, F, lmbda, mu_mu_0, sigma_mu0):
    """eigenmodel.py:914-925: posterior precision and mean-dot-precision of mu_0."""
    A = np.asarray(A, dtype=float)
    prec = 1.0 / sigma_mu0 + Omega.sum()
    FdotF = (F * lmbda[None, :]).dot(F.T)
    np.fill_diagonal(FdotF, 0)
    h = mu_mu_0 / sigma_mu0 + ((A - 0.5) - Omega * FdotF).sum()
    return prec, h


def eigen_lmbda_conditional(A, Omega, F, mu_0, mu_lmbda, sigma_lmbda):
    """eigenmodel.py:938-958."""
    N, D = F.shape
    A = np.asarray(A, dtype=float)
    prec = np.eye(D) / sigma_lmbda
    h = mu_lmbda / sigma_lmbda
    zcent = (A - 0.5) / Omega - mu_0
    for n1 in range(N):
        for n2 in range(n1 + 1, N):
            prec = prec + (Omega[n2, n1] + Omega[n1, n2]) * np.outer(F[n1], F[n1]) * np.outer(F[n2], F[n2])
            f12 = F[n1] * F[n2]
            h = h + Omega[n1, n2] * zcent[n1, n2] * f12 + Omega[n2, n1] * zcent[n2, n1] * f12
    return prec, h
