"""Synthetic data generator shared by tests, smoke and bench.

Gaussian-mixture base vectors, k-means codebooks (any k-means is allowed: learning is offline in
the reference, J/quantization/*), residuals with the reference's sign (centroid - vector,
J/quantization/ResidualVectorComputation.java:34).
"""
import numpy as np


def kmeans(X, k, iters=8, seed=0):
    rng = np.random.default_rng(seed)
    X = np.asarray(X, np.float64)
    cent = X[rng.choice(X.shape[0], size=k, replace=X.shape[0] < k)].copy()
    for _ in range(iters):
        d = (X * X).sum(1)[:, None] - 2.0 * X @ cent.T + (cent * cent).sum(1)[None, :]
        a = d.argmin(1)
        for c in range(k):
            sel = a == c
            if sel.any():
                cent[c] = X[sel].mean(0)
            else:
                cent[c] = X[rng.integers(X.shape[0])] + 1e-3 * rng.standard_normal(X.shape[1])
    return cent


def mixture(n, D, G, sigma=0.15, seed=1234):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((G, D))
    g = rng.integers(0, G, size=n)
    return mu[g] + sigma * rng.standard_normal((n, D)), mu


def make_ivfpq_problem(n=4000, D=32, C=16, m=8, ks=32, nq=24, seed=7, sigma=0.15, qsigma=0.01, iters=6):
    """Returns dict(base, coarse, pq, queries). ks may be < 256 to keep tests fast."""
    rng = np.random.default_rng(seed)
    base, mu = mixture(n, D, C, sigma=sigma, seed=seed)
    coarse = kmeans(base[: min(n, 20000)], C, iters=iters, seed=seed + 1)
    d = ((base[:, None, :] - coarse[None, :, :]) ** 2).sum(-1) if n * C * D < 5e7 else None
    if d is None:
        d = (base * base).sum(1)[:, None] - 2 * base @ coarse.T + (coarse * coarse).sum(1)[None]
    cell = d.argmin(1)
    resid = coarse[cell] - base  # centroid - vector
    dsub = D // m
    pq = np.zeros((m, ks, dsub))
    for s in range(m):
        pq[s] = kmeans(resid[:, s * dsub:(s + 1) * dsub], ks, iters=iters, seed=seed + 10 + s)
    qi = rng.integers(0, n, size=nq)
    queries = base[qi] + qsigma * rng.standard_normal((nq, D))
    return dict(base=base, coarse=coarse, pq=pq, queries=queries, qi=qi)


def make_pq_problem(n=5000, D=32, m=4, ks=32, nq=16, seed=11, iters=6):
    """Flat PQ: iid N(0, I) base vectors (a tight mixture would collapse to a few codes and
    create thousands of exact distance ties)."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, D))
    dsub = D // m
    pq = np.zeros((m, ks, dsub))
    for s in range(m):
        pq[s] = kmeans(base[:, s * dsub:(s + 1) * dsub], ks, iters=iters, seed=seed + s)
    queries = rng.standard_normal((nq, D))
    return dict(base=base, pq=pq, queries=queries)


def has_topk_tie(dists_sorted_kplus1):
    """True when any two of the first k+1 ascending distances are exactly equal."""
    d = np.asarray(dists_sorted_kplus1)
    return bool(np.any(d[1:] == d[:-1]))
