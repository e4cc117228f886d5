"""Seeded synthetic datasets of the shapes BASELINE.json names (there is no network for SIFT/DEEP).

sift_like: Gaussian mixture mapped and rounded to integers in [0, 218] -- SIFT1M's value range, so squared
distances are integers and exact float ties occur as they do on the real data.
"""
import numpy as np


def sift_like(n, d=128, n_queries=10000, n_clusters=1024, seed=2024, within=0.5, query_seed=None):
    """Returns (vectors f32[n,d], queries f32[n_queries,d]); queries come from the same mixture, disjoint stream."""
    rs = np.random.RandomState(seed)
    cent = rs.randn(n_clusters, d).astype(np.float32)

    def draw(cnt, r):
        out = np.empty((cnt, d), dtype=np.float32)
        step = 1 << 18
        for s in range(0, cnt, step):
            e = min(cnt, s + step)
            a = r.randint(0, n_clusters, size=e - s)
            p = cent[a] + within * r.randn(e - s, d).astype(np.float32)
            p = (p + 4.0) * (218.0 / 8.0)
            out[s:e] = np.clip(np.rint(p), 0, 218)
        return out

    x = draw(n, rs)
    q = draw(n_queries, np.random.RandomState(seed + 1 if query_seed is None else query_seed))
    return x, q


def unit_mixture(n, d=1536, n_queries=1000, n_clusters=256, seed=7, within=1.0):
    """Unit-norm clustered vectors (text-embedding-like); L2^2 = 2 - 2*IP on these."""
    rs = np.random.RandomState(seed)
    cent = rs.randn(n_clusters, d).astype(np.float32)

    def draw(cnt, r):
        a = r.randint(0, n_clusters, size=cnt)
        p = cent[a] + within * r.randn(cnt, d).astype(np.float32)
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        return p.astype(np.float32)

    return draw(n, rs), draw(n_queries, np.random.RandomState(seed + 1))


def recall_at_k(ids, gt, k=10):
    """mean |pred[:k] & gt[:k]| / k (dataset_benchmark.py:120-124)."""
    hit = 0
    for a, b in zip(ids[:, :k], gt[:, :k]):
        hit += len(set(a.tolist()) & set(b.tolist()))
    return hit / (k * len(ids))
