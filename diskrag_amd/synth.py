"""Seeded synthetic datasets of the shapes BASELINE.json names (there is no network for SIFT/DEEP).

sift_like: low-intrinsic-dimension Gaussian mixture mapped and rounded to integers in [0, 218] -- SIFT1M's value
range, so squared distances are integers and exact float ties occur as they do on the real data.
"""
import numpy as np


def sift_like(n, d=128, n_queries=10000, n_clusters=1024, seed=2024, latent=32, within=1.0, noise=0.2,
              query_seed=None, rounded=True, queries_only=False):
    """Returns (vectors f32[n,d], queries f32[n_queries,d]); queries come from the same mixture, disjoint stream.
    queries_only: (None, queries) -- the same queries without drawing the n vectors (their stream is their own).

    x = round(affine(B z + noise)), z ~ mixture of n_clusters Gaussians in a `latent`-dimensional space, B a fixed
    random latent x d map. Real SIFT descriptors have a local intrinsic dimension far below 128; an isotropic
    128-d mixture does not (every cluster-mate is equidistant, robust pruning cannot work and a Vamana graph
    needs L >> 100 for 0.95 recall). With latent = 32 a graph built with R = 64, L_build = 100, alpha = 1.2 has
    mean degree ~50 and a search at L = 100 scores ~4000 nodes per query, as on SIFT1M.
    """
    rs = np.random.RandomState(seed)
    B = (rs.randn(latent, d) / np.sqrt(latent)).astype(np.float32)
    cent = rs.randn(n_clusters, latent).astype(np.float32)

    def draw(cnt, r):
        out = np.empty((cnt, d), dtype=np.float32)
        step = 1 << 18
        for s in range(0, cnt, step):
            e = min(cnt, s + step)
            a = r.randint(0, n_clusters, size=e - s)
            z = cent[a] + within * r.randn(e - s, latent).astype(np.float32)
            p = z @ B + noise * r.randn(e - s, d).astype(np.float32)
            v = (p + 4.0) * (218.0 / 8.0)
            out[s:e] = np.clip(np.rint(v), 0, 218) if rounded else v       # rounded=False: SURVEY 8d's "un-rounded variant"
        return out

    x = None if queries_only else draw(n, rs)
    q = draw(n_queries, np.random.RandomState(seed + 1 if query_seed is None else query_seed))
    return x, q


def unit_mixture(n, d=1536, n_queries=1000, n_clusters=256, seed=7, latent=64, within=1.0, noise=0.05):
    """Unit-norm clustered vectors (text-embedding-like, low intrinsic dimension); L2^2 = 2 - 2*IP on these."""
    rs = np.random.RandomState(seed)
    B = (rs.randn(latent, d) / np.sqrt(latent)).astype(np.float32)
    cent = rs.randn(n_clusters, latent).astype(np.float32)

    def draw(cnt, r):
        out = np.empty((cnt, d), dtype=np.float32)
        step = 1 << 15
        for s in range(0, cnt, step):
            e = min(cnt, s + step)
            a = r.randint(0, n_clusters, size=e - s)
            z = cent[a] + within * r.randn(e - s, latent).astype(np.float32)
            p = z @ B + noise * r.randn(e - s, d).astype(np.float32)
            p /= np.linalg.norm(p, axis=1, keepdims=True)
            out[s:e] = p
        return out

    return draw(n, rs), draw(n_queries, np.random.RandomState(seed + 1))


def unit_mixture_parallel(n, d=1536, n_queries=1000, n_clusters=256, seed=7, latent=64, within=1.0, noise=0.05,
                          threads=32):
    """The same family as unit_mixture, generated chunk-parallel with counter-based streams (one Philox stream per
    32768-row chunk, float32 normals): 10M x 1536 in well under a minute on the GPU box's host cores. Not the same
    numbers as unit_mixture(seed) -- a different, equally seeded dataset."""
    from concurrent.futures import ThreadPoolExecutor
    root = np.random.SeedSequence(seed)
    g0 = np.random.Generator(np.random.Philox(root.spawn(1)[0]))
    B = (g0.standard_normal((latent, d), dtype=np.float32) / np.float32(np.sqrt(latent)))
    cent = g0.standard_normal((n_clusters, latent), dtype=np.float32)
    step = 1 << 15

    def draw(cnt, ss):
        out = np.empty((cnt, d), dtype=np.float32)
        starts = list(range(0, cnt, step))
        seeds = ss.spawn(len(starts))

        def fill(a):
            s0, sq = a
            e = min(cnt, s0 + step)
            r = np.random.Generator(np.random.Philox(sq))
            idx = r.integers(0, n_clusters, size=e - s0)
            z = cent[idx] + np.float32(within) * r.standard_normal((e - s0, latent), dtype=np.float32)
            p = z @ B
            p += np.float32(noise) * r.standard_normal((e - s0, d), dtype=np.float32)
            p /= np.linalg.norm(p, axis=1, keepdims=True)
            out[s0:e] = p

        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(fill, zip(starts, seeds)))
        return out

    kids = root.spawn(3)
    return draw(n, kids[1]), draw(n_queries, kids[2])


def recall_at_k(ids, gt, k=10):
    """mean |pred[:k] & gt[:k]| / k (dataset_benchmark.py:120-124)."""
    hit = 0
    for a, b in zip(ids[:, :k], gt[:, :k]):
        hit += len(set(a.tolist()) & set(b.tolist()))
    return hit / (k * len(ids))


def recall_at_k_ties(dist, gt_dist, k=10):
    """Distance-based recall@k (what ANN benchmarks use when distances tie): a returned entry is a hit when its distance
    is <= the k-th ground-truth distance. With PQ-only shards many points share a code word, hence an ADC distance, and
    the id-based recall_at_k charges a search for returning a DIFFERENT member of a tie than the brute force picked
    (smallest ids). dist[nq, >=k] as returned by the search (NaN padded), gt_dist[nq, >=k] in the SAME arithmetic."""
    d = np.asarray(dist)[:, :k]
    thr = np.asarray(gt_dist)[:, k - 1:k]
    hit = (d <= thr) & ~np.isnan(d)
    return float(hit.sum()) / (k * len(d))


class UnitMixtureStream:
    """The unit_mixture family as a row-addressable stream: rows [start, start + rows) are a pure function of
    (seed, row block), so a dataset far larger than host memory (config c5: 1.25e8 x 1536 per shard) can be generated,
    encoded and forgotten chunk by chunk, and any chunk can be regenerated later. One Philox stream per 32768-row block."""

    BLOCK = 1 << 15

    def __init__(self, d=1536, n_clusters=4096, seed=7, latent=64, within=1.0, noise=0.05, threads=32):
        self.d, self.n_clusters, self.within, self.noise, self.threads = d, n_clusters, within, noise, threads
        self.root = np.random.SeedSequence(seed)
        g0 = np.random.Generator(np.random.Philox(self.root.spawn(1)[0]))
        self.B = (g0.standard_normal((latent, d), dtype=np.float32) / np.float32(np.sqrt(latent)))
        self.cent = g0.standard_normal((n_clusters, latent), dtype=np.float32)
        self.latent = latent

    def _block(self, stream, b, rows):
        # always the whole block (then cut): a row's value must not depend on how the caller chunks the stream
        r = np.random.Generator(np.random.Philox(np.random.SeedSequence([self.root.entropy, stream, b])))
        idx = r.integers(0, self.n_clusters, size=self.BLOCK)
        z = self.cent[idx] + np.float32(self.within) * r.standard_normal((self.BLOCK, self.latent), dtype=np.float32)
        p = z @ self.B
        p += np.float32(self.noise) * r.standard_normal((self.BLOCK, self.d), dtype=np.float32)
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        return p[:rows]

    def draw(self, start, rows, stream=0):
        """rows [start, start + rows) of stream `stream` (0 = data, 1 = queries); start must be a multiple of BLOCK."""
        from concurrent.futures import ThreadPoolExecutor
        assert start % self.BLOCK == 0
        out = np.empty((rows, self.d), dtype=np.float32)
        blocks = list(range(0, rows, self.BLOCK))

        def fill(o):
            n = min(self.BLOCK, rows - o)
            out[o:o + n] = self._block(stream, (start + o) // self.BLOCK, n)

        with ThreadPoolExecutor(max_workers=self.threads) as ex:
            list(ex.map(fill, blocks))
        return out
