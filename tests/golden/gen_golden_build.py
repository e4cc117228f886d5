#!/usr/bin/env python3
"""Golden fixture for the graph BUILDER and the PQ fit (N2), produced by running the REFERENCE
(read-only at /root/reference) in the same /tmp scratch copy gen_golden.py uses. Writes DATA ONLY:
tests/golden/build_int32.npz

  * build_vamana_index_cython (pydiskann/cython_utils.pyx:269-369) on small INTEGER-valued points, Python's RNG seeded
    (the builder shuffles with random.shuffle): the adjacency lists. Integer coordinates make every float32 summation
    order give the same bits, so the -ffast-math build of the reference is reproducible bit for bit.
  * DiskANNPQ.fit (pydiskann/pq/fast_pq.py:188-243, sklearn KMeans with k-means++ and n_init restarts) on the
    sift128 / deep96 fixture vectors: the quantisation error (sum over sub-quantisers of KMeans.inertia_) the device
    trainer is compared with.

Run (dev container only): python tests/golden/gen_golden_build.py"""
import random
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import setup_reference  # noqa: E402


def main():
    setup_reference()
    from pydiskann import cython_utils
    from pydiskann.pq.fast_pq import DiskANNPQ
    out = {}
    rs = np.random.RandomState(77)
    n, d = 500, 32
    cent = rs.randint(0, 12, size=(10, d))
    pts = (cent[rs.randint(0, 10, size=n)] + rs.randint(-2, 3, size=(n, d))).astype(np.float32)
    pts[rs.choice(n, 25, replace=False)] = pts[rs.randint(0, n, 25)]          # exact duplicates: distance ties
    for tag, (R, L, alpha, medoid, seed) in {"a": (8, 16, 1.2, 17, 123), "b": (16, 40, 1.2, 250, 7), "c": (4, 8, 1.0, 0, 99)}.items():
        random.seed(seed)
        adj = cython_utils.build_vamana_index_cython(pts, R, L, alpha, medoid, False)
        deg = np.array([len(a) for a in adj], dtype=np.uint32)
        pad = np.full((n, int(deg.max())), 0xFFFFFFFF, dtype=np.uint32)
        for i, a in enumerate(adj):
            pad[i, :len(a)] = a
        out[f"adj_{tag}"] = pad
        out[f"deg_{tag}"] = deg
        out[f"params_{tag}"] = np.array([R, L, medoid, seed], dtype=np.int64)
        out[f"alpha_{tag}"] = np.float32(alpha)
        print(tag, "mean degree", deg.mean(), "max", deg.max())
    out["points"] = pts
    # PQ fit quality of the reference on two fixture datasets
    for name, m in (("sift128", 32), ("deep96", 16)):
        x = np.load(HERE / f"data_{name}.npz")["vectors"]
        np.random.seed(5)
        pq = DiskANNPQ(m, 256)
        pq.fit(x)
        inertia = float(sum(km.inertia_ for km in pq.kmeans_list))
        codes = pq.encode(x)
        cb = np.stack([km.cluster_centers_ for km in pq.kmeans_list]).astype(np.float32)
        sd = x.shape[1] // m
        rec = np.concatenate([cb[j][codes[:, j]] for j in range(m)], axis=1)
        out[f"pqfit_{name}"] = np.array([m, inertia, float(((x - rec) ** 2).sum())], dtype=np.float64)
        print(name, "m", m, "sklearn inertia", inertia)
    np.savez_compressed(HERE / "build_int32.npz", **out)


if __name__ == "__main__":
    main()
