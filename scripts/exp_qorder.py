"""Experiment (GPU box): does the ORDER of the queries inside a batch matter? Queries that share graph regions, run by
neighbouring wavefront slots at the same time, can hit each other's rows in L2 / Infinity Cache."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
def measure(qq, bw, tag):
    ix.batch_upload(qq)
    for _ in range(3): ix.batch_run(10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
    ix.batch_sync()
    ts = []
    for _ in range(10):
        ix.batch_run(10, L=100, beam_width=bw, mode=_ffi.MODE_M1); ix.batch_sync(); ts.append(ix.timing()["search_kernel_ms"])
    print(f"{tag:28s} bw={bw} kernel_ms min {min(ts):.3f} med {sorted(ts)[5]:.3f}")
rs = np.random.RandomState(1)
for npiv in (256, 1024, 4096):
    piv = x[rs.choice(len(x), npiv, replace=False)]
    d = (q * q).sum(1)[:, None] - 2.0 * q @ piv.T + (piv * piv).sum(1)[None, :]
    key = d.argmin(1)
    order = np.argsort(key, kind="stable")
    for bw in (8, 0):
        measure(q, bw, "original order")
        measure(q[order], bw, f"sorted by nearest of {npiv}")
measure(q[rs.permutation(len(q))], 8, "random permutation")
