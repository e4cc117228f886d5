"""Diagnostic: no-trim (beam_width=None) M1 at the bench size through the resident-batch path, per batch, device vs oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like, recall_at_k
from oracle import pyoracle as orc
nb, nq = 8, 10000
x, q = sift_like(1000000, 128, n_queries=nq * nb, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
medoid, _ = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100000, iters=8)
codes = ix.pq_encode(cb, want_codes=True)
adj = ix.get_adjacency()
gt, _ = ix.bruteforce_topk(q, 10)
for b in range(nb):
    ix.batch_select(b); ix.batch_upload(q[b * nq:(b + 1) * nq])
for bw in (8, 0):
    for rep in range(2):
        if rep == 1:            # many back-to-back launches first, like the bench
            for i in range(300):
                ix.batch_select(i % nb); ix.batch_run(10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
            ix.batch_sync()
        rec = []
        for b in range(nb):
            ix.batch_select(b)
            ix.batch_run(10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
            ids, dist, cnt, st = ix.batch_download()
            sl = slice(b * nq, (b + 1) * nq)
            w = orc.search_batch(x, adj, q[sl][:300], medoid, orc.M1, 10, L=100, bw=bw, codes=codes, codebook=cb, nthreads=64)
            rec.append((round(recall_at_k(ids, gt[sl], 10), 4), bool(np.array_equal(ids[:300], w[0])), int(st["status"].max()), round(float(st["exact"].mean()), 1)))
        print("bw", bw, "after-many-launches" if rep else "fresh", rec, flush=True)
