"""Hit rate of the adjacency prefetch (stats.adj_prefetch_hits counts expansions whose row was already in LDS)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=100000, iters=8); ix.pq_encode(cb)
for bw in (8, 0):
    ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
    print("bw", bw, "variant", ix.timing()["variant"], "expansions", st["steps"].mean(), "prefetch hits", st["adj_prefetch_hits"].mean(),
          "rate", st["adj_prefetch_hits"].sum() / st["steps"].sum())
