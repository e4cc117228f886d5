import sys, numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like, recall_at_k
x, q = sift_like(1000000, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
gt, _ = ix.bruteforce_topk(q, 10)
ix.batch_upload(q)
for kind in (8, 12, -1):
    ix.debug_force_kind(kind)
    for mode, kw in ((2, dict(L=100, beam_width=8)), (2, dict(L=100, beam_width=64)), (4, dict(L=100, beam_width=0))):
        for _ in range(2): ix.batch_run(10, mode=mode, **kw)
        ix.batch_sync()
        for _ in range(5): ix.batch_run(10, mode=mode, **kw)
        ix.batch_sync(); t = ix.timing()
        ids, dist, cnt, st = ix.batch_download()
        print("kind", kind, "mode", mode, kw, "kernel_ms %.3f" % t["search_kernel_ms"], "block", t["block"], "recall %.4f" % recall_at_k(ids, gt, 10))
