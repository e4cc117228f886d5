"""A/B helper (GPU box): resident-batch throughput of small and mid-size batches (64 ... 3000 queries: the 4-wavefront workgroup variants 16 / 17) on the
1M bench index at L = 100 and at the API's L = 20. -> JSON lines (compare two builds / switches by running it twice)"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=4096, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
for (k, L) in ((10, 100), (5, 20)):
    for nq in (64, 256, 1000, 3000):
        ix.batch_upload(q[:nq])
        for _ in range(3): ix.batch_run(k, L=L, beam_width=8, mode=_ffi.MODE_M1)
        ix.batch_sync()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(20): ix.batch_run(k, L=L, beam_width=8, mode=_ffi.MODE_M1)
            ix.batch_sync()
            best = min(best, (time.perf_counter() - t0) / 20)
        print(json.dumps({"k": k, "L": L, "queries": nq, "ms_per_batch_best": round(best * 1e3, 4), "kernel_ms": round(ix.timing()["search_kernel_ms"], 4), "variant": ix.timing()["variant"]}), flush=True)
