"""A/B helper (GPU box): reference-faithful M1 on LIVE data (unit-norm mixtures: the rerank policy counts on every row) -- resident batches of 4096 queries,
D = 96 (c4-shaped, m = 16) and D = 1536 (c3-shaped, m = 32), L = 100 / 250, both band policies. One JSON line per point; run once per library (DR_LIB)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture
for (N, D, m) in ((1000000, 96, 16), (200000, 1536, 32)):
    x, q = unit_mixture(N, D, n_queries=4096, n_clusters=512, seed=5, latent=32)
    ix = HipIndex.create_empty(x, R=64)
    ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
    cb = ix.pq_train(m, n_sample=50000, iters=5); ix.pq_encode(cb)
    for (L, bw, pol) in ((100, 8, 0), (250, 64, 0), (100, 8, 1)):
        ix.batch_upload(q)
        for _ in range(2): ix.batch_run(10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
        ix.batch_sync()
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            for _ in range(5): ix.batch_run(10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
            ix.batch_sync()
            best = min(best, (time.perf_counter() - t0) / 5)
        ids, dist, cnt, st = ix.batch_download()
        print(json.dumps({"D": D, "L": L, "beam_width": bw, "policy": pol, "ms_per_4096": round(best * 1e3, 3), "kernel_ms": round(ix.timing()["search_kernel_ms"], 3), "variant": ix.timing()["variant"],
                          "exact_per_query": round(float(st["exact"].mean()), 1), "sum_ids": int(ids.astype(np.uint64).sum())}), flush=True)
    ix.close()
