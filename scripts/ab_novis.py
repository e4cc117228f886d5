"""A/B (GPU box): the engine's ADC traversals with and without a visited set (round 4: DR_F_NO_VISITED_SET; + DR_PQ_ROW_PREFETCH=1: the next row's ids landed in LDS),
interleaved in ONE process on ONE index, each with and without inline neighbour codes (dr_index_inline_codes). usage: ab_novis.py c5s|c3|c4 N  -> prints one JSON line per run, checksums must agree"""
import hashlib
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel
shape, n = sys.argv[1], int(sys.argv[2])
D, m, ncl, latent, R, Lb = {"c3": (1536, 32, 4096, 64, 64, 100), "c4": (96, 16, 4096, 32, 64, 100), "c5s": (1536, 32, 4096, 64, 32, 64)}[shape]
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=10000, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=R)
ix.build_vamana(L_build=Lb, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(m, n_sample=100000, iters=5); ix.pq_encode(cb)
gt, _ = ix.bruteforce_topk(q[:1000], 10)
ix.batch_upload(q)
runs = [("PQ L=100 bw=8", dict(L=100, beam_width=8, mode=_ffi.MODE_PQ)), ("PQ L=200 no trim", dict(L=200, beam_width=0, mode=_ffi.MODE_PQ)),
        ("PQ+rerank L=250 no trim", dict(L=250, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ L=400 no trim", dict(L=400, beam_width=0, mode=_ffi.MODE_PQ))]
for rep in range(2):
    for tag, kw0 in runs:
        for vis, inline, pre in ((1, 0, 0), (0, 0, 0), (0, 0, 1), (1, 1, 0), (0, 1, 0)):
            kw = dict(kw0, flags=kw0.get("flags", 0) | (0 if vis else _ffi.F_NO_VISITED_SET))
            if pre: os.environ["DR_PQ_ROW_PREFETCH"] = "1"
            else: os.environ.pop("DR_PQ_ROW_PREFETCH", None)
            ix.inline_codes(bool(inline))
            ix.batch_run(10, **kw); ix.batch_sync()
            t0 = time.perf_counter()
            for _ in range(3): ix.batch_run(10, **kw)
            ix.batch_sync()
            dt = (time.perf_counter() - t0) / 3
            ids, dist, cnt, st = ix.batch_download()
            t = ix.timing()
            rec = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids[:1000], gt)]))
            print(json.dumps({"shape": shape, "N": n, "run": tag, "visited_set": bool(vis), "inline_codes": bool(inline), "next_row_prefetch": bool(pre), "prefetch_hits": float(st["adj_prefetch_hits"].mean()), "kernel_ms": t["search_kernel_ms"], "table_kernel_ms": t["lut_kernel_ms"],
                              "qps": 10000 / dt, "variant": t["variant"], "waves_per_cu": t["waves_per_cu"], "recall_vs_exact": rec, "steps": float(st["steps"].mean()),
                              "pq": float(st["pq"].mean()), "results_sha1": hashlib.sha1(ids.tobytes() + dist.tobytes()).hexdigest()[:12]}), flush=True)
