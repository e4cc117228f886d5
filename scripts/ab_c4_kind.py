"""A/B (GPU box): the rerank-policy-live M1 at the c4 shape with LONG lists (the recall >= 0.95 operating points need L = 400-500):
per-query table in LDS (variant 0, 6 wavefronts per CU at this list size) against the shared codebook (variant 3, 8 wavefronts per CU,
table entries recomputed per neighbour), which the engine picks today because 8 tables do not fit. usage: ab_c4_kind.py N [D m]"""
import json
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel
n = int(sys.argv[1])
D, m, ncl, latent, R = 96, 16, 4096, 32, 64
if len(sys.argv) > 3:      # another unit-norm shape: D m   (e.g. 128 32: does the table at FOUR wavefronts per CU beat the codebook at eight?)
    D, m = int(sys.argv[2]), int(sys.argv[3])
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=10000, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=R)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(m, n_sample=100000, iters=5); ix.pq_encode(cb)
gt, _ = ix.bruteforce_topk(q[:1000], 10)
ix.batch_upload(q)
ix.batch_run(10, L=100, beam_width=8, mode=_ffi.MODE_M1); ix.batch_sync()      # regime probe
for rep in range(2):
    for (L, bw) in ((500, 0), (400, 0), (300, 0), (100, 8)):
        for kind in (3, 0, -1):
            ix.debug_force_kind(kind)
            ix.batch_run(10, L=L, beam_width=bw, mode=_ffi.MODE_M1); ix.batch_sync()
            t0 = time.perf_counter()
            for _ in range(3): ix.batch_run(10, L=L, beam_width=bw, mode=_ffi.MODE_M1)
            ix.batch_sync()
            dt = (time.perf_counter() - t0) / 3
            ids, dist, cnt, st = ix.batch_download()
            t = ix.timing()
            rec = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids[:1000], gt)]))
            print(json.dumps({"N": n, "L": L, "bw": bw, "forced": kind, "variant": t["variant"], "waves_per_cu": t["waves_per_cu"], "kernel_ms": t["search_kernel_ms"],
                              "table_kernel_ms": t["lut_kernel_ms"], "qps": 10000 / dt, "recall": rec, "exact": float(st["exact"].mean()), "pq_eval": float(st["pq_evaluated"].mean()),
                              "steps": float(st["steps"].mean())}), flush=True)
ix.debug_force_kind(-1)
# the engine's PQ traversal + rerank at the same shape: shared codebook (variant 5, the engine's pick at D <= 128) against the per-query table (variant 2)
for rep in range(2):
    for (L, bw) in ((400, 32), (350, 0), (100, 8)):
        for kind in (5, 2, -1):
            ix.debug_force_kind(kind)
            kw = dict(L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
            ix.batch_run(10, **kw); ix.batch_sync()
            t0 = time.perf_counter()
            for _ in range(3): ix.batch_run(10, **kw)
            ix.batch_sync()
            dt = (time.perf_counter() - t0) / 3
            ids, dist, cnt, st = ix.batch_download()
            t = ix.timing()
            rec = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids[:1000], gt)]))
            print(json.dumps({"N": n, "mode": "PQ+rerank", "L": L, "bw": bw, "forced": kind, "variant": t["variant"], "waves_per_cu": t["waves_per_cu"], "kernel_ms": t["search_kernel_ms"],
                              "table_kernel_ms": t["lut_kernel_ms"], "qps": 10000 / dt, "recall": rec, "pq_eval": float(st["pq_evaluated"].mean()), "steps": float(st["steps"].mean())}), flush=True)
ix.debug_force_kind(-1)
