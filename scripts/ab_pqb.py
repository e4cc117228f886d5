"""A/B (GPU box): the engine's PQ-only traversals on ONE index, interleaved in ONE process -- DR_MODE_PQ (the sequential statement: variant 15,
with and without its visited set) against DR_MODE_PQB (round 5: a batch per step on a total order, 1 / 2 / 4 frontier entries per step), with the
table layouts of the new kernel (DR_PQB_TREG) and inline neighbour codes. Recall@10 against the brute-force ADC ranking (dr_pq_scan_topk) and the
exact neighbours, 1000 queries. usage: ab_pqb.py c5s|c3|c4 N [quick]  -> one JSON line per run"""
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
quick = len(sys.argv) > 3 and sys.argv[3] == "quick"
D, m, ncl, latent, R, Lb = {"c3": (1536, 32, 4096, 64, 64, 100), "c4": (96, 16, 4096, 32, 64, 100), "c5s": (1536, 32, 4096, 64, 32, 64),
                            "c5w": (1536, 32, 4096, 64, 128, 128)}[shape]
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=10000, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=R)
t0 = time.perf_counter()
if shape == "c5w":      # the c5 shard's own construction: graph from code words, R = 128
    cb = ix.pq_train(m, n_sample=100000, iters=5); ix.pq_encode(cb)
    ix.build_vamana_pq(L_build=Lb, alpha=1.2, passes=2, seed=7)
else:
    ix.build_vamana(L_build=Lb, alpha=1.2, passes=2, seed=7)
    cb = ix.pq_train(m, n_sample=100000, iters=5); ix.pq_encode(cb)
print(json.dumps({"shape": shape, "N": n, "D": D, "m": m, "R": R, "build_s": time.perf_counter() - t0}), flush=True)
gt, _ = ix.bruteforce_topk(q[:1000], 10)
gta, _, _ = ix.pq_scan_topk(q[:1000], 10)
ix.batch_upload(q)
rr = _ffi.F_RERANK if shape in ("c3", "c4") else 0
if shape == "c5s": pts = [(100, 8), (200, 0)] if quick else [(100, 8), (200, 0), (400, 0), (100, 32)]
elif shape == "c5w": pts = [(100, 32), (150, 16)] if quick else [(100, 32), (150, 16), (100, 0), (200, 8)]
elif shape == "c3": pts = [(250, 0), (100, 8)] if quick else [(250, 0), (100, 8), (400, 0), (300, 64)]
else: pts = [(400, 32), (100, 8)] if quick else [(400, 32), (100, 8), (350, 64), (350, 0)]
variants = [("PQ", dict(mode=_ffi.MODE_PQ), {}), ("PQ_novis", dict(mode=_ffi.MODE_PQ, flags=_ffi.F_NO_VISITED_SET), {})]
for pops in (1, 2, 4):
    if pops * (1 << int(np.ceil(np.log2(R)))) <= 256:
        variants.append((f"PQB_pops{pops}", dict(mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(pops)), {}))
for pops in (2, 4):       # steps of several passes: the visited filter + compaction (default) against the plain form, and 24 register rows (spills at 4 passes)
    if 64 < pops * (1 << int(np.ceil(np.log2(R)))) <= 256:
        if m == 32: variants.append((f"PQB_pops{pops}_treg24", dict(mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(pops)), {"DR_PQB_TREG": "24"}))
best_pops = 2 if R <= 64 else 1
if m == 32:
    for treg in ("16", "24"):
        variants.append((f"PQB_pops{best_pops}_treg{treg}", dict(mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(best_pops)), {"DR_PQB_TREG": treg}))
if n * R * m <= 40e9:
    variants.append((f"PQB_pops{best_pops}_inline", dict(mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(best_pops)), {"inline": 1}))
    variants.append(("PQ_novis_inline", dict(mode=_ffi.MODE_PQ, flags=_ffi.F_NO_VISITED_SET), {"inline": 1}))
for rep in range(2):
    for L, bw in pts:
        for tag, kw0, env in variants:
            kw = dict(kw0, L=L, beam_width=bw, flags=kw0.get("flags", 0) | rr)
            for k_, v_ in env.items():
                if k_ != "inline": os.environ[k_] = v_
            try:
                ix.inline_codes(bool(env.get("inline")))
                ix.batch_run(10, **kw); ix.batch_sync()
                t0 = time.perf_counter()
                for _ in range(3): ix.batch_run(10, **kw)
                ix.batch_sync()
                dt = (time.perf_counter() - t0) / 3
                ids, dist, cnt, st = ix.batch_download()
                t = ix.timing()
                rec = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids[:1000], gt)]))
                reca = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids[:1000], gta)]))
                print(json.dumps({"shape": shape, "N": n, "L": L, "bw": bw, "run": tag, "rep": rep, "kernel_ms": round(t["search_kernel_ms"], 4), "table_kernel_ms": round(t["lut_kernel_ms"], 4),
                                  "qps": round(10000 / dt), "variant": t["variant"], "waves_per_cu": t["waves_per_cu"], "lds": t["lds_bytes"],
                                  "recall_vs_exact": round(rec, 4), "recall_vs_adc": round(reca, 4), "steps": float(st["steps"].mean()), "pq": float(st["pq"].mean()),
                                  "status": int(st["status"].max()), "results_sha1": hashlib.sha1(ids.tobytes() + dist.tobytes()).hexdigest()[:12]}), flush=True)
            except Exception as e:      # noqa: BLE001
                print(json.dumps({"shape": shape, "L": L, "bw": bw, "run": tag, "error": str(e)[:200]}), flush=True)
            for k_ in env:
                os.environ.pop(k_, None)
ix.inline_codes(False)
