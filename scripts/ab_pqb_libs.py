"""A/B (GPU box): DR_MODE_PQB of several LIBRARY BUILDS on ONE index, interleaved by the calling shell (one process per library: DR_LIB is read when
the library is loaded). `prep` builds the index once and saves what a PQ-only traversal needs (graph, code words, codebook, queries; the rows too for
the shapes that rerank); `run` loads it under the library DR_LIB names and times the same points, printing one JSON line per point with a checksum of
the results (the builds must agree bit for bit).
usage: ab_pqb_libs.py prep c5s|c3|c4|c5w N DIR
       DR_LIB=... ab_pqb_libs.py run DIR TAG [reps]"""
import hashlib
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi           # noqa: E402

SHAPES = {"c3": (1536, 32, 4096, 64, 64, 100), "c4": (96, 16, 4096, 32, 64, 100), "c5s": (1536, 32, 4096, 64, 32, 64), "c5w": (1536, 32, 4096, 64, 128, 128)}
POINTS = {"c5s": [(100, 8), (200, 0)], "c5w": [(100, 32), (150, 16)], "c3": [(250, 0), (100, 8)], "c4": [(400, 32), (100, 8)]}

if sys.argv[1] == "prep":
    from diskrag_amd.synth import unit_mixture, unit_mixture_parallel
    shape, n, d = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    D, m, ncl, latent, R, Lb = SHAPES[shape]
    os.makedirs(d, exist_ok=True)
    gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
    x, q = gen(n, D, n_queries=10000, n_clusters=ncl, seed=11, latent=latent)
    ix = HipIndex.create_empty(x, R=R)
    t0 = time.perf_counter()
    if shape == "c5w":
        cb = ix.pq_train(m, n_sample=100000, iters=5); codes = ix.pq_encode(cb, want_codes=True)
        ix.build_vamana_pq(L_build=Lb, alpha=1.2, passes=2, seed=7)
    else:
        ix.build_vamana(L_build=Lb, alpha=1.2, passes=2, seed=7)
        cb = ix.pq_train(m, n_sample=100000, iters=5); codes = ix.pq_encode(cb, want_codes=True)
    np.save(os.path.join(d, "adj.npy"), ix.get_adjacency()); np.save(os.path.join(d, "codes.npy"), codes); np.save(os.path.join(d, "cb.npy"), cb)
    np.save(os.path.join(d, "q.npy"), q)
    gt, _ = ix.bruteforce_topk(q[:1000], 10)
    gta, _, _ = ix.pq_scan_topk(q[:1000], 10)
    np.save(os.path.join(d, "gt.npy"), gt); np.save(os.path.join(d, "gta.npy"), gta)
    if shape in ("c3", "c4"):
        np.save(os.path.join(d, "x.npy"), x)
    json.dump({"shape": shape, "N": n, "D": D, "m": m, "R": R, "medoid": int(ix.medoid), "build_s": time.perf_counter() - t0}, open(os.path.join(d, "meta.json"), "w"))
    print(open(os.path.join(d, "meta.json")).read(), flush=True)
    sys.exit(0)

d, tag = sys.argv[2], sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
meta = json.load(open(os.path.join(d, "meta.json")))
shape = meta["shape"]
adj, codes, cb, q = (np.load(os.path.join(d, f)) for f in ("adj.npy", "codes.npy", "cb.npy", "q.npy"))
gt, gta = np.load(os.path.join(d, "gt.npy")), np.load(os.path.join(d, "gta.npy"))
rr = 0
if shape in ("c3", "c4"):
    ix = HipIndex.create(np.load(os.path.join(d, "x.npy"), mmap_mode="r"), adj, meta["medoid"])
    ix.set_pq(cb, codes)
    rr = _ffi.F_RERANK
else:
    ix = HipIndex.create_codes(adj, meta["medoid"], meta["D"], cb, codes)
ix.batch_upload(q)
extra = [e for e in os.environ.get("AB_POPS", "").split(",") if e]
for L, bw in POINTS[shape]:
    for pops in [0] + [int(e) for e in extra]:
        kw = dict(mode=_ffi.MODE_PQB, L=L, beam_width=bw, flags=rr | (_ffi.F_POPS(pops) if pops else 0))
        ix.batch_run(10, **kw); ix.batch_sync()
        ker, lut, wall = [], [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            ix.batch_run(10, **kw); ix.batch_sync()
            wall.append(time.perf_counter() - t0)
            t = ix.timing()
            ker.append(t["search_kernel_ms"]); lut.append(t["lut_kernel_ms"])
        ids, dist, cnt, st = ix.batch_download()
        t = ix.timing()
        rec = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids[:1000], gt)]))
        reca = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids[:1000], gta)]))
        print(json.dumps({"lib": tag, "shape": shape, "N": meta["N"], "L": L, "bw": bw, "pops": pops, "kernel_ms": round(float(np.median(ker)), 4),
                          "kernel_ms_min": round(float(np.min(ker)), 4), "table_kernel_ms": round(float(np.median(lut)), 4), "call_ms": round(1e3 * float(np.median(wall)), 4),
                          "waves_per_cu": t["waves_per_cu"], "lds": t["lds_bytes"], "recall_vs_exact": round(rec, 4), "recall_vs_adc": round(reca, 4),
                          "steps": float(st["steps"].mean()), "pq": float(st["pq"].mean()), "status": int(st["status"].max()),
                          "results_sha1": hashlib.sha1(ids.tobytes() + dist.tobytes()).hexdigest()[:12]}), flush=True)
