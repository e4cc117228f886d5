"""Latency of small calls through the C ABI (GPU box): 1 / 8 / 64 queries per blocking dr_search_batch on the bench index (the /search and
/faq-search routes ask one query at a time, search_engine.py:530-614, app.py:84-130), reference-faithful M1 at the API defaults (k 5, L 20,
beam_width 8) and at the bench point (k 10, L 100), and the engine's PQ traversal + rerank (DR_MODE_PQB). Host buffers in, results out.
DR_REPREFETCH=1 (the adjacency prefetch with a second chance, 97 % hits) is tried beside the default. -> one JSON object"""
import ctypes as C
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=4096, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
L_ = _ffi.load_library()
out = {}
pts = (("M1_api_default_k5_L20_bw8", dict(k=5, L=20, bw=8, mode=_ffi.MODE_M1, flags=0)), ("M1_k10_L100_bw8", dict(k=10, L=100, bw=8, mode=_ffi.MODE_M1, flags=0)),
       ("PQB_rerank_k10_L100_bw8", dict(k=10, L=100, bw=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK)))
for env in ("0", "1", "k3", "k17"):
    if env == "1": os.environ["DR_REPREFETCH"] = "1"
    else: os.environ.pop("DR_REPREFETCH", None)
    ix.debug_force_kind(int(env[1:]) if env.startswith("k") else -1)
    for tag, kw in pts:
        if env.startswith("k") and kw["mode"] != _ffi.MODE_M1: continue
        for nq in (1, 8, 64):
            k = kw["k"]
            qq = np.ascontiguousarray(q, dtype=np.float32)
            oi = np.empty((nq, k), np.uint32); od = np.empty((nq, k), np.float32); oc = np.empty(nq, np.uint32)
            pi, pd, pc = oi.ctypes.data_as(C.POINTER(C.c_uint32)), od.ctypes.data_as(C.POINTER(C.c_float)), oc.ctypes.data_as(C.POINTER(C.c_uint32))
            ts, ks = [], []
            nrep = 600
            for i in range(nrep + 50):
                pq_ = qq[(i * nq) % (4096 - nq):].ctypes.data_as(C.POINTER(C.c_float))
                t0 = time.perf_counter()
                rc = L_.dr_search_batch(ix._h, pq_, nq, k, kw["L"], kw["bw"], kw["mode"], 0, kw["flags"], pi, pd, pc, None)
                t1 = time.perf_counter()
                assert rc == 0
                if i >= 50:
                    ts.append(t1 - t0)
                    if i % 20 == 0: ks.append(ix.timing()["search_kernel_ms"])
            ts = np.array(ts) * 1e3
            out.setdefault(tag, {})["nq%d%s" % (nq, "_second_chance_prefetch" if env == "1" else "_forced_variant_" + env[1:] if env.startswith("k") else "")] = {
                "p50_ms": round(float(np.percentile(ts, 50)), 4), "p99_ms": round(float(np.percentile(ts, 99)), 4), "search_kernel_ms_mean": round(float(np.mean(ks)), 4),
                "variant": ix.timing()["variant"]}
print(json.dumps(out, indent=1))
