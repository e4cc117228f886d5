"""One-query blocking calls on an EMBEDDING-shaped index (unit-norm mixture, D = 1536 by default, m = 32: the rerank policy really consults the ADC
there): p50 through the C ABI at the API defaults (k 5, L 20, beam_width 8) and at L = 100, both band policies, variant 18 (DR_LAT_ALL=1) beside the
engine's choice, and the engine's PQ traversal + rerank (DR_MODE_PQB). usage: latency_embeddings.py [points] [D]  -> one JSON object"""
import ctypes as C
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
x, q = unit_mixture(N, D, n_queries=1024, n_clusters=256, seed=5, latent=32)
ix = HipIndex.create_empty(x, R=64)
t0 = time.time(); ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True); tb = time.time() - t0
cb = ix.pq_train(32, n_sample=50000, iters=5); ix.pq_encode(cb)
if os.environ.get("LAT_INLINE_CODES"): ix.inline_codes(True)       # dr_index_inline_codes: every adjacency slot carries its neighbour's code word
L_ = _ffi.load_library()
out = {"index": "%d x %d unit-norm mixture, R 64, m 32; built in %.1f s" % (N, D, tb)}
pts = (("M1_k5_L20_bw8_policy0", dict(k=5, L=20, bw=8, mode=_ffi.MODE_M1, pol=0, flags=0)), ("M1_k5_L20_bw8_policy1", dict(k=5, L=20, bw=8, mode=_ffi.MODE_M1, pol=1, flags=0)),
       ("M1_k10_L100_bw8_policy0", dict(k=10, L=100, bw=8, mode=_ffi.MODE_M1, pol=0, flags=0)), ("PQB_rerank_k10_L100_bw8", dict(k=10, L=100, bw=8, mode=_ffi.MODE_PQB, pol=0, flags=_ffi.F_RERANK)),
       ("M2_k8_bw8", dict(k=8, L=0, bw=8, mode=_ffi.MODE_M2, pol=0, flags=0)))
qq = np.ascontiguousarray(q, dtype=np.float32)
NQS = tuple(int(v) for v in os.environ.get("LAT_NQ", "1,16").split(","))
for tag, kw in pts:
    for nq in NQS:
        ent = {}
        legs = (("engine", {"DR_NO_LATENCY": "1"}), ("workgroup_per_query", {"DR_LAT_ALL": "1"})) if kw["mode"] in (_ffi.MODE_M1, _ffi.MODE_M2) else (("engine", {}),)
        if os.environ.get("LAT_WANTS") and kw["mode"] in (_ffi.MODE_M1, _ffi.MODE_M2):
            legs = legs + tuple(("workgroup_per_query_want%s" % w, {"DR_LAT_ALL": "1", "DR_LAT_WANT": w}) for w in os.environ["LAT_WANTS"].split(","))
        k = kw["k"]
        oi = np.empty((nq, k), np.uint32); od = np.empty((nq, k), np.float32); oc = np.empty(nq, np.uint32)
        pi, pd, pc = oi.ctypes.data_as(C.POINTER(C.c_uint32)), od.ctypes.data_as(C.POINTER(C.c_float)), oc.ctypes.data_as(C.POINTER(C.c_uint32))
        ts = {n: [] for n, _ in legs}; ks = {n: [] for n, _ in legs}; var = {}; stt = {}
        for blk in range(4):
            for name, env in legs:
                for kk in ("DR_LAT_ALL", "DR_LAT_WANT", "DR_NO_LATENCY"): os.environ.pop(kk, None)
                os.environ.update(env)
                for i in range(90):
                    pq_ = qq[((blk * 90 + i) * nq) % (1024 - nq):].ctypes.data_as(C.POINTER(C.c_float))
                    t0 = time.perf_counter()
                    rc = L_.dr_search_batch(ix._h, pq_, nq, k, kw["L"], kw["bw"], kw["mode"], kw["pol"], kw["flags"], pi, pd, pc, None)
                    t1 = time.perf_counter()
                    assert rc == 0
                    if i >= 10:
                        ts[name].append(t1 - t0)
                        if i % 20 == 0: ks[name].append(ix.timing()["search_kernel_ms"])
                var[name] = ix.timing()["variant"]
                r = ix.search_batch(qq[:nq], k, L=kw["L"], beam_width=kw["bw"], mode=kw["mode"], band_policy=kw["pol"], flags=kw["flags"])
                stt[name] = {f: round(float(r[3][f].mean()), 1) for f in ("steps", "visited", "exact", "pq", "pq_evaluated")}
        for kk in ("DR_LAT_ALL", "DR_LAT_WANT", "DR_NO_LATENCY"): os.environ.pop(kk, None)
        for name in ts:
            t = np.array(ts[name]) * 1e3
            ent[name] = {"p50_ms": round(float(np.percentile(t, 50)), 4), "p99_ms": round(float(np.percentile(t, 99)), 4), "search_kernel_ms_mean": round(float(np.mean(ks[name])), 4),
                         "variant": var[name], "per_query": stt[name]}
        out.setdefault(tag, {})["nq%d" % nq] = ent
print(json.dumps(out, indent=1))
