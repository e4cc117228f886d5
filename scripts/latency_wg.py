"""Latency of small blocking calls: the workgroup-per-query kernel (variant 18, csrc/latency_kernel.hpp: DR_LAT_ALL=1) beside the batch kernels
(the engine's choice) with and without "ask later" (DR_NO_ASK_LATER=1: round 4's behaviour), interleaved, 1 / 8 / 64 queries per dr_search_batch on the 1M-point bench index.
Reference-faithful M1 at the API defaults (k 5, L 20, beam_width 8) and at the bench point (k 10, L 100), M2 at beam_width 8 (the CLI's other
search). Also checks that both give the same ids / distance bits / counters. -> one JSON object.  argv[1]: points (default 1000000)"""
import ctypes as C
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
x, q = sift_like(N, 128, n_queries=4096, n_clusters=1024, seed=2024, query_seed=9000)
if len(sys.argv) > 2 and sys.argv[2] == "float":      # un-rounded rows: the float-row kernels
    x = x + np.float32(0.25); q = q + np.float32(0.25)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
L_ = _ffi.load_library()
out = {}
pts = (("M1_api_default_k5_L20_bw8", dict(k=5, L=20, bw=8, mode=_ffi.MODE_M1)), ("M1_k10_L100_bw8", dict(k=10, L=100, bw=8, mode=_ffi.MODE_M1)),
       ("M2_k10_bw8", dict(k=8, L=100, bw=8, mode=_ffi.MODE_M2)))
if os.environ.get("LAT_SWEEP"):      # where the workgroup kernel stops paying: list sizes between the API default and the bench point
    pts = tuple(("M1_k10_L%d_bw8" % L, dict(k=10, L=L, bw=8, mode=_ffi.MODE_M1)) for L in (10, 32, 48, 64, 80, 128, 200))
LEGS = (("workgroup_per_query", {"DR_LAT_ALL": "1"}), ("batch_kernels", {"DR_NO_LATENCY": "1"}),
        ("batch_kernels_policy_asked_first", {"DR_NO_LATENCY": "1", "DR_NO_ASK_LATER": "1"}))      # (the last one: round 4's behaviour at short lists)
def set_leg(env):
    for k in ("DR_LAT_ALL", "DR_NO_LATENCY", "DR_NO_ASK_LATER"): os.environ.pop(k, None)
    os.environ.update(env)
qq = np.ascontiguousarray(q, dtype=np.float32)
NQS = tuple(int(v) for v in os.environ.get("LAT_NQ", "1,8,64").split(","))
if max(NQS) > 64: ix.debug_force_kind(-1)
for tag, kw in pts:
    for nq in NQS:
        k = kw["k"]
        res = {}
        # same answers first
        for name, env in LEGS:
            set_leg(env)
            ix.debug_force_kind(18 if (nq > 256 and name == "workgroup_per_query") else -1)
            for _ in range(3):
                r = ix.search_batch(qq[:nq], k, L=kw["L"], beam_width=kw["bw"], mode=kw["mode"])
            res[name] = (r, ix.timing()["variant"])
        a = res["workgroup_per_query"][0]
        same = all(bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and
                        all(np.array_equal(a[3][f], b[3][f]) for f in ("steps", "visited", "exact", "pq", "status", "inserts")))
                   for b in (res["batch_kernels"][0], res["batch_kernels_policy_asked_first"][0]))
        oi = np.empty((nq, k), np.uint32); od = np.empty((nq, k), np.float32); oc = np.empty(nq, np.uint32)
        pi, pd, pc = oi.ctypes.data_as(C.POINTER(C.c_uint32)), od.ctypes.data_as(C.POINTER(C.c_float)), oc.ctypes.data_as(C.POINTER(C.c_uint32))
        ts = {n: [] for n, _ in LEGS}
        ks = {n: [] for n, _ in LEGS}
        var = {}
        for blk in range(6):       # interleaved blocks of 100 calls
            for name, env in LEGS:
                set_leg(env)
                ix.debug_force_kind(18 if (nq > 256 and name == "workgroup_per_query") else -1)
                for i in range(110):
                    pq_ = qq[((blk * 110 + i) * nq) % (4096 - nq):].ctypes.data_as(C.POINTER(C.c_float))
                    t0 = time.perf_counter()
                    rc = L_.dr_search_batch(ix._h, pq_, nq, k, kw["L"], kw["bw"], kw["mode"], 0, 0, pi, pd, pc, None)
                    t1 = time.perf_counter()
                    assert rc == 0
                    if i >= 10:
                        ts[name].append(t1 - t0)
                        if i % 20 == 0: ks[name].append(ix.timing()["search_kernel_ms"])
                var[name] = ix.timing()["variant"]
        set_leg({})
        ix.debug_force_kind(-1)
        ent = {"same_results": same, "mean_expansions": round(float(a[3]["steps"].mean()), 1), "mean_rounds_hits": round(float(a[3]["adj_prefetch_hits"].mean()), 1)}
        for name in ts:
            t = np.array(ts[name]) * 1e3
            ent[name] = {"p50_ms": round(float(np.percentile(t, 50)), 4), "p99_ms": round(float(np.percentile(t, 99)), 4),
                         "search_kernel_ms_mean": round(float(np.mean(ks[name])), 4), "variant": var[name]}
        out.setdefault(tag, {})["nq%d" % nq] = ent
print(json.dumps(out, indent=1))
