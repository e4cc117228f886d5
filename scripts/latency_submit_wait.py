"""One-query requests through dr_search_submit + dr_search_wait (what the facade's _pq_accelerated_graph_search does, search_engine.py _one):
p50 / p99 per request at the API defaults (k 5, L 20, beam_width 8) on the 1M-point bench index, variant 18 (DR_LAT_ALL=1) beside the engine's
choice (the one-wavefront kernels), interleaved; and 16 request threads sharing launches. -> one JSON object"""
import json
import os
import sys
import threading
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=4096, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
kw = dict(L=20, beam_width=8, mode=_ffi.MODE_M1)
for _ in range(3): ix.search_batch(q[:4], 5, **kw)
out = {}
ts = {"workgroup_per_query": [], "batch_kernels": []}
var = {}
for blk in range(6):
    for name, env in (("workgroup_per_query", None), ("batch_kernels", "1")):
        if env: os.environ["DR_NO_LATENCY"] = env; os.environ.pop("DR_LAT_ALL", None)
        else: os.environ.pop("DR_NO_LATENCY", None); os.environ["DR_LAT_ALL"] = "1"
        for i in range(110):
            qq = q[(blk * 110 + i) % 4000:][:1]
            t0 = time.perf_counter()
            ix.search_submit(qq, 5, **kw).wait()
            t1 = time.perf_counter()
            if i >= 10: ts[name].append(t1 - t0)
        var[name] = ix.timing()["variant"]
for name in ts:
    t = np.array(ts[name]) * 1e3
    out["one_request_at_a_time_" + name] = {"p50_ms": round(float(np.percentile(t, 50)), 4), "p99_ms": round(float(np.percentile(t, 99)), 4), "variant": var[name]}
# 16 request threads, one query per request
for name, env in (("workgroup_per_query", None), ("batch_kernels", "1")):
    if env: os.environ["DR_NO_LATENCY"] = env; os.environ.pop("DR_LAT_ALL", None)
    else: os.environ.pop("DR_NO_LATENCY", None); os.environ["DR_LAT_ALL"] = "1"
    lat = []
    def worker(t):
        for i in range(200):
            qq = q[(t * 200 + i) % 4000:][:1]
            t0 = time.perf_counter()
            ix.search_submit(qq, 5, **kw).wait()
            lat.append(time.perf_counter() - t0)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    a = np.array(lat) * 1e3
    out["16_request_threads_" + name] = {"requests_per_s": round(16 * 200 / wall, 1), "p50_ms": round(float(np.percentile(a, 50)), 4), "p99_ms": round(float(np.percentile(a, 99)), 4),
                                          "pipeline_since_start": ix.pipeline_stats()}
os.environ.pop("DR_NO_LATENCY", None); os.environ.pop("DR_LAT_ALL", None)
print(json.dumps(out, indent=1))
