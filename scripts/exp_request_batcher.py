"""Measurement (GPU box): T client threads asking ONE query per request (the /search route's shape) on the bench index --
directly (every request its own dr_search_batch call; the handle serialises them), through RequestBatcher, and (round 4) as
submit + wait per request: dr_search_wait does not hold the handle, so concurrent requests are coalesced into shared launches
by the library itself.
usage: exp_request_batcher.py [k L]  -> gpurun_out/request_batcher_k<k>_L<L>.json (default k 10, L 100; round 5 also at the API defaults 5 20)"""
import json
import sys
import threading
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.batching import RequestBatcher
from diskrag_amd.synth import sift_like

K_ = int(sys.argv[1]) if len(sys.argv) > 2 else 10
L_ = int(sys.argv[2]) if len(sys.argv) > 2 else 100
x, q = sift_like(1000000, 128, n_queries=20000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)


class Eng:      # the facade's search_batch over a bare handle
    def search_batch(self, qs, k=10, L=None, beam_width=8, use_pq_search=True, band_policy=0):
        return ix.search_batch(qs, k, L=L, beam_width=beam_width or 0, mode=_ffi.MODE_M1)


def run(threads, per_thread, fn):
    lat = [[] for _ in range(threads)]
    def client(t):
        for i in range(per_thread):
            t0 = time.perf_counter(); fn(q[(t * per_thread + i) % len(q)]); lat[t].append(time.perf_counter() - t0)
    th = [threading.Thread(target=client, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    l = np.concatenate(lat) * 1e3
    return {"requests_per_s": threads * per_thread / dt, "p50_ms": float(np.percentile(l, 50)), "p99_ms": float(np.percentile(l, 99))}


out = {"index": "1M x 128, R 64, m 32 (the bench index)", "request": "M1, k %d, L %d, beam_width 8" % (K_, L_), "runs": {}}
direct = lambda v: ix.search_batch(v, K_, L=L_, beam_width=8, mode=_ffi.MODE_M1)
for v in q[:64]: direct(v)
for T in (1, 16, 64, 256):
    s0 = ix.pipeline_stats()
    rec = {"direct": run(T, max(20, 2000 // T), direct),
           "submit_wait_per_request": run(T, max(20, 4000 // T), lambda v: ix.search_submit(v.reshape(1, -1), K_, L=L_, beam_width=8, mode=_ffi.MODE_M1).wait())}
    s1 = ix.pipeline_stats()
    rec["submit_wait_per_request"]["requests_per_launch"] = (s1["tickets"] - s0["tickets"]) / max(1, s1["launches"] - s0["launches"])
    for wait in (0.0, 0.2, 1.0):
        with RequestBatcher(Eng(), k_max=K_, L=L_, beam_width=8, max_batch=4096, max_wait_ms=wait) as rb:
            r = run(T, max(20, 4000 // T), lambda v: rb.search(v))
            r["mean_batch"] = rb.queries_sent / max(rb.batches_sent, 1)
        rec[f"batcher_wait_{wait}ms"] = r
    out["runs"][f"{T}_threads"] = rec
    print(T, rec, flush=True)
import os
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/request_batcher_k%d_L%d.json" % (K_, L_), "w"), indent=1)
