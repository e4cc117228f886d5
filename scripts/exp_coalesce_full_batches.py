"""Experiment (GPU box): the tail of a 10k-query launch. 10 000 queries on 4096 wavefront slots are 2.4 queries per slot: towards the end of
a launch slots run out of tickets and idle until the longest query is done, and the next batch's kernel (same stream) cannot start before.
Does coalescing FULL 10k-query submits into launches of 20k-30k queries (dr_set_coalesce(32768), more tickets in flight) shorten the time per
query?  usage: exp_coalesce_full_batches.py -> JSON lines"""
import json
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like, recall_at_k

nq, nb = 10000, 8
x, q = sift_like(1000000, 128, n_queries=nq * nb, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
ix.pq_encode(ix.pq_train(32, n_sample=100000, iters=8))
qb = []
for b in range(nb):
    a = _ffi.pinned_empty((nq, 128), np.float32)
    a[:] = q[b * nq:(b + 1) * nq]
    qb.append(a)
want = [ix.search_batch(a, 10, L=100, beam_width=8, mode=_ffi.MODE_M1) for a in qb[:2]]


def run(n, depth):
    jobs, done, last = [], 0, None
    t0 = time.perf_counter()
    for i in range(n):
        jobs.append(ix.search_submit(qb[i % nb], 10, L=100, beam_width=8, mode=_ffi.MODE_M1, reuse_outputs=True))
        if len(jobs) - done >= depth:
            jobs[done].wait(); jobs[done] = None; done += 1
    for j in range(done, len(jobs)):
        last = jobs[j].wait()
    return time.perf_counter() - t0, last


for rep in range(3):
    for cap, depth in ((10240, 4), (20480, 8), (32768, 10), (32768, 14), (10240, 8)):
        ix.set_coalesce(cap)
        run(3 * depth, depth); ix.batch_sync()
        s0 = ix.pipeline_stats()
        n = 400
        el, last = run(n, depth)
        ix.batch_sync()
        s1 = ix.pipeline_stats()
        lb = (n - 1) % nb
        ok = lb >= 2 or (np.array_equal(last[0], want[lb][0]) and np.array_equal(last[1].view(np.uint32), want[lb][1].view(np.uint32)))
        t = ix.timing()
        ql = (s1["queries"] - s0["queries"]) / max(1, s1["launches"] - s0["launches"])
        print(json.dumps({"coalesce_cap": cap, "tickets_in_flight": depth, "qps": nq * n / el, "queries_per_launch": ql, "kernel_ms_per_launch": t["search_kernel_ms"],
                          "kernel_ms_per_10k_queries": t["search_kernel_ms"] * 10000 / ql, "same_bits": bool(ok)}), flush=True)
