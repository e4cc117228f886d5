"""Experiment (GPU box): where the time of a stream of SMALL submits goes. For submit sizes 10000 / 5000 / 2500 / 1250 and a few (coalesce cap,
tickets in flight) settings: queries/s, the host thread's time inside dr_search_submit and inside dr_search_wait (perf_counter around the
ctypes calls), queries per launch and the search stream's busy share (kernel time x launches / wall time).
usage: exp_small_submit_host.py -> JSON lines"""
import json
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like

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


def run(n, srcs, depth):
    jobs, done = [], 0
    t_sub = t_wait = 0.0
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        jobs.append(ix.search_submit(srcs[i % nb], 10, L=100, beam_width=8, mode=_ffi.MODE_M1, reuse_outputs=True))
        b = time.perf_counter(); t_sub += b - a
        if len(jobs) - done >= depth:
            jobs[done].wait(); jobs[done] = None; done += 1
            t_wait += time.perf_counter() - b
    b = time.perf_counter()
    for j in range(done, len(jobs)):
        jobs[j].wait()
    t_wait += time.perf_counter() - b
    return time.perf_counter() - t0, t_sub, t_wait


grid = {10000: ((32768, 14), (32768, 8), (10240, 4)),
        5000: ((32768, 26), (32768, 16), (20480, 16), (10240, 8)),
        2500: ((32768, 54), (32768, 30), (20480, 30), (20480, 20), (10240, 16)),
        1250: ((32768, 62), (32768, 40), (20480, 40), (10240, 30))}
for rep in range(2):
    for n_g, settings in grid.items():
        srcs = [a[:n_g] for a in qb]
        for cap, depth in settings:
            ix.set_coalesce(cap)
            run(3 * depth, srcs, depth); ix.batch_sync()
            s0 = ix.pipeline_stats()
            n = 400 * (nq // n_g)
            el, t_sub, t_wait = run(n, srcs, depth)
            ix.batch_sync()
            s1 = ix.pipeline_stats()
            t = ix.timing()
            nl = max(1, s1["launches"] - s0["launches"])
            print(json.dumps({"queries_per_submit": n_g, "coalesce_cap": cap, "tickets_in_flight": depth, "qps": n_g * n / el,
                              "host_us_per_submit": t_sub / n * 1e6, "host_us_per_wait": t_wait / n * 1e6, "host_share_in_submit": t_sub / el,
                              "queries_per_launch": (s1["queries"] - s0["queries"]) / nl, "kernel_ms_per_launch": t["search_kernel_ms"],
                              "search_stream_busy": t["search_kernel_ms"] * 1e-3 * nl / el}), flush=True)
