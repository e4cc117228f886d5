"""Experiment (GPU box): two independent pipelines in ONE process -- two index handles (two copies of the bench index, each with its
own streams, visited words and output sets), one host thread each -- against one handle. Two PROCESSES on one GPU reached
8.58 M QPS (profiles/r03/bench_2ranks_on_one_gpu.json) where one pipelined handle reaches 7.7 M; two search streams inside one
handle were slower. Which of the two is it?
usage: exp_two_handles.py -> gpurun_out/r03/two_handles.json"""
import json
import sys
import threading
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like

x, q = sift_like(1000000, 128, n_queries=80000, n_clusters=1024, seed=2024, query_seed=9000)
a = HipIndex.create_empty(x, R=64)
med, _ = a.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = a.pq_train(32, n_sample=100000, iters=5)
codes = a.pq_encode(cb, want_codes=True)
adj = a.get_adjacency()
b = HipIndex.create(x, adj, med)
b.set_pq(cb, codes)
batches = [np.ascontiguousarray(q[i * 10000:(i + 1) * 10000]) for i in range(8)]


def pipeline(ix, n_launch, out, key):
    jobs, done = [], 0
    t0 = time.perf_counter()
    for i in range(n_launch):
        jobs.append(ix.search_submit(batches[i % 8], 10, L=100, beam_width=8, mode=_ffi.MODE_M1, reuse_outputs=True))
        if len(jobs) - done >= _ffi.PIPE_DEPTH:
            jobs[done].wait(); done += 1
    while done < len(jobs):
        jobs[done].wait(); done += 1
    out[key] = time.perf_counter() - t0


res = {}
for rep in range(3):
    o = {}
    pipeline(a, 40, o, "warm")
    pipeline(a, 400, o, "one")
    one = 400 * 10000 / o["one"]
    o2 = {}
    pipeline(b, 40, o2, "warm")
    th = [threading.Thread(target=pipeline, args=(ix, 400, o2, k)) for ix, k in ((a, "a"), (b, "b"))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    two = 800 * 10000 / (time.perf_counter() - t0)
    res[f"rep{rep}"] = {"one_handle_qps": one, "two_handles_two_threads_qps": two}
    print(res[f"rep{rep}"], flush=True)
json.dump(res, open("gpurun_out/r03/two_handles.json", "w"), indent=1)
