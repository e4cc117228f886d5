"""Operating points of DR_MODE_PQB | DR_F_RERANK with the rerank cut to the ADC top of the list (DR_F_RERANK_TOP, round 6) on a c3- / c4-shaped
index built on the device (GPU box): for every (L, beam_width) the recall@10 against the EXACT neighbours and the QPS as the rerank depth shrinks,
resident launches first, then the best points at or above the recall bar as a host -> host stream (dr_search_submit / dr_search_wait).
usage: op_rerank_top.py c3|c4 N [recall bar = 0.95]  -> JSON lines on stdout and in gpurun_out/op_rerank_top_<shape>_<N>.jsonl"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                      # noqa: E402
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel, recall_at_k     # noqa: E402

shape, n = sys.argv[1], int(sys.argv[2])
bar = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
D, m, ncl, latent = {"c3": (1536, 32, 4096, 64), "c4": (96, 16, 4096, 32)}[shape]
import os
nq = int(os.environ.get("OP_NQ", "10000"))
out = open(f"gpurun_out/op_rerank_top_{shape}_{n}{os.environ.get('OP_TAG', '')}.jsonl", "w")


def emit(rec):
    out.write(json.dumps(rec) + "\n"); out.flush()
    print(json.dumps(rec), flush=True)


t0 = time.perf_counter()
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=nq, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=64)
med, bsec = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(m, n_sample=100000, iters=5)
ix.pq_encode(cb)
gt, _ = ix.bruteforce_topk(q, 10)
emit({"setup": {"shape": shape, "N": n, "D": D, "m": m, "build_s": bsec, "setup_s": time.perf_counter() - t0}})
del x
ix.batch_upload(q)
grid = {"c3": [(L, bw) for L in (100, 125, 150, 200, 250, 300) for bw in (0, 128)], "c4": [(L, bw) for L in (100, 150, 200, 300, 400) for bw in (8, 32, 64)]}[shape]
tops = {"c3": (0, 200, 150, 125, 100, 80, 72, 64), "c4": (0, 100, 64, 48, 32, 24, 16)}[shape]
if os.environ.get("OP_GRID"):        # "L:bw,L:bw,..." and OP_TOPS="0,200,..." override the shape's grid
    grid = [tuple(int(v) for v in g.split(":")) for g in os.environ["OP_GRID"].split(",")]
    tops = tuple(int(v) for v in os.environ.get("OP_TOPS", "0").split(","))
good = []
for L, bw in grid:
    for top in tops:
        if top >= L:
            continue
        kw = dict(L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_RERANK_TOP(top))
        ix.batch_run(10, **kw); ix.batch_sync()
        t1 = time.perf_counter()
        for _ in range(3):
            ix.batch_run(10, **kw)
        ix.batch_sync()
        dt = (time.perf_counter() - t1) / 3
        ids, dist, cnt, st = ix.batch_download()
        tm = ix.timing()
        rec = recall_at_k(ids, gt, 10)
        r = {"L": L, "bw": bw, "rerank_top": top or L, "qps_resident": nq / dt, "ms_per_batch": dt * 1e3, "recall_at_10": rec, "traversal_kernel_ms": tm["search_kernel_ms"],
             "table_kernel_ms": tm["lut_kernel_ms"], "rerank_and_rest_ms": dt * 1e3 - tm["search_kernel_ms"] - tm["lut_kernel_ms"], "steps": float(st["steps"].mean()),
             "exact": float(st["exact"].mean()), "status": int(st["status"].max())}
        emit(r)
        if rec >= bar:
            good.append((nq / dt, L, bw, top))
# the five fastest points at or above the bar, as a host -> host stream: 10k-query submits, 14 in flight (bench.py's shape)
qp = _ffi.pinned_empty((nq, D), np.float32)
qp[:] = q
for _, L, bw, top in sorted(good, reverse=True)[:5]:
    kw = dict(L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_RERANK_TOP(top))
    def stream(nb):
        jobs, last = [], None
        t1 = time.perf_counter()
        for i in range(nb):
            jobs.append(ix.search_submit(qp, 10, reuse_outputs=False, **kw))
            if len(jobs) >= 14:
                last = jobs.pop(0).wait()
        for j in jobs:
            last = j.wait()
        return time.perf_counter() - t1, last
    stream(20)
    el, last = stream(60)
    emit({"stream": True, "L": L, "bw": bw, "rerank_top": top or L, "qps_stream": 60 * nq / el, "recall_at_10": recall_at_k(last[0], gt, 10), "pipeline": ix.pipeline_stats()})
