"""Kernel-trace target: a stream of small pipelined batches through dr_search_submit / dr_search_wait, to be
run under `rocprofv3 --kernel-trace`. With DR_TWO_LANES=1 (round-3 library) consecutive small batches alternate between two
search streams; scripts/analyse_overlap.py then says from the trace whether their search kernels overlapped in time.
usage: exp_two_lanes_trace.py <queries_per_batch> <launches> [depth]"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like

nqb, launches = int(sys.argv[1]), int(sys.argv[2])
depth = int(sys.argv[3]) if len(sys.argv) > 3 else _ffi.PIPE_DEPTH
x, q = sift_like(1000000, 128, n_queries=8 * nqb, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
ix.pq_encode(ix.pq_train(32, n_sample=100000, iters=8))
qb = []
for b in range(8):
    a = _ffi.pinned_empty((nqb, 128), np.float32)
    a[:] = q[b * nqb:(b + 1) * nqb]
    qb.append(a)


def run(n):
    jobs, done = [], 0
    t0 = time.perf_counter()
    for i in range(n):
        jobs.append(ix.search_submit(qb[i % 8], 10, L=100, beam_width=8, mode=_ffi.MODE_M1, reuse_outputs=True))
        if len(jobs) - done >= depth:
            jobs[done].wait(); done += 1
    while done < len(jobs):
        jobs[done].wait(); done += 1
    return time.perf_counter() - t0


run(24)
ix.batch_sync()
el = run(launches)
ix.batch_sync()
print("QPS %.0f batch %d launches %d depth %d kernel_ms %.4f variant %d" % (nqb * launches / el, nqb, launches, depth, ix.timing()["search_kernel_ms"], ix.timing()["variant"]))
