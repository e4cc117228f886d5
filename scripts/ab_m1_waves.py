"""A/B (GPU box): the c2 M1 kernel at different occupancies -- the bench's index and batches (1M x 128 SIFT-like, R = 64, m = 32, L = 100, beam_width 8),
resident 10 000-query launches, the variant forced: 13 (one 16-wavefront workgroup per CU) and 17 (4-wavefront workgroups; in the A/B builds
-DDR_AB_RB17=32 -DDR_AB_MINW17=5|6 the same code at 20 / 24 wavefronts per CU: bursts of 32 rows, 96 / 80 VGPRs). One process per library (DR_LIB);
prints kernel ms per launch (HIP events inside the library), QPS and a checksum of the results. usage: DR_LIB=... ab_m1_waves.py TAG [nq per launch]"""
import hashlib
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                      # noqa: E402
from diskrag_amd.synth import sift_like                     # noqa: E402

tag = sys.argv[1]
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
x, q = sift_like(1_000_000, 128, n_queries=nq * 4, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100_000, iters=8)
ix.pq_encode(cb)
for b in range(4):
    ix.batch_select(b); ix.batch_upload(q[b * nq:(b + 1) * nq])
import os
for kind in [int(v) for v in os.environ.get("AB_KINDS", "13,17,13,17").split(",")]:
    ix.debug_force_kind(kind)
    for i in range(4):
        ix.batch_select(i % 4); ix.batch_run(10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    ix.batch_sync()
    t0 = time.perf_counter()
    n = 24
    for i in range(n):
        ix.batch_select(i % 4); ix.batch_run(10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    ix.batch_sync()
    el = time.perf_counter() - t0
    tm = ix.timing()
    ix.batch_select(0)
    ids, dist, cnt, st = ix.batch_download()
    print(json.dumps({"lib": tag, "forced_kind": kind, "variant": tm["variant"], "waves_per_cu": tm["waves_per_cu"], "lds": tm["lds_bytes"], "block": tm["block"],
                      "kernel_ms": round(tm["search_kernel_ms"], 4), "qps_resident": round(n * nq / el), "nq": nq,
                      "results_sha1": hashlib.sha1(ids.tobytes() + dist.tobytes()).hexdigest()[:12], "status": int(st["status"].max())}), flush=True)
