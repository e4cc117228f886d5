"""A/B (GPU box): batches of 10 000 queries at the API's list size (k 5, L 20, beam_width 8) and at L = 48 through the batch kernels, "ask later"
(search_kernel.hpp, round 5) on and off (DR_NO_ASK_LATER=1), interleaved, resident batches on the 1M-point bench index. -> JSON lines"""
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
for (k, L) in ((5, 20), (10, 48), (10, 100)):
    res = {}
    for rep in range(3):
        for name, env in (("ask_later", None), ("always_ask_first", "1")):
            if env: os.environ["DR_NO_ASK_LATER"] = env
            else: os.environ.pop("DR_NO_ASK_LATER", None)
            ix.batch_upload(q)
            for _ in range(2): ix.batch_run(k, L=L, beam_width=8, mode=_ffi.MODE_M1)
            ix.batch_sync()
            t0 = time.perf_counter()
            for _ in range(10): ix.batch_run(k, L=L, beam_width=8, mode=_ffi.MODE_M1)
            ix.batch_sync()
            dt = (time.perf_counter() - t0) / 10
            ids, dist, cnt, st = ix.batch_download()
            res.setdefault(name, []).append({"ms_per_batch": round(dt * 1e3, 4), "kernel_ms": round(ix.timing()["search_kernel_ms"], 4), "variant": ix.timing()["variant"],
                                             "pq_evaluated_share": round(float(st["pq_evaluated"].sum()) / max(1, float(st["pq"].sum())), 4), "ids_sha": int(ids.astype(np.uint64).sum())})
    os.environ.pop("DR_NO_ASK_LATER", None)
    print(json.dumps({"k": k, "L": L, "beam_width": 8, "queries": len(q), **{n: {"qps_median": round(len(q) / (np.median([r["ms_per_batch"] for r in v]) * 1e-3)), "runs": v} for n, v in res.items()}}))
