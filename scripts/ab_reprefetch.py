"""A/B (GPU box): the adjacency prefetch of the byte-query variants with and without its second chance (DR_REPREFETCH=1 per launch):
kernel time, prefetch hits per expansion, same results. One process, one index, forms alternated.  usage: ab_reprefetch.py -> JSON lines"""
import hashlib
import json
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like

nq, nb = 10000, 8
x, q = sift_like(1000000, 128, n_queries=nq * nb, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
ix.pq_encode(ix.pq_train(32, n_sample=100000, iters=8))
for b in range(nb):
    ix.batch_select(b); ix.batch_upload(q[b * nq:(b + 1) * nq])
for rep in range(3):
    for (L, bw, mode) in ((100, 8, _ffi.MODE_M1), (100, 0, _ffi.MODE_M1), (0, 8, _ffi.MODE_M2)):
        for on in ("0", "1"):
            os.environ["DR_REPREFETCH"] = on
            for i in range(8):
                ix.batch_select(i % nb); ix.batch_run(10, L=L, beam_width=bw, mode=mode)
            ix.batch_sync()
            for i in range(40):
                ix.batch_select(i % nb); ix.batch_run(10, L=L, beam_width=bw, mode=mode)
            ix.batch_sync()
            t = ix.timing()
            ix.batch_select(0); ix.batch_run(10, L=L, beam_width=bw, mode=mode)
            ids, dist, cnt, st = ix.batch_download()
            h = hashlib.sha1(ids.tobytes() + dist.tobytes() + st["steps"].tobytes() + st["visited"].tobytes()).hexdigest()[:12]
            print(json.dumps({"mode": int(mode), "L": L, "beam_width": bw, "second_chance": on == "1", "kernel_ms": t["search_kernel_ms"], "variant": t["variant"],
                              "prefetch_hits_per_expansion": float(st["adj_prefetch_hits"].sum() / st["steps"].sum()), "results_sha1": h}), flush=True)
