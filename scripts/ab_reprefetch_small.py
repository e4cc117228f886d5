"""A/B (GPU box): the second-chance adjacency prefetch (DR_REPREFETCH=1) where the chip is NOT full -- one query, small batches: there an
expansion's chain of round trips is the whole cost.  usage: ab_reprefetch_small.py -> JSON lines"""
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
ix.pq_encode(ix.pq_train(32, n_sample=100000, iters=8))
for rep in range(2):
    for nq in (1, 8, 64, 512, 1250, 2500, 4096):
        for (L, bw, mode) in ((100, 8, _ffi.MODE_M1), (0, 8, _ffi.MODE_M2)):
            rec = {"nq": nq, "mode": int(mode), "L": L, "beam_width": bw}
            for on in ("0", "1"):
                os.environ["DR_REPREFETCH"] = on
                ix.batch_upload(q[:nq])
                for i in range(5):
                    ix.batch_run(10, L=L, beam_width=bw, mode=mode)
                ix.batch_sync()
                t0 = time.perf_counter()
                for i in range(50):
                    ix.batch_run(10, L=L, beam_width=bw, mode=mode)
                ix.batch_sync()
                wall = (time.perf_counter() - t0) / 50
                t = ix.timing()
                ids, dist, cnt, st = ix.batch_download()
                rec["kernel_ms_" + ("second_chance" if on == "1" else "plain")] = t["search_kernel_ms"]
                rec["hits_" + ("second_chance" if on == "1" else "plain")] = float(st["adj_prefetch_hits"].sum() / max(1, st["steps"].sum()))
                rec["variant"] = t["variant"]
                rec.setdefault("ids_sum", []).append(int(ids.astype(np.uint64).sum()))
            rec["same"] = rec["ids_sum"][0] == rec["ids_sum"][1]; del rec["ids_sum"]
            rec["speedup"] = rec["kernel_ms_plain"] / rec["kernel_ms_second_chance"]
            print(json.dumps(rec), flush=True)
