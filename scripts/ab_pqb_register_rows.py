"""A/B (GPU box, one process, one index): DR_MODE_PQB at m = 64 (or 48) with 32 / 40 / 48 (16 / 32) of the table rows in registers (DR_PQB_TREG; 32
rows in LDS are 4 wavefronts per CU, 24 are 6, 16 are 8) on a PQ-only index of N x 1536 points (the c5 shape at bench scale, graph built from the code
words); the last entry of the list, 0, is the engine's own choice.
usage: ab_pqb_register_rows.py [N=1000000] [R=128] [m=64]  -> JSON lines on stdout"""
import hashlib
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                     # noqa: E402
from diskrag_amd.synth import unit_mixture, recall_at_k    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 128
D = 1536
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
x, q = unit_mixture(n, D, n_queries=10000, n_clusters=4096, seed=11, latent=64)
full = HipIndex.create_empty(x, R=8)
cb = full.pq_train(m, n_sample=100000, iters=5)
codes = full.pq_encode(cb, want_codes=True)
gt, _ = full.bruteforce_topk(q[:1000], 10)
full.close()
sh = HipIndex.create_codes_empty(n, D, R, cb)
sh.set_pq(cb, codes)
t0 = time.perf_counter()
sh.build_vamana_pq(L_build=128, alpha=1.2, passes=2, seed=7)
print(json.dumps({"setup": {"N": n, "R": R, "m": m, "graph_s": time.perf_counter() - t0}}), flush=True)
gta, _, _ = sh.pq_scan_topk(q[:1000], 10)
sh.batch_upload(q)
for rnd in range(3):
    for treg in ((32, 40, 48, 0) if m == 64 else (16, 32, 0)):
        if treg: os.environ["DR_PQB_TREG"] = str(treg)
        else: os.environ.pop("DR_PQB_TREG", None)
        for L, bw in ((100, 32), (200, 32)):
            kw = dict(L=L, beam_width=bw, mode=_ffi.MODE_PQB)
            sh.batch_run(10, **kw); sh.batch_sync()
            t1 = time.perf_counter()
            for _ in range(3):
                sh.batch_run(10, **kw)
            sh.batch_sync()
            dt = (time.perf_counter() - t1) / 3
            ids, dist, cnt, st = sh.batch_download()
            tm = sh.timing()
            print(json.dumps({"round": rnd, "treg": treg, "L": L, "bw": bw, "ms_per_batch": dt * 1e3, "kernel_ms": tm["search_kernel_ms"], "table_kernel_ms": tm["lut_kernel_ms"],
                              "recall_vs_adc": recall_at_k(ids[:1000], gta, 10), "recall_vs_exact": recall_at_k(ids[:1000], gt, 10), "status": int(st["status"].max()),
                              "results_sha1": hashlib.sha1(ids.tobytes() + dist.tobytes()).hexdigest()[:12]}), flush=True)
