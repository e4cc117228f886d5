"""Scaled runs of the other BASELINE shapes on one MI355X (GPU box): c3-shaped (D=1536 unit-norm) and c4-shaped
(D=96 unit-norm, DEEP-like), index built on the device, M1 with the reference's default beam_width and without trim.
Usage: python scripts/scale_measurements.py c3 1000000 | c4 10000000 [nq]   -> one JSON object on stdout."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                      # noqa: E402
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel, recall_at_k     # noqa: E402

shape, n = sys.argv[1], int(sys.argv[2])
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
D, m, ncl, latent = {"c3": (1536, 32, 4096, 64), "c4": (96, 16, 4096, 32), "u128": (128, 32, 4096, 32)}[shape]
R = 64
t0 = time.perf_counter()
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=nq, n_clusters=ncl, seed=11, latent=latent)
gen_s = time.perf_counter() - t0
t0 = time.perf_counter()
ix = HipIndex.create_empty(x, R=R)
up_s = time.perf_counter() - t0
med, bsec = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
t0 = time.perf_counter()
cb = ix.pq_train(m, n_sample=100000, iters=5)
ix.pq_encode(cb)
pq_s = time.perf_counter() - t0
gt, _ = ix.bruteforce_topk(q, 10)
out = {"shape": shape, "N": n, "D": D, "R": R, "m": m, "nq": nq, "data": f"{gen.__name__}(latent={latent}, clusters={ncl})",
       "generate_s": gen_s, "upload_s": up_s, "build_s": bsec, "pq_s": pq_s, "runs": {}}


def run(tag, **kw):
    ix.batch_upload(q)
    ix.batch_run(10, **kw); ix.batch_sync()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.batch_run(10, **kw)
    ix.batch_sync()
    dt = (time.perf_counter() - t0) / reps
    ids, dist, cnt, st = ix.batch_download()
    t = ix.timing()
    alg = float((4.0 * D + st["steps"] * 4.0 * R + st["pq_evaluated"] * float(m) + st["exact"] * 4.0 * D + 80).sum())
    out["runs"][tag] = {"qps": nq / dt, "recall_at_10": recall_at_k(ids, gt, 10), "kernel_ms": t["search_kernel_ms"],
                        "launch": {k: t[k] for k in ("grid", "block", "lds_bytes", "waves_per_cu")},
                        "steps": float(st["steps"].mean()), "exact": float(st["exact"].mean()),
                        "pq": float(st["pq"].mean()), "pq_evaluated": float(st["pq_evaluated"].mean()),
                        "status_max": int(st["status"].max()), "alg_bytes_per_query": alg / nq,
                        "alg_GBps": alg / (t["search_kernel_ms"] * 1e-3) / 1e9,
                        "frac_of_8TBps": alg / (t["search_kernel_ms"] * 1e-3) / 8e12}


run("M1_L100_bw8_policy0", L=100, beam_width=8, mode=_ffi.MODE_M1, band_policy=0)
run("M1_L100_notrim_policy0", L=100, beam_width=0, mode=_ffi.MODE_M1, band_policy=0)
run("M1_L100_notrim_policy1", L=100, beam_width=0, mode=_ffi.MODE_M1, band_policy=1)
run("M2_bw8", L=100, beam_width=8, mode=_ffi.MODE_M2)
# PQ-only traversal (the c5 path): result heap = k, Q9 trim; then the same with the vectors freed (a c5 shard)
run("M3_PQ_k10_bw8", L=10, beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
run("M3_PQ_k10_bw64", L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
ix.drop_vectors()
run("M3_PQ_k10_bw64_codes_only", L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
print(json.dumps(out))
