"""Secondary measurements for DESIGN.md (GPU box): flat PQ scan, exact-mode searches, D=1536 configuration."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like, unit_mixture, recall_at_k
out = {}
# ---- c3-shaped (scaled down): D=1536 unit-norm, m=32
x, q = unit_mixture(200000, 1536, n_queries=2000, n_clusters=64, seed=11)
ix = HipIndex.create_empty(x, R=64)
med, bsec = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=50000, iters=5); ix.pq_encode(cb)
gt, _ = ix.bruteforce_topk(q, 10)
def run2(tag, **kw):
    ix.batch_upload(q)
    ix.batch_run(10, **kw); ix.batch_sync()
    t0 = time.perf_counter()
    for _ in range(3): ix.batch_run(10, **kw)
    ix.batch_sync()
    dt = (time.perf_counter() - t0) / 3
    ids, dist, cnt, st = ix.batch_download()
    out[tag] = {"N": 200000, "D": 1536, "build_s": bsec, "qps": len(q) / dt, "recall_at_10": recall_at_k(ids, gt, 10), "kernel_ms": ix.timing()["search_kernel_ms"],
                "steps": float(st["steps"].mean()), "exact": float(st["exact"].mean()), "pq_evaluated": float(st["pq_evaluated"].mean()), "status_max": int(st["status"].max()),
                "alg_GBps": float((4 * 1536 + st["steps"] * 4.0 * 64 + st["pq_evaluated"] * 32.0 + st["exact"] * 4.0 * 1536 + 80).sum()) / (ix.timing()["search_kernel_ms"] * 1e-3) / 1e9}
run2("D1536_M1_L100_policy0", L=100, beam_width=0, mode=_ffi.MODE_M1, band_policy=0)
run2("D1536_M1_L100_policy1", L=100, beam_width=0, mode=_ffi.MODE_M1, band_policy=1)
print(json.dumps(out, indent=1))
