"""Diagnostic (GPU box): health of a trained PQ codebook on unit-mixture data -- inertia, centroids in use per sub-quantiser,
distinct code words, brute-force ADC top-10 against exact top-10. usage: DR_LIB=... dbg_codebook.py N n_clusters [n_sample] [max_iter]"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex
from diskrag_amd.synth import UnitMixtureStream, recall_at_k
N, NCL = int(sys.argv[1]), int(sys.argv[2])
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 15
gen = UnitMixtureStream(d=1536, n_clusters=NCL, seed=11, latent=64, threads=64)
x = gen.draw(0, N); q = gen.draw(0, 200, stream=1)
ix = HipIndex.create_empty(x, R=32)
t0 = time.perf_counter()
cb, inertia = ix.pq_train_ex(32, n_sample=ns, max_iter=iters, n_init=1, seed=5)
t1 = time.perf_counter()
codes = ix.pq_encode(cb, want_codes=True)
t2 = time.perf_counter()
used = [len(np.unique(codes[:, j])) for j in range(32)]
uniq = len(np.unique(codes.view([("", codes.dtype)] * 32)))
rec = np.concatenate([cb[j][codes[:200000, j]] for j in range(32)], axis=1)
err = float(((x[:200000].astype(np.float64) - rec) ** 2).sum(axis=1).mean())
gt, _ = ix.bruteforce_topk(q, 10)
try:
    adc = ix.pq_scan_topk(q, 10)[0]
    r = recall_at_k(adc, gt, 10)
except Exception as e:          # (older library)
    r = str(e)
print(f"N={N} ncl={NCL} train_s {t1 - t0:.2f} encode_s {t2 - t1:.2f} inertia/sample {inertia / ns:.6f} mean sq err/vector {err:.6f} "
      f"centroids used min {min(used)} max {max(used)} distinct code words {uniq} of {N} adc_top10_vs_exact {r} cb finite {np.isfinite(cb).all()} "
      f"cb norm range {np.linalg.norm(cb.reshape(32 * 256, -1), axis=1).min():.4f}..{np.linalg.norm(cb.reshape(32 * 256, -1), axis=1).max():.4f}")
