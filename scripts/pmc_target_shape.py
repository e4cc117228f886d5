"""PMC target for the other shapes (run under rocprofv3 --pmc ...): builds a c3- or c4-shaped index on the device and
runs a few M1 launches; prints the algorithmic bytes of a launch and a calibration byte count.
usage: pmc_target_shape.py c3|c4|c5s N [bw] [L]   (c5s: PQ-only traversal DR_MODE_PQ on R = 32 rows, the c5 shard shape)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel
shape, n = sys.argv[1], int(sys.argv[2])
bw = int(sys.argv[3]) if len(sys.argv) > 3 else 8
L = int(sys.argv[4]) if len(sys.argv) > 4 else 100
D, m, ncl, latent = {"c3": (1536, 32, 4096, 64), "c4": (96, 16, 4096, 32), "c5s": (1536, 32, 4096, 64)}[shape]
R = 32 if shape == "c5s" else 64
import os
mode = _ffi.MODE_PQ if shape == "c5s" else _ffi.MODE_M1
flags = 0
if os.environ.get("PMC_MODE") == "pqb":
    mode = _ffi.MODE_PQB                # round 5: the batch-per-step traversal (csrc/pqb_kernel.hpp), default pops
    if shape != "c5s":
        flags = _ffi.F_RERANK           # c3 / c4: PQ traversal + exact rerank of the L list (bench.py --config c3 | c4)
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=10000, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=R)
ix.build_vamana(L_build=100 if shape != "c5s" else 64, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(m, n_sample=100000, iters=5); ix.pq_encode(cb)
if os.environ.get("DR_INLINE") == "1":
    ix.inline_codes(True)               # code words of a node's neighbours beside its adjacency row (dr_index_inline_codes)
ix.bruteforce_topk(q[:1], 10)           # calibration: streams the whole vector table once
ix.batch_upload(q)
for _ in range(4):
    ix.batch_run(10, L=L, beam_width=bw, mode=mode, flags=flags)
ids, dist, cnt, st = ix.batch_download()
S, V, X = st["steps"].astype(np.float64), st["pq_evaluated"].astype(np.float64), st["exact"].astype(np.float64)
print("ALG_BYTES_PER_LAUNCH", float((4 * D + S * 4 * R + V * m + X * 4 * D + 80).sum() + 4 * 256 * D))
print("CALIB_BYTES", n * D * 4)
print("N", n); print("L", L); print("BW", bw)
print("KERNEL_MS", ix.timing()["search_kernel_ms"], "VARIANT", ix.timing()["variant"])
print("PER_QUERY steps %.1f exact %.1f pq_eval %.1f" % (S.mean(), X.mean(), V.mean()))
