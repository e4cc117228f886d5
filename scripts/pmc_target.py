"""PMC target (run under rocprofv3 --pmc ...): one calibration kernel with a known HBM byte count (brute force over
the whole vector table for ONE query = N*D*4 bytes, same 128-byte-per-row access shape as the search kernel),
then a few search steps of the bench workload."""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
import hashlib, os
NQ = int(os.environ.get("PMC_NQ", "10000"))        # queries per launch (20000: the shape of the bench's coalesced launches)
x, q = sift_like(1000000, 128, n_queries=NQ, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=100000, iters=8); ix.pq_encode(cb)
ix.bruteforce_topk(q[:1], 10)
bw = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ix.batch_upload(q)
for _ in range(4):
    ix.batch_run(10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
ids, dist, cnt, st = ix.batch_download()
S, V, X = st["steps"].astype(np.float64), st["pq_evaluated"].astype(np.float64), st["exact"].astype(np.float64)
print("ALG_BYTES_PER_LAUNCH", float((4 * 128 + S * 4 * 64 + V * 32 + X * 4 * 128 + 80).sum() + 4 * 256 * 128))
print("ALG_BYTES_OWN_PER_LAUNCH", float((4 * 128 + S * 4 * 64 + V * 32 + X * (128 if ix.timing()["variant"] in (11, 13, 16, 17) else 512) + 80).sum() + 4 * 256 * 128))
print("QUERIES_PER_LAUNCH", NQ)
print("KERNEL_MS", ix.timing()["search_kernel_ms"], "VARIANT", ix.timing()["variant"])
lib = os.environ.get("DR_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diskrag_amd", "libdiskrag_hip.so")
print("BUILD_SHA1", hashlib.sha1(open(lib, "rb").read()).hexdigest()[:16])
print("CALIB_BYTES", 1000000 * 128 * 4)
print("BEAM_WIDTH", bw)
print("EXPANSIONS_PER_LAUNCH", float(S.sum()))
