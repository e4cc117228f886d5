"""Diagnostic (GPU box): where one M1 expansion spends its cycles. Needs a -DDR_PHASE_TIMING build (DR_LIB)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
x, q = sift_like(n, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=20000, iters=3); ix.pq_encode(cb)
for _ in range(2):
    ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=int(sys.argv[2]) if len(sys.argv) > 2 else 0, mode=_ffi.MODE_M1)
ph = np.array(ix.debug_phase_cycles())
names = ["setup+LUT", "pop/stop", "adjacency", "visited", "ADC", "exact", "decisions", "output"]
tot = ph.sum()
t = ix.timing()
print("kernel_ms", t["search_kernel_ms"], "finalize_ms", t["finalize_kernel_ms"], "steps", st["steps"].mean(), "visited", st["visited"].mean(), "inserts", st["inserts"].mean(), t)
for nme, v in zip(names, ph):
    print(f"{nme:12s} {v/tot*100:6.2f}%  cycles/query {v/len(q):10.0f}  per-step {v/st['steps'].sum():8.0f}")
