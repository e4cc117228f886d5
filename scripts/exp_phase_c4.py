"""Diagnostic (GPU box): phase shares of one expansion on c4- / c3-shaped A4-live data. Needs a -DDR_PHASE_TIMING build (DR_LIB).
usage: exp_phase_c4.py N beam_width [D=96|1536] [M1|M2|PQ] [inline]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
D = int(sys.argv[3]) if len(sys.argv) > 3 else 96            # 96: c4 shape; 1536: c3 shape
mode = {"M1": _ffi.MODE_M1, "M2": _ffi.MODE_M2, "PQ": _ffi.MODE_PQ}[sys.argv[4] if len(sys.argv) > 4 else "M1"]
inline = len(sys.argv) > 5 and sys.argv[5] == "inline"
x, q = unit_mixture(n, D, n_queries=10000, n_clusters=4096, seed=11, latent=32 if D == 96 else 64)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(16 if D == 96 else 32, n_sample=100000, iters=5); ix.pq_encode(cb)
if inline:
    ix.inline_codes(True)
for _ in range(2):
    ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=bw, mode=mode)
ph = np.array(ix.debug_phase_cycles())
names = ["setup+LUT", "pop/stop", "adjacency", "visited", "ADC", "exact", "decisions", "output"]
tot = ph.sum()
t = ix.timing()
print("kernel_ms", t["search_kernel_ms"], "steps", st["steps"].mean(), "visited", st["visited"].mean(), "exact", st["exact"].mean(), "inserts", st["inserts"].mean(), t)
for nme, v in zip(names, ph):
    print(f"{nme:12s} {v/tot*100:6.2f}%  cycles/query {v/len(q):10.0f}  per-step {v/st['steps'].sum():8.0f}")
