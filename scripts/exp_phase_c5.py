"""Diagnostic (GPU box): phase shares of the PQ-only traversal (DR_MODE_PQ) on the c5 shard shape (D = 1536, m = 32, R = 32).
Needs a -DDR_PHASE_TIMING build (DR_LIB). usage: exp_phase_c5.py N"""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture_parallel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
x, q = unit_mixture_parallel(n, 1536, n_queries=10000, n_clusters=4096, seed=11, latent=64)
ix = HipIndex.create_empty(x, R=32)
ix.build_vamana(L_build=64, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
names = ["setup+LUT", "pop/stop", "adjacency", "visited", "ADC", "exact", "decisions", "output"]
for L, bw in ((100, 8), (200, 0), (400, 0)):
    for _ in range(2):
        ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_PQ)
    ph = np.array(ix.debug_phase_cycles())
    tot = ph.sum()
    t = ix.timing()
    print(f"# DR_MODE_PQ N={n} L={L} bw={bw}: kernel_ms {t['search_kernel_ms']:.3f} variant {t['variant']} waves/CU {t['waves_per_cu']} steps {st['steps'].mean():.1f} "
          f"visited {st['visited'].mean():.1f} pq_eval {st['pq_evaluated'].mean():.1f} inserts {st['inserts'].mean():.1f}")
    for nme, v in zip(names, ph):
        print(f"{nme:12s} {v/tot*100:6.2f}%  cycles/query {v/len(q):10.0f}  per-step {v/st['steps'].sum():8.0f}")
