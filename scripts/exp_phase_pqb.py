"""Diagnostic (GPU box): shader-clock shares of the phases of a DR_MODE_PQB step (csrc/pqb_kernel.hpp). Needs a -DDR_PHASE_TIMING build (DR_LIB).
usage: exp_phase_pqb.py c5s|c3 N"""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel
shape, n = sys.argv[1], int(sys.argv[2])
D, m, ncl, latent, R, Lb = {"c3": (1536, 32, 4096, 64, 64, 100), "c5s": (1536, 32, 4096, 64, 32, 64)}[shape]
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=10000, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=R)
ix.build_vamana(L_build=Lb, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(m, n_sample=100000, iters=5); ix.pq_encode(cb)
names = ["table landing", "pop", "rows", "code words", "ADC", "candidates", "merge+trim", "output"]
for L, bw in (((100, 8), (200, 0)) if shape == "c5s" else ((250, 0), (100, 8))):
    for pops in (1, 2):
        for treg in ("16", "24"):
            os.environ["DR_PQB_TREG"] = treg
            for _ in range(2):
                ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(pops))
            ph = np.array(ix.debug_phase_cycles())
            tot = ph.sum()
            t = ix.timing()
            nsteps = st["steps"].sum() / pops
            print(f"# DR_MODE_PQB {shape} N={n} L={L} bw={bw} pops={pops} treg={treg}: kernel_ms {t['search_kernel_ms']:.3f} waves/CU {t['waves_per_cu']} expanded {st['steps'].mean():.1f} "
                  f"scored {st['pq'].mean():.1f} inserts {st['inserts'].mean():.1f}")
            for nme, v in zip(names, ph):
                print(f"{nme:14s} {v/tot*100:6.2f}%  cycles/query {v/len(q):10.0f}  per-step {v/nsteps:8.0f}")
