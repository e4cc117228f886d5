"""Is the reference-faithful PQ traversal (M3 with PQ) deterministic at scale, and equal to the oracle?
usage: python scripts/dbg_m3_determinism.py [N] [D]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture
from oracle import pyoracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
m = 32 if D >= 128 else 16
x, q = unit_mixture(n, D, n_queries=10000, n_clusters=4096, seed=11, latent=64)
ix = HipIndex.create_empty(x, R=64)
med, _ = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(m, n_sample=100000, iters=5)
codes = ix.pq_encode(cb, want_codes=True)
adj = ix.get_adjacency()
runs = []
for rep in range(4):
    ids, dist, cnt, st = ix.search_batch(q, 10, L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
    runs.append((ids.copy(), dist.copy(), st["steps"].copy(), st["pq"].copy()))
    if rep:
        d = np.nonzero((runs[0][0] != ids).any(axis=1) | (runs[0][1].view(np.uint32) != dist.view(np.uint32)).any(axis=1) | (runs[0][2] != st["steps"]))[0]
        print(f"run {rep} vs run 0: {d.size} of {len(q)} queries differ", d[:10], flush=True)
nchk = 400
w = orc.search_batch(x, adj, q[:nchk], med, orc.M3, 10, L=10, bw=64, flags=orc.F_USE_PQ, codes=codes, codebook=cb, nthreads=16)
for rep, r in enumerate(runs):
    bad = np.nonzero((r[0][:nchk] != w[0]).any(axis=1))[0]
    print(f"run {rep} vs oracle (first {nchk}): {bad.size} differ", bad[:10], "steps equal:", bool((r[2][:nchk] == w[3][:, 0]).all()))
    if bad.size:
        b = bad[0]
        print(" device", r[0][b], r[1][b], "steps", r[2][b], "pq", r[3][b]); print(" oracle", w[0][b], w[1][b], w[3][b])
