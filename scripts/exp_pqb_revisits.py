"""Diagnostic (GPU box, CPU side): of the code words DR_MODE_PQB scores more than once, how many belong to nodes that sit IN the list at that
moment (a list-membership filter in LDS would skip them before their code words are fetched) and how many to nodes rejected or evicted earlier?
Runs the CPU restatement (oracle, ORC_PQB_DIAG=1) on the device-built graph for a few hundred queries. usage: exp_pqb_revisits.py c5s|c3|c5w N"""
import json
import os
import sys
import numpy as np
sys.path.insert(0, ".")
os.environ["ORC_PQB_DIAG"] = "1"
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel
from oracle import pyoracle as orc
shape, n = sys.argv[1], int(sys.argv[2])
D, m, ncl, latent, R, Lb = {"c3": (1536, 32, 4096, 64, 64, 100), "c5s": (1536, 32, 4096, 64, 32, 64), "c5w": (1536, 32, 4096, 64, 128, 128)}[shape]
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=256, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=R)
if shape == "c5w":
    cb = ix.pq_train(m, n_sample=100000, iters=5); codes = ix.pq_encode(cb, want_codes=True)
    ix.build_vamana_pq(L_build=Lb, alpha=1.2, passes=2, seed=7)
else:
    ix.build_vamana(L_build=Lb, alpha=1.2, passes=2, seed=7)
    cb = ix.pq_train(m, n_sample=100000, iters=5); codes = ix.pq_encode(cb, want_codes=True)
adj = ix.get_adjacency()
med = ix.medoid
for L, bw in ((100, 8), (250, 0), (100, 32)):
    for pops in (1, 2, 4):
        if pops * (1 << int(np.ceil(np.log2(R)))) > 256: continue
        w = orc.search_batch(x, adj, q, med, orc.PQB, 10, L=L, bw=bw, flags=orc.F_POPS(pops), codes=codes, codebook=cb, nthreads=64)
        st = w[3].astype(np.float64)
        a = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_PQ)[3]
        print(json.dumps({"shape": shape, "N": n, "L": L, "bw": bw, "pops": pops, "expanded": st[:, 0].mean(), "scored": st[:, 1].mean(),
                          "in_list_when_scored": st[:, 2].mean(), "distinct_nodes(DR_MODE_PQ visited)": float(a["visited"].mean())}), flush=True)
