"""Diagnostic (GPU box, under rocprofv3 --kernel-trace --stats): kernel breakdown of the PQ-only builder (dr_build_vamana_pq).
usage: exp_build_pq_profile.py N R L_build   (DR_PQ_PRUNE_LDS=1: the prune with its table rows in LDS; the graph hash must not change)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex
from diskrag_amd.synth import UnitMixtureStream
N, R, LB = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gen = UnitMixtureStream(d=1536, n_clusters=4096, seed=11, latent=64, threads=64)
x = gen.draw(0, N)
tmp = HipIndex.create_empty(x[:262144], R=R)
cb, _ = tmp.pq_train_ex(32, n_sample=50000, max_iter=15, n_init=1, seed=5)
tmp.close()
sh = HipIndex.create_codes_empty(N, 1536, R, cb)
for r0 in range(0, N, 1 << 20):
    sh.encode_rows(x[r0:r0 + (1 << 20)], r0)
med, secs = sh.build_vamana_pq(L_build=LB, alpha=1.2, passes=2, seed=7)
print("BUILD_S", secs, "N", N, "R", R, "L_build", LB)
import hashlib
adj = sh.get_adjacency()
print("GRAPH_SHA1", hashlib.sha1(np.ascontiguousarray(adj).tobytes()).hexdigest()[:16], "mean_degree", float((adj != 0xFFFFFFFF).sum(1).mean()))
