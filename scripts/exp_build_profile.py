"""Diagnostic (GPU box, under rocprofv3 --kernel-trace --stats): kernel breakdown of the exact-vector builder (dr_build_vamana).
usage: exp_build_profile.py N D R L_build"""
import hashlib
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex
from diskrag_amd.synth import UnitMixtureStream, sift_like
N, D, R, LB = (int(v) for v in sys.argv[1:5])
if D == 1536:
    x = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=64).draw(0, N)
else:
    x = sift_like(N, D, n_queries=16, seed=1)[0]
ix = HipIndex.create_empty(x, R=R)
med, secs = ix.build_vamana(L_build=LB, alpha=1.2, passes=2, seed=7, pad_with_zero=False)
adj = ix.get_adjacency()
print("BUILD_S", secs, "N", N, "D", D, "R", R, "L_build", LB)
print("GRAPH_SHA1", hashlib.sha1(np.ascontiguousarray(adj).tobytes()).hexdigest()[:16], "mean_degree", float((adj != 0xFFFFFFFF).sum(1).mean()))
