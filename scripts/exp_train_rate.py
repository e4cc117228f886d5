"""Diagnostic (GPU box): time of dr_pq_train_ex (k-means++ seeding + Lloyd on the device) with a hash of the codebook (A/B of library
builds through DR_LIB: the hash must not change).
usage: exp_train_rate.py N D m n_sample max_iter"""
import hashlib
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex
from diskrag_amd.synth import UnitMixtureStream, sift_like
N, D, m, ns, it = (int(v) for v in sys.argv[1:6])
x = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=64).draw(0, N) if D == 1536 else sift_like(N, D, n_queries=16, seed=1)[0]
ix = HipIndex.create_empty(x, R=32)
ix.pq_train_ex(m, n_sample=min(ns, 4096), max_iter=2, n_init=1, seed=5)       # warm-up
t0 = time.perf_counter()
cb, inertia = ix.pq_train_ex(m, n_sample=ns, max_iter=it, n_init=1, seed=5)
dt = time.perf_counter() - t0
print("TRAIN N", N, "D", D, "m", m, "n_sample", ns, "max_iter", it, "seconds %.3f" % dt, "inertia %.6g" % inertia, "codebook_sha1", hashlib.sha1(cb.tobytes()).hexdigest()[:16])
