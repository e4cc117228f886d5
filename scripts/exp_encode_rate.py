"""Diagnostic (GPU box): rate of dr_pq_encode_rows (streamed chunks -> code words) and of dr_pq_encode (stored vectors), with a hash
of the code words (A/B of library builds through DR_LIB: the hash must not change).
usage: exp_encode_rate.py N D m"""
import hashlib
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex
from diskrag_amd.synth import UnitMixtureStream, sift_like
N, D, m = (int(v) for v in sys.argv[1:4])
x = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=64).draw(0, N) if D == 1536 else sift_like(N, D, n_queries=16, seed=1)[0]
full = HipIndex.create_empty(x, R=32)
cb, _ = full.pq_train_ex(m, n_sample=50000, max_iter=10, n_init=1, seed=5)
t0 = time.perf_counter()
ca = full.pq_encode(cb, want_codes=True)
t_stored = time.perf_counter() - t0
sh = HipIndex.create_codes_empty(N, D, 32, cb)
sh.encode_rows(x[:4096], 0)
t0 = time.perf_counter()
for r0 in range(0, N, 1 << 20):
    sh.encode_rows(x[r0:r0 + (1 << 20)], r0)
t_rows = time.perf_counter() - t0
# (a codes-only shard has no download: its table is compared through the flat ADC scan of a few queries)
q = x[:64] + 0.01
ia, da, _ = full.pq_scan_topk(q, 10)
ib, db, _ = sh.pq_scan_topk(q, 10)
print("ENCODE N", N, "D", D, "m", m, "stored_rows_per_s %.0f" % (N / t_stored), "streamed_rows_per_s_incl_upload %.0f" % (N / t_rows),
      "codes_sha1", hashlib.sha1(ca.tobytes()).hexdigest()[:16], "streamed_scan_equal", bool(np.array_equal(ia, ib) and np.array_equal(da, db)))
