"""Stress (GPU box): 12 threads on ONE handle mixing blocking calls of 1 ... 3000 queries (the direct result slab, variant 17 / 13, "ask later" lists)
with submit + wait, every answer compared with a single-threaded reference of the same queries. -> prints a summary, exits non-zero on a mismatch"""
import sys, threading
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(300000, 128, n_queries=6000, n_clusters=512, seed=11, query_seed=12)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=80, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
ix.pq_encode(ix.pq_train(32, n_sample=50000, iters=3))
cfgs = [(5, 20, 8), (10, 100, 8), (10, 48, 0)]
ref = {c: ix.search_batch(q, c[0], L=c[1], beam_width=c[2], mode=_ffi.MODE_M1) for c in cfgs}
bad = []
def worker(t):
    rs = np.random.RandomState(100 + t)
    for it in range(60):
        c = cfgs[rs.randint(len(cfgs))]
        n = int(rs.choice([1, 1, 2, 7, 64, 300, 1500, 3000]))
        a = int(rs.randint(0, len(q) - n))
        if rs.rand() < 0.5: out = ix.search_batch(q[a:a + n], c[0], L=c[1], beam_width=c[2], mode=_ffi.MODE_M1)
        else: out = ix.search_submit(q[a:a + n], c[0], L=c[1], beam_width=c[2], mode=_ffi.MODE_M1).wait()
        r = ref[c]
        ok = np.array_equal(out[0], r[0][a:a + n]) and np.array_equal(out[1].view(np.uint32), r[1][a:a + n].view(np.uint32)) and \
             all(np.array_equal(out[3][f], r[3][f][a:a + n]) for f in ("steps", "visited", "exact", "pq", "status"))
        if not ok: bad.append((t, it, c, n, a))
th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
for t in th: t.start()
for t in th: t.join()
print("mixed blocking / pipelined calls from 12 threads:", "all %d calls equal the single-threaded answers" % (12 * 60) if not bad else "MISMATCH %s" % bad[:5])
sys.exit(1 if bad else 0)
