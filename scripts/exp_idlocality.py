"""Experiment (GPU box): how much does id LOCALITY buy? Same data, ids either in generation (random) order or sorted by
nearest pivot, so that graph neighbours have nearby ids: visited-bitmap lines, DRAM pages and TLB entries get reused."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like, recall_at_k
x, q = sift_like(int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
rs = np.random.RandomState(1)
piv = x[rs.choice(len(x), 4096, replace=False)]
lab = np.empty(len(x), dtype=np.int64)
pp = (piv * piv).sum(1)[None, :]
for s in range(0, len(x), 65536):
    xs = x[s:s + 65536]
    lab[s:s + 65536] = (pp - 2.0 * xs @ piv.T).argmin(1)
order = np.argsort(lab, kind="stable")
for tag, data in (("generation order", x), ("sorted by nearest of 4096 pivots", x[order])):
    ix = HipIndex.create_empty(data, R=64)
    ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
    cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
    gt, _ = ix.bruteforce_topk(q, 10)
    for bw in (8, 0):
        ix.batch_upload(q)
        for _ in range(3): ix.batch_run(10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
        ix.batch_sync()
        ts = []
        for _ in range(10):
            ix.batch_run(10, L=100, beam_width=bw, mode=_ffi.MODE_M1); ix.batch_sync(); ts.append(ix.timing()["search_kernel_ms"])
        ids, dist, cnt, st = ix.batch_download()
        print(f"{tag:34s} bw={bw} kernel_ms min {min(ts):.3f} med {sorted(ts)[5]:.3f} steps {st['steps'].mean():.1f} exact {st['exact'].mean():.0f} recall {recall_at_k(ids, gt, 10):.4f}")
    ix.close()
