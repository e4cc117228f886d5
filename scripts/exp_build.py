"""Experiment (GPU box): recall/QPS at 1M for builder settings."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import recall_at_k, sift_like

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
x, q = sift_like(n, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
ix0 = HipIndex.create_empty(x, R=64)
gt, _ = ix0.bruteforce_topk(q, 10)
ix0.close()
for (Lb, alpha, passes, mb) in [(100, 1.2, 2, 32768), (128, 1.2, 2, 32768), (160, 1.2, 2, 32768), (100, 1.2, 3, 32768), (100, 1.2, 2, 8192), (128, 1.3, 2, 32768)]:
    ix = HipIndex.create_empty(x, R=64)
    med, secs = ix.build_vamana(L_build=Lb, alpha=alpha, passes=passes, seed=7, pad_with_zero=True, max_batch=mb)
    cb = ix.pq_train(32, n_sample=20000, iters=3)
    ix.pq_encode(cb)
    adj = ix.get_adjacency()
    deg = (adj != 0).sum(1).mean()
    res = []
    for L, bw in ((100, 0), (100, 8)):
        ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1)
        res.append(f"L={L},bw={bw}: recall={recall_at_k(ids, gt):.4f} steps={st['steps'].mean():.0f} vis={st['visited'].mean():.0f} ins={st['inserts'].mean():.0f} ms={ix.timing()['search_kernel_ms']:.2f}")
    print(f"Lb={Lb} alpha={alpha} passes={passes} maxbatch={mb}: build={secs:.1f}s deg={deg:.1f} | " + " | ".join(res), flush=True)
    ix.close()
