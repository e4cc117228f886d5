"""40 blocking dr_search_batch calls of 10 000 queries (SURVEY 8d's literal metric) and 300 one-query calls at the API defaults on the bench index --
the program scripts/profile_run_r05b.sh puts under rocprofv3 --kernel-trace --stats (search / finalize / bound kernels of a blocking call). -> JSON"""
import json
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=40000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
ts = []
for i in range(44):
    src = np.array(q[(i % 4) * 10000:(i % 4 + 1) * 10000])
    t0 = time.perf_counter()
    ix.search_batch(src, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[4:])
one = []
for i in range(330):
    t0 = time.perf_counter()
    ix.search_batch(q[i:i + 1], 5, L=20, beam_width=8, mode=_ffi.MODE_M1)
    one.append(time.perf_counter() - t0)
print(json.dumps({"blocking_call_10000_queries_ms_median": round(ts[len(ts) // 2] * 1e3, 4), "qps": round(10000 / ts[len(ts) // 2]),
                  "one_query_api_default_p50_ms": round(float(np.percentile(np.array(one[30:]) * 1e3, 50)), 4)}))
