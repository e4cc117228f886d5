"""dr_sharded_search with several LOCAL shards on one GPU: S PQ-only shards of n points each, every query on
every shard, device merge, one-rank RCCL exchange. Round 2 searched the shards one after another with a stream sync each and
allocated nine arrays per call; round 3 queues them on their own streams and keeps the work area. usage: DR_LIB=... ab_sharded.py [S] [n]"""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
shards, bases = [], []
q = None
for s in range(S):
    x, qq = sift_like(n, 128, n_queries=10000, n_clusters=1024, seed=3000 + s, query_seed=77)
    q = qq if q is None else q
    ix = HipIndex.create_empty(x, R=64)
    ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=False)
    ix.pq_encode(ix.pq_train(32, n_sample=100000, iters=8, seed=42))
    ix.drop_vectors()
    shards.append(ix); bases.append(s * n)
comm = _ffi.Comm(_ffi.Comm.unique_id(), 1, 0, 0)
for use_comm in (comm, None):
    for _ in range(3):
        _ffi.sharded_search(shards, bases, q, 10, L=100, beam_width=8, mode=_ffi.MODE_PQ, comm=use_comm)
    t0 = time.perf_counter()
    ms = np.zeros(3)
    for _ in range(10):
        ids, dist, status, m3 = _ffi.sharded_search(shards, bases, q, 10, L=100, beam_width=8, mode=_ffi.MODE_PQ, comm=use_comm)
        ms += m3
    el = (time.perf_counter() - t0) / 10
    chk = int(np.bitwise_xor.reduce(ids.astype(np.uint64).ravel() * np.uint64(0x9E3779B97F4A7C15) + dist.view(np.uint32).astype(np.uint64).ravel()))
    print(f"{S} local shards x {n}, 10000 queries, {'one-rank RCCL' if use_comm else 'no communicator'}: {el * 1e3:.3f} ms per call ({10000 / el:.0f} QPS), "
          f"search {ms[0] / 10:.3f} gather {ms[1] / 10:.3f} merge {ms[2] / 10:.3f} ms, status {int(status.max())} chk {chk:016x}", flush=True)
