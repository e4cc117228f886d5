"""N > 1 host path on CPU: two gloo ranks exercise query sharding, the graph-sharded top-k all-gather + merge and
the max-over-ranks timing, checked against single-process oracle runs on a golden fixture."""
import os
import socket

import numpy as np
import pytest

from tests.conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    from diskrag_amd import parallel
    from oracle import pyoracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("sift128_R64_m32")
        k = 10
        # --- query-sharded replicas: each rank searches its slice with the whole index, rows are concatenated
        sl = parallel.shard_slice(len(g.queries), world, rank)
        ids, dist_, cnt, _ = orc.search_batch(g.vectors, g.adj, g.queries[sl], g.medoid, orc.M1, k, L=50, bw=0,
                                              codes=g.codes, codebook=g.codebook)
        all_ids = parallel.gather_rows(ids)
        # --- graph-sharded: each rank owns half of the ids and searches ALL queries by brute force over its half
        n = len(g.vectors)
        own = parallel.shard_slice(n, world, rank)
        lids = orc.bruteforce_topk(g.vectors[own], g.queries, k)
        ldist = np.array([[orc.sqdist(g.vectors[own][i], q) for i in row] for row, q in zip(lids, g.queries)],
                         dtype=np.float32)
        mids, mdist = parallel.allgather_merge_topk(lids, ldist, own.start, k)
        slow = parallel.max_over_ranks(1.0 + rank)
        # --- the graph-sharded driver itself, two shards per rank, with the oracle standing in for the device index
        from diskrag_amd.sharded import GraphShard, ShardedSearch

        class OracleShard:   # HipIndex.search_batch's signature over a brute-force shard (host logic under test)
            def __init__(self, vecs):
                self.v = vecs

            def search_batch(self, queries, k, L=100, beam_width=0, mode=0, band_policy=0, flags=0):
                li = orc.bruteforce_topk(self.v, queries, k)
                ld = np.array([[orc.sqdist(self.v[i], q) for i in row] for row, q in zip(li, queries)], dtype=np.float32)
                st = np.zeros(len(queries), dtype=[("status", np.uint32)])
                return li, ld, np.full(len(queries), k, dtype=np.uint32), st

        quarters = [parallel.shard_slice(n, 2 * world, 2 * rank + j) for j in range(2)]
        eng = ShardedSearch([GraphShard(OracleShard(g.vectors[sl]), sl.start) for sl in quarters], group=dist.group.WORLD)
        sids, sdist, _ = eng.search_batch(g.queries, k)
        ret[rank] = (all_ids, mids, mdist, slow, sids, sdist)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_merge():
    import torch.multiprocessing as mp
    from oracle import pyoracle as orc
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    g = load_golden("sift128_R64_m32")
    want_ids, _, _, _ = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M1, 10, L=50, bw=0,
                                         codes=g.codes, codebook=g.codebook)
    gt = orc.bruteforce_topk(g.vectors, g.queries, 10)
    gt_d = np.array([[orc.sqdist(g.vectors[i], q) for i in row] for row, q in zip(gt, g.queries)], dtype=np.float32)
    for r in range(2):
        all_ids, mids, mdist, slow, sids, sdist = ret[r]
        # 2 ranks x 2 local shards through ShardedSearch == the 2-shard exchange == brute force over everything
        assert np.array_equal(sids, mids) and np.array_equal(sdist.view(np.uint32), mdist.view(np.uint32))
        assert np.array_equal(all_ids, want_ids)                  # query-sharded == unsharded
        assert np.array_equal(np.sort(mdist, axis=1).view(np.uint32), np.sort(gt_d, axis=1).view(np.uint32))
        assert slow == 2.0                                        # max over ranks
        # merged lists are in canonical (distance, id) order
        for row_i, row_d in zip(mids, mdist):
            keys = list(zip(row_d.tolist(), row_i.tolist()))
            assert keys == sorted(keys)


def test_shard_slice_and_merge_padding():
    from diskrag_amd.parallel import PAD, merge_topk, shard_slice
    sizes = [shard_slice(10, 4, r) for r in range(4)]
    assert [s.stop - s.start for s in sizes] == [3, 3, 2, 2] and sizes[0].start == 0 and sizes[-1].stop == 10
    a_ids = np.array([[5, 7, PAD]], dtype=np.uint32); a_d = np.array([[1.0, 3.0, np.nan]], dtype=np.float32)
    b_ids = np.array([[9, PAD, PAD]], dtype=np.uint32); b_d = np.array([[1.0, np.nan, np.nan]], dtype=np.float32)
    ids, d = merge_topk([a_ids, b_ids], [a_d, b_d], 4)
    assert ids.tolist() == [[5, 9, 7, int(PAD)]]
    assert d[0, :3].tolist() == [1.0, 1.0, 3.0] and np.isnan(d[0, 3])


def test_distance_based_recall_counts_ties():
    """recall_at_k_ties (diskrag_amd/synth.py): a returned entry as near as the k-th ground-truth entry is a hit, whatever its id."""
    from diskrag_amd.synth import recall_at_k, recall_at_k_ties
    gt_ids = np.array([[0, 1, 2]], dtype=np.uint32)
    gt_dist = np.array([[1.0, 2.0, 2.0]], dtype=np.float32)
    got_ids = np.array([[0, 7, 9]], dtype=np.uint32)             # 7 and 9 tie with the 2nd / 3rd ground-truth entries
    got_dist = np.array([[1.0, 2.0, 2.0]], dtype=np.float32)
    assert recall_at_k(got_ids, gt_ids, 3) == 1 / 3
    assert recall_at_k_ties(got_dist, gt_dist, 3) == 1.0
    assert recall_at_k_ties(np.array([[1.0, 2.5, np.nan]], dtype=np.float32), gt_dist, 3) == 1 / 3


def test_strong_scaling_slices_tile_any_batch():
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for nq in (1, 7, 1250, 10000, 10001):
        for world in (1, 2, 3, 8):
            sl = [bench.slice_of(nq, world, r) for r in range(world)]
            assert sl[0][0] == 0 and sl[-1][1] == nq and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert max(h - l for l, h in sl) - min(h - l for l, h in sl) <= 1
