"""N > 1 host path on CPU: two gloo ranks exercise query sharding, the graph-sharded top-k all-gather + merge and
the max-over-ranks timing, checked against single-process oracle runs on a golden fixture. The torch.distributed side is
tests/gloo_twin.py (test infrastructure); the numpy statements under test are diskrag_amd/parallel.py."""
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
    from tests import gloo_twin
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
        all_ids = gloo_twin.gather_rows(ids)
        # --- graph-sharded: each rank owns half of the ids and searches ALL queries by brute force over its half
        n = len(g.vectors)
        own = parallel.shard_slice(n, world, rank)
        lids = orc.bruteforce_topk(g.vectors[own], g.queries, k)
        ldist = np.array([[orc.sqdist(g.vectors[own][i], q) for i in row] for row, q in zip(lids, g.queries)],
                         dtype=np.float32)
        mids, mdist = gloo_twin.allgather_merge_topk(lids, ldist, own.start, k)
        slow = gloo_twin.max_over_ranks(1.0 + rank)
        # --- the graph-sharded driver itself, two shards per rank, with the oracle standing in for the device index
        from diskrag_amd.sharded import GraphShard
        from tests.gloo_twin import HostShardedSearch as ShardedSearch

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


def test_packed_keys_order_and_round_trip():
    """the form the lists travel in (ONE all-gather of 64-bit words): order = (distance, id), distances survive bit for bit"""
    from diskrag_amd.parallel import PAD, merge_topk, pack_keys, unpack_keys
    rs = np.random.RandomState(4)
    d = np.abs(rs.randn(6, 9)).astype(np.float32)
    d[0, 0] = np.nan; d[1, 1] = np.inf; d[2, 2] = 0.0; d[3, 3] = d[3, 4]            # a tie on distance: id decides
    ids = rs.randint(0, 2 ** 32 - 2, size=(6, 9)).astype(np.uint32)
    ids[4, 4] = PAD
    keys = pack_keys(ids, d)
    i2, d2 = unpack_keys(keys)
    valid = ~np.isnan(d) & (d != np.inf) & (ids != PAD)
    assert np.array_equal(i2[valid], ids[valid]) and np.array_equal(d2[valid].view(np.uint32), d[valid].view(np.uint32))
    assert (i2[~valid] == PAD).all() and np.isnan(d2[~valid]).all()
    # sorting the keys IS the canonical merge
    want_ids, want_d = merge_topk([ids], [d], 9)
    si, sd = unpack_keys(np.sort(keys, axis=1))
    assert np.array_equal(si, want_ids) and np.array_equal(sd.view(np.uint32), want_d.view(np.uint32))
    assert pack_keys(np.array([[3]], np.uint32), np.array([[-0.0]], np.float32)) == pack_keys(np.array([[3]], np.uint32), np.array([[0.0]], np.float32))


def _failing_worker(rank, world, port, ret):
    """rank 1's shard raises in its local phase: BOTH ranks must come back (with an error), nobody blocks in the collective;
    the next call, with healthy shards, works -- the protocol left nothing behind"""
    import torch.distributed as dist
    from diskrag_amd import parallel
    from diskrag_amd.sharded import GraphShard
    from tests.gloo_twin import HostShardedSearch as ShardedSearch
    from oracle import pyoracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = load_golden("sift128_R64_m32")
        n, k = len(g.vectors), 5
        own = parallel.shard_slice(n, world, rank)

        class Shard:
            def __init__(self, vecs, broken):
                self.v, self.broken = vecs, broken

            def search_batch(self, queries, k, L=100, beam_width=0, mode=0, band_policy=0, flags=0):
                if self.broken:
                    raise RuntimeError("shard without PQ data")
                li = orc.bruteforce_topk(self.v, queries, k)
                ld = np.array([[orc.sqdist(self.v[i], q) for i in row] for row, q in zip(li, queries)], dtype=np.float32)
                return li, ld, np.full(len(queries), k, dtype=np.uint32), np.zeros(len(queries), dtype=[("status", np.uint32)])

        q = g.queries[:8]
        outcome = []
        for broken in (rank == 1, False, rank == 0):
            eng = ShardedSearch([GraphShard(Shard(g.vectors[own], broken), own.start)], group=dist.group.WORLD)
            try:
                ids, d, _ = eng.search_batch(q, k)
                outcome.append(("ok", ids))
            except parallel.ShardExchangeError as e:
                outcome.append(("remote", e.statuses))
            except RuntimeError as e:
                outcome.append(("local", str(e)))
        ret[rank] = outcome
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_fails_the_call_on_every_rank_and_hangs_nobody():
    import torch.multiprocessing as mp
    from oracle import pyoracle as orc
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "a rank hung or died"
    g = load_golden("sift128_R64_m32")
    gt = orc.bruteforce_topk(g.vectors, g.queries[:8], 5)
    r0, r1 = ret[0], ret[1]
    assert r0[0] == ("remote", [0, 1]) and r1[0] == ("local", "shard without PQ data")      # call 1: rank 1 failed
    assert r0[1][0] == "ok" and r1[1][0] == "ok"                                            # call 2: healthy
    assert np.array_equal(r0[1][1], r1[1][1])
    assert all(set(a.tolist()) == set(b.tolist()) for a, b in zip(r0[1][1], gt))            # (sets: ties may order differently from the brute force)
    assert r0[2] == ("local", "shard without PQ data") and r1[2] == ("remote", [1, 0])      # call 3: rank 0 failed


def test_the_product_package_imports_no_torch():
    """diskrag_amd/ is numpy + ctypes: the gloo twin of the exchange lives under tests/ """
    import re
    from pathlib import Path
    pkg = Path(__file__).resolve().parent.parent / "diskrag_amd"
    for f in pkg.glob("*.py"):
        assert not re.search(r"^\s*(import|from)\s+torch", f.read_text(), re.M), f.name
