"""Graph-sharded search (c5 layout) on one device: four shards with their own device-built sub-graphs, PQ-only
traversal (M3, the reference's beam_search_with_pq) per shard, canonical merge. Anchor: the merged
result equals the merge of the per-shard ORACLE runs on the same sub-graphs, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_four_shards_equal_the_merge_of_per_shard_oracle_runs():
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.parallel import merge_topk, shard_slice
    from diskrag_amd.sharded import GraphShard, ShardedSearch, globalize
    from diskrag_amd.synth import sift_like, recall_at_k
    from oracle import pyoracle as orc

    n, nshard, k = 32000, 4, 10
    x, q = sift_like(n, 128, n_queries=96, n_clusters=64, seed=21, query_seed=22)
    shards, parts = [], []
    cb = None
    for s in range(nshard):
        sl = shard_slice(n, nshard, s)
        ix = HipIndex.create_empty(x[sl], R=32)
        medoid, _ = ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=5 + s, pad_with_zero=False)
        if cb is None:
            cb = ix.pq_train(16, n_sample=8000, iters=4)          # one codebook for the whole id space
        codes = ix.pq_encode(cb, want_codes=True)
        shards.append(GraphShard(ix, sl.start))
        parts.append((sl, medoid, ix.get_adjacency(), codes))
    eng = ShardedSearch(shards)
    for bw, kk in ((8, k), (16, 5)):
        ids, dist, stats = eng.search_batch(q, kk, L=kk, beam_width=bw)
        o_ids, o_dist = [], []
        for (sl, medoid, adj, codes) in parts:
            oi, od, oc, ost = orc.search_batch(x[sl], adj, q, medoid, orc.M3, kk, L=kk, bw=bw, flags=orc.F_USE_PQ,
                                               codes=codes, codebook=cb)
            o_ids.append(globalize(oi, sl.start)); o_dist.append(od)
        w_ids, w_dist = merge_topk(o_ids, o_dist, kk)
        assert np.array_equal(ids, w_ids)
        assert np.array_equal(dist.view(np.uint32), w_dist.view(np.uint32))
        # canonical order and global id range
        valid = ids != 0xFFFFFFFF
        assert (ids[valid] < n).all()
        d = np.where(valid, dist, np.inf)
        assert (np.diff(d, axis=1) >= 0).all()
    # PQ-only traversal with a k-sized result heap (M3's rule, Q9 trim) is a coarse search: the merged lists only
    # have to be far better than chance (10 of 32000) -- recall, not parity, is all SURVEY 8e promises across shards
    ids, dist, _ = eng.search_batch(q, k, L=k, beam_width=32)
    gt = orc.bruteforce_topk(x, q, k)
    assert recall_at_k(ids, gt, k) > 0.15
    for sh in shards:
        sh.index.close()


def test_pq_only_shard_without_vectors():
    """c5's shards hold no vectors. A codes-only index serves the PQ-only traversal bit for bit like the full index
    (and the reference golden), and refuses what needs a stored vector."""
    from diskrag_amd import DiskragHipError, HipIndex, _ffi
    from tests.conftest import load_golden
    from tests.test_gpu_parity import get_index
    name = "randn128_R16_m32"
    g = load_golden(name)
    ci = [i for i, c in enumerate(g.cases) if c["mode"] == "M3" and c.get("use_pq")][0]
    c = g.case(ci)
    full = get_index(name, mem=True)
    want = full.search_batch(c["queries"], c["k"], L=c["k"], beam_width=c["bw"], mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
    assert np.array_equal(want[0], c["ids"])                                   # the reference's own output
    shard = HipIndex.create_codes(g.mem_adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes)
    got = shard.search_batch(c["queries"], c["k"], L=c["k"], beam_width=c["bw"], mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    assert np.array_equal(shard.adc(g.queries[:3], np.arange(40, dtype=np.uint32)), full.adc(g.queries[:3], np.arange(40, dtype=np.uint32)))
    for call in (lambda: shard.search_batch(g.queries, 10, L=50, mode=_ffi.MODE_M1),
                 lambda: shard.search_batch(g.queries, 10, beam_width=8, mode=_ffi.MODE_M2),
                 lambda: shard.search_batch(g.queries, 5, beam_width=8, mode=_ffi.MODE_M3),           # exact M3
                 lambda: shard.search_batch_f64(g.queries.astype(np.float64), 10, L=50),
                 lambda: shard.exact_distances(g.queries[:1], np.arange(4, dtype=np.uint32)),
                 lambda: shard.bruteforce_topk(g.queries[:1], 5), lambda: shard.get_node(0),
                 lambda: shard.build_vamana(), lambda: shard.pq_train(32)):
        with pytest.raises(DiskragHipError):
            call()
    shard.close()
    # a device-built index turned into a shard: same PQ-only answers before and after the vectors are freed
    from diskrag_amd.synth import sift_like
    x, q = sift_like(20000, 128, n_queries=64, n_clusters=64, seed=41, query_seed=42)
    ix = HipIndex.create_empty(x, R=32)
    ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=9, pad_with_zero=False)
    ix.pq_encode(ix.pq_train(16, n_sample=8000, iters=4))
    before = ix.search_batch(q, 10, L=10, beam_width=16, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
    ix.drop_vectors()
    after = ix.search_batch(q, 10, L=10, beam_width=16, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1].view(np.uint32), after[1].view(np.uint32))
    with pytest.raises(DiskragHipError):
        ix.search_batch(q, 10, L=50, mode=_ffi.MODE_M1)
    ix.close()
