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


def test_pipelined_sharded_submits_and_the_failure_protocol():
    """Round 4: dr_sharded_submit / dr_sharded_wait -- two batches in flight, any wait order, more submits
    than work areas -- return the bits of the blocking call; the lists travel as packed keys + a status word in ONE
    all-gather (a real one-rank RCCL communicator here); a rank whose shard cannot run the mode reports the error through
    the exchange instead of leaving it, and the handles keep working."""
    import pytest
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.parallel import shard_slice
    from diskrag_amd.synth import sift_like
    n, nshard, k = 12000, 3, 10
    x, q = sift_like(n, 128, n_queries=120, n_clusters=32, seed=31, query_seed=32)
    shards, bases, cb = [], [], None
    for s in range(nshard):
        sl = shard_slice(n, nshard, s)
        ix = HipIndex.create_empty(x[sl], R=32)
        ix.build_vamana(L_build=50, alpha=1.2, passes=2, seed=9 + s, pad_with_zero=False)
        if cb is None:
            cb = ix.pq_train(16, n_sample=4000, iters=4)
        ix.pq_encode(cb)
        shards.append(ix); bases.append(sl.start)
    nopq = HipIndex.create_empty(x[:4000], R=32)           # a shard that was never encoded: DR_MODE_PQ fails on it (DR_E_NOPQ)
    nopq.build_vamana(L_build=50, alpha=1.2, passes=2, seed=3, pad_with_zero=False)
    comm = _ffi.Comm(_ffi.Comm.unique_id(), 1, 0, 0)
    try:
        batches = [np.ascontiguousarray(q[a:b]) for a, b in ((0, 120), (0, 7), (7, 60), (60, 61), (30, 120))]
        for c in (comm, None):
            want = [_ffi.sharded_search(shards, bases, b, k, L=60, beam_width=8, mode=_ffi.MODE_PQ, comm=c) for b in batches]
            jobs = [_ffi.sharded_submit(shards, bases, b, k, L=60, beam_width=8, mode=_ffi.MODE_PQ, comm=c) for b in batches]
            for i in (4, 0, 2, 1, 3):
                ids, dist, status, ms = jobs[i].wait()
                assert np.array_equal(ids, want[i][0]) and np.array_equal(dist.view(np.uint32), want[i][1].view(np.uint32))
                assert np.array_equal(status, want[i][2])
            # the local phase fails on the second shard: the call comes back with THAT error (the exchange still ran) ...
            with pytest.raises(_ffi.DiskragHipError) as ei:
                _ffi.sharded_search([shards[0], nopq, shards[2]], [0, 4000, 8000], batches[0], k, L=60, beam_width=8, mode=_ffi.MODE_PQ, comm=c)
            assert ei.value.code == _ffi.E_NOPQ
            # ... a failing submit between two good ones does not disturb them ...
            j0 = _ffi.sharded_submit(shards, bases, batches[2], k, L=60, beam_width=8, mode=_ffi.MODE_PQ, comm=c)
            jbad = _ffi.sharded_submit([nopq] + shards[1:], bases, batches[2], k, L=60, beam_width=8, mode=_ffi.MODE_PQ, comm=c)
            j1 = _ffi.sharded_submit(shards, bases, batches[4], k, L=60, beam_width=8, mode=_ffi.MODE_PQ, comm=c)
            assert np.array_equal(j1.wait()[0], want[4][0])
            assert np.array_equal(j0.wait()[0], want[2][0])
            with pytest.raises(_ffi.DiskragHipError):
                jbad.wait()
            # ... and everything still answers afterwards, the shard that had work queued when the call failed included
            ids, dist, _, _ = _ffi.sharded_search(shards, bases, batches[0], k, L=60, beam_width=8, mode=_ffi.MODE_PQ, comm=c)
            assert np.array_equal(ids, want[0][0]) and np.array_equal(dist.view(np.uint32), want[0][1].view(np.uint32))
            one = shards[0].search_batch(batches[1], k, L=60, beam_width=8, mode=_ffi.MODE_PQ)
            assert (one[3]["status"] == 0).all()
    finally:
        comm.close()
        nopq.close()
        for ix in shards:
            ix.close()


def test_grouped_exchanges_carry_several_submits():
    """End of round 4: dr_sharded_set_group(n) -- n consecutive submits ride in ONE exchange (one launch per shard over their
    concatenated queries, one all-gather); the rule is a count, a wait or a flush, never a timing. Every ticket gets the bits of a
    blocking call of its own; parameters that differ split an exchange; a failing local phase fails every ticket of its exchange
    and nothing else."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.parallel import shard_slice
    from diskrag_amd.synth import sift_like
    n, nshard, k = 12000, 3, 10
    x, q = sift_like(n, 128, n_queries=120, n_clusters=32, seed=41, query_seed=42)
    shards, bases, cb = [], [], None
    for s in range(nshard):
        sl = shard_slice(n, nshard, s)
        ix = HipIndex.create_empty(x[sl], R=32)
        ix.build_vamana(L_build=50, alpha=1.2, passes=2, seed=19 + s, pad_with_zero=False)
        if cb is None:
            cb = ix.pq_train(16, n_sample=4000, iters=4)
        ix.pq_encode(cb)
        shards.append(ix); bases.append(sl.start)
    nopq = HipIndex.create_empty(x[:4000], R=32)
    nopq.build_vamana(L_build=50, alpha=1.2, passes=2, seed=3, pad_with_zero=False)
    comm = _ffi.Comm(_ffi.Comm.unique_id(), 1, 0, 0)
    kw = dict(L=60, beam_width=8, mode=_ffi.MODE_PQ)

    def same(got, want):
        return (np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)) and np.array_equal(got[2], want[2]))
    try:
        batches = [np.ascontiguousarray(q[a:b]) for a, b in ((0, 120), (0, 7), (7, 60), (60, 61), (30, 120))]
        for c in (comm, None):
            want = [_ffi.sharded_search(shards, bases, b, k, comm=c, **kw) for b in batches]
            want40 = _ffi.sharded_search(shards, bases, batches[2], k, L=40, beam_width=8, mode=_ffi.MODE_PQ, comm=c)
            _ffi.sharded_set_group(shards[0], 3)
            # a blocking call under a group size of 3: its wait launches the exchange it rides in alone
            assert same(_ffi.sharded_search(shards, bases, batches[0], k, comm=c, **kw), want[0])
            # five submits: three fill the first exchange, the wait for the fifth launches the second (two submits)
            jobs = [_ffi.sharded_submit(shards, bases, b, k, comm=c, **kw) for b in batches]
            got = {}
            for i in (4, 0, 3, 2, 1):
                got[i] = jobs[i].wait()
                assert same(got[i], want[i]), i
            assert np.array_equal(got[0][3], got[1][3]) and np.array_equal(got[0][3], got[2][3])        # one exchange: the same three phase times
            assert np.array_equal(got[3][3], got[4][3]) and not np.array_equal(got[0][3], got[3][3])
            # other parameters close the collecting exchange; a flush launches what is held
            ja = _ffi.sharded_submit(shards, bases, batches[2], k, comm=c, **kw)
            jb = _ffi.sharded_submit(shards, bases, batches[2], k, L=40, beam_width=8, mode=_ffi.MODE_PQ, comm=c)
            _ffi.sharded_flush(shards[0])
            assert same(jb.wait(), want40) and same(ja.wait(), want[2])
            # more submits than exchanges in flight (4 x 3 tickets): the oldest are finished by the library, every wait still answers
            many = [_ffi.sharded_submit(shards, bases, batches[i % 5], k, comm=c, **kw) for i in range(17)]
            for i in reversed(range(17)):
                assert same(many[i].wait(), want[i % 5]), i
            # a local phase that fails (a shard without PQ data): every ticket of THAT exchange answers the error, the next exchange is fine
            bad = [shards[0], nopq, shards[2]]
            j0 = _ffi.sharded_submit(bad, [0, 4000, 8000], batches[1], k, comm=c, **kw)
            j1 = _ffi.sharded_submit(bad, [0, 4000, 8000], batches[3], k, comm=c, **kw)
            j2 = _ffi.sharded_submit(shards, bases, batches[4], k, comm=c, **kw)        # (other shards: closes the failing exchange, opens its own)
            assert same(j2.wait(), want[4])
            for j in (j1, j0):
                with pytest.raises(_ffi.DiskragHipError) as ei:
                    j.wait()
                assert ei.value.code == _ffi.E_NOPQ
            _ffi.sharded_set_group(shards[0], 1)
            assert same(_ffi.sharded_submit(shards, bases, batches[0], k, comm=c, **kw).wait(), want[0])
        # round 6: an exchange holds up to 65536 queries (32768 before): five 12000-query submits ride in ONE launch per shard, a sixth
        # would not fit next to them and opens the next exchange; every ticket still gets the bits of its own rows
        big = np.ascontiguousarray(np.tile(q, (100, 1)))
        wantb = _ffi.sharded_search(shards, bases, big, k, comm=None, **kw)
        _ffi.sharded_set_group(shards[0], 6)
        jobs = [_ffi.sharded_submit(shards, bases, big, k, comm=None, **kw) for _ in range(6)]
        got = [j.wait() for j in jobs]
        for g in got:
            assert same(g, wantb)
        assert np.array_equal(got[0][3], got[4][3]) and not np.array_equal(got[0][3], got[5][3])      # five of 12000 in one exchange (60000 <= 65536), the sixth alone
        _ffi.sharded_set_group(shards[0], 1)
        with pytest.raises(_ffi.DiskragHipError):
            _ffi.sharded_set_group(shards[0], 17)
    finally:
        comm.close()
        nopq.close()
        for ix in shards:
            ix.close()
