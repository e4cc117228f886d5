"""Round-2 entry points through the C ABI, on an MI355X: pipelined batches, several resident batches, the engine's
PQ-only traversal (DR_MODE_PQ, checked against the oracle's restatement of it) with and without the exact rerank,
result lists up to 1024 entries, the device merge kernel, the sharded search with its RCCL exchange, and the PQ
encoder against the reference-produced codes of every golden fixture."""
import os
import numpy as np
import pytest

from tests.conftest import INDEX_FIXTURES, load_golden
from tests.test_gpu_parity import PQ_FIXTURES, bits, get_index

pytestmark = pytest.mark.gpu


def _stats4(st):
    return np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1)


@pytest.mark.parametrize("name", ["sift128_R64_m32", "unit1536_R16_m32"])
def test_pipelined_submit_wait_equals_blocking_calls(name):
    from diskrag_amd import _ffi
    g = load_golden(name)
    ix = get_index(name)
    rs = np.random.RandomState(3)
    batches = [g.queries[rs.permutation(len(g.queries))[:n]] for n in (len(g.queries), 7, 1, 16, len(g.queries), 5, 9)]
    params = [dict(L=100, beam_width=8, mode=_ffi.MODE_M1), dict(L=30, beam_width=0, mode=_ffi.MODE_M1),
              dict(L=0, beam_width=8, mode=_ffi.MODE_M2), dict(L=100, beam_width=8, mode=_ffi.MODE_M1, band_policy=1)]
    want = [ix.search_batch(b, 10, **params[i % len(params)]) for i, b in enumerate(batches)]
    # pageable sources (staged by the library) and page-locked sources (read in place), more submits than pipeline slots
    for pinned in (False, True):
        srcs = []
        for b in batches:
            if pinned:
                a = _ffi.pinned_empty(b.shape, np.float32); a[:] = b
            else:
                a = np.array(b)
            srcs.append(a)
        jobs = [ix.search_submit(a, 10, **params[i % len(params)]) for i, a in enumerate(srcs)]
        if not pinned:
            for a in srcs:
                a[:] = -7.0              # pageable memory may be reused as soon as submit returns
        for j, w in zip(reversed(jobs), reversed(want)):      # any wait order
            ids, dist, cnt, st = j.wait()
            assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist), bits(w[1])) and np.array_equal(cnt, w[2])
            assert np.array_equal(_stats4(st), _stats4(w[3]))
    ix.batch_sync()


def test_a_dropped_pipelined_job_does_not_lose_its_buffers():
    """the library copies a batch's results into the caller's arrays when the batch is FINISHED (a later submit
    that reuses its slot, or wait). A PendingSearch dropped without wait() must keep those arrays alive until then."""
    import gc
    from diskrag_amd import _ffi
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    want = ix.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    for _ in range(8):
        ix.search_submit(np.array(g.queries), 10, L=100, beam_width=8, mode=_ffi.MODE_M1)      # dropped at once
        gc.collect()
        junk = [np.empty(len(g.queries) * 10, dtype=np.uint32) for _ in range(4)]              # churn the allocator
        del junk
    assert len(ix.__dict__["_inflight"]) <= _ffi.MAX_TICKETS
    j = ix.search_submit(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    ids, dist, cnt, st = j.wait()
    assert np.array_equal(ids, want[0]) and np.array_equal(bits(dist), bits(want[1]))
    ix.batch_sync()


def test_resident_batches_are_independent():
    from diskrag_amd import _ffi
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    qa, qb = g.queries[:12], g.queries[12:24][::-1].copy()
    wa = ix.search_batch(qa, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    wb = ix.search_batch(qb, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    ix.batch_select(3); ix.batch_upload(qa)
    ix.batch_select(9); ix.batch_upload(qb)
    for slot, w in ((3, wa), (9, wb), (3, wa)):
        ix.batch_select(slot)
        ix.batch_run(10, L=100, beam_width=8, mode=_ffi.MODE_M1)
        ids, dist, cnt, st = ix.batch_download()
        assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist), bits(w[1]))
    with pytest.raises(_ffi.DiskragHipError):
        ix.batch_select(16)
    ix.batch_select(0)


@pytest.mark.parametrize("name", ["randn128_R16_m32", "sift128_R64_m32", "unit1536_R16_m64", "deep96_R32_m16"])
def test_pq_mode_matches_its_oracle_restatement(name):
    """DR_MODE_PQ: M1's loop on squared ADC distances (no reference counterpart; oracle mode 5). Bit-exact ids, distance
    bits, counts and counters; then the exact rerank of the final list in (distance, id) order."""
    from diskrag_amd import HipIndex, _ffi
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    shard = HipIndex.create_codes(g.adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes)
    try:
        for (L, bw, k) in ((100, 8, 10), (40, 0, 10), (10, 3, 10), (200, 16, 25), (64, 8, 64)):
            w = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, codes=g.codes, codebook=g.codebook)
            for eng, inline in ((ix, False), (shard, False), (shard, True)):
                eng.inline_codes(inline)      # (True: the neighbours' code words read as one block beside the adjacency row)
                ids, dist, cnt, st = eng.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ)
                assert int(st["status"].max()) == 0
                assert np.array_equal(ids, w[0]), (name, L, bw, inline)
                valid = w[0] != 0xFFFFFFFF
                assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid])
                assert np.array_equal(cnt, w[2]) and np.array_equal(_stats4(st), w[3])
                if inline:      # the reference-faithful PQ traversal reads the same block
                    w3 = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M3, min(k, 10), L=min(k, 10), bw=8, flags=orc.F_USE_PQ,
                                          codes=g.codes, codebook=g.codebook)
                    i3, d3, c3, s3 = eng.search_batch(g.queries, min(k, 10), L=min(k, 10), beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
                    assert np.array_equal(i3, w3[0]) and np.array_equal(c3, w3[2]) and np.array_equal(_stats4(s3), w3[3])
            shard.inline_codes(False)
            wr = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, flags=orc.F_RERANK, codes=g.codes,
                                  codebook=g.codebook)
            ids, dist, cnt, st = ix.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
            assert np.array_equal(ids, wr[0]) and np.array_equal(cnt, wr[2]) and np.array_equal(_stats4(st), wr[3])
            valid = wr[0] != 0xFFFFFFFF
            assert np.array_equal(bits(dist)[valid], bits(wr[1].astype(np.float32))[valid])
        with pytest.raises(_ffi.DiskragHipError):        # the rerank needs the stored vectors
            shard.search_batch(g.queries, 10, L=50, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
    finally:
        shard.close()


@pytest.mark.parametrize("name", ["randn128_R16_m32", "sift128_R64_m32", "unit1536_R16_m32", "unit1536_R16_m64", "deep96_R32_m16", "unit768_R16_m96"])
def test_pq_mode_without_a_visited_set(name, monkeypatch):
    """DR_F_NO_VISITED_SET (round 4): the engine's PQ traversal with no visited words -- every neighbour scored, one that would
    enter the list looked up in the list. Bit-exact against the oracle's statement of the same rule (ids, distance bits, counts,
    evaluation counters), and the same ids / distances as the traversal WITH a visited set; with and without inline neighbour
    codes (their code words then leave with the adjacency row), the exact rerank, and the next row's ids prefetched into LDS."""
    from diskrag_amd import HipIndex, _ffi
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    shard = HipIndex.create_codes(g.adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes)
    try:
        for (L, bw, k) in ((100, 8, 10), (40, 0, 10), (10, 3, 10), (200, 16, 25), (64, 8, 64), (1, 1, 1)):
            w = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, flags=orc.F_NO_VISITED_SET, codes=g.codes, codebook=g.codebook)
            base = ix.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ)
            for eng, inline, pre in ((ix, False, False), (shard, False, False), (shard, True, False), (ix, True, True), (shard, False, True)):
                eng.inline_codes(inline)
                if pre: monkeypatch.setenv("DR_PQ_ROW_PREFETCH", "1")
                else: monkeypatch.delenv("DR_PQ_ROW_PREFETCH", raising=False)
                ids, dist, cnt, st = eng.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_NO_VISITED_SET)
                assert int(st["status"].max()) == 0
                assert np.array_equal(ids, w[0]) and np.array_equal(cnt, w[2]), (name, L, bw, inline, pre)
                valid = w[0] != 0xFFFFFFFF
                assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid])
                assert np.array_equal(_stats4(st), w[3]), (name, L, bw, inline, pre)
                assert np.array_equal(ids, base[0]) and np.array_equal(bits(dist)[valid], bits(base[1])[valid])      # == with a visited set
                if pre and bw != 1: assert st["adj_prefetch_hits"].sum() > 0
            monkeypatch.delenv("DR_PQ_ROW_PREFETCH", raising=False)
            ix.inline_codes(False); shard.inline_codes(False)
            wr = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, flags=orc.F_RERANK | orc.F_NO_VISITED_SET,
                                  codes=g.codes, codebook=g.codebook)
            ids, dist, cnt, st = ix.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK | _ffi.F_NO_VISITED_SET)
            assert np.array_equal(ids, wr[0]) and np.array_equal(cnt, wr[2]) and np.array_equal(_stats4(st), wr[3])
        with pytest.raises(_ffi.DiskragHipError):        # the flag belongs to DR_MODE_PQ
            ix.search_batch(g.queries, 10, L=50, beam_width=8, mode=_ffi.MODE_M1, flags=_ffi.F_NO_VISITED_SET)
    finally:
        shard.close()


@pytest.mark.parametrize("name", ["unit1536_R16_m32", "unit1536_R16_m64"])
def test_split_table_variant_returns_the_same_bits(name):
    """Variant 15 (round 3): the table rows of the last 16 sub-quantisers in registers (ds_bpermute lookups), the rest in LDS
    -- against variant 2 (whole table in LDS) and the oracle, for the engine's PQ traversal and the reference's M3 with PQ,
    on a full index and on a PQ-only shard, every list-size class."""
    from diskrag_amd import HipIndex, _ffi
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    shard = HipIndex.create_codes(g.adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes)
    try:
        for (L, bw, k) in ((100, 8, 10), (40, 0, 10), (200, 16, 25), (300, 0, 10), (600, 8, 10), (1024, 0, 10)):
            w = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQ, k, L=L, bw=bw, codes=g.codes, codebook=g.codebook)
            for eng in (ix, shard):
                for kind in (15, 2):
                    eng.debug_force_kind(kind)
                    ids, dist, cnt, st = eng.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ)
                    assert eng.timing()["variant"] == kind
                    assert int(st["status"].max()) == 0
                    valid = w[0] != 0xFFFFFFFF
                    assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid]), (name, L, bw, kind)
                    assert np.array_equal(cnt, w[2]) and np.array_equal(_stats4(st), w[3])
        w3 = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M3, 10, L=10, bw=8, flags=orc.F_USE_PQ, codes=g.codes, codebook=g.codebook)
        for kind in (15, 2):
            shard.debug_force_kind(kind)
            i3, d3, c3, s3 = shard.search_batch(g.queries, 10, L=10, beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
            assert shard.timing()["variant"] == kind
            assert np.array_equal(i3, w3[0]) and np.array_equal(c3, w3[2]) and np.array_equal(_stats4(s3), w3[3])
        shard.debug_force_kind(-1)
        shard.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_PQ)
        assert shard.timing()["variant"] == 15 and shard.timing()["lut_kernel_ms"] > 0.0      # the engine's own choice
    finally:
        ix.debug_force_kind(-1)
        shard.close()


@pytest.mark.parametrize("name", ["sift128_R64_m32", "unit1536_R16_m64", "deep96_R32_m16"])
def test_pq_scan_topk_equals_the_sorted_flat_scan(name):
    """dr_pq_scan_topk (brute-force ADC search, the ground truth of the PQ-only traversals) == the k smallest
    (distance, id) pairs of the full flat scan, whose sums are pinned on the reference's ADC bits."""
    g = load_golden(name)
    ix = get_index(name)
    full, _ = ix.pq_scan(g.queries)
    for k in (1, 10, 64):
        ids, sq, ms = ix.pq_scan_topk(g.queries, k)
        for qi in range(len(g.queries)):
            order = np.lexsort((np.arange(full.shape[1]), full[qi]))[:k]
            assert np.array_equal(ids[qi], order.astype(np.uint32)), (name, k, qi)
            assert np.array_equal(bits(sq[qi]), bits(full[qi][order]))
    # duplicated code words: ties go to the smaller id
    from diskrag_amd import HipIndex
    codes = np.repeat(g.codes[:50], 3, axis=0)
    sh = HipIndex.create_codes(np.zeros((150, 4), dtype=np.uint32), 0, g.vectors.shape[1], g.codebook, codes)
    try:
        ids, sq, _ = sh.pq_scan_topk(g.queries[:4], 12)
        f2, _ = sh.pq_scan(g.queries[:4])
        for qi in range(4):
            order = np.lexsort((np.arange(150), f2[qi]))[:12]
            assert np.array_equal(ids[qi], order.astype(np.uint32))
    finally:
        sh.close()


@pytest.mark.parametrize("name", ["randn128_R16_m32", "unit1536_R16_m32"])
def test_result_lists_up_to_1024_entries(name):
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    for (L, bw) in ((600, 0), (1024, 8), (513, 16)):
        w = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M1, 10, L=L, bw=bw, codes=g.codes, codebook=g.codebook)
        ids, dist, cnt, st = ix.search_batch(g.queries, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1)
        assert int(st["status"].max()) == 0
        assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist), bits(w[1].astype(np.float32))) and np.array_equal(_stats4(st), w[3])
    with pytest.raises(_ffi.DiskragHipError):
        ix.search_batch(g.queries, 10, L=1025, mode=_ffi.MODE_M1)


def test_deep_lists_on_a_large_index_do_not_overflow_the_insert_log():
    """L = 512 on a 100k-point index (insert counts grow with N): no status bits, results equal the oracle's."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like
    from oracle import pyoracle as orc
    x, q = sift_like(100000, 128, n_queries=64, n_clusters=128, seed=5, query_seed=6)
    ix = HipIndex.create_empty(x, R=64)
    try:
        medoid, _ = ix.build_vamana(L_build=80, alpha=1.2, passes=2, seed=3)
        cb = ix.pq_train(32, n_sample=20000, iters=3)
        codes = ix.pq_encode(cb, want_codes=True)
        adj = ix.get_adjacency()
        for bw in (0, 8):
            ids, dist, cnt, st = ix.search_batch(q, 10, L=512, beam_width=bw, mode=_ffi.MODE_M1)
            assert int(st["status"].max()) == 0
            w = orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=512, bw=bw, codes=codes, codebook=cb, nthreads=8)
            assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist), bits(w[1].astype(np.float32)))
            assert np.array_equal(_stats4(st), w[3])
    finally:
        ix.close()


def test_device_merge_kernel_equals_the_host_merge():
    from diskrag_amd import _ffi
    from diskrag_amd.parallel import merge_topk
    rs = np.random.RandomState(11)
    for (S, nq, k, k_out) in ((8, 257, 10, 10), (3, 40, 7, 12), (1, 5, 10, 10), (8, 64, 64, 64), (5, 33, 10, 4)):
        ids = np.empty((S, nq, k), dtype=np.uint32)
        dist = np.empty((S, nq, k), dtype=np.float32)
        for s in range(S):
            for qi in range(nq):
                n_valid = rs.randint(0, k + 1)
                row = np.sort(rs.randint(0, 40, size=k).astype(np.float32) * 0.5)        # many equal distances
                ids[s, qi] = s * 1000000 + rs.permutation(100000)[:k]
                dist[s, qi] = row
                ids[s, qi, n_valid:] = 0xFFFFFFFF
                dist[s, qi, n_valid:] = np.nan
        dist[0, 0, 0] = np.nan; dist[0, 1 % nq, 0] = -0.0 if nq > 1 else dist[0, 0, 0]
        w_ids, w_dist = merge_topk(list(ids), list(dist), k_out)
        g_ids, g_dist = _ffi.merge_topk_device(ids, dist, k_out)
        assert np.array_equal(g_ids, w_ids), (S, nq, k, k_out)
        assert np.array_equal(np.isnan(g_dist), np.isnan(w_dist))
        ok = ~np.isnan(w_dist)
        assert np.array_equal(g_dist[ok], w_dist[ok])


def test_sharded_search_with_the_rccl_exchange():
    """Four PQ-only shards on one device + a one-rank RCCL communicator: dr_sharded_search (device merge, ncclAllGather,
    device merge) == the host merge of the per-shard ORACLE runs, bit for bit; same answer without a communicator."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.parallel import merge_topk, shard_slice
    from diskrag_amd.sharded import globalize
    from diskrag_amd.synth import sift_like
    from oracle import pyoracle as orc
    n, nshard, k = 24000, 4, 10
    x, q = sift_like(n, 128, n_queries=80, n_clusters=64, seed=21, query_seed=22)
    shards, bases, parts, cb = [], [], [], None
    for s in range(nshard):
        sl = shard_slice(n, nshard, s)
        ix = HipIndex.create_empty(x[sl], R=32)
        medoid, _ = ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=5 + s, pad_with_zero=False)
        if cb is None:
            cb = ix.pq_train(16, n_sample=6000, iters=4)
        codes = ix.pq_encode(cb, want_codes=True)
        ix.drop_vectors()                       # a c5 shard holds no vectors
        shards.append(ix); bases.append(sl.start); parts.append((sl, medoid, ix.get_adjacency(), codes))
    comm = _ffi.Comm(_ffi.Comm.unique_id(), 1, 0, 0)
    try:
        for (L, bw) in ((100, 8), (30, 0)):
            o_ids, o_dist = [], []
            for (sl, medoid, adj, codes) in parts:
                oi, od, oc, ost = orc.search_batch(x[sl], adj, q, medoid, orc.PQ, k, L=L, bw=bw, codes=codes, codebook=cb)
                o_ids.append(globalize(oi, sl.start)); o_dist.append(od.astype(np.float32))
            w_ids, w_dist = merge_topk(o_ids, o_dist, k)
            for c in (comm, None):
                ids, dist, status, ms = _ffi.sharded_search(shards, bases, q, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ, comm=c)
                assert int(status.max()) == 0
                assert np.array_equal(ids, w_ids) and np.array_equal(bits(dist), bits(w_dist))
        gt = orc.bruteforce_topk(x, q, k)
        from diskrag_amd.synth import recall_at_k
        ids, _, _, _ = _ffi.sharded_search(shards, bases, q, k, L=100, beam_width=0, mode=_ffi.MODE_PQ, comm=comm)
        assert recall_at_k(ids, gt, k) > 0.5          # ADC-only distances, m = 16: far above the reference M3's ~0.01
    finally:
        comm.close()
        for ix in shards:
            ix.close()


@pytest.mark.parametrize("name", PQ_FIXTURES)
def test_pq_encode_reproduces_the_reference_codes(name):
    """N2: nearest-centroid codes for the reference's own codebook == the codes the reference produced (DiskANNPQ.encode,
    fast_pq.py:245-267, sklearn KMeans.predict). Where they differ the two centroids must be equidistant to float32
    rounding (sklearn's pairwise-distance expansion |x|^2 - 2xc + |c|^2 is not the direct form)."""
    from diskrag_amd import HipIndex
    g = load_golden(name)
    ix = HipIndex.create(g.vectors, g.adj, g.medoid)
    try:
        codes = ix.pq_encode(g.codebook, want_codes=True)
    finally:
        ix.close()
    diff = np.argwhere(codes != g.codes)
    assert len(diff) <= 0.001 * codes.size, f"{len(diff)} of {codes.size} code words differ"
    sd = g.vectors.shape[1] // g.m
    for (i, j) in diff:
        sub = g.vectors[i, j * sd:(j + 1) * sd].astype(np.float64)
        da = ((g.codebook[j, codes[i, j]].astype(np.float64) - sub) ** 2).sum()
        db = ((g.codebook[j, g.codes[i, j]].astype(np.float64) - sub) ** 2).sum()
        assert da <= db * (1 + 1e-6) + 1e-12, (i, j, da, db)     # ours is at least as near: the reference's pick was a rounding tie


@pytest.mark.parametrize("data", ["int32", "randn64", "unit96"])
def test_prune_kernel_equals_the_textbook_form_of_the_reference_prune(data):
    """N1: prune_kernel on explicit candidate lists == oracle/pybuild.robust_prune_fast(stale_reads=False): same picks in
    the same order (the restatement itself is pinned on graphs the reference built: tests/test_oracle_build.py)."""
    from diskrag_amd import HipIndex
    from oracle import pybuild
    from tests.conftest import GOLDEN
    rs = np.random.RandomState(9)
    if data == "int32":
        pts = np.load(GOLDEN / "build_int32.npz")["points"]
    elif data == "randn64":
        pts = rs.randn(700, 64).astype(np.float32)
    else:
        pts = rs.randn(600, 96).astype(np.float32); pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    ix = HipIndex.create_empty(pts, R=16)
    try:
        for trial in range(60):
            p = int(rs.randint(len(pts)))
            n = int(rs.choice([3, 17, 40, 64, 130, 300]))
            cands = rs.choice(len(pts), size=min(n, len(pts)), replace=False).astype(np.uint32)
            if trial % 7 == 0:
                cands = np.concatenate([cands, cands[:5], [p]]).astype(np.uint32)       # duplicates and the point itself
            alpha, R = float(rs.choice([1.0, 1.2, 2.0])), int(rs.choice([4, 8, 16, 64]))
            got = ix.debug_prune(p, cands, alpha, R)
            row, picked = pybuild.robust_prune_fast(pts, p, cands, alpha, R, return_order=True, stale_reads=False)
            assert got.tolist() == picked, (data, trial, p, alpha, R)
    finally:
        ix.close()

@pytest.mark.parametrize("D,m", [(128, 32), (128, 16), (96, 8), (256, 64)])
def test_pq_prune_kernel_equals_its_restatement(D, m):
    """prune_pq_kernel (rows in registers for m = 32 / 16, rows in LDS for m = 8 / 64) on explicit candidate lists ==
    oracle/pybuild.robust_prune_pq: the textbook robust prune over code-word distances built from the oracle's A2 table and
    A3 sum (both pinned on the reference's goldens) -- same picks in the same order. Clustered data: many code words repeat,
    so equal distances and the (distance, id) tie order are exercised."""
    from diskrag_amd import HipIndex
    from diskrag_amd.synth import unit_mixture
    from oracle import pybuild
    rs = np.random.RandomState(11 + m)
    x, _ = unit_mixture(3000, D, n_queries=4, n_clusters=24, seed=6, latent=12)
    full = HipIndex.create_empty(x, R=16)
    cb = full.pq_train(m, n_sample=3000, iters=4)
    codes = full.pq_encode(cb, want_codes=True)
    full.close()
    sh = HipIndex.create_codes_empty(len(x), D, 16, cb)
    try:
        sh.encode_rows(x, 0)
        for trial in range(40):
            p = int(rs.randint(len(x)))
            n = int(rs.choice([2, 17, 63, 64, 65, 130, 300]))
            cands = rs.choice(len(x), size=n, replace=False).astype(np.uint32)
            if trial % 5 == 0:
                cands = np.concatenate([cands[:-6], cands[:5], [p]]).astype(np.uint32)      # duplicates and the point itself
            alpha, R = float(rs.choice([1.0, 1.2, 2.0])), int(rs.choice([4, 16, 64, 128]))
            got = sh.debug_prune_pq(p, cands, alpha, R)
            want = pybuild.robust_prune_pq(cb, codes, p, cands, alpha, R)
            assert got.tolist() == want.tolist(), (D, m, trial, p, n, alpha, R)
    finally:
        sh.close()


def test_centroid_pair_table_follows_the_codebook():
    """the centroid-pair table of the PQ-only builder is cached on the handle; replacing the codebook (set_pq with
    another m, pq_encode) must rebuild it -- dr_debug_prune_pq right after each change against the restatement on the NEW
    codebook (a stale table scores with the old centroids; with a larger m it was read out of bounds)."""
    from diskrag_amd import HipIndex
    from diskrag_amd.synth import unit_mixture
    from oracle import pybuild
    rs = np.random.RandomState(3)
    x, _ = unit_mixture(2500, 128, n_queries=4, n_clusters=16, seed=8, latent=10)
    ix = HipIndex.create_empty(x, R=16)
    try:
        books = {}
        for m in (16, 32, 8, 32):
            cb = ix.pq_train(m, n_sample=2500, iters=3, seed=100 + len(books) + m)     # (the second m = 32 book differs from the first)
            codes = ix.pq_encode(cb, want_codes=True)          # replaces codebook and m on the handle
            books[(m, len(books))] = (cb, codes)
            for trial in range(6):
                p = int(rs.randint(len(x)))
                cands = rs.choice(len(x), size=int(rs.choice([17, 64, 130])), replace=False).astype(np.uint32)
                got = ix.debug_prune_pq(p, cands, 1.2, 16)
                want = pybuild.robust_prune_pq(cb, codes, p, cands, 1.2, 16)
                assert got.tolist() == want.tolist(), (m, trial)
        # set_pq (host-side codes + codebook) invalidates it too
        (cb, codes) = books[(16, 0)]
        ix.set_pq(cb, codes)
        p, cands = 7, rs.choice(len(x), size=90, replace=False).astype(np.uint32)
        assert ix.debug_prune_pq(p, cands, 1.2, 16).tolist() == pybuild.robust_prune_pq(cb, codes, p, cands, 1.2, 16).tolist()
    finally:
        ix.close()


@pytest.mark.parametrize("name,m", [("sift128", 32), ("deep96", 16)])
def test_pq_trainer_reaches_the_reference_quantisation_error(name, m):
    """N2: k-means++ / n_init / Lloyd on the device vs DiskANNPQ.fit (sklearn) on the same vectors: the summed inertia is
    within 2 % of the reference's (tests/golden/gen_golden_build.py recorded it), with the reference's own parameters
    for this data size (n_init 10, max_iter 300: fast_pq.py:188-195)."""
    from diskrag_amd import HipIndex
    from tests.conftest import GOLDEN
    x = np.load(GOLDEN / f"data_{name}.npz")["vectors"]
    ref_m, ref_inertia, _ = np.load(GOLDEN / "build_int32.npz")[f"pqfit_{name}"]
    assert int(ref_m) == m
    ix = HipIndex.create_empty(x, R=16)
    try:
        cb, inertia = ix.pq_train_ex(m, n_sample=len(x), max_iter=300, n_init=10, tol=1e-4, seed=7)
        codes = ix.pq_encode(cb, want_codes=True)
    finally:
        ix.close()
    sd = x.shape[1] // m
    rec = np.concatenate([cb[j][codes[:, j]] for j in range(m)], axis=1)
    err = float(((x.astype(np.float64) - rec) ** 2).sum())
    assert abs(err - inertia) <= 1e-3 * inertia          # the encoder assigns what the trainer assigned
    assert inertia <= 1.02 * ref_inertia, (inertia, ref_inertia)

def test_copy_codes_gives_a_second_shard_over_the_same_points():
    """dr_index_copy_codes: another degree R over points whose vectors are gone -- the copy scans like the source."""
    from diskrag_amd import HipIndex, _ffi
    g = load_golden("unit1536_R16_m32")
    a = HipIndex.create_codes(g.adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes)
    b = HipIndex.create_codes_empty(len(g.vectors), g.vectors.shape[1], 48, g.codebook[:, ::-1].copy())     # (another codebook: replaced by the copy)
    try:
        b.copy_codes_from(a)
        fa, _ = a.pq_scan(g.queries)
        fb, _ = b.pq_scan(g.queries)
        assert np.array_equal(bits(fa), bits(fb))
        medoid, _ = b.build_vamana_pq(L_build=32, alpha=1.2, passes=2, seed=3)
        ids, dist, cnt, st = b.search_batch(g.queries, 10, L=64, beam_width=0, mode=_ffi.MODE_PQ)
        # the copy's searches are the oracle's on the graph it built, with the SOURCE's code words and codebook
        from oracle import pyoracle as orc
        w = orc.search_batch(g.vectors, b.get_adjacency(), g.queries, medoid, orc.PQ, 10, L=64, bw=0, codes=g.codes, codebook=g.codebook)
        assert int(st["status"].max()) == 0 and np.array_equal(ids, w[0]) and np.array_equal(_stats4(st), w[3])
        with pytest.raises(_ffi.DiskragHipError):
            b.copy_codes_from(b)
    finally:
        a.close(); b.close()


def test_pq_only_builder_makes_a_searchable_shard():
    """c5's construction at test scale: code words streamed in chunks (vectors never resident), Vamana graph built from the
    code words alone (dr_build_vamana_pq), searched with DR_MODE_PQ. No reference counterpart: held to graph quality --
    the merged lists must find most of the brute-force ADC top-10 -- and to the oracle on the graph it built."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import unit_mixture, recall_at_k
    from oracle import pyoracle as orc
    n, D, m, R = 40000, 256, 32, 32
    x, q = unit_mixture(n, D, n_queries=200, n_clusters=64, seed=4, latent=24)
    full = HipIndex.create_empty(x[:8192], R=R)
    cb = full.pq_train(m, n_sample=8192, iters=6)                # the codebook comes from a sample, as it would at 1e9
    full.close()
    sh = HipIndex.create_codes_empty(n, D, R, cb)
    try:
        for r0 in range(0, n, 7000):
            sh.encode_rows(x[r0:r0 + 7000], r0)
        with pytest.raises(_ffi.DiskragHipError):          # the prune's candidate list: L_build + R + 64 <= 320
            sh.build_vamana_pq(L_build=256, alpha=1.2, passes=2, seed=3)
        medoid, secs = sh.build_vamana_pq(L_build=64, alpha=1.2, passes=2, seed=3)
        adj = sh.get_adjacency()
        deg = (adj != 0xFFFFFFFF).sum(axis=1)
        assert deg.min() >= 1 and deg.max() <= R and deg.mean() > R / 3
        valid = adj[adj != 0xFFFFFFFF]
        assert valid.max() < n
        # codes as the one-shot encoder makes them
        ref = HipIndex.create_empty(x, R=R)
        codes = ref.pq_encode(cb, want_codes=True)
        ref.close()
        # ground truth in the shard's own metric: brute-force ADC top-10
        scan_ids = []
        for i in range(0, 200, 8):
            _, _, _, d_all = sh.pq_scan_best(q[i:i + 8], want_output=True)
            scan_ids.append(np.argsort(d_all, axis=1, kind="stable")[:, :10])
        gt_adc = np.concatenate(scan_ids).astype(np.uint32)
        ids, dist, cnt, st = sh.search_batch(q, 10, L=100, beam_width=0, mode=_ffi.MODE_PQ)
        assert int(st["status"].max()) == 0
        assert recall_at_k(ids, gt_adc, 10) > 0.9
        # and the traversal itself is the oracle's on that graph
        w = orc.search_batch(x, adj, q, medoid, orc.PQ, 10, L=100, bw=0, codes=codes, codebook=cb, nthreads=8)
        assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist), bits(w[1].astype(np.float32)))
    finally:
        sh.close()

def test_pq_only_builder_with_the_widest_row():
    """R = 128 (two 64-lane passes per row, L_build + R + 64 = 320 candidates in the prune: both limits of dr_build_vamana_pq), the
    degree the full-size c5 shard is built with: valid rows, a searchable graph, and the device traversal equal to the oracle's on
    the graph it built."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import unit_mixture, recall_at_k
    from oracle import pyoracle as orc
    n, D, m, R = 30000, 128, 32, 128
    x, q = unit_mixture(n, D, n_queries=100, n_clusters=64, seed=4, latent=24)
    full = HipIndex.create_empty(x, R=16)
    cb = full.pq_train(m, n_sample=8192, iters=6)
    codes = full.pq_encode(cb, want_codes=True)
    full.close()
    sh = HipIndex.create_codes_empty(n, D, R, cb)
    try:
        sh.encode_rows(x, 0)
        medoid, _ = sh.build_vamana_pq(L_build=128, alpha=1.2, passes=2, seed=3)
        adj = sh.get_adjacency()
        deg = (adj != 0xFFFFFFFF).sum(axis=1)
        assert deg.min() >= 1 and deg.max() <= R and adj[adj != 0xFFFFFFFF].max() < n
        assert all(len(set(r[r != 0xFFFFFFFF].tolist())) == int((r != 0xFFFFFFFF).sum()) for r in adj[:2000])     # no repeated neighbour
        gt_adc = sh.pq_scan_topk(q, 10)[0]
        ids, dist, cnt, st = sh.search_batch(q, 10, L=100, beam_width=8, mode=_ffi.MODE_PQ)
        assert int(st["status"].max()) == 0 and recall_at_k(ids, gt_adc, 10) > 0.9
        w = orc.search_batch(x, adj, q, medoid, orc.PQ, 10, L=100, bw=8, codes=codes, codebook=cb, nthreads=8)
        assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist), bits(w[1].astype(np.float32)))
    finally:
        sh.close()


_PRUNE_FORMS_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex
from diskrag_amd.synth import unit_mixture
m = int(sys.argv[1])
x, q = unit_mixture(30000, 128, n_queries=8, n_clusters=64, seed=4, latent=24)
full = HipIndex.create_empty(x[:8192], R=32)
cb = full.pq_train(m, n_sample=8192, iters=6)
full.close()
sh = HipIndex.create_codes_empty(len(x), 128, 32, cb)
sh.encode_rows(x, 0)
sh.build_vamana_pq(L_build=64, alpha=1.2, passes=2, seed=3)
print("GRAPH", hashlib.sha1(sh.get_adjacency().tobytes()).hexdigest())
"""


@pytest.mark.parametrize("m", [16, 32])
def test_pq_prune_forms_build_the_same_graph(m, tmp_path):
    """prune_pq_kernel keeps its centroid-pair rows in registers (m = 16, 32: ds_bpermute lookups, 8 wavefronts per CU) or in
    LDS (any m; DR_PQ_PRUNE_LDS=1 forces it): the same sums in the same order, so the built graph is the same bit for bit.
    The switch is read once per process: one child process per form."""
    import subprocess
    import sys
    script = tmp_path / "build_once.py"
    script.write_text(_PRUNE_FORMS_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for force_lds in (False, True):
        env = dict(os.environ)
        env.pop("DR_PQ_PRUNE_LDS", None)
        if force_lds:
            env["DR_PQ_PRUNE_LDS"] = "1"
        r = subprocess.run([sys.executable, str(script), str(m)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes.append([ln for ln in r.stdout.splitlines() if ln.startswith("GRAPH")][-1])
    assert hashes[0] == hashes[1]


_EXACT_PRUNE_FORMS_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex
from diskrag_amd.synth import sift_like, unit_mixture
D = int(sys.argv[1])
x = sift_like(20000, D, n_queries=4, seed=3)[0] if D == 128 else unit_mixture(20000, D, n_queries=4, n_clusters=64, seed=4, latent=24)[0]
ix = HipIndex.create_empty(x, R=32)
ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=5)
print("GRAPH", hashlib.sha1(ix.get_adjacency().tobytes()).hexdigest())
"""


def test_pq_builder_searches_without_a_visited_set_build_the_same_graph(monkeypatch):
    """Round 4: on large shards the PQ-only builder's greedy searches keep no visited set (DR_BUILD_PQ_NO_VISITED_SET; automatic from
    2^25 points): the lists they return are the same, so the graph must be the same bit for bit -- forced on and off here."""
    import hashlib
    from diskrag_amd import HipIndex
    from diskrag_amd.synth import unit_mixture
    x, _ = unit_mixture(20000, 128, n_queries=4, n_clusters=32, seed=13, latent=12)
    full = HipIndex.create_empty(x, R=16)
    cb = full.pq_train(32, n_sample=8000, iters=3)
    full.close()
    shas = {}
    for env in ("0", "1"):
        monkeypatch.setenv("DR_BUILD_PQ_NO_VISITED_SET", env)
        sh = HipIndex.create_codes_empty(len(x), 128, 64, cb)
        try:
            sh.encode_rows(x, 0)
            medoid, _ = sh.build_vamana_pq(L_build=80, alpha=1.2, passes=2, seed=5)
            shas[env] = (medoid, hashlib.sha1(sh.get_adjacency().tobytes()).hexdigest())
        finally:
            sh.close()
    assert shas["0"] == shas["1"]


@pytest.mark.parametrize("D", [96, 128, 256])
def test_exact_prune_forms_build_the_same_graph(D, tmp_path):
    """prune_kernel<D, true> (the next <= 4 likely picks scored in one pass over the candidates' rows; the default at D <= 256)
    and the plain form (one pass per pick; DR_PRUNE_PLAIN=1) make the same picks in the same order: the same graph bit for
    bit. The switch is read once per process: one child process per form."""
    import subprocess
    import sys
    script = tmp_path / "build_once.py"
    script.write_text(_EXACT_PRUNE_FORMS_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for plain in (False, True):
        env = dict(os.environ)
        env.pop("DR_PRUNE_PLAIN", None)
        if plain:
            env["DR_PRUNE_PLAIN"] = "1"
        r = subprocess.run([sys.executable, str(script), str(D)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        hashes.append([ln for ln in r.stdout.splitlines() if ln.startswith("GRAPH")][-1])
    assert hashes[0] == hashes[1]


@pytest.mark.parametrize("d", [7, 64, 96, 128, 130, 960, 1536])
def test_scalar_kernels_on_the_device(d):
    """C8: l2_distance_fast_cython / cosine_similarity_cython on the device against the reference's own outputs
    (tests/golden/k_scalar.npz), at the reference's tolerance (test_pydiskann_cython.sh:50-54: rtol 1e-5, atol 1e-6)."""
    from diskrag_amd import _ffi
    from tests.conftest import GOLDEN
    z = np.load(GOLDEN / "k_scalar.npz")
    a, b = z[f"a{d}"], z[f"b{d}"]
    l2, cs = _ffi.scalar_kernels(a, b)
    np.testing.assert_allclose(l2, z[f"l2_{d}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cs, z[f"cos_{d}"], rtol=1e-5, atol=1e-6)
    zero = np.zeros_like(a[:2])
    _, c0 = _ffi.scalar_kernels(zero, b[:2])
    assert (c0 == 0.0).all()                      # either norm zero -> 0.0 (cython_utils.pyx:66-67)
