"""Round-2 entry points through the C ABI, on an MI355X: pipelined batches, several resident batches, the engine's
PQ-only traversal (DR_MODE_PQ, checked against the oracle's restatement of it) with and without the exact rerank,
result lists up to 1024 entries, the device merge kernel, the sharded search with its RCCL exchange, and the PQ
encoder against the reference-produced codes of every golden fixture."""
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
            for eng in (ix, shard):
                ids, dist, cnt, st = eng.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQ)
                assert int(st["status"].max()) == 0
                assert np.array_equal(ids, w[0]), (name, L, bw)
                valid = w[0] != 0xFFFFFFFF
                assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid])
                assert np.array_equal(cnt, w[2]) and np.array_equal(_stats4(st), w[3])
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


@pytest.mark.parametrize("name", ["unit1536_R16_m32", "unit1536_R16_m64"])
def test_multi_wave_queries_match_the_single_wave_kernels_and_the_oracle(name):
    """D = 1536: variants 15 / 16 (four wavefronts share one query and one table) against variants 0 / 2 and the oracle,
    both band policies, several capacities, a batch larger than the grid."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    q = np.concatenate([g.queries] * 40)[:2500]           # more queries than workgroup slots (4 x 256 CUs)
    try:
        for (L, bw, pol, k) in ((100, 8, 0, 10), (100, 0, 1, 10), (20, 8, 0, 5), (300, 16, 1, 10), (600, 0, 0, 10)):
            w = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M1, k, L=L, bw=bw, policy=pol, codes=g.codes,
                                 codebook=g.codebook, nthreads=8)
            for kind in (15, 0):
                if kind == 15 and L > 512:
                    continue
                ix.debug_force_kind(kind)
                ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
                assert ix.timing()["variant"] == kind
                assert int(st["status"].max()) == 0
                n = len(g.queries)
                for rep in range(0, len(q) - n + 1, n * 13):
                    sl = slice(rep, rep + n)
                    assert np.array_equal(ids[sl], w[0]), (kind, L, bw, pol)
                    assert np.array_equal(bits(dist[sl]), bits(w[1].astype(np.float32)))
                    assert np.array_equal(_stats4(st[sl]), w[3])
        for (mode, omode, fl, ofl, L, bw, k) in ((_ffi.MODE_PQ, orc.PQ, 0, 0, 100, 8, 10), (_ffi.MODE_M3, orc.M3, _ffi.F_USE_PQ, orc.F_USE_PQ, 10, 8, 10),
                                                 (_ffi.MODE_PQ, orc.PQ, 0, 0, 64, 0, 20)):
            gi = get_index(name, mem=(mode == _ffi.MODE_M3))
            adj = g.mem_adj if mode == _ffi.MODE_M3 else g.adj
            w = orc.search_batch(g.vectors, adj, g.queries, g.medoid, omode, k, L=L, bw=bw, flags=ofl, codes=g.codes, codebook=g.codebook)
            for kind in (16, 2):
                gi.debug_force_kind(kind)
                ids, dist, cnt, st = gi.search_batch(g.queries, k, L=L, beam_width=bw, mode=mode, flags=fl)
                assert gi.timing()["variant"] == kind
                assert np.array_equal(ids, w[0]) and np.array_equal(cnt, w[2]) and np.array_equal(_stats4(st), w[3])
                valid = w[0] != 0xFFFFFFFF
                assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid])
    finally:
        ix.debug_force_kind(-1)
