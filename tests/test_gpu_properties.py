"""Size-independent properties of the search output on a device-built index that is too large for the oracle to
sweep quickly (N = 200k): what the reference's result list must look like whatever the data. Needs a GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PAD = np.uint32(0xFFFFFFFF)


@pytest.fixture(scope="module")
def big():
    from diskrag_amd import HipIndex
    from diskrag_amd.synth import sift_like
    x, q = sift_like(200000, 128, n_queries=3000, n_clusters=256, seed=77)
    ix = HipIndex.create_empty(x, R=64)
    ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=5, pad_with_zero=True)
    cb = ix.pq_train(32, n_sample=50000, iters=5)
    codes = ix.pq_encode(cb, want_codes=True)
    return ix, x, q, cb, codes


@pytest.mark.parametrize("mode,L,bw,k", [(1, 100, 8, 10), (1, 100, 0, 20), (1, 40, 8, 30), (2, 0, 32, 10), (4, 64, 0, 10)])
def test_result_list_properties(big, mode, L, bw, k):
    ix, x, q, cb, codes = big
    ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=mode)
    assert (st["status"] == 0).all()
    cap = bw if mode == 2 else L
    for i in range(0, len(q), 7):
        n = int(cnt[i])
        assert 1 <= n <= min(k, cap)
        row, d = ids[i, :n], dist[i, :n]
        assert (ids[i, n:] == PAD).all() and np.isnan(dist[i, n:]).all()
        assert len(set(row.tolist())) == n                      # a node is returned once
        assert (np.diff(d) >= 0).all()                           # ascending distances
    # returned distances are the exact distances of the returned ids (A1 bits; sqrt for the norm modes)
    sel = np.arange(0, len(q), 101)
    for i in sel:
        n = int(cnt[i])
        ex = ix.exact_distances(q[i:i + 1], ids[i, :n])[0]
        want = np.sqrt(ex) if mode in (2, 4) else ex
        assert np.array_equal(dist[i, :n].view(np.uint32), want.astype(np.float32).view(np.uint32))
    # idempotence: the same batch again gives the same bits (visited bitmaps and double buffers are reusable)
    ids2, dist2, cnt2, st2 = ix.search_batch(q, k, L=L, beam_width=bw, mode=mode)
    assert np.array_equal(ids, ids2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32))
    assert np.array_equal(st["steps"], st2["steps"]) and np.array_equal(st["visited"], st2["visited"])


def test_counters_are_consistent(big):
    ix, x, q, cb, codes = big
    ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=0, mode=1)
    assert (st["visited"] == st["pq"] + 1).all()                 # every visited node but the start gets one ADC
    assert (st["exact"] <= st["visited"]).all() and (st["pq_evaluated"] <= st["pq"]).all()
    assert (st["steps"] <= 1000).all() and (st["inserts"] <= st["exact"]).all()
    # byte-row variant, R = 64, bit order on: the adjacency row of the predicted next node is prefetched into LDS
    assert ix.timing()["variant"] in (13, 17)      # (17: the same kernel in 4-wavefront workgroups, batches below 4096 queries)
    assert (st["adj_prefetch_hits"] < st["steps"]).all() and st["adj_prefetch_hits"].sum() > 0.5 * st["steps"].sum()
    ix.debug_force_kind(9)
    try:
        ids9, dist9, cnt9, st9 = ix.search_batch(q, 10, L=100, beam_width=0, mode=1)
        assert ix.timing()["variant"] == 9 and (st9["adj_prefetch_hits"] == 0).all()
        assert np.array_equal(ids, ids9) and np.array_equal(dist.view(np.uint32), dist9.view(np.uint32))
        assert np.array_equal(st["steps"], st9["steps"]) and np.array_equal(st["exact"], st9["exact"])
    finally:
        ix.debug_force_kind(-1)


def test_sample_matches_oracle_at_scale(big):
    from oracle import pyoracle as orc
    ix, x, q, cb, codes = big
    adj = ix.get_adjacency()
    for L, bw in ((100, 8), (100, 0)):
        ids, dist, cnt, st = ix.search_batch(q[:300], 10, L=L, beam_width=bw, mode=1)
        oids, odist, ocnt, ost = orc.search_batch(x, adj, q[:300], ix.medoid, orc.M1, 10, L=L, bw=bw, codes=codes,
                                                  codebook=cb, nthreads=8)
        assert np.array_equal(ids, oids)
        assert np.array_equal(dist.view(np.uint32), odist.astype(np.float32).view(np.uint32))
        assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)


def test_live_policy_sample_matches_oracle_at_scale():
    """The A4-live regime (unit-norm data, ADC evaluated, rows fetched only for neighbours that can pass) at a size
    where the locality bit order of the visited bitmap is active (N >= 32768): oracle sample, both band policies,
    and the second batch (served by the variant the measured regime selects) equals the first."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import unit_mixture
    from oracle import pyoracle as orc
    x, q = unit_mixture(150000, 96, n_queries=400, n_clusters=512, seed=13, latent=32)
    ix = HipIndex.create_empty(x, R=48)
    medoid, _ = ix.build_vamana(L_build=80, alpha=1.2, passes=2, seed=4)
    cb = ix.pq_train(16, n_sample=40000, iters=4)
    codes = ix.pq_encode(cb, want_codes=True)
    adj = ix.get_adjacency()
    try:
        for L, bw, pol in ((100, 8, 0), (100, 0, 1), (50, 8, 1)):
            first = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
            again = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
            oi, od, oc, ost = orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=L, bw=bw, policy=pol, codes=codes,
                                               codebook=cb, nthreads=8)
            for ids, dist, cnt, st in (first, again):
                assert (st["status"] == 0).all()
                assert np.array_equal(ids, oi) and np.array_equal(cnt, oc)
                valid = oi != PAD
                assert np.array_equal(dist[valid].view(np.uint32), od[valid].astype(np.float32).view(np.uint32))
                assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)
        assert ix.debug_force_kind(-1) == 1          # measured: live
        st = first[3]
        assert st["exact"].sum() < 0.8 * st["visited"].sum()
    finally:
        ix.close()


def test_other_modes_sample_matches_oracle_at_scale(big):
    """M2, M3 (exact and PQ-only) and M4 on the 200k index: oracle sample, bit for bit (the same 0-padded rows are
    given to both, so the phantom neighbour 0 of short rows is a real neighbour for both)."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    ix, x, q, cb, codes = big
    adj = ix.get_adjacency()
    qs = q[:200]
    P = orc.F_PAIRWISE
    for (mode, omode, k, L, bw, fl, ofl) in ((2, orc.M2, 8, 0, 8, 0, P), (2, orc.M2, 10, 0, 64, 0, P),
                                             (4, orc.M4, 10, 64, 0, _ffi.F_SQDIST, orc.F_CYTHON | P), (4, orc.M4, 10, 64, 0, 0, P),
                                             (3, orc.M3, 10, 10, 32, _ffi.F_USE_PQ, orc.F_USE_PQ | P),
                                             (3, orc.M3, 5, 5, 8, 0, P)):
        ids, dist, cnt, st = ix.search_batch(qs, k, L=L, beam_width=bw, mode=mode, flags=fl)
        oi, od, oc, ost = orc.search_batch(x, adj, qs, ix.medoid, omode, k, L=L, bw=bw, flags=ofl, codes=codes, codebook=cb,
                                           nthreads=8)
        assert (st["status"] == 0).all()
        assert np.array_equal(ids, oi) and np.array_equal(cnt, oc), (mode, k, L, bw)
        valid = oi != PAD
        assert np.array_equal(dist[valid].view(np.uint32), od[valid].astype(np.float32).view(np.uint32)), (mode, k, L, bw)
        assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)


@pytest.mark.parametrize("name", ["sift128_R64_m32", "unit1536_R16_m32", "deep96_R32_m16", "faq32_R16_nopq"])
def test_small_calls_direct_path_equals_the_general_path(name, monkeypatch):
    """dr_search_batch with <= 256 queries writes its results straight into the page-locked slab and runs the tie-order pass only when a query was
    listed for it (round 5); DR_NO_DIRECT=1 sends the same call through the general path (download copies, tie-order pass on its own stream):
    the same ids, distance bits, counts and counters in every mode, ties included (the integer-valued fixture)."""
    from diskrag_amd import _ffi
    from tests.conftest import load_golden
    from tests.test_gpu_parity import bits, get_index
    g = load_golden(name)
    ix = get_index(name)
    runs = [dict(L=100, beam_width=8, mode=_ffi.MODE_M2), dict(L=50, beam_width=0, mode=_ffi.MODE_M4)]
    if g.m:
        runs += [dict(L=100, beam_width=8, mode=_ffi.MODE_M1), dict(L=20, beam_width=8, mode=_ffi.MODE_M1, band_policy=1), dict(L=100, beam_width=16, mode=_ffi.MODE_PQB),
                 dict(L=60, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK), dict(L=10, beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)]
    for kw in runs:
        for q in (g.queries, g.queries[:1], g.queries[:7]):
            monkeypatch.delenv("DR_NO_DIRECT", raising=False)
            a = ix.search_batch(q, 10, **kw)
            monkeypatch.setenv("DR_NO_DIRECT", "1")
            b = ix.search_batch(q, 10, **kw)
            monkeypatch.delenv("DR_NO_DIRECT", raising=False)
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2]), (name, kw, len(q))
            for f in ("steps", "visited", "exact", "pq", "status", "inserts"):
                assert np.array_equal(a[3][f], b[3][f]), (name, kw, f)
