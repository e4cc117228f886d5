"""The skewed flat PQ scan (round 6, pq_scan_skew_kernel: every lane of a lane group on its own table row, conflict-free LDS lookups, code words
read from a scan-order copy) against the reference's ADC (A3, fast_pq.py:325-326, restated in the oracle) and against the kernel it replaces
(DR_PQ_SCAN_NO_SKEW=1: pq_scan_kernel), bit for bit: every m, ragged sizes around the 64-point rows, several block rows, non-finite tables,
and a code change between two scans."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _index(n, D, m, seed):
    from diskrag_amd import HipIndex
    rs = np.random.RandomState(seed)
    codes = rs.randint(0, 256, size=(n, m)).astype(np.uint8)
    cb = rs.randn(m, 256, D // m).astype(np.float32)
    return HipIndex.create_codes(np.zeros((n, 1), dtype=np.uint32), 0, D, cb, codes), cb, codes, rs


@pytest.mark.parametrize("D,m", [(64, 16), (128, 32), (96, 48), (128, 64)])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 4097, 70001])
def test_the_skewed_scan_is_the_reference_adc(D, m, n, monkeypatch):
    from oracle import pyoracle as orc
    ix, cb, codes, rs = _index(n, D, m, 3 * m + n % 97)
    try:
        if n > 200:
            codes[100] = codes[n - 7]      # equal sums: the smaller id wins
            ix.set_pq(cb, codes)
        for nq in (1, 3):
            q = rs.randn(nq, D).astype(np.float32)
            if nq > 1:
                monkeypatch.setenv("DR_PQ_SCAN_PER_QUERY", "1")      # one block row per query: the skewed kernel with gridDim.y = nq
            bid, bsq, ms, allsq = ix.pq_scan_best(q, want_output=True)
            bid2, bsq2, _ = ix.pq_scan_best(q)
            monkeypatch.setenv("DR_PQ_SCAN_NO_SKEW", "1")
            bid0, bsq0, _, all0 = ix.pq_scan_best(q, want_output=True)
            monkeypatch.delenv("DR_PQ_SCAN_NO_SKEW")
            monkeypatch.delenv("DR_PQ_SCAN_PER_QUERY", raising=False)
            assert np.array_equal(bits(allsq), bits(all0)) and np.array_equal(bid, bid0) and np.array_equal(bits(bsq), bits(bsq0))
            assert np.array_equal(bid, bid2) and np.array_equal(bits(bsq), bits(bsq2))
            for qi in range(nq):
                want = orc.adc(orc.build_lut(cb, q[qi]), codes)[0]
                assert np.array_equal(bits(allsq[qi]), bits(want))
                assert int(bid[qi]) == int(np.flatnonzero(want == want.min())[0])
                assert bits(np.float32(bsq[qi])) == bits(want.min())
    finally:
        ix.close()


@pytest.mark.parametrize("D,m", [(64, 16), (128, 32), (128, 64)])
def test_non_finite_table_entries_stay_in_their_own_code_word(D, m, monkeypatch):
    """a centroid with an infinite / NaN coordinate: only the code words that use it get inf / NaN; the kernel's body with selects == pq_scan_kernel"""
    ix, cb, codes, rs = _index(9000, D, m, 17 + m)
    try:
        cb[3, 11, 0] = np.inf
        cb[m - 1, 200, 1] = np.nan
        cb[0, 0, 0] = -np.inf
        ix.set_pq(cb, codes)
        q = rs.randn(1, D).astype(np.float32)
        _, _, _, a1 = ix.pq_scan_best(q, want_output=True)
        monkeypatch.setenv("DR_PQ_SCAN_NO_SKEW", "1")
        _, _, _, a0 = ix.pq_scan_best(q, want_output=True)
        monkeypatch.delenv("DR_PQ_SCAN_NO_SKEW")
        assert np.array_equal(bits(a1), bits(a0))
        hit = (codes[:, 3] == 11) | (codes[:, m - 1] == 200) | (codes[:, 0] == 0)
        assert np.isfinite(a1[0][~hit]).all() and not np.isfinite(a1[0][hit]).any() and hit.sum() > 30
    finally:
        ix.close()


def test_the_scan_order_copy_follows_the_code_words():
    """dr_index_set_pq / encode between two scans: the copy is rebuilt, never stale"""
    from oracle import pyoracle as orc
    ix, cb, codes, rs = _index(5000, 128, 32, 5)
    try:
        q = rs.randn(1, 128).astype(np.float32)
        _, _, _, a = ix.pq_scan_best(q, want_output=True)
        codes2 = rs.randint(0, 256, size=codes.shape).astype(np.uint8)
        ix.set_pq(cb, codes2)
        _, _, _, b = ix.pq_scan_best(q, want_output=True)
        lut = orc.build_lut(cb, q[0])
        assert np.array_equal(bits(a[0]), bits(orc.adc(lut, codes)[0])) and np.array_equal(bits(b[0]), bits(orc.adc(lut, codes2)[0]))
        cb3 = rs.randn(16, 256, 8).astype(np.float32)      # another m on the same index
        codes3 = rs.randint(0, 256, size=(5000, 16)).astype(np.uint8)
        ix.set_pq(cb3, codes3)
        _, _, _, c = ix.pq_scan_best(q, want_output=True)
        assert np.array_equal(bits(c[0]), bits(orc.adc(orc.build_lut(cb3, q[0]), codes3)[0]))
    finally:
        ix.close()


@pytest.mark.parametrize("contiguous", [False, True])
def test_several_streams_per_wavefront_and_both_layouts_of_the_copy(contiguous, monkeypatch):
    """three block rows share the device (fewer wavefronts than streams of the scan-order copy: a wavefront walks several streams); the interleaved
    layout and DR_PQ_SCAN_CONTIGUOUS_STREAMS=1 (the A/B layout: every stream in one piece) give the same bits"""
    from oracle import pyoracle as orc
    if contiguous:
        monkeypatch.setenv("DR_PQ_SCAN_CONTIGUOUS_STREAMS", "1")
    ix, cb, codes, rs = _index(300007, 128, 32, 23)
    try:
        q = rs.randn(3, 128).astype(np.float32)
        _, _, _, a = ix.pq_scan_best(q[:1], want_output=True)
        monkeypatch.setenv("DR_PQ_SCAN_PER_QUERY", "1")
        bid, bsq, _, b = ix.pq_scan_best(q, want_output=True)
        for qi in range(3):
            want = orc.adc(orc.build_lut(cb, q[qi]), codes)[0]
            assert np.array_equal(bits(b[qi]), bits(want))
            assert int(bid[qi]) == int(np.flatnonzero(want == want.min())[0])
        assert np.array_equal(bits(a[0]), bits(b[0]))
    finally:
        ix.close()


@pytest.mark.parametrize("D,m", [(64, 16), (128, 32), (96, 48), (128, 64)])
def test_brute_force_adc_search_of_one_query_on_the_skewed_kernel(D, m, monkeypatch):
    """dr_pq_scan_topk with one query (and one block row per query): the k smallest (distance, id) pairs of the flat scan, the smaller id first among
    equal sums; == pq_scan_topk_kernel (DR_PQ_SCAN_NO_SKEW=1)"""
    for n in (50, 4097, 250003):
        ix, cb, codes, rs = _index(n, D, m, 7 * m + n % 89)
        try:
            if n > 200:
                codes[17] = codes[n - 3]; codes[n // 2] = codes[n - 3]      # equal sums
                ix.set_pq(cb, codes)
            q = rs.randn(3, D).astype(np.float32)
            full = ix.pq_scan_best(q, want_output=True)[3]
            for k in (1, 10, 64):
                got1 = ix.pq_scan_topk(q[:1], k)
                monkeypatch.setenv("DR_PQ_SCAN_PER_QUERY", "1")
                got3 = ix.pq_scan_topk(q, k)
                monkeypatch.setenv("DR_PQ_SCAN_NO_SKEW", "1")
                old3 = ix.pq_scan_topk(q, k)
                monkeypatch.delenv("DR_PQ_SCAN_NO_SKEW")
                monkeypatch.delenv("DR_PQ_SCAN_PER_QUERY")
                assert np.array_equal(got3[0], old3[0]) and np.array_equal(bits(got3[1]), bits(old3[1]))
                assert np.array_equal(got1[0][0], got3[0][0]) and np.array_equal(bits(got1[1][0]), bits(got3[1][0]))
                for qi in range(3):
                    order = np.lexsort((np.arange(n), full[qi]))[:k]
                    kk = min(k, n)
                    assert np.array_equal(got3[0][qi][:kk], order.astype(np.uint32)[:kk]), (n, k, qi)
                    assert np.array_equal(bits(got3[1][qi][:kk]), bits(full[qi][order][:kk]))
        finally:
            ix.close()
