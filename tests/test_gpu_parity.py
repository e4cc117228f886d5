"""Parity of the HIP engine (through the C ABI) with the oracle and the golden vectors. Needs an MI355X."""
import numpy as np
import pytest

from tests.conftest import INDEX_FIXTURES, all_cases, load_golden

pytestmark = pytest.mark.gpu

PQ_FIXTURES = [n for n in INDEX_FIXTURES if "nopq" not in n]
_index_cache = {}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def get_index(name, mem=False):
    """HBM-resident index for a golden fixture (disk slot order, or the in-memory set order for M3/M4)."""
    from diskrag_amd import HipIndex
    key = (name, mem)
    if key not in _index_cache:
        g = load_golden(name)
        ix = HipIndex.create(g.vectors, g.mem_adj if mem else g.adj, g.medoid)
        if g.m:
            ix.set_pq(g.codebook, g.codes)
        _index_cache[key] = ix
    return _index_cache[key]


def test_library_loads_on_gpu():
    import diskrag_amd
    assert diskrag_amd.device_count() >= 1


# ------------------------------------------------------------------------------------ kernel-level parity

@pytest.mark.parametrize("name", list(INDEX_FIXTURES))
def test_exact_distance_bits(name):
    """A1 on the device == np.sum(diff*diff) of the reference, bit for bit (golden K1 + oracle on more nodes)."""
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    nodes = g.z["k1_nodes"]
    got = ix.exact_distances(g.queries[:4], nodes)
    assert np.array_equal(bits(got), bits(g.z["k1_exact"]))
    rs = np.random.RandomState(5)
    more = rs.randint(0, len(g.vectors), size=301).astype(np.uint32)
    got = ix.exact_distances(g.queries, more)
    want = np.array([[orc.sqdist(g.vectors[n], q) for n in more] for q in g.queries], dtype=np.float32)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("name", PQ_FIXTURES)
def test_distance_table_and_adc_bits(name):
    """A2/A3 on the device == compute_distance_table / asymmetric_distance of the reference, bit for bit."""
    g = load_golden(name)
    ix = get_index(name)
    nodes = g.z["k1_nodes"]
    lut = ix.distance_table(g.queries[:4])
    assert np.array_equal(bits(lut), bits(g.z["k1_lut"]))
    sq, rt = ix.adc(g.queries[:4], nodes)
    assert np.array_equal(bits(sq), bits(g.z["k1_adc_sq"]))
    assert np.array_equal(bits(rt), bits(g.z["k1_adc"]))


@pytest.mark.parametrize("name", ["randn128_R16_m32", "unit1536_R16_m32", "deep96_R32_m16"])
def test_get_node_roundtrip(name):
    """T1: what went into HBM (chain-major) reads back as the reference's (vector, neighbours) record."""
    g = load_golden(name)
    ix = get_index(name)
    for nid in (0, 1, g.medoid, len(g.vectors) - 1):
        vec, nbrs = ix.get_node(nid)
        assert np.array_equal(bits(vec), bits(g.vectors[nid]))
        assert np.array_equal(nbrs, g.adj[nid])


@pytest.mark.parametrize("name", ["sift128_R64_m32", "deep96_R32_m16"])
def test_pq_scan_matches_oracle(name):
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    out, ms = ix.pq_scan(g.queries[:3])
    for qi in range(3):
        lut = orc.build_lut(g.codebook, g.queries[qi])
        sq, _ = orc.adc(lut, g.codes)
        assert np.array_equal(bits(out[qi]), bits(sq))


@pytest.mark.parametrize("name", ["randn128_R16_m32", "sift128_R64_m32", "deep96_R32_m16"])
def test_bruteforce_topk(name):
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    ids, dist = ix.bruteforce_topk(g.queries, 10)
    want = orc.bruteforce_topk(g.vectors, g.queries, 10)
    # same distance multiset; ids may differ only inside exact ties
    for qi in range(len(g.queries)):
        wd = np.array([orc.sqdist(g.vectors[n], g.queries[qi]) for n in want[qi]], dtype=np.float32)
        assert np.array_equal(bits(np.sort(dist[qi])), bits(np.sort(wd)))


# ------------------------------------------------------------------------------------ search parity (golden)

def run_case(name, c):
    from diskrag_amd import _ffi
    mode = {"M1": _ffi.MODE_M1, "M2": _ffi.MODE_M2, "M3": _ffi.MODE_M3, "M4": _ffi.MODE_M4}[c["mode"]]
    flags = 0
    if c["mode"] == "M3" and c["use_pq"]:
        flags |= _ffi.F_USE_PQ
    if c["mode"] == "M4" and c.get("cython"):
        flags |= _ffi.F_SQDIST
    ix = get_index(name, mem=c["mode"] in ("M3", "M4"))
    return ix.search_batch(c["queries"], c["k"], L=c.get("L", 100), beam_width=c.get("bw", 0) or 0, mode=mode,
                           band_policy=c.get("policy", 0), flags=flags)


@pytest.mark.parametrize("name,ci", all_cases(modes=("M1",), pred=lambda c: not c.get("f64")))
def test_m1_bit_exact_vs_reference(name, ci):
    """M1 through the C ABI == the reference's own output: ids, distance bits, counts and the four counters."""
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, st = run_case(name, c)
    assert (st["status"] == 0).all()
    assert np.array_equal(cnt, c["count"])
    assert np.array_equal(ids, c["ids"])
    assert np.array_equal(bits(dist), bits(c["dist"]))
    got = np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1)
    assert np.array_equal(got, c["stats"])


@pytest.mark.parametrize("name,ci", all_cases(modes=("M3",), pred=lambda c: c["use_pq"]))
def test_m3_pq_bit_exact_vs_reference(name, ci):
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, st = run_case(name, c)
    assert (st["status"] == 0).all()
    assert np.array_equal(cnt, c["count"])
    assert np.array_equal(ids, c["ids"])
    assert np.array_equal(bits(dist), bits(c["dist"]))


@pytest.mark.parametrize("name,ci", all_cases(modes=("M2", "M4")) + all_cases(modes=("M3",), pred=lambda c: not c["use_pq"]))
def test_exact_modes_bit_exact_vs_oracle(name, ci):
    """M2 / M4 / M3-without-PQ: the reference's distance goes through BLAS or a -ffast-math loop, so the device
    is held bit-exact to the ORACLE (same traversal, numpy summation order) and to the reference within 1e-4."""
    from oracle import pyoracle as orc
    g = load_golden(name)
    c = g.case(ci)
    ids, dist, cnt, st = run_case(name, c)
    assert (st["status"] == 0).all()
    omode = {"M2": orc.M2, "M3": orc.M3, "M4": orc.M4}[c["mode"]]
    oflags = (orc.F_CYTHON if c.get("cython") else 0) | orc.F_PAIRWISE
    adj = g.adj if c["mode"] == "M2" else g.mem_adj
    oids, odist, ocnt, _ = orc.search_batch(g.vectors, adj, c["queries"], g.medoid, omode, c["k"], L=c.get("L", 100),
                                            bw=c.get("bw", 0) or 0, flags=oflags, codes=g.codes, codebook=g.codebook)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(dist), bits(odist.astype(np.float32)))
    assert np.array_equal(cnt, c["count"])
    if c["mode"] != "M4":
        for qi in range(len(cnt)):
            n = cnt[qi]
            np.testing.assert_allclose(np.sort(dist[qi, :n]), np.sort(c["dist"][qi, :n]), rtol=1e-4)


# ------------------------------------------------------------------------------------ search parity (oracle, wider)

@pytest.mark.parametrize("L,bw,k", [(100, 0, 10), (100, 8, 10), (64, 0, 64), (65, 5, 30), (128, 0, 20), (129, 16, 10),
                                    (256, 0, 50), (300, 0, 10), (1, 0, 1), (2, 1, 2)])
@pytest.mark.parametrize("name", ["sift128_R64_m32", "randn128_R16_m32"])
def test_m1_vs_oracle_capacity_sweep(name, L, bw, k):
    """Result-list capacities across every size class of the kernel (64/128/256/512), incl. odd sizes."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    ids, dist, cnt, st = ix.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_M1)
    oids, odist, ocnt, ost = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M1, k, L=L, bw=bw,
                                              codes=g.codes, codebook=g.codebook)
    assert (st["status"] == 0).all()
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(dist), bits(odist.astype(np.float32)))
    assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)


def test_many_queries_persistent_workgroups():
    """More queries than resident workgroups: the ticket loop and generation-tagged visited tables are reused."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    rs = np.random.RandomState(3)
    q = g.vectors[rs.randint(0, len(g.vectors), size=3000)] + rs.randint(-3, 4, size=(3000, 128)).astype(np.float32)
    ids, dist, cnt, st = ix.search_batch(q, 10, L=50, beam_width=0, mode=_ffi.MODE_M1)
    oids, odist, ocnt, ost = orc.search_batch(g.vectors, g.adj, q, g.medoid, orc.M1, 10, L=50, bw=0, codes=g.codes,
                                              codebook=g.codebook, nthreads=8)
    assert (st["status"] == 0).all()
    assert np.array_equal(ids, oids)
    assert np.array_equal(bits(dist), bits(odist.astype(np.float32)))
    # second run on the same handle must give identical results (generation counters advance)
    ids2, dist2, _, _ = ix.search_batch(q, 10, L=50, beam_width=0, mode=_ffi.MODE_M1)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(dist), bits(dist2))


def test_error_behaviour():
    from diskrag_amd import DiskragHipError, HipIndex, _ffi
    g = load_golden("faq32_R16_nopq")
    ix = get_index("faq32_R16_nopq")
    with pytest.raises(DiskragHipError):       # M1 without PQ data
        ix.search_batch(g.queries, 5, L=20, mode=_ffi.MODE_M1)
    with pytest.raises(ValueError):            # wrong query dimension (search_engine.py:547-551)
        ix.search_batch(np.zeros((1, 7), dtype=np.float32), 5)
    with pytest.raises(DiskragHipError):       # neighbour id out of range
        bad = g.adj.copy()
        bad[3, 2] = 1000
        HipIndex.create(g.vectors, bad, g.medoid)
    # a dimension without compiled kernels is served by the generic traversal since round 6 (tests/test_gpu_shapes.py); what needs a compiled one says so
    odd = HipIndex.create(np.zeros((4, 7), dtype=np.float32), np.zeros((4, 2), dtype=np.uint32), 0)
    with pytest.raises(DiskragHipError) as e:
        odd.build_vamana(L_build=4)
    assert e.value.code == _ffi.E_UNSUPPORTED
    odd.close()
    with pytest.raises(DiskragHipError):       # a dimension beyond the generic traversal's stack (32768)
        HipIndex.create(np.zeros((1, 40000), dtype=np.float32), np.zeros((1, 2), dtype=np.uint32), 0)


def test_large_batch_is_chunked():
    """dr_search_batch splits batches larger than its per-query scratch budget (32768) transparently."""
    from diskrag_amd import _ffi
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    c = g.case(2)   # L=20, bw=8
    reps = 40000 // len(g.queries) + 1
    q = np.tile(g.queries, (reps, 1))[:40000]
    ids, dist, cnt, st = ix.search_batch(q, 10, L=20, beam_width=8, mode=_ffi.MODE_M1)
    want = np.tile(c["ids"], (reps, 1))[:40000]
    assert (st["status"] == 0).all()
    assert np.array_equal(ids, want)
    assert np.array_equal(bits(dist), bits(np.tile(c["dist"], (reps, 1))[:40000]))


@pytest.mark.parametrize("name", ["sift128_R64_m32", "randn128_R64_m16", "unit1536_R16_m64", "deep96_R32_m16"])
def test_flat_pq_scan_matches_adc_and_finds_the_nearest_code(name):
    """dr_pq_scan_best (the isolated ADC kernel): every distance == A3 of the reference (oracle), bit for bit, and
    the folded best key is the argmin with the smallest id among equal sums."""
    from oracle import pyoracle as orc
    g = load_golden(name)
    ix = get_index(name)
    q = g.queries[:5]
    bid, bsq, ms, allsq = ix.pq_scan_best(q, want_output=True)
    for qi in range(len(q)):
        lut = orc.build_lut(g.codebook, q[qi])
        want = orc.adc(lut, g.codes)[0]
        assert np.array_equal(bits(allsq[qi]), bits(want))
        assert int(bid[qi]) == int(np.flatnonzero(want == want.min())[0])
        assert bits(np.float32(bsq[qi])) == bits(want.min())
    only_best = ix.pq_scan_best(q)
    assert np.array_equal(only_best[0], bid) and np.array_equal(bits(only_best[1]), bits(bsq))


@pytest.mark.parametrize("D,m", [(64, 16), (128, 32), (96, 48), (128, 64)])
def test_flat_pq_scan_several_queries_per_pass(D, m, monkeypatch):
    """nq >= 2: pq_scan_multi_kernel (one pass over the code words serves 4 -- m <= 32 -- or 2 queries, tables interleaved in LDS) == A3 of the
    reference per query (oracle), bit for bit, == the one-query-per-block-row kernel; ragged last groups, n not a multiple of the block."""
    from diskrag_amd import HipIndex
    from oracle import pyoracle as orc
    rs = np.random.RandomState(11 + m)
    n = 5231
    codes = rs.randint(0, 256, size=(n, m)).astype(np.uint8)
    codes[100] = codes[4000]      # (equal sums: the smaller id wins)
    cb = rs.randn(m, 256, D // m).astype(np.float32)
    ix = HipIndex.create_codes(np.zeros((n, 1), dtype=np.uint32), 0, D, cb, codes)
    try:
        for nq in (2, 3, 4, 5, 9):
            q = rs.randn(nq, D).astype(np.float32)
            q[-1] = cb[:, 7, :].reshape(-1) if nq == 3 else q[-1]
            bid, bsq, ms, allsq = ix.pq_scan_best(q, want_output=True)
            bid2, bsq2, _ = ix.pq_scan_best(q)
            monkeypatch.setenv("DR_PQ_SCAN_PER_QUERY", "1")
            bid1, bsq1, _, all1 = ix.pq_scan_best(q, want_output=True)
            monkeypatch.delenv("DR_PQ_SCAN_PER_QUERY")
            assert np.array_equal(bits(allsq), bits(all1)) and np.array_equal(bid, bid1) and np.array_equal(bits(bsq), bits(bsq1))
            for qi in range(nq):
                want = orc.adc(orc.build_lut(cb, q[qi]), codes)[0]
                assert np.array_equal(bits(allsq[qi]), bits(want))
                assert int(bid[qi]) == int(np.flatnonzero(want == want.min())[0]) == int(bid2[qi])
                assert bits(np.float32(bsq[qi])) == bits(want.min()) == bits(np.float32(bsq2[qi]))
    finally:
        ix.close()


@pytest.mark.parametrize("D,m", [(64, 16), (128, 32), (96, 48), (128, 64)])
def test_pq_scan_topk_several_queries_per_pass(D, m, monkeypatch):
    """dr_pq_scan_topk with nq >= 2 (pq_scan_topk_multi_kernel: 4 or 2 queries share a pass over the code words) == the k smallest (distance, id)
    pairs of the flat scan (pinned on the oracle above) == the one-query-per-block-row kernel; duplicated code words: the smaller id wins."""
    from diskrag_amd import HipIndex
    rs = np.random.RandomState(5 + m)
    n = 70001
    codes = rs.randint(0, 256, size=(n, m)).astype(np.uint8)
    codes[n // 2:n // 2 + 3000] = codes[:3000]
    cb = rs.randn(m, 256, D // m).astype(np.float32)
    ix = HipIndex.create_codes(np.zeros((n, 1), dtype=np.uint32), 0, D, cb, codes)
    try:
        for nq, k in ((2, 10), (5, 64), (9, 1), (23, 10)):
            q = rs.randn(nq, D).astype(np.float32)
            full = ix.pq_scan_best(q, want_output=True)[3]
            ids, sq, _ = ix.pq_scan_topk(q, k)
            monkeypatch.setenv("DR_PQ_SCAN_PER_QUERY", "1")
            ids1, sq1, _ = ix.pq_scan_topk(q, k)
            monkeypatch.delenv("DR_PQ_SCAN_PER_QUERY")
            assert np.array_equal(ids, ids1) and np.array_equal(bits(sq), bits(sq1))
            for qi in range(nq):
                order = np.lexsort((np.arange(n), full[qi]))[:k]
                assert np.array_equal(ids[qi], order.astype(np.uint32)), (nq, k, qi)
                assert np.array_equal(bits(sq[qi]), bits(full[qi][order]))
    finally:
        ix.close()


def test_flat_pq_scan_generic_m():
    """m not a multiple of 16 (D=96, m=24): the generic scan path; same contract as the fast kernel."""
    from diskrag_amd import HipIndex
    from oracle import pyoracle as orc
    rs = np.random.RandomState(3)
    n, D, m = 3000, 96, 24
    codes = rs.randint(0, 256, size=(n, m)).astype(np.uint8)
    cb = rs.randn(m, 256, D // m).astype(np.float32)
    q = rs.randn(4, D).astype(np.float32)
    ix = HipIndex.create_codes(np.zeros((n, 1), dtype=np.uint32), 0, D, cb, codes)
    try:
        bid, bsq, ms, allsq = ix.pq_scan_best(q, want_output=True)
        bid2, bsq2, _ = ix.pq_scan_best(q)
        for qi in range(len(q)):
            want = orc.adc(orc.build_lut(cb, q[qi]), codes)[0]
            assert np.array_equal(bits(allsq[qi]), bits(want))
            assert int(bid[qi]) == int(np.flatnonzero(want == want.min())[0]) == int(bid2[qi])
            assert bits(np.float32(bsq[qi])) == bits(want.min()) == bits(np.float32(bsq2[qi]))
    finally:
        ix.close()


@pytest.mark.parametrize("name", ["randn128_R16_m32", "unit1536_R16_m32", "deep96_R32_m16"])
def test_m3_cosine_traversal(name):
    """DR_F_COSINE: the in-memory M3 with distance_metric='cosine' (vamana_graph.py:324-329, cython_utils.pyx:53-70) against
    the reference's goldens (tests/golden/gen_golden_cosine.py) at the reference's own tolerance for this kernel (1e-5; its
    -ffast-math summation order is unpinned) -- ids equal except where two distances lie within that tolerance."""
    import json
    from diskrag_amd import _ffi
    from tests.conftest import GOLDEN
    g = load_golden(name)
    z = np.load(GOLDEN / f"cos_{name}.npz")
    ix = get_index(name, mem=True)
    for ci, c in enumerate(json.loads(str(z["cases"]))):
        ids, dist, cnt, st = ix.search_batch(g.queries, c["k"], L=c["k"], beam_width=c["bw"], mode=_ffi.MODE_M3, flags=_ffi.F_COSINE)
        assert int(st["status"].max()) == 0
        w_ids, w_dist = z[f"c{ci}_ids"], z[f"c{ci}_dist"]
        assert np.array_equal(cnt, z[f"c{ci}_count"])
        valid = w_ids != 0xFFFFFFFF
        assert np.allclose(dist[valid], w_dist[valid], rtol=0, atol=1e-5), (name, c)
        for qi, pos in np.argwhere(ids != w_ids):      # a swap is only allowed between near-equal distances
            assert abs(float(dist[qi, pos]) - float(w_dist[qi, pos])) <= 1e-5
    with pytest.raises(_ffi.DiskragHipError):
        ix.search_batch(g.queries, 5, L=5, beam_width=8, mode=_ffi.MODE_M1, flags=_ffi.F_COSINE)
