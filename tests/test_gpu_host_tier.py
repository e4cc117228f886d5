"""The host tier of the stored vectors (dr_index_*_tiered with DR_TIER_HOST; SURVEY.md 8f N3's optional tier, the
reference's counterpart being MMapNodeReader over index.dat, pydiskann/io/diskann_persist.py:201-234): graph and code
words in HBM, full-precision rows in pinned host memory read by the same kernels. Same bits as the HBM-resident index and
as the reference's goldens in every mode; an index.dat opened straight into the tier; the builder over host-tier rows."""
import numpy as np
import pytest

from tests.conftest import INDEX_FIXTURES, load_golden
from tests.test_gpu_parity import bits, get_index

pytestmark = pytest.mark.gpu

_MODES = {"M1": 1, "M2": 2, "M3": 3, "M4": 4}


def _host_index(name, mem=False):
    from diskrag_amd import HipIndex, _ffi
    g = load_golden(name)
    ix = HipIndex.create(g.vectors, g.mem_adj if mem else g.adj, g.medoid, vector_tier=_ffi.TIER_HOST)
    if g.m:
        ix.set_pq(g.codebook, g.codes)
    return ix


def _stats(st):
    return np.stack([st["steps"], st["visited"], st["exact"], st["pq"], st["status"]], axis=1)


@pytest.mark.parametrize("name", list(INDEX_FIXTURES))
def test_every_golden_case_from_the_host_tier(name):
    """Every case the reference produced for this fixture, run on a host-tier index: the reference's ids, distance bits
    and counters where the golden is bit-pinned (M1, M3 with PQ), and bit-identical to the HBM-resident index in all."""
    from diskrag_amd import _ffi
    g = load_golden(name)
    tiers = {}
    try:
        for ci in range(len(g.cases)):
            c = g.case(ci)
            if c.get("f64"):
                continue
            mem = c["mode"] in ("M3", "M4")
            if mem not in tiers:
                tiers[mem] = _host_index(name, mem)
            flags = 0
            if c["mode"] == "M3" and c["use_pq"]:
                flags |= _ffi.F_USE_PQ
            if c["mode"] == "M4" and c.get("cython"):
                flags |= _ffi.F_SQDIST
            kw = dict(L=c.get("L", 100), beam_width=c.get("bw", 0) or 0, mode=_MODES[c["mode"]], band_policy=c.get("policy", 0), flags=flags)
            a = tiers[mem].search_batch(c["queries"], c["k"], **kw)
            b = get_index(name, mem=mem).search_batch(c["queries"], c["k"], **kw)
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
            assert np.array_equal(_stats(a[3]), _stats(b[3]))
            if c["mode"] == "M1" or (c["mode"] == "M3" and c["use_pq"]):
                assert np.array_equal(a[0], c["ids"]) and np.array_equal(bits(a[1]), bits(c["dist"])) and np.array_equal(a[2], c["count"])
    finally:
        for ix in tiers.values():
            ix.close()


def test_pq_traversal_with_rerank_reads_the_rows_from_host_memory():
    """The mode the tier is for: DR_MODE_PQ walks the graph on code words in HBM, DR_F_RERANK scores the L list against the
    full-precision rows -- here in host memory. Same output as the HBM index, and as the oracle."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    name = "unit1536_R16_m32"
    g = load_golden(name)
    q = g.case(0)["queries"]
    host = _host_index(name)
    try:
        for L, bw in ((32, 0), (100, 8)):
            a = host.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
            b = get_index(name).search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
            assert int(a[3]["status"].max()) == 0
            assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
            w = orc.search_batch(g.vectors, g.adj, q, g.medoid, orc.PQ, 10, L=L, bw=bw, flags=orc.F_RERANK, codes=g.codes, codebook=g.codebook)
            assert np.array_equal(a[0], w[0]) and np.array_equal(bits(a[1]), bits(w[1].astype(np.float32)))
        # the kernel-level seams read the tier too
        ids = np.arange(0, len(g.vectors), 7, dtype=np.uint32)[:64]
        assert np.array_equal(bits(host.exact_distances(q[:4], ids)), bits(get_index(name).exact_distances(q[:4], ids)))
        v, nb = host.get_node(5)
        assert np.array_equal(bits(v), bits(g.vectors[5]))
    finally:
        host.close()


def test_index_dat_opens_into_the_host_tier_and_the_facade_serves_it(tmp_path):
    """An index directory in the reference's formats opened with vector_tier='host': the facade's answers are those of the
    HBM-resident engine on the same files."""
    from diskrag_amd.search_engine import SearchEngineCorrect
    from tests.test_gpu_facade import write_collection
    g = load_golden("sift128_R64_m32")
    write_collection(tmp_path, "col", g)
    hbm = SearchEngineCorrect("col", base_dir=tmp_path)
    host = SearchEngineCorrect("col", base_dir=tmp_path, vector_tier="host")
    with pytest.raises(ValueError):
        SearchEngineCorrect("col", base_dir=tmp_path, vector_tier="disk")
    c = g.case(1)       # L = 100, beam_width = 8 (the API default), k = 10
    for eng in (host, hbm):
        for qi in range(6):
            res, stats = eng._pq_accelerated_graph_search(g.queries[qi], k=10, L=100, beam_width=8)
            assert [int(i) for _, i in res] == [int(i) for i in c["ids"][qi][:c["count"][qi]]]
            assert np.array_equal(np.array([d for d, _ in res], dtype=np.float32).view(np.uint32), c["dist"][qi][:c["count"][qi]].view(np.uint32))
    for qi in range(6):      # M2 through the facade (beam_width 8 hard-coded by the reference, Q6)
        ra, _ = host._exact_graph_search(g.queries[qi], k=10, L=100)
        rb, _ = hbm._exact_graph_search(g.queries[qi], k=10, L=100)
        assert len(ra) > 0 and [(float(d), int(i)) for d, i in ra] == [(float(d), int(i)) for d, i in rb]
    a = host.search_batch(g.queries, k=10, L=100, beam_width=8)
    b = hbm.search_batch(g.queries, k=10, L=100, beam_width=8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[0], c["ids"])
    host.close(); hbm.close()


def test_builder_over_host_tier_rows_builds_the_same_graph():
    """dr_build_vamana reads its rows through the same pointer: a graph built over host-tier rows is the graph built over
    HBM rows, bit for bit (the byte-row copy is the only thing a host-tier index does not make)."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like
    x, _ = sift_like(6000, 128, n_queries=4, seed=3)
    a = HipIndex.create_empty(x, R=24)
    b = HipIndex.create_empty(x, R=24, vector_tier=_ffi.TIER_HOST)
    try:
        ma, _ = a.build_vamana(L_build=40, alpha=1.2, passes=2, seed=5)
        mb, _ = b.build_vamana(L_build=40, alpha=1.2, passes=2, seed=5)
        assert ma == mb and np.array_equal(a.get_adjacency(), b.get_adjacency())
    finally:
        a.close(); b.close()
    with pytest.raises(_ffi.DiskragHipError):
        HipIndex.create_empty(x, R=24, vector_tier=7)


@pytest.mark.parametrize("tier", ["hbm", "host"])
def test_rows_streamed_in_chunks_give_the_same_index(tier):
    """dr_index_write_rows (round 4): an index filled chunk by chunk -- the form an index larger than host memory twice needs --
    builds the same graph and answers the same bits as one created from the whole array."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like
    x, q = sift_like(6000, 128, n_queries=40, n_clusters=16, seed=9, query_seed=10)
    vt = _ffi.TIER_HOST if tier == "host" else _ffi.TIER_HBM
    a = HipIndex.create_empty(x, R=32, vector_tier=vt)
    b = HipIndex.create_rows_empty(len(x), 128, 32, vector_tier=vt)
    try:
        for r0 in (3000, 0, 5000, 1000):                    # any order, uneven chunks
            r1 = {3000: 5000, 0: 1000, 5000: 6000, 1000: 3000}[r0]
            b.write_rows(x[r0:r1], r0)
        outs = []
        for ix in (a, b):
            medoid, _ = ix.build_vamana(L_build=50, alpha=1.2, passes=2, seed=4, pad_with_zero=True)
            cb = ix.pq_train(32, n_sample=6000, iters=3)
            ix.pq_encode(cb)
            outs.append((medoid, ix.get_adjacency(), ix.search_batch(q, 10, L=60, beam_width=8, mode=_ffi.MODE_M1),
                         ix.search_batch(q, 8, L=0, beam_width=8, mode=_ffi.MODE_M2)))
        assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
        for s in (2, 3):
            assert np.array_equal(outs[0][s][0], outs[1][s][0]) and np.array_equal(outs[0][s][1].view(np.uint32), outs[1][s][1].view(np.uint32))
        vec, _ = b.get_node(4321)
        assert np.array_equal(vec.view(np.uint32), x[4321].view(np.uint32))
        with pytest.raises(_ffi.DiskragHipError):
            b.write_rows(x[:10], len(x) - 5)                # outside the index
    finally:
        a.close(); b.close()
