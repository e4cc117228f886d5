"""DR_MODE_PQB (round 5): the engine's batch-per-step PQ-only beam search (csrc/pqb_kernel.hpp) against the oracle's restatement
of it (oracle/diskrag_oracle.c pqb_search_one) -- ids, distance bits, counts and counters bit for bit -- on full indexes and
PQ-only shards, every list-size class, one to eight frontier entries per step, rows narrower and wider than a wavefront, inline
neighbour codes, the exact rerank, every table layout (DR_PQB_TREG), the pipelined path; and against DR_MODE_PQ in recall terms."""
import numpy as np
import pytest

from tests.conftest import load_golden
from tests.test_gpu_parity import bits, get_index

pytestmark = pytest.mark.gpu


def _stats4(st):
    return np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1)


def _check(eng, g, k, L, bw, pops, flags=0, tag=None):
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    top = (flags >> 12) & 1023
    w = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.PQB, k, L=L, bw=bw,
                         flags=orc.F_POPS(pops) | (orc.F_RERANK if flags & _ffi.F_RERANK else 0) | orc.F_RERANK_TOP(top), codes=g.codes, codebook=g.codebook)
    ids, dist, cnt, st = eng.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=flags | _ffi.F_POPS(pops))
    assert int(st["status"].max()) == 0, tag
    assert np.array_equal(ids, w[0]), tag
    assert np.array_equal(cnt, w[2]), tag
    valid = w[0] != 0xFFFFFFFF
    assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid]), tag
    assert np.array_equal(_stats4(st), w[3]), tag
    return ids


SHAPES = ["randn128_R16_m32", "sift128_R64_m32", "sift128_R16_m32", "unit1536_R16_m32", "unit1536_R16_m64", "deep96_R32_m16",
          "unit768_R16_m96", "unit256_R16_m128", "unit256_R16_m4", "randn128_R64_m16"]


@pytest.mark.parametrize("name", SHAPES)
def test_pqb_matches_its_oracle_restatement(name):
    from diskrag_amd import HipIndex, _ffi
    g = load_golden(name)
    ix = get_index(name)
    shard = HipIndex.create_codes(g.adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes)
    try:
        for (L, bw, k) in ((100, 8, 10), (40, 0, 10), (10, 3, 10), (200, 16, 25), (64, 8, 64), (1, 1, 1), (300, 0, 10), (600, 32, 10), (1024, 0, 10)):
            for pops in (0, 1, 2, 4):      # (0: the default -- the rows that fill 64 neighbour slots)
                if pops * (1 << int(np.ceil(np.log2(g.R)))) > 256:
                    continue
                for eng, inline in ((ix, False), (shard, False), (shard, True)):
                    if inline and g.m % 4:
                        continue
                    eng.inline_codes(inline)
                    _check(eng, g, k, L, bw, pops, tag=(name, L, bw, k, pops, inline))
                shard.inline_codes(False)
            _check(ix, g, k, L, bw, 1, flags=_ffi.F_RERANK, tag=(name, L, bw, k, "rerank"))
        with pytest.raises(_ffi.DiskragHipError):        # the rerank needs the stored vectors
            shard.search_batch(g.queries, 10, L=50, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK)
        with pytest.raises(_ffi.DiskragHipError):        # pops belong to DR_MODE_PQB
            ix.search_batch(g.queries, 10, L=50, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_POPS(2))
    finally:
        shard.close()


def test_pqb_eight_pops_and_wide_rows():
    """pops * next_pow2(R) up to 256 slots per step: eight rows of 16 per step, and R = 64 rows four at a time."""
    g = load_golden("unit1536_R16_m32")
    ix = get_index("unit1536_R16_m32")
    for pops in (3, 5, 8):
        _check(ix, g, 10, 100, 16, pops, tag=("R16", pops))
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    for pops in (3, 4):
        _check(ix, g, 10, 100, 8, pops, tag=("R64", pops))
    from diskrag_amd import _ffi
    with pytest.raises(_ffi.DiskragHipError):            # 8 x 64 slots per step is more than the kernel family holds
        ix.search_batch(g.queries, 10, L=50, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(8))


@pytest.mark.parametrize("treg", ["16", "24"])
def test_pqb_table_layouts_return_the_same_bits(treg, monkeypatch):
    """m = 32: the last 16 / the last 24 table rows in registers (ds_bpermute lookups), the rest in LDS -- the same sums in the same order (the oracle's)."""
    monkeypatch.setenv("DR_PQB_TREG", treg)
    for name in ("unit1536_R16_m32", "sift128_R64_m32"):
        g = load_golden(name)
        ix = get_index(name)
        for (L, bw, k, pops) in ((100, 8, 10, 1), (300, 0, 10, 2), (64, 8, 64, 4)):
            _check(ix, g, k, L, bw, pops, tag=(name, treg, L, bw, pops))


def test_pqb_the_same_node_through_several_rows_of_a_step():
    """Several frontier entries per step: a node reached through two of the step's rows is accepted once (round 6 finds the copies in the loop that
    ranks the accepted keys and takes the counts again without them) -- rows of 16 and 64 slots, up to eight rows per step, lists that fill
    (long candidate sets: the LDS form of that loop) and lists that are full (the v_readlane form)."""
    for name, pops in (("sift128_R64_m32", 4), ("unit1536_R16_m32", 8), ("randn128_R16_m32", 5), ("sift128_R16_m32", 8), ("deep96_R32_m16", 4)):
        g = load_golden(name)
        ix = get_index(name)
        for (L, bw, k) in ((100, 8, 10), (250, 0, 10), (40, 0, 10), (20, 2, 5)):
            _check(ix, g, k, L, bw, pops, tag=(name, pops, L, bw))


def test_pqb_rerank_of_the_adc_top_of_the_list():
    """DR_F_RERANK_TOP(n): only the n list entries with the smallest ADC are scored exactly (round 6: c3's rerank is a third of a call at D = 1536) --
    ids, distance bits, counts and the exact-distance counter against the restatement; n >= L is the whole list; refused without DR_F_RERANK."""
    from diskrag_amd import _ffi
    for name in ("unit1536_R16_m32", "sift128_R64_m32", "deep96_R32_m16"):
        g = load_golden(name)
        ix = get_index(name)
        for (L, bw, k, top) in ((100, 8, 10, 30), (250, 0, 10, 64), (64, 8, 20, 20), (100, 0, 10, 100), (100, 0, 10, 1000), (40, 0, 10, 5)):
            _check(ix, g, k, L, bw, 0, flags=_ffi.F_RERANK | _ffi.F_RERANK_TOP(top), tag=(name, L, bw, k, top))
        a = ix.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_RERANK_TOP(100))
        b = ix.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
        with pytest.raises(_ffi.DiskragHipError):
            ix.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK_TOP(10))
        with pytest.raises(_ffi.DiskragHipError):
            ix.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK | _ffi.F_RERANK_TOP(10))


def test_pqb_through_the_pipelined_path():
    """dr_search_submit / dr_search_wait with DR_MODE_PQB: coalesced tickets get the bits of blocking calls."""
    from diskrag_amd import _ffi
    g = load_golden("unit1536_R16_m32")
    ix = get_index("unit1536_R16_m32")
    kw = dict(L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(2))
    ref = ix.search_batch(g.queries, 10, **kw)
    jobs = [ix.search_submit(g.queries[i::3], 10, **kw) for i in range(3)]
    for i, j in enumerate(jobs):
        ids, dist, cnt, st = j.wait()
        assert np.array_equal(ids, ref[0][i::3]) and np.array_equal(bits(dist), bits(ref[1][i::3])) and np.array_equal(cnt, ref[2][i::3])


def test_pqb_finds_what_the_sequential_traversal_finds():
    """Same L, same frontier trim: the batch-per-step statement returns (nearly) the list DR_MODE_PQ returns -- recall against the
    brute-force ADC ranking is what it is held to at scale (scripts/ab_pqb.py); here: overlap of the two top-10 lists."""
    from diskrag_amd import _ffi
    for name in ("unit1536_R16_m32", "deep96_R32_m16"):
        g = load_golden(name)
        ix = get_index(name)
        a = ix.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_PQ)[0]
        for pops in (1, 2, 4):
            b = ix.search_batch(g.queries, 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(pops))[0]
            ov = np.mean([len(set(x) & set(y)) / 10 for x, y in zip(a, b)])
            assert ov >= 0.97, (name, pops, ov)


def test_inner_product_flag_on_unit_norm_data():
    """DR_F_IP (BASELINE c3 / c5 name the inner product; the reference has none, vamana_graph.py:294-299): with the exact rerank on unit-norm rows
    the order is the squared-L2 order and out_dist = |q - v|^2 / 2 = 1 - <q, v>; refused on rows that are not unit-norm; a query that is not gets
    NaN and status bit 4."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    g = load_golden("unit1536_R16_m32")
    ix = get_index("unit1536_R16_m32")
    q = g.queries / np.linalg.norm(g.queries.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    for mode, omode in ((_ffi.MODE_PQB, orc.PQB), (_ffi.MODE_PQ, orc.PQ)):
        base = ix.search_batch(q, 10, L=100, beam_width=8, mode=mode, flags=_ffi.F_RERANK)
        ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=8, mode=mode, flags=_ffi.F_RERANK | _ffi.F_IP)
        assert int(st["status"].max()) == 0
        assert np.array_equal(ids, base[0]) and np.array_equal(bits(dist), bits(base[1] * np.float32(0.5)))
        w = orc.search_batch(g.vectors, g.adj, q, g.medoid, omode, 10, L=100, bw=8, flags=orc.F_RERANK | orc.F_IP, codes=g.codes, codebook=g.codebook)
        assert np.array_equal(ids, w[0]) and np.array_equal(bits(dist), bits(w[1].astype(np.float32)))
        ip = 1.0 - np.einsum("qd,qkd->qk", q.astype(np.float64), g.vectors[ids].astype(np.float64))
        assert np.abs(dist - ip).max() < 2e-6                                    # = 1 - <q, v> on unit vectors
    bad = q.copy(); bad[3] *= 1.5
    ids, dist, cnt, st = ix.search_batch(bad, 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_IP)
    assert st["status"][3] & 16 and np.isnan(dist[3]).all() and not np.isnan(np.delete(dist, 3, axis=0)).any()
    with pytest.raises(_ffi.DiskragHipError):                                    # SIFT-scale rows are not unit-norm
        get_index("sift128_R64_m32").search_batch(load_golden("sift128_R64_m32").queries, 10, L=50, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_IP)
    with pytest.raises(_ffi.DiskragHipError):                                    # the flag goes with the rerank
        ix.search_batch(q, 10, L=50, mode=_ffi.MODE_PQB, flags=_ffi.F_IP)


@pytest.mark.parametrize("R,dim,m", [(48, 64, 16), (128, 128, 32), (96, 96, 48), (32, 128, 64), (20, 64, 8)])
def test_pqb_on_device_built_graphs_with_ties(R, dim, m):
    """Shapes the goldens do not reach, on integer-valued data (ADC sums tie: the id decides), graphs built on the device: rows that are not a
    power of two (R = 48, 96, 20: lanes of a pass left empty), rows wider than a wavefront, m = 48 / 64 / 8 (the generic table-in-LDS kernel),
    odd list sizes, every pops setting the row width allows, > 256 queries (the general blocking path) -- ids, distance bits and counters
    against the restatement; and the PQ-only builder's DR_PAD-padded rows."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like
    from oracle import pyoracle as orc
    x, q = sift_like(24000, dim, n_queries=320, n_clusters=48, seed=71 + R, query_seed=72)
    x[12000:12160] = x[:160]            # duplicate points: the same code word, the same ADC sum -- only the id orders them
    q[:160] = x[:160]                   # ... and queries that sit on them
    ix = HipIndex.create_empty(x, R=R)
    try:
        medoid, _ = ix.build_vamana(L_build=70, alpha=1.2, passes=2, seed=3)
        cb = ix.pq_train(m, n_sample=8000, iters=3)
        codes = ix.pq_encode(cb, want_codes=True)
        adj = ix.get_adjacency()
        rs = 1 << int(np.ceil(np.log2(R)))
        for (L, bw, k) in ((37, 5, 10), (130, 0, 10), (300, 64, 20), (64, 64, 64)):
            for pops in (0, 1, 2, 3, 4):
                if pops * rs > 256:
                    continue
                w = orc.search_batch(x, adj, q, medoid, orc.PQB, k, L=L, bw=bw, flags=orc.F_POPS(pops), codes=codes, codebook=cb, nthreads=8)
                ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(pops))
                assert int(st["status"].max()) == 0
                assert np.array_equal(ids, w[0]) and np.array_equal(cnt, w[2]), (R, m, L, bw, pops)
                valid = w[0] != 0xFFFFFFFF
                assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid])
                assert np.array_equal(_stats4(st), w[3]), (R, m, L, bw, pops)
        assert (bits(dist)[:, 1:] == bits(dist)[:, :-1]).any()          # equal ADC sums among the results: the id broke the tie
        if m in (16, 32):       # the same points as a PQ-only shard with a graph built from the code words (rows padded with DR_PAD)
            sh = HipIndex.create_codes_empty(len(x), dim, R, cb)
            sh.encode_rows(x, 0)
            med2, _ = sh.build_vamana_pq(L_build=70, alpha=1.2, passes=2, seed=3)
            adj2 = sh.get_adjacency()
            w = orc.search_batch(x, adj2, q, med2, orc.PQB, 10, L=100, bw=16, codes=codes, codebook=cb, nthreads=8)
            ids, dist, cnt, st = sh.search_batch(q, 10, L=100, beam_width=16, mode=_ffi.MODE_PQB)
            assert np.array_equal(ids, w[0]) and np.array_equal(_stats4(st), w[3])
            sh.close()
    finally:
        ix.close()
