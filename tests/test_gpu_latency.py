"""Variant 18 (csrc/latency_kernel.hpp): a workgroup of eight wavefronts per query for the handful of queries of one request
(search_engine.py:530-614, app.py:84-130). Held to the reference's goldens and the oracle bit for bit -- ids, distance bits,
counts and the four counters -- with the variant forced, and as the engine's own choice for small blocking calls. Needs an MI355X."""
import os

import numpy as np
import pytest

from tests.conftest import all_cases, load_golden
from tests.test_gpu_parity import bits, get_index, run_case

pytestmark = pytest.mark.gpu


class forced:
    """the variant pinned for a block (process-wide hook: always released)"""

    def __init__(self, ix, kind):
        self.ix, self.kind = ix, kind

    def __enter__(self):
        self.ix.debug_force_kind(self.kind)

    def __exit__(self, *a):
        self.ix.debug_force_kind(-1)


@pytest.mark.parametrize("name,ci", all_cases(modes=("M1",), pred=lambda c: not c.get("f64")))
def test_m1_goldens_with_the_workgroup_kernel(name, ci):
    g = load_golden(name)
    c = g.case(ci)
    ix = get_index(name)
    with forced(ix, 18):
        ids, dist, cnt, st = run_case(name, c)
        assert ix.timing()["variant"] == 18 and ix.timing()["block"] == 512
    assert (st["status"] == 0).all()
    assert np.array_equal(cnt, c["count"])
    assert np.array_equal(ids, c["ids"])
    assert np.array_equal(bits(dist), bits(c["dist"]))
    assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), c["stats"])


@pytest.mark.parametrize("name,ci", all_cases(modes=("M2", "M4")) + all_cases(modes=("M3",), pred=lambda c: not c["use_pq"]))
def test_exact_traversals_with_the_workgroup_kernel(name, ci):
    """M2 / M4 / M3 without PQ: the same lists as search_kernel.hpp (which test_gpu_parity holds to the oracle), bit for bit."""
    g = load_golden(name)
    c = g.case(ci)
    ix = get_index(name, mem=c["mode"] in ("M3", "M4"))
    ix.debug_force_kind(-1)
    os.environ["DR_NO_LATENCY"] = "1"        # (with long rows variant 18 is the engine's own choice for small launches: the other family here)
    try:
        w_ids, w_dist, w_cnt, w_st = run_case(name, c)
    finally:
        del os.environ["DR_NO_LATENCY"]
    assert ix.timing()["variant"] != 18
    with forced(ix, 18):
        ids, dist, cnt, st = run_case(name, c)
        assert ix.timing()["variant"] == 18
    assert (st["status"] == 0).all()
    assert np.array_equal(cnt, w_cnt) and np.array_equal(ids, w_ids) and np.array_equal(bits(dist), bits(w_dist))
    for f in ("steps", "visited", "exact", "pq", "inserts"):
        assert np.array_equal(st[f], w_st[f]), f


def test_exact_traversals_vs_oracle():
    from oracle import pyoracle as orc
    name = "sift128_R64_m32"
    g = load_golden(name)
    ix = get_index(name)
    from diskrag_amd import _ffi
    with forced(ix, 18):
        for (L, bw) in ((100, 8), (40, 16), (10, 4)):
            ids, dist, cnt, st = ix.search_batch(g.queries, 10, L=L, beam_width=bw, mode=_ffi.MODE_M2)
            assert ix.timing()["variant"] == 18 and (st["status"] == 0).all()
            oids, odist, ocnt, ost = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M2, 10, L=L, bw=bw, flags=orc.F_PAIRWISE)
            assert np.array_equal(ids, oids) and np.array_equal(cnt, ocnt)
            assert np.array_equal(bits(dist), bits(odist.astype(np.float32)))
            assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)


@pytest.mark.parametrize("D,m", [(128, 32), (96, 16), (256, 32), (768, 32), (1536, 32)])
def test_live_policy_small_calls_pick_the_workgroup_kernel(D, m):
    """Blocking calls of 1 ... 70 queries through variant 18 (after the one call per list-size class that measures the A4 regime): the oracle's
    results under both band policies, on data where the rerank policy really consults the ADC; and the engine's own rule for taking it."""
    from diskrag_amd import _ffi
    from diskrag_amd.synth import unit_mixture
    from oracle import pyoracle as orc
    from tests.test_gpu_live_regime import _index
    x, q = unit_mixture(12000, D, n_queries=70, n_clusters=64, seed=7, latent=24)
    ix, medoid, adj, cb, codes = _index(x, 32, m)
    os.environ["DR_LAT_ALL"] = "1"          # (every eligible small call; the engine's own rule -- lists shorter than 64 entries -- is checked at the end)
    try:
        for (L, bw, pol) in ((100, 8, 0), (20, 8, 1), (100, 0, 0), (150, 8, 1)):
            w = orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=L, bw=bw, policy=pol, codes=codes, codebook=cb, nthreads=8)
            for nq in (1, 1, 7, 64, 70):
                ids, dist, cnt, st = ix.search_batch(q[:nq], 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
                assert int(st["status"].max()) == 0
                assert np.array_equal(ids, w[0][:nq]), (L, bw, pol, nq)
                valid = w[0][:nq] != 0xFFFFFFFF
                assert np.array_equal(dist[valid].view(np.uint32), w[1][:nq][valid].astype(np.float32).view(np.uint32))
                assert np.array_equal(cnt, w[2][:nq])
                assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), w[3][:nq])
            assert ix.timing()["variant"] == 18                                 # (launches of up to 256 queries)
            # the scoring wavefronts may leave the ADC to the decisions (the engine's choice on data where the policy rarely asks): same bits
            for env in ("DR_LAT_LAZY_ADC", "DR_LAT_EAGER_ADC"):
                os.environ[env] = "1"
                try:
                    ids, dist, cnt, st = ix.search_batch(q[:9], 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
                finally:
                    del os.environ[env]
                assert ix.timing()["variant"] == 18 and int(st["status"].max()) == 0
                assert np.array_equal(ids, w[0][:9]) and np.array_equal(cnt, w[2][:9]), (env, L, bw, pol)
                valid = w[0][:9] != 0xFFFFFFFF
                assert np.array_equal(dist[valid].view(np.uint32), w[1][:9][valid].astype(np.float32).view(np.uint32))
                assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), w[3][:9])
            ix.search_batch(q[:3], 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
            assert ix.timing()["variant"] == 18 and ix.timing()["grid"] == 3     # a workgroup per query
        del os.environ["DR_LAT_ALL"]
        # the engine's own rule: long rows only -- M1 above 960 dimensions, the exact traversals above 256; launches of <= 256 queries
        for L in (20, 48, 64, 100):
            ix.search_batch(q[:2], 5, L=L, beam_width=8, mode=_ffi.MODE_M1)
            ix.search_batch(q[:2], 5, L=L, beam_width=8, mode=_ffi.MODE_M1)
            assert (ix.timing()["variant"] == 18) == (D > 960), L
        ix.search_batch(q[:2], 5, L=20, beam_width=8, mode=_ffi.MODE_M2)
        assert (ix.timing()["variant"] == 18) == (D > 256)
    finally:
        os.environ.pop("DR_LAT_ALL", None)
        ix.close()


def test_visited_set_spills_to_global_memory():
    """An LDS table of 256 slots: nearly every query continues its visited ids in the workgroup's global table -- same results, and the table
    is wiped behind each query (the second pass over the same queries and a different batch see an empty one)."""
    from diskrag_amd import _ffi
    name = "sift128_R64_m32"
    g = load_golden(name)
    ix = get_index(name)
    ix.debug_force_kind(-1)
    os.environ["DR_NO_LATENCY"] = "1"
    try:
        want = [ix.search_batch(g.queries, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1) for (L, bw) in ((100, 8), (30, 8), (200, 0))]
    finally:
        del os.environ["DR_NO_LATENCY"]
    os.environ["DR_LAT_VH_BITS"] = "8"
    try:
        with forced(ix, 18):
            for rep in range(2):
                for w, (L, bw) in zip(want, ((100, 8), (30, 8), (200, 0))):
                    got = ix.search_batch(g.queries, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1)
                    assert ix.timing()["variant"] == 18 and (got[3]["status"] == 0).all()
                    assert np.array_equal(got[0], w[0]) and np.array_equal(bits(got[1]), bits(w[1])) and np.array_equal(got[2], w[2])
                    for f in ("steps", "visited", "exact", "pq", "inserts"):
                        assert np.array_equal(got[3][f], w[3][f]), f
    finally:
        del os.environ["DR_LAT_VH_BITS"]
        ix.debug_force_kind(-1)


def test_one_query_requests_through_submit_and_wait():
    """The facade's one-query requests go through dr_search_submit / dr_search_wait (search_engine.py _one): with DR_LAT_ALL=1 launches of a
    handful of queries run the workgroup kernel there too -- the same bits as the batch kernels."""
    from diskrag_amd import _ffi
    name = "sift128_R64_m32"
    g = load_golden(name)
    ix = get_index(name)
    ix.debug_force_kind(-1)
    os.environ["DR_NO_LATENCY"] = "1"
    try:
        want = ix.search_batch(g.queries, 5, L=20, beam_width=8, mode=_ffi.MODE_M1)
    finally:
        del os.environ["DR_NO_LATENCY"]
    os.environ["DR_LAT_ALL"] = "1"
    try:
        for qi in range(6):
            ids, dist, cnt, st = ix.search_submit(g.queries[qi:qi + 1], 5, L=20, beam_width=8, mode=_ffi.MODE_M1).wait()
            assert ix.timing()["variant"] == 18
            assert np.array_equal(ids, want[0][qi:qi + 1]) and np.array_equal(bits(dist), bits(want[1][qi:qi + 1])) and int(st["status"][0]) == 0
        pend = [ix.search_submit(g.queries[qi:qi + 3], 5, L=20, beam_width=8, mode=_ffi.MODE_M1) for qi in range(0, 12, 3)]
        for i, pnd in reversed(list(enumerate(pend))):
            ids, dist, cnt, st = pnd.wait()
            assert np.array_equal(ids, want[0][3 * i:3 * i + 3]) and np.array_equal(bits(dist), bits(want[1][3 * i:3 * i + 3]))
            for f in ("steps", "visited", "exact", "pq"):
                assert np.array_equal(st[f], want[3][f][3 * i:3 * i + 3])
    finally:
        del os.environ["DR_LAT_ALL"]


def test_visited_set_overflow_falls_back():
    """A visited-id set too small for the query (a tiny LDS table and no global continuation): the forced variant reports status bit 0
    through the resident path, the small blocking call answers through search_kernel.hpp instead -- same results."""
    from diskrag_amd import _ffi
    name = "sift128_R64_m32"
    g = load_golden(name)
    ix = get_index(name)
    ix.debug_force_kind(-1)
    want = ix.search_batch(g.queries[:5], 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
    os.environ["DR_LAT_VH_BITS"] = "8"
    os.environ["DR_LAT_SPILL_BITS"] = "0"
    os.environ["DR_LAT_ALL"] = "1"
    try:
        got = ix.search_batch(g.queries[:5], 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
        assert ix.timing()["variant"] != 18
        assert np.array_equal(got[0], want[0]) and np.array_equal(bits(got[1]), bits(want[1]))
        for f in ("steps", "visited", "exact", "pq", "status"):
            assert np.array_equal(got[3][f], want[3][f])
        with forced(ix, 18):
            ix.batch_upload(g.queries[:5])
            ix.batch_run(10, L=100, beam_width=8, mode=_ffi.MODE_M1)
            _, _, _, st = ix.batch_download()
            assert ix.timing()["variant"] == 18 and ((st["status"] & 1) != 0).all()
        os.environ["DR_LAT_SPILL_BITS"] = "9"           # (a continuation that is itself too small: 512 slots)
        got = ix.search_batch(g.queries[:5], 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
        assert ix.timing()["variant"] != 18
        assert np.array_equal(got[0], want[0]) and np.array_equal(bits(got[1]), bits(want[1]))
        del os.environ["DR_LAT_VH_BITS"]
        del os.environ["DR_LAT_SPILL_BITS"]
        got = ix.search_batch(g.queries[:5], 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
        assert ix.timing()["variant"] == 18
        assert np.array_equal(got[0], want[0]) and np.array_equal(bits(got[1]), bits(want[1]))
    finally:
        os.environ.pop("DR_LAT_VH_BITS", None)
        os.environ.pop("DR_LAT_SPILL_BITS", None)
        os.environ.pop("DR_LAT_ALL", None)
        ix.debug_force_kind(-1)


def test_ties_and_long_lists():
    """Integer-valued data (tied distances: the tie-order pass reads the workgroup kernel's insert log) and every list-size class."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    name = "sift128_R64_m32"
    g = load_golden(name)
    ix = get_index(name)
    with forced(ix, 18):
        for (L, bw, k) in ((64, 8, 10), (128, 0, 10), (200, 8, 50), (300, 16, 10), (600, 8, 100), (1000, 8, 10)):
            ids, dist, cnt, st = ix.search_batch(g.queries, k, L=L, beam_width=bw, mode=_ffi.MODE_M1)
            assert ix.timing()["variant"] == 18 and (st["status"] == 0).all()
            oids, odist, ocnt, ost = orc.search_batch(g.vectors, g.adj, g.queries, g.medoid, orc.M1, k, L=L, bw=bw, codes=g.codes,
                                                      codebook=g.codebook, nthreads=8)
            assert np.array_equal(ids, oids), (L, bw, k)
            valid = oids != 0xFFFFFFFF
            assert np.array_equal(dist[valid].view(np.uint32), odist[valid].astype(np.float32).view(np.uint32))
            assert np.array_equal(cnt, ocnt)
            assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)


@pytest.mark.parametrize("live", [False, True])
def test_ask_later_on_short_lists(live):
    """search_kernel.hpp's "ask later" (lists of at most 64 entries: every new row first, the rerank policy's ADC only when the sharper test
    fails): the batch kernels with it and without it (DR_NO_ASK_LATER=1) against the oracle, on SIFT-scale data (the test nearly always
    proves the policy true) and on unit-norm data (it nearly never does: the late evaluation runs), every M1 variant."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    os.environ["DR_NO_LATENCY"] = "1"
    try:
        if live:
            from diskrag_amd.synth import unit_mixture
            from tests.test_gpu_live_regime import _index
            x, q = unit_mixture(12000, 128, n_queries=96, n_clusters=64, seed=11, latent=24)
            ix, medoid, adj, cb, codes = _index(x, 32, 32)
            kinds = (-1, 0, 3, 9)
        else:
            g = load_golden("sift128_R64_m32")
            ix, x, q, adj, medoid, cb, codes = get_index("sift128_R64_m32"), g.vectors, g.queries, g.adj, g.medoid, g.codebook, g.codes
            kinds = (-1, 0, 3, 9, 11, 13, 17)
        for (L, bw, pol) in ((20, 8, 0), (20, 8, 1), (48, 0, 0), (64, 8, 1), (10, 4, 0)):
            w = orc.search_batch(x, adj, q, medoid, orc.M1, 5, L=L, bw=bw, policy=pol, codes=codes, codebook=cb, nthreads=8)
            evaluated = {}
            for env in (None, "1"):
                if env: os.environ["DR_NO_ASK_LATER"] = env
                else: os.environ.pop("DR_NO_ASK_LATER", None)
                for kind in kinds:
                    ix.debug_force_kind(kind)
                    ids, dist, cnt, st = ix.search_batch(q, 5, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
                    assert int(st["status"].max()) == 0
                    assert np.array_equal(ids, w[0]) and np.array_equal(cnt, w[2]), (live, L, bw, pol, env, kind)
                    valid = w[0] != 0xFFFFFFFF
                    assert np.array_equal(dist[valid].view(np.uint32), w[1][valid].astype(np.float32).view(np.uint32))
                    assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), w[3])
                    if kind == 0: evaluated[env] = int(st["pq_evaluated"].sum())
            if not live and L < 64:
                assert evaluated[None] < evaluated["1"], (L, evaluated)      # the sharper test spares evaluations (a third of them on this 2 000-point fixture, four fifths on the 1M bench index)
    finally:
        os.environ.pop("DR_NO_LATENCY", None)
        os.environ.pop("DR_NO_ASK_LATER", None)
        ix.debug_force_kind(-1)
        if live: ix.close()


def test_tie_replay_by_the_whole_wavefront_equals_the_one_lane_replay():
    """finalize_kernel's wave-parallel heappush / heappop (capacity <= 128) against the one-lane form (DR_FINALIZE_SERIAL=1) and the oracle's
    CPython heap, on data made of few distinct values (nearly every query ties inside its first k) at capacities around the switch."""
    from diskrag_amd import _ffi
    from oracle import pyoracle as orc
    from tests.test_gpu_live_regime import _index
    rs = np.random.RandomState(17)
    x = rs.randint(0, 12, size=(6000, 128)).astype(np.float32)         # squared distances are small integers: ties in most lists (values 0..3 tie so often that the 64-entry side list of tied evictions overflows: status bit 1)
    q = rs.randint(0, 12, size=(150, 128)).astype(np.float32)
    ix, medoid, adj, cb, codes = _index(x, 32, 32)
    try:
        for (L, bw, k) in ((100, 8, 10), (128, 0, 50), (64, 8, 64), (20, 8, 5), (129, 8, 20), (7, 4, 7), (300, 16, 100)):
            w = orc.search_batch(x, adj, q, medoid, orc.M1, k, L=L, bw=bw, codes=codes, codebook=cb, nthreads=8)
            res = {}
            for env in (None, "1"):
                if env: os.environ["DR_FINALIZE_SERIAL"] = env
                else: os.environ.pop("DR_FINALIZE_SERIAL", None)
                for nq in (150, 3):          # the general path and the direct path of small calls
                    ids, dist, cnt, st = ix.search_batch(q[:nq], k, L=L, beam_width=bw, mode=_ffi.MODE_M1)
                    assert int(st["status"].max()) == 0
                    assert np.array_equal(ids, w[0][:nq]), (L, bw, k, env, nq)
                    valid = w[0][:nq] != 0xFFFFFFFF
                    assert np.array_equal(dist[valid].view(np.uint32), w[1][:nq][valid].astype(np.float32).view(np.uint32))
            # the data really ties: a good share of the queries hold equal distances inside their first k (a quarter at k = 10)
            d = w[1]
            assert (np.diff(d, axis=1) == 0).any(axis=1).mean() > (0.15 if k >= 10 else 0.03), (L, bw, k)
        # M2 sorts full tuples (no replay), M4 / M3 replay with their own keys
        for (mode, omode, k, L, bw, fl, ofl) in ((_ffi.MODE_M4, orc.M4, 10, 50, 0, _ffi.F_SQDIST, orc.F_CYTHON | orc.F_PAIRWISE), (_ffi.MODE_M2, orc.M2, 8, 0, 8, 0, orc.F_PAIRWISE)):
            w = orc.search_batch(x, adj, q, medoid, omode, k, L=L, bw=bw, flags=ofl, nthreads=8)
            ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=mode, flags=fl)
            assert np.array_equal(ids, w[0]) and np.array_equal(dist.view(np.uint32), w[1].astype(np.float32).view(np.uint32))
    finally:
        os.environ.pop("DR_FINALIZE_SERIAL", None)
        ix.close()
