"""Round 4: coalesced launches of the pipelined path (dr_search_submit holds small submits that find the search stream busy
and launches them as ONE ticket space) and the 4-wavefront workgroup form of the byte-row kernels (variants 16 / 17).
Every ticket must carry exactly the bits of a blocking dr_search_batch call of its own -- which the other GPU tests hold
to the reference's goldens and the oracle (search_engine.py:398-506; app.py:84-130 is the one-query-per-request shape)."""
import numpy as np
import pytest

from tests.conftest import load_golden
from tests.test_gpu_parity import bits, get_index

pytestmark = pytest.mark.gpu


def _stats4(st):
    return np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1)


def _same(got, want):
    return (np.array_equal(got[0], want[0]) and np.array_equal(bits(got[1]), bits(want[1])) and np.array_equal(got[2], want[2])
            and np.array_equal(_stats4(got[3]), _stats4(want[3])) and np.array_equal(got[3]["status"], want[3]["status"]))


@pytest.mark.parametrize("name,params", [
    ("sift128_R64_m32", dict(L=100, beam_width=8, mode=1)),
    ("sift128_R64_m32", dict(L=40, beam_width=0, mode=1)),
    ("sift128_R64_m32", dict(L=0, beam_width=8, mode=2)),
    ("unit1536_R16_m32", dict(L=100, beam_width=8, mode=1, band_policy=1)),
    ("unit1536_R16_m32", dict(L=60, beam_width=8, mode=5)),
    ("deep96_R32_m16", dict(L=100, beam_width=8, mode=1)),
])
def test_held_submits_ride_in_one_launch_and_return_their_own_bits(name, params):
    g = load_golden(name)
    ix = get_index(name)
    rs = np.random.RandomState(11)
    sizes = (5, 1, 17, 3, 9, 1, len(g.queries), 2)
    batches = [np.ascontiguousarray(g.queries[rs.permutation(len(g.queries))[:n]]) for n in sizes]
    want = [ix.search_batch(b, 10, **params) for b in batches]
    ix.debug_hold(True)
    try:
        before = ix.pipeline_stats()
        jobs = [ix.search_submit(np.array(b), 10, **params) for b in batches]
        assert ix.pipeline_stats()["launches"] == before["launches"]          # all held
        order = rs.permutation(len(jobs))
        got = {}
        got[order[0]] = jobs[order[0]].wait()                                 # the first wait launches the whole group
        after = ix.pipeline_stats()
        assert after["launches"] == before["launches"] + 1 and after["tickets"] == before["tickets"] + len(jobs)
        assert after["max_tickets_per_launch"] >= len(jobs) and after["queries"] == before["queries"] + sum(len(b) for b in batches)
        for i in order[1:]:
            got[i] = jobs[i].wait()
        for i, w in enumerate(want):
            assert _same(got[i], w), (name, params, i)
    finally:
        ix.debug_hold(False)
    ix.batch_sync()


def test_groups_split_on_parameters_capacity_and_flush():
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    q = g.queries
    pa, pb = dict(L=100, beam_width=8, mode=1), dict(L=50, beam_width=8, mode=1)
    wa, wb, wk = ix.search_batch(q[:6], 10, **pa), ix.search_batch(q[6:12], 10, **pb), ix.search_batch(q[:6], 5, **pa)
    ix.debug_hold(True)
    try:
        s0 = ix.pipeline_stats()
        ja = ix.search_submit(q[:6], 10, **pa)
        jb = ix.search_submit(q[6:12], 10, **pb)          # other L: closes (launches) the first group, opens its own
        s1 = ix.pipeline_stats()
        assert s1["launches"] == s0["launches"] + 1
        jk = ix.search_submit(q[:6], 5, **pa)             # other k: the same
        assert ix.pipeline_stats()["launches"] == s0["launches"] + 2
        ix.search_flush()                                 # the third group, nobody waiting yet
        assert ix.pipeline_stats()["launches"] == s0["launches"] + 3
        assert _same(jk.wait(), wk) and _same(ja.wait(), wa) and _same(jb.wait(), wb)
        # capacity: a group of 16 queries takes two 6-query submits; the third does not fit and starts the next one
        ix.set_coalesce(16)
        s2 = ix.pipeline_stats()
        jobs = [ix.search_submit(q[:6], 10, **pa) for _ in range(3)]
        assert ix.pipeline_stats()["launches"] == s2["launches"] + 1
        for j in jobs:
            assert _same(j.wait(), wa)
        assert ix.pipeline_stats()["launches"] == s2["launches"] + 2
        # coalescing off: every submit is its own launch, at once
        ix.set_coalesce(0)
        s3 = ix.pipeline_stats()
        jobs = [ix.search_submit(q[:6], 10, **pa) for _ in range(3)]
        assert ix.pipeline_stats()["launches"] == s3["launches"] + 3
        for j in jobs:
            assert _same(j.wait(), wa)
    finally:
        ix.set_coalesce(32768)
        ix.debug_hold(False)
    ix.batch_sync()


def test_more_tickets_than_slots_and_a_stream_of_small_submits():
    """free-running policy (no hold hook): 200 small submits, at most MAX_TICKETS waited-for late, any grouping the timing gives"""
    from diskrag_amd import _ffi
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    rs = np.random.RandomState(5)
    params = dict(L=100, beam_width=8, mode=_ffi.MODE_M1)
    pool = [np.ascontiguousarray(g.queries[rs.permutation(len(g.queries))[:n]]) for n in (1, 4, 9, 16, 2, 30)]
    want = [ix.search_batch(b, 10, **params) for b in pool]
    jobs = []
    for i in range(200):
        jobs.append((i % len(pool), ix.search_submit(pool[i % len(pool)], 10, **params)))
        if len(jobs) > _ffi.MAX_TICKETS + 7:                   # older tickets were finished by the library when their slot was needed
            w, j = jobs.pop(0)
            assert _same(j.wait(), want[w])
    for w, j in jobs:
        assert _same(j.wait(), want[w])
    s = ix.pipeline_stats()
    assert s["tickets"] >= 200 and s["launches"] <= s["tickets"]
    ix.batch_sync()


def test_a_failed_launch_answers_every_ticket_that_rode_in_it():
    from diskrag_amd import _ffi
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    good = dict(L=100, beam_width=8, mode=_ffi.MODE_M1)
    want = ix.search_batch(g.queries[:5], 10, **good)
    ix.debug_hold(True)
    try:
        bad = [ix.search_submit(g.queries[:3], 10, L=5000, beam_width=8, mode=_ffi.MODE_M1) for _ in range(3)]   # capacity > 1024: the launch fails
        with pytest.raises(_ffi.DiskragHipError):
            bad[1].wait()
        with pytest.raises(_ffi.DiskragHipError):
            bad[0].wait()
        ok = ix.search_submit(g.queries[:5], 10, **good)      # the handle keeps working
        assert _same(ok.wait(), want)
        with pytest.raises(_ffi.DiskragHipError):
            bad[2].wait()
        # a lone failing submit (launched at once) fails in the submit itself
        ix.debug_hold(False)
        with pytest.raises(_ffi.DiskragHipError):
            ix.search_submit(g.queries[:3], 10, L=5000, beam_width=8, mode=_ffi.MODE_M1)
        assert _same(ix.search_submit(g.queries[:5], 10, **good).wait(), want)
    finally:
        ix.debug_hold(False)
    ix.batch_sync()


def test_quiesce_collects_held_submits():
    """set_pq / build / close wait for everything queued on the handle -- held submits included"""
    from diskrag_amd import _ffi
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32", mem=True)
    params = dict(L=100, beam_width=8, mode=_ffi.MODE_M1)
    want = ix.search_batch(g.queries[:7], 10, **params)
    ix.debug_hold(True)
    try:
        jobs = [ix.search_submit(g.queries[:7], 10, **params) for _ in range(4)]
        ix.set_pq(g.codebook, g.codes)                    # quiesces: the held group is launched and its tickets finished
        for j in jobs:
            assert _same((j.ids, j.dist, j.cnt, j.stats), want)
            j.wait()
    finally:
        ix.debug_hold(False)
    ix.batch_sync()


@pytest.mark.parametrize("bw", [8, 0])
def test_small_workgroup_variants_return_the_same_bits(bw):
    """variants 16 / 17 = 11 / 13 in 4-wavefront workgroups: chosen for batches below the chip's wavefront slots; forced here
    against the 16-wavefront forms, the float-row form and the oracle"""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like
    from oracle import pyoracle as orc
    x, q = sift_like(40000, 128, n_queries=300, n_clusters=64, seed=5, query_seed=6)
    ix = HipIndex.create_empty(x, R=64)
    medoid, _ = ix.build_vamana(L_build=64, alpha=1.2, passes=2, seed=3, pad_with_zero=True)
    cb = ix.pq_train(32, n_sample=20000, iters=4)
    codes = ix.pq_encode(cb, want_codes=True)
    adj = ix.get_adjacency()
    outs = {}
    for kind in (9, 11, 13, 16, 17, -1):
        ix.debug_force_kind(kind)
        outs[kind] = ix.search_batch(q, 10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
        assert ix.timing()["variant"] == (17 if kind == -1 else kind)      # 300 byte queries < 4096 slots: the small form by itself
    ix.debug_force_kind(-1)
    o = orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=100, bw=bw, codes=codes, codebook=cb)
    for kind, got in outs.items():
        assert np.array_equal(got[0], o[0]) and np.array_equal(bits(got[1]), bits(o[1].astype(np.float32))), kind
        assert np.array_equal(_stats4(got[3]), o[3]), kind
    # float queries on byte rows: 16 by itself
    qf = q + np.float32(0.5)
    got = ix.search_batch(qf, 10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
    assert ix.timing()["variant"] == 16
    o = orc.search_batch(x, adj, qf, medoid, orc.M1, 10, L=100, bw=bw, codes=codes, codebook=cb)
    assert np.array_equal(got[0], o[0]) and np.array_equal(bits(got[1]), bits(o[1].astype(np.float32)))
    ix.close()


def test_request_threads_share_launches():
    """a pool of request handlers, one query per request (app.py:84-130), each calling submit + wait: dr_search_wait does not
    hold the handle while it waits, so the other threads' submits ride in the same launches -- and every request still gets
    the bits of a call of its own"""
    import threading
    from diskrag_amd import _ffi
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    params = dict(L=100, beam_width=8, mode=_ffi.MODE_M1)
    want = ix.search_batch(g.queries, 10, **params)
    s0 = ix.pipeline_stats()
    bad, nthreads, per = [], 16, 40

    def client(t):
        for i in range(per):
            qi = (t * 7 + i) % len(g.queries)
            ids, dist, cnt, st = ix.search_submit(g.queries[qi:qi + 1], 10, **params).wait()
            if not (np.array_equal(ids[0], want[0][qi]) and np.array_equal(bits(dist[0]), bits(want[1][qi])) and cnt[0] == want[2][qi]):
                bad.append((t, i))

    th = [threading.Thread(target=client, args=(t,)) for t in range(nthreads)]
    for t in th: t.start()
    for t in th: t.join()
    assert not bad
    s1 = ix.pipeline_stats()
    assert s1["tickets"] - s0["tickets"] == nthreads * per
    assert s1["launches"] - s0["launches"] <= nthreads * per          # (how many share a launch depends on timing; the bits do not)
    ix.batch_sync()


@pytest.mark.parametrize("name,params", [
    ("sift128_R64_m32", dict(L=100, beam_width=8, mode=1)),
    ("unit1536_R16_m32", dict(L=60, beam_width=8, mode=5)),
])
def test_large_groups_full_batches_share_a_launch(name, params):
    """end of round 4: launches hold up to 32768 queries (10240 before) and 64 tickets (16): three 6000-query submits ride in ONE launch of
    18000 queries -- the per-query ADC bounds of M1 are computed once for the whole launch -- and a fourth that would overflow the
    capacity opens the next; 70 small tickets in one launch; every ticket the bits of a blocking call of its own"""
    g = load_golden(name)
    ix = get_index(name)
    rs = np.random.RandomState(5)
    nq = len(g.queries)
    big = [np.ascontiguousarray(g.queries[rs.randint(0, nq, size=n)]) for n in (6000, 6000, 6000, 15000)]
    want = [ix.search_batch(b, 10, **params) for b in big]
    ix.debug_hold(True)
    try:
        s0 = ix.pipeline_stats()
        jobs = [ix.search_submit(b, 10, **params) for b in big[:3]]
        assert ix.pipeline_stats()["launches"] == s0["launches"]                   # 18000 queries held in one group
        jobs.append(ix.search_submit(big[3], 10, **params))                        # 18000 + 15000 > 32768: launches the first group
        s1 = ix.pipeline_stats()
        assert s1["launches"] == s0["launches"] + 1 and s1["queries"] == s0["queries"] + 18000 and s1["tickets"] == s0["tickets"] + 3
        got = [j.wait() for j in jobs]
        for i, w in enumerate(want):
            assert _same(got[i], w), (name, i)
        # many tickets in one launch
        small = [np.ascontiguousarray(g.queries[rs.randint(0, nq, size=1 + (i % 5))]) for i in range(70)]
        want_s = [ix.search_batch(b, 10, **params) for b in small]
        s2 = ix.pipeline_stats()
        jobs = [ix.search_submit(b, 10, **params) for b in small]
        s3 = ix.pipeline_stats()
        assert s3["launches"] - s2["launches"] == 1 and s3["tickets"] - s2["tickets"] == 64     # the 64th ticket filled (and launched) the first group
        got = [j.wait() for j in jobs]
        assert ix.pipeline_stats()["launches"] == s2["launches"] + 2
        for i, w in enumerate(want_s):
            assert _same(got[i], w), (name, "small", i)
    finally:
        ix.debug_hold(False)
    ix.batch_sync()


def test_bulk_groups_collect_towards_the_launch_capacity():
    """Round 6: a group that already holds >= 8192 queries keeps collecting while ONE search is running, for as long as another such batch fits its
    capacity (dr_set_coalesce, now up to 65 536): a stream of 9000-query submits rides in launches of several submits, mixed with small requests, any
    wait order -- and every ticket still carries the bits of a blocking call of its own; a lone bulk submit is launched at once (nothing in flight)."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like
    x, q = sift_like(60000, 128, n_queries=9000, n_clusters=128, seed=31, query_seed=32)
    ix = HipIndex.create_empty(x, R=32)
    ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=3, pad_with_zero=True)
    ix.pq_encode(ix.pq_train(32, n_sample=20000, iters=3))
    kw = dict(L=60, beam_width=8, mode=_ffi.MODE_M1)
    try:
        want = ix.search_batch(q, 10, **kw)
        lone = ix.search_submit(q, 10, **kw)                       # nothing in flight: launched at once
        assert _same(lone.wait(), want)
        for cap in (32768, 65536):
            ix.set_coalesce(cap)
            before = ix.pipeline_stats()
            rs = np.random.RandomState(cap)
            jobs, sizes = [], []
            for i in range(22):
                n = 9000 if i % 4 else int(rs.choice([1, 7, 300]))
                a = int(rs.randint(0, len(q) - n + 1))
                sizes.append((a, n))
                jobs.append(ix.search_submit(q[a:a + n], 10, **kw))
            for i in rs.permutation(len(jobs)):
                a, n = sizes[i]
                got = jobs[i].wait()
                assert _same(got, tuple(w[a:a + n] for w in want)), (cap, i, a, n)
            after = ix.pipeline_stats()
            assert after["tickets"] - before["tickets"] == 22
            assert 1 <= after["launches"] - before["launches"] <= 22          # (how many submits share a launch depends on timing: the bits must not)
        with pytest.raises(_ffi.DiskragHipError):
            ix.set_coalesce(65537)
        ix.set_coalesce(32768)
    finally:
        ix.close()
