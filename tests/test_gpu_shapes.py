"""Shapes the goldens do not reach: adjacency rows wider than one wavefront (R = 96, 128: two 64-slot chunks per
expansion), extreme k / L / beam_width, a one-query batch, and a batch larger than the engine's chunk size. All
against the oracle on a device-built graph. Needs a GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PAD = np.uint32(0xFFFFFFFF)


def _check(ix, x, adj, medoid, cb, codes, q, mode, k, L, bw, flags=0, pol=0):
    from oracle import pyoracle as orc
    omode = {1: orc.M1, 2: orc.M2, 3: orc.M3, 4: orc.M4}[mode]
    oflags = (orc.F_USE_PQ if (mode == 3 and flags & 1) else 0)
    ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=mode, band_policy=pol, flags=flags)
    oi, od, oc, ost = orc.search_batch(x, adj, q, medoid, omode, k, L=L, bw=bw, policy=pol, flags=oflags, codes=codes,
                                       codebook=cb, nthreads=8)
    assert (st["status"] == 0).all()
    assert np.array_equal(cnt, oc), (mode, k, L, bw)
    assert np.array_equal(ids, oi), (mode, k, L, bw)
    valid = oi != PAD
    if mode in (1, 3):
        assert np.array_equal(dist[valid].view(np.uint32), od[valid].astype(np.float32).view(np.uint32))
    else:
        assert np.allclose(dist[valid], od[valid], rtol=1e-4)
    assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost)


@pytest.mark.parametrize("R", [96, 128])
def test_rows_wider_than_a_wavefront(R):
    from diskrag_amd import HipIndex
    from diskrag_amd.synth import sift_like
    x, q = sift_like(30000, 128, n_queries=96, n_clusters=64, seed=51, query_seed=52)
    ix = HipIndex.create_empty(x, R=R)
    medoid, _ = ix.build_vamana(L_build=120, alpha=1.2, passes=2, seed=2)
    cb = ix.pq_train(32, n_sample=10000, iters=3)
    codes = ix.pq_encode(cb, want_codes=True)
    adj = ix.get_adjacency()
    assert (adj != 0).sum(axis=1).max() > 64            # some rows really use the second chunk
    try:
        for (mode, k, L, bw) in ((1, 10, 100, 8), (1, 10, 100, 0), (1, 20, 40, 16), (2, 10, 0, 32), (2, 8, 0, 8)):
            _check(ix, x, adj, medoid, cb, codes, q, mode, k, L, bw)
    finally:
        ix.close()


def test_extreme_parameters_and_batch_sizes():
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like
    x, q = sift_like(20000, 64, n_queries=200, n_clusters=64, seed=61, query_seed=62)
    ix = HipIndex.create_empty(x, R=32)
    medoid, _ = ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=2)
    cb = ix.pq_train(16, n_sample=8000, iters=3)
    codes = ix.pq_encode(cb, want_codes=True)
    adj = ix.get_adjacency()
    try:
        for (mode, k, L, bw) in ((1, 1, 1, 0), (1, 1, 1, 1), (1, 64, 20, 8),      # k > L: at most L hits
                                 (1, 10, 512, 0), (1, 10, 300, 200), (1, 5, 7, 3),
                                 (2, 10, 0, 1), (2, 64, 0, 100), (2, 5, 0, 512)):
            _check(ix, x, adj, medoid, cb, codes, q[:48], mode, k, L, bw)
        # one query, and a batch beyond the engine's 32768-query chunk (rows repeat: every copy must agree)
        _check(ix, x, adj, medoid, cb, codes, q[:1], 1, 10, 50, 8)
        big = np.tile(q, (170, 1))[:33000]
        ids, dist, cnt, st = ix.search_batch(big, 10, L=30, beam_width=8, mode=_ffi.MODE_M1)
        ref = ix.search_batch(q, 10, L=30, beam_width=8, mode=_ffi.MODE_M1)
        for r0 in (0, 200, 32600, 32800):
            n = min(200, 33000 - r0)
            sl = slice(r0, r0 + n)
            off = r0 % 200
            assert np.array_equal(ids[sl], np.roll(ref[0], -off, axis=0)[:n])
    finally:
        ix.close()


@pytest.mark.parametrize("D,m", [(32, 8), (64, 16), (256, 32), (768, 32), (960, 48)])
def test_every_built_dimension(D, m):
    """The pairwise tree and the chain-major layout are compiled per dimension: each built D (the goldens cover 96,
    128 and 1536) against the oracle, SIFT-scale and unit-scale data, M1 / M2 / M3-PQ / M4."""
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import sift_like, unit_mixture
    for gen in ("sift", "unit"):
        if gen == "sift":
            x, q = sift_like(6000, D, n_queries=40, n_clusters=32, seed=70 + D, query_seed=71 + D, latent=min(32, D))
        else:
            x, q = unit_mixture(6000, D, n_queries=40, n_clusters=32, seed=72 + D, latent=min(32, D))
        ix = HipIndex.create_empty(x, R=32)
        medoid, _ = ix.build_vamana(L_build=50, alpha=1.2, passes=2, seed=2, pad_with_zero=(gen == "sift"))
        cb = ix.pq_train(m, n_sample=6000, iters=3)
        codes = ix.pq_encode(cb, want_codes=True)
        adj = ix.get_adjacency()
        try:
            if gen == "sift":       # 0-padded rows: the engine paths (M1, M2)
                for (mode, k, L, bw, pol) in ((1, 10, 60, 8, 0), (1, 10, 60, 0, 1), (2, 8, 0, 8, 0)):
                    _check(ix, x, adj, medoid, cb, codes, q, mode, k, L, bw, pol=pol)
            else:                   # PAD-padded in-memory rows: M1 with the policy live, M3 with PQ, M4
                for (mode, k, L, bw, pol, fl) in ((1, 10, 60, 8, 0, 0), (1, 10, 60, 0, 1, 0), (3, 5, 5, 8, 0, _ffi.F_USE_PQ), (4, 10, 40, 0, 0, 0)):
                    _check(ix, x, adj, medoid, cb, codes, q, mode, k, L, bw, flags=fl, pol=pol)
        finally:
            ix.close()


# ---- dimensions without compiled kernels (round 6): the generic traversal, numpy's pairwise tree evaluated from D at run time ------------------
def _toy_index(N, D, m, R, seed, pad):
    """a small kNN + random graph, a sampled codebook and its code words, all in numpy (the builder needs a compiled dimension; any graph serves a
    parity test -- the oracle walks the same rows)"""
    rs = np.random.RandomState(seed)
    x = rs.randn(N, D).astype(np.float32)
    x[: N // 2] += 2.0 * rs.randn(1, D).astype(np.float32)
    q = (x[rs.choice(N, 24, replace=False)] + 0.1 * rs.randn(24, D)).astype(np.float32)
    n2 = (x.astype(np.float64) ** 2).sum(1)
    d2 = n2[:, None] + n2[None, :] - 2.0 * x.astype(np.float64) @ x.astype(np.float64).T
    np.fill_diagonal(d2, np.inf)
    knn = np.argsort(d2, axis=1)[:, : R - 5].astype(np.uint32)
    adj = np.full((N, R), pad, dtype=np.uint32)
    adj[:, : R - 5] = knn
    adj[:, R - 5: R - 2] = rs.randint(0, N, size=(N, 3)).astype(np.uint32)          # (repeats of a kNN entry happen: first-occurrence masks at work)
    medoid = int(np.argmin(((x - x.mean(0)) ** 2).sum(1)))
    sd = D // m
    cb = np.stack([x[rs.choice(N, 256, replace=False), j * sd:(j + 1) * sd] for j in range(m)]).astype(np.float32)       # [m][256][sd]
    codes = np.empty((N, m), dtype=np.uint8)
    for j in range(m):
        sub = x[:, j * sd:(j + 1) * sd].astype(np.float64)
        dd = (sub ** 2).sum(1)[:, None] + (cb[j].astype(np.float64) ** 2).sum(1)[None, :] - 2.0 * sub @ cb[j].astype(np.float64).T
        codes[:, j] = np.argmin(dd, axis=1)
    return x, q, adj, medoid, cb, codes


GENERIC_CASES = ((1, 10, 40, 8, 0, 0), (1, 10, 40, 0, 1, 0), (1, 5, 12, 4, 0, 0), (2, 8, 0, 8, 0, 0), (2, 10, 0, 32, 0, 0),
                 (3, 5, 5, 8, 0, 1), (3, 10, 10, 16, 0, 1), (3, 5, 5, 8, 0, 0), (4, 10, 30, 0, 0, 0), (4, 10, 30, 0, 0, 2))


def _check_generic(ix, x, adj, medoid, cb, codes, q):
    from oracle import pyoracle as orc
    for (mode, k, L, bw, pol, fl) in GENERIC_CASES:
        omode = {1: orc.M1, 2: orc.M2, 3: orc.M3, 4: orc.M4}[mode]
        # (the device sums M3-without-PQ and the Cython twin of M4 in numpy's pairwise order: the reference's -ffast-math order is unpinned)
        oflags = (orc.F_USE_PQ if (mode == 3 and fl & 1) else 0) | (orc.F_PAIRWISE if mode in (3, 4) else 0) | (orc.F_CYTHON if (mode == 4 and fl & 2) else 0)
        ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=mode, band_policy=pol, flags=fl)
        oi, od, oc, ost = orc.search_batch(x, adj, q, medoid, omode, k, L=L, bw=bw, policy=pol, flags=oflags, codes=codes, codebook=cb, nthreads=8)
        tag = (x.shape[1], mode, k, L, bw, pol, fl)
        assert (st["status"] == 0).all(), tag
        assert np.array_equal(cnt, oc) and np.array_equal(ids, oi), tag
        valid = oi != PAD
        assert np.array_equal(dist[valid].view(np.uint32), od[valid].astype(np.float32).view(np.uint32)), tag       # every mode bit for bit: one summation order
        assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), ost), tag


@pytest.mark.parametrize("D,m", [(48, 8), (100, 10), (384, 32), (512, 8), (1024, 4), (20, 4), (7, 1), (1000, 8)])
def test_dimensions_without_compiled_kernels(D, m):
    """pydiskann's searches take any D (vamana_graph.py:719-760, 535-640); the engine compiles eight. Any other dimension gets an index in original
    element order and the generic traversal (search_f64.hpp, D = 0): M1 / M2 / M3 (PQ and exact) / M4 (norm and squared) against the oracle, ids,
    distance bits, counts and counters -- D not a multiple of 8 (numpy's leftover elements), sub-vectors longer than 128 (the tree's recursion inside
    a table row), D below 8 (the sequential form); float64 queries for M1 / M2 as the CLI hands them; what needs a compiled dimension says so."""
    from diskrag_amd import HipIndex, _ffi
    from oracle import pyoracle as orc
    for pad in (0, int(PAD)):
        x, q, adj, medoid, cb, codes = _toy_index(1200, D, m, 16, 100 + D, pad)
        ix = HipIndex.create(x, adj, medoid)
        ix.set_pq(cb, codes)
        try:
            _check_generic(ix, x, adj, medoid, cb, codes, q)
            if pad == 0:
                q64 = q.astype(np.float64) + 1e-9
                for (mode, k, L, bw) in ((1, 10, 40, 8), (2, 8, 0, 8)):
                    ids, dist, cnt, st = ix.search_batch_f64(q64, k, L=L, beam_width=bw, mode=mode)
                    oi, od, oc, ost = orc.search_batch(x, adj, q64, medoid, {1: orc.M1, 2: orc.M2}[mode], k, L=L, bw=bw, codes=codes, codebook=cb, nthreads=8)
                    assert np.array_equal(ids, oi) and np.array_equal(cnt, oc), (D, mode)
                    assert np.array_equal(dist[oi != PAD].view(np.uint64), od[oi != PAD].view(np.uint64)), (D, mode)
                for call in (lambda: ix.build_vamana(L_build=20), lambda: ix.bruteforce_topk(q, 5), lambda: ix.exact_distances(q, np.arange(4, dtype=np.uint32)),
                             lambda: ix.search_batch(q, 5, L=20, mode=_ffi.MODE_PQB), lambda: ix.search_submit(q, 5, L=20).wait()):
                    with pytest.raises(_ffi.DiskragHipError) as e:
                        call()
                    assert e.value.code == _ffi.E_UNSUPPORTED
        finally:
            ix.close()


@pytest.mark.parametrize("D,m", [(32, 8), (64, 16), (96, 16), (128, 32), (256, 32), (768, 32), (960, 48), (1536, 32)])
def test_the_run_time_tree_equals_the_compiled_trees(D, m, monkeypatch):
    """On the eight built dimensions the generic traversal (DR_FORCE_GENERIC=1: rows read through the chain-major position table) and the compiled
    kernels return the same ids, distance bits and counters -- and both equal the oracle."""
    from diskrag_amd import HipIndex
    x, q, adj, medoid, cb, codes = _toy_index(1200, D, m, 16, 300 + D, 0)
    ix = HipIndex.create(x, adj, medoid)
    ix.set_pq(cb, codes)
    try:
        monkeypatch.setenv("DR_FORCE_GENERIC", "1")
        _check_generic(ix, x, adj, medoid, cb, codes, q)
        a = {c: ix.search_batch(q, c[1], L=c[2], beam_width=c[3], mode=c[0], band_policy=c[4], flags=c[5]) for c in GENERIC_CASES}
        monkeypatch.delenv("DR_FORCE_GENERIC")
        for c, (ids, dist, cnt, st) in a.items():
            i2, d2, c2, s2 = ix.search_batch(q, c[1], L=c[2], beam_width=c[3], mode=c[0], band_policy=c[4], flags=c[5])
            assert np.array_equal(ids, i2) and np.array_equal(dist.view(np.uint32), d2.view(np.uint32)) and np.array_equal(cnt, c2), (D, c)
            assert all(np.array_equal(st[f], s2[f]) for f in ("steps", "visited", "exact", "pq")), (D, c)
    finally:
        ix.close()


def test_a_very_long_vector_through_the_generic_traversal():
    """D = 17 000 (nine levels of numpy's pairwise recursion; the queries' two copies take 136 KB of the kernel's LDS): the exact traversals against the oracle"""
    from diskrag_amd import HipIndex
    from oracle import pyoracle as orc
    rs = np.random.RandomState(5)
    N, D, R = 300, 17000, 8
    x = rs.randn(N, D).astype(np.float32)
    q = (x[:6] + 0.05 * rs.randn(6, D)).astype(np.float32)
    adj = np.stack([rs.permutation(N)[:R] for _ in range(N)]).astype(np.uint32)
    ix = HipIndex.create(x, adj, 0)
    try:
        for (mode, omode, k, L, bw, oflags) in ((2, orc.M2, 5, 0, 8, 0), (4, orc.M4, 5, 20, 0, orc.F_PAIRWISE)):
            ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=mode)
            oi, od, oc, ost = orc.search_batch(x, adj, q, 0, omode, k, L=L, bw=bw, flags=oflags, nthreads=4)
            assert np.array_equal(ids, oi) and np.array_equal(cnt, oc), mode
            assert np.array_equal(dist[oi != PAD].view(np.uint32), od[oi != PAD].astype(np.float32).view(np.uint32)), mode
    finally:
        ix.close()
