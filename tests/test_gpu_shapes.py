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
