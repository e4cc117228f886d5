"""A4-live data (unit-norm: the rerank policy really consults the ADC) through every M1 kernel variant, against the
oracle on the same device-built graph; and the engine's regime probe (unit-scale -> live, SIFT-scale -> not)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLOCK = {0: 64, 3: 512, 9: 768}


def _index(x, R, m):
    from diskrag_amd import HipIndex
    ix = HipIndex.create_empty(x, R=R)
    medoid, _ = ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=3)
    cb = ix.pq_train(m, n_sample=10000, iters=4)
    codes = ix.pq_encode(cb, want_codes=True)
    return ix, medoid, ix.get_adjacency(), cb, codes


@pytest.mark.parametrize("D,m,kinds", [(128, 32, (0, 3, 9)), (128, 16, (0, 3, 9)), (96, 16, (0, 3)), (64, 8, (0, 3))])
def test_live_policy_every_variant_matches_oracle(D, m, kinds):
    from diskrag_amd import _ffi
    from diskrag_amd.synth import unit_mixture
    from oracle import pyoracle as orc
    x, q = unit_mixture(24000, D, n_queries=160, n_clusters=128, seed=31, latent=24)
    ix, medoid, adj, cb, codes = _index(x, 32, m)
    try:
        ix.debug_force_kind(-1)
        cases = [(100, 8, 0), (100, 0, 0), (100, 0, 1), (30, 8, 1), (64, 0, 0), (128, 8, 0)]
        want = {}
        for (L, bw, pol) in cases:
            want[(L, bw, pol)] = orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=L, bw=bw, policy=pol, codes=codes, codebook=cb,
                                                    nthreads=8)
        ran = set()
        for kind in (-1,) + tuple(kinds):
            ix.debug_force_kind(kind)
            for (L, bw, pol) in cases:
                ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
                w_ids, w_dist, w_cnt, w_st = want[(L, bw, pol)]
                assert int(st["status"].max()) == 0
                assert np.array_equal(ids, w_ids), (kind, L, bw, pol)
                valid = w_ids != 0xFFFFFFFF
                assert np.array_equal(dist[valid].view(np.uint32), w_dist[valid].astype(np.float32).view(np.uint32))
                assert np.array_equal(cnt, w_cnt)
                assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), w_st), kind
                if kind >= 0 and ix.timing()["block"] == BLOCK[kind]:
                    ran.add(kind)
        assert ix.debug_force_kind(-1) == 1                      # measured: the policy is live on this data
        assert ran == set(kinds)                                 # every pinned variant really ran
        # inline neighbour codes (the code words of a row's neighbours read as one block): same bits, every variant
        ix.inline_codes(True)
        for kind in (-1,) + tuple(kinds):
            ix.debug_force_kind(kind)
            for (L, bw, pol) in cases[:4]:
                ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
                w_ids, w_dist, w_cnt, w_st = want[(L, bw, pol)]
                assert int(st["status"].max()) == 0
                assert np.array_equal(ids, w_ids), ("inline", kind, L, bw, pol)
                valid = w_ids != 0xFFFFFFFF
                assert np.array_equal(dist[valid].view(np.uint32), w_dist[valid].astype(np.float32).view(np.uint32))
                assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), w_st), kind
        ix.inline_codes(False)
        ix.debug_force_kind(-1)
        # the policy is live: a good share of the visited neighbours was NOT exact-scored, and the ADC was evaluated
        ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=0, mode=_ffi.MODE_M1, band_policy=1)
        assert st["exact"].sum() < 0.8 * st["visited"].sum() and st["pq_evaluated"].sum() > 0.5 * st["pq"].sum()
    finally:
        ix.debug_force_kind(-1)
        ix.close()


def test_regime_probe_on_sift_scale_data():
    from diskrag_amd import _ffi
    from diskrag_amd.synth import sift_like
    x, q = sift_like(24000, 128, n_queries=96, n_clusters=64, seed=5, query_seed=6)
    ix, medoid, adj, cb, codes = _index(x, 32, 32)
    try:
        assert ix.debug_force_kind(-1) == -1                     # not probed yet
        ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
        assert ix.debug_force_kind(-1) == 0                      # Q1: the policy is provably true, ADC skipped
        assert st["pq_evaluated"].sum() < 0.1 * st["pq"].sum()
        assert ix.timing()["variant"] == 17 and ix.timing()["block"] == 256    # byte rows landed in LDS; 96 queries < 4096 slots: the 4-wavefront workgroups
        ix.set_pq(cb, codes)                                     # a PQ change resets the probe
        assert ix.debug_force_kind(-1) == -1
    finally:
        ix.close()


def test_byte_rows_are_lossless_and_only_for_integer_data():
    """Variants 10 / 11 / 13 read a byte copy of the vectors, built only when every component is an integer in [0, 255]
    (13 / 14: the queries of the batch as well): same bits as the float-row variant 9 and as the oracle; data that does
    not qualify never takes them."""
    from diskrag_amd import _ffi
    from diskrag_amd.synth import sift_like
    from oracle import pyoracle as orc
    x, q = sift_like(24000, 128, n_queries=128, n_clusters=64, seed=91, query_seed=92)
    assert np.array_equal(x, np.rint(x)) and x.min() >= 0 and x.max() <= 255
    ix, medoid, adj, cb, codes = _index(x, 32, 32)
    try:
        want = orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=100, bw=8, codes=codes, codebook=cb, nthreads=8)
        blocks = {}
        for kind in (9, 11, 13, 16, 17, -1):
            ix.debug_force_kind(kind)
            for (L, bw) in ((20, 8), (100, 8), (100, 0), (300, 16)):      # (the engine's own choice at L = 20 and 128 queries is variant 18: not the last one asked)
                ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1)
                w = want if (L, bw) == (100, 8) else orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=L, bw=bw, codes=codes,
                                                                      codebook=cb, nthreads=8)
                assert np.array_equal(ids, w[0]) and np.array_equal(dist.view(np.uint32), w[1].astype(np.float32).view(np.uint32))
                assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), w[3])
            blocks[kind] = (ix.timing()["block"], ix.timing()["lds_bytes"])
            # integer data AND integer queries: 13 -- as 17 (4-wavefront workgroups) for a batch below the chip's 4096 wavefront slots
            assert ix.timing()["variant"] == (kind if kind >= 0 else 17)
        assert blocks[9][0] == 768 and blocks[11][0] == 1024 and blocks[11][1] < blocks[9][1] + 4 * 8192
        assert blocks[13][0] == blocks[11][0] and blocks[13][1] == blocks[11][1] + 16 * 528   # + adjacency landing areas
        assert blocks[-1] == blocks[17] and blocks[17] == (256, blocks[13][1] // 4) and blocks[16] == (256, blocks[11][1] // 4)
        # byte queries (13 / 14: v_dot4_u32_u8 distances) are taken only when EVERY component of the batch is an integer
        # in [0, 255]; one fractional, negative or too-large component and the batch runs on the float-query variants.
        # Extreme integer queries (all 0 / all 255 against rows up to 218) stay exact as well.
        for name, qq in (("frac", q + np.float32(0.5) * (np.arange(q.size).reshape(q.shape) == 777)),
                         ("neg", np.where(np.arange(q.size).reshape(q.shape) == 5, np.float32(-1), q)),
                         ("big", np.where(np.arange(q.size).reshape(q.shape) == q.size - 1, np.float32(256), q)),
                         ("extreme", np.concatenate([np.zeros((2, 128), np.float32), np.full((2, 128), 255, np.float32), q[:28]]))):
            qq = np.ascontiguousarray(qq, dtype=np.float32)
            w = orc.search_batch(x, adj, qq, medoid, orc.M1, 10, L=100, bw=8, codes=codes, codebook=cb, nthreads=8)
            for kind in (13, -1):
                ix.debug_force_kind(kind)
                ids, dist, cnt, st = ix.search_batch(qq, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
                # (16 / 17: the 4-wavefront workgroups of 11 / 13 -- these batches are far below the chip's 4096 wavefront slots; a forced
                # variant that applies is kept as forced)
                assert ix.timing()["variant"] == ((13 if kind == 13 else 17) if name == "extreme" else 16), name
                assert np.array_equal(ids, w[0]) and np.array_equal(dist.view(np.uint32), w[1].astype(np.float32).view(np.uint32)), name
                assert np.array_equal(np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1), w[3]), name
        # the exact traversals: float rows (8) == byte rows (12) == byte rows and queries (14) == oracle
        for (mode, omode, k, L, bw, fl, ofl) in ((2, orc.M2, 8, 0, 8, 0, orc.F_PAIRWISE), (4, orc.M4, 10, 50, 0, _ffi.F_SQDIST, orc.F_CYTHON | orc.F_PAIRWISE)):
            w = orc.search_batch(x, adj, q, medoid, omode, k, L=L, bw=bw, flags=ofl, nthreads=8)
            for kind, blk in ((8, 512), (12, 1024), (14, 1024), (-1, 1024)):
                ix.debug_force_kind(kind)
                ids, dist, cnt, st = ix.search_batch(q, k, L=L, beam_width=bw, mode=mode, flags=fl)
                assert ix.timing()["block"] == blk and ix.timing()["variant"] == (kind if kind >= 0 else 14)
                assert np.array_equal(ids, w[0]) and np.array_equal(dist.view(np.uint32), w[1].astype(np.float32).view(np.uint32))
    finally:
        ix.debug_force_kind(-1)
        ix.close()
    # the same descriptors shifted by a quarter: not integers any more -> float rows, still the oracle's bits
    y = x + np.float32(0.25)
    iy, medoid, adj, cb, codes = _index(y, 32, 32)
    try:
        for kind in (11, 13, -1):
            iy.debug_force_kind(kind)
            ids, dist, cnt, st = iy.search_batch(q, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
            assert iy.timing()["block"] == 768 and iy.timing()["lds_bytes"] > 140000 and iy.timing()["variant"] == 9
            w = orc.search_batch(y, adj, q, medoid, orc.M1, 10, L=100, bw=8, codes=codes, codebook=cb, nthreads=8)
            assert np.array_equal(ids, w[0]) and np.array_equal(dist.view(np.uint32), w[1].astype(np.float32).view(np.uint32))
    finally:
        iy.debug_force_kind(-1)
        iy.close()
