"""Round 4: the reference's legal PQ shapes OUTSIDE the shapes the other GPU tests use.
`pydiskann/pq/adaptive_pq.py:29,80-91` admits n_subvectors in {4, 8, 16, 32, 48, 64, 96, 128} with sub_dim in [2, 64]; a
"high_accuracy" index of <= 50k points carries m = 96 or 128 (`:98-101`). Three such shapes have goldens from the reference
itself (tests/golden: unit768_R16_m96, unit256_R16_m128, unit256_R16_m4 -- every golden test of tests/test_gpu_parity.py
runs on them); here every shape of the sweep is searched by the device and the oracle on a device-built index: M1 under both
band policies, the reference's PQ-only traversal (M3 with PQ), the engine's (DR_MODE_PQ) with and without the exact rerank,
and the kernel-level seams (A2 table, A3 sums). A 96- or 128-KiB table is one wavefront per CU: correct, not fast."""
import numpy as np
import pytest

from tests.test_gpu_parity import bits

pytestmark = pytest.mark.gpu

SHAPES = [(1536, 48), (1536, 96), (1536, 128), (768, 96), (960, 96), (256, 128), (128, 4), (256, 4), (128, 8), (960, 64), (768, 48)]


def _stats4(st):
    return np.stack([st["steps"], st["visited"], st["exact"], st["pq"]], axis=1)


@pytest.mark.parametrize("D,m", SHAPES)
def test_every_legal_pq_shape_matches_the_oracle(D, m):
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import unit_mixture
    from oracle import pyoracle as orc
    n = 1500
    x, q = unit_mixture(n, D, n_queries=20, n_clusters=12, seed=40 + m, latent=16)
    ix = HipIndex.create_empty(x, R=24)
    try:
        medoid, _ = ix.build_vamana(L_build=40, alpha=1.2, passes=2, seed=2, pad_with_zero=True)
        cb = ix.pq_train(m, n_sample=n, iters=3)
        codes = ix.pq_encode(cb, want_codes=True)
        adj = ix.get_adjacency()
        # kernel-level seams
        lut = ix.distance_table(q[:3])
        assert np.array_equal(bits(lut), bits(np.stack([orc.build_lut(cb, qq) for qq in q[:3]])))
        nodes = np.arange(0, n, 97, dtype=np.uint32)
        sq, rt = ix.adc(q[:3], nodes)
        assert np.array_equal(bits(sq), bits(np.stack([orc.adc(orc.build_lut(cb, qq), codes[nodes])[0] for qq in q[:3]])))
        # M1 (rerank policy A4 live on unit-scale data), both band policies, trimmed and not
        for (L, bw, pol) in ((60, 8, 0), (60, 0, 1), (20, 4, 1)):
            w = orc.search_batch(x, adj, q, medoid, orc.M1, 10, L=L, bw=bw, policy=pol, codes=codes, codebook=cb)
            ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
            assert int(st["status"].max()) == 0
            assert np.array_equal(ids, w[0]) and np.array_equal(cnt, w[2]), (D, m, L, bw, pol)
            valid = w[0] != 0xFFFFFFFF
            assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid])
            assert np.array_equal(_stats4(st), w[3])
        # the reference's PQ-only traversal and the engine's, then the rerank
        w = orc.search_batch(x, adj, q, medoid, orc.M3, 5, L=5, bw=8, flags=orc.F_USE_PQ, codes=codes, codebook=cb)
        ids, dist, cnt, st = ix.search_batch(q, 5, L=5, beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
        assert np.array_equal(ids, w[0]) and np.array_equal(cnt, w[2]) and np.array_equal(_stats4(st), w[3])
        for flags, oflags in ((0, 0), (_ffi.F_RERANK, orc.F_RERANK)):
            w = orc.search_batch(x, adj, q, medoid, orc.PQ, 10, L=50, bw=8, flags=oflags, codes=codes, codebook=cb)
            ids, dist, cnt, st = ix.search_batch(q, 10, L=50, beam_width=8, mode=_ffi.MODE_PQ, flags=flags)
            assert int(st["status"].max()) == 0
            assert np.array_equal(ids, w[0]) and np.array_equal(cnt, w[2]) and np.array_equal(_stats4(st), w[3]), (D, m, flags)
            valid = w[0] != 0xFFFFFFFF
            assert np.array_equal(bits(dist)[valid], bits(w[1].astype(np.float32))[valid])
        # a PQ-only shard of the same shape (no stored vectors)
        sh = HipIndex.create_codes(adj, medoid, D, cb, codes)
        try:
            w = orc.search_batch(x, adj, q, medoid, orc.PQ, 10, L=50, bw=8, codes=codes, codebook=cb)
            ids, dist, cnt, st = sh.search_batch(q, 10, L=50, beam_width=8, mode=_ffi.MODE_PQ)
            assert np.array_equal(ids, w[0]) and np.array_equal(_stats4(st), w[3])
        finally:
            sh.close()
    finally:
        ix.close()
