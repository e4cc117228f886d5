"""The disk tier of the full-precision rows (dr_index_attach_row_file -- the reference's MMapNodeReader, io/diskann_persist.py:201-234): a PQ-only
index whose rows stay in index.dat answers PQ traversal + exact rerank with the same ids and distance bits as the index that holds them in HBM."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,buffered", [(128, False), (1536, False), (96, True)])
def test_rerank_from_the_index_file_equals_rerank_from_hbm(tmp_path, D, buffered):
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import unit_mixture
    N, R, m = 6000, 32, 32 if D != 96 else 16
    x, q = unit_mixture(N, D, n_queries=300, n_clusters=48, seed=21, latent=24)
    full = HipIndex.create_empty(x, R=R)
    medoid, _ = full.build_vamana(L_build=60, alpha=1.2, passes=2, seed=3)
    cb = full.pq_train(m, n_sample=6000, iters=4)
    codes = full.pq_encode(cb, want_codes=True)
    adj = full.get_adjacency()
    # index.dat as the reference writes it: record i = D float32 then R uint32 (diskann_persist.py:17-24)
    path = tmp_path / "index.dat"
    rec = np.empty((N, D + R), dtype=np.uint32)
    rec[:, :D] = x.view(np.uint32)
    rec[:, D:] = adj
    rec.tofile(path)
    shard = HipIndex.create_codes(adj, medoid, D, cb, codes)
    try:
        # without the file the rerank is refused
        with pytest.raises(_ffi.DiskragHipError):
            shard.search_batch(q[:4], 10, L=50, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK)
        if buffered: os.environ["DR_ROW_FILE_BUFFERED"] = "1"
        try:
            shard.attach_row_file(path)
        finally:
            os.environ.pop("DR_ROW_FILE_BUFFERED", None)
        for mode in (_ffi.MODE_PQB, _ffi.MODE_PQ):
            for (k, L, bw, nq) in ((10, 100, 8, 300), (5, 20, 8, 1), (50, 120, 0, 33)):
                want = full.search_batch(q[:nq], k, L=L, beam_width=bw, mode=mode, flags=_ffi.F_RERANK)
                got = shard.search_batch(q[:nq], k, L=L, beam_width=bw, mode=mode, flags=_ffi.F_RERANK)
                assert np.array_equal(got[0], want[0]), (mode, k, L, bw, nq)
                assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
                assert np.array_equal(got[2], want[2])
                assert np.array_equal(got[3]["exact"], want[3]["exact"]) and int(got[3]["status"].max()) == 0
        # the pipelined path too
        want = full.search_batch(q[:64], 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK)
        got = shard.search_submit(q[:64], 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK).wait()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
        # a file that is too short is refused
        short = tmp_path / "short.dat"
        rec[: N // 2].tofile(short)
        with pytest.raises(_ffi.DiskragHipError):
            shard.attach_row_file(short)
    finally:
        shard.close(); full.close()
