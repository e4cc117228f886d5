"""Diagnostic (GPU box, -DDR_TRACE_VIS build via DR_LIB): dumps the visited-bitmap positions tested by the first 256
bench queries, expansion by expansion, to gpurun_out/vis_trace_bw<bw>.npy (analysed offline: sizing of an LDS-resident
visited set)."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=1024, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=20000, iters=3); ix.pq_encode(cb)
for bw in (8, 0):
    ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
    ids, dist, cnt, st = ix.search_batch(q, 10, L=100, beam_width=bw, mode=_ffi.MODE_M1)
    buf = np.zeros((256, 16384), dtype=np.uint32)
    rc = _ffi.load_library().dr_debug_phase_cycles(ix._h, buf.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    np.save(f"gpurun_out/vis_trace_bw{bw}.npy", buf)
    print(bw, "trace words", buf[:, 0].mean(), "visited", st["visited"][:256].mean(), "steps", st["steps"][:256].mean())
