"""Exact traversals (M2 beam_width 8, M4 L = 100) on the bench index: float rows (8), byte rows (12), byte rows + byte
queries (14). Kernel ms per 10k-query batch."""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=10000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
ix.batch_upload(q)
for name, kw in (("M2_bw8", dict(L=0, beam_width=8, mode=2)), ("M4_L100", dict(L=100, beam_width=0, mode=4, flags=_ffi.F_SQDIST))):
    for kind in (8, 12, 14):
        ix.debug_force_kind(kind)
        for _ in range(3):
            ix.batch_run(10, **kw)
        ix.batch_sync()
        for _ in range(10):
            ix.batch_run(10, **kw)
        ix.batch_sync()
        t = ix.timing()
        print(name, "variant", t["variant"], "kernel_ms %.3f" % t["search_kernel_ms"])
ix.debug_force_kind(-1)
