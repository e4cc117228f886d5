"""Diagnostic (GPU box): where the workgroup-per-query kernel (variant 18, csrc/latency_kernel.hpp) spends a lone query's time -- wavefront 0's
shader-clock sums. Needs a -DDR_PHASE_TIMING build (DR_LIB=diskrag_amd/libdiskrag_hip_ph.so). usage: exp_phase_latency.py [points]"""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
os.environ["DR_LAT_ALL"] = "1"      # (variant 18 at every list size, not only where the engine prefers it)
if len(sys.argv) > 2 and sys.argv[2] == "emb":      # the embedding shape: unit-norm mixture, D = 1536, the rerank policy live
    from diskrag_amd.synth import unit_mixture
    x, q = unit_mixture(N, 1536, n_queries=512, n_clusters=256, seed=5, latent=32)
else:
    x, q = sift_like(N, 128, n_queries=512, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=20000, iters=3); ix.pq_encode(cb)
ix.search_batch(q, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
ix.search_batch(q, 5, L=20, beam_width=8, mode=_ffi.MODE_M1)
names = ["setup", "decisions", "scheduling", "rounds", "(own scoring)", "#rounds", "#pops", "output"]
for (k, L) in ((10, 100), (5, 20)):
    for nq in (1, 64):
        acc, steps, kms = np.zeros(8), 0.0, []
        for i in range(0, 256, nq):
            ids, dist, cnt, st = ix.search_batch(q[i:i + nq], k, L=L, beam_width=8, mode=_ffi.MODE_M1)
            assert ix.timing()["variant"] == 18
            acc += np.array(ix.debug_phase_cycles()); steps += st["steps"].sum(); kms.append(ix.timing()["search_kernel_ms"])
            if os.environ.get("LAT_SUB2"):
                pe = st["pq_evaluated"].astype(np.uint64)
                hist = hist + np.array([(pe & 255).sum(), ((pe >> 8) & 255).sum(), ((pe >> 16) & 255).sum(), ((pe >> 24) & 255).sum()]) if i else np.array([(pe & 255).sum(), ((pe >> 8) & 255).sum(), ((pe >> 16) & 255).sum(), ((pe >> 24) & 255).sum()])
        tot = acc[[0, 1, 2, 3, 7]].sum()
        print(f"# M1 k={k} L={L} bw=8, {nq} per call: kernel_ms {np.mean(kms):.3f} expansions/query {steps / 256:.1f} rounds/query {acc[5] / 256:.1f} "
              f"pops consumed/query {acc[6] / 256:.1f} cycles/query {tot / 256:.0f}")
        for i in (0, 1, 2, 3, 4, 7):
            per = acc[i] / (acc[6] if i == 1 else acc[5] if i in (2, 3, 4) else 256)
            print(f"{names[i]:14s} {acc[i] / tot * 100:6.2f}%  cycles per {'pop' if i == 1 else 'round' if i in (2, 3, 4) else 'query'} {per:8.0f}")
        # (a -DDR_LAT_SUBPHASES build adds wavefront 0's scoring sub-phases into the same slots: 0 += adjacency row wait, 2 += visited probe +
        #  compaction, 7 += exact rows, 6 += ADC -- read with LAT_SUB=1: cycles per round; the lines above are then polluted)
        if os.environ.get("LAT_SUB"):
            print("# sub-phases of wavefront 0's scoring, cycles per round: adjacency %.0f  probe+compaction %.0f  rows %.0f  ADC %.0f" %
                  (acc[0] / acc[5], acc[2] / acc[5], acc[7] / acc[5], acc[6] / acc[5]))
        if os.environ.get("LAT_SUB2"):      # a -DDR_PHASE_TIMING -DDR_LAT_SUB2 build (slots shared with the phases above: read the differences)
            print("# rows consumed per query with 0 / 1-2 / 3+ candidates: %.1f %.1f %.1f; of those through the general decisions: %.1f" % tuple(hist / 256.0))
            print("# decisions of wavefront 0, cycles per pop (raw slot sums / pops): [0] setup + peek/stop/lookup/commit %.0f  [2] scheduling + slot read + visited %.0f  "
                  "[6] policy set-up %.0f  [7] output + accept + merge %.0f  [4] own scoring + trim %.0f; [3] rounds + accept-all count %.0f; [1] %.0f" %
                  (acc[0] / steps, acc[2] / steps, acc[6] / steps, acc[7] / steps, acc[4] / steps, acc[3] / steps, acc[1] / steps))
