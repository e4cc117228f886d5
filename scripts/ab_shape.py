"""Compact A/B target: one c3- / c4- / c5-shaped index built on the device, a fixed list of runs, one line each.
usage: DR_LIB=... python scripts/ab_shape.py c3|c4|c5s N [kinds...]   (kinds: DR_FORCE_KIND values tried for M1, -1 = engine's choice)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                      # noqa: E402
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel, recall_at_k     # noqa: E402

shape, n = sys.argv[1], int(sys.argv[2])
kinds = [int(a) for a in sys.argv[3:]] or [-1]
nq = 10000
D, m, ncl, latent, R = {"c3": (1536, 32, 4096, 64, 64), "c4": (96, 16, 4096, 32, 64), "c5s": (1536, 32, 4096, 64, 32)}[shape]
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=nq, n_clusters=ncl, seed=11, latent=latent)
ix = HipIndex.create_empty(x, R=R)
med, bsec = ix.build_vamana(L_build=100 if shape != "c5s" else 64, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(m, n_sample=100000, iters=5)
ix.pq_encode(cb)
gt, _ = ix.bruteforce_topk(q, 10)
import os
if os.environ.get("DR_INLINE") == "1":
    ix.inline_codes(True)
print(f"# {shape} N={n} D={D} m={m} R={R} build {bsec:.1f}s inline_codes={os.environ.get('DR_INLINE', '0')}", flush=True)


def run(tag, kind=-1, **kw):
    ix.debug_force_kind(kind)
    ix.batch_upload(q)
    ix.batch_run(10, **kw); ix.batch_sync()
    ms = []
    for _ in range(3):
        ix.batch_run(10, **kw); ix.batch_sync()
        ms.append(ix.timing()["search_kernel_ms"])
    ids, dist, cnt, st = ix.batch_download()
    t = ix.timing()
    alg = float((4.0 * D + st["steps"] * 4.0 * R + st["pq_evaluated"] * float(m) + st["exact"] * 4.0 * D + 80).sum())
    chk = int(np.bitwise_xor.reduce(ids.astype(np.uint64).ravel() * np.uint64(0x9E3779B97F4A7C15) + dist.view(np.uint32).astype(np.uint64).ravel()))
    print(f"{tag:34s} kind {t['variant']:2d} w/CU {t['waves_per_cu']:2d} kernel_ms {min(ms):8.3f} (med {sorted(ms)[1]:8.3f}) lut_ms {t.get('lut_kernel_ms', 0.0):6.3f} recall {recall_at_k(ids, gt, 10):.4f} "
          f"steps {st['steps'].mean():6.1f} exact {st['exact'].mean():7.1f} pq_eval {st['pq_evaluated'].mean():7.1f} alg_frac {alg / (min(ms) * 1e-3) / 8e12:.3f} "
          f"status {int(st['status'].max())} chk {chk:016x}", flush=True)


if shape == "c5s":      # the PQ-only traversal: kinds = the ADC-only variants to compare (2: table in LDS, 15: split LDS / registers)
    for kd in kinds:
        for L, bw in ((100, 8), (200, 0), (400, 0), (800, 0)):
            run(f"PQ_L{L}_bw{bw or 'None'}[k{kd}]", kd, L=L, beam_width=bw, mode=_ffi.MODE_PQ)
        run(f"M3_PQ_k10_bw64[k{kd}]", kd, L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
    sys.exit(0)
for kd in kinds:
    run(f"M1_L100_bw8[k{kd}]", kd, L=100, beam_width=8, mode=_ffi.MODE_M1)
    run(f"M1_L100_notrim[k{kd}]", kd, L=100, beam_width=0, mode=_ffi.MODE_M1)
run("M1_L400_bw8", -1, L=400, beam_width=8, mode=_ffi.MODE_M1)
run("M2_bw8", -1, L=100, beam_width=8, mode=_ffi.MODE_M2)
run("M2_bw128", -1, L=100, beam_width=128, mode=_ffi.MODE_M2)
run("PQ_L100_bw8", -1, L=100, beam_width=8, mode=_ffi.MODE_PQ)
run("PQ_L200_notrim", -1, L=200, beam_width=0, mode=_ffi.MODE_PQ)
run("PQ_L200_notrim_rerank", -1, L=200, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
run("M3_PQ_k10_bw64", -1, L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
