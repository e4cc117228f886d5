"""Diagnostic (GPU box): where a LONE query's expansion spends its time (one wavefront on an otherwise idle chip: the /search route's shape).
Needs a -DDR_PHASE_TIMING build (DR_LIB): every stamp drains the memory queues, so a phase's cycles are its exposed latency. usage: exp_phase_single.py"""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=512, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
cb = ix.pq_train(32, n_sample=20000, iters=3); ix.pq_encode(cb)
ix.search_batch(q, 10, L=100, beam_width=8, mode=_ffi.MODE_M1)          # (the regime of this list-size class, byte rows)
names = ["setup", "pop/stop", "adjacency", "visited", "ADC", "exact rows", "decisions", "output"]
for nq in (1, 64):
    acc, nrows, steps, kms = np.zeros(8), np.zeros(8), 0.0, []
    for i in range(0, 256, nq):
        ids, dist, cnt, st = ix.search_batch(q[i:i + nq], 10, L=100, beam_width=8, mode=_ffi.MODE_M1)
        raw = np.array(ix.debug_phase_cycles(), dtype=np.uint64)
        nrows += (raw >> np.uint64(36)).astype(np.float64); raw &= np.uint64((1 << 36) - 1)      # (-DDR_DEC_SUB: row counts ride in the high bits)
        acc += raw.astype(np.float64); steps += st["steps"].sum(); kms.append(ix.timing()["search_kernel_ms"])
    print(f"# M1 L=100 bw=8, {nq} quer{'y' if nq == 1 else 'ies'} per call: kernel_ms {np.mean(kms):.3f} variant {ix.timing()['variant']} expansions/query {steps / 256:.1f}")
    for nme, v in zip(names, acc):
        print(f"{nme:12s} {v / acc.sum() * 100:6.2f}%  per expansion {v / steps:8.0f}")
    if nrows.sum() > 0:      # a -DDR_DEC_SUB build: slots 4 / 7 / 0 hold the decision pass of rows without candidates / on the accept-all path / on the general path
        base = {4: 274.0, 7: 48.0, 0: 45.0, 6: 0.0}      # (what those slots hold per expansion without the sub-stamps: profiles/r05/phase_shares_single_query_m1_end_of_round.txt)
        for slot, nme in ((4, "no candidate"), (7, "all accepted"), (6, "closed form"), (0, "general path")):
            cyc = acc[slot] - base[slot] * steps
            print(f"# decisions, {nme:16s}: {nrows[slot] / 256:6.1f} rows per query, {cyc / max(nrows[slot], 1):7.0f} cycles per row, {cyc / steps:6.0f} per expansion")
