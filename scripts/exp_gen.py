"""Experiment (GPU box): generator difficulty at 1M."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import recall_at_k

def gen_lowrank(n, nq, d=128, latent=24, ncl=1024, within=0.6, noise=0.05, seed=1):
    rs = np.random.RandomState(seed)
    B = rs.randn(latent, d).astype(np.float32) / np.sqrt(latent)
    cent = rs.randn(ncl, latent).astype(np.float32)
    def draw(cnt, r):
        out = np.empty((cnt, d), dtype=np.float32)
        for s in range(0, cnt, 1 << 18):
            e = min(cnt, s + (1 << 18))
            a = r.randint(0, ncl, size=e - s)
            z = cent[a] + within * r.randn(e - s, latent).astype(np.float32)
            p = z @ B + noise * r.randn(e - s, d).astype(np.float32)
            out[s:e] = np.clip(np.rint((p + 4.0) * (218.0 / 8.0)), 0, 218)
        return out
    return draw(n, rs), draw(nq, np.random.RandomState(seed + 1))

n = 1000000
for (latent, within, noise) in [(32, 0.7, 0.1), (48, 0.7, 0.1), (32, 1.0, 0.2), (64, 0.7, 0.15)]:
    x, q = gen_lowrank(n, 10000, latent=latent, within=within, noise=noise)
    ix = HipIndex.create_empty(x, R=64)
    med, secs = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
    gt, _ = ix.bruteforce_topk(q, 10)
    cb = ix.pq_train(32, n_sample=20000, iters=3)
    ix.pq_encode(cb)
    adj = ix.get_adjacency()
    deg = (adj != 0).sum(1).mean()
    res = []
    for L, bw in ((100, 0), (100, 8)):
        ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1)
        ties = sum(1 for r in dist if len(np.unique(r)) < len(r))
        res.append(f"L={L},bw={bw}: recall={recall_at_k(ids, gt):.4f} steps={st['steps'].mean():.0f} vis={st['visited'].mean():.0f} ins={st['inserts'].mean():.0f} ms={ix.timing()['search_kernel_ms']:.2f} fin={ix.timing()['finalize_kernel_ms']:.2f} tieq={ties}")
    print(f"latent={latent} within={within} noise={noise}: build={secs:.1f}s deg={deg:.1f} | " + " | ".join(res), flush=True)
    ix.close()
