"""Experiment (GPU box): recall of M1 at L=100 for a few synthetic generators / builder settings."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import recall_at_k


def gen_lowrank(n, nq, d=128, latent=24, ncl=256, within=0.6, noise=0.05, seed=1):
    rs = np.random.RandomState(seed)
    B = rs.randn(latent, d).astype(np.float32) / np.sqrt(latent)
    cent = rs.randn(ncl, latent).astype(np.float32)
    def draw(cnt, r):
        a = r.randint(0, ncl, size=cnt)
        z = cent[a] + within * r.randn(cnt, latent).astype(np.float32)
        p = z @ B + noise * r.randn(cnt, d).astype(np.float32)
        p = (p + 4.0) * (218.0 / 8.0)
        return np.clip(np.rint(p), 0, 218).astype(np.float32)
    return draw(n, rs), draw(nq, np.random.RandomState(seed + 1))


def gen_iso(n, nq, d=128, ncl=1024, within=0.5, seed=1):
    from diskrag_amd.synth import sift_like
    return sift_like(n, d, nq, ncl, seed, within)


def run(name, x, q, R=64, Lb=100, alpha=1.2):
    t0 = time.time()
    ix = HipIndex.create_empty(x, R=R)
    med, secs = ix.build_vamana(L_build=Lb, alpha=alpha, passes=2, seed=3, pad_with_zero=True)
    cb = ix.pq_train(32, n_sample=20000, iters=4)
    ix.pq_encode(cb)
    gt, _ = ix.bruteforce_topk(q, 10)
    adj = ix.get_adjacency()
    deg = (adj != 0).sum(1).mean()
    out = []
    for L, bw in ((100, 0), (100, 8), (200, 0)):
        ids, dist, cnt, st = ix.search_batch(q, 10, L=L, beam_width=bw, mode=_ffi.MODE_M1)
        t = ix.timing()
        out.append(f"L={L},bw={bw}: recall={recall_at_k(ids, gt):.4f} steps={st['steps'].mean():.0f} visited={st['visited'].mean():.0f} kern_ms={t['search_kernel_ms']:.2f}")
    print(f"[{name}] N={len(x)} build={secs:.1f}s deg={deg:.1f} total={time.time()-t0:.1f}s | " + " | ".join(out), flush=True)
    ix.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    nq = 2000
    x, q = gen_iso(n, nq, ncl=max(16, n // 1000)); run("iso within=0.5", x, q)
    x, q = gen_iso(n, nq, ncl=max(16, n // 1000), within=1.0); run("iso within=1.0", x, q)
    x, q = gen_lowrank(n, nq, latent=24, ncl=max(16, n // 1000), within=0.6); run("lowrank24 w=0.6", x, q)
    x, q = gen_lowrank(n, nq, latent=16, ncl=max(16, n // 1000), within=1.0); run("lowrank16 w=1.0", x, q)
