"""Cross-process determinism of the reference-faithful PQ traversal (M3 with PQ) at 1M x 1536.
usage: dbg_m3_xproc.py run <tag>   -> /tmp/m3_<tag>.npz (ids, dist, steps) [+ /tmp/m3_index.npz from tag 'a']
       dbg_m3_xproc.py cmp         -> differences between runs a and b, each against the oracle"""
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture

N, D, m = 1000000, 1536, 32
if sys.argv[1] == "run":
    tag = sys.argv[2]
    x, q = unit_mixture(N, D, n_queries=10000, n_clusters=4096, seed=11, latent=64)
    ix = HipIndex.create_empty(x, R=64)
    med, _ = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
    cb = ix.pq_train(m, n_sample=100000, iters=5)
    codes = ix.pq_encode(cb, want_codes=True)
    # the same launches ab_shape.py queues before its M3 run
    ix.batch_upload(q)
    for kw in (dict(L=100, beam_width=8, mode=_ffi.MODE_PQ), dict(L=200, beam_width=0, mode=_ffi.MODE_PQ)):
        for _ in range(4):
            ix.batch_run(10, **kw)
        ix.batch_sync()
    out = {}
    for rep in range(2):
        for _ in range(4):
            ix.batch_run(10, L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
        ix.batch_sync()
        ids, dist, cnt, st = ix.batch_download()
        out[f"ids{rep}"], out[f"dist{rep}"], out[f"steps{rep}"] = ids, dist, st["steps"]
    np.savez(f"/tmp/m3_{tag}.npz", **out)
    if tag == "a":
        np.savez("/tmp/m3_index.npz", adj=ix.get_adjacency(), codes=codes, cb=cb, med=med)
    import hashlib
    h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
    adj_now = ix.get_adjacency()
    print("run", tag, "in-process reps equal:", bool((out["ids0"] == out["ids1"]).all()), "| sha1 adjacency", h(adj_now), "sorted rows", h(np.sort(adj_now, axis=1)),
          "codebook", h(cb), "codes", h(codes), "medoid", med)
else:
    from oracle import pyoracle as orc
    a, b = np.load("/tmp/m3_a.npz"), np.load("/tmp/m3_b.npz")
    z = np.load("/tmp/m3_index.npz")
    x, q = unit_mixture(N, D, n_queries=10000, n_clusters=4096, seed=11, latent=64)
    for ka, kb in (("ids0", "ids0"), ("ids1", "ids1")):
        d = np.nonzero((a[ka] != b[kb]).any(axis=1))[0]
        print(f"a.{ka} vs b.{kb}: {d.size} queries differ", d[:20])
    d = np.nonzero((a["ids0"] != b["ids0"]).any(axis=1) | (a["ids0"] != a["ids1"]).any(axis=1) | (b["ids0"] != b["ids1"]).any(axis=1))[0]
    sel = np.concatenate([d[:50], np.arange(50)])
    w = orc.search_batch(x, z["adj"], q[sel], int(z["med"]), orc.M3, 10, L=10, bw=64, flags=orc.F_USE_PQ, codes=z["codes"], codebook=z["cb"], nthreads=16)
    for name, r in (("a0", a["ids0"]), ("a1", a["ids1"]), ("b0", b["ids0"]), ("b1", b["ids1"])):
        bad = np.nonzero((r[sel] != w[0]).any(axis=1))[0]
        print(name, "vs oracle on", len(sel), "queries:", bad.size, "differ", sel[bad][:10])
    if d.size:
        i = d[0]; j = list(sel).index(i)
        print("query", i, "\n a0", a["ids0"][i], a["dist0"][i], a["steps0"][i], "\n b0", b["ids0"][i], b["dist0"][i], b["steps0"][i], "\n or", w[0][j], w[1][j], w[3][j])
