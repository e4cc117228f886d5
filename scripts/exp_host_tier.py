"""Measurement (GPU box): the host tier of the stored vectors against the HBM-resident index, c3-shaped (N x 1536, m = 32,
R = 64): PQ traversal + exact rerank of the L list (the mode the tier is for), M1 (reads one row per expansion) and M2 (reads
every scored row). One graph, built once on the HBM index; the host-tier index gets the same adjacency and codes.
usage: exp_host_tier.py [N]   -> gpurun_out/r03/host_tier.json"""
import json
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import UnitMixtureStream, recall_at_k
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2097152
D, m, R, nq = 1536, 32, 64, 10000
gen = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=64)
x = gen.draw(0, N)
q = gen.draw(0, nq, stream=1)
hbm = HipIndex.create_empty(x, R=R)
medoid, bsec = hbm.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=False)
cb, _ = hbm.pq_train_ex(m, n_sample=50000, max_iter=15, n_init=1, seed=5)
codes = hbm.pq_encode(cb, want_codes=True)
gt, _ = hbm.bruteforce_topk(q[:1000], 10)
adj = hbm.get_adjacency()
t0 = time.perf_counter()
host = HipIndex.create(x, adj, medoid, vector_tier=_ffi.TIER_HOST)
host.set_pq(cb, codes)
out = {"N": N, "D": D, "m": m, "R": R, "nq": nq, "build_s": bsec, "host_tier_load_s": time.perf_counter() - t0,
       "bytes": {"rows_in_host_memory": N * D * 4, "hbm_adjacency": N * R * 4, "hbm_codes": N * m}, "runs": {}}
RUNS = [("PQ+rerank L100 bw8", dict(L=100, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L200", dict(L=200, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L400", dict(L=400, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ only L400 (no row is read)", dict(L=400, beam_width=0, mode=_ffi.MODE_PQ)),
        ("M1 L100 bw8", dict(L=100, beam_width=8, mode=_ffi.MODE_M1)),
        ("M2 bw64", dict(L=100, beam_width=64, mode=_ffi.MODE_M2))]
for tag, kw in RUNS:
    rec = {}
    res = {}
    for name, ix in (("hbm", hbm), ("host", host)):
        ix.batch_upload(q)
        ix.batch_run(10, **kw); ix.batch_sync()
        t1 = time.perf_counter()
        ix.batch_run(10, **kw); ix.batch_sync()
        dt = time.perf_counter() - t1
        ids, dist, cnt, st = ix.batch_download()
        res[name] = (ids, dist)
        rows = float(st["exact"].mean())
        rec[name] = {"qps": nq / dt, "ms": dt * 1e3, "recall_at_10": recall_at_k(ids[:1000], gt, 10), "rows_read_per_query": rows,
                     "row_GBps": rows * D * 4 * nq / dt / 1e9, "status_max": int(st["status"].max())}
    rec["bit_identical"] = bool(np.array_equal(res["hbm"][0], res["host"][0]) and np.array_equal(res["hbm"][1].view(np.uint32), res["host"][1].view(np.uint32)))
    out["runs"][tag] = rec
    print(tag, rec, flush=True)
json.dump(out, open("gpurun_out/r03/host_tier.json", "w"), indent=1)
print(json.dumps(out))
