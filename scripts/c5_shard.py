"""One shard of BASELINE config c5 at its full size on one MI355X (GPU box): N = 1.25e8 x d = 1536, PQ-only.
The vectors are generated chunk by chunk on the host (UnitMixtureStream), encoded on the device and forgotten -- they
are never stored (1.25e8 x 1536 x 4 B = 768 GB); the Vamana graph is built from the code words alone
(dr_build_vamana_pq) and searched with the engine's PQ-only traversal (DR_MODE_PQ) and the reference's (M3 with PQ).
Usage: python scripts/c5_shard.py [N] [chunk_rows] [n_gt_queries]  -> gpurun_out/scale_c5_shard.json"""
import json
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "64")     # the image's BLAS is built for 64 threads; 96 generator threads call into it

import numpy as np  # noqa: E402

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                       # noqa: E402
from diskrag_amd.synth import UnitMixtureStream, recall_at_k  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * 1024 * 1024
NGT = int(sys.argv[3]) if len(sys.argv) > 3 else 64
D, m, R, nq = 1536, 32, 32, 10000
CH -= CH % UnitMixtureStream.BLOCK
out = {"shape": "c5 shard", "N": N, "D": D, "m": m, "R": R, "nq": nq, "chunk_rows": CH}
gen = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=96)


def save():
    json.dump(out, open("gpurun_out/scale_c5_shard.json", "w"), indent=1)


# codebook from a sample of the first chunk (DiskANNPQ.fit on a sample: k-means++ seeding, Lloyd)
t0 = time.perf_counter()
sample = gen.draw(0, 262144)
tmp = HipIndex.create_empty(sample, R=R)
cb, inertia = tmp.pq_train_ex(m, n_sample=50000, max_iter=15, n_init=1, seed=5)
tmp.close()
out["codebook_s"] = time.perf_counter() - t0
del sample

sh = HipIndex.create_codes_empty(N, D, R, cb)
t_gen = t_enc = 0.0
t0 = time.perf_counter()
for r0 in range(0, N, CH):
    rows = min(CH, N - r0)
    t1 = time.perf_counter()
    x = gen.draw(r0, rows)
    t2 = time.perf_counter()
    sh.encode_rows(x, r0)
    t3 = time.perf_counter()
    t_gen += t2 - t1; t_enc += t3 - t2
    del x
out["generate_s"], out["encode_s"], out["stream_total_s"] = t_gen, t_enc, time.perf_counter() - t0
save()
print("encoded", out, flush=True)

medoid, bsec = sh.build_vamana_pq(L_build=64, alpha=1.2, passes=2, seed=7)
out["build_s"], out["medoid"] = bsec, medoid
out["memory_bytes"] = {"codes": N * m, "adjacency": N * R * 4, "first_masks": N * 8, "codebook": 256 * D * 4, "centroid_pair_table": m * 65536 * 4,
                       "build_scratch_rows": N * (R + 64) * 4, "visited_words_per_slot": ((N + 23) // 24 + 3) // 4 * 16,
                       "stored_vectors": 0, "vectors_if_stored": N * D * 4}
save()
print("built", bsec, flush=True)

q = gen.draw(0, nq, stream=1)
# ground truth in the shard's own metric: brute-force ADC top-10 of the first NGT queries (flat scan of all code words)
t0 = time.perf_counter()
gt = np.empty((NGT, 10), dtype=np.uint32)
scan_ms = []
for i in range(NGT):
    _, _, ms, d_all = sh.pq_scan_best(q[i:i + 1], want_output=True)
    part = np.argpartition(d_all[0], 10)[:10]
    gt[i] = part[np.lexsort((part, d_all[0][part]))]
    scan_ms.append(ms)
out["ground_truth"] = {"queries": NGT, "seconds": time.perf_counter() - t0, "flat_scan_kernel_ms": float(np.mean(scan_ms)),
                       "flat_scan_GBps": N * m / (float(np.mean(scan_ms)) * 1e-3) / 1e9}
save()

sh.batch_upload(q)
out["runs"] = {}


def run(tag, **kw):
    sh.batch_run(10, **kw); sh.batch_sync()
    t1 = time.perf_counter()
    for _ in range(2):
        sh.batch_run(10, **kw)
    sh.batch_sync()
    dt = (time.perf_counter() - t1) / 2
    ids, dist, cnt, st = sh.batch_download()
    t = sh.timing()
    alg = float((4.0 * D + st["steps"] * 4.0 * R + st["pq_evaluated"] * float(m) + 80).sum())
    out["runs"][tag] = {"qps": nq / dt, "kernel_ms": t["search_kernel_ms"], "variant": t["variant"], "waves_per_cu": t["waves_per_cu"],
                        "recall_at_10_vs_bruteforce_adc": recall_at_k(ids[:NGT], gt, 10), "steps": float(st["steps"].mean()),
                        "pq_evaluated": float(st["pq_evaluated"].mean()), "status_max": int(st["status"].max()),
                        "alg_bytes_per_query": alg / nq, "alg_frac_of_8TBps": alg / (t["search_kernel_ms"] * 1e-3) / 8e12}
    save()
    print(tag, out["runs"][tag], flush=True)


for L in (100, 200, 400):
    for bw in (8, 0):
        run(f"PQ_L{L}_bw{bw or 'None'}", L=L, beam_width=bw, mode=_ffi.MODE_PQ)
run("M3_PQ_k10_bw8_reference_faithful", L=10, beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
run("M3_PQ_k10_bw64_reference_faithful", L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
sh.close()
print(json.dumps(out))
