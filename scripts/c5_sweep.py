"""c5 shard graph quality at a size that fits a short GPU call, with the FULL-SIZE density: the full shard holds
1.25e8 points in 4096 clusters (30.5 k points per cluster); N points in n_clusters = 4096 * N / 1.25e8 clusters have the
same number of cluster-mates per point, which is what decides how hard the ADC ranking is to navigate (the 4M-point /
4096-cluster shard of round 2 reached recall 0.96 where the full shard reached 0.80).
For each (R, L_build) the PQ-only builder (dr_build_vamana_pq) builds the graph from the code words; one more graph is
built from the exact vectors (dr_build_vamana) and searched on the same code words: is the builder's metric the limit,
or the ADC ranking itself? Recall@10 of DR_MODE_PQ against (a) brute-force ADC and (b) exact L2, 1000 queries.
usage: c5_sweep.py N n_clusters "R:Lb,R:Lb,..." [exact]   -> gpurun_out/c5_sweep_<N>_<ncl>.jsonl"""
import json
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "64")
import numpy as np  # noqa: E402

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                       # noqa: E402
from diskrag_amd.synth import UnitMixtureStream, recall_at_k  # noqa: E402

N = int(sys.argv[1]); NCL = int(sys.argv[2])
cfgs = [tuple(int(v) for v in c.split(":")) for c in sys.argv[3].split(",")]
with_exact_graph = len(sys.argv) > 4 and sys.argv[4] == "exact"
D, m, nq, NGT = 1536, int(os.environ.get("C5_M", "32")), 10000, 1000
out = open(f"gpurun_out/c5_sweep_{N}_{NCL}.jsonl", "w")


def emit(rec):
    out.write(json.dumps(rec) + "\n"); out.flush()
    print(json.dumps(rec), flush=True)


gen = UnitMixtureStream(d=D, n_clusters=NCL, seed=11, latent=64, threads=64)
t0 = time.perf_counter()
x = gen.draw(0, N)
q = gen.draw(0, nq, stream=1)
gen_s = time.perf_counter() - t0
full = HipIndex.create_empty(x, R=64)
cb, inertia = full.pq_train_ex(m, n_sample=50000, max_iter=15, n_init=1, seed=5)
t0 = time.perf_counter()
gt_exact, _ = full.bruteforce_topk(q[:NGT], 10)
gt_s = time.perf_counter() - t0
emit({"setup": {"N": N, "n_clusters": NCL, "D": D, "m": m, "generate_s": gen_s, "exact_gt_s": gt_s, "gt_queries": NGT}})

gt_adc = None


def adc_gt(sh):
    return sh.pq_scan_topk(q[:NGT], 10)[0]      # brute-force ADC search on the device (flat scan, top-10 per query)


def sweep(sh, tag, R):
    sh.batch_upload(q)
    for L, bw in ((100, 8), (100, 0), (200, 0), (400, 0), (800, 0)):
        sh.batch_run(10, L=L, beam_width=bw, mode=_ffi.MODE_PQ); sh.batch_sync()
        sh.batch_run(10, L=L, beam_width=bw, mode=_ffi.MODE_PQ)
        ids, dist, cnt, st = sh.batch_download()
        t = sh.timing()
        emit({"graph": tag, "R": R, "L": L, "bw": bw, "kernel_ms": t["search_kernel_ms"], "waves_per_cu": t["waves_per_cu"],
              "recall_vs_adc": recall_at_k(ids[:NGT], gt_adc, 10), "recall_vs_exact": recall_at_k(ids[:NGT], gt_exact, 10),
              "steps": float(st["steps"].mean()), "pq_evaluated": float(st["pq_evaluated"].mean()), "status_max": int(st["status"].max())})


for R, Lb in cfgs:
    sh = HipIndex.create_codes_empty(N, D, R, cb)
    CH = 1 << 20
    for r0 in range(0, N, CH):
        sh.encode_rows(x[r0:r0 + CH], r0)
    if gt_adc is None:
        t0 = time.perf_counter()
        gt_adc = adc_gt(sh)
        emit({"adc_gt_s": time.perf_counter() - t0, "adc_top10_vs_exact_top10": recall_at_k(gt_adc, gt_exact, 10)})
    medoid, bsec = sh.build_vamana_pq(L_build=Lb, alpha=1.2, passes=2, seed=7)
    emit({"build": "pq", "R": R, "L_build": Lb, "build_s": bsec})
    sweep(sh, f"pq_R{R}_Lb{Lb}", R)
    sh.close()

if with_exact_graph:
    R = 64
    medoid, bsec = full.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=False)
    emit({"build": "exact", "R": R, "L_build": 100, "build_s": bsec})
    full.pq_encode(cb)
    sweep(full, "exact_R64_Lb100", R)
full.close()
