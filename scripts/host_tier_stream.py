"""The host tier at scale, filled as a STREAM (GPU box): N x 1536 rows live in pinned host memory (DR_TIER_HOST), generated chunk
by chunk and written straight into the tier (dr_index_write_rows, round 4) -- the host never holds the rows twice; code words, the
graph built from them (dr_build_vamana_pq, R = 128) and the visited words live in HBM. Searches: the PQ traversal in HBM + exact
rerank of the L list from the host rows -- DiskANN's recipe with host DRAM as the slow tier. Exact ground truth for NGT queries is a
running brute-force merge over the chunks while they are generated.
The pool's pods are limited to 300 GiB of host memory (profiles/r04/box_limits.txt: cgroup memory.max = 322 GB), so an index whose
rows exceed the 288 GB of HBM cannot be hosted here; the script refuses any N whose tier + working set would pass 60 % of the limit.
usage: host_tier_stream.py [N=20000000] [n_gt_queries=1000]  -> gpurun_out/r04/host_tier_stream.json"""
import json
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "64")
import numpy as np  # noqa: E402

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                       # noqa: E402
from diskrag_amd.parallel import merge_topk                   # noqa: E402
from diskrag_amd.synth import UnitMixtureStream, recall_at_k  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
NGT = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
D, m, R, LB, nq = 1536, 32, 128, 128, 10000
CH = 32 * UnitMixtureStream.BLOCK          # 1M rows = 6.4 GB per chunk
out = {"N": N, "D": D, "m": m, "R": R, "L_build": LB, "nq": nq, "rows_bytes_in_host_memory": N * D * 4, "chunk_rows": CH}
OUT = "gpurun_out/r04/host_tier_stream.json"
os.makedirs(os.path.dirname(OUT), exist_ok=True)


def save():
    json.dump(out, open(OUT, "w"), indent=1)


def cgroup_limit():
    for pth in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(pth).read().strip()
            if v.isdigit() and int(v) < (1 << 60):
                return int(v)
        except OSError:
            pass
    return None


need = N * D * 4 + 3 * CH * D * 4 + (8 << 30)          # the tier + chunks in flight + working memory
lim = cgroup_limit()
out["cgroup_limit_bytes"], out["host_bytes_needed"] = lim, need
if lim is None:
    sys.exit("host_tier_stream: no cgroup memory limit found -- refusing to pin %.0f GB blind" % (need / 1e9))
if need > 0.6 * lim:
    sys.exit("host_tier_stream: needs %.0f GB of host memory, 60 %% of the pod's limit is %.0f GB -- choose a smaller N" % (need / 1e9, 0.6 * lim / 1e9))

gen = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=64)
q = gen.draw(0, nq, stream=1)
t0 = time.perf_counter()
ix = HipIndex.create_rows_empty(N, D, R, vector_tier=_ffi.TIER_HOST)
out["tier_alloc_s"] = time.perf_counter() - t0
gt_ids = gt_dist = None
t_gen = t_wr = t_gt = 0.0
for r0 in range(0, N, CH):
    rows = min(CH, N - r0)
    t1 = time.perf_counter()
    x = gen.draw(r0, rows)
    t2 = time.perf_counter()
    ix.write_rows(x, r0)
    t3 = time.perf_counter()
    part = HipIndex.create_empty(x, R=1)
    ci, cd = part.bruteforce_topk(q[:NGT], 10)
    part.close()
    ci = (ci.astype(np.uint64) + r0).astype(np.uint32)
    gt_ids, gt_dist = (ci, cd) if gt_ids is None else merge_topk([gt_ids, ci], [gt_dist, cd], 10)
    t_gen += t2 - t1; t_wr += t3 - t2; t_gt += time.perf_counter() - t3
    del x
out["generate_s"], out["write_rows_s"], out["exact_ground_truth_s"] = t_gen, t_wr, t_gt
out["write_rows_GBps"] = N * D * 4 / t_wr / 1e9
save()
print("filled", out, flush=True)

t0 = time.perf_counter()
cb, _ = ix.pq_train_ex(m, n_sample=50000, max_iter=15, n_init=1, seed=5)
out["codebook_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
ix.pq_encode(cb)
out["encode_from_host_rows_s"] = time.perf_counter() - t0
medoid, bsec = ix.build_vamana_pq(L_build=LB, alpha=1.2, passes=2, seed=7)
out["build_s"] = bsec
save()
print("built", out, flush=True)

gt_adc = ix.pq_scan_topk(q[:NGT], 10)[0]
out["adc_top10_vs_exact_top10"] = recall_at_k(gt_adc, gt_ids, 10)
out["runs"] = {}
ix.batch_upload(q)
RUNS = [("PQ+rerank L100 bw8", dict(L=100, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L150 bw16", dict(L=150, beam_width=16, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L200 bw8", dict(L=200, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L300 bw16", dict(L=300, beam_width=16, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L400 bw8", dict(L=400, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L400", dict(L=400, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ only L200 bw8 (no row is read)", dict(L=200, beam_width=8, mode=_ffi.MODE_PQ))]
for tag, kw in RUNS:
    ix.batch_run(10, **kw); ix.batch_sync()
    t1 = time.perf_counter()
    ix.batch_run(10, **kw); ix.batch_sync()
    dt = time.perf_counter() - t1
    ids, dist, cnt, st = ix.batch_download()
    rows = float(st["exact"].mean())
    out["runs"][tag] = {"qps": nq / dt, "ms": dt * 1e3, "recall_at_10_vs_exact": recall_at_k(ids[:NGT], gt_ids, 10),
                        "recall_at_10_vs_bruteforce_adc": recall_at_k(ids[:NGT], gt_adc, 10), "rows_read_per_query": rows,
                        "row_GBps_over_the_link": rows * D * 4 * nq / dt / 1e9, "status_max": int(st["status"].max())}
    save()
    print(tag, out["runs"][tag], flush=True)
ix.close()
print(json.dumps(out))
