"""An index whose full-precision rows do not fit in HBM, served from the host tier (GPU box with >= 1 TB of host memory):
N x 1536 rows (N = 5e7: 307 GB > the 288 GB of HBM) live in pinned host memory (DR_TIER_HOST); code words, the graph built from
them (dr_build_vamana_pq, R = 128) and the visited words live in HBM. Searches: the PQ traversal in HBM + exact rerank of the L list
from the host rows -- DiskANN's recipe with host DRAM as the slow tier. Exact ground truth for NGT queries is a running brute-force
merge over the chunks while they are generated (as scripts/c5_shard.py does).
usage: host_tier_big.py [N] [n_gt_queries]  -> gpurun_out/r03/host_tier_big.json

NOT MEASURED in round 3: the one attempt (N = 5e7: a 307 GB numpy array + 307 GB of pinned memory) took the pool's box down about
when the tier was being filled -- `free` shows the node's 3 TB, the pod's own memory limit is lower. The script now reads the
cgroup limit and refuses to start when it would need more than 70 % of it; the tier itself is measured at 2M rows
(scripts/exp_host_tier.py, profiles/r03/host_tier_2M_d1536.json)."""
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "64")
import numpy as np  # noqa: E402

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                       # noqa: E402
from diskrag_amd.parallel import merge_topk                   # noqa: E402
from diskrag_amd.synth import UnitMixtureStream, recall_at_k  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
NGT = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
D, m, R, LB, nq, CH = 1536, 32, 128, 128, 10000, 4 * 1024 * 1024
CH -= CH % UnitMixtureStream.BLOCK
out = {"N": N, "D": D, "m": m, "R": R, "L_build": LB, "nq": nq, "rows_bytes_in_host_memory": N * D * 4, "hbm_bytes_total": 288 * 2 ** 30}
OUT = "gpurun_out/r03/host_tier_big.json"


def save():
    json.dump(out, open(OUT, "w"), indent=1)


def vram_used():
    try:
        t = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
        return int([l for l in t.splitlines() if "Used" in l][0].split(":")[-1])
    except Exception:
        return None


def cgroup_limit():
    for pth in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(pth).read().strip()
            if v.isdigit() and int(v) < (1 << 60):
                return int(v)
        except OSError:
            pass
    return None


need = 2 * N * D * 4 + (32 << 30)          # the generated array + the pinned copy + working memory
lim = cgroup_limit()
if lim is None and os.environ.get("HOST_TIER_BIG_FORCE") != "1":
    sys.exit("host_tier_big: no cgroup memory limit found -- refusing to allocate %.0f GB blind (HOST_TIER_BIG_FORCE=1 overrides)" % (need / 1e9))
if lim is not None and need > 0.7 * lim:
    sys.exit("host_tier_big: needs %.0f GB of host memory, the pod's limit is %.0f GB -- choose a smaller N" % (need / 1e9, lim / 1e9))

gen = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=64)
q = gen.draw(0, nq, stream=1)
x = np.empty((N, D), dtype=np.float32)
gt_ids = gt_dist = None
t0 = time.perf_counter()
t_gen = t_gt = 0.0
for r0 in range(0, N, CH):
    rows = min(CH, N - r0)
    t1 = time.perf_counter()
    x[r0:r0 + rows] = gen.draw(r0, rows)
    t2 = time.perf_counter()
    part = HipIndex.create_empty(x[r0:r0 + rows], R=1)
    ci, cd = part.bruteforce_topk(q[:NGT], 10)
    part.close()
    ci = (ci.astype(np.uint64) + r0).astype(np.uint32)
    gt_ids, gt_dist = (ci, cd) if gt_ids is None else merge_topk([gt_ids, ci], [gt_dist, cd], 10)
    t_gen += t2 - t1; t_gt += time.perf_counter() - t2
out["generate_s"], out["exact_ground_truth_s"] = t_gen, t_gt
save()
print("generated", out, flush=True)

t0 = time.perf_counter()
ix = HipIndex.create_empty(x, R=R, vector_tier=_ffi.TIER_HOST)
out["load_into_host_tier_s"] = time.perf_counter() - t0
del x
t0 = time.perf_counter()
cb, _ = ix.pq_train_ex(m, n_sample=50000, max_iter=15, n_init=1, seed=5)
out["codebook_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
ix.pq_encode(cb)
out["encode_from_host_rows_s"] = time.perf_counter() - t0
medoid, bsec = ix.build_vamana_pq(L_build=LB, alpha=1.2, passes=2, seed=7)
out["build_s"] = bsec
out["vram_used_bytes_after_build"] = vram_used()
save()
print("built", out, flush=True)

gt_adc = ix.pq_scan_topk(q[:NGT], 10)[0]
out["adc_top10_vs_exact_top10"] = recall_at_k(gt_adc, gt_ids, 10)
out["runs"] = {}
ix.batch_upload(q)
RUNS = [("PQ+rerank L100 bw8", dict(L=100, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L200 bw8", dict(L=200, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L400 bw8", dict(L=400, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L400", dict(L=400, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
        ("PQ+rerank L800 bw8", dict(L=800, beam_width=8, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
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
