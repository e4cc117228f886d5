"""One shard of BASELINE config c5 at its full size on one MI355X (GPU box): N = 1.25e8 x d = 1536, PQ-only.
The vectors are generated chunk by chunk on the host (UnitMixtureStream), encoded on the device and forgotten -- they
are never stored (1.25e8 x 1536 x 4 B = 768 GB); the Vamana graph is built from the code words alone
(dr_build_vamana_pq) and searched with the engine's PQ-only traversal (DR_MODE_PQ) and the reference's (M3 with PQ).
Ground truth for NGT queries, both kinds: EXACT top-10 (a running brute-force merge over the streamed chunks: every chunk
is a temporary index, dr_bruteforce_topk, ids offset by the chunk's first row) and ADC top-10 (dr_pq_scan_topk, a flat
scan of the finished code table).
Usage: python scripts/c5_shard.py [N] [chunk_rows] [n_gt_queries] [R:L_build[,R:L_build...]] [m] [n_clusters]  -> gpurun_out/scale_c5_shard.json
(C5_OUT=<path> names the output, C5_GRID="L:bw,L:bw,..." replaces the standard (L, beam_width) grid by a finer sweep)"""
import json
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "64")     # the image's BLAS is built for 64 threads: no more generator threads than that (more end in "Bad memory unallocation" at exit)

import numpy as np  # noqa: E402

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                       # noqa: E402
from diskrag_amd.parallel import merge_topk                   # noqa: E402
from diskrag_amd.synth import UnitMixtureStream, recall_at_k, recall_at_k_ties  # noqa: E402

argv = sys.argv[1:] + [None] * 8
N = int(argv[0] or 125_000_000)
CH = int(argv[1] or 4 * 1024 * 1024)
NGT = int(argv[2] or 1000)
CFG = [tuple(int(v) for v in c.split(":")) for c in (argv[3] or "32:64").split(",")]      # "R:L_build[,R:L_build...]": graphs built over ONE code table
R, LB = CFG[0]
m = int(argv[4] or 32)
NCL = int(argv[5] or 4096)
D, nq = 1536, 10000
CH -= CH % UnitMixtureStream.BLOCK
out = {"shape": "c5 shard", "N": N, "D": D, "m": m, "graphs": [{"R": r, "L_build": l} for r, l in CFG], "n_clusters": NCL, "nq": nq, "chunk_rows": CH}
gen = UnitMixtureStream(d=D, n_clusters=NCL, seed=11, latent=64, threads=64)
OUT = os.environ.get("C5_OUT", "gpurun_out/scale_c5_shard.json")


def save():
    json.dump(out, open(OUT, "w"), indent=1)


# codebook from a sample of the first chunk (DiskANNPQ.fit on a sample: k-means++ seeding, Lloyd)
t0 = time.perf_counter()
sample = gen.draw(0, 262144)
tmp = HipIndex.create_empty(sample, R=R)
cb, inertia = tmp.pq_train_ex(m, n_sample=50000, max_iter=15, n_init=1, seed=5)
tmp.close()
out["codebook_s"] = time.perf_counter() - t0
del sample

q = gen.draw(0, nq, stream=1)
sh = HipIndex.create_codes_empty(N, D, R, cb)
t_gen = t_enc = t_gt = 0.0
gt_ids = gt_dist = None
t0 = time.perf_counter()
for r0 in range(0, N, CH):
    rows = min(CH, N - r0)
    t1 = time.perf_counter()
    x = gen.draw(r0, rows)
    t2 = time.perf_counter()
    sh.encode_rows(x, r0)
    t3 = time.perf_counter()
    if NGT:
        # exact top-10 of this chunk, merged into the running lists in canonical (distance, id) order
        part = HipIndex.create_empty(x, R=1)
        ci, cd = part.bruteforce_topk(q[:NGT], 10)
        part.close()
        ci = (ci.astype(np.uint64) + r0).astype(np.uint32)
        gt_ids, gt_dist = (ci, cd) if gt_ids is None else merge_topk([gt_ids, ci], [gt_dist, cd], 10)
    t4 = time.perf_counter()
    t_gen += t2 - t1; t_enc += t3 - t2; t_gt += t4 - t3
    del x
out["generate_s"], out["encode_s"], out["exact_ground_truth_s"], out["stream_total_s"] = t_gen, t_enc, t_gt, time.perf_counter() - t0
save()
print("encoded", out, flush=True)

def run(tag, **kw):
    sh.batch_upload(q)          # (the sharded path uses the selected resident batch as scratch: put the bench batch back)
    sh.batch_run(10, **kw); sh.batch_sync()
    t1 = time.perf_counter()
    for _ in range(2):
        sh.batch_run(10, **kw)
    sh.batch_sync()
    dt = (time.perf_counter() - t1) / 2
    ids, dist, cnt, st = sh.batch_download()
    t = sh.timing()
    alg = float((4.0 * D + st["steps"] * 4.0 * R + st["pq_evaluated"] * float(m) + 80).sum())
    out["runs"][tag] = {"qps": nq / dt, "kernel_ms": t["search_kernel_ms"], "table_build_kernel_ms": t["lut_kernel_ms"], "variant": t["variant"],
                        "waves_per_cu": t["waves_per_cu"],
                        "recall_at_10_vs_bruteforce_adc": recall_at_k(ids[:NGT], gt_adc, 10),
                        # distance-based (ties count): a returned code word as near as the 10th of the brute force is a hit
                        "recall_at_10_vs_bruteforce_adc_by_distance": recall_at_k_ties(dist[:NGT], gt_adc_sq, 10),
                        "recall_at_10_vs_exact": recall_at_k(ids[:NGT], gt_ids, 10) if NGT else None, "steps": float(st["steps"].mean()),
                        "pq_evaluated": float(st["pq_evaluated"].mean()), "status_max": int(st["status"].max()),
                        "alg_bytes_per_query": alg / nq, "alg_frac_of_8TBps": alg / (t["search_kernel_ms"] * 1e-3) / 8e12}
    save()
    print(tag, out["runs"][tag], flush=True)


def run_stream(tag, depth=14, n_sub=48, **kw):
    """the same operating point as a host -> host stream of nq-query submits (dr_search_submit / dr_search_wait): the library runs the submits
    that wait for the search stream as one launch (end of round 4: a 10 000-query launch is 4.9 queries per wavefront slot on this kernel)"""
    src = _ffi.pinned_empty(q.shape, np.float32); src[:] = q

    def go(n):
        jobs, done, last = [], 0, None
        t1 = time.perf_counter()
        for i in range(n):
            jobs.append(sh.search_submit(src, 10, reuse_outputs=True, **kw))
            if len(jobs) - done >= depth:
                last = jobs[done].wait(); jobs[done] = None; done += 1
        for j in range(done, len(jobs)):
            last = jobs[j].wait()
        return time.perf_counter() - t1, last
    go(2 * depth); sh.batch_sync()
    s0 = sh.pipeline_stats()
    dt, last = go(n_sub)
    sh.batch_sync()
    s1 = sh.pipeline_stats()
    t = sh.timing()
    qpl = (s1["queries"] - s0["queries"]) / max(1, s1["launches"] - s0["launches"])
    out["runs"][tag] = {"path": "dr_search_submit / dr_search_wait, host -> host, %d submits in flight" % depth, "qps": nq * n_sub / dt, "queries_per_launch": qpl,
                        "kernel_ms_per_launch": t["search_kernel_ms"], "kernel_ms_per_10k_queries": t["search_kernel_ms"] * 10000.0 / qpl,
                        "table_build_kernel_ms_per_launch": t["lut_kernel_ms"], "variant": t["variant"],
                        "recall_at_10_vs_bruteforce_adc": recall_at_k(last[0][:NGT], gt_adc, 10), "status_max": int(last[3]["status"].max())}
    save()
    print(tag, out["runs"][tag], flush=True)


def run_exchange(tag, group=3, n_sub=48, **kw):
    """the same operating point through the graph-sharded path (dr_sharded_submit / dr_sharded_wait, one-rank RCCL communicator): `group`
    consecutive submits per exchange (dr_sharded_set_group: one launch per shard, one all-gather), 3 exchanges' worth of submits in flight"""
    global _xch_comm
    if _xch_comm is None:       # (one communicator for the whole run)
        _xch_comm = _ffi.Comm(_ffi.Comm.unique_id(), 1, 0, 0)
    comm = _xch_comm
    if True:
        src = _ffi.pinned_empty(q.shape, np.float32); src[:] = q
        _ffi.sharded_set_group(sh, group)
        depth = 2 if group == 1 else 3 * group

        def go(n):
            jobs, last = [], None
            t1 = time.perf_counter()
            for i in range(n):
                jobs.append(_ffi.sharded_submit([sh], [0], src, 10, comm=comm, **kw))
                if len(jobs) >= depth:
                    last = jobs.pop(0).wait()
            for j in jobs:
                last = j.wait()
            return time.perf_counter() - t1, last
        go(2 * depth)
        dt, last = go(n_sub)
        _ffi.sharded_set_group(sh, 1)
        out["runs"][tag] = {"path": "dr_sharded_submit / dr_sharded_wait, one-rank RCCL exchange, %d submits per exchange, %d in flight" % (group, depth),
                            "qps": nq * n_sub / dt, "recall_at_10_vs_bruteforce_adc": recall_at_k(last[0][:NGT], gt_adc, 10), "status_max": int(last[2].max()),
                            "ms_of_the_last_exchange": {"search": float(last[3][0]), "all_gather": float(last[3][1]), "merge": float(last[3][2])}}
        save()
        print(tag, out["runs"][tag], flush=True)


_xch_comm = None


# ground truth in the shard's own metric: brute-force ADC top-10 (flat scan of all code words, top-k kept on the device)
t0 = time.perf_counter()
gt_adc, gt_adc_sq, scan_ms = sh.pq_scan_topk(q[:max(NGT, 1)], 10)
# how much of the table shares code words: distinct ADC distances among each query's 64 nearest code words
t64 = sh.pq_scan_topk(q[:min(max(NGT, 1), 100)], 64)[1]
out["adc_ties"] = {"distinct_distances_among_64_nearest_mean": float(np.mean([len(np.unique(r)) for r in t64])),
                   "queries_whose_10th_and_11th_distance_tie": float(np.mean(t64[:, 9] == t64[:, 10]))}
out["ground_truth"] = {"queries": NGT, "adc_seconds": time.perf_counter() - t0, "flat_scan_kernel_ms_per_query": scan_ms / max(NGT, 1),
                       "flat_scan_GBps": N * m * max(NGT, 1) / (scan_ms * 1e-3) / 1e9,
                       "adc_top10_vs_exact_top10": recall_at_k(gt_adc, gt_ids, 10) if NGT else None}
save()

out["runs"] = {}
for gi, (R, LB) in enumerate(CFG):
    if gi > 0:      # another degree over the same code table (the vectors are gone: the code words are copied on the device)
        nxt = HipIndex.create_codes_empty(N, D, R, cb)
        nxt.copy_codes_from(sh)
        sh.close()
        sh = nxt
    medoid, bsec = sh.build_vamana_pq(L_build=LB, alpha=1.2, passes=2, seed=7)
    G = f"R{R}_Lb{LB}"
    out.setdefault("build", {})[G] = {"build_s": bsec, "medoid": medoid,
                                      "memory_bytes": {"codes": N * m, "adjacency": N * R * 4, "first_masks": N * 8 * ((R + 63) // 64), "codebook": 256 * D * 4,
                                                       "centroid_pair_table": m * 65536 * 4, "build_scratch_rows": N * (R + 64) * 4,
                                                       "visited_words_per_slot": ((N + 23) // 24 + 3) // 4 * 16, "stored_vectors": 0, "vectors_if_stored": N * D * 4}}
    save()
    print("built", G, bsec, flush=True)
    sh.batch_upload(q)
    grid = os.environ.get("C5_GRID")        # "L:bw,L:bw,...": a finer sweep around an operating point instead of the standard grid
    if grid:
        for g in grid.split(","):
            parts = g.split(":")
            L, bw = int(parts[0]), int(parts[1])
            nv = "nv" in parts[2:]                            # "L:bw:nv": the same run without a visited set (DR_F_NO_VISITED_SET, round 4)
            pre = "pre" in parts[2:]                          # "...:pre": ... with the next row's ids prefetched into LDS (DR_PQ_ROW_PREFETCH)
            pqb = [p_ for p_ in parts[2:] if p_.startswith("pqb")]      # "L:bw:pqb" / "pqb2": DR_MODE_PQB (round 5), default / 2 pops per step
            if pre: os.environ["DR_PQ_ROW_PREFETCH"] = "1"
            else: os.environ.pop("DR_PQ_ROW_PREFETCH", None)
            if pqb:
                pops = int(pqb[0][3:] or 0)
                kwm = dict(mode=_ffi.MODE_PQB, flags=_ffi.F_POPS(pops))
                tag0 = f"{G}/PQB_L{L}_bw{bw or 'None'}" + (f"_pops{pops}" if pops else "")
            else:
                kwm = dict(mode=_ffi.MODE_PQ, flags=_ffi.F_NO_VISITED_SET if nv else 0)
                tag0 = f"{G}/PQ_L{L}_bw{bw or 'None'}" + ("_no_visited_set" if nv else "") + ("_next_row_prefetch" if pre else "")
            run(tag0, L=L, beam_width=bw, **kwm)
            if parts[-1] == "xch":                            # "...:xch": the same point through the graph-sharded path, 1 and 3 submits per exchange
                for grp in (1, 3):
                    run_exchange(tag0 + "_sharded_path_%d_per_exchange" % grp, group=grp, L=L, beam_width=bw, **kwm)
            if parts[-1] == "stream":                         # "...:stream": the same point again as a host -> host stream with shared launches
                run_stream(tag0 + "_stream_shared_launches", L=L, beam_width=bw, **kwm)
        continue
    for L in (100, 200, 400, 800):
        for bw in (8, 0):
            run(f"{G}/PQ_L{L}_bw{bw or 'None'}", L=L, beam_width=bw, mode=_ffi.MODE_PQ)
    sh.debug_force_kind(2)
    run(f"{G}/PQ_L400_bwNone_variant2_table_in_LDS", L=400, beam_width=0, mode=_ffi.MODE_PQ)
    sh.debug_force_kind(-1)
    run(f"{G}/M3_PQ_k10_bw8_reference_faithful", L=10, beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
    run(f"{G}/M3_PQ_k10_bw64_reference_faithful", L=10, beam_width=64, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
if _xch_comm is not None:
    _xch_comm.close()
sh.close()
print(json.dumps(out))
