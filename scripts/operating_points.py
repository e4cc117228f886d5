"""Operating points of the other BASELINE shapes on one MI355X (GPU box): where does recall@10 reach 0.95, and at what QPS?
Sweeps the reference-faithful M1 (L, beam_width, band policy) and the engine's PQ traversal + exact rerank of the L list
(DR_MODE_PQ | DR_F_RERANK: SURVEY.md 8d's definition of c3) on a c3- / c4-shaped index built on the device.
Usage: python scripts/operating_points.py c3 10000000 [nq] [quick]   -> JSON lines in gpurun_out/op_<shape>_<N>.jsonl
(one line per run, written as it goes: a run that dies keeps what it measured). OP_R=<degree> builds the graph with another R."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi                      # noqa: E402
from diskrag_amd.synth import unit_mixture, unit_mixture_parallel, recall_at_k     # noqa: E402

shape, n = sys.argv[1], int(sys.argv[2])
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
spec = sys.argv[4] if len(sys.argv) > 4 else "full"
quick = spec.startswith("quick")
both = spec.endswith("+extra") and spec != "extra"       # "quick+extra" / "full+extra": the grid, then the extra pass, on ONE built index
# d256 / d768 / d960: the other entries of the reference's SUPPORTED_DIMENSIONS (preprocessing/config.py:88), unit-norm mixture
D, m, ncl, latent = {"c3": (1536, 32, 4096, 64), "c4": (96, 16, 4096, 32), "d256": (256, 32, 4096, 32), "d768": (768, 32, 4096, 64),
                     "d960": (960, 32, 4096, 64)}[shape]
R = int(os.environ.get("OP_R", "64"))          # graph degree (OP_R=128: the widest the builder takes)
path = f"gpurun_out/op_{shape}_{n}{'' if spec in ('full', 'quick') else '_' + spec.replace('+', '_')}{'' if R == 64 else '_R%d' % R}.jsonl"
out = open(path, "w")


def emit(rec):
    out.write(json.dumps(rec) + "\n"); out.flush()
    print(json.dumps(rec), flush=True)


t0 = time.perf_counter()
gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
x, q = gen(n, D, n_queries=nq, n_clusters=ncl, seed=11, latent=latent)
gen_s = time.perf_counter() - t0
t0 = time.perf_counter()
ix = HipIndex.create_empty(x, R=R)
up_s = time.perf_counter() - t0
med, bsec = ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
t0 = time.perf_counter()
cb = ix.pq_train(m, n_sample=100000, iters=5)
ix.pq_encode(cb)
pq_s = time.perf_counter() - t0
t0 = time.perf_counter()
gt, _ = ix.bruteforce_topk(q, 10)
emit({"setup": {"shape": shape, "N": n, "D": D, "R": R, "m": m, "nq": nq, "generate_s": gen_s, "upload_s": up_s, "build_s": bsec,
                "pq_s": pq_s, "ground_truth_s": time.perf_counter() - t0}})
del x
ix.batch_upload(q)


def run(tag, **kw):
    try:
        ix.batch_run(10, **kw); ix.batch_sync()
        reps = 2
        t1 = time.perf_counter()
        for _ in range(reps):
            ix.batch_run(10, **kw)
        ix.batch_sync()
        dt = (time.perf_counter() - t1) / reps
        ids, dist, cnt, st = ix.batch_download()
        t = ix.timing()
        alg = float((4.0 * D + st["steps"] * 4.0 * R + st["pq_evaluated"] * float(m) + st["exact"] * 4.0 * D + 80).sum())
        emit({"run": tag, "args": {k: int(v) for k, v in kw.items()}, "qps": nq / dt, "recall_at_10": recall_at_k(ids, gt, 10),
              "kernel_ms": t["search_kernel_ms"], "table_build_kernel_ms": t.get("lut_kernel_ms", 0.0), "variant": t["variant"], "waves_per_cu": t["waves_per_cu"],
              "steps": float(st["steps"].mean()), "exact": float(st["exact"].mean()), "pq_evaluated": float(st["pq_evaluated"].mean()),
              "status_max": int(st["status"].max()), "alg_bytes_per_query": alg / nq,
              "alg_frac_of_8TBps": alg / (t["search_kernel_ms"] * 1e-3) / 8e12})
    except Exception as e:          # a capacity or LDS limit: record it and go on
        emit({"run": tag, "error": str(e)})


def extra_pass():
    # second pass: pin the first operating points at recall >= 0.95 between the grid points of the full sweep, and the
    # exact beam search (M2, beam_search_from_disk) with wider beams
    for bwx in (96, 128, 192, 256):
        run(f"M2_bw{bwx}", L=100, beam_width=bwx, mode=_ffi.MODE_M2)
    for L in ((500, 600) if shape == "c4" else (250, 300)):
        run(f"M1_L{L}_bwNone_policy0", L=L, beam_width=0, mode=_ffi.MODE_M1, band_policy=0)
        run(f"M1_L{L}_bw8_policy0", L=L, beam_width=8, mode=_ffi.MODE_M1, band_policy=0)
    for L in ((300, 350) if shape == "c4" else (250, 300, 350)):
        run(f"PQ_rerank_L{L}_bwNone", L=L, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)

def run_stream(tag, cap, depth, n_sub=48, **kw):
    """host memory -> host memory: a stream of nq-query submits (dr_search_submit / dr_search_wait), `depth` in flight; the library runs the
    submits that wait for the search stream as one launch of up to `cap` queries"""
    try:
        ix.set_coalesce(cap)
        src = _ffi.pinned_empty(q.shape, np.float32); src[:] = q

        def go(n):
            jobs, done, last = [], 0, None
            t1 = time.perf_counter()
            for i in range(n):
                jobs.append(ix.search_submit(src, 10, reuse_outputs=True, **kw))
                if len(jobs) - done >= depth:
                    last = jobs[done].wait(); jobs[done] = None; done += 1
            for j in range(done, len(jobs)):
                last = jobs[j].wait()
            return time.perf_counter() - t1, last
        go(2 * depth); ix.batch_sync()
        s0 = ix.pipeline_stats()
        dt, last = go(n_sub)
        ix.batch_sync()
        s1 = ix.pipeline_stats()
        t = ix.timing()
        nl = max(1, s1["launches"] - s0["launches"])
        emit({"run": tag, "args": {k: int(v) for k, v in kw.items()}, "path": "dr_search_submit/wait, host -> host", "coalesce_cap": cap, "tickets_in_flight": depth,
              "qps": nq * n_sub / dt, "recall_at_10": recall_at_k(last[0], gt, 10), "queries_per_launch": (s1["queries"] - s0["queries"]) / nl,
              "kernel_ms_per_launch": t["search_kernel_ms"], "kernel_ms_per_10k_queries": t["search_kernel_ms"] * 10000.0 / ((s1["queries"] - s0["queries"]) / nl),
              "variant": t["variant"], "status_max": int(last[3]["status"].max())})
    except Exception as e:
        emit({"run": tag, "error": str(e)})
    finally:
        ix.set_coalesce(32768)


if spec == "stream":
    # end of round 4: the shape's operating points at recall >= 0.95 as a host -> host stream of nq-query submits, one launch per submit
    # against shared launches (the tail of a launch: 10 000 queries are 2.4-4.9 queries per wavefront slot)
    pts = ((("M1_L500_bw64", dict(L=500, beam_width=64, mode=_ffi.MODE_M1)), ("PQ_rerank_L400_bw32", dict(L=400, beam_width=32, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)),
            ("M2_bw128", dict(L=100, beam_width=128, mode=_ffi.MODE_M2))) if shape == "c4" else
           (("PQ_rerank_L250_bwNone", dict(L=250, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)), ("M1_L200_bw64", dict(L=200, beam_width=64, mode=_ffi.MODE_M1))))
    for tag, kw in pts:
        run(tag + "_resident", **kw)
        run_stream(tag + "_stream_one_launch_per_submit", nq, 4, **kw)
        run_stream(tag + "_stream_shared_launches", 32768, 14, **kw)
    ix.close()
    sys.exit(0)
if spec == "pqb":
    # round 5: DR_MODE_PQB (the PQ traversal as a batch per step, csrc/pqb_kernel.hpp) + exact rerank of the L list, around the shape's
    # recall-0.95 points, next to round 4's DR_MODE_PQ at the same L / beam_width; then the best points as host -> host streams
    grid = (((350, 64), (400, 32), (350, 0), (300, 64), (500, 32)) if shape == "c4" else ((250, 0), (250, 128), (300, 64), (200, 0), (300, 0), (400, 32)))
    for L, bw in grid:
        run(f"PQ_rerank_L{L}_bw{bw or 'None'}", L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
        for pops in (1, 2):
            run(f"PQB_rerank_L{L}_bw{bw or 'None'}_pops{pops}", L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_POPS(pops))
    # inline neighbour codes (N*R*m bytes: 20 GB at the full c3 size): a row's code words are R*m contiguous bytes = R*m/128 lines instead of R
    # scattered 128-byte lines -- at 10M points the code-word gathers come from HBM and the traversal runs at the chip's rate of random lines
    ix.inline_codes(True)
    for L, bw in grid[:3]:
        for pops in (1, 2):
            run(f"PQB_rerank_L{L}_bw{bw or 'None'}_pops{pops}_inline_codes", L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_POPS(pops))
    for L, bw in grid[:2]:
        kw = dict(L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_POPS(1))
        run_stream(f"PQB_rerank_L{L}_bw{bw or 'None'}_pops1_inline_codes_stream_shared_launches", 32768, 14, **kw)
    ix.inline_codes(False)
    for L, bw in grid[:2]:
        kw = dict(L=L, beam_width=bw, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | _ffi.F_POPS(1))
        run_stream(f"PQB_rerank_L{L}_bw{bw or 'None'}_pops1_stream_one_launch_per_submit", nq, 4, **kw)
        run_stream(f"PQB_rerank_L{L}_bw{bw or 'None'}_pops1_stream_shared_launches", 32768, 14, **kw)
    ix.close()
    sys.exit(0)
if spec == "extra":
    extra_pass()
    ix.close()
    sys.exit(0)
if spec == "nvsweep":
    # round 4: DR_F_NO_VISITED_SET at full size (on the 1.25e8-point c5 shard it is 9-33 % FASTER, at 1M-10M points it was slower)
    for L, bw in (((400, 32), (350, 64), (350, 0)) if shape == "c4" else ((250, 0), (250, 128), (300, 64))):
        for nv in (0, 1):
            run(f"PQ_rerank_L{L}_bw{bw or 'None'}" + ("_no_visited_set" if nv else ""), L=L, beam_width=bw, mode=_ffi.MODE_PQ,
                flags=_ffi.F_RERANK | (_ffi.F_NO_VISITED_SET if nv else 0))
    ix.close()
    sys.exit(0)
if spec == "bwsweep":
    # round 4: intermediate frontier trims (beam_width is a parameter of the reference's search, search_engine.py:477-479; rounds 1-3
    # swept 8 and None only): at equal recall a trim of 32-128 halves the expansions of the no-trim operating points
    for L in ((350, 400, 500) if shape == "c4" else (250, 300, 400)):
        for bw in ((128, 64, 32) if shape == "c4" else (0, 128, 64, 32)):
            run(f"PQ_rerank_L{L}_bw{bw or 'None'}", L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
    if shape != "c4":
        for L in (200, 400):
            for bw in (0, 64, 32):
                run(f"M1_L{L}_bw{bw or 'None'}_policy0", L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=0)
        for bwx in (64, 128):
            run(f"M2_bw{bwx}", L=100, beam_width=bwx, mode=_ffi.MODE_M2)
    ix.close()
    sys.exit(0)
if spec == "m1fine":
    # round 4: the reference-faithful M1 around its recall-0.95 point (long lists: the per-query table is now preferred down to
    # five wavefronts per CU), with and without a frontier trim, next to the shape's other operating points
    for L in ((400, 450, 500) if shape == "c4" else (200, 300, 400)):
        for bw in (0, 64, 32):
            run(f"M1_L{L}_bw{bw or 'None'}_policy0", L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=0)
    ix.debug_force_kind(3)
    run("M1_L500_bwNone_policy0_forced_variant3_shared_codebook" if shape == "c4" else "M1_L400_bwNone_policy0_forced_variant3", L=500 if shape == "c4" else 400,
        beam_width=0, mode=_ffi.MODE_M1, band_policy=0)
    ix.debug_force_kind(-1)
    for L in ((300, 350) if shape == "c4" else (250, 300)):
        run(f"PQ_rerank_L{L}_bwNone", L=L, beam_width=0, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
    run("M2_bw128", L=100, beam_width=128, mode=_ffi.MODE_M2)
    ix.close()
    sys.exit(0)
Ls = (100, 200, 400) if quick else (100, 200, 400, 800)
for L in Ls:
    for bw in (8, 0):
        for pol in (0, 1):
            run(f"M1_L{L}_bw{bw or 'None'}_policy{pol}", L=L, beam_width=bw, mode=_ffi.MODE_M1, band_policy=pol)
for L in Ls:
    for bw in (8, 0):
        run(f"PQ_rerank_L{L}_bw{bw or 'None'}", L=L, beam_width=bw, mode=_ffi.MODE_PQ, flags=_ffi.F_RERANK)
run("M2_bw8", L=100, beam_width=8, mode=_ffi.MODE_M2)
run("M2_bw64", L=100, beam_width=64, mode=_ffi.MODE_M2)
if both:
    extra_pass()
ix.close()
