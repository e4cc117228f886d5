#!/usr/bin/env python3
"""bench.py -- QPS at recall@10 >= 0.95 on the SIFT1M-shaped configuration (BASELINE.json configs[1]):
N = 1,000,000 x d = 128 L2, R = 64 slots, PQ m = 32, L_search = 100, batch = 10,000 queries, reference-faithful M1
(SearchEngineCorrect._pq_accelerated_graph_search semantics, search_engine.py:398-506), one MI355X per rank.

A "step" is one pass of the hot path over one 10k-query batch that is already resident in HBM. Multi-GPU runs are
query-sharded replicas: every rank holds the whole index and searches its own 10k batch, there is no
data-path collective, and the job value is the sum of the ranks' queries over the slowest rank's time (weak scaling).

Prints ONE JSON line on rank 0. Synthetic data (no network): diskrag_amd/synth.py; graph, PQ codebook and codes are
built on the device by the engine's own builder before the timed region.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy reaches


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num-vectors", dest="n", type=int, default=1_000_000)
    ap.add_argument("--num-queries", dest="nq", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--R", type=int, default=64)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--bw", type=int, default=8, help="beam_width: 8 = the reference default of search()/the API routes (search_engine.py:530, app.py:96); 0 = None (no frontier trim)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the beam_width=None secondary measurement")
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--L-build", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=2000, help="queries timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("DR_BENCH_BACKEND", "nccl")     # "gloo" lets the N>1 path run where ranks share a GPU
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    import diskrag_amd
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import recall_at_k, sift_like

    if diskrag_amd.device_count() < 1:
        raise RuntimeError("no HIP device: the engine has no CPU fallback")

    # ---------------------------------------------------------------- setup (untimed): data, graph, PQ, ground truth
    t0 = time.time()
    x, q = sift_like(args.n, args.dim, n_queries=args.nq, n_clusters=1024, seed=2024, query_seed=9000 + rank)
    log(f"synthetic data {x.shape} + {q.shape[0]} queries in {time.time() - t0:.1f}s")
    t0 = time.time()
    ix = HipIndex.create_empty(x, R=args.R, device=local_rank % max(1, diskrag_amd.device_count()))
    medoid, build_s = ix.build_vamana(L_build=args.L_build, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
    log(f"vamana graph built on device in {build_s:.1f}s (upload+build {time.time() - t0:.1f}s), medoid {medoid}")
    t0 = time.time()
    cb = ix.pq_train(args.m, n_sample=100_000, iters=8)
    codes = ix.pq_encode(cb, want_codes=(rank == 0 and world == 1 and not args.no_cpu))
    log(f"PQ m={args.m} trained + {args.n} vectors encoded in {time.time() - t0:.1f}s")
    t0 = time.time()
    gt, _ = ix.bruteforce_topk(q, args.k)
    log(f"brute-force ground truth in {time.time() - t0:.1f}s")

    mode = _ffi.MODE_M1
    ix.batch_upload(q)

    def sync_all():
        if dist is not None:
            import torch
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    # ---------------------------------------------------------------- warmup + timed region
    for _ in range(args.warmup):
        ix.batch_run(args.k, L=args.L, beam_width=args.bw, mode=mode)
    ix.batch_sync()
    sync_all()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        ix.batch_run(args.k, L=args.L, beam_width=args.bw, mode=mode)   # queues the step: search kernel on the engine's
    ix.batch_sync()                                                     # stream, tie-order pass overlapping the next step;
    sync_all()                                                          # everything is waited for here, inside the clock
    elapsed = time.perf_counter() - t_start
    kernel_ms = [ix.timing()["search_kernel_ms"]]   # mean launch duration over the timed steps (HIP events on the engine's stream)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ids, dist_out, cnt, st = ix.batch_download()
    timing = ix.timing()
    recall = recall_at_k(ids, gt, args.k)
    if (st["status"] != 0).any():
        raise RuntimeError("work-area overflow during the bench")

    # PCIe-inclusive rate (queries from host memory, results back to host), reported beside the headline
    t1 = time.perf_counter()
    ix.search_batch(q, args.k, L=args.L, beam_width=args.bw, mode=mode)
    pcie_qps = args.nq / (time.perf_counter() - t1)

    # ---------------------------------------------------------------- roofline of the dominant kernel
    # algorithmic bytes per query: B_q = 4D + S*4R + V*m + X*4D + 8k, counters from the engine
    S, V, X = st["steps"].astype(np.float64), st["pq_evaluated"].astype(np.float64), st["exact"].astype(np.float64)
    bytes_q = 4 * args.dim + S * 4 * args.R + V * args.m + X * 4 * args.dim + 8 * args.k
    alg_bytes = float(bytes_q.sum()) + 4 * 256 * args.dim       # + codebook once per batch
    k_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    # HBM traffic per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950
    # read correction applied): measured offline on this same workload, committed under profiles/
    traffic = None
    pmc = ROOT / "profiles" / "r01" / "pmc_traffic.json"
    if pmc.exists() and (args.n, args.nq, args.dim, args.R, args.L, args.m) == (1_000_000, 10_000, 128, 64, 100, 32):
        rec = json.loads(pmc.read_text()).get("beam_width_%d" % args.bw)
        if rec:
            traffic = rec["hbm_bytes_per_launch"]

    # secondary measurement (same index, same queries): beam_width=None, the reference's no-trim mode
    secondary = None
    if not args.no_secondary and args.bw != 0:
        for _ in range(2):
            ix.batch_run(args.k, L=args.L, beam_width=0, mode=mode)
        ix.batch_sync()
        sync_all()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            ix.batch_run(args.k, L=args.L, beam_width=0, mode=mode)
        ix.batch_sync()
        k2 = [ix.timing()["search_kernel_ms"]]
        el2 = time.perf_counter() - t2
        ids2, _, _, st2 = ix.batch_download()
        b2 = (4 * args.dim + st2["steps"].astype(np.float64) * 4 * args.R + st2["pq_evaluated"].astype(np.float64) * args.m +
              st2["exact"].astype(np.float64) * 4 * args.dim + 8 * args.k).sum() + 4 * 256 * args.dim
        secondary = {"beam_width": None, "qps_rank0": args.nq * args.steps / el2, "recall_at_10": recall_at_k(ids2, gt, args.k),
                     "kernel_ms": float(np.mean(k2)), "roofline_frac": b2 / (float(np.mean(k2)) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "exact_distances_per_query": float(st2["exact"].mean())}

    # the same step on the variants that less special data gets: float32 rows (variant 9: data that is not
    # integer-valued) and byte rows with float32 queries (variant 11: integer data, queries that are not)
    float_rows = float_queries = None
    byte_rows = timing["variant"] in (10, 11, 13)

    def forced(kind):
        ix.debug_force_kind(kind)
        for _ in range(2):
            ix.batch_run(args.k, L=args.L, beam_width=args.bw, mode=mode)
        ix.batch_sync()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            ix.batch_run(args.k, L=args.L, beam_width=args.bw, mode=mode)
        ix.batch_sync()
        el3 = time.perf_counter() - t3
        k3 = ix.timing()["search_kernel_ms"]
        ran = ix.timing()["variant"]
        ix.debug_force_kind(-1)
        return {"variant": ran, "qps_rank0": args.nq * args.steps / el3, "kernel_ms": k3,
                "roofline_frac": alg_bytes / (k3 * 1e-3) / 1e9 / HBM_PEAK_GBPS}

    if byte_rows and not args.no_secondary:
        float_rows = forced(9)
        if timing["variant"] == 13:
            float_queries = forced(11)

    total_q = args.nq * world * args.steps
    value = total_q / elapsed
    out = {
        "metric": "QPS @ recall@10>=0.95, SIFT1M-shaped d=128 L2, batch=10k",
        "value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SIFT1M-shaped synthetic (configs[1]): N=%d d=%d L2, R=%d, L_search=%d, PQ m=%d, "
                               "beam_width=%s, k=%d, batch=%d queries/GPU, mode=M1 reference-faithful"
                               % (args.n, args.dim, args.R, args.L, args.m, args.bw or None, args.k, args.nq),
                   "recall_at_10": recall, "build_seconds": build_s, "parallelism": "query-sharded replicas x%d" % world,
                   "qps_pcie_inclusive_rank0": pcie_qps,
                   "per_query": {"expansions": float(S.mean()), "pq_distances": float(st["pq"].mean()), "pq_evaluated": float(V.mean()),
                                 "exact_distances": float(X.mean()), "algorithmic_bytes": float(bytes_q.mean())},
                   "launch": {k_: timing[k_] for k_ in ("variant", "grid", "block", "lds_bytes", "waves_per_cu")},
                   "finalize_kernel_ms": timing["finalize_kernel_ms"], "secondary_no_trim": secondary,
                   "row_storage": ("u8: lossless byte copy of the integer-valued vectors (every component checked; distances "
                                   "bit-identical); roofline.achieved counts the reference's 4*D bytes per scored vector, "
                                   "roofline.traffic is what HBM really moved") if byte_rows else "f32",
                   "query_storage": ("u8: every component of the batch is an integer in [0, 255] (checked per upload), "
                                     "distances by v_dot4_u32_u8 -- the integer sum IS the reference's float32 sum "
                                     "(all partial sums < 2^24)") if timing["variant"] == 13 else "f32",
                   "float32_rows": float_rows, "byte_rows_float32_queries": float_queries},
        "roofline": {"bound": "hbm", "kernel": "search_kernel<128,M1>", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes},
    }

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1): the oracle, timed
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import pyoracle as orc
        adj = ix.get_adjacency()
        cores = os.cpu_count() or 1
        ns = min(args.cpu_sample, args.nq)
        t2 = time.perf_counter()
        oids, odist, ocnt, ost = orc.search_batch(x, adj, q[:ns], medoid, orc.M1, args.k, L=args.L, bw=args.bw,
                                                  codes=codes, codebook=cb, nthreads=cores)
        cpu_s = time.perf_counter() - t2
        same = bool(np.array_equal(oids, ids[:ns]) and
                    np.array_equal(odist.astype(np.float32).view(np.uint32), dist_out[:ns].view(np.uint32)))
        out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": "first %d of the %d bench queries, same index, oracle/ C restatement of M1 "
                                         "on OpenMP threads; GPU results bit-identical on the sample: %s"
                                         % (ns, args.nq, same)}
        if not same:
            raise RuntimeError("GPU results differ from the oracle on the CPU-baseline sample")
    if rank == 0:
        print(json.dumps(out), flush=True)
    ix.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
