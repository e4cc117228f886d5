#!/usr/bin/env python3
"""bench.py -- QPS at recall@10 >= 0.95 on the SIFT1M-shaped configuration (BASELINE.json configs[1]):
N = 1,000,000 x d = 128 L2, R = 64 slots, PQ m = 32, L_search = 100, batch = 10,000 queries, reference-faithful M1
(SearchEngineCorrect._pq_accelerated_graph_search semantics, search_engine.py:398-506), one MI355X per rank.

What is measured:
  * `value`: queries per second of a stream of 10k-query batches that start in HOST memory and end as results in HOST
    memory -- dr_search_submit / dr_search_wait (include/diskrag_hip.h): one submit per batch, 14 batches in flight; upload,
    search, tie-order pass and download overlap on separate HIP streams, and the library runs the batches that wait for the
    search stream as ONE launch (2-3 batches per launch: `config.queries_per_launch`; a lone 10k-query launch ends in a tail
    of idle wavefront slots). Distinct batches rotate (--distinct-batches, default 8), so no step replays the previous
    step's queries. `roofline` is per launch of the timed region; `config.one_launch_per_batch` is the same stream with
    one launch per batch and four in flight (how rounds 2-3 ran it).
  * `config.qps_resident`: the same rotation with every batch already resident in HBM (dr_batch_select / dr_batch_run).
  * a "step" is --batches-per-step (default 40) consecutive 10k-query batches, so that the timed region of the
    driver's 20 steps lasts about a second; `ms_per_step` is per step, `config.ms_per_batch` per 10k-query batch.

Multi-GPU (query-sharded replicas): `python bench.py --gpus N` starts N worker processes itself -- before
any GPU call is made in this process -- one per device, each with the whole index and its own batches (weak scaling);
they meet at file barriers in a scratch directory and rank 0 prints the one JSON line (value = all ranks' queries /
slowest rank's time). No data-path collective, no PyTorch. The same workers also run under
`python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE from the environment, same file barriers).

Synthetic data (no network): diskrag_amd/synth.py; graph, PQ codebook and codes are built on the device by the
engine's own builder before the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

# (the host BLAS of this image is built for 64 threads: on a 256-thread box it warns and, with many Python threads,
# corrupts its buffer table at exit -- cap it before numpy loads it; the bench's host work is not BLAS-bound)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "64")

import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what a streaming copy reaches

# (L, beam_width, rerank depth; 0 = the whole list) of DR_MODE_PQB | DR_F_RERANK at recall@10 >= 0.95: "bench" = the default bench-scale index
# (c3 1M, c4 4M points), "full" = the configuration's own size (from `full_from` points on) -- scripts/op_rerank_top.py, profiles/r06/
# c3 1M: L = 100 no trim, rerank of the ADC top 72: 3.46 M QPS resident / 3.65-3.73 M as a stream at recall 0.959 (the whole list: 0.977 at 3.1 M);
# c3 10M: the rerank depth IS the recall (top 200 of L = 250: 0.936; L = 300 top 200: 0.943), and on 20 000 queries L = 250 reads 0.9476, L = 256
# 0.9497 (round 5's 0.9504 was a 10 000-query sample): the first points safely above the bar are L = 264 (0.9528, 1.01 M) and L = 272, beam_width 128
# (0.9539, 1.12 M resident and as a stream; op_rerank_top_c3_10000000_fine.jsonl) -- lists beyond 256 entries are the next kernel size class;
# c4 4M: L = 200, beam_width 8: 4.8 M / 6.65 M at 0.955 (96-d rows: the rerank is a tenth of the call, its depth buys nothing)
OPERATING_POINTS = {"c3": {"bench": (100, 0, 72), "full": (272, 128, 0), "full_from": 5_000_000},
                    "c4": {"bench": (200, 8, 0), "full": (400, 32, 0), "full_from": 50_000_000}}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batches-per-step", dest="bps", type=int, default=40, help="10k-query batches per step")
    ap.add_argument("--distinct-batches", dest="nb", type=int, default=8, help="distinct query batches rotated (<= 16)")
    ap.add_argument("--num-vectors", dest="n", type=int, default=1_000_000)
    ap.add_argument("--num-queries", dest="nq", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=None, help="default 128 (c2), 1536 (c5)")
    ap.add_argument("--R", type=int, default=None, help="default 64 (c2), 128 (c5)")
    ap.add_argument("--L", type=int, default=None, help="default 100 (c2 and c5)")
    ap.add_argument("--bw", type=int, default=None, help="beam_width: default 8 (c2) = the reference default of search()/the API routes (search_engine.py:530, app.py:96), 32 (c5); 0 = None (no frontier trim)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (no trim, float rows, un-rounded data)")
    ap.add_argument("--m", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--L-build", type=int, default=None, help="default 100 (c2), 128 (c5)")
    ap.add_argument("--cpu-sample", type=int, default=2000, help="queries timed on the CPU oracle, all cores (rank 0, N=1 only)")
    ap.add_argument("--cpu-sample-1t", type=int, default=150, help="queries timed on ONE CPU thread")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed region (+ the 8 launches the recall is computed from): what scripts/profile_run.sh runs under "
                         "rocprofv3 --kernel-trace --stats, so that the search kernel's average duration in the stats is the timed region's")
    ap.add_argument("--min-recall", type=float, default=0.95, help="the metric's recall bar: the bench fails below it")
    ap.add_argument("--c5-group", type=int, default=3, help="c5: submits per exchange (dr_sharded_set_group): one launch per shard and one all-gather for that many "
                                                            "consecutive 10k-query submits; 1 = every submit its own exchange")
    ap.add_argument("--c5-mode", default="pqb", choices=["pqb", "pq"], help="c5: the traversal -- pqb = DR_MODE_PQB (round 5: a batch per step on a total order; "
                    "default), pq = DR_MODE_PQ | DR_F_NO_VISITED_SET (round 4's: the sequential statement)")
    ap.add_argument("--rows", default="auto", choices=["auto", "f32"], help="c2: auto = the engine's choice (lossless byte rows + byte queries on integer-valued data: variant 13); "
                    "f32 = the float32-row kernel (variant 9) as the headline -- what every embedding workload gets")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2: the headline (query-sharded replicas); c3: d=1536 PQ traversal + full-precision rerank of the L list; c4: d=96 replicated index, "
                         "reference-faithful M1; c5: graph-sharded PQ-only search with the RCCL top-k exchange. c3 / c4 default to a bench-scale index "
                         "(--num-vectors 10000000 / 100000000 = the configurations' full sizes: they fit one MI355X)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank its own batches of --num-queries; strong (SURVEY.md 8e): ONE stream of "
                         "--num-queries batches, every batch cut into contiguous slices of nq/N queries, one per rank")
    ap.add_argument("--pq-scan-codes", type=int, default=64_000_000, help="code words of the isolated PQ-scan measurement (config.pq_scan)")
    ap.add_argument("--blocking-calls", type=int, default=24, help="blocking dr_search_batch calls timed for config.qps_blocking_call (median)")
    ap.add_argument("--rerank-top", dest="rerank_top", type=int, default=None,
                    help="c3 / c4: rerank only the n list entries with the smallest ADC (DR_F_RERANK_TOP); default: the operating point's; 0 = the whole list")
    ap.add_argument("--full-size", action="store_true",
                    help="c3 / c4 / c5: the configuration's own size on ONE MI355X (c3 10 000 000 x 1536: ~9 min of set-up; c4 100 000 000 x 96: ~12 min; "
                         "c5 one 125 000 000-point shard of the 1B x 1536 shape: ~25 min) instead of the bench-scale default")
    ap.add_argument("--coalesce", type=int, default=65536, help="c2 / c3 / c4: queries one launch of the pipelined path may hold (dr_set_coalesce; the library's own default is 32768)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    # the shape of the configuration (BASELINE.json configs[1] / configs[4]) unless given
    dflt = {"c2": dict(dim=128, R=64, L=100, bw=8, L_build=100), "c5": dict(dim=1536, R=128, L=100, bw=32, L_build=128),
            "c3": dict(dim=1536, R=64, L_build=100), "c4": dict(dim=96, R=64, L_build=100)}[args.config]
    given = argv if argv is not None else sys.argv
    if args.config in ("c3", "c4"):
        if args.n == 1_000_000 and "--num-vectors" not in given:
            args.n = 1_000_000 if args.config == "c3" else 4_000_000
        if args.full_size:
            args.n = 10_000_000 if args.config == "c3" else 100_000_000
        if args.config == "c4" and "--m" not in given:
            args.m = 16
        # The metric reads "QPS @ recall@10 >= 0.95": the default (L, beam_width, rerank depth) is the fastest measured point at or above 0.95 of
        # an index of THIS size (profiles/r06/op_rerank_top_*.jsonl; the bench-scale index reaches the bar with a shorter list than the full-size one)
        op = OPERATING_POINTS[args.config]
        pt = op["full"] if args.n >= op["full_from"] else op["bench"]
        if args.L is None: args.L = pt[0]
        if args.bw is None: args.bw = pt[1]
        if args.rerank_top is None: args.rerank_top = pt[2] if (args.L, args.bw) == (pt[0], pt[1]) else 0
    if args.config == "c5" and args.full_size:
        args.n = 125_000_000
    for name, v in dflt.items():
        if getattr(args, name) is None:
            setattr(args, name, v)
    return args


# ------------------------------------------------------------------------------------------------ ranks and barriers
class Ranks:
    """Rank bookkeeping + file barriers in a scratch directory shared by the ranks of ONE node (no torch, no sockets).

    The directory is fresh for every job. The script's own launcher makes it (mkdtemp) and hands it over in the environment.
    Under torchrun nothing in the environment is unique per job (MASTER_PORT 29500 and run id 'none' by default), so the
    ranks agree on a fresh directory through a handshake in a base directory named from port + run id: every rank r
    publishes a random nonce (`join.r`), rank 0 creates the directory (mkdtemp) and answers each nonce it sees with
    `assign.r` = (nonce, directory), rank r accepts only an answer that carries ITS nonce and confirms with `here.r` inside
    the new directory. Files a crashed earlier job left behind carry other nonces and are ignored, so a barrier can
    never be satisfied, nor a time or an RCCL id read, from another run's files. Every file is written atomically
    (os.replace). A rank that fails drops an `abort` file, which ends the waits of the others at once."""

    def __init__(self):
        env = os.environ
        self._n = 0
        self._base = None
        if "DR_BENCH_RANK" in env:                       # started by this script's own launcher
            self.rank, self.world = int(env["DR_BENCH_RANK"]), int(env["DR_BENCH_WORLD"])
            self.local_rank, self.dir = self.rank, env["DR_BENCH_SYNC_DIR"]
        elif "RANK" in env and "WORLD_SIZE" in env:       # torch.distributed.run / torchrun
            self.rank, self.world = int(env["RANK"]), int(env["WORLD_SIZE"])
            self.local_rank = int(env.get("LOCAL_RANK", self.rank))
            tag = "%s_%s" % (env.get("MASTER_PORT", "0"), env.get("TORCHELASTIC_RUN_ID", "run"))
            self._base = os.path.join(tempfile.gettempdir(), "diskrag_bench_" + "".join(c if c.isalnum() else "_" for c in tag))
            os.makedirs(self._base, exist_ok=True)
            self.dir = self._handshake() if self.world > 1 else None
        else:
            self.rank, self.world, self.local_rank, self.dir = 0, 1, 0, None

    @staticmethod
    def _write(path, data):
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)

    @staticmethod
    def _read(path):
        try:
            with open(path, "rb") as f:
                return f.read()
        except OSError:
            return None

    def _handshake(self, timeout=1800.0):
        B, t0 = self._base, time.time()
        if self.rank == 0:
            fresh = tempfile.mkdtemp(prefix="job_", dir=B)
            seen = {}
            while True:
                missing = [r for r in range(1, self.world) if not os.path.exists(os.path.join(fresh, "here.%d" % r))]
                if not missing:
                    return fresh
                for r in missing:
                    nonce = self._read(os.path.join(B, "join.%d" % r))
                    if nonce and seen.get(r) != nonce:
                        self._write(os.path.join(B, "assign.%d" % r), nonce + b"\n" + fresh.encode())
                        seen[r] = nonce
                if time.time() - t0 > timeout:
                    raise RuntimeError("rank 0: ranks %s never joined" % missing)
                time.sleep(0.002)
        nonce = os.urandom(16).hex().encode()
        self._write(os.path.join(B, "join.%d" % self.rank), nonce)
        while True:
            a = self._read(os.path.join(B, "assign.%d" % self.rank))
            if a and a.split(b"\n", 1)[0] == nonce:
                fresh = a.split(b"\n", 1)[1].decode()
                self._write(os.path.join(fresh, "here.%d" % self.rank), b"x")
                return fresh
            if time.time() - t0 > timeout:
                raise RuntimeError("rank %d: no directory assigned by rank 0" % self.rank)
            time.sleep(0.002)

    def _check_abort(self):
        if self.dir and os.path.exists(os.path.join(self.dir, "abort")):
            raise RuntimeError("rank %d: another rank aborted: %s" % (self.rank, (self._read(os.path.join(self.dir, "abort")) or b"").decode()))

    def abort(self, why):
        """called by a failing rank: the others stop waiting for it"""
        if self.world > 1 and self.dir:
            try:
                self._write(os.path.join(self.dir, "abort"), ("rank %d: %s" % (self.rank, why)).encode())
            except OSError:
                pass

    def barrier(self, timeout=1800.0):
        if self.world == 1:
            return
        self._n += 1
        self._write(os.path.join(self.dir, "b%d.%d" % (self._n, self.rank)), b"x")
        t0 = time.time()
        while True:
            if all(os.path.exists(os.path.join(self.dir, "b%d.%d" % (self._n, r))) for r in range(self.world)):
                return
            self._check_abort()
            if time.time() - t0 > timeout:
                raise RuntimeError("barrier %d timed out on rank %d" % (self._n, self.rank))
            time.sleep(0.0005)

    def put(self, name, obj):
        if self.world == 1:
            self._single = getattr(self, "_single", {})
            self._single[name] = obj
            return
        self._write(os.path.join(self.dir, "%s.%d" % (name, self.rank)), obj if isinstance(obj, bytes) else json.dumps(obj).encode())

    def get(self, name, rank, raw=False, timeout=1800.0):
        if self.world == 1:
            return self._single[name]
        path = os.path.join(self.dir, "%s.%d" % (name, rank))
        t0 = time.time()
        while not os.path.exists(path):
            self._check_abort()
            if time.time() - t0 > timeout:
                raise RuntimeError("waiting for %s timed out" % path)
            time.sleep(0.001)
        data = open(path, "rb").read()
        return data if raw else json.loads(data)

    def gather(self, name, obj):
        """every rank's object, on every rank (call on all ranks)"""
        self.put(name, obj)
        self.barrier()
        return [self.get(name, r) for r in range(self.world)]

    def finish(self):
        """last call of a rank: after it nobody reads the directory any more; under torchrun rank 0 removes it (the
        script's own launcher removes the one it made)."""
        if self.world == 1 or not self.dir:
            return
        self.barrier()
        if self._base is None:
            return
        self._write(os.path.join(self.dir, "bye.%d" % self.rank), b"x")
        if self.rank != 0:
            try:
                os.remove(os.path.join(self._base, "join.%d" % self.rank))
            except OSError:
                pass
            return
        t0 = time.time()
        while not all(os.path.exists(os.path.join(self.dir, "bye.%d" % r)) for r in range(self.world)) and time.time() - t0 < 60:
            time.sleep(0.002)
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)
        for r in range(1, self.world):
            try:
                os.remove(os.path.join(self._base, "assign.%d" % r))
            except OSError:
                pass
        try:
            os.rmdir(self._base)
        except OSError:
            pass


def log(rk, msg):
    if rk.rank == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def _pci_bus_id(device):
    """PCI address (domain:bus:device.function) of HIP device `device`, or None"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
        return None
    return buf.value.decode().strip().lower()


def local_cpus_of(bdf, sysroot="/sys"):
    """the CPUs next to a PCI function (its `local_cpulist`) and its NUMA node, or (None, None) where the tree does not show them"""
    base = os.path.join(sysroot, "bus/pci/devices", bdf)
    try:
        cpus = set()
        for part in open(os.path.join(base, "local_cpulist")).read().strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        node = int(open(os.path.join(base, "numa_node")).read().strip())
    except (OSError, ValueError):
        return None, None
    return cpus, node


def pin_to_gpu_socket(rk, device, sysroot=None, bus_id=_pci_bus_id):
    """N > 1 ranks on one node: every rank's host work -- the byte scan and staging of its batches, its waits -- stays on the CPUs next to ITS
    GPU (`local_cpulist` of the device's PCI function). One rank keeps the whole box (the CPU baseline wants every core). Best effort: a
    container that hides /sys or a runtime without hipDeviceGetPCIBusId leaves the affinity alone. Returns what it did, for the JSON line.
    (`sysroot` / `bus_id`: tests/test_bench_launcher.py runs it against a fake tree of eight devices on two sockets; DR_BENCH_SYSFS_ROOT does
    the same from outside.)"""
    if rk.world < 2 or os.environ.get("DR_BENCH_NO_PIN"):
        return None
    try:
        bdf = bus_id(device)
        if not bdf:
            return None
        cpus, node = local_cpus_of(bdf, sysroot or os.environ.get("DR_BENCH_SYSFS_ROOT", "/sys"))
        if not cpus:
            return None
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "cpus": len(cpus)}
    except Exception:       # noqa: BLE001
        return None


# ------------------------------------------------------------------------------------------------ launcher
def launch(args):
    """`bench.py --gpus N` without a torchrun environment: N worker processes, started BEFORE this process makes any
    GPU call (it never does), one device each, synchronised through files. Rank 0's stdout is this process's stdout."""
    syncdir = tempfile.mkdtemp(prefix="diskrag_bench_")
    procs = []
    argv = [a for a in sys.argv[1:] if a != "--worker"] + ["--worker"]
    for r in range(args.gpus):
        env = dict(os.environ, DR_BENCH_RANK=str(r), DR_BENCH_WORLD=str(args.gpus), DR_BENCH_SYNC_DIR=syncdir)
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # watch every child: the first failure ends the job (the others would sit in a barrier until its timeout)
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for o in live:
                    o.terminate()
        time.sleep(0.02)
    for f in os.listdir(syncdir):
        try:
            os.remove(os.path.join(syncdir, f))
        except OSError:
            pass
    try:
        os.rmdir(syncdir)
    except OSError:
        pass
    return rc


# ------------------------------------------------------------------------------------------------ engines
class StubIndex:
    """Launcher self-test double (DR_BENCH_STUB=1, tests/test_bench_launcher.py): no GPU, no search -- it only lets the
    rank / barrier / JSON plumbing run on a CPU box. Never a fallback: the real path raises without a HIP device."""

    def __init__(self, nq, k):
        self.nq, self.k = nq, k

    def results(self):
        st = np.zeros(self.nq, dtype=[("steps", "<u4"), ("visited", "<u4"), ("exact", "<u4"), ("pq", "<u4"), ("status", "<u4"),
                                      ("inserts", "<u4"), ("pq_evaluated", "<u4"), ("adj_prefetch_hits", "<u4")])
        st["steps"], st["exact"] = 40, 2000
        ids = np.tile(np.arange(self.k, dtype=np.uint32), (self.nq, 1))
        return ids, np.zeros((self.nq, self.k), np.float32), np.full(self.nq, self.k, np.uint32), st


def alg_bytes(st, D, R, m, k, row_bytes=None):
    """algorithmic bytes of one launch: sum_q 4D + S*4R + V*m + X*row_bytes + 8k, + the codebook once.
    row_bytes = 4D is the reference's accounting (a float32 vector per exact distance); the byte-row kernels read D."""
    S, V, X = st["steps"].astype(np.float64), st["pq_evaluated"].astype(np.float64), st["exact"].astype(np.float64)
    per_q = 4 * D + S * 4 * R + V * m + X * (4 * D if row_bytes is None else row_bytes) + 8 * k
    return float(per_q.sum()) + 4 * 256 * D, per_q


def isolated_pq_scan(device, n_codes=64_000_000, m=32, D=128, nq=4):
    """north_star's "PQ-scan kernel >= 40 % of the HBM-read roofline": the isolated ADC kernel (round 6: pq_scan_skew_kernel -- table image
    in LDS with every lane of a lane group on its own table row, code words streamed from the scan-order copy, strict sequential float sum;
    `previous_kernel`: pq_scan_kernel, DR_PQ_SCAN_NO_SKEW=1) over a table of random code words far larger than L2 + Infinity Cache (2 GB);
    algorithmic bytes = N*m per query, duration from HIP events around the table-image kernel + the scan launch."""
    from diskrag_amd import HipIndex
    rs = np.random.default_rng(5)
    codes = rs.integers(0, 256, size=(n_codes, m), dtype=np.uint8)
    cb = rs.standard_normal((m, 256, D // m), dtype=np.float32)
    q = rs.standard_normal((nq, D), dtype=np.float32)
    sc = HipIndex.create_codes(np.zeros((n_codes, 1), dtype=np.uint32), 0, D, cb, codes, device=device)
    del codes
    sc.pq_scan_best(q)
    ms = sorted(sc.pq_scan_best(q[:1])[2] for _ in range(5))          # one query: the code stream against the HBM peak
    os.environ["DR_PQ_SCAN_NO_SKEW"] = "1"
    try:
        ms_old = sorted(sc.pq_scan_best(q[:1])[2] for _ in range(5))
    finally:
        del os.environ["DR_PQ_SCAN_NO_SKEW"]
    msn = sorted(sc.pq_scan_best(q)[2] for _ in range(5))             # nq queries sharing the pass (pq_scan_multi_kernel: 4 per group at m <= 32)
    sc.pq_scan_topk(q[:1], 10)
    mst = sorted(sc.pq_scan_topk(q[:1], 10)[2] for _ in range(5))     # the brute-force ADC search of ONE query (the skewed kernel with a list of k keys per wavefront)
    sc.close()
    gbps = n_codes * m / (ms[2] * 1e-3) / 1e9
    per_pass = 4 if m <= 32 and nq > 2 else 2
    passes = -(-nq // per_pass)
    return {"kernel": "pq_scan_skew_table_kernel + pq_scan_skew_kernel<2, 512, 2, false>", "code_bytes_per_launch": n_codes * m, "kernel_ms_median": ms[2], "GBps": gbps,
            "frac": gbps / HBM_PEAK_GBPS, "queries_per_launch": 1,
            "brute_force_adc_top10_one_query": {"kernel": "pq_scan_skew_kernel<2, 512, 2, false, true> (dr_pq_scan_topk)", "kernel_ms_median": mst[2],
                                                "frac": n_codes * m / (mst[2] * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "previous_kernel": {"kernel": "pq_scan_kernel<2>", "kernel_ms_median": ms_old[2], "frac": n_codes * m / (ms_old[2] * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "shared_pass": {"kernel": "pq_scan_multi_kernel<2, 4, 768>", "queries_per_launch": nq, "queries_per_pass": per_pass, "kernel_ms_median": msn[2],
                            "ms_per_query": msn[2] / nq, "GBps_algorithmic": nq * n_codes * m / (msn[2] * 1e-3) / 1e9,
                            "code_stream_GBps": passes * n_codes * m / (msn[2] * 1e-3) / 1e9,
                            "code_stream_frac": passes * n_codes * m / (msn[2] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "bound": "lds"}}


def worker(args):
    rk = Ranks()
    try:
        rc = worker_c5(args, rk) if args.config == "c5" else worker_shape(args, rk) if args.config in ("c3", "c4") else worker_c2(args, rk)
    except BaseException as e:          # the other ranks stop waiting for this one
        rk.abort("%s: %s" % (type(e).__name__, e))
        raise
    rk.finish()
    return rc


def slice_of(nq, world, rank):
    """strong scaling: rank r owns the contiguous slice [lo, hi) of every nq-query batch"""
    return (nq * rank) // world, (nq * (rank + 1)) // world


def worker_c2(args, rk):
    stub = os.environ.get("DR_BENCH_STUB") == "1"
    strong = args.scaling == "strong"
    lo, hi = slice_of(args.nq, rk.world, rk.rank) if strong else (0, args.nq)
    nb = max(1, min(args.nb, 16))
    nq_job, k, D = args.nq, args.k, args.dim     # queries per batch of the whole job
    nq = hi - lo                                 # ... and of this rank (strong scaling: its slice)
    if nq < 1:
        raise RuntimeError("strong scaling: rank %d has an empty slice (%d queries over %d ranks)" % (rk.rank, nq_job, rk.world))
    launches = args.steps * args.bps

    if stub:
        if os.environ.get("DR_BENCH_STUB_FAIL_RANK") == str(rk.rank):
            raise RuntimeError("stub failure requested on rank %d" % rk.rank)
        time.sleep(0.05 * (1 + rk.rank))
        rk.barrier()
        t0 = time.perf_counter()
        time.sleep(0.001 * launches)
        el = time.perf_counter() - t0
        times = rk.gather("t", el)
        slices = rk.gather("slice", [lo, hi])
        if rk.rank == 0:
            total_q = nq_job * launches * (1 if strong else rk.world)
            print(json.dumps({"metric": "launcher self-test (stub engine)", "value": total_q / max(times),
                              "unit": "queries/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": max(times) / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
                              "vs_baseline": None, "dtype": "none", "data": "stub",
                              "config": {"workload": "stub", "per_rank_seconds": times, "per_rank_slice": slices}}), flush=True)
        return 0

    import diskrag_amd
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import recall_at_k, sift_like

    ndev = diskrag_amd.device_count()
    if ndev < 1:
        raise RuntimeError("no HIP device: the engine has no CPU fallback")
    device = rk.local_rank % ndev
    mode = _ffi.MODE_M1
    base_kind = 9 if args.rows == "f32" else -1          # --rows f32: the float32-row kernel for every launch of the run
    pinned_to = rk.gather("pin", pin_to_gpu_socket(rk, device))

    # ---------------------------------------------------------------- setup (untimed): data, graph, PQ, ground truth
    t_setup = time.time()
    t0 = time.time()
    if strong:      # the job's batches are the same on every rank; this rank keeps its slice of each
        x, q_job = sift_like(args.n, D, n_queries=nq_job * nb, n_clusters=1024, seed=2024, query_seed=9000)
        q_all = np.ascontiguousarray(q_job.reshape(nb, nq_job, D)[:, lo:hi].reshape(nb * nq, D))
        del q_job
    else:
        x, q_all = sift_like(args.n, D, n_queries=nq * nb, n_clusters=1024, seed=2024, query_seed=9000 + rk.rank)
    log(rk, f"synthetic data {x.shape} + {nb} distinct batches of {nq} queries in {time.time() - t0:.1f}s")
    t0 = time.time()
    ix = HipIndex.create_empty(x, R=args.R, device=device)
    medoid, build_s = ix.build_vamana(L_build=args.L_build, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
    log(rk, f"vamana graph built on device {device} in {build_s:.1f}s (upload+build {time.time() - t0:.1f}s), medoid {medoid}")
    t0 = time.time()
    if args.headline_only:
        args.no_cpu = args.no_secondary = True
    want_cpu = rk.rank == 0 and rk.world == 1 and not args.no_cpu
    cb = ix.pq_train(args.m, n_sample=100_000, iters=8)
    codes = ix.pq_encode(cb, want_codes=want_cpu)
    log(rk, f"PQ m={args.m} trained + {args.n} vectors encoded in {time.time() - t0:.1f}s")
    t0 = time.time()
    gt, _ = ix.bruteforce_topk(q_all, k)
    log(rk, f"brute-force ground truth for {nq * nb} queries in {time.time() - t0:.1f}s")

    # what a rank's set-up costs the node (every rank generates the data, builds the graph and computes its own ground truth: N ranks = N times
    # this host memory and N concurrent builds -- DESIGN.md section 7 states both at the driver's N = 8)
    import resource
    setup_all = rk.gather("setup", {"setup_seconds": time.time() - t_setup, "max_rss_mb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0})

    # the batches as a caller would hold them: page-locked host arrays (dr_host_alloc) -- and pageable copies for the
    # secondary figure
    qb = []
    for b in range(nb):
        a = _ffi.pinned_empty((nq, D), np.float32)
        a[:] = q_all[b * nq:(b + 1) * nq]
        qb.append(a)

    def run_resident(bw, n_launch, kind=None, batches=None):
        """n_launch launches rotating over the resident batches; returns seconds (host clock) and mean kernel ms"""
        ix.debug_force_kind(base_kind if kind is None else kind)
        nbb = len(batches) if batches is not None else nb
        for i in range(min(4, n_launch)):
            ix.batch_select(i % nbb); ix.batch_run(k, L=args.L, beam_width=bw, mode=mode)
        ix.batch_sync()
        t1 = time.perf_counter()
        for i in range(n_launch):
            ix.batch_select(i % nbb)
            ix.batch_run(k, L=args.L, beam_width=bw, mode=mode)
        ix.batch_sync()
        el = time.perf_counter() - t1
        tm = ix.timing()
        ix.debug_force_kind(base_kind)
        return el, tm

    def collect(bw, kind=None):
        """one launch per resident batch, results of all of them (recall, counters)"""
        ix.debug_force_kind(base_kind if kind is None else kind)
        outs = []
        for b in range(nb):
            ix.batch_select(b)
            ix.batch_run(k, L=args.L, beam_width=bw, mode=mode)
            outs.append(ix.batch_download())
        ix.debug_force_kind(base_kind)
        ids = np.concatenate([o[0] for o in outs]); st = np.concatenate([o[3] for o in outs])
        dist = np.concatenate([o[1] for o in outs])
        if (st["status"] != 0).any():
            raise RuntimeError("work-area overflow during the bench")
        return ids, dist, st

    for b in range(nb):
        ix.batch_select(b)
        ix.batch_upload(qb[b])
    ix.batch_select(0)
    ix.debug_force_kind(base_kind)

    # Launches of the pipelined path hold up to `coalesce` queries: the library's default is 32 768 (three 10 000-query batches); the bench asks for
    # 65 536 (dr_set_coalesce: six batches per launch) -- a launch on 4096 persistent wavefronts ends about one query (0.3 ms) after its ideal finish
    # whatever it holds, so twice the queries per launch is 0.98 -> 0.94 ms of kernel per 10 000 queries and +3 % host -> host, four of four interleaved
    # rounds (profiles/r06/ab/ab_launches_of_up_to_65536_queries.jsonl); the price is latency: a batch spends ~25 ms in the stream instead of ~14
    # (--coalesce 32768 restores the default, 10240 one batch per launch)
    coalesce = args.coalesce
    ix.set_coalesce(coalesce)

    def tickets_in_flight(n_q):
        """submits a caller keeps in flight: PIPE_DEPTH launches' worth (a launch coalesces submits up to `coalesce` queries) + 2"""
        return min(_ffi.MAX_TICKETS - 2, _ffi.PIPE_DEPTH * max(1, coalesce // n_q) + 2)

    # ---------------------------------------------------------------- the headline: host memory -> host memory, pipelined
    def run_pipelined(n_launch, sources, depth=_ffi.PIPE_DEPTH):
        """a stream of n_launch submits, `depth` tickets in flight; the library coalesces the submits that find the search stream
        busy into one launch of up to 32768 queries (10k-query batches: 2-3 per launch -- a 10k-query launch is 2.4 queries per
        wavefront slot and ends in a tail of idle slots; small batches: up to PIPE_DEPTH launches' worth of submits)"""
        jobs, done = [], 0
        t1 = time.perf_counter()
        for i in range(n_launch):
            jobs.append(ix.search_submit(sources[i % len(sources)], k, L=args.L, beam_width=args.bw, mode=mode, reuse_outputs=True))
            if len(jobs) - done >= depth:
                jobs[done].wait(); jobs[done] = None; done += 1
        last = None
        for j in range(done, len(jobs)):
            last = jobs[j].wait()
            if j < len(jobs) - 1:
                jobs[j] = None
        el = time.perf_counter() - t1
        return el, last

    depth_head = tickets_in_flight(nq)
    run_pipelined(max(3 * depth_head, args.warmup * args.bps), qb, depth_head)     # warm-up (also sizes every buffer of the pipeline)
    ix.batch_sync()
    rk.barrier()
    ps0 = ix.pipeline_stats()
    elapsed, last = run_pipelined(launches, qb, depth_head)
    ix.batch_sync()                                            # everything is waited for inside the clock (it already is)
    tm_head = ix.timing()
    ps1 = ix.pipeline_stats()
    n_kernel_launches = max(1, ps1["launches"] - ps0["launches"])
    q_per_launch = (ps1["queries"] - ps0["queries"]) / n_kernel_launches       # queries one search-kernel launch of the timed region held
    submits_per_launch = (ps1["tickets"] - ps0["tickets"]) / n_kernel_launches
    times = rk.gather("t_head", elapsed)
    elapsed_job = max(times)
    slices = rk.gather("slice", [lo, hi])
    if strong and rk.rank == 0:      # the slices tile every batch exactly once
        edges = sorted(slices)
        if edges[0][0] != 0 or edges[-1][1] != nq_job or any(a[1] != b[0] for a, b in zip(edges, edges[1:])):
            raise RuntimeError("strong scaling: slices %s do not tile [0, %d)" % (edges, nq_job))
    total_q = launches * (nq_job if strong else nq * rk.world)      # queries the whole job answered in the timed region
    value = total_q / elapsed_job

    # ---------------------------------------------------------------- N > 1, weak scaling: the STRONG-scaling figure in the same line
    # (SURVEY.md 8e / BASELINE metric "batch=10k; 1/2/4/8": ONE stream of nq-query batches, every batch cut into N contiguous
    # slices, one per GPU -- the same job batches on every rank, generated from the common seed)
    strong_cfg = None
    if rk.world > 1 and not strong:
        slo, shi = slice_of(nq_job, rk.world, rk.rank)
        _, q_job = sift_like(args.n, D, n_queries=nq_job * nb, n_clusters=1024, seed=2024, query_seed=9000, queries_only=True)
        n_s = shi - slo
        q_s = np.ascontiguousarray(q_job.reshape(nb, nq_job, D)[:, slo:shi].reshape(nb * n_s, D))
        del q_job
        ix.batch_select(15)             # (brute force and blocking calls use the selected resident batch as scratch: keep them off the bench's)
        gt_s, _ = ix.bruteforce_topk(q_s, k)
        src_s = []
        for b in range(nb):
            a = _ffi.pinned_empty((n_s, D), np.float32)
            a[:] = q_s[b * n_s:(b + 1) * n_s]
            src_s.append(a)
        depth_s = tickets_in_flight(n_s)
        run_pipelined(3 * depth_s, src_s, depth_s)
        ix.batch_sync()
        rk.barrier()
        sp0 = ix.pipeline_stats()
        el_s, _ = run_pipelined(launches, src_s, depth_s)
        ix.batch_sync()
        sp1 = ix.pipeline_stats()
        t_s = rk.gather("t_strong", el_s)
        ids_s = np.concatenate([ix.search_batch(a, k, L=args.L, beam_width=args.bw, mode=mode)[0] for a in src_s])
        if nb == 16:
            ix.batch_upload(qb[15])
        ix.batch_select(0)
        rec_s = rk.gather("recall_strong", [recall_at_k(ids_s, gt_s, k), n_s])
        strong_cfg = {"value": nq_job * launches / max(t_s), "unit": "queries/s", "scaling": "strong",
                      "what": "ONE stream of %d-query batches, every batch cut into %d contiguous slices, one per GPU; value = the stream's "
                              "queries / slowest rank's time, host memory -> host memory" % (nq_job, rk.world),
                      "queries_per_batch_per_gpu": n_s, "batches": launches, "per_rank_seconds": t_s, "tickets_in_flight": depth_s,
                      "recall_at_10": sum(r * n for r, n in rec_s) / sum(n for _, n in rec_s),
                      "rank0_submits_per_launch": (sp1["tickets"] - sp0["tickets"]) / max(1, sp1["launches"] - sp0["launches"])}

    # ---------------------------------------------------------------- the same rotation, batches resident in HBM
    rk.barrier()
    if args.headline_only:
        res_times, tm_res, qps_resident = None, {"search_kernel_ms": None}, None
    else:
        el_res, tm_res = run_resident(args.bw, launches)
        res_times = rk.gather("t_res", el_res)
        qps_resident = total_q / max(res_times)

    # results of every distinct batch: recall, counters, algorithmic bytes
    if args.headline_only:
        # (under rocprofv3: every launch of the process is then one of the pipelined path's, so that the kernel's average duration in
        # the --stats summary is the timed region's)
        jobs_c = [ix.search_submit(qb[b], k, L=args.L, beam_width=args.bw, mode=mode) for b in range(nb)]
        outs_c = [j.wait() for j in jobs_c]
        ids = np.concatenate([o[0] for o in outs_c]); dist_out = np.concatenate([o[1] for o in outs_c]); st = np.concatenate([o[3] for o in outs_c])
        if (st["status"] != 0).any():
            raise RuntimeError("work-area overflow during the bench")
    else:
        ids, dist_out, st = collect(args.bw)
    # the pipelined path (what `value` times) must have produced the same answers as the resident path the recall is
    # computed from: its last waited batch against the same batch of collect()
    lb = (launches - 1) % nb
    if not (np.array_equal(last[0], ids[lb * nq:(lb + 1) * nq]) and
            np.array_equal(last[1].view(np.uint32), dist_out[lb * nq:(lb + 1) * nq].view(np.uint32)) and
            np.array_equal(last[3]["status"], st["status"][lb * nq:(lb + 1) * nq])):
        raise RuntimeError("the pipelined path (dr_search_submit/wait) and the resident path disagree on batch %d" % lb)
    recall_local = recall_at_k(ids, gt, k)
    recalls = rk.gather("recall", [recall_local, nq])
    recall = sum(r * n for r, n in recalls) / sum(n for _, n in recalls) if strong else recall_local
    if recall < args.min_recall:
        raise RuntimeError(f"recall@{k} = {recall:.4f} is below the metric's bar {args.min_recall}")
    variant_launched = ix.timing()["variant"]
    variant = {16: 11, 17: 13}.get(variant_launched, variant_launched)      # (16 / 17: the same kernels in 4-wavefront workgroups, batches below 4096 queries)
    a_all, per_q = alg_bytes(st, D, args.R, args.m, k)
    per_launch = q_per_launch / nq                               # the timed region's launches hold this many batches on average
    alg_launch = ((a_all - 4 * 256 * D) / nb) * per_launch + 4 * 256 * D      # per launch of the timed region (q_per_launch queries)
    alg_launch_1 = (a_all - 4 * 256 * D) / nb + 4 * 256 * D                   # per one-batch launch (the resident legs below)
    k_ms = float(tm_head["search_kernel_ms"])                    # mean launch duration over the timed pipelined region

    # PCIe-inclusive rate from PAGEABLE caller memory (the library stages it), for reference
    qb_pageable = [np.array(a) for a in qb[:min(nb, 4)]]
    qps_pageable = None
    if not args.headline_only:
        el_pg, _ = run_pipelined(max(8, launches // 8), qb_pageable, depth_head)
        qps_pageable = nq * max(8, launches // 8) / el_pg
    # ... and SURVEY.md 8d's literal metric: nq / wall time of ONE blocking dr_search_batch call (upload, search, tie order,
    # download, nothing overlapped), median over --blocking-calls calls rotating the distinct batches (pageable sources)
    ix.batch_select(15)             # (a blocking call uploads into the selected resident batch: keep it off the bench's)
    call_s, call_pieces = [], []
    for i in range(0 if args.headline_only else max(3, args.blocking_calls) + 2):
        src = qb_pageable[i % len(qb_pageable)]
        t1 = time.perf_counter()
        ix.search_batch(src, k, L=args.L, beam_width=args.bw, mode=mode)
        call_s.append(time.perf_counter() - t1)
        tmc = ix.timing()
        call_pieces.append((call_s[-1] * 1e3, tmc["h2d_ms"], tmc["lut_kernel_ms"], tmc["search_kernel_ms"], tmc["finalize_kernel_ms"]))
    # the pieces of the median call (device events; "host_and_gaps" = wall - their sum: the staging pass over the pageable batch that runs
    # ahead of the copies, the bound kernels, launch gaps, the final synchronisation and the copy out of the page-locked slab)
    blocking_pieces = None
    if len(call_pieces) > 2:
        mid = sorted(call_pieces[2:])[len(call_pieces[2:]) // 2]
        blocking_pieces = {"wall": mid[0], "h2d_copy": mid[1], "table_kernel": mid[2], "search_kernel": mid[3], "tie_order_pass": mid[4],
                           "host_and_gaps": mid[0] - sum(mid[1:])}
    call_s = sorted(call_s[2:]) or [float("inf")]     # (the first two calls size the slot's buffers)
    qps_one_call = nq / call_s[len(call_s) // 2]
    # the reference's real serving shape (search_engine.py:530-614, app.py:84-130): ONE query per blocking call, host buffer in, results out --
    # p50 over 300 calls at the API's defaults (k 5, L 20, beam_width 8) and at this bench's point; and a 10 000-query batch at the API's
    # list size through the resident path (round 5: "ask later")
    one_query_ms, qps_api_default = {}, None
    if not args.headline_only and mode == _ffi.MODE_M1:
        for tag, kk, LL in (("api_default_k5_L20", 5, 20), ("bench_point", k, args.L)):
            ts = []
            for i in range(330):
                src = qb_pageable[0][i % nq: i % nq + 1]
                t1 = time.perf_counter()
                ix.search_batch(src, kk, L=LL, beam_width=args.bw, mode=mode)
                ts.append(time.perf_counter() - t1)
            one_query_ms[tag] = float(np.percentile(np.array(ts[30:]) * 1e3, 50))
        ix.batch_upload(qb_pageable[0])
        for _ in range(2): ix.batch_run(5, L=20, beam_width=args.bw, mode=mode)
        ix.batch_sync()
        t1 = time.perf_counter()
        for _ in range(10): ix.batch_run(5, L=20, beam_width=args.bw, mode=mode)
        ix.batch_sync()
        qps_api_default = nq * 10 / (time.perf_counter() - t1)
    if nb == 16:
        ix.batch_upload(qb[15])
    ix.batch_select(0)

    # HBM traffic per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 read
    # correction applied): collected offline on this same workload by scripts/profile_run.sh, committed under profiles/
    traffic, traffic_src, traffic_2 = None, None, None
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        pmc = ROOT / "profiles" / rnd / ("pmc_traffic.json" if args.rows == "auto" else "pmc_traffic_f32_rows.json")
        if traffic is None and pmc.exists() and (args.n, nq, D, args.R, args.L, args.m) == (1_000_000, 10_000, 128, 64, 100, 32):
            doc = json.loads(pmc.read_text())
            rec = doc.get("beam_width_%d" % args.bw)
            if rec and (rnd != "r01" or variant == 13):
                traffic = rec["hbm_bytes_per_launch"]
                # (round 6: the counters were also collected with two / three batches per launch, the shape of the bench's coalesced launches)
                for nq2 in (30000, 20000):
                    t2 = (doc.get("queries_per_launch_%d" % nq2) or {}).get("hbm_bytes_per_launch") if args.bw == 8 else None
                    if t2 and traffic_2 is None:
                        traffic_2 = (t2, nq2 / 10000.0)
                traffic_src = ("profiles/%s/" + pmc.name + ": rocprofv3 --pmc read requests by size (TCC_EA0_RDREQ_32B/64B/128B; FETCH_SIZE x2 before round 4) + WRITE_SIZE, separate passes over scripts/pmc_target.py "
                               "-- the same workload and search kernel (the file carries the library's hash), collected by scripts/profile_run_r06.sh in ANOTHER run on another "
                               "box of the pool (a PMC pass cannot share a process with the timed region) with one AND three batches per launch (round 6; one only before): "
                               "interpolated to this run's batches per launch; hbm_frac divides it by THIS run's kernel time") % rnd

    byte_rows = variant in (11, 13)
    # the kernel's own necessary bytes: a scored vector is D bytes for the byte-row variants, 4D for float rows
    a_k, _ = alg_bytes(st, D, args.R, args.m, k, row_bytes=D if byte_rows else 4 * D)
    alg_kernel = ((a_k - 4 * 256 * D) / nb) * per_launch + 4 * 256 * D
    alg_kernel_1 = (a_k - 4 * 256 * D) / nb + 4 * 256 * D
    achieved = alg_kernel / (k_ms * 1e-3) / 1e9
    if traffic is not None:
        # the PMC passes ran one (and, since round 6, two) batches per launch; this run's launches hold `per_launch` batches: linear in between
        traffic = traffic + (per_launch - 1.0) * (traffic_2[0] - traffic) / (traffic_2[1] - 1.0) if traffic_2 else traffic * per_launch
    # one launch per batch, PIPE_DEPTH in flight (how rounds 2-3 ran the headline): the same stream without coalescing
    one_per_launch = None
    if not args.headline_only:
        ix.set_coalesce(nq)
        run_pipelined(3 * _ffi.PIPE_DEPTH, qb)
        ix.batch_sync()
        el_1, _ = run_pipelined(max(40, launches // 2), qb)
        ix.batch_sync()
        one_per_launch = {"qps": nq * max(40, launches // 2) / el_1, "kernel_ms": float(ix.timing()["search_kernel_ms"]),
                          "tickets_in_flight": _ffi.PIPE_DEPTH}
        ix.set_coalesce(coalesce)
    secondary = float_rows = float_queries = unrounded = None
    if not args.no_secondary and rk.world == 1:
        n_sec = max(40, launches // 4)
        if args.bw != 0:        # beam_width=None, the reference's no-trim mode
            el2, tm2 = run_resident(0, n_sec)
            ids2, _, st2 = collect(0)
            a2, _ = alg_bytes(st2, D, args.R, args.m, k, row_bytes=D if byte_rows else 4 * D)
            a2 = (a2 - 4 * 256 * D) / nb + 4 * 256 * D
            secondary = {"beam_width": None, "qps_resident": nq * n_sec / el2, "recall_at_10": recall_at_k(ids2, gt, k),
                         "kernel_ms": tm2["search_kernel_ms"], "roofline_frac": a2 / (tm2["search_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "exact_distances_per_query": float(st2["exact"].mean())}

        def forced(kind):
            el3, tm3 = run_resident(args.bw, n_sec, kind=kind)
            return {"variant": tm3["variant"], "qps_resident": nq * n_sec / el3, "kernel_ms": tm3["search_kernel_ms"],
                    "roofline_frac": (alg_launch_1 if tm3["variant"] not in (11, 13) else alg_kernel_1) / (tm3["search_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}

        if byte_rows:
            float_rows = forced(9)          # float32 rows: what data that is not integer-valued gets
            if variant == 13:
                float_queries = forced(11)  # byte rows, float32 queries: integer data, queries that are not

    # What one GPU does with the per-GPU slice of a strong-scaling job ("1250 at G = 8 no longer fills the
    # chip -- report that knee"): the same pipelined stream with every batch cut to nq/G queries, on this one GPU.
    small_batch = pq_scan = None
    if not args.no_secondary and rk.world == 1:
        small_batch = {}
        for g in (2, 4, 8):
            n_g = nq // g
            if n_g < 1:
                continue
            srcs = [a[:n_g] for a in qb]
            depth_g = tickets_in_flight(n_g)
            n_sub = max(40, launches // 2) * g // 2          # (about the same number of queries for every slice size)
            run_pipelined(3 * depth_g, srcs, depth_g)
            s0 = ix.pipeline_stats()
            el_g, last_g = run_pipelined(n_sub, srcs, depth_g)
            ix.batch_sync()
            s1 = ix.pipeline_stats()
            lbg = (n_sub - 1) % nb
            if not (np.array_equal(last_g[0], ids[lbg * nq:lbg * nq + n_g]) and
                    np.array_equal(last_g[1].view(np.uint32), dist_out[lbg * nq:lbg * nq + n_g].view(np.uint32))):
                raise RuntimeError("a coalesced %d-query submit and the resident path disagree" % n_g)
            small_batch["%d_queries_per_batch" % n_g] = {
                "qps": n_g * n_sub / el_g, "as_gpus_of_a_strong_scaling_job": g, "tickets_in_flight": depth_g,
                "submits_per_launch": (s1["tickets"] - s0["tickets"]) / max(1, s1["launches"] - s0["launches"]),
                "queries_per_launch": (s1["queries"] - s0["queries"]) / max(1, s1["launches"] - s0["launches"]),
                "kernel_ms_per_launch": ix.timing()["search_kernel_ms"], "variant": ix.timing()["variant"]}
        pq_scan = isolated_pq_scan(device, n_codes=args.pq_scan_codes)

    out = {
        "metric": "QPS @ recall@10>=0.95, SIFT1M-shaped d=128 L2, batch=10k",
        "value": value, "unit": "queries/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed_job / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": ("f32 (lossless u8 rows+queries, v_dot4: integer arithmetic that equals the reference's float32 sums bit for bit)"
                  if variant == 13 else "f32 (lossless u8 rows)" if variant == 11 else "f32"),
        "data": "synthetic",
        "config": {"workload": "SIFT1M-shaped synthetic (configs[1]): N=%d d=%d L2, R=%d, L_search=%d, PQ m=%d, beam_width=%s, "
                               "k=%d, batch=%d queries, mode=M1 reference-faithful; a step = %d consecutive batches, %d distinct "
                               "batches per GPU rotating; value = host memory -> host memory (dr_search_submit/wait, one submit per batch, %d batches in flight; "
                               "the library runs the batches that wait for the search stream as one launch: %.2f batches = %.0f queries per launch)"
                               % (args.n, D, args.R, args.L, args.m, args.bw or None, k, nq_job, args.bps, nb, depth_head, submits_per_launch, q_per_launch)
                               + ("; STRONG scaling: every batch is cut into %d contiguous slices, one per GPU (slice of rank 0: %d queries)" % (rk.world, nq) if strong else ""),
                   "recall_at_10": recall, "build_seconds": build_s,
                   "parallelism": ("query-sharded replicas x%d, one batch split over the ranks" if strong else "query-sharded replicas x%d") % rk.world,
                   "rank_cpu_affinity": pinned_to if rk.world > 1 else None,
                   "per_rank_setup": setup_all,
                   "per_rank_slice": slices if strong else None,
                   "tickets_in_flight": depth_head, "coalesce_cap_queries": coalesce, "queries_per_launch": q_per_launch, "submits_per_launch": submits_per_launch,
                   "ms_a_batch_spends_in_the_stream": depth_head * nq / (value / rk.world) * 1e3 if value else None,
                   "kernel_launches_in_timed_region": n_kernel_launches, "kernel_ms_per_batch": k_ms / per_launch,
                   "one_launch_per_batch": one_per_launch,
                   "ms_per_batch": elapsed_job / launches * 1e3, "timed_region_s": elapsed_job, "per_rank_seconds": times,
                   "per_rank_qps": [(sl[1] - sl[0]) * launches / t for sl, t in zip(slices, times)],
                   "qps_resident": qps_resident, "ms_per_batch_resident": (max(res_times) / launches * 1e3) if res_times else None,
                   "kernel_ms_resident": tm_res["search_kernel_ms"],
                   "qps_pcie_inclusive_pageable_source": qps_pageable,
                   "qps_blocking_call_median": None if args.headline_only else qps_one_call,       # (scalar twin of the dict below: SURVEY 8d's literal metric)
                   "one_query_call_p50_ms_api_default_k5_L20": one_query_ms.get("api_default_k5_L20"),
                   "one_query_call_p50_ms_bench_point": one_query_ms.get("bench_point"),
                   "qps_resident_api_default_k5_L20": qps_api_default,
                   "float32_rows_qps_resident": float_rows["qps_resident"] if float_rows else None,
                   "float32_rows_kernel_ms": float_rows["kernel_ms"] if float_rows else None,
                   "float32_rows_roofline_frac": float_rows["roofline_frac"] if float_rows else None,
                   "rows": args.rows,
                   "qps_blocking_call": None if args.headline_only else {"median": qps_one_call, "calls": len(call_s), "best": nq / call_s[0], "worst": nq / call_s[-1],
                                         "ms_of_the_median_call": blocking_pieces,
                                         "note": "SURVEY.md 8d's literal metric: nq / wall time of one blocking dr_search_batch call "
                                                 "(pageable source; upload + search + tie order + download, nothing overlapped)"},
                   "per_query": {"expansions": float(st["steps"].mean()), "pq_distances": float(st["pq"].mean()),
                                 "pq_evaluated": float(st["pq_evaluated"].mean()), "exact_distances": float(st["exact"].mean()),
                                 "algorithmic_bytes": float(per_q.mean())},
                   "launch": {k_: tm_head[k_] for k_ in ("variant", "grid", "block", "lds_bytes", "waves_per_cu")},
                   "finalize_kernel_ms": tm_head["finalize_kernel_ms"],
                   # queries whose returned top-k holds equal distances: what the tie-order pass (finalize_kernel) replays per batch
                   "tied_queries_per_batch": float((dist_out[:, 1:] == dist_out[:, :-1]).any(axis=1).sum()) / nb,
                   "secondary_no_trim": secondary,
                   "row_storage": ("u8: lossless byte copy of the integer-valued vectors (every component checked; distances "
                                   "bit-identical); roofline.achieved counts D bytes per scored vector (what a byte-row kernel has to move), "
                                   "roofline.traffic is what HBM really moved") if byte_rows else "f32",
                   "query_storage": ("u8: every component of the batch is an integer in [0, 255] (checked per batch on the host)"
                                     if variant == 13 else "f32"),
                   "float32_rows": float_rows, "byte_rows_float32_queries": float_queries,
                   "small_batches_on_one_gpu": small_batch, "strong_scaling": strong_cfg, "pq_scan": pq_scan},
        "roofline": {"bound": "hbm", "kernel": "search_kernel<128,M1> variant %d" % variant_launched, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "hbm_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                     "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_kernel, "queries_per_launch": q_per_launch,
                     "row_bytes": D if byte_rows else 4 * D,
                     "note": "frac = the kernel's own algorithmic bytes (D bytes per scored vector on byte rows) / kernel time / peak; "
                             "hbm_frac is what the PMC counters say HBM moved; the kernel is bound by the chip's rate of random "
                             "requests (DESIGN.md 4.1: ~55 G requests/s measured by tools/gather_probe.hip), not by bytes"},
    }

    # ---------------------------------------------------------------- the un-rounded generator: its own data, graph, recall
    adj = ix.get_adjacency() if want_cpu else None
    if not args.no_secondary and rk.world == 1:
        t0 = time.time()
        xu, qu = sift_like(args.n, D, n_queries=nq * min(nb, 4), n_clusters=1024, seed=2024, query_seed=9000, rounded=False)
        iu = HipIndex.create_empty(xu, R=args.R, device=device)
        _, bsu = iu.build_vamana(L_build=args.L_build, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
        iu.pq_encode(iu.pq_train(args.m, n_sample=100_000, iters=8))
        gtu, _ = iu.bruteforce_topk(qu, k)
        nbu = min(nb, 4)
        for b in range(nbu):
            iu.batch_select(b); iu.batch_upload(qu[b * nq:(b + 1) * nq])
        n_sec = max(40, launches // 4)
        for i in range(4):
            iu.batch_select(i % nbu); iu.batch_run(k, L=args.L, beam_width=args.bw, mode=mode)
        iu.batch_sync()
        t1 = time.perf_counter()
        for i in range(n_sec):
            iu.batch_select(i % nbu); iu.batch_run(k, L=args.L, beam_width=args.bw, mode=mode)
        iu.batch_sync()
        elu = time.perf_counter() - t1
        tmu = iu.timing()
        outs = []
        for b in range(nbu):
            iu.batch_select(b); iu.batch_run(k, L=args.L, beam_width=args.bw, mode=mode); outs.append(iu.batch_download())
        idu = np.concatenate([o[0] for o in outs]); stu = np.concatenate([o[3] for o in outs])
        au, _ = alg_bytes(stu, D, args.R, args.m, k)
        au = (au - 4 * 256 * D) / nbu + 4 * 256 * D
        unrounded = {"data": "the same mixture without rounding (float32-valued descriptors: no byte rows)", "variant": tmu["variant"],
                     "qps_resident": nq * n_sec / elu, "recall_at_10": recall_at_k(idu, gtu, k), "kernel_ms": tmu["search_kernel_ms"],
                     "roofline_frac": au / (tmu["search_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "build_seconds": bsu,
                     "exact_distances_per_query": float(stu["exact"].mean()), "setup_seconds": time.time() - t0}
        out["config"]["unrounded_data"] = unrounded
        iu.close()

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1): the oracle, timed
    if want_cpu:
        from oracle import pyoracle as orc
        cores = os.cpu_count() or 1
        ns = min(args.cpu_sample, nq)
        q0 = np.array(qb[0])
        t2 = time.perf_counter()
        oids, odist, ocnt, ost = orc.search_batch(x, adj, q0[:ns], medoid, orc.M1, k, L=args.L, bw=args.bw,
                                                  codes=codes, codebook=cb, nthreads=cores)
        cpu_s = time.perf_counter() - t2
        same = bool(np.array_equal(oids, ids[:ns]) and
                    np.array_equal(odist.astype(np.float32).view(np.uint32), dist_out[:ns].view(np.uint32)))
        out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": "first %d of the bench queries, same index, oracle/ C restatement of M1 on OpenMP "
                                         "threads; GPU results bit-identical on the sample: %s" % (ns, same)}
        n1 = min(args.cpu_sample_1t, ns)
        t2 = time.perf_counter()
        orc.search_batch(x, adj, q0[:n1], medoid, orc.M1, k, L=args.L, bw=args.bw, codes=codes, codebook=cb, nthreads=1)
        out["cpu_baseline_1t"] = {"value": n1 / (time.perf_counter() - t2), "unit": "queries/s", "cores": 1, "kind": "port",
                                  "sample": "first %d of the bench queries, one thread" % n1}
        if not same:
            raise RuntimeError("GPU results differ from the oracle on the CPU-baseline sample")
    if rk.rank == 0:
        print(json.dumps(out), flush=True)
    ix.close()
    return 0



# ------------------------------------------------------------------------------------------------ c3 / c4: the other query-sharded shapes
def worker_shape(args, rk):
    """BASELINE configs[2] (c3: 10M x 1536 inner product on unit vectors = L2, PQ + full-precision rerank) and configs[3] (c4: DEEP100M-shaped
    100M x 96, index replicated, queries sharded) at --num-vectors points, index built by the engine's own builder.
    Both: DR_MODE_PQB | DR_F_RERANK -- the PQ traversal (round 5's batch-per-step kernel) + exact rerank of the L list (SURVEY.md 8d's definition
    of c3) -- at the recall-0.95 point of the FULL-SIZE index: c3 L = 250 without frontier trim (1.09 M QPS resident / 1.21 M as a stream at 10M points,
    profiles/r05/op_c3_10M_d1536_pqb.jsonl), c4 L = 400, beam_width 32 (1.75 M / 2.13 M at 100M points, op_c4_100M_d96_pqb.jsonl). c4 also runs the
    reference-faithful M1 (_pq_accelerated_graph_search, search_engine.py:398-506) at ITS recall-0.95 point, L = 500, beam_width 64 (862 k / 1.00 M at
    100M points): `config.m1_reference_faithful`.
    value = host memory -> host memory stream of --num-queries batches (dr_search_submit / dr_search_wait, shared launches), every rank its own
    replica (weak scaling, no collective). roofline = the traversal kernel's algorithmic bytes (SURVEY.md 8d's B_q from the engine's per-query
    counters) / its mean launch duration over the timed region (HIP events on the search stream)."""
    import diskrag_amd
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.synth import recall_at_k, unit_mixture, unit_mixture_parallel
    ndev = diskrag_amd.device_count()
    if ndev < 1:
        raise RuntimeError("no HIP device: the engine has no CPU fallback")
    device = rk.local_rank % ndev
    pin_to_gpu_socket(rk, device)
    c3 = args.config == "c3"
    nq, k, D, m, R, n = args.nq, args.k, args.dim, args.m, args.R, args.n
    nb = max(1, min(args.nb, 4))
    t0 = time.time()
    gen = unit_mixture_parallel if n * D >= (1 << 32) else unit_mixture
    x, q_all = gen(n, D, n_queries=nq * nb, n_clusters=4096, seed=11, latent=64 if c3 else 32)          # (the replicas hold the same index; every rank streams the same batches)
    log(rk, f"synthetic unit-norm mixture {x.shape} + {nb} batches of {nq} queries in {time.time() - t0:.1f}s")
    ix = HipIndex.create_empty(x, R=R, device=device)
    medoid, build_s = ix.build_vamana(L_build=args.L_build, alpha=1.2, passes=2, seed=7)
    t0 = time.time()
    want_cpu = rk.rank == 0 and rk.world == 1 and not args.no_cpu
    cb = ix.pq_train(m, n_sample=100_000, iters=8)
    codes = ix.pq_encode(cb, want_codes=want_cpu)
    pq_s = time.time() - t0
    gt, _ = ix.bruteforce_topk(q_all[:nq * min(nb, 2)], k)
    log(rk, f"graph in {build_s:.1f}s, PQ m={m} in {pq_s:.1f}s, ground truth done")
    if not want_cpu:
        del x
    # c4's 16-byte code words cost a 128-byte line each when gathered (6.2x traffic, profiles/r06/pmc_pqb_c4.json): beside the adjacency row they are
    # one contiguous read per expansion (dr_index_inline_codes: N R m bytes -- 102 GB at 100M; same results, +6.5 % at 100M in round 5)
    # (at 4M points the two forms measure the same: 6.65 against 6.67 M QPS, profiles/r06 -- the footprint only pays where the gathers leave the caches)
    inline = (not c3) and n >= 50_000_000 and n * R * m <= 120e9 and not os.environ.get("DR_BENCH_NO_INLINE")
    if inline:
        ix.inline_codes(True)
    qn = np.linalg.norm(q_all[:64].astype(np.float64), axis=1)
    ip = _ffi.F_IP if c3 and np.abs(qn * qn - 1.0).max() < 5e-4 else 0       # c3 is named an inner-product config: unit-norm rows and queries
    top = int(args.rerank_top or 0)
    kw = dict(L=args.L, beam_width=args.bw, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK | ip | _ffi.F_RERANK_TOP(top if top < args.L else 0))
    # the reference-faithful M1 (_pq_accelerated_graph_search, search_engine.py:398-506) beside it, at ITS recall-0.95 point of the full-size index
    # (c4 100M: L = 500, beam_width 64; c3 10M: L = 200, beam_width 64 -- profiles/r04/op_c4_100M_d96_m1fine.jsonl, op_c3_10M_d1536_bwsweep.jsonl)
    kw2 = dict(L=500, beam_width=64, mode=_ffi.MODE_M1) if not c3 else dict(L=200, beam_width=64, mode=_ffi.MODE_M1)
    if args.no_secondary:
        kw2 = None
    qb = []
    for b in range(nb):
        a = _ffi.pinned_empty((nq, D), np.float32)
        a[:] = q_all[b * nq:(b + 1) * nq]
        qb.append(a)
    ix.set_coalesce(args.coalesce)          # (launches of up to 65 536 queries: worker_c2 has the measurement)
    depth = min(_ffi.MAX_TICKETS - 2, _ffi.PIPE_DEPTH * max(1, args.coalesce // nq) + 2)

    def stream(n_sub, kw_):
        jobs, done, last = [], 0, None
        t1 = time.perf_counter()
        for i in range(n_sub):
            jobs.append(ix.search_submit(qb[i % nb], k, reuse_outputs=True, **kw_))
            if len(jobs) - done >= depth:
                last = jobs[done].wait(); jobs[done] = None; done += 1
        for j in range(done, len(jobs)):
            last = jobs[j].wait()
        return time.perf_counter() - t1, last

    # a step = --batches-per-step batches (default here: 4: these launches take 5-25 ms each at full size)
    bps = args.bps if "--batches-per-step" in sys.argv else 4
    n_sub = args.steps * bps
    stream(max(2 * depth, args.warmup * bps), kw); ix.batch_sync()
    rk.barrier()
    ps0 = ix.pipeline_stats()
    el, last = stream(n_sub, kw)
    ix.batch_sync()
    tm = ix.timing()
    ps1 = ix.pipeline_stats()
    times = rk.gather("t_shape", el)
    nl = max(1, ps1["launches"] - ps0["launches"])
    qpl = (ps1["queries"] - ps0["queries"]) / nl
    # counters, recall: one blocking call per ground-truth batch
    outs = [ix.search_batch(qb[b], k, **kw) for b in range(min(nb, 2))]
    ids = np.concatenate([o[0] for o in outs]); st = np.concatenate([o[3] for o in outs]); dist = np.concatenate([o[1] for o in outs])
    if (st["status"] != 0).any():
        raise RuntimeError("work-area overflow during the bench")
    lbq = (n_sub - 1) % nb
    if lbq < len(outs) and not np.array_equal(last[0], outs[lbq][0]):
        raise RuntimeError("the pipelined path and a blocking call disagree")
    recall = recall_at_k(ids, gt, k)
    recalls = rk.gather("recall", recall)
    if min(recalls) < args.min_recall:
        raise RuntimeError(f"recall@{k} = {min(recalls):.4f} is below the metric's bar {args.min_recall}")
    # algorithmic bytes per query: query + adjacency rows + code words scored + full-precision rows scored + output
    S, V, X = st["steps"].astype(np.float64), st["pq_evaluated"].astype(np.float64), st["exact"].astype(np.float64)
    per_q = 4.0 * D + S * 4.0 * R + V * float(m) + X * 4.0 * D + 8.0 * k
    k_ms = float(tm["search_kernel_ms"])
    alg_launch = float(per_q.mean()) * qpl + 4.0 * 256 * D
    # the rerank pass of c3 runs in its own kernel: its rows are not the traversal kernel's
    alg_search_launch = float((per_q - X * 4.0 * D).mean()) * qpl + 4.0 * 256 * D
    achieved = alg_search_launch / (k_ms * 1e-3) / 1e9
    # one resident launch per batch, nothing overlapped: where a batch's time goes (traversal kernel, table kernel, rerank pass + the rest)
    ix.batch_select(15); ix.batch_upload(qb[0])
    ix.batch_run(k, **kw); ix.batch_sync()
    t1 = time.perf_counter()
    for _ in range(3):
        ix.batch_run(k, **kw)
    ix.batch_sync()
    res_ms = (time.perf_counter() - t1) / 3 * 1e3
    tmr = ix.timing()
    resident_pieces = {"ms_per_batch": res_ms, "traversal_kernel": float(tmr["search_kernel_ms"]), "table_kernel": float(tmr["lut_kernel_ms"]),
                       "rerank_pass_and_rest": res_ms - float(tmr["search_kernel_ms"]) - float(tmr["lut_kernel_ms"]), "qps_resident": nq / (res_ms * 1e-3),
                       "rows_reranked_per_query": float(X.mean())}
    ix.batch_select(0)
    # HBM traffic of the traversal kernel from the PMC counters, collected offline on this workload (scripts/profile_run_r06.sh, separate --pmc passes;
    # one 10 000-query batch per launch there: scaled by this run's queries per launch)
    traffic, traffic_src = None, None
    pmc = ROOT / "profiles" / "r06" / ("pmc_pqb_%s.json" % args.config)
    if pmc.exists():
        rec = json.loads(pmc.read_text())
        if (rec.get("N"), rec.get("L"), rec.get("bw")) == (n, args.L, args.bw) and rec.get("hbm_bytes_per_launch"):
            traffic = rec["hbm_bytes_per_launch"] * qpl / 10000.0
            traffic_src = "profiles/r06/%s: rocprofv3 --pmc TCC_EA0_RDREQ by size + WRITE_SIZE over scripts/pmc_target_shape.py (same index recipe, 10 000 queries per launch), scaled to this run's queries per launch" % pmc.name
    m1_ref = None
    if kw2 is not None:
        stream(2 * depth, kw2); ix.batch_sync()
        el2, _ = stream(max(8, n_sub // 2), kw2); ix.batch_sync()
        o2 = [ix.search_batch(qb[b], k, **kw2) for b in range(min(nb, 2))]
        m1_ref = {"what": "DR_MODE_M1 L=%d beam_width=%d: the reference-faithful _pq_accelerated_graph_search at ITS recall-0.95 point of the full-size index" % (kw2["L"], kw2["beam_width"]),
                  "qps": nq * max(8, n_sub // 2) / el2, "recall_at_10": recall_at_k(np.concatenate([o[0] for o in o2]), gt, k),
                  "kernel_ms_per_launch": float(ix.timing()["search_kernel_ms"]), "variant": ix.timing()["variant"]}
    out = {"metric": "QPS @ recall@10>=0.95, %s, PQ traversal + full-precision rerank, batch=%d" % ("d=1536 unit-norm (inner product = L2)" if c3
                                                                                                  else "DEEP-shaped d=96 L2, index replicated", nq),
           "value": nq * n_sub * rk.world / max(times), "unit": "queries/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": max(times) / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (ADC sums over u8 codes; float32 rerank rows)", "data": "synthetic",
           "config": {"workload": "%s-shaped synthetic (BASELINE configs[%d]): N=%d d=%d unit-norm mixture (4096 clusters), R=%d, PQ m=%d, %s L=%d beam_width=%s, k=%d, "
                                  "batch=%d queries; a step = %d batches; host memory -> host memory (dr_search_submit/wait, %d submits in flight, %.0f queries per launch); "
                                  "index built on the device (dr_build_vamana L_build=%d); full size = %d points"
                                  % (args.config, 2 if c3 else 3, n, D, R, m, "DR_MODE_PQB | DR_F_RERANK" + (" | DR_F_IP" if ip else "") + (" | DR_F_RERANK_TOP(%d)" % top if 0 < top < args.L else ""), args.L, args.bw or None, k, nq, bps, depth, qpl,
                                     args.L_build, 10_000_000 if c3 else 100_000_000),
                      "recall_at_10": float(np.mean(recalls)), "build_seconds": build_s, "pq_seconds": pq_s, "parallelism": "query-sharded replicas x%d" % rk.world,
                      "per_rank_seconds": times, "queries_per_launch": qpl, "kernel_ms_per_launch": k_ms, "kernel_ms_per_10k_queries": k_ms * 10000.0 / qpl,
                      "table_kernel_ms_per_launch": float(tm["lut_kernel_ms"]),
                      "per_query": {"expansions": float(S.mean()), "code_words_scored": float(V.mean()), "full_precision_rows_scored": float(X.mean()),
                                    "algorithmic_bytes": float(per_q.mean())},
                      "launch": {k_: tm[k_] for k_ in ("variant", "grid", "block", "lds_bytes", "waves_per_cu")},
                      "rerank_top": top if 0 < top < args.L else args.L, "one_resident_batch_ms": resident_pieces, "inline_neighbour_codes": bool(inline),
                      "m1_reference_faithful": m1_ref},
           "roofline": {"bound": "hbm", "kernel": "pqb_search_kernel (DR_MODE_PQB traversal; the rerank pass is its own kernel)",
                        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                        "hbm_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                        "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_search_launch, "queries_per_launch": qpl,
                        "whole_call_algorithmic_bytes_per_launch": alg_launch,
                        "note": "algorithmic bytes = SURVEY 8d's B_q from the engine's per-query counters x the queries of a launch; these traversals are bound by "
                                "instruction issue and dependent LDS / memory round trips at 8-12 wavefronts per CU, not by HBM bytes (DESIGN.md 4.5)"}}
    if want_cpu:
        from oracle import pyoracle as orc
        cores = os.cpu_count() or 1
        adj = ix.get_adjacency()
        ns = min(args.cpu_sample, nq)
        q0 = np.array(qb[0][:ns])
        omode, oflags = orc.PQB, orc.F_RERANK | (orc.F_IP if ip else 0) | orc.F_RERANK_TOP(top if top < args.L else 0)
        t2 = time.perf_counter()
        oids, odist, ocnt, ost = orc.search_batch(x, adj, q0, medoid, omode, k, L=args.L, bw=args.bw, flags=oflags, codes=codes, codebook=cb, nthreads=cores)
        cpu_s = time.perf_counter() - t2
        same = bool(np.array_equal(oids, outs[0][0][:ns]) and np.array_equal(odist.astype(np.float32).view(np.uint32), outs[0][1][:ns].view(np.uint32)))
        out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": "first %d of the bench queries, same index, oracle/ C restatement of the same mode on OpenMP threads; GPU results "
                                         "bit-identical on the sample: %s" % (ns, same)}
        if not same:
            raise RuntimeError("GPU results differ from the oracle on the CPU-baseline sample")
    if rk.rank == 0:
        print(json.dumps(out), flush=True)
    ix.close()
    return 0

# ------------------------------------------------------------------------------------------------ c5: graph-sharded
def worker_c5(args, rk):
    """BASELINE config c5 (1B x 1536, PQ-only, graph sharded 8 ways) at bench scale, through the pipeline of the full-size run
    (scripts/c5_shard.py): every rank owns ONE shard of --num-vectors points of the 1536-d unit-mixture
    stream -- generated chunk by chunk, encoded on the device (dr_pq_encode_rows) and forgotten: the vectors are never stored --,
    builds its Vamana sub-graph from the code words alone (dr_build_vamana_pq, R = 128), and every query runs on every shard
    (DR_MODE_PQ | DR_F_NO_VISITED_SET, L = 100, beam_width 32: the full-size shard's operating point, 1.46 M QPS at recall 0.961 vs the
    ADC ranking, profiles/r04/scale_c5_shard_R128_no_visited_set.json); the per-shard top-k lists travel as packed keys in ONE
    RCCL all-gather and are merged on the device (dr_sharded_submit / dr_sharded_wait, two batches in flight). Recall is
    against the brute-force ADC ranking of the union (dr_pq_scan_topk per shard, merged): the metric a PQ-only index can be
    held to."""
    import diskrag_amd
    from diskrag_amd import HipIndex, _ffi
    from diskrag_amd.parallel import merge_topk
    from diskrag_amd.synth import UnitMixtureStream, recall_at_k
    ndev = diskrag_amd.device_count()
    if ndev < 1:
        raise RuntimeError("no HIP device: the engine has no CPU fallback")
    device = rk.local_rank % ndev
    pin_to_gpu_socket(rk, device)
    nq, k, D, m, R = args.nq, args.k, args.dim, args.m, args.R
    n_s = args.n
    blk = UnitMixtureStream.BLOCK
    stride = -(-n_s // blk) * blk                       # every shard starts on a block of the stream
    gen = UnitMixtureStream(d=D, n_clusters=4096, seed=11, latent=64, threads=min(32, os.cpu_count() or 1))
    t0 = time.time()
    # one codebook for the whole index: every rank trains on the SAME sample with the same seed (deterministic on the device)
    sample = gen.draw(0, min(262144, stride))
    tmp = HipIndex.create_empty(sample, R=R, device=device)
    cb, _ = tmp.pq_train_ex(m, n_sample=50000, max_iter=15, n_init=1, seed=5)
    tmp.close()
    del sample
    cb_s = time.time() - t0
    q = gen.draw(0, nq, stream=1)                        # the same queries on every rank
    sh = HipIndex.create_codes_empty(n_s, D, R, cb, device=device)
    t0 = time.time()
    ch = 8 * blk
    ngt = min(nq, 1000)
    ex_ids = ex_dist = None          # exact top-k of this shard for the first ngt queries, accumulated chunk by chunk (the vectors are not kept)
    ex_s = 0.0
    for r0 in range(0, n_s, ch):
        rows = min(ch, n_s - r0)
        x = gen.draw(rk.rank * stride + r0, rows)
        sh.encode_rows(x, r0)
        t1 = time.time()
        part = HipIndex.create_empty(x, R=1, device=device)
        ci, cd = part.bruteforce_topk(q[:ngt], k)
        part.close()
        ci = (ci.astype(np.int64) + rk.rank * n_s + r0).astype(np.uint32)
        ex_ids, ex_dist = (ci, cd) if ex_ids is None else merge_topk([ex_ids, ci], [ex_dist, cd], k)
        ex_s += time.time() - t1
        del x
    enc_s = time.time() - t0 - ex_s
    medoid, build_s = sh.build_vamana_pq(L_build=args.L_build, alpha=1.2, passes=2, seed=7)
    log(rk, f"shard of {n_s} x {D} encoded in {enc_s:.1f}s (codebook {cb_s:.1f}s), graph R={R} built from code words in {build_s:.1f}s")
    base = rk.rank * n_s
    # ground truth in the index's own metric: brute-force ADC top-k of every shard, merged over the ranks
    g_ids, g_sq, _ = sh.pq_scan_topk(q[:ngt], k)
    gts = rk.gather("gt", {"ids": (g_ids.astype(np.int64) + base).tolist(), "dist": g_sq.tolist()})
    gt, _ = merge_topk([np.array(g["ids"], dtype=np.uint32) for g in gts], [np.array(g["dist"], dtype=np.float32) for g in gts], k)
    # ... and against the EXACT neighbours (a PQ-only index cannot rerank: this is bounded by the quantiser)
    exs = rk.gather("gt_exact", {"ids": ex_ids.tolist(), "dist": ex_dist.tolist()})
    gt_exact, _ = merge_topk([np.array(g["ids"], dtype=np.uint32) for g in exs], [np.array(g["dist"], dtype=np.float32) for g in exs], k)
    # communicator: rank 0 makes the id, the others read it from the scratch directory
    if rk.rank == 0:
        rk.put("rccl_id", _ffi.Comm.unique_id())
    uid = rk.get("rccl_id", 0, raw=True)
    comm = _ffi.Comm(uid, rk.world, rk.rank, device)
    qp = _ffi.pinned_empty((nq, D), np.float32)
    qp[:] = q

    if args.c5_mode == "pqb":
        c5_mode, c5_flags = _ffi.MODE_PQB, 0              # round 5: one third of DR_MODE_PQ's instruction stream (pops: the rows that fill 64 slots)
    else:
        c5_mode, c5_flags = _ffi.MODE_PQ, _ffi.F_NO_VISITED_SET        # (same results as with a visited set; 24-33 % faster at the full shard size)

    # Exchanges of --c5-group submits (a count, never a timing: every rank forms the same exchanges), three exchanges' worth of submits in
    # flight: a 10k-query launch of this kernel is 4.9 queries per wavefront slot and ends in a tail of idle slots
    # (profiles/r04/scale_c5_shard_R128_stream.json: 1.44 -> 1.76 M QPS on the full-size shard with 26.7 k queries per launch)
    grp = max(1, min(16, args.c5_group, 65536 // max(1, nq)))
    _ffi.sharded_set_group(sh, grp)
    depth_c5 = 2 if grp == 1 else 3 * grp

    def stream(n_calls):
        jobs, out, ms = [], None, np.zeros(3)
        t1 = time.perf_counter()
        for i in range(n_calls):
            jobs.append(_ffi.sharded_submit([sh], [base], qp, k, L=args.L, beam_width=args.bw, mode=c5_mode, flags=c5_flags, comm=comm))
            if len(jobs) >= depth_c5:
                out = jobs.pop(0).wait(); ms += out[3]
        for j in jobs:
            out = j.wait(); ms += out[3]
        return time.perf_counter() - t1, out, ms

    stream(max(2 * depth_c5, args.warmup))
    rk.barrier()
    el, (ids, dist, status, _), ms = stream(args.steps)
    group_ab = None
    if os.environ.get("DR_BENCH_C5_GROUPS"):      # A/B of the exchange size in ONE process (one index): "3,6,3,6" -> QPS of a stream of 2 * steps batches per entry
        group_ab = []
        for g in [int(v) for v in os.environ["DR_BENCH_C5_GROUPS"].split(",")]:
            _ffi.sharded_set_group(sh, g)
            depth_c5 = 2 if g == 1 else 3 * g
            stream(2 * depth_c5)
            rk.barrier()
            el_g = stream(2 * args.steps)[0]
            group_ab.append({"submits_per_exchange": g, "qps": nq * 2 * args.steps / max(rk.gather("t_c5_g%d" % len(group_ab), el_g))})
        depth_c5 = 2 if grp == 1 else 3 * grp
    _ffi.sharded_set_group(sh, 1)
    times = rk.gather("t_c5", el)
    if int(status.max()) != 0:
        raise RuntimeError("work-area overflow during the bench")
    # one blocking call for the per-phase times (nothing overlapped) and the search kernel of the shard
    _, _, _, ms1 = _ffi.sharded_search([sh], [base], qp, k, L=args.L, beam_width=args.bw, mode=c5_mode, flags=c5_flags, comm=comm)
    sh.batch_sync()                                     # (publishes the shard's kernel timings)
    tm = sh.timing()
    recall = recall_at_k(ids[:ngt], gt, k)
    out = {"metric": "QPS, graph-sharded PQ-only search (c5 layout at bench scale), batch=%d" % nq,
           "value": nq * args.steps / max(times), "unit": "queries/s", "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": max(times) / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (ADC sums over u8 codes)", "data": "synthetic",
           "config": {"workload": "c5 layout at bench scale: %d shards x %d points of the 1536-d unit-mixture stream (4096 clusters), vectors "
                                  "encoded on the fly and never stored, PQ m=%d, graph built from code words (dr_build_vamana_pq R=%d, L_build=%d), "
                                  "%s L=%d beam_width=%s, k=%d; every query on every shard; exchange = ONE RCCL all-gather of (nq*k + 1) "
                                  "packed 64-bit words per rank + device merge; a step = one %d-query batch (dr_sharded_submit/wait), %d submits per exchange "
                                  "(one launch per shard, one all-gather: dr_sharded_set_group), %d submits in flight"
                                  % (rk.world, n_s, m, R, args.L_build, "DR_MODE_PQB" if args.c5_mode == "pqb" else "DR_MODE_PQ | DR_F_NO_VISITED_SET", args.L, args.bw or None, k, nq, grp, depth_c5),
                      "recall_at_10_vs_bruteforce_adc": recall, "recall_at_10_vs_exact_neighbours": recall_at_k(ids[:ngt], gt_exact, k),
                      "adc_ranking_recall_at_10_vs_exact_neighbours": recall_at_k(gt, gt_exact, k),
                      "ground_truth_queries": ngt, "rccl_ranks": rk.world, "exact_ground_truth_seconds": ex_s,
                      "build_seconds": build_s, "encode_seconds": enc_s, "codebook_seconds": cb_s,
                      "per_rank_seconds": times, "exchange_bytes_per_rank_per_batch": (nq * k + 1) * 8, "submits_per_exchange": grp, "submits_in_flight": depth_c5, "exchange_size_ab": group_ab,
                      "one_blocking_call_ms": {"search": float(ms1[0]), "all_gather": float(ms1[1]), "merge": float(ms1[2])},
                      "search_kernel": {"variant": tm["variant"], "kernel_ms": tm["search_kernel_ms"], "table_kernel_ms": tm["lut_kernel_ms"],
                                        "waves_per_cu": tm["waves_per_cu"]}}}
    if recall < args.min_recall:
        raise RuntimeError(f"c5: recall@{k} vs the brute-force ADC ranking = {recall:.4f} is below {args.min_recall}")
    # What a USABLE PQ-only index costs: at m = 32 a 1536-d code word keeps recall@10 against the exact neighbours at the quantiser's
    # ceiling (0.2-0.5); the same shard with m = 64 (sub-vectors of 24 elements, 64-byte code words, a 64-KiB table per query: the kernel with 32
    # table rows in registers and 32 in LDS) -- one rank, the same stream, points regenerated from the stream (bench scale only)
    m64 = None
    if rk.world == 1 and not args.no_secondary and m == 32 and n_s <= 8_000_000:
        t0 = time.time()
        sample = gen.draw(0, min(262144, stride))
        tmp = HipIndex.create_empty(sample, R=R, device=device)
        cb64, _ = tmp.pq_train_ex(64, n_sample=50000, max_iter=15, n_init=1, seed=5)
        tmp.close()
        del sample
        sh64 = HipIndex.create_codes_empty(n_s, D, R, cb64, device=device)
        for r0 in range(0, n_s, ch):
            sh64.encode_rows(gen.draw(rk.rank * stride + r0, min(ch, n_s - r0)), r0)
        _, b64 = sh64.build_vamana_pq(L_build=args.L_build, alpha=1.2, passes=2, seed=7)
        g64, _, _ = sh64.pq_scan_topk(q[:ngt], k)
        sh64.batch_upload(q)
        pts = []
        for (L64, bw64) in ((args.L, args.bw), (150, 32), (200, 32)):
            kw64 = dict(L=L64, beam_width=bw64, mode=_ffi.MODE_PQB)
            sh64.batch_run(k, **kw64); sh64.batch_sync()
            t1 = time.perf_counter()
            for _ in range(3):
                sh64.batch_run(k, **kw64)
            sh64.batch_sync()
            dt = (time.perf_counter() - t1) / 3
            i64 = sh64.batch_download()[0]
            t64 = sh64.timing()
            pts.append({"L": L64, "beam_width": bw64, "qps_resident": nq / dt, "kernel_ms": t64["search_kernel_ms"], "table_kernel_ms": t64["lut_kernel_ms"],
                        "waves_per_cu": t64["waves_per_cu"], "recall_at_10_vs_bruteforce_adc": recall_at_k(i64[:ngt], g64.astype(np.uint32), k),
                        "recall_at_10_vs_exact_neighbours": recall_at_k((i64[:ngt].astype(np.int64) + base).astype(np.uint32), gt_exact, k)})
        m64 = {"m": 64, "code_bytes": 64, "setup_seconds": time.time() - t0, "graph_build_seconds": b64,
               "adc_ranking_recall_at_10_vs_exact_neighbours": recall_at_k((g64.astype(np.int64) + base).astype(np.uint32), gt_exact, k), "points": pts,
               "m32_same_launch_shape": {"note": "the m = 32 shard of this line, one resident 10k launch", "kernel_ms": tm["search_kernel_ms"], "waves_per_cu": tm["waves_per_cu"]}}
        sh64.close()
    out["config"]["m64"] = m64
    comm.close()
    sh.close()
    if rk.rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)       # (librccl prints a version banner through the C stdout: out before the one JSON line)
        print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse_args()
    in_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not args.worker and not in_torchrun and args.gpus > 1:
        return launch(args)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
