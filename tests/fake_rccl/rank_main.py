"""One rank of the multi-process sharded-search tests (tests/test_gpu_sharded_procs.py: 2 and 8 ranks). Started as a fresh child process BEFORE any GPU call;
DR_RCCL_LIB points at tests/fake_rccl/libfake_rccl.so (an all-gather over shared memory: RCCL refuses two ranks on one GPU).
usage: rank_main.py <scenario> <rank> <nranks> <scratch dir>   -> writes <scratch>/result.<rank>.json"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
scenario, rank, nranks, scratch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), Path(sys.argv[4])

from diskrag_amd import HipIndex, _ffi          # noqa: E402
from tests.conftest import load_golden           # noqa: E402


def put(name, data):
    tmp = scratch / (name + ".tmp%d" % rank)
    tmp.write_bytes(data)
    os.replace(tmp, scratch / name)


def get(name, timeout=60):
    t0 = time.time()
    while not (scratch / name).exists():
        if time.time() - t0 > timeout:
            raise RuntimeError("timed out waiting for " + name)
        time.sleep(0.01)
    return (scratch / name).read_bytes()


g = load_golden("unit1536_R16_m32")
N = len(g.vectors)
# every rank owns ONE shard: the fixture's graph under its own id base (the shards of a graph-sharded index are independent indexes)
mine = HipIndex.create_codes(g.adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes)
bases = [r * N for r in range(nranks)]
# what the exchange must return: the same shards, all local, merged without a communicator
local = [HipIndex.create_codes(g.adj, g.medoid, g.vectors.shape[1], g.codebook, g.codes) for _ in range(nranks)]
if rank == 0:
    put("rccl_id", _ffi.Comm.unique_id())
comm = _ffi.Comm(get("rccl_id"), nranks, rank, 0)
out = {"rank": rank, "checks": []}
q = g.queries
kw = dict(L=100, beam_width=16, mode=_ffi.MODE_PQB)


def same(a, b):
    return bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))


try:
    if scenario == "ok":
        want = _ffi.sharded_search(local, bases, q, 10, **kw)
        got = _ffi.sharded_search([mine], [bases[rank]], q, 10, comm=comm, **kw)
        out["checks"].append(["blocking call == all-local merge", same(got, want)])
        # exchanges of three submits, mixed sizes, seven submits over the ring of four, waited in reverse order
        _ffi.sharded_set_group(mine, 3)
        sizes = [len(q), 5, 9, 1, len(q), 7, 3]
        jobs = [_ffi.sharded_submit([mine], [bases[rank]], q[:n], 10, comm=comm, **kw) for n in sizes]
        _ffi.sharded_flush(mine)
        res = [j.wait() for j in reversed(jobs)][::-1]
        out["checks"].append(["grouped submits == blocking calls", all(same(r, (want[0][:n], want[1][:n])) for r, n in zip(res, sizes))])
        _ffi.sharded_set_group(mine, 1)
        # the reference-faithful PQ traversal through the same exchange
        kw3 = dict(L=10, beam_width=8, mode=_ffi.MODE_M3, flags=_ffi.F_USE_PQ)
        out["checks"].append(["M3 with PQ", same(_ffi.sharded_search([mine], [bases[rank]], q, 10, comm=comm, **kw3), _ffi.sharded_search(local, bases, q, 10, **kw3))])
    elif scenario == "local_failure":
        # rank 1's shard cannot serve the mode (no stored vectors for the rerank): its status word fails the call on EVERY rank, the next call works
        bad = dict(kw, flags=_ffi.F_RERANK) if rank == 1 else kw
        try:
            _ffi.sharded_search([mine], [bases[rank]], q, 10, comm=comm, **bad)
            out["checks"].append(["failing call raised", False])
        except _ffi.DiskragHipError as e:
            out["checks"].append(["failing call raised", True])
            out["code"] = e.code
        want = _ffi.sharded_search(local, bases, q, 10, **kw)
        got = _ffi.sharded_search([mine], [bases[rank]], q, 10, comm=comm, **kw)
        out["checks"].append(["the next exchange works", same(got, want)])
    elif scenario == "exchange_failure":
        # rank 1's all-gather itself fails (fake_rccl injects it): rank 1 aborts its communicator and reports the error; rank 0's all-gather never
        # completes -- its bounded wait (DR_EXCHANGE_TIMEOUT_MS) aborts and answers DR_E_REMOTE instead of hanging forever
        t0 = time.time()
        try:
            _ffi.sharded_search([mine], [bases[rank]], q, 10, comm=comm, **kw)
            out["checks"].append(["failing exchange raised", False])
        except _ffi.DiskragHipError as e:
            out["checks"].append(["failing exchange raised", True])
            out["code"] = e.code
        out["seconds"] = time.time() - t0
        try:        # the communicator is dead on both ranks: refused at once, no hang
            _ffi.sharded_search([mine], [bases[rank]], q, 10, comm=comm, **kw)
            out["checks"].append(["dead communicator refused", False])
        except _ffi.DiskragHipError:
            out["checks"].append(["dead communicator refused", True])
        # ... and the shard itself is still usable
        a = mine.search_batch(q, 10, **kw)
        b = local[0].search_batch(q, 10, **kw)
        out["checks"].append(["shard usable afterwards", bool(np.array_equal(a[0], b[0]))])
except Exception as e:      # noqa: BLE001
    out["error"] = "%s: %s" % (type(e).__name__, e)
put("result.%d.json" % rank, json.dumps(out).encode())
