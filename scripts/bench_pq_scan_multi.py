"""The flat PQ scan with several queries per pass (GPU box): pq_scan_multi_kernel (entry_points.inc: tables interleaved in LDS as T[j][c][q],
one 16-byte lookup per code byte for 4 queries, the code stream read once per group) beside pq_scan_kernel (one block row per query), same
table of random code words (far larger than L2 + Infinity Cache), same queries. Rate = nq * N * m bytes / kernel time (SURVEY 8d's algorithmic
bytes: every query scans the whole table); stream = the bytes the multi kernel actually reads from HBM per second.
Usage: python scripts/bench_pq_scan_multi.py [N=64000000] [m=32] [D=128]  -> one JSON object on stdout."""
import json
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rs = np.random.default_rng(5)
codes = rs.integers(0, 256, size=(N, m), dtype=np.uint8)
cb = rs.standard_normal((m, 256, D // m), dtype=np.float32)
qs = rs.standard_normal((16, D), dtype=np.float32)
ix = HipIndex.create_codes(np.zeros((N, 1), dtype=np.uint32), 0, D, cb, codes)
del codes
out = {"N": N, "m": m, "D": D, "code_bytes_per_query": N * m, "rows": []}
for nq in (1, 2, 4, 8, 16):
    NQ = 4 if (m <= 32 and nq > 2) else 2
    row = {"queries_per_launch": nq, "queries_per_pass": NQ if nq >= 2 else 1}
    res = {}
    for mode in ("per_query", "shared_pass"):
        if mode == "per_query":
            os.environ["DR_PQ_SCAN_PER_QUERY"] = "1"
        else:
            os.environ.pop("DR_PQ_SCAN_PER_QUERY", None)
        ix.pq_scan_best(qs[:nq])
        runs = [ix.pq_scan_best(qs[:nq]) for _ in range(5)]
        ms = sorted(r[2] for r in runs)[2]
        res[mode] = (runs[0][0].tolist(), runs[0][1].tobytes())
        row[mode] = {"kernel_ms_median": ms, "GBps_algorithmic": nq * N * m / (ms * 1e-3) / 1e9, "frac_of_8TBps": nq * N * m / (ms * 1e-3) / 8e12,
                     "ms_per_query": ms / nq}
        if mode == "shared_pass" and nq >= 2:
            row[mode]["code_stream_GBps"] = -(-nq // NQ) * N * m / (ms * 1e-3) / 1e9
    row["same_winners_bit_for_bit"] = res["per_query"] == res["shared_pass"]
    row["speedup"] = row["per_query"]["kernel_ms_median"] / row["shared_pass"]["kernel_ms_median"]
    out["rows"].append(row)
print(json.dumps(out))
