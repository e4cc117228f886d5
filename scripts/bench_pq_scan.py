"""The isolated ADC kernel against the HBM roofline (GPU box): a PQ-only index of N random code words (far larger than
L2 + Infinity Cache) is scanned flat, one query per block row; bytes = N*m per query.
Usage: python scripts/bench_pq_scan.py [N=64000000] [m=32] [nq=8] [D=128]  -> one JSON object on stdout."""
import json
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 8
D = int(sys.argv[4]) if len(sys.argv) > 4 else 128
rs = np.random.default_rng(5)
codes = rs.integers(0, 256, size=(N, m), dtype=np.uint8)
cb = rs.standard_normal((m, 256, D // m), dtype=np.float32)
q = rs.standard_normal((nq, D), dtype=np.float32)
adj = np.zeros((N, 1), dtype=np.uint32)
ix = HipIndex.create_codes(adj, 0, D, cb, codes)
out = {"N": N, "m": m, "D": D, "nq": nq, "code_bytes_per_query": N * m, "runs": []}
for rep in range(4):
    bid, bsq, ms = ix.pq_scan_best(q)
    out["runs"].append({"kernel_ms": ms, "code_GBps": nq * N * m / (ms * 1e-3) / 1e9, "lookups_per_s": nq * N * m / (ms * 1e-3),
                        "frac_of_8TBps": nq * N * m / (ms * 1e-3) / 8e12})
# check the winner of query 0 on the host (strict sequential float32 sum, A3)
lut0 = ((cb - q[0].reshape(m, 1, D // m)) ** 2).astype(np.float32)
best = int(bid[0])
s = np.float32(0)
for j in range(m):
    t = np.float32(0)
    for v in lut0[j, codes[best, j]]:
        t = np.float32(t + v)
    s = np.float32(s + t)
out["best_id_q0"] = best
out["best_sq_q0_device"] = float(bsq[0])
out["best_sq_q0_host"] = float(s)
print(json.dumps(out))
