"""The brute-force ADC search (dr_pq_scan_topk) with several queries per pass (GPU box): pq_scan_topk_multi_kernel beside pq_scan_topk_kernel on the
same table of random code words and the same queries; kernel time per query and whether the answers agree bit for bit.
Usage: python scripts/bench_pq_scan_topk.py [N=64000000] [m=32] [D=128] [nq=64] [k=10]  -> one JSON object on stdout."""
import json
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 64
k = int(sys.argv[5]) if len(sys.argv) > 5 else 10
rs = np.random.default_rng(5)
codes = rs.integers(0, 256, size=(N, m), dtype=np.uint8)
cb = rs.standard_normal((m, 256, D // m), dtype=np.float32)
qs = rs.standard_normal((nq, D), dtype=np.float32)
ix = HipIndex.create_codes(np.zeros((N, 1), dtype=np.uint32), 0, D, cb, codes)
del codes
out = {"N": N, "m": m, "D": D, "nq": nq, "k": k, "code_bytes_per_query": N * m}
res = {}
for mode in ("per_query", "shared_pass"):
    if mode == "per_query":
        os.environ["DR_PQ_SCAN_PER_QUERY"] = "1"
    else:
        os.environ.pop("DR_PQ_SCAN_PER_QUERY", None)
    ix.pq_scan_topk(qs, k)
    runs = [ix.pq_scan_topk(qs, k) for _ in range(3)]
    ms = sorted(r[2] for r in runs)[1]
    res[mode] = (runs[0][0].tobytes(), runs[0][1].tobytes())
    out[mode] = {"kernel_ms_median": ms, "ms_per_query": ms / nq, "GBps_algorithmic": nq * N * m / (ms * 1e-3) / 1e9,
                 "frac_of_8TBps": nq * N * m / (ms * 1e-3) / 8e12}
out["same_answers_bit_for_bit"] = res["per_query"] == res["shared_pass"]
out["speedup"] = out["per_query"]["kernel_ms_median"] / out["shared_pass"]["kernel_ms_median"]
print(json.dumps(out))
