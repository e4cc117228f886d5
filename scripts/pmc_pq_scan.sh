#!/bin/bash
# isolated ADC scan: timing at nq = 1 and 8, and FETCH_SIZE (HBM bytes actually read) for both (GPU box)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pqscan; rm -rf $OUT; mkdir -p $OUT
timeout 300 python3 scripts/bench_pq_scan.py 64000000 32 1 > $OUT/nq1.json 2> $OUT/nq1.err
timeout 300 python3 scripts/bench_pq_scan.py 64000000 32 8 > $OUT/nq8.json 2> $OUT/nq8.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 scripts/bench_pq_scan.py 64000000 32 1 > $OUT/pmc1.out 2> $OUT/pmc1.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc8 -- python3 scripts/bench_pq_scan.py 64000000 32 8 > $OUT/pmc8.out 2> $OUT/pmc8.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/bench_pq_scan.py 64000000 32 1 > $OUT/stats.out 2> $OUT/stats.err
python3 - <<'PY'
import csv, glob, json
res = {}
for nq in (1, 8):
    d = json.load(open(f"gpurun_out/pqscan/nq{nq}.json"))
    ms = sorted(r["kernel_ms"] for r in d["runs"])[1]
    rows = list(csv.DictReader(open(glob.glob(f"gpurun_out/pqscan/pmc{nq}/*/*_counter_collection.csv")[0])))
    fs = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "FETCH_SIZE" and ("pq_scan_skew_kernel" in r["Kernel_Name"] if nq == 1 else "pq_scan_multi_kernel" in r["Kernel_Name"])]      # (round 6: the one-query scan is pq_scan_skew_kernel)
    hbm = sum(fs) / len(fs) * 1024 * 2      # KiB, and gfx950 counts 128-B requests as 64 B (MI355X_MICROARCH.md)
    res[f"nq{nq}"] = {"kernel_ms_median": ms, "code_bytes": nq * d["code_bytes_per_query"], "code_GBps": nq * d["code_bytes_per_query"] / ms / 1e6,
                      "frac_of_8TBps_algorithmic": nq * d["code_bytes_per_query"] / ms / 1e6 / 8000, "FETCH_SIZE_bytes_per_launch": hbm,
                      "hbm_GBps_measured": hbm / ms / 1e6, "launches_profiled": len(fs)}
json.dump(res, open("gpurun_out/pqscan/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
grep pq_scan $OUT/stats/*/*kernel_stats.csv | cut -c1-200
