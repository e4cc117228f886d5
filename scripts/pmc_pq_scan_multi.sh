#!/bin/bash
# the flat scan with 4 queries per pass (pq_scan_multi_kernel): timing, FETCH_SIZE (HBM bytes actually read) and the LDS counters
# (bank-conflict cycles / all LDS-array cycles) beside the one-query kernel's (GPU box; separate --pmc passes)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pqscan_multi; rm -rf $OUT; mkdir -p $OUT
for nq in 1 4; do
python3 scripts/bench_pq_scan.py 64000000 32 $nq > $OUT/nq$nq.json 2> $OUT/nq$nq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch$nq -- python3 scripts/bench_pq_scan.py 64000000 32 $nq > $OUT/fetch$nq.out 2> $OUT/fetch$nq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/lds$nq -- python3 scripts/bench_pq_scan.py 64000000 32 $nq > $OUT/lds$nq.out 2> $OUT/lds$nq.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/bench_pq_scan.py 64000000 32 4 > $OUT/stats.out 2> $OUT/stats.err
python3 - <<'PY'
import csv, glob, json
res = {}
for nq, kname in ((1, "pq_scan_kernel"), (4, "pq_scan_multi_kernel")):
    d = json.load(open(f"gpurun_out/pqscan_multi/nq{nq}.json"))
    ms = sorted(r["kernel_ms"] for r in d["runs"])[1]
    def counters(sub):
        rows = list(csv.DictReader(open(glob.glob(f"gpurun_out/pqscan_multi/{sub}{nq}/*/*_counter_collection.csv")[0])))
        acc = {}
        for r in rows:
            if kname in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in acc.items()}
    f, l = counters("fetch"), counters("lds")
    hbm = f["FETCH_SIZE"] * 1024 * 2      # KiB, and gfx950 counts 128-B requests as 64 B (MI355X_MICROARCH.md)
    res[f"nq{nq}"] = {"kernel": kname, "kernel_ms_median": ms, "algorithmic_bytes": nq * d["code_bytes_per_query"],
                      "GBps_algorithmic": nq * d["code_bytes_per_query"] / ms / 1e6, "FETCH_SIZE_bytes_per_launch": hbm, "hbm_GBps_measured": hbm / ms / 1e6,
                      "lds_counters_per_launch": l, "lds_bank_conflict_share": l.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, l.get("SQ_LDS_IDX_ACTIVE", 0))}
json.dump(res, open("gpurun_out/pqscan_multi/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
grep pq_scan $OUT/stats/*/*kernel_stats.csv | cut -c1-220
