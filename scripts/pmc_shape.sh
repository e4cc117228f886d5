#!/bin/bash
# PMC evidence for the c3 / c4 shapes: FETCH_SIZE, WRITE_SIZE and the SQ wait / instruction counters of
# the M1 kernel, separate passes. usage: pmc_shape.sh c3 1000000 | c4 10000000 | c5s 4000000   -> gpurun_out/pmc_<shape>/summary.json
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
S=$1; N=$2; OUT=gpurun_out/pmc_$S; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 scripts/pmc_target_shape.py $S $N > $OUT/$1.out 2> $OUT/$1.err; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sqa "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
run tcca "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"
run tccb "TCC_REQ_sum TCC_READ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
run tccc "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum TCC_WRITE_sum"
run sqb "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"
python3 - $S $N <<'PY'
import csv, glob, json, sys
S, N = sys.argv[1], int(sys.argv[2])
out = {"shape": S, "N": N, "source": "scripts/pmc_shape.sh: rocprofv3 --pmc, separate passes over scripts/pmc_target_shape.py (M1 -- c5s: DR_MODE_PQ on R = 32 rows --, L=100, beam_width 8, 10000 queries; mean of the last launches)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B, calibrated on the brute-force stream of the whole vector table); SQ_* are summed over wavefronts"}
def rows(d):
    f = glob.glob(f"gpurun_out/pmc_{S}/{d}/*/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []
def mean_last(rs, counter, sub, k=3):
    v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter and sub in r["Kernel_Name"]]
    return sum(v[-k:]) / max(1, len(v[-k:])) if v else None
kern = "search_kernel<"
for line in open(f"gpurun_out/pmc_{S}/fetch.out"):
    p = line.split()
    if p and p[0] in ("ALG_BYTES_PER_LAUNCH", "CALIB_BYTES"): out[p[0].lower()] = float(p[1])
    if p and p[0] == "KERNEL_MS": out["kernel_ms_under_profiler"] = float(p[1]); out["variant"] = int(p[3])
    if p and p[0] == "PER_QUERY": out["per_query"] = line.strip()
f, w = rows("fetch"), rows("write")
def wanted(nm):      # M1 kernels have FILTER = true; the PQ-only traversal (c5s) is <D, false, 2 (DIST_ADC_SQ), ...>
    a = [t.strip() for t in nm.split(",")]
    return kern in nm and len(a) > 2 and ((a[1] == "false" and a[2] == "2") if S == "c5s" else a[1] == "true")
names = sorted({r["Kernel_Name"] for r in f if wanted(r["Kernel_Name"])})
out["kernel"] = names[0] if names else None
sub = names[0] if names else kern
fs, ws = mean_last(f, "FETCH_SIZE", sub), mean_last(w, "WRITE_SIZE", sub)
cal = mean_last(f, "FETCH_SIZE", "bruteforce_kernel", 1)
if fs is not None:
    out["read_bytes_per_launch"] = fs * 1024 * 2; out["write_bytes_per_launch"] = (ws or 0) * 1024
    out["hbm_bytes_per_launch"] = out["read_bytes_per_launch"] + out["write_bytes_per_launch"]
    out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out.get("alg_bytes_per_launch", float("nan"))
if cal: out["calibration"] = {"known_bytes": out.get("calib_bytes"), "corrected_bytes": cal * 1024 * 2}
sq = {}
tcc = {}
for d in ("tcca", "tccb", "tccc"):
    rs = rows(d)
    for c in sorted({r["Counter_Name"] for r in rs}):
        v = mean_last(rs, c, sub)
        if v is not None: tcc[c] = v
out["tcc"] = tcc
# byte-exact read traffic from the request-size split (profiles/r04/tcc_calibration.json: FETCH_SIZE counts 64 B per request whatever its size)
if "TCC_EA0_RDREQ_sum" in tcc and "TCC_EA0_RDREQ_128B_sum" in tcc:
    n128, n64 = tcc["TCC_EA0_RDREQ_128B_sum"], tcc.get("TCC_EA0_RDREQ_64B_sum", 0.0)
    n32 = tcc.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    rest = max(0.0, tcc["TCC_EA0_RDREQ_sum"] - n128 - n64 - n32)
    out["read_bytes_by_request_size"] = {"requests": tcc["TCC_EA0_RDREQ_sum"], "128B": n128, "64B": n64, "32B": n32, "unsized_counted_as_64B": rest,
                                         "bytes": 128 * n128 + 64 * (n64 + rest) + 32 * n32}
    if "write_bytes_per_launch" in out:
        out["hbm_bytes_per_launch_by_request_size"] = out["read_bytes_by_request_size"]["bytes"] + out["write_bytes_per_launch"]
        out["traffic_over_algorithmic_by_request_size"] = out["hbm_bytes_per_launch_by_request_size"] / out.get("alg_bytes_per_launch", float("nan"))
for d in ("sqa", "sqb"):
    rs = rows(d)
    for c in sorted({r["Counter_Name"] for r in rs}):
        v = mean_last(rs, c, sub)
        if v is not None: sq[c] = v
out["sq"] = sq
if "SQ_WAVE_CYCLES" in sq and "SQ_WAIT_ANY" in sq: out["wave_cycles_waiting"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
json.dump(out, open(f"gpurun_out/pmc_{S}/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "sq"}, indent=1))
PY
# the raw counter CSVs (one row per dispatch, the builder launches hundreds of thousands) stay on the box: gpurun_out is capped at 64 MiB
rm -rf $OUT/fetch $OUT/write $OUT/sqa $OUT/sqb $OUT/tcca $OUT/tccb $OUT/tccc
