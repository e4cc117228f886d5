#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcsq; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -- python3 scripts/pmc_target.py > $OUT/a.out 2> $OUT/a.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/b -- python3 scripts/pmc_target.py > $OUT/b.out 2> $OUT/b.err
tail -2 $OUT/a.err $OUT/b.err
python3 - <<'PY'
import csv, glob
for d in ("a","b"):
    fs=glob.glob(f"gpurun_out/pmcsq/{d}/*/*counter_collection.csv")
    if not fs: print("no csv", d); continue
    rows=list(csv.DictReader(open(fs[0])))
    agg={}
    for r in rows:
        if "search_kernel<128, true" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,sum(v)/len(v), len(v))
PY
python3 - <<'PY'
import csv, glob, json
out = {"source": "scripts/pmc_sq.sh: rocprofv3 --pmc SQ_* (two passes) over scripts/pmc_target.py: the c2 workload, search_kernel<128, true, ...> (M1), mean per launch, summed over wavefronts"}
for d in ("a", "b"):
    fs = glob.glob(f"gpurun_out/pmcsq/{d}/*/*counter_collection.csv")
    if not fs: continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if "search_kernel<128, true" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            out["kernel"] = r["Kernel_Name"]
    for k, v in agg.items(): out[k] = sum(v) / len(v)
for line in open("gpurun_out/pmcsq/a.out"):
    p = line.split()
    if p and p[0] == "EXPANSIONS_PER_LAUNCH": out["expansions_per_launch"] = float(p[1])
if "expansions_per_launch" in out:
    e = out["expansions_per_launch"]
    out["per_expansion"] = {k.replace("SQ_INSTS_", "").lower(): out[k] / e for k in out if k.startswith("SQ_INSTS_")}
if "SQ_WAIT_ANY" in out and "SQ_WAVE_CYCLES" in out: out["wave_cycles_waiting"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
json.dump(out, open("gpurun_out/pmcsq/sq_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/a $OUT/b
