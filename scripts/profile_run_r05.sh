#!/bin/bash
# rocprofv3 evidence of round 5 (run on the GPU box through gpurun). Output: gpurun_out/prof5/ -> copied to profiles/r05/ by hand.
#   1. the headline kernel (c2, variant 13): --kernel-trace --stats of `bench.py --headline-only`, PMC traffic (separate passes)
#   2. the FLOAT-ROW kernel as a headline of its own (bench.py --rows f32, variant 9: what embedding workloads get): the same two
#   3. the DR_MODE_PQB kernel on the c5s shape (4M x 1536, R = 32, L = 100, beam_width 8): kernel stats + PMC traffic + SQ counters
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof5; rm -rf $OUT; mkdir -p $OUT
pmc() { # tag, counters, program args...
  local tag=$1; local ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$tag -- python3 "$@" > $OUT/$tag.out 2> $OUT/$tag.err
}
# ---- 1 + 2: c2, engine's choice and float rows
for rows in auto f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$rows -- python3 bench.py --steps 5 --warmup 1 --headline-only --rows $rows > $OUT/bench_under_rocprof_$rows.json 2> $OUT/stats_$rows.err
  cp $(ls $OUT/stats_$rows/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_$rows.csv
  rm -rf $OUT/stats_$rows
  export DR_FORCE_KIND=$([ $rows = f32 ] && echo 9 || echo -1)
  pmc fetch_$rows FETCH_SIZE scripts/pmc_target.py 8
  pmc write_$rows WRITE_SIZE scripts/pmc_target.py 8
  pmc rdreq_$rows "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" scripts/pmc_target.py 8
  unset DR_FORCE_KIND
done
python3 - <<'PY'
import csv, glob, json
def load(path): return list(csv.DictReader(open(path)))
def vals(rows, c, sub): return [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == c and sub in r['Kernel_Name']]
for rows in ("auto", "f32"):
    out = {"source": "scripts/profile_run_r05.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_{sum,32B,64B,128B} (separate passes) on scripts/pmc_target.py 8"
                     + (" with DR_FORCE_KIND=9 (float32 rows)" if rows == "f32" else "") + ", MI355X, ROCm 7.2",
           "units": "FETCH_SIZE / WRITE_SIZE are KiB; read bytes = 128 / 64 / 32 bytes per read request of each size (profiles/r04/tcc_calibration.json)"}
    try:
        f = load(glob.glob(f'gpurun_out/prof5/fetch_{rows}/*/*_counter_collection.csv')[0])
        w = load(glob.glob(f'gpurun_out/prof5/write_{rows}/*/*_counter_collection.csv')[0])
        rq = load(glob.glob(f'gpurun_out/prof5/rdreq_{rows}/*/*_counter_collection.csv')[0])
        alg = [float(l.split()[1]) for l in open(f'gpurun_out/prof5/fetch_{rows}.out') if l.startswith('ALG_BYTES_PER_LAUNCH')][0]
        sub = 'search_kernel<128, true'
        fs, ws = vals(f, 'FETCH_SIZE', sub), vals(w, 'WRITE_SIZE', sub)
        mean = lambda c: (lambda v: sum(v) / len(v) if v else 0.0)(vals(rq, c, sub))
        n_all, n32, n64, n128 = mean('TCC_EA0_RDREQ_sum'), mean('TCC_EA0_RDREQ_32B_sum'), mean('TCC_EA0_RDREQ_64B_sum'), mean('TCC_EA0_RDREQ_128B_sum')
        rest = max(0.0, n_all - n32 - n64 - n128)
        rd = 128 * n128 + 64 * (n64 + rest) + 32 * n32
        wr = sum(ws) / len(ws) * 1024
        out["beam_width_8"] = {"kernel": sorted({r['Kernel_Name'] for r in f if sub in r['Kernel_Name']})[0], "read_requests": {"all": n_all, "128B": n128, "64B": n64, "32B": n32},
                               "read_bytes_per_launch": rd, "read_bytes_fetch_size_doubled": sum(fs) / len(fs) * 2048, "write_bytes_per_launch": wr,
                               "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch_4D_rows": alg,
                               "queries_per_launch": 10000}
    except Exception as e:
        out["error"] = str(e)
    json.dump(out, open('gpurun_out/prof5/pmc_traffic%s.json' % ("" if rows == "auto" else "_f32_rows"), 'w'), indent=1)
    print(rows, json.dumps(out.get("beam_width_8", out), indent=1))
PY
rm -rf $OUT/fetch_auto $OUT/write_auto $OUT/rdreq_auto $OUT/fetch_f32 $OUT/write_f32 $OUT/rdreq_f32
# ---- 3: DR_MODE_PQB on the c5s shape
export PMC_MODE=pqb
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pqb -- python3 scripts/pmc_target_shape.py c5s 4000000 > $OUT/stats_pqb.out 2> $OUT/stats_pqb.err
cp $(ls $OUT/stats_pqb/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_pqb_c5s_4M.csv; rm -rf $OUT/stats_pqb
pmc pqb_fetch FETCH_SIZE scripts/pmc_target_shape.py c5s 4000000
pmc pqb_write WRITE_SIZE scripts/pmc_target_shape.py c5s 4000000
pmc pqb_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" scripts/pmc_target_shape.py c5s 4000000
pmc pqb_sqa "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" scripts/pmc_target_shape.py c5s 4000000
pmc pqb_sqb "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" scripts/pmc_target_shape.py c5s 4000000
python3 - <<'PY'
import csv, glob, json
def rows(d):
    f = glob.glob(f"gpurun_out/prof5/{d}/*/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []
def mean_last(rs, counter, sub, k=3):
    v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter and sub in r["Kernel_Name"]]
    return sum(v[-k:]) / max(1, len(v[-k:])) if v else None
sub = "pqb_search_kernel"
out = {"shape": "c5s", "N": 4000000, "mode": "DR_MODE_PQB, default pops (2 rows per step at R = 32), L = 100, beam_width 8, 10000 queries per launch",
       "source": "scripts/profile_run_r05.sh: rocprofv3 --pmc, separate passes over scripts/pmc_target_shape.py c5s 4000000 with PMC_MODE=pqb; mean of the last 3 launches"}
for line in open("gpurun_out/prof5/pqb_fetch.out"):
    p = line.split()
    if p and p[0] in ("ALG_BYTES_PER_LAUNCH", "CALIB_BYTES"): out[p[0].lower()] = float(p[1])
    if p and p[0] == "KERNEL_MS": out["kernel_ms_under_profiler"] = float(p[1]); out["variant"] = int(p[3])
    if p and p[0] == "PER_QUERY": out["per_query"] = line.strip()
f, w, rq = rows("pqb_fetch"), rows("pqb_write"), rows("pqb_rdreq")
out["kernel"] = sorted({r["Kernel_Name"] for r in f if sub in r["Kernel_Name"]})[:1]
n_all, n32, n64, n128 = (mean_last(rq, c, sub) or 0.0 for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
rd = 128 * n128 + 64 * (n64 + max(0.0, n_all - n32 - n64 - n128)) + 32 * n32
wr = (mean_last(w, "WRITE_SIZE", sub) or 0.0) * 1024
out["read_requests"] = {"all": n_all, "128B": n128, "64B": n64, "32B": n32}
out["read_bytes_per_launch"], out["write_bytes_per_launch"], out["hbm_bytes_per_launch"] = rd, wr, rd + wr
if out.get("alg_bytes_per_launch"): out["traffic_over_algorithmic"] = (rd + wr) / out["alg_bytes_per_launch"]
sq = {}
for d in ("pqb_sqa", "pqb_sqb"):
    rs = rows(d)
    for c in sorted({r["Counter_Name"] for r in rs}):
        v = mean_last(rs, c, sub)
        if v is not None: sq[c] = v
out["sq"] = sq
if "SQ_WAVE_CYCLES" in sq and "SQ_WAIT_ANY" in sq: out["wave_cycles_waiting"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
json.dump(out, open("gpurun_out/prof5/pmc_pqb_c5s_4M.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "sq"}, indent=1))
PY
rm -rf $OUT/pqb_fetch $OUT/pqb_write $OUT/pqb_rdreq $OUT/pqb_sqa $OUT/pqb_sqb
ls -la $OUT
