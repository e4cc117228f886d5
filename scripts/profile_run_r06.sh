#!/bin/bash
# rocprofv3 evidence of round 6 (run on the GPU box through gpurun). Output: gpurun_out/prof6/ -> copied to profiles/r06/ by hand.
#   1. the headline kernel (c2, variant 13): --kernel-trace --stats of `bench.py --headline-only`; PMC traffic (separate --pmc passes) in the
#      shape of the bench's COALESCED launches (PMC_NQ=30000 queries per launch) and with one batch per launch; the library's hash in every file
#   2. where the written bytes go (VERDICT r5 item 2 iii): the same passes with DR_NO_LOG=1 DR_SKIP_FINALIZE=1 (no insert log) -- WRITE_SIZE and the
#      kernel's duration with and without
#   3. SQ counters of the occupancy A/B (variant 17 at 16 wavefronts per CU against the 24-wavefront build, when that library is present)
#   4. the DR_MODE_PQB traversal on the shapes of `bench.py --config c3 | c4` at their default operating points and on c5s: kernel stats + PMC traffic
# usage: profile_run_r06.sh [parts: any of 1 2 3 4, default all]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
PARTS=${*:-1 2 3 4}
OUT=gpurun_out/prof6; mkdir -p $OUT
pmc() { # tag, counters, program args...
  local tag=$1; local ctr=$2; shift 2
  rm -rf $OUT/$tag
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$tag -- python3 "$@" > $OUT/$tag.out 2> $OUT/$tag.err
}
RD="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
WR="WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
SQA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
SQB="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"
for part in $PARTS; do case $part in
1)
  rm -rf $OUT/stats_c2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 bench.py --steps 5 --warmup 1 --headline-only > $OUT/bench_under_rocprof.json 2> $OUT/stats_c2.err
  cp $(ls $OUT/stats_c2/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/stats_c2
  for nq in 30000 10000; do
    export PMC_NQ=$nq
    pmc rd_$nq "$RD" scripts/pmc_target.py 8
    pmc wr_$nq "$WR" scripts/pmc_target.py 8
    unset PMC_NQ
  done ;;
2)
  export PMC_NQ=30000 DR_NO_LOG=1 DR_SKIP_FINALIZE=1
  pmc wr_nolog "$WR" scripts/pmc_target.py 8
  pmc rd_nolog "$RD" scripts/pmc_target.py 8
  unset PMC_NQ DR_NO_LOG DR_SKIP_FINALIZE ;;
3)
  export DR_FORCE_KIND=17
  pmc sqa_k17 "$SQA" scripts/pmc_target.py 8
  pmc sqb_k17 "$SQB" scripts/pmc_target.py 8
  pmc rd_k17 "$RD" scripts/pmc_target.py 8
  if [ -f diskrag_amd/csrc/build_ab/r6_m1_w6.so ]; then
    export DR_LIB=$PWD/diskrag_amd/csrc/build_ab/r6_m1_w6.so
    pmc sqa_k17_w6 "$SQA" scripts/pmc_target.py 8
    pmc sqb_k17_w6 "$SQB" scripts/pmc_target.py 8
    pmc rd_k17_w6 "$RD" scripts/pmc_target.py 8
    unset DR_LIB
  fi
  unset DR_FORCE_KIND ;;
4)
  export PMC_MODE=pqb
  # (shape, N, beam_width, L): bench.py's default operating points (OPERATING_POINTS) + the c5s shape of round 5's profile
  for spec in "c3 1000000 ${C3_BW:-0} ${C3_L:-100}" "c4 4000000 ${C4_BW:-8} ${C4_L:-200}" "c5s 4000000 8 100"; do
    set -- $spec; tag=$1
    rm -rf $OUT/stats_pqb_$tag
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pqb_$tag -- python3 scripts/pmc_target_shape.py $spec > $OUT/stats_pqb_$tag.out 2> $OUT/stats_pqb_$tag.err
    cp $(ls $OUT/stats_pqb_$tag/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_pqb_$tag.csv; rm -rf $OUT/stats_pqb_$tag
    pmc pqb_rd_$tag "$RD" scripts/pmc_target_shape.py $spec
    pmc pqb_wr_$tag "$WR" scripts/pmc_target_shape.py $spec
  done
  unset PMC_MODE ;;
esac; done
python3 - <<'PY'
import csv, glob, json, os
OUT = "gpurun_out/prof6"
def rows(tag):
    f = glob.glob(f"{OUT}/{tag}/*/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []
def mean_last(rs, counter, sub, k=3):
    v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter and sub in r["Kernel_Name"]]
    return sum(v[-k:]) / max(1, len(v[-k:])) if v else None
def outvals(tag):
    d = {}
    p = f"{OUT}/{tag}.out"
    if os.path.exists(p):
        for line in open(p):
            w = line.split()
            if w and w[0].isupper() and len(w) > 1:
                d[w[0].lower()] = w[1] if w[0] in ("BUILD_SHA1", "PER_QUERY") else float(w[1]) if w[1].replace(".", "", 1).replace("e+", "", 1).replace("-", "", 1).isdigit() else w[1]
                if w[0] == "PER_QUERY": d["per_query"] = line.strip()
    return d
def traffic(rd_tag, wr_tag, sub):
    rq, w = rows(rd_tag), rows(wr_tag)
    if not rq or not w: return None
    n_all, n32, n64, n128 = (mean_last(rq, c, sub) or 0.0 for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
    rest = max(0.0, n_all - n32 - n64 - n128)
    rd = 128 * n128 + 64 * (n64 + rest) + 32 * n32
    wr = (mean_last(w, "WRITE_SIZE", sub) or 0.0) * 1024
    o = {"kernel": sorted({r["Kernel_Name"] for r in rq if sub in r["Kernel_Name"]})[:1], "read_requests": {"all": n_all, "128B": n128, "64B": n64, "32B": n32},
         "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "write_requests": {"all": mean_last(w, "TCC_EA0_WRREQ_sum", sub), "64B": mean_last(w, "TCC_EA0_WRREQ_64B_sum", sub)},
         "hbm_bytes_per_launch": rd + wr}
    o.update(outvals(rd_tag))
    if o.get("alg_bytes_own_per_launch"): o["traffic_over_own_algorithmic_bytes"] = (rd + wr) / o["alg_bytes_own_per_launch"]
    elif o.get("alg_bytes_per_launch"): o["traffic_over_algorithmic_bytes"] = (rd + wr) / o["alg_bytes_per_launch"]
    return o
src = "scripts/profile_run_r06.sh: rocprofv3 --pmc TCC_EA0_RDREQ_{sum,32B,64B,128B} and WRITE_SIZE + TCC_EA0_WRREQ (separate passes), mean of the last 3 launches, MI355X, ROCm 7.2; read bytes = 128 / 64 / 32 per request of each size (profiles/r04/tcc_calibration.json), WRITE_SIZE in KiB"
c2 = {"source": src + "; target scripts/pmc_target.py 8 (the bench workload, resident launches)"}
for nq in (30000, 20000, 10000):
    t = traffic(f"rd_{nq}", f"wr_{nq}", "search_kernel<128, true")
    if t: c2["queries_per_launch_%d" % nq] = t
t = traffic("rd_nolog", "wr_nolog", "search_kernel<128, true")
if t: c2["queries_per_launch_30000_no_insert_log"] = dict(t, note="DR_NO_LOG=1 DR_SKIP_FINALIZE=1: the insert log is not written (tie order wrong: counters and timing only)")
if len(c2) > 1:
    # bench.py reads beam_width_8 (one batch per launch, scaled by its batches per launch) -- kept for that reader
    if "queries_per_launch_10000" in c2: c2["beam_width_8"] = dict(c2["queries_per_launch_10000"], queries_per_launch=10000)
    json.dump(c2, open(f"{OUT}/pmc_traffic.json", "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("hbm_bytes_per_launch", "read_bytes_per_launch", "write_bytes_per_launch", "kernel_ms", "build_sha1", "traffic_over_own_algorithmic_bytes") if kk in v}) for k, v in c2.items() if k != "source"}, indent=1))
sq = {}
for tag in ("k17", "k17_w6"):
    o = {}
    for part in ("sqa", "sqb"):
        rs = rows(f"{part}_{tag}")
        for c in sorted({r["Counter_Name"] for r in rs}):
            o[c] = mean_last(rs, c, "search_kernel<128, true")
    if o:
        o.update(outvals(f"sqa_{tag}"))
        if o.get("SQ_WAIT_ANY") and o.get("SQ_WAVE_CYCLES"): o["wave_cycles_waiting"] = o["SQ_WAIT_ANY"] / o["SQ_WAVE_CYCLES"]
        if o.get("expansions_per_launch"): o["per_expansion"] = {c.replace("SQ_INSTS_", "").lower(): o[c] / o["expansions_per_launch"] for c in o if c.startswith("SQ_INSTS_") and o[c]}
        t = rows(f"rd_{tag}")
        if t: o["read_requests_per_launch"] = mean_last(t, "TCC_EA0_RDREQ_sum", "search_kernel<128, true")
        sq["variant_17_16_waves_per_cu" if tag == "k17" else "variant_17_ab_build_24_waves_per_cu_80_vgprs"] = o
if sq:
    sq["source"] = "scripts/profile_run_r06.sh part 3: rocprofv3 --pmc SQ_* (two passes) + TCC_EA0_RDREQ over scripts/pmc_target.py 8 with DR_FORCE_KIND=17; the A/B build is diskrag_amd/csrc/build_ab/r6_m1_w6.so (-DDR_AB_RB17=32 -DDR_AB_MINW17=6)"
    json.dump(sq, open(f"{OUT}/sq_counters_occupancy_ab.json", "w"), indent=1)
    print(json.dumps({k: {kk: v[kk] for kk in ("kernel_ms", "wave_cycles_waiting", "per_expansion", "read_requests_per_launch", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY") if kk in v} for k, v in sq.items() if isinstance(v, dict)}, indent=1))
for tag in ("c3", "c4", "c5s"):
    t = traffic(f"pqb_rd_{tag}", f"pqb_wr_{tag}", "pqb_search_kernel")
    if not t: continue
    spec = open(f"{OUT}/stats_pqb_{tag}.out").read() if os.path.exists(f"{OUT}/stats_pqb_{tag}.out") else ""
    t["source"] = src + "; target scripts/pmc_target_shape.py with PMC_MODE=pqb (DR_MODE_PQB; c3 / c4 with DR_F_RERANK: the rerank pass is its own kernel and not in these figures)"
    t["shape"] = tag
    for key, tgt in (("n", "N"), ("l", "L"), ("bw", "bw")):
        if key in t: t[tgt] = int(t.pop(key)) if key != "bw" else int(t[key])
    json.dump(t, open(f"{OUT}/pmc_pqb_{tag}.json", "w"), indent=1)
    print(tag, json.dumps({kk: t[kk] for kk in ("hbm_bytes_per_launch", "kernel_ms", "alg_bytes_per_launch", "traffic_over_algorithmic_bytes", "per_query") if kk in t}))
PY
rm -rf $OUT/rd_* $OUT/wr_* $OUT/sqa_* $OUT/sqb_* $OUT/pqb_rd_* $OUT/pqb_wr_*
ls $OUT
