#!/bin/bash
# is the exact-vector builder's prune bound by bytes? FETCH_SIZE of prune_kernel / search_kernel over one 2M x 1536 build
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O; rm -rf gpurun_out/bpmc
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/bpmc -- python3 scripts/exp_build_profile.py 2097152 1536 64 100 > $O/build_pmc.out 2> $O/build_pmc.err
python3 - <<'PY'
import csv, glob, json
cc = glob.glob('gpurun_out/bpmc/*/*_counter_collection.csv')[0]
kt = glob.glob('gpurun_out/bpmc/*/*_kernel_trace.csv')[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
agg = {}
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'FETCH_SIZE': continue
    name = r['Kernel_Name'].split('(')[0][:60]
    a = agg.setdefault(name, [0.0, 0.0, 0])
    a[0] += float(r['Counter_Value']) * 1024 * 2          # KiB, 128-byte requests counted as 64 (gfx950): doubled
    a[1] += dur.get(r['Dispatch_Id'], 0.0); a[2] += 1
out = {k: {"launches": v[2], "seconds": v[1], "fetch_bytes": v[0], "fetch_TBps": (v[0] / v[1] / 1e12) if v[1] else None} for k, v in agg.items() if v[1] > 0.05}
json.dump(out, open('gpurun_out/r03/build_exact_pmc_2M_d1536.json', 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['seconds']): print(k, v)
PY
cat $O/build_pmc.out; rm -rf gpurun_out/bpmc
