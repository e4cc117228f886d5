#!/bin/bash
# rocprofv3 evidence for the bench kernel (run on the GPU box through gpurun). Output: gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 1 --headline-only > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for bw in 8 0; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_bw$bw -- python3 scripts/pmc_target.py $bw > $OUT/pmc_fetch_bw$bw.out 2> $OUT/pmc_fetch_bw$bw.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_bw$bw -- python3 scripts/pmc_target.py $bw > $OUT/pmc_write_bw$bw.out 2> $OUT/pmc_write_bw$bw.err
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/pmc_rdreq_bw$bw -- python3 scripts/pmc_target.py $bw > $OUT/pmc_rdreq_bw$bw.out 2> $OUT/pmc_rdreq_bw$bw.err
done
python3 - <<'PY'
import csv, glob, json
def load(path): return list(csv.DictReader(open(path)))
def vals(rows, c, sub): return [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == c and sub in r['Kernel_Name']]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_EA0_RDREQ_{sum,32B,64B,128B} (separate passes) on scripts/pmc_target.py <beam_width>, MI355X, ROCm 7.2",
       "units": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE = 64 B x read requests whatever their size (profiles/r04/tcc_calibration.json): "
                "read_bytes_per_launch doubles it (the guide's correction, exact when every request is 128 bytes: an upper bound here), "
                "read_bytes_by_request_size adds 128 / 64 / 32 bytes per request of each size (exact); hbm_bytes_per_launch uses the latter when it was collected"}
for bw in (8, 0):
    f = load(glob.glob(f'gpurun_out/prof/pmc_fetch_bw{bw}/*/*_counter_collection.csv')[0])
    w = load(glob.glob(f'gpurun_out/prof/pmc_write_bw{bw}/*/*_counter_collection.csv')[0])
    alg = [float(l.split()[1]) for l in open(f'gpurun_out/prof/pmc_fetch_bw{bw}.out') if l.startswith('ALG_BYTES_PER_LAUNCH')][0]
    fs, ws = vals(f, 'FETCH_SIZE', 'search_kernel<128, true'), vals(w, 'WRITE_SIZE', 'search_kernel<128, true')
    cal = vals(f, 'FETCH_SIZE', 'bruteforce_kernel')
    rd, wr = sum(fs) / len(fs) * 1024 * 2, sum(ws) / len(ws) * 1024
    rd_doubled, by_size = rd, None
    try:
        rq = load(glob.glob(f'gpurun_out/prof/pmc_rdreq_bw{bw}/*/*_counter_collection.csv')[0])
        mean = lambda c: (lambda v: sum(v) / len(v) if v else 0.0)(vals(rq, c, 'search_kernel<128, true'))
        n_all, n32, n64, n128 = mean('TCC_EA0_RDREQ_sum'), mean('TCC_EA0_RDREQ_32B_sum'), mean('TCC_EA0_RDREQ_64B_sum'), mean('TCC_EA0_RDREQ_128B_sum')
        rest = max(0.0, n_all - n32 - n64 - n128)
        by_size = {"requests": n_all, "128B": n128, "64B": n64, "32B": n32, "unsized_counted_as_64B": rest, "bytes": 128 * n128 + 64 * (n64 + rest) + 32 * n32}
        if n_all > 0: rd = by_size["bytes"]
    except Exception as e:
        by_size = {"error": str(e)}
    out[f"beam_width_{bw}"] = {"kernel": sorted({r['Kernel_Name'] for r in f if 'search_kernel<128, true' in r['Kernel_Name']})[0], "FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": ws,
                               "read_bytes_per_launch": rd, "read_bytes_fetch_size_doubled": rd_doubled, "read_bytes_by_request_size": by_size,
                               "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                               "algorithmic_bytes_per_launch": alg,
                               "calibration": {"kernel": "bruteforce_kernel<128>, 1 query", "known_bytes": 512000000, "FETCH_SIZE_KiB": cal[0],
                                               "corrected_bytes": cal[0] * 1024 * 2}}
json.dump(out, open('gpurun_out/prof/pmc_traffic.json', 'w'), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if 'bytes' in kk}) for k, v in out.items()}, indent=1))
PY
# the bench line quotes roofline.traffic from profiles/r04/pmc_traffic.json: refresh it first, then run the bench
mkdir -p profiles/r04; cp $OUT/pmc_traffic.json profiles/r04/pmc_traffic.json
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cp $OUT/bench.json profiles/r04/bench.json; cp $OUT/bench_under_rocprof.json profiles/r04/bench_under_rocprof.json
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) profiles/r04/kernel_stats.csv
mkdir -p gpurun_out/r04prof; cp profiles/r04/bench.json profiles/r04/bench_under_rocprof.json profiles/r04/kernel_stats.csv profiles/r04/pmc_traffic.json gpurun_out/r04prof/
# the raw traces stay on the box (gpurun_out is capped at 64 MiB)
rm -rf $OUT/stats $OUT/pmc_fetch_bw8 $OUT/pmc_fetch_bw0 $OUT/pmc_write_bw8 $OUT/pmc_write_bw0 $OUT/pmc_rdreq_bw8 $OUT/pmc_rdreq_bw0
