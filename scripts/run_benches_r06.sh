cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b6
bash scripts/run_gpu_tests.sh
python bench.py > gpurun_out/b6/bench.json 2> gpurun_out/b6/bench.err
python bench.py --config c3 > gpurun_out/b6/bench_config_c3.json 2> gpurun_out/b6/c3.err
python bench.py --config c4 > gpurun_out/b6/bench_config_c4.json 2> gpurun_out/b6/c4.err
DR_BENCH_NO_INLINE=1 python bench.py --config c4 --no-cpu > gpurun_out/b6/bench_config_c4_no_inline.json 2> gpurun_out/b6/c4b.err
python bench.py --config c5 > gpurun_out/b6/bench_config_c5.json 2> gpurun_out/b6/c5.err
python bench.py --gpus 8 --no-secondary --steps 5 --warmup 1 > gpurun_out/b6/bench_8_ranks_on_one_gpu.json 2> gpurun_out/b6/g8.err
for f in bench bench_config_c3 bench_config_c4 bench_config_c4_no_inline bench_config_c5 bench_8_ranks_on_one_gpu; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/b6/$f.json") if l.startswith("{")][-1]); c=d["config"]
    print("$f", round(d["value"]), d.get("roofline",{}).get("frac"), c.get("recall_at_10", c.get("recall_at_10_vs_bruteforce_adc")), c.get("qps_blocking_call_median"), c.get("one_resident_batch_ms"), c.get("per_rank_setup"), (c.get("m64") or {}).get("points"))
except Exception as e: print("$f FAILED", e)
PY
done
tail -3 gpurun_out/b6/*.err | cut -c1-300
