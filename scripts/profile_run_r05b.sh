#!/bin/bash
# rocprofv3 evidence at the END of round 5 (after "ask later", the wave-parallel tie replay and the direct result slab): kernel stats of the
# headline-only bench and of a run of blocking calls, then the bench line itself. Output: gpurun_out/prof5b/ -> copied to profiles/r05/ by hand.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof5b; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 1 --headline-only > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_blk -- python3 scripts/blocking_calls.py > $OUT/blocking_calls.json 2> $OUT/stats_blk.err
cp $(ls $OUT/stats_blk/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_blocking_calls.csv; rm -rf $OUT/stats_blk
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 3000 $OUT/bench.json
