#!/bin/bash
# rocprofv3 evidence for the bench kernel (run on the GPU box through gpurun). Output: gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 scripts/pmc_target.py > $OUT/pmc_fetch.out 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 scripts/pmc_target.py > $OUT/pmc_write.out 2> $OUT/pmc_write.err
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" | head -20; tail -2 $OUT/pmc_fetch.out; tail -3 $OUT/pmc_fetch.err
