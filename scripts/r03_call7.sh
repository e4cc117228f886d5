#!/bin/bash
# round 3, GPU call 7: GPU suite; c2 companions A/B (new upload path vs round 2's); small batches with and without the second
# search lane; D = 960 operating points again (table kernel for sub_dim 30); c4 at its full per-GPU size; PMC of the c4 shape
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests7.log 2>&1; grep -E "passed|failed" $O/gputests7.log | tail -1
bash scripts/ab_companions.sh 2 > $O/ab_companions_v3.log 2>&1; cat $O/ab_companions_v3.log
for lane in two one; do
  [ $lane = one ] && export DR_ONE_LANE=1 || unset DR_ONE_LANE
  timeout 600 python bench.py --no-cpu > $O/bench_lanes_$lane.json 2> $O/bench_lanes_$lane.err
done
unset DR_ONE_LANE
timeout 900 python scripts/operating_points.py d960 1000000 10000 quick > $O/op_d960.log 2>&1; cp gpurun_out/op_d960_1000000.jsonl $O/
timeout 2400 python scripts/operating_points.py c4 100000000 10000 quick+extra > $O/op_c4_100M.log 2>&1; cp gpurun_out/op_c4_100000000_quick_extra.jsonl $O/
timeout 900 bash scripts/pmc_shape.sh c4 10000000 > $O/pmc_c4.log 2>&1; cp gpurun_out/pmc_c4/summary.json $O/pmc_c4_10M.json
DR_INLINE=1 timeout 900 bash scripts/pmc_shape.sh c4 10000000 > $O/pmc_c4_inline.log 2>&1; cp gpurun_out/pmc_c4/summary.json $O/pmc_c4_10M_inline_codes.json
du -sh gpurun_out
