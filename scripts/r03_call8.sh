#!/bin/bash
# round 3, GPU call 8: pipeline depth and ballot counts on the 16-wave variants: interleaved bench A/B of three builds
# (A: depth 3, ballots on one-wave variants; C: depth 4; B: depth 4 + ballots everywhere); GPU suite on B; c5 bench (1 rank)
# round 2 vs now; c3 at its full size
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests8.log 2>&1; grep -E "passed|failed" $O/gputests8.log | tail -1
for round in 1 2 3; do for lib in libdiskrag_hip_a.so libdiskrag_hip_c.so libdiskrag_hip.so; do
  echo -n "$lib: " >> $O/ab_depth_ballots.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f kernel_ms %.4f | blocking %.0f recall %.4f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['kernel_ms_resident'], c['qps_blocking_call']['median'], c['recall_at_10']))" >> $O/ab_depth_ballots.log
done; done
cat $O/ab_depth_ballots.log
for lib in libdiskrag_hip_r02.so libdiskrag_hip.so; do
  echo -n "$lib: " >> $O/ab_c5_bench_1rank.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python bench.py --config c5 --steps 10 --warmup 2 2>/dev/null >> $O/ab_c5_bench_1rank.log
done
cat $O/ab_c5_bench_1rank.log | cut -c1-900
timeout 2400 python scripts/operating_points.py c3 10000000 10000 quick+extra > $O/op_c3_10M.log 2>&1; cp gpurun_out/op_c3_10000000_quick_extra.jsonl $O/
du -sh gpurun_out
