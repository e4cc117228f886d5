#!/bin/bash
# experiment: two search lanes for FULL batches (two processes on one GPU gave 8.58 M QPS against 7.66 M for one)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for round in 1 2 3; do for v in off on; do
  [ $v = on ] && export DR_LANES_ALL=1 || unset DR_LANES_ALL
  echo -n "lanes_all=$v: " >> $O/ab_lanes_all.log
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f kernel_ms %.4f recall %.4f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['kernel_ms_resident'], c['recall_at_10']))" >> $O/ab_lanes_all.log
done; done
cat $O/ab_lanes_all.log
