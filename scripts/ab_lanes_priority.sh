#!/bin/bash
# A/B (GPU box): one search lane / + high-priority companion streams / two lanes + high-priority companions, interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/lanes_priority.jsonl
for r in 1 2 3; do for cfg in base lanes lanes_prio lanes_prio2 lanes_prio3 lanes_prio4; do
  unset DR_TWO_LANES DR_COMPANION_PRIORITY
  [ $cfg = prio ] && export DR_COMPANION_PRIORITY=1
  [ $cfg = lanes ] && export DR_TWO_LANES=1
  [ $cfg = lanes_prio ] && export DR_COMPANION_PRIORITY=1 DR_TWO_LANES=1
  [ $cfg = lanes_prio2 ] && export DR_COMPANION_PRIORITY=2 DR_TWO_LANES=1
  [ $cfg = lanes_prio3 ] && export DR_COMPANION_PRIORITY=3 DR_TWO_LANES=1
  [ $cfg = lanes_prio4 ] && export DR_COMPANION_PRIORITY=4 DR_TWO_LANES=1
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print(json.dumps({'config': '$cfg', 'value': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'queries_per_launch': c['queries_per_launch'], 'timed_region_s': c['timed_region_s']}))" >> gpurun_out/ab/lanes_priority.jsonl
done; done
cat gpurun_out/ab/lanes_priority.jsonl
