#!/bin/bash
# A/B (GPU box): the pipelined path's launch policy for bulk streams -- round 6's default (a group of >= 8192 queries collects on to 30 000 while one search
# is running) against round 5's (DR_KICK_MIN_QUERIES=0: launched as soon as fewer than two searches are in flight), interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/kick_min.jsonl
for r in 1 2 3 4 5; do for mn in 0 default; do
  if [ $mn = default ]; then unset DR_KICK_MIN_QUERIES; else export DR_KICK_MIN_QUERIES=$mn; fi
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print(json.dumps({'kick_min_queries': '$mn', 'value': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'queries_per_launch': c['queries_per_launch'], 'frac': d['roofline']['frac'], 'kernel_ms_per_batch': c['kernel_ms_per_batch']}))" >> gpurun_out/ab/kick_min.jsonl
done; done
cat gpurun_out/ab/kick_min.jsonl
