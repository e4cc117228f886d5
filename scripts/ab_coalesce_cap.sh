#!/bin/bash
# A/B (GPU box): launches of the pipelined path up to 65 536 queries (DR_COALESCE_CAP=65536: six 10 000-query submits per launch) against the default 32 768
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/coalesce_cap.jsonl
for r in 1 2 3 4; do for cap in 32768 65536; do
  export DR_COALESCE_CAP=$cap
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print(json.dumps({'coalesce_cap': $cap, 'value': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'queries_per_launch': c['queries_per_launch'], 'kernel_ms_per_batch': c['kernel_ms_per_batch'], 'tickets_in_flight': c['tickets_in_flight'], 'frac': d['roofline']['frac']}))" >> gpurun_out/ab/coalesce_cap.jsonl
done; done
cat gpurun_out/ab/coalesce_cap.jsonl
