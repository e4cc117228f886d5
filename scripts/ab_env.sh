#!/bin/bash
# bench under different environment settings: usage ab_env.sh "VAR=1 VAR2=x" "..." (each arg = one setting; "-" = none)
for round in 1 2; do for setting in "$@"; do
  echo -n "[$setting] "
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['config']['secondary_no_trim']; print('QPS %.0f kernel_ms %.3f | no-trim QPS %.0f kernel_ms %.3f | build %.1f' % (d['value'], d['roofline']['kernel_ms'], s['qps_rank0'], s['kernel_ms'], d['config']['build_seconds']))"
done; done
