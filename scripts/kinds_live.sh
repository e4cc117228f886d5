#!/bin/bash
# A4-live (unit-norm) data under each M1 kernel variant: usage kinds_live.sh <shape> <n> kinds...
shape=$1; n=$2; shift 2
for kd in "$@"; do
  echo "== $shape N=$n DR_FORCE_KIND=$kd"
  DR_FORCE_KIND=$kd timeout 300 python scripts/scale_measurements.py $shape $n 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,r in d['runs'].items():
    if k.startswith('M1'): print(k, 'QPS %.0f recall %.3f kernel_ms %.2f steps %.1f exact %.0f pq_eval %.0f launch %s' % (r['qps'], r['recall_at_10'], r['kernel_ms'], r['steps'], r['exact'], r['pq_evaluated'], r['launch']))"
done
