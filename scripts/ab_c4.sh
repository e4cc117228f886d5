#!/bin/bash
# A/B two libraries on c4-shaped A4-live data (D=96 unit-norm, N=2M) + c2 headline: usage ab_c4.sh libA libB
for lib in "$@"; do
  echo "== $lib"
  DR_LIB=$PWD/$lib timeout 300 python scripts/scale_measurements.py c4 2000000 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,r in d['runs'].items():
    print(k, 'QPS %.0f recall %.3f kernel_ms %.2f steps %.1f exact %.0f pq_eval %.0f GB/s %.0f' % (r['qps'], r['recall_at_10'], r['kernel_ms'], r['steps'], r['exact'], r['pq_evaluated'], r['alg_GBps']))"
done
