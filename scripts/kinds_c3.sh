#!/bin/bash
# c3-shaped run (D = 1536 unit-norm) under the single-wave (0) and multi-wave (15) M1 variants. usage: kinds_c3.sh N
N=${1:-1000000}
for kd in 15 0; do
  echo "== DR_FORCE_KIND=$kd"
  DR_FORCE_KIND=$kd timeout 900 python scripts/scale_measurements.py c3 $N 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for k,r in d['runs'].items(): print('%-32s qps %9.0f kernel_ms %7.2f recall %.3f alg_frac %.3f waves/CU %s exact %.0f pq_eval %.0f' % (k, r['qps'], r['kernel_ms'], r['recall_at_10'], r['frac_of_8TBps'], r['launch']['waves_per_cu'], r['exact'], r['pq_evaluated']))"
done
