#!/bin/bash
# bench under each M1 kernel variant (0: per-query table, 1 wave/WG; 3: shared codebook 8 waves; 4: 16 waves)
for kd in 9 6 7 3 4; do
  echo "== DR_FORCE_KIND=$kd"
  DR_FORCE_KIND=$kd timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('QPS %.0f recall %.4f ms/step %.2f kernel_ms %.2f finalize_ms %.2f frac %.4f GB/s %.0f launch %s' % (d['value'], c['recall_at_10'], d['ms_per_step'], r['kernel_ms'], c['finalize_kernel_ms'], r['frac'], r['achieved'], c['launch']))"
done
