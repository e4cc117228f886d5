#!/bin/bash
# A/B two libraries in one box, interleaved (variance control): usage ab.sh libA libB
for round in 1 2 3; do for lib in "$@"; do
  echo -n "$lib: "; DR_LIB=$PWD/$lib timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; s=c['secondary_no_trim'] or {}
print('value(host->host) %.0f resident %.0f kernel_ms %.3f recall %.4f | no-trim kernel_ms %.3f | f32rows %.3f | unrounded %.3f' % (d['value'], c['qps_resident'], d['roofline']['kernel_ms'], c['recall_at_10'], s.get('kernel_ms', 0), (c['float32_rows'] or {}).get('kernel_ms', 0), (c.get('unrounded_data') or {}).get('kernel_ms', 0)))"
done; done
