#!/bin/bash
# A/B two libraries in one box, interleaved (variance control): usage ab.sh libA libB
for round in 1 2 3; do for lib in "$@"; do
  echo -n "$lib: "; DR_LIB=$PWD/$lib DR_FORCE_KIND=3 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('QPS %.0f kernel_ms %.3f ms/step %.3f' % (d['value'], d['roofline']['kernel_ms'], d['ms_per_step']))"
done; done
