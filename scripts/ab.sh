#!/bin/bash
# A/B two libraries in one box, interleaved (variance control): usage ab.sh libA libB
for round in 1 2 3; do for lib in "$@"; do
  echo -n "$lib: "; DR_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['config']['secondary_no_trim']; print('QPS %.0f kernel_ms %.3f recall %.4f | no-trim QPS %.0f kernel_ms %.3f' % (d['value'], d['roofline']['kernel_ms'], d['config']['recall_at_10'], s['qps_rank0'], s['kernel_ms']))"
done; done
