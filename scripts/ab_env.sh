#!/bin/bash
# A/B an environment switch in one box, interleaved: usage ab_env.sh VAR
for round in 1 2 3; do for v in 0 1; do
  echo -n "$1=$v: "; env $1=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('QPS %.0f kernel_ms %.3f ms/step %.3f recall %.4f' % (d['value'], d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['recall_at_10']))"
done; done
