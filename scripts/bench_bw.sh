#!/bin/bash
for bw in 0 8; do echo -n "bw=$bw: "; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --bw $bw 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('QPS %.0f recall %.4f ms/step %.2f kernel_ms %.2f frac %.4f GB/s %.0f' % (d['value'], c['recall_at_10'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['achieved']))"; done
