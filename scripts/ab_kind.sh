#!/bin/bash
# A/B libraries in one box, interleaved, plus the float-query variant 11 of the first library as the in-box reference:
# usage ab_kind.sh lib0 libA ...
for round in 1 2 3; do
  for lib in "$@" "$1:11"; do
    k=""; l=$lib; case $lib in *:*) l=${lib%%:*}; k=${lib##*:};; esac
    if [ -n "$k" ]; then export DR_FORCE_KIND=$k; else unset DR_FORCE_KIND; fi
    echo -n "$lib: "; DR_LIB=$PWD/$l timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('QPS %.0f kernel_ms %.3f recall %.4f' % (d['value'], d['roofline']['kernel_ms'], d['config']['recall_at_10']))"
  done
done
