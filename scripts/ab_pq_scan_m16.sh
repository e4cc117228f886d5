#!/bin/bash
# interleaved A/B of the one-query flat PQ scan at m = 16 over library builds (block sizes / records in flight): usage ab_pq_scan_m16.sh OUT lib1 lib2 ...
# ("tree" = the in-tree build runs first in every round) -> JSON lines of scripts/bench_pq_scan.py 64000000 16 1 128
out=$1; shift
for round in 1 2 3; do for v in tree "$@"; do
  unset DR_LIB; [ "$v" = tree ] || export DR_LIB=$PWD/$v
  echo "{\"variant\": \"$(basename $v .so)\", \"round\": $round, \"m\": 16, \"run\": $(timeout 300 python scripts/bench_pq_scan.py 64000000 16 1 128 2>>$out.err)}" >> $out
done; done
