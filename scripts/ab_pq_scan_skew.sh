#!/bin/bash
# interleaved A/B of the one-query flat PQ scan (round 6): the skewed kernel (in-tree build; 'contiguous': every stream of the scan-order copy in one
# piece, DR_PQ_SCAN_CONTIGUOUS_STREAMS=1) against pq_scan_kernel (DR_PQ_SCAN_NO_SKEW=1) and
# against other builds of the library (given as DR_LIB paths: the round's variants were builds with other block sizes, records in flight and load kinds).
# usage: ab_pq_scan_skew.sh OUT [lib ...]     -> JSON lines (scripts/bench_pq_scan.py: 64M code words, one query, m = 32, 16 and 64)
out=$1; shift
for round in 1 2; do
  for v in skew contiguous noskew "$@"; do
    unset DR_LIB DR_PQ_SCAN_NO_SKEW DR_PQ_SCAN_CONTIGUOUS_STREAMS
    case $v in skew) ;; contiguous) export DR_PQ_SCAN_CONTIGUOUS_STREAMS=1 ;; noskew) export DR_PQ_SCAN_NO_SKEW=1 ;; *) export DR_LIB=$PWD/$v ;; esac
    ms="32 16 64"
    for m in $ms; do
      echo "{\"variant\": \"$(basename $v .so)\", \"round\": $round, \"m\": $m, \"run\": $(timeout 600 python scripts/bench_pq_scan.py 64000000 $m 1 128 2>>$out.err)}" >> $out
    done
  done
done
