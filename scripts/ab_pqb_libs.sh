#!/bin/bash
# interleaved A/B of DR_MODE_PQB over library builds (scripts/ab_pqb_libs.py): usage ab_pqb_libs.sh SHAPE N OUT lib1 lib2 ...
# (a library named "tree" is the in-tree build)
shape=$1; n=$2; out=$3; shift 3
d=/tmp/abpqb_${shape}_${n}
[ -f $d/meta.json ] || python scripts/ab_pqb_libs.py prep $shape $n $d > $out.prep 2>&1
for round in 1 2; do for lib in "$@"; do
  if [ "$lib" = tree ]; then unset DR_LIB; else export DR_LIB=$PWD/$lib; fi
  timeout 900 python scripts/ab_pqb_libs.py run $d $(basename $lib .so) 2>>$out.err >> $out
done; done
