#!/bin/bash
# A/B of library builds on the c3 / c4 / c5-small shapes (GPU box): usage ab_shapes.sh "c3 1000000 -1" libA libB ...
SPEC="$1"; shift
for lib in "$@"; do
  echo "== $lib  [$SPEC]"
  DR_LIB=$PWD/$lib timeout 900 python scripts/ab_shape.py $SPEC 2>&1 | grep -v "^$" | tail -40
done
