#!/bin/bash
# round-2 A/B: GPU parity suite under the candidate library, then interleaved bench runs. usage: ab_r2.sh paritylib libA libB ...
plib=$1; shift
if [ "$plib" != "-" ]; then echo "== parity under $plib"; DR_LIB=$PWD/$plib timeout 900 python -m pytest tests -m gpu -q --timeout=120 -x 2>&1 | tail -4; fi
bash scripts/ab.sh "$@"
