#!/bin/bash
# the quantiser's ceiling on a PQ-only shard: m = 32 against m = 64 at 16M points (same generator, R = 64, L_build = 128)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for m in 32 64; do
  C5_OUT=$O/scale_c5_16M_m$m.json timeout 1500 python scripts/c5_shard.py 16777216 2097152 1000 "64:128" $m > $O/c5_16M_m$m.log 2>&1
  tail -1 $O/c5_16M_m$m.log | cut -c1-200
done
