#!/bin/bash
# a wider graph on the PQ-only shard: 16M points, R = 128, L_build = 128 (candidates 128 + 128 + 64 = 320)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
C5_OUT=$O/scale_c5_16M_R128.json timeout 1500 python scripts/c5_shard.py 16777216 2097152 1000 "128:128" 32 > $O/c5_16M_lb.log 2>&1
tail -1 $O/c5_16M_lb.log | cut -c1-200
