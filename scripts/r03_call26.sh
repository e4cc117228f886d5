#!/bin/bash
# ONE c5 shard at its size on the final tree (register-row prune in the builder, final search kernels): R = 64, L_build = 128
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
C5_OUT=$O/scale_c5_shard_final.json timeout 2700 python scripts/c5_shard.py 125000000 4194304 1000 "64:128" > $O/c5_shard_final.log 2>&1
tail -1 $O/c5_shard_final.log | cut -c1-300
