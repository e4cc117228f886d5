#!/bin/bash
# ONE c5 shard at its size with the widest graph the builder's prune takes: R = 128, L_build = 128 (16M-point runs: recall 0.97 at L = 100, beam_width 8)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
C5_OUT=$O/scale_c5_shard_R128.json timeout 3000 python scripts/c5_shard.py 125000000 4194304 1000 "128:128" > $O/c5_shard_R128.log 2>&1
tail -1 $O/c5_shard_R128.log | cut -c1-300
