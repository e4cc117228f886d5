#!/bin/bash
# the c5 shard's operating point, finer: R = 128 graph at full size, (L, beam_width) around L = 200 / beam_width 8
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
C5_GRID="200:8,150:8,150:16,200:16,125:0,150:0,100:16,250:8,300:8,100:32,150:32,200:32" C5_OUT=$O/scale_c5_shard_R128_fine.json timeout 1500 python scripts/c5_shard.py 125000000 4194304 1000 "128:128" > $O/c5_shard_R128_fine.log 2>&1
echo "rc=$?"; tail -1 $O/c5_shard_R128_fine.log | cut -c1-200
