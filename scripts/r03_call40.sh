#!/bin/bash
# an index whose rows do not fit in HBM (5e7 x 1536 = 307 GB) served from the host tier
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python scripts/host_tier_big.py 50000000 1000 > $O/host_tier_big.log 2>&1
echo "rc=$?"; grep -E "^PQ|^built|Error|error|Killed" $O/host_tier_big.log | cut -c1-400; tail -2 $O/host_tier_big.log | cut -c1-300
