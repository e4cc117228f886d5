#!/bin/bash
# round 3, GPU call 5: one c5 shard at its full size (1.25e8 x 1536, PQ-only), two graphs over one code table, exact + ADC ground
# truth on 1000 queries; a 2M-point dry run of the same script first
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
C5_OUT=$O/scale_c5_dryrun_2M.json timeout 600 python scripts/c5_shard.py 2097152 1048576 200 "32:64,64:128" > $O/c5_dryrun.log 2>&1 || { tail -20 $O/c5_dryrun.log; exit 1; }
tail -2 $O/c5_dryrun.log | cut -c1-600
C5_OUT=$O/scale_c5_shard.json timeout 3300 python scripts/c5_shard.py 125000000 4194304 1000 "32:64,64:128" > $O/c5_shard.log 2>&1
tail -5 $O/c5_shard.log | cut -c1-800
du -sh gpurun_out
