#!/bin/bash
# round 3, GPU call 3: GPU suite on the new entry points (device k-means, cosine M3, dr_pq_scan_topk, sliced brute force,
# sharded search with persistent scratch, split-table builder), then the c5 graph-quality sweep at the full shard's density
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests3.log 2>&1; tail -15 $O/gputests3.log
timeout 2400 python scripts/c5_sweep.py 8388608 256 "32:64,64:128" exact > $O/c5_sweep_8M_256.log 2>&1
cp gpurun_out/c5_sweep_8388608_256.jsonl $O/
du -sh gpurun_out
