#!/bin/bash
# PQ-only builder: prune with table rows in registers (8 waves/CU) against rows in LDS (2 waves/CU): same graph? how much faster?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round2.py -q -x -k "pq_only or copy_codes or builder" 2>&1 | tail -3
for rep in 1 2; do
  timeout 600 python scripts/exp_build_pq_profile.py 4194304 64 128 2>&1 | grep -E "BUILD_S|GRAPH" | sed 's/^/regs: /'
  DR_PQ_PRUNE_LDS=1 timeout 600 python scripts/exp_build_pq_profile.py 4194304 64 128 2>&1 | grep -E "BUILD_S|GRAPH" | sed 's/^/lds:  /'
done
