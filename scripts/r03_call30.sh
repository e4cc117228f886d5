#!/bin/bash
# host tier of the stored vectors: parity tests, then the c3-shaped measurement against the HBM-resident index
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_host_tier.py -q -x > $O/host_tier_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" $O/host_tier_tests.log | tail -5
timeout 1200 python scripts/exp_host_tier.py 2097152 > $O/host_tier.log 2>&1; grep -E "^PQ|^M1|^M2|Error|error" $O/host_tier.log | cut -c1-400
