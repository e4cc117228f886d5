#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_facade.py -q -x > $O/facade_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" $O/facade_tests.log | tail -5
timeout 900 python scripts/exp_request_batcher.py > $O/request_batcher.log 2>&1; grep -E "^[0-9]+ \{|Error" $O/request_batcher.log | cut -c1-900
