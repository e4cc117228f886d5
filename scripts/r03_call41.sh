#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round2.py -q -x -k "prune or pq_only or widest" > gpurun_out/prune_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror|assert" gpurun_out/prune_tests.log | tail -8
