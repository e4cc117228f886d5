#!/bin/bash
# the GPU suite, summary line last (RCCL prints its banner at exit: the pytest summary is taken from the log, not from the tail)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3
