#!/bin/bash
# final evidence of the round: the GPU suite, smoke(), then the rocprofv3 / PMC / bench run of scripts/profile_run.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
bash scripts/profile_run.sh 2>&1 | tail -5 | cut -c1-1500
