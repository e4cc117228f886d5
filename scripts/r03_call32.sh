#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 900 python scripts/exp_two_handles.py 2>&1 | grep -E "rep|one_handle|Error|error" | cut -c1-300
