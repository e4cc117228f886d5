#!/bin/bash
# exact prune, multi-pick form (DR_PRUNE_MULTI=1) against the plain form: parity tests under both, then build time and graph hash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
DR_PRUNE_MULTI=1 timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_build.py -q -x -k "prune_kernel_equals or reproducible or recall_and_oracle or graph_structure" > $O/multi_tests.log 2>&1; echo "multi pytest rc=$?"; grep -E "passed|failed|rror" $O/multi_tests.log | tail -3
for rep in 1 2; do
  for cfg in "1000000 128 64 100" "4000000 96 32 100"; do
    timeout 600 python scripts/exp_build_profile.py $cfg 2>&1 | grep -E "BUILD_S|GRAPH" | tr '\n' ' ' | sed 's/^/plain: /'; echo
    DR_PRUNE_MULTI=1 timeout 600 python scripts/exp_build_profile.py $cfg 2>&1 | grep -E "BUILD_S|GRAPH" | tr '\n' ' ' | sed 's/^/multi: /'; echo
  done
done 2>&1 | tee $O/ab_prune_multi.txt
