#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for rep in 1 2; do
  for cfg in "262144 1536 32 50000 15" "1000000 128 32 100000 10" "1000000 96 16 100000 10"; do
    timeout 600 python scripts/exp_train_rate.py $cfg 2>&1 | grep -E "TRAIN|Error" | sed 's/^/new:  /'
    DR_LIB=diskrag_amd/libdiskrag_hip_prev.so timeout 600 python scripts/exp_train_rate.py $cfg 2>&1 | grep -E "TRAIN|Error" | sed 's/^/prev: /'
  done
done 2>&1 | tee $O/ab_train.txt
