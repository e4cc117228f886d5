#!/bin/bash
# encode kernels with the sub-vector in registers: rate and code-word hash against the previous build; GPU suite; exact-builder profile
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for rep in 1 2; do
  for cfg in "2097152 1536 32" "4000000 128 32" "4000000 96 16"; do
    timeout 600 python scripts/exp_encode_rate.py $cfg 2>&1 | grep ENCODE | sed 's/^/new:  /'
    DR_LIB=diskrag_amd/libdiskrag_hip_prev.so timeout 600 python scripts/exp_encode_rate.py $cfg 2>&1 | grep ENCODE | sed 's/^/prev: /'
  done
done 2>&1 | tee $O/ab_encode.txt
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
bash scripts/r03_call27.sh
