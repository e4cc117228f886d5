#!/bin/bash
# k-means++ seeding on a column-major sample: time and codebook hash against the previous build
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for rep in 1 2; do
  for cfg in "262144 1536 32 50000 15" "1000000 128 32 100000 10" "1000000 96 16 100000 10"; do
    timeout 600 python scripts/exp_train_rate.py $cfg 2>&1 | grep TRAIN | sed 's/^/new:  /'
    DR_LIB=diskrag_amd/libdiskrag_hip_prev.so timeout 600 python scripts/exp_train_rate.py $cfg 2>&1 | grep TRAIN | sed 's/^/prev: /'
  done
done 2>&1 | tee $O/ab_train.txt
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_build.py -q -x > $O/train_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" $O/train_tests.log | tail -3
