#!/bin/bash
# round 3, GPU call 1: baselines on the round-2 build
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
bash scripts/ab_companions.sh > gpurun_out/r03/ab_companions.log 2>&1
for nq in 1250 2500 5000; do
  timeout 600 python bench.py --num-queries $nq --no-cpu --no-secondary > gpurun_out/r03/bench_nq$nq.json 2> gpurun_out/r03/bench_nq$nq.err
done
DR_LIB=$PWD/diskrag_amd/libdiskrag_hip_phase.so timeout 900 python scripts/exp_phase_c5.py 4000000 > gpurun_out/r03/phase_c5s_4M.txt 2>&1
timeout 1500 bash scripts/pmc_shape.sh c5s 4000000 > gpurun_out/r03/pmc_c5s.log 2>&1
cp gpurun_out/pmc_c5s/summary.json gpurun_out/r03/pmc_c5s_4M.json
