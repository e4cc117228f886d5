#!/bin/bash
# round 3, GPU call 2: GPU test suite on the new build, c2 companions A/B, the default bench line, c5-shape A/B of the
# table-build kernel and the split-table variant against the round-2 library, phase shares, PMC of the mode-5 kernel
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
for lib in libdiskrag_hip_r02.so libdiskrag_hip.so; do
  echo "## $lib" >> $O/ab_c5s_4M.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 900 python scripts/ab_shape.py c5s 4000000 $([ $lib = libdiskrag_hip.so ] && echo "15 2" || echo "2") >> $O/ab_c5s_4M.log 2>&1
done
for env in "" "DR_NO_TREG=1"; do
  echo "## phase build $env" >> $O/phase_c5s_4M.txt
  env $env DR_LIB=$PWD/diskrag_amd/libdiskrag_hip_phase.so timeout 900 python scripts/exp_phase_c5.py 4000000 >> $O/phase_c5s_4M.txt 2>&1
done
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash scripts/ab_companions.sh > $O/ab_companions.log 2>&1
timeout 1500 bash scripts/pmc_shape.sh c5s 4000000 > $O/pmc_c5s.log 2>&1
cp gpurun_out/pmc_c5s/summary.json $O/pmc_c5s_4M.json
du -sh gpurun_out
