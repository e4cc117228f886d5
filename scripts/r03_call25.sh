#!/bin/bash
# kernel breakdown of the PQ-only builder after the register-row prune (same command as r03_call22.sh)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O; rm -rf gpurun_out/bprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bprof -- python3 scripts/exp_build_pq_profile.py 8388608 64 128 > $O/build_pq_profile_regs.out 2> $O/build_pq_profile_regs.err
cp $(ls gpurun_out/bprof/*/*kernel_stats.csv | head -1) $O/build_pq_kernel_stats_8M_R64_L128_regs.csv; rm -rf gpurun_out/bprof
cat $O/build_pq_profile_regs.out; cut -c1-160 $O/build_pq_kernel_stats_8M_R64_L128_regs.csv | head -14
