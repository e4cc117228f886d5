#!/bin/bash
# kernel breakdown of the PQ-only builder at the degree of the full-size shard: 8M points, R = 128, L_build = 128
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O; rm -rf gpurun_out/bprof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bprof -- python3 scripts/exp_build_pq_profile.py 8388608 128 128 > $O/build_pq_profile_R128.out 2> $O/build_pq_profile_R128.err
cp $(ls gpurun_out/bprof/*/*kernel_stats.csv | head -1) $O/build_pq_kernel_stats_8M_R128_L128.csv; rm -rf gpurun_out/bprof
cat $O/build_pq_profile_R128.out; cut -c1-150 $O/build_pq_kernel_stats_8M_R128_L128.csv | head -8
