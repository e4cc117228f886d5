#!/bin/bash
# kernel breakdown of the exact-vector builder: c3 shape (2M x 1536, R 64, L_build 100) and c2 shape (1M x 128)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for cfg in "2097152 1536 64 100" "1000000 128 64 100"; do
  set -- $cfg; tag=${1}x${2}
  rm -rf gpurun_out/bprof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bprof -- python3 scripts/exp_build_profile.py $cfg > $O/build_profile_$tag.out 2> $O/build_profile_$tag.err
  cp $(ls gpurun_out/bprof/*/*kernel_stats.csv | head -1) $O/build_kernel_stats_$tag.csv; rm -rf gpurun_out/bprof
  cat $O/build_profile_$tag.out; cut -c1-150 $O/build_kernel_stats_$tag.csv | head -9
done
