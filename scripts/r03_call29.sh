#!/bin/bash
# where a one-query call spends its 0.45 ms: device spans, the C ABI alone, and the HIP API calls behind it
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O; rm -rf gpurun_out/ltrace
timeout 600 python scripts/latency_single.py > $O/latency_breakdown.json 2> $O/latency_breakdown.err
cat $O/latency_breakdown.json
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/ltrace -- python3 scripts/latency_single.py > /dev/null 2> $O/latency_trace.err
cp $(ls gpurun_out/ltrace/*/*hip_api_stats.csv | head -1) $O/latency_hip_api_stats.csv
cp $(ls gpurun_out/ltrace/*/*kernel_stats.csv | head -1) $O/latency_kernel_stats.csv
rm -rf gpurun_out/ltrace
head -16 $O/latency_hip_api_stats.csv | cut -c1-120; grep -E "search_kernel|finalize|permute|pq_bound|f64" $O/latency_kernel_stats.csv | cut -c1-160
