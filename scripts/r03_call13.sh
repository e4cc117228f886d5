#!/bin/bash
# round 3, GPU call 13: the round's rocprofv3 evidence on the final build (kernel stats, PMC traffic, the bench line), SQ counters
# of the c2 kernel, single-query latency, sharded search with 4 local shards (round 2 vs now)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests13.log 2>&1; grep -E "passed|failed" $O/gputests13.log | tail -1
timeout 1500 bash scripts/profile_run.sh > $O/profile_run.log 2>&1; tail -3 $O/profile_run.log | cut -c1-400
timeout 600 bash scripts/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmcsq/sq_counters.json $O/ 2>/dev/null
timeout 300 python scripts/latency_single.py > $O/latency_single.log 2>&1; cp gpurun_out/latency_single_query.json $O/ 2>/dev/null
for lib in libdiskrag_hip_r02.so libdiskrag_hip.so; do echo "## $lib" >> $O/ab_sharded_4_local_shards.log; DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python scripts/ab_sharded.py 4 1000000 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> $O/ab_sharded_4_local_shards.log; done
cat $O/ab_sharded_4_local_shards.log
du -sh gpurun_out
