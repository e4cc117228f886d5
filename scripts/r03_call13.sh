#!/bin/bash
# round 3, final evidence run: GPU suite; rocprofv3 kernel stats + PMC traffic + the bench line (profile_run.sh); SQ counters of the
# c2 kernel; single-query latency
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests_final.log 2>&1; grep -E "passed|failed" $O/gputests_final.log | tail -1
timeout 1500 bash scripts/profile_run.sh > $O/profile_run.log 2>&1; tail -1 $O/profile_run.log | cut -c1-300
timeout 600 bash scripts/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmcsq/sq_counters.json $O/ 2>/dev/null
timeout 300 python scripts/latency_single.py > $O/latency_single.log 2>&1; cp gpurun_out/latency_single_query.json $O/ 2>/dev/null
du -sh gpurun_out
