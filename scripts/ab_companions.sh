#!/bin/bash
# what does the c2 search kernel cost without its companions? Mean search-kernel duration (HIP events
# around every launch, harvested by the library) of the bench workload
#   pipelined  = dr_search_submit/wait: per batch an upload, permute_queries_kernel, pq_bound_kernel, the search kernel,
#                finalize_kernel on the second stream, the download
#   resident   = dr_batch_run on batches already in HBM: the search kernel + finalize_kernel only
# each with and without the tie-order pass (DR_SKIP_FINALIZE=1: timing experiment, tie order then wrong).
# Interleaved, 3 rounds. usage: ab_companions.sh  -> gpurun_out/ab_companions.log
cd "$GRAFT_REPO_ROOT"
for round in 1 2 3; do for skip in 0 1; do
  if [ $skip = 1 ]; then export DR_SKIP_FINALIZE=1; else unset DR_SKIP_FINALIZE; fi
  echo -n "round $round skip_finalize=$skip: "
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f ms/batch %.4f kernel_ms %.4f | finalize_ms %.3f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['ms_per_batch_resident'], c['kernel_ms_resident'], c['finalize_kernel_ms']))"
done; done
