#!/bin/bash
# what does the c2 search kernel cost without its companions? Mean search-kernel duration (HIP events
# around every launch, harvested by the library) of the bench workload
#   pipelined  = dr_search_submit/wait: per batch an upload, the bound kernels, the search kernel, finalize_kernel on the
#                second stream, the download
#   resident   = dr_batch_run on batches already in HBM: the search kernel + finalize_kernel only
# each with and without the tie-order pass (DR_SKIP_FINALIZE=1: timing experiment, tie order then wrong), and with the
# round-2 upload path (DR_SUBMIT_PERMUTE=1 DR_PQ_BOUND_BLOCK=1: permute_queries_kernel + block-per-query pq_bound_kernel).
# Interleaved. usage: ab_companions.sh [rounds]
cd "$GRAFT_REPO_ROOT"
for round in $(seq 1 ${1:-2}); do for cfg in "new 0" "new 1" "r2upload 0" "r2upload 1"; do
  set -- $cfg
  unset DR_SKIP_FINALIZE DR_SUBMIT_PERMUTE DR_PQ_BOUND_BLOCK
  [ $2 = 1 ] && export DR_SKIP_FINALIZE=1
  [ $1 = r2upload ] && export DR_SUBMIT_PERMUTE=1 DR_PQ_BOUND_BLOCK=1
  echo -n "round $round upload=$1 skip_finalize=$2: "
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f ms/batch %.4f kernel_ms %.4f | finalize_ms %.3f tied/batch %.1f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['ms_per_batch_resident'], c['kernel_ms_resident'], c['finalize_kernel_ms'], c.get('tied_queries_per_batch', -1)))"
done; done
