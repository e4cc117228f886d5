#!/bin/bash
# round 3, GPU call 9: build D (cached worst distance, branch-selected list reads) against build B: GPU suite, c2 bench, c5 / c4 shapes
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gputests9.log 2>&1; grep -E "passed|failed" $O/gputests9.log | tail -1
for round in 1 2 3; do for lib in libdiskrag_hip_b.so libdiskrag_hip.so; do
  echo -n "$lib: " >> $O/ab_pop.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f kernel_ms %.4f recall %.4f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['kernel_ms_resident'], c['recall_at_10']))" >> $O/ab_pop.log
done; done
for sh in "c5s 4000000 15" "c4 4000000 0"; do
  set -- $sh
  for lib in libdiskrag_hip_b.so libdiskrag_hip.so; do
    echo "## $lib $sh" >> $O/ab_pop.log
    DR_LIB=$PWD/diskrag_amd/$lib timeout 900 python scripts/ab_shape.py $1 $2 $3 >> $O/ab_pop.log 2>&1
  done
done
cut -c1-230 $O/ab_pop.log
