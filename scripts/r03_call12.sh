#!/bin/bash
# round 3, GPU call 12: build F (no store-acknowledgement wait on a prefetch hit) against build D: interleaved c2 bench (experiment G: drain before every expansion)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for round in 1 2 3; do for lib in libdiskrag_hip_d.so libdiskrag_hip.so; do
  echo -n "$lib: " >> $O/ab_top_wait_G.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; s=c['secondary_no_trim']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f kernel_ms %.4f recall %.4f | no trim kernel_ms %.4f recall %.4f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['kernel_ms_resident'], c['recall_at_10'], s['kernel_ms'], s['recall_at_10']))" >> $O/ab_top_wait_G.log
done; done
cat $O/ab_top_wait_G.log
