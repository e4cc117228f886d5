#!/bin/bash
# control: the round-2 library against the current one on ONE box (is the 1.57-ms kernel of call 13 the box or the build?)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
/opt/rocm/bin/rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | head -30 > $O/box_state.txt
for round in 1 2; do for lib in libdiskrag_hip_s0.so libdiskrag_hip_f0.so libdiskrag_hip.so; do
  echo -n "$lib: " >> $O/ab_hit_wait_on_fast_base.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f kernel_ms %.4f recall %.4f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['kernel_ms_resident'], c['recall_at_10']))" >> $O/ab_hit_wait_on_fast_base.log
done; done
cat $O/ab_hit_wait_on_fast_base.log; cat $O/box_state.txt
