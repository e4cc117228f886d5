#!/bin/bash
# round 3, GPU call 11: where does the late prefetch correction lose? phase shares of builds D and E (issue at merge start), bench D vs E
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for v in d e; do echo "## phase build $v" >> $O/phase_c2_late_prefetch.txt; DR_LIB=$PWD/diskrag_amd/libdiskrag_hip_phase_$v.so timeout 300 python scripts/exp_phase.py 1000000 8 >> $O/phase_c2_late_prefetch.txt 2>&1; done
cat $O/phase_c2_late_prefetch.txt
for round in 1 2; do for lib in libdiskrag_hip_d.so libdiskrag_hip.so; do
  echo -n "$lib: " >> $O/ab_late_prefetch2.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('pipelined: value %.0f ms/batch %.4f kernel_ms %.4f | resident: qps %.0f kernel_ms %.4f recall %.4f' % (d['value'], c['ms_per_batch'], d['roofline']['kernel_ms'], c['qps_resident'], c['kernel_ms_resident'], c['recall_at_10']))" >> $O/ab_late_prefetch2.log
done; done
cat $O/ab_late_prefetch2.log
