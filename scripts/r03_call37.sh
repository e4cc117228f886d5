#!/bin/bash
# graph degree on the c3 shape: 2M x 1536, R = 64 against R = 128 (same data, same sweeps)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for r in 64 128; do
  OP_R=$r timeout 900 python scripts/operating_points.py c3 2097152 10000 quick > $O/op_c3_2M_R$r.log 2>&1
  grep -E '"setup"|PQ_rerank|M1_L(100|200)_bw(8|None)_policy0|M2_bw64|error' $O/op_c3_2M_R$r.log | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l[:200]); continue
    if 'setup' in d: print('R', d['setup']['R'], 'build_s %.1f' % d['setup']['build_s'])
    elif 'error' in d: print(d)
    else: print('  %-28s qps %9.0f recall %.4f kernel_ms %.3f steps %.1f' % (d['run'], d['qps'], d['recall_at_10'], d['kernel_ms'], d['steps']))
"
done
cp gpurun_out/op_c3_2097152.jsonl $O/op_c3_2M_R64.jsonl; cp gpurun_out/op_c3_2097152_R128.jsonl $O/op_c3_2M_R128.jsonl
