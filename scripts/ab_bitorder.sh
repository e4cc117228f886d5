#!/bin/bash
# A/B (GPU box): cells of the visited bitmap's locality order (DR_BITORDER_P pivots grouped by DR_BITORDER_S super pivots): kernel ms of the c2 kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/bitorder.jsonl
for spec in "4096 64" "2048 64" "1304 64" "1024 32" "512 32" "4096 256" "8192 128" "15624 256"; do
  set -- $spec
  DR_BITORDER_P=$1 DR_BITORDER_S=$2 AB_KINDS=13,13 timeout 300 python scripts/ab_m1_waves.py "P$1_S$2" 30000 2>/dev/null >> gpurun_out/ab/bitorder.jsonl
done
python - <<PY
import json
for l in open("gpurun_out/ab/bitorder.jsonl"):
    r=json.loads(l); print(r["lib"], r["kernel_ms"], r["qps_resident"], r["results_sha1"])
PY
