#!/bin/bash
# A/B (GPU box; RECORD: runs only on commit 2cfb1e0, where variants 22 / 23 exist): the visited set of the c2 M1 kernel as a hash set of ids per wavefront
# slot (DR_VHASH=1) against the stamped bitmap (13 / 17). Result: profiles/r06/ab/ab_visited_id_hash_set.jsonl (bit-identical, 40-48 % slower); removed after.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/vhash.jsonl
export DR_VHASH=1
DR_VHASH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_live_regime.py tests/test_gpu_coalesce.py -x -q 2>&1 | tail -3
for r in 1 2; do for nq in 10000 30000; do
  AB_KINDS=13,22,13,22 timeout 300 python scripts/ab_m1_waves.py vhash $nq 2>/dev/null >> gpurun_out/ab/vhash.jsonl
done; done
python - <<PY
import json
for l in open("gpurun_out/ab/vhash.jsonl"):
    r=json.loads(l); print(r["nq"], r["forced_kind"], r["variant"], r["kernel_ms"], r["qps_resident"], r["results_sha1"], r["status"])
PY
for r in 1 2 3; do for vh in 0 1; do
  export DR_VHASH=$vh
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print(json.dumps({'vhash': $vh, 'value': d['value'], 'kernel_ms': d['roofline']['kernel_ms'], 'queries_per_launch': c['queries_per_launch'], 'variant': c['launch']['variant'], 'qps_resident': c['qps_resident'], 'recall': c['recall_at_10']}))"
done; done
