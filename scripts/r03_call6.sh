#!/bin/bash
# round 3, GPU call 6: GPU suite on the build with adaptive ballot counts; interleaved A/B (A = build before, B = with) on the
# c5 / c4 / c3 shapes; c2 companions with the leaner upload path; operating points of D = 256 / 768 / 960 at 1M points
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests6.log 2>&1; tail -4 $O/gputests6.log | head -2
for sh in "c5s 4000000 15" "c4 4000000 0" "c3 1000000 0"; do
  set -- $sh
  for lib in libdiskrag_hip_a.so libdiskrag_hip.so; do
    echo "## $lib $sh" >> $O/ab_ballot_counts.log
    DR_LIB=$PWD/diskrag_amd/$lib timeout 900 python scripts/ab_shape.py $1 $2 $3 >> $O/ab_ballot_counts.log 2>&1
  done
done
bash scripts/ab_companions.sh > $O/ab_companions_v2.log 2>&1
for d in d256 d768 d960; do
  timeout 900 python scripts/operating_points.py $d 1000000 10000 quick > $O/op_$d.log 2>&1
  cp gpurun_out/op_${d}_1000000.jsonl $O/ 2>/dev/null
done
du -sh gpurun_out
