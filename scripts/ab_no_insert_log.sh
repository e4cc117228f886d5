cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for r in 1 2 3; do for nolog in 0 1; do
  export DR_SKIP_FINALIZE=1; if [ $nolog = 1 ]; then export DR_NO_LOG=1; else unset DR_NO_LOG; fi
  timeout 300 python scripts/ab_m1_waves.py nolog$nolog 20000 2>/dev/null | grep '"forced_kind": 13' >> gpurun_out/ab/no_insert_log.jsonl
done; done
cat gpurun_out/ab/no_insert_log.jsonl
