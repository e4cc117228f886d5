#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for ncl in 256 4096; do for lib in libdiskrag_hip_r02.so libdiskrag_hip.so; do
  echo -n "$lib: " >> $O/dbg_codebook.log
  DR_LIB=$PWD/diskrag_amd/$lib timeout 600 python scripts/dbg_codebook.py 1048576 $ncl >> $O/dbg_codebook.log 2>&1
done; done
cat $O/dbg_codebook.log
