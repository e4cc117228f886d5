#!/bin/bash
# repeat the GPU suite and an open/close loop: flakiness and leak check (GPU box)
for i in 1 2 3; do timeout 600 python -m pytest tests -m gpu -q --timeout=300 -x -p no:cacheprovider > /tmp/stress_$i.log 2>&1; echo "run $i rc=$? $(grep -E "passed|failed|error" /tmp/stress_$i.log | tail -1)"; done     # (RCCL prints its banner last: the summary is taken from the log)
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(100000, 128, n_queries=512, n_clusters=128, seed=3)
ref = None
for it in range(12):
    ix = HipIndex.create_empty(x, R=32)
    ix.build_vamana(L_build=50, alpha=1.2, passes=1, seed=7)
    ix.pq_encode(ix.pq_train(32, n_sample=20000, iters=2))
    out = ix.search_batch(q, 10, L=60, beam_width=8)
    if ref is None: ref = out
    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1].view(np.uint32), ref[1].view(np.uint32)), it
    ix.close()
import subprocess
print("open/close loop ok; rocm-smi VRAM:", subprocess.run(["rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout.strip().splitlines()[-3:])
PY
