"""Stress (GPU box): P processes sharing ONE GPU build the same small index over and over (bench.py --gpus 8 on a one-GPU pool does exactly that);
the builder is deterministic, so every build of every process must give the same adjacency. usage: stress_concurrent_builds.py [P=8] [rounds=6] [N=30000]"""
import hashlib
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    sys.path.insert(0, ".")
    from diskrag_amd import HipIndex
    from diskrag_amd.synth import sift_like
    rounds, n = int(sys.argv[2]), int(sys.argv[3])
    x, q = sift_like(n, 128, n_queries=100, n_clusters=1024, seed=2024, query_seed=9000)
    for r in range(rounds):
        try:
            ix = HipIndex.create_empty(x, R=64)
            ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7, pad_with_zero=True)
            adj = ix.get_adjacency()
            print("ok", hashlib.sha1(adj.tobytes()).hexdigest()[:12], int((adj >= n).sum()), flush=True)
            ix.close()
        except Exception as e:      # noqa: BLE001
            print("FAIL", str(e)[:200], flush=True)
            try:        # what the bad rows look like
                adj = ix.get_adjacency()
                bad = np.argwhere(adj >= n)
                rows = np.unique(bad[:, 0])
                print("DIAG rows", rows[:10].tolist(), "n_rows", len(rows), "values", np.unique(adj[adj >= n])[:5].tolist(), "cols", np.unique(bad[:, 1])[:70].tolist(), flush=True)
                for r_ in rows[:2]:
                    print("DIAG row", int(r_), adj[r_].tolist(), flush=True)
            except Exception as e2:      # noqa: BLE001
                print("DIAG failed", str(e2)[:100], flush=True)
    sys.exit(0)

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
procs = [subprocess.Popen([sys.executable, __file__, "child", str(rounds), str(n)], stdout=subprocess.PIPE, text=True) for _ in range(P)]
lines = []
for p in procs:
    out, _ = p.communicate()
    lines += out.strip().splitlines()
from collections import Counter
print(Counter(l for l in lines if not l.startswith("DIAG")))
for l in lines:
    if l.startswith("DIAG"):
        print(l[:1500])
