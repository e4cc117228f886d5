"""Measurement (GPU box): the disk tier of the rows (dr_index_attach_row_file) against rows in HBM -- PQ traversal (DR_MODE_PQB) + exact rerank of the
L = 100 list on a 200k x 1536 unit-norm index written to index.dat in the reference's record layout; batches of 1 / 64 / 2000 queries, recall@10 against
the exact neighbours, O_DIRECT and buffered reads. -> one JSON object. usage: exp_disk_tier.py [directory for index.dat]"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import unit_mixture
N, D, R, m = 200000, 1536, 64, 32
x, q = unit_mixture(N, D, n_queries=2000, n_clusters=256, seed=5, latent=32)
full = HipIndex.create_empty(x, R=R)
medoid, _ = full.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = full.pq_train(m, n_sample=50000, iters=5); codes = full.pq_encode(cb, want_codes=True)
adj = full.get_adjacency()
d = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp()
path = os.path.join(d, "index.dat")
rec = np.empty((N, D + R), dtype=np.uint32); rec[:, :D] = x.view(np.uint32); rec[:, D:] = adj; rec.tofile(path); del rec
gt, _ = full.bruteforce_topk(q, 10)
out = {"index": "%d x %d unit-norm, R %d, m %d; index.dat %.2f GB at %s" % (N, D, R, m, os.path.getsize(path) / 1e9, d)}
def run(ix, nq, reps):
    ix.search_batch(q[:nq], 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK)
    ts = []
    for r in range(reps):
        t0 = time.perf_counter(); res = ix.search_batch(q[:nq], 10, L=100, beam_width=8, mode=_ffi.MODE_PQB, flags=_ffi.F_RERANK); ts.append(time.perf_counter() - t0)
    rec10 = float(np.mean([len(set(res[0][i]) & set(gt[i])) / 10.0 for i in range(nq)]))
    return {"ms_per_call_median": round(float(np.median(ts)) * 1e3, 3), "qps": round(nq / float(np.median(ts))), "recall_at_10": round(rec10, 4)}
for nq, reps in ((1, 50), (64, 20), (2000, 5)):
    out.setdefault("rows_in_hbm", {})["nq%d" % nq] = run(full, nq, reps)
for tag, env in (("disk_tier_o_direct", None), ("disk_tier_buffered", "1")):
    shard = HipIndex.create_codes(adj, medoid, D, cb, codes)
    if env: os.environ["DR_ROW_FILE_BUFFERED"] = env
    shard.attach_row_file(path)
    os.environ.pop("DR_ROW_FILE_BUFFERED", None)
    for nq, reps in ((1, 50), (64, 20), (2000, 5)):
        out.setdefault(tag, {})["nq%d" % nq] = run(shard, nq, reps)
    shard.close()
print(json.dumps(out, indent=1))
