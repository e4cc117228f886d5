"""Golden vectors for the in-memory M3 with distance_metric='cosine' (beam_search_with_pq, pydiskann/vamana_graph.py:535-605;
compute_query_distance -> cosine_similarity_cython, :324-329, cython_utils.pyx:53-70). Dev container only: imports the
REFERENCE from a scratch copy (the recipe of gen_golden.py) and writes DATA only -> tests/golden/cos_<fixture>.npz.
The graph is the committed fixture's in-memory graph (mem_adj rows, set iteration order as dumped by gen_golden.py) handed
to the reference as Node objects whose `neighbors` are lists in that order, so the traversal order is the fixture's."""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import gen_golden as gg          # noqa: E402  (setup_reference, pack_results)

CASES = [dict(bw=8, k=5), dict(bw=16, k=10), dict(bw=64, k=10), dict(bw=3, k=3)]


def main():
    gg.setup_reference()
    from pydiskann.vamana_graph import VamanaGraphWithPQ, Node, beam_search_with_pq
    for data, idx in (("randn128", "randn128_R16_m32"), ("unit1536", "unit1536_R16_m32"), ("deep96", "deep96_R32_m16")):
        d = np.load(HERE / f"data_{data}.npz")
        z = np.load(HERE / f"idx_{idx}.npz")
        x, queries = d["vectors"], d["queries"]
        g = VamanaGraphWithPQ(int(z["R"]), None, distance_metric="cosine")
        for i in range(len(x)):
            nd = Node(i, x[i])
            nd.neighbors = [int(v) for v in z["mem_adj"][i][:int(z["deg"][i])]]
            g.nodes[i] = nd
        g.medoid_idx = int(z["medoid"])
        out = {}
        import warnings
        for ci, c in enumerate(CASES):
            res = []
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")          # sqrt of a cosine distance that rounded below zero
                for q in queries:
                    res.append(beam_search_with_pq(g, q, start_idx=None, beam_width=c["bw"], k=c["k"], use_pq=False))
            ids, dist, dist64, cnt = gg.pack_results(res, c["k"])
            out[f"c{ci}_ids"], out[f"c{ci}_dist"], out[f"c{ci}_count"] = ids, dist, cnt
        out["cases"] = np.array(json.dumps(CASES))
        out["provenance"] = np.array(json.dumps(dict(generator="tests/golden/gen_golden_cosine.py", fixture=idx, numpy=np.__version__,
                                                     note="expected values produced by the reference at /root/reference")))
        np.savez_compressed(HERE / f"cos_{idx}.npz", **out)
        print(f"[golden] cosine {idx}: {len(queries)} queries, {len(CASES)} cases")


if __name__ == "__main__":
    main()
