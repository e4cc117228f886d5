#!/usr/bin/env python3
"""Golden vectors for the LITERAL rerank policy A4 (search_engine.py:381-397): in the 0.8 - 1.2 band the reference flips a coin,
`np.random.random() < 0.2`, on numpy's global MT19937 stream (quirk Q2). The other goldens pin the two deterministic branches by
patching np.random.random; these run the reference UNPATCHED, with `np.random.seed(seed0 + qi)` called before query qi -- the
convention the engine's band_policy = 2 | seed0 << 8 restates (one query per request is how the reference is called; a batch has no
order of its own). Dev container only (needs /root/reference); writes DATA ONLY: coinflip_<fixture>.npz.

    python tests/golden/gen_golden_coinflip.py
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE))

import gen_golden as gg                      # noqa: E402
from diskrag_amd import persist              # noqa: E402
from tests.conftest import load_golden       # noqa: E402

CASES = [dict(L=100, bw=8, k=10, seed0=12345, f64=False), dict(L=40, bw=0, k=10, seed0=7, f64=False), dict(L=100, bw=8, k=10, seed0=99, f64=True),
         dict(L=20, bw=8, k=5, seed0=2024, f64=True)]


def main():
    work = gg.setup_reference()
    from search_engine import SearchEngineCorrect
    for name in ("unit1536_R16_m32", "unit1536_R16_m64", "deep96_R32_m16"):
        g = load_golden(name)
        if g.vectors.shape[1] not in (128, 256, 768, 960, 1536):
            # (the facade refuses other dimensions, Q14: the engine object is built around the files by hand, as gen_golden.py does)
            pass
        cname = "coinflip_" + name
        cdir = gg.make_collection(work, cname, g.vectors if g.vectors.shape[1] in (128, 256, 768, 960, 1536) else g.vectors)
        persist.write_index(cdir / "index", g.vectors, g.z["mem_adj"], g.medoid, R=g.R, degrees=g.z["deg"], codes=g.codes, codebook=g.codebook,
                            build_params={"L": 40}, pq_pickle=True)
        try:
            eng = SearchEngineCorrect(cname)
        except Exception as e:      # noqa: BLE001  (unsupported dimension)
            print(f"{name}: skipped ({e})")
            continue
        assert eng.use_pq
        out, meta = {}, []
        for ci, c in enumerate(CASES):
            qs = g.queries.astype(np.float64) if c["f64"] else g.queries
            ids = np.full((len(qs), c["k"]), 0xFFFFFFFF, np.uint32)
            dist = np.full((len(qs), c["k"]), np.nan, np.float64)
            cnt = np.zeros(len(qs), np.uint32)
            stats = np.zeros((len(qs), 4), np.uint32)
            draws = np.zeros(len(qs), np.uint32)
            orig = np.random.random
            for qi, q in enumerate(qs):
                np.random.seed(c["seed0"] + qi)
                n = [0]

                def counted():
                    n[0] += 1
                    return orig()
                np.random.random = counted
                try:
                    res, st = eng._pq_accelerated_graph_search(q, k=c["k"], L=c["L"], beam_width=c["bw"] or None)
                finally:
                    np.random.random = orig
                cnt[qi] = len(res)
                for t, (d, i) in enumerate(res):
                    ids[qi, t] = int(i); dist[qi, t] = float(d)
                stats[qi] = [st["search_steps"], st["nodes_visited"], st["exact_distance_computations"], st["pq_distance_computations"]]
                draws[qi] = n[0]
            out[f"c{ci}_ids"], out[f"c{ci}_dist"], out[f"c{ci}_count"], out[f"c{ci}_stats"], out[f"c{ci}_draws"] = ids, dist, cnt, stats, draws
            meta.append(c)
            print(f"{name} case {ci} {c}: coin flips per query {draws.mean():.1f} (max {draws.max()})")
        out["cases"] = np.array(json.dumps(meta))
        out["provenance"] = np.array(json.dumps(dict(generator="tests/golden/gen_golden_coinflip.py", numpy=np.__version__,
                                                     note="reference run UNPATCHED, np.random.seed(seed0 + qi) before query qi; index = the fixture idx_%s.npz" % name)))
        np.savez_compressed(HERE / f"coinflip_{name}.npz", **out)


if __name__ == "__main__":
    main()
