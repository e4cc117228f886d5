#!/usr/bin/env python3
"""Dev-container-only check (needs /root/reference; never runs on the GPU box, not collected by pytest):
an index directory written by diskrag_amd.persist.write_index is opened by the REFERENCE's SearchEngineCorrect
(its own readers: MMapNodeReader, load_pq_codes, load_pq_codebook -> pq_model.pkl) and its M1 search returns the
golden results recorded from the reference-written directory. Prints one line per fixture.

    python tests/golden/check_persist_with_reference.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE))

import gen_golden as gg                      # noqa: E402  (setup_reference, make_collection: harness helpers)
from diskrag_amd import persist              # noqa: E402
from tests.conftest import load_golden       # noqa: E402


def main():
    work = gg.setup_reference()
    from search_engine import SearchEngineCorrect
    ok = True
    for name in ("sift128_R16_m32", "randn128_R64_m16", "unit1536_R16_m32"):
        g = load_golden(name)
        cname = "persist_" + name
        cdir = gg.make_collection(work, cname, g.vectors)
        persist.write_index(cdir / "index", g.vectors, g.z["mem_adj"], g.medoid, R=g.R, degrees=g.z["deg"],
                            codes=g.codes, codebook=g.codebook, build_params={"L": 40}, pq_pickle=True)
        eng = SearchEngineCorrect(cname)
        assert eng.use_pq, "reference fell back to exact mode: pq_model.pkl not accepted"
        bad = 0
        ncase = 0
        for ci in range(len(g.cases)):
            c = g.case(ci)
            if c["mode"] != "M1" or c.get("f64") or c.get("policy", 0) != 0:
                continue
            ncase += 1
            orig = np.random.random
            np.random.random = lambda: 0.0       # band policy 0, as the generator pins it
            try:
                for qi in range(min(8, g.queries.shape[0])):
                    res, _ = eng._pq_accelerated_graph_search(g.queries[qi], k=c["k"], L=c["L"], beam_width=c["bw"] or None)
                    ids = [int(i) for _, i in res]
                    want = [int(i) for i in c["ids"][qi][:c["count"][qi]]]
                    db = np.array([d for d, _ in res], dtype=np.float32).view(np.uint32)
                    if ids != want or not np.array_equal(db, c["dist"][qi][:c["count"][qi]].view(np.uint32)):
                        bad += 1
            finally:
                np.random.random = orig
        print(f"{name}: reference engine on a write_index directory, {ncase} M1 cases x 8 queries, mismatches: {bad}")
        ok = ok and bad == 0 and ncase > 0
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
