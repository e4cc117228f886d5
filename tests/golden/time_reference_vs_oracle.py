#!/usr/bin/env python3
"""Dev-container-only (needs /root/reference; not collected by pytest): how fast is the REFERENCE's own M1 search
against the oracle's C restatement of it, one thread each, same index, same queries? SURVEY 8d asks for this ratio
beside the GPU numbers (the reference's Python cannot travel to the GPU box, the restatement can).

    python tests/golden/time_reference_vs_oracle.py
"""
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE))

import gen_golden as gg                      # noqa: E402
from diskrag_amd import persist              # noqa: E402
from oracle import pyoracle as orc           # noqa: E402
from tests.conftest import load_golden       # noqa: E402


def main():
    work = gg.setup_reference()
    from search_engine import SearchEngineCorrect
    for name in ("sift128_R64_m32", "unit1536_R16_m32"):
        g = load_golden(name)
        cname = "timing_" + name
        cdir = gg.make_collection(work, cname, g.vectors)
        persist.write_index(cdir / "index", g.vectors, g.z["mem_adj"], g.medoid, R=g.R, degrees=g.z["deg"],
                            codes=g.codes, codebook=g.codebook, pq_pickle=True)
        eng = SearchEngineCorrect(cname)
        q = g.queries
        orig = np.random.random
        np.random.random = lambda: 0.0
        try:
            t0 = time.perf_counter()
            for qi in range(len(q)):
                eng._pq_accelerated_graph_search(q[qi], k=10, L=100, beam_width=8)
            t_ref = time.perf_counter() - t0
        finally:
            np.random.random = orig
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.search_batch(g.vectors, g.adj, q, g.medoid, orc.M1, 10, L=100, bw=8, codes=g.codes, codebook=g.codebook,
                             nthreads=1)
        t_orc = (time.perf_counter() - t0) / reps
        print(f"{name}: N={len(g.vectors)} D={g.vectors.shape[1]} queries={len(q)}  reference {len(q) / t_ref:8.1f} QPS/thread   "
              f"oracle {len(q) / t_orc:9.1f} QPS/thread   ratio {t_ref / t_orc:6.1f}x")


if __name__ == "__main__":
    main()
