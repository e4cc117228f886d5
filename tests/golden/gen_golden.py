#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE implementation (read-only at
/root/reference) in a scratch directory under /tmp.

This script is the only place the reference is imported. It never copies reference
sources into the repo: it copies them to a /tmp scratch dir, builds the Cython
extension there, installs harness-side stubs for modules absent from this image
(numba, polars, openai: SURVEY.md section 8c), runs the reference search functions on
seeded synthetic data and writes DATA ONLY (inputs + expected outputs) as .npz files
next to this script.

Run (dev container only; the GPU box has no /root/reference):
    python tests/golden/gen_golden.py [--only NAME]

Reference functions driven (file:line are into /root/reference):
  M1  SearchEngineCorrect._pq_accelerated_graph_search   search_engine.py:398-506
  M2  beam_search_from_disk                               pydiskann/vamana_graph.py:719-760
  M3  beam_search_with_pq                                 pydiskann/vamana_graph.py:535-605
  M4  greedy_search / greedy_search_cython                vamana_graph.py:607-640, cython_utils.pyx:72-122
  A1  _compute_exact_distance                             search_engine.py:374-379
  A2  DiskANNPQ.compute_distance_table                    pydiskann/pq/fast_pq.py:294-318
  A3  DiskANNPQ.asymmetric_distance[_sq]                  pydiskann/pq/fast_pq.py:320-333
  C8  l2_distance_fast_cython / cosine_similarity_cython  pydiskann/cython_utils.pyx:18-70
"""
import argparse
import json
import logging
import os
import shutil
import subprocess
import sys
import time
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
SCRATCH = Path("/tmp/diskrag_golden_scratch")
PAD = np.uint32(0xFFFFFFFF)


def setup_reference():
    if not REF.exists():
        sys.exit("reference not present: golden vectors can only be generated in the dev container")
    work = SCRATCH / "ref"
    if not (work / "pydiskann").exists():
        work.mkdir(parents=True, exist_ok=True)
        for item in ["pydiskann", "preprocessing", "scripts", "search_engine.py"]:
            src = REF / item
            dst = work / item
            if src.is_dir():
                shutil.copytree(src, dst)
            else:
                shutil.copy(src, dst)
    so = list((work / "pydiskann").glob("cython_utils*.so"))
    if not so:
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"],
                              cwd=work / "pydiskann", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # harness-side stubs for modules missing from the image (no reference file is modified)
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)
    sys.modules["numba"] = nb
    pl = types.ModuleType("polars")
    pl.DataFrame = object
    pl.String = str
    sys.modules["polars"] = pl
    oa = types.ModuleType("openai")
    oa.OpenAI = object
    sys.modules["openai"] = oa
    os.chdir(work)
    sys.path.insert(0, str(work))
    logging.disable(logging.CRITICAL)
    return work


# ----------------------------------------------------------------------------- datasets

def ds_randn(seed, n, d, nq):
    rs = np.random.RandomState(seed)
    x = rs.randn(n, d).astype(np.float32)
    q = rs.randn(nq, d).astype(np.float32)
    return x, q


def ds_siftlike(seed, n, d, nq, ncl=32, step=4.0, dup_frac=0.15):
    """Integer-valued clustered data in [0, 218] (SIFT value range), quantised to multiples of `step`, with a
    fraction of exact duplicate points: exact f32 distance ties (same distance, different id) are common."""
    rs = np.random.RandomState(seed)
    cent = rs.randn(ncl, d) * 1.0
    def draw(cnt):
        a = rs.randint(0, ncl, size=cnt)
        p = cent[a] + 0.5 * rs.randn(cnt, d)
        p = (p - (-4.0)) / 8.0 * 218.0
        return (np.clip(np.rint(p / step) * step, 0, 218)).astype(np.float32)
    x = draw(n)
    ndup = int(n * dup_frac)
    dst = rs.choice(n, size=ndup, replace=False)
    src = rs.randint(0, n, size=ndup)
    x[dst] = x[src]
    return x, draw(nq)


def ds_unit(seed, n, d, nq, ncl=16, noise=0.7):
    """Unit-norm clustered data (text-embedding-like): squared distances < 2, Q1/Q2 band is live."""
    rs = np.random.RandomState(seed)
    cent = rs.randn(ncl, d)
    def draw(cnt):
        a = rs.randint(0, ncl, size=cnt)
        p = cent[a] + noise * rs.randn(cnt, d)
        p /= np.linalg.norm(p, axis=1, keepdims=True)
        return p.astype(np.float32)
    return draw(n), draw(nq)


# ----------------------------------------------------------------------------- helpers

def make_collection(work, name, x):
    cdir = work / "collections" / name
    if cdir.exists():
        shutil.rmtree(cdir)
    (cdir / "index").mkdir(parents=True)
    np.save(cdir / "vectors.npy", x)
    info = dict(name=name, config={}, dimension=int(x.shape[1]), num_vectors=int(x.shape[0]),
                created_at="2025-01-01T00:00:00", updated_at="2025-01-01T00:00:00", source_files=[],
                text_hashes=[], vector_offsets={}, chunk_stats={})
    (cdir / "collection_info.json").write_text(json.dumps(info))
    return cdir


def pack_results(res_list, k):
    """list (per query) of [(dist, id)] -> ids[nq,k] (PAD), dist[nq,k] f32 (nan pad), count[nq]"""
    nq = len(res_list)
    ids = np.full((nq, k), PAD, dtype=np.uint32)
    dist = np.full((nq, k), np.nan, dtype=np.float32)
    dist64 = np.full((nq, k), np.nan, dtype=np.float64)
    cnt = np.zeros(nq, dtype=np.uint32)
    for i, r in enumerate(res_list):
        cnt[i] = len(r)
        for j, (d, n) in enumerate(r):
            ids[i, j] = n
            dist[i, j] = np.float32(d)
            dist64[i, j] = np.float64(d)
    return ids, dist, dist64, cnt


def build_index_fixture(work, name, x, queries, R, Lb, alpha, m, seed, cases, with_mem_modes=True,
                        via_engine=True, k1_nodes=48):
    """Build PQ + graph with the reference, persist with the reference writer, run the cases."""
    from pydiskann.pq.fast_pq import DiskANNPQ
    from pydiskann.vamana_graph import (build_vamana_with_pq, beam_search_from_disk, beam_search_with_pq,
                                        greedy_search, compute_query_distance)
    from pydiskann.io.diskann_persist import DiskANNPersist, MMapNodeReader
    from pydiskann.cython_utils import greedy_search_cython
    import random as pyrandom

    t0 = time.time()
    n, d = x.shape
    np.random.seed(seed)
    pyrandom.seed(seed)
    pq = None
    codes = None
    if m:
        pq = DiskANNPQ(m, 256)
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            pq.fit(x)
        codes = pq.encode(x)
    graph = build_vamana_with_pq(x, pq, R=R, L=Lb, alpha=alpha, use_pq_in_build=False)
    medoid = int(graph.medoid_idx)

    cdir = make_collection(work, name, x)
    persist = DiskANNPersist(dim=d, R=R)
    persist.save_index(str(cdir / "index" / "index.dat"), graph)
    meta = dict(D=d, R=R, L=Lb, alpha=alpha, N=n, medoid_idx=medoid, n_subvectors=m or 0, pq_centroids=256,
                use_pq=bool(m))
    persist.save_meta(str(cdir / "index" / "meta.json"), meta)
    if m:
        persist.save_pq_codes(str(cdir / "index" / "pq_codes.bin"), codes)
        persist.save_pq_codebook(str(cdir / "index" / "pq_model.pkl"), pq)

    raw = np.fromfile(cdir / "index" / "index.dat", dtype=np.uint8)
    assert raw.size == n * 4 * (d + R)
    rec = raw.reshape(n, 4 * (d + R))
    vec_disk = rec[:, :4 * d].copy().view(np.float32).reshape(n, d)
    adj_disk = rec[:, 4 * d:].copy().view(np.uint32).reshape(n, R)
    assert np.array_equal(vec_disk, x)
    # in-memory neighbour order (python set iteration), PAD-padded: M3/M4 iterate this, no phantom 0
    mem_adj = np.full((n, R), PAD, dtype=np.uint32)
    deg = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        nb = list(graph.nodes[i].neighbors)
        deg[i] = len(nb)
        mem_adj[i, :len(nb)] = nb[:R]
        assert len(nb) <= R
        assert np.array_equal(adj_disk[i, :len(nb)], mem_adj[i, :len(nb)])

    out = dict(adj=adj_disk, mem_adj=mem_adj, deg=deg, medoid=np.uint32(medoid), R=np.uint32(R),
               m=np.uint32(m or 0))
    if m:
        cb = np.stack([km.cluster_centers_ for km in pq.kmeans_list]).astype(np.float32)
        assert cb.dtype == np.float32 and pq.kmeans_list[0].cluster_centers_.dtype == np.float32
        out["codebook"] = cb
        out["codes"] = codes.astype(np.uint8)

    engine = None
    if via_engine:
        from search_engine import SearchEngineCorrect
        engine = SearchEngineCorrect(name)
    reader = MMapNodeReader(str(cdir / "index" / "index.dat"), dim=d, R=R)

    # ---- K1: kernel-level known answers
    rs = np.random.RandomState(seed + 7)
    nodes = rs.choice(n, size=min(k1_nodes, n), replace=False).astype(np.uint32)
    nk = min(4, len(queries))
    out["k1_nodes"] = nodes
    ex = np.zeros((nk, len(nodes)), dtype=np.float32)
    nrm = np.zeros((nk, len(nodes)), dtype=np.float32)
    for qi in range(nk):
        for j, nid in enumerate(nodes):
            v, _ = reader.get_node(int(nid))
            diff = v - queries[qi]
            ex[qi, j] = np.sum(diff * diff)                 # A1, search_engine.py:378-379
            nrm[qi, j] = np.linalg.norm(v - queries[qi])    # M2/M4 distance, vamana_graph.py:726
    out["k1_exact"] = ex
    out["k1_norm"] = nrm
    if m:
        lut = np.stack([pq.compute_distance_table(queries[qi]) for qi in range(nk)])
        assert lut.dtype == np.float32
        out["k1_lut"] = lut
        adc_sq = np.stack([pq.asymmetric_distance_sq(codes[nodes], lut[qi]) for qi in range(nk)])
        adc = np.stack([pq.asymmetric_distance(codes[nodes], lut[qi]) for qi in range(nk)])
        # the M1 hot loop evaluates ADC one code at a time (search_engine.py:369)
        one = np.array([[pq.asymmetric_distance(codes[nid].reshape(1, -1), lut[qi])[0] for nid in nodes]
                        for qi in range(nk)], dtype=np.float32)
        assert np.array_equal(one, adc)
        out["k1_adc_sq"] = adc_sq.astype(np.float32)
        out["k1_adc"] = adc.astype(np.float32)

    # ---- search cases
    case_meta = []
    orig_random = np.random.random
    for ci, c in enumerate(cases):
        mode = c["mode"]
        k = c["k"]
        qs = queries[:c.get("nq", len(queries))]
        if c.get("f64"):
            qs = qs.astype(np.float64)
        res, stats = [], []
        if mode == "M1":
            pol = c.get("policy", 0)
            np.random.random = (lambda: 0.0) if pol == 0 else (lambda: 1.0)
            try:
                for q in qs:
                    r, st = engine._pq_accelerated_graph_search(q, k=k, L=c["L"], beam_width=c.get("bw") or None)
                    res.append(r)
                    stats.append([st["search_steps"], st["nodes_visited"], st["exact_distance_computations"],
                                  st["pq_distance_computations"]])
            finally:
                np.random.random = orig_random
        elif mode == "M2":
            for q in qs:
                if c.get("via_engine"):
                    r, st = engine._exact_graph_search(q, k=k, L=c.get("L", 100))
                else:
                    r = beam_search_from_disk(reader, q, start_id=medoid, beam_width=c["bw"], k=k)
                res.append(r)
                stats.append([0, 0, 0, 0])
        elif mode == "M3":
            for q in qs:
                r = beam_search_with_pq(graph, q, start_idx=None, beam_width=c["bw"], k=k, use_pq=c["use_pq"])
                res.append(r)
                stats.append([0, 0, 0, 0])
        elif mode == "M4":
            graph.use_pq_for_search = False
            for q in qs:
                if c.get("cython"):
                    ids = greedy_search_cython(graph, medoid, q, c["L"], compute_query_distance)
                else:
                    ids = greedy_search(graph, medoid, q, c["L"])
                res.append([(np.nan, i) for i in ids[:k]])
                stats.append([0, 0, 0, 0])
        else:
            raise ValueError(mode)
        ids, dist, dist64, cnt = pack_results(res, k)
        out[f"c{ci}_ids"] = ids
        out[f"c{ci}_dist"] = dist
        if c.get("f64"):
            out[f"c{ci}_dist64"] = dist64
        out[f"c{ci}_count"] = cnt
        out[f"c{ci}_stats"] = np.array(stats, dtype=np.uint32)
        case_meta.append(c)
    out["cases"] = np.array(json.dumps(case_meta))
    out["provenance"] = np.array(json.dumps(dict(
        generator="tests/golden/gen_golden.py", seed=seed, R=R, L_build=Lb, alpha=alpha, m=m,
        numpy=np.__version__, note="expected values produced by the reference at /root/reference")))
    np.savez_compressed(HERE / f"idx_{name}.npz", **out)
    reader.close()
    print(f"[golden] {name}: N={n} D={d} R={R} m={m} medoid={medoid} mean_deg={deg.mean():.1f} "
          f"cases={len(cases)} ({time.time() - t0:.1f}s)")


def gen_scalar_kernels():
    """C8 known answers: the reference's own numeric check (scripts/test_pydiskann_cython.sh:36-56)."""
    from pydiskann.cython_utils import l2_distance_fast_cython, cosine_similarity_cython
    rs = np.random.RandomState(0)
    out = {}
    for d in (7, 64, 96, 128, 130, 960, 1536):
        a = rs.randn(40, d).astype(np.float32)
        b = rs.randn(40, d).astype(np.float32)
        out[f"a{d}"] = a
        out[f"b{d}"] = b
        out[f"l2_{d}"] = np.array([l2_distance_fast_cython(a[i], b[i]) for i in range(40)], dtype=np.float32)
        out[f"cos_{d}"] = np.array([cosine_similarity_cython(a[i], b[i]) for i in range(40)], dtype=np.float32)
        diff = a - b
        out[f"npsum_{d}"] = np.array([np.sum(diff[i] * diff[i]) for i in range(40)], dtype=np.float32)
        out[f"norm_{d}"] = np.array([np.linalg.norm(a[i] - b[i]) for i in range(40)], dtype=np.float32)
    np.savez_compressed(HERE / "k_scalar.npz", **out)
    print("[golden] k_scalar")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    work = setup_reference()

    def want(name):
        return args.only is None or args.only == name

    if want("k_scalar"):
        gen_scalar_kernels()

    m1 = lambda L, bw, k=10, pol=0, **kw: dict(mode="M1", L=L, bw=bw, k=k, policy=pol, **kw)

    if want("randn128"):
        x, q = ds_randn(1234, 2000, 128, 24)
        np.savez_compressed(HERE / "data_randn128.npz", vectors=x, queries=q)
        build_index_fixture(work, "randn128_R16_m32", x, q, R=16, Lb=32, alpha=1.2, m=32, seed=11, cases=[
            m1(100, 0), m1(100, 8), m1(20, 8), m1(20, 0), m1(200, 0, k=20), m1(7, 3, k=5), m1(100, 0, f64=True, nq=8),
            dict(mode="M2", bw=8, k=10, via_engine=True), dict(mode="M2", bw=24, k=10), dict(mode="M2", bw=64, k=10),
            dict(mode="M3", bw=8, k=5, use_pq=True), dict(mode="M3", bw=5, k=3, use_pq=True),
            dict(mode="M3", bw=8, k=5, use_pq=False),
            dict(mode="M4", L=50, k=10), dict(mode="M4", L=100, k=10), dict(mode="M4", L=100, k=10, cython=True),
        ])
        build_index_fixture(work, "randn128_R64_m16", x, q, R=64, Lb=100, alpha=1.2, m=16, seed=12, cases=[
            m1(100, 0), m1(100, 8), m1(20, 8),
            dict(mode="M2", bw=8, k=10, via_engine=True), dict(mode="M2", bw=32, k=10),
            dict(mode="M3", bw=8, k=5, use_pq=True),
            dict(mode="M4", L=100, k=10),
        ])

    if want("sift128"):
        x, q = ds_siftlike(4321, 2000, 128, 24)
        np.savez_compressed(HERE / "data_sift128.npz", vectors=x, queries=q)
        build_index_fixture(work, "sift128_R64_m32", x, q, R=64, Lb=100, alpha=1.2, m=32, seed=21, cases=[
            m1(100, 0), m1(100, 8), m1(20, 8), m1(20, 0), m1(50, 16, k=20),
            dict(mode="M2", bw=8, k=10, via_engine=True), dict(mode="M2", bw=48, k=10),
            dict(mode="M3", bw=8, k=5, use_pq=True),
            dict(mode="M4", L=100, k=10),
        ])
        build_index_fixture(work, "sift128_R16_m32", x, q, R=16, Lb=32, alpha=1.2, m=32, seed=22, cases=[
            m1(100, 0), m1(100, 8), m1(20, 8),
        ])

    if want("unit1536"):
        x, q = ds_unit(777, 600, 1536, 12, ncl=3, noise=1.0)
        np.savez_compressed(HERE / "data_unit1536.npz", vectors=x, queries=q)
        build_index_fixture(work, "unit1536_R16_m32", x, q, R=16, Lb=32, alpha=1.2, m=32, seed=31, cases=[
            m1(100, 0, pol=0), m1(100, 0, pol=1), m1(20, 8, pol=0), m1(20, 8, pol=1), m1(40, 0, pol=1),
            m1(20, 8, pol=0, f64=True, nq=6),
            dict(mode="M2", bw=8, k=10, via_engine=True),
            dict(mode="M3", bw=8, k=5, use_pq=True),
        ])
        build_index_fixture(work, "unit1536_R16_m64", x, q, R=16, Lb=32, alpha=1.2, m=64, seed=32, cases=[
            m1(40, 0, pol=0), m1(40, 0, pol=1), m1(20, 8, pol=1),
        ])

    if want("faq32"):
        # shape of BASELINE config 1: 32 x 1536, no PQ, R=16, served by M2 with beam_width=8
        x, q = ds_unit(99, 32, 1536, 8, ncl=4)
        np.savez_compressed(HERE / "data_faq32.npz", vectors=x, queries=q)
        build_index_fixture(work, "faq32_R16_nopq", x, q, R=16, Lb=32, alpha=1.2, m=0, seed=41, cases=[
            dict(mode="M2", bw=8, k=15, via_engine=True), dict(mode="M2", bw=8, k=3, via_engine=True),
            dict(mode="M4", L=20, k=10),
        ])

    if want("deep96"):
        # D=96 cannot pass through the facade (SUPPORTED_DIMENSIONS, Q14): pydiskann-level API only
        x, q = ds_unit(555, 1500, 96, 16, ncl=24)
        np.savez_compressed(HERE / "data_deep96.npz", vectors=x, queries=q)
        build_index_fixture(work, "deep96_R32_m16", x, q, R=32, Lb=64, alpha=1.2, m=16, seed=51, via_engine=False,
                            cases=[
            dict(mode="M2", bw=24, k=10), dict(mode="M2", bw=64, k=10),
            dict(mode="M3", bw=8, k=5, use_pq=True), dict(mode="M3", bw=16, k=10, use_pq=True),
            dict(mode="M4", L=100, k=10),
        ])

    # ---- round 4: the edges of the reference's legal PQ shapes (pydiskann/pq/adaptive_pq.py:29,80-91: m in {4 ... 128},
    # sub_dim 2 ... 64; "high_accuracy" indexes of <= 50k points carry m = 96 / 128). New names only: the fixtures above are
    # never regenerated by these (their graphs are time-seeded, Q15).
    if want("unit768"):
        x, q = ds_unit(778, 500, 768, 12, ncl=3, noise=1.0)
        np.savez_compressed(HERE / "data_unit768.npz", vectors=x, queries=q)
        build_index_fixture(work, "unit768_R16_m96", x, q, R=16, Lb=32, alpha=1.2, m=96, seed=61, cases=[
            m1(100, 0, pol=0), m1(100, 0, pol=1), m1(20, 8, pol=0), m1(20, 8, pol=1),
            dict(mode="M3", bw=8, k=5, use_pq=True),
        ])
    if want("unit256"):
        x, q = ds_unit(779, 600, 256, 12, ncl=4, noise=1.0)
        np.savez_compressed(HERE / "data_unit256.npz", vectors=x, queries=q)
        build_index_fixture(work, "unit256_R16_m128", x, q, R=16, Lb=32, alpha=1.2, m=128, seed=62, cases=[
            m1(100, 0, pol=0), m1(100, 0, pol=1), m1(20, 8, pol=1),
            dict(mode="M3", bw=8, k=5, use_pq=True),
        ])
        build_index_fixture(work, "unit256_R16_m4", x, q, R=16, Lb=32, alpha=1.2, m=4, seed=63, cases=[
            m1(100, 0, pol=0), m1(100, 0, pol=1), m1(20, 8, pol=0),
            dict(mode="M3", bw=8, k=5, use_pq=True),
        ])


if __name__ == "__main__":
    main()
