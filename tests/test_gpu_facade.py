"""The drop-in facade (diskrag_amd.search_engine.SearchEngineCorrect) over a collection directory laid out and
written exactly like the reference's (index.dat records, pq_codes.bin, meta.json; T1/T2/T4). Needs a GPU."""
import json

import numpy as np
import pytest

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def write_collection(base, name, g, with_pq=True):
    cdir = base / name
    (cdir / "index").mkdir(parents=True)
    n, d = g.vectors.shape
    rec = np.empty((n, d + g.R), dtype=np.uint32)          # DiskANNPersist.save_index layout (diskann_persist.py:17-24)
    rec[:, :d] = g.vectors.view(np.uint32)
    rec[:, d:] = g.adj
    rec.tofile(cdir / "index" / "index.dat")
    meta = {"D": d, "R": g.R, "N": n, "medoid_idx": g.medoid, "use_pq": bool(with_pq and g.m), "n_subvectors": g.m}
    (cdir / "index" / "meta.json").write_text(json.dumps(meta))
    if with_pq and g.m:
        g.codes.tofile(cdir / "index" / "pq_codes.bin")
        g.codebook.astype(np.float32).tofile(cdir / "index" / "pq_codebook.f32")
    (cdir / "collection_info.json").write_text(json.dumps({"name": name, "dimension": d, "num_vectors": n}))
    return cdir


def test_facade_m1_matches_reference_golden(tmp_path):
    from diskrag_amd.search_engine import SearchEngineCorrect
    g = load_golden("sift128_R64_m32")
    write_collection(tmp_path, "c", g)
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    assert eng.use_pq and eng.n_subvectors == 32
    c = g.case(1)   # L=100, beam_width=8 (the API default), k=10
    for qi in range(6):
        res, stats = eng._pq_accelerated_graph_search(g.queries[qi], k=10, L=100, beam_width=8)
        assert [int(i) for _, i in res] == [int(i) for i in c["ids"][qi][:c["count"][qi]]]
        assert np.array_equal(np.array([d for d, _ in res], dtype=np.float32).view(np.uint32),
                              c["dist"][qi][:c["count"][qi]].view(np.uint32))
        assert [stats["search_steps"], stats["nodes_visited"], stats["exact_distance_computations"],
                stats["pq_distance_computations"]] == c["stats"][qi].tolist()
        assert isinstance(res[0][0], np.float32) and isinstance(res[0][1], np.uint32)
    # batched entry point == per-query calls
    ids, dist, cnt, st = eng.search_batch(g.queries, k=10, L=100, beam_width=8)
    assert np.array_equal(ids, c["ids"])
    eng.close()


def test_facade_search_and_faq_dedup(tmp_path):
    from diskrag_amd.search_engine import SearchEngineCorrect
    g = load_golden("randn128_R16_m32")
    write_collection(tmp_path, "c", g)

    def lookup(idx):   # every 3 consecutive vectors share a qa_id; odd ids are not FAQ rows
        return f"text {idx}", {"type": "faq" if idx % 2 == 0 else "chunk", "qa_id": f"qa{idx // 6}"}

    eng = SearchEngineCorrect("c", base_dir=tmp_path, text_lookup=lookup)
    q = g.queries[0]
    out = eng.search("hello", k=5, embedding_fn=lambda s: q)
    assert [r["text"] for r in out["results"]] == [f"text {int(i)}" for i in g.case(2)["ids"][0][:5]]   # L=max(2k,20)=20, bw=8
    assert out["stats"]["search_type"] == "pq_accelerated" and out["stats"]["L_search"] == 20
    faq = eng.faq_search("hello", k=3, embedding_fn=lambda s: q)
    qa = [r["metadata"]["qa_id"] for r in faq["results"]]
    assert len(qa) == len(set(qa)) and all(r["metadata"]["type"] == "faq" for r in faq["results"])
    assert faq["stats"]["total_results_before_dedup"] <= 9
    with pytest.raises(ValueError):
        eng.search("x", embedding_fn=None)
    with pytest.raises(ValueError):
        eng.search("x", embedding_fn=lambda s: np.zeros(7, dtype=np.float32))
    eng.close()


def test_facade_exact_mode_when_pq_missing(tmp_path):
    """No PQ files -> the engine serves M2 with beam_width 8 (search_engine.py:49-51, 508-528; Q6: <= 8 hits)."""
    from diskrag_amd.search_engine import SearchEngineCorrect
    from oracle import pyoracle as orc
    g = load_golden("randn128_R16_m32")
    write_collection(tmp_path, "c", g, with_pq=False)
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    assert not eng.use_pq
    res, st = eng._exact_graph_search(g.queries[0], k=15, L=100)
    oids, odist, ocnt, _ = orc.search_batch(g.vectors, g.adj, g.queries[:1], g.medoid, orc.M2, 15, bw=8)
    assert len(res) == int(ocnt[0]) <= 8
    assert [int(i) for _, i in res] == oids[0, :ocnt[0]].tolist()
    assert st["search_type"] == "exact_beam_search"
    eng.close()


def test_device_built_index_survives_the_reference_file_formats(tmp_path):
    """N1+N2+N3 end to end: build graph and PQ on the device, write the reference's files (persist.write_index), open
    them through the facade (dr_index_open on index.dat): same results as the index that never left HBM."""
    import json as _json
    from diskrag_amd import HipIndex, persist
    from diskrag_amd.search_engine import SearchEngineCorrect
    from diskrag_amd.synth import sift_like
    x, q = sift_like(40000, 128, n_queries=64, n_clusters=64, seed=11, query_seed=12)   # > 32768: the locality bit order is on
    ix = HipIndex.create_empty(x, R=32)
    medoid, _ = ix.build_vamana(L_build=60, alpha=1.2, passes=2, seed=3)
    cb = ix.pq_train(32, n_sample=10000, iters=4)
    codes = ix.pq_encode(cb, want_codes=True)
    adj = ix.get_adjacency()
    want = ix.search_batch(q, 10, L=50, beam_width=8)
    cdir = tmp_path / "built"
    meta = persist.write_index(cdir / "index", x, adj, medoid, codes=codes, codebook=cb,
                               build_params={"L": 60, "alpha": 1.2}, pq_pickle=False)
    assert meta["N"] == 40000 and meta["n_subvectors"] == 32 and meta["use_pq"]
    (cdir / "collection_info.json").write_text(_json.dumps({"name": "built", "dimension": 128, "num_vectors": 40000}))
    eng = SearchEngineCorrect("built", base_dir=tmp_path)
    assert eng.use_pq and eng.medoid_idx == medoid
    got = eng.search_batch(q, k=10, L=50, beam_width=8)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    back = persist.read_index(cdir / "index")
    assert np.array_equal(back.adjacency, adj) and np.array_equal(back.codes, codes)
    eng.close(); ix.close()


def test_facade_float64_query_takes_the_float64_path(tmp_path):
    """The CLI's query is float64 (diskrag.py:194): the facade must return the reference's float64 results for it."""
    from diskrag_amd.search_engine import SearchEngineCorrect
    g = load_golden("unit1536_R16_m32")
    write_collection(tmp_path, "c", g)
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    ci = [i for i, c in enumerate(g.cases) if c.get("f64")][0]
    c = g.case(ci)
    for qi in range(len(c["queries"])):
        res, stats = eng._pq_accelerated_graph_search(c["queries"][qi], k=c["k"], L=c["L"], beam_width=c["bw"] or None)
        n = int(c["count"][qi])
        assert [int(i) for _, i in res] == [int(i) for i in c["ids"][qi][:n]]
        assert all(isinstance(d, np.float64) for d, _ in res)
        assert np.array_equal(np.array([d for d, _ in res]).view(np.uint64), c["dist64"][qi][:n].view(np.uint64))
        assert [stats["search_steps"], stats["nodes_visited"], stats["exact_distance_computations"],
                stats["pq_distance_computations"]] == c["stats"][qi].tolist()
    # the same vector as float32 takes the float32 kernel and returns np.float32 distances
    res32, _ = eng._pq_accelerated_graph_search(c["queries"][0].astype(np.float32), k=c["k"], L=c["L"], beam_width=c["bw"] or None)
    assert all(isinstance(d, np.float32) for d, _ in res32)
    # search(): the CLI route, float64 from the embedding function
    out = eng.search("q", k=3, embedding_fn=lambda s: np.array(c["queries"][0].tolist()), L_search=20)
    assert out["stats"]["search_type"] == "pq_accelerated"
    eng.close()


def test_concurrent_searches_on_one_handle():
    """The reference engine is a shared singleton hit by several threads (search_engine.py:879, Q18): concurrent
    dr_search_batch / dr_search_batch_f64 calls on one handle must each get their own answer."""
    import threading
    from diskrag_amd import _ffi
    from tests.test_gpu_parity import get_index
    g = load_golden("sift128_R64_m32")
    ix = get_index("sift128_R64_m32")
    cases = [(1, 100, 8), (1, 100, 0), (1, 20, 8), (2, 100, 8)]      # (mode, L, bw)
    want = {c: ix.search_batch(g.queries, 10, L=c[1], beam_width=c[2], mode=c[0]) for c in cases}
    q64 = g.queries[:8].astype(np.float64)
    want64 = ix.search_batch_f64(q64, 10, L=50, beam_width=8)
    errors = []

    def worker(tid):
        try:
            for it in range(12):
                c = cases[(tid + it) % len(cases)]
                ids, dist, cnt, st = ix.search_batch(g.queries, 10, L=c[1], beam_width=c[2], mode=c[0])
                if not (np.array_equal(ids, want[c][0]) and np.array_equal(dist.view(np.uint32), want[c][1].view(np.uint32))):
                    errors.append((tid, it, c))
                if it % 4 == tid % 4:
                    i64, d64, _, _ = ix.search_batch_f64(q64, 10, L=50, beam_width=8)
                    if not (np.array_equal(i64, want64[0]) and np.array_equal(d64.view(np.uint64), want64[1].view(np.uint64))):
                        errors.append((tid, it, "f64"))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors[:3]


def test_b5_seams_diagnostics_and_search_with_debug(tmp_path):
    """The per-kernel seams as METHODS (_compute_exact_distance, _build_pq_lut_fixed, _get_pq_distance)
    against the reference's golden K1 bits, the metadata join from metadata.parquet (one read, nested metadata merged as
    CollectionManager.get_text_by_index does), and search_with_debug (search_engine.py:616-660)."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from diskrag_amd.search_engine import SearchEngineCorrect
    g = load_golden("sift128_R64_m32")
    cdir = write_collection(tmp_path, "c", g)
    n = len(g.vectors)
    rows = {"vector_index": list(range(n)), "text": [f"t{i}" for i in range(n)],
            "metadata": [json.dumps({"type": "faq", "qa_id": f"qa{i // 4}", "metadata": json.dumps({"source": "s%d" % i, "type": "x"})})
                         if i % 5 else "not json" for i in range(n)]}
    pq.write_table(pa.table(rows), cdir / "metadata.parquet")
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    try:
        nodes = g.z["k1_nodes"]
        for qi in range(2):
            q = g.queries[qi]
            lut = eng._build_pq_lut_fixed(q)
            assert np.array_equal(lut.view(np.uint32), g.z["k1_lut"][qi].view(np.uint32))
            for t, node in enumerate(nodes[:6]):
                e = eng._compute_exact_distance(q, int(node))
                assert np.float32(e).view(np.uint32) == g.z["k1_exact"][qi, t].view(np.uint32)
                p = eng._get_pq_distance(lut, g.codes[int(node)])
                assert np.float32(p).view(np.uint32) == g.z["k1_adc"][qi, t].view(np.uint32)
        text, md = eng._get_text_by_index(7)
        assert text == "t7" and md["qa_id"] == "qa1" and md["type"] == "faq" and md["source"] == "s7"   # nested merged, top level kept
        text, md = eng._get_text_by_index(10)
        assert md == {"text": "t10", "id": 10}                                                          # unparsable metadata
        assert eng._get_text_by_index(n + 5) is None
        assert eng._run_diagnostic_check() is True
        emb = lambda s: g.queries[3]          # noqa: E731
        dbg = eng.search_with_debug("x", k=5, embedding_fn=emb, debug_mode=True)
        assert dbg["diagnostic_passed"] is True and len(dbg["exact_results"]) == 5 and len(dbg["pq_results"]) == 5
        assert dbg["debug_info"]["medoid_idx"] == g.medoid and len(dbg["debug_info"]["neighbor_info"]) == 5
        plain = eng.search_with_debug("x", k=5, embedding_fn=emb)
        assert [r["text"] for r in plain["results"]] == [r["text"] for r in eng.search("x", k=5, embedding_fn=emb)["results"]]
        res = eng.faq_search("x", k=3, embedding_fn=emb)
        assert all(r["metadata"]["type"] == "faq" for r in res["results"])
    finally:
        eng.close()


def test_pq_load_surfaces_io_errors_and_never_writes(tmp_path):
    """loading must not write into the index directory, and only an unreadable model downgrades to exact mode."""
    import os
    from diskrag_amd.search_engine import SearchEngineCorrect
    g = load_golden("randn128_R16_m32")
    cdir = write_collection(tmp_path, "c", g)
    before = sorted(os.listdir(cdir / "index"))
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    assert eng.use_pq and sorted(os.listdir(cdir / "index")) == before
    eng.close()
    # a truncated code file is a model-format error: exact mode, as the reference does for a model it cannot load
    (cdir / "index" / "pq_codes.bin").write_bytes(b"\x00" * 10)
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    assert not eng.use_pq
    eng.close()
    meta = json.loads((cdir / "index" / "meta.json").read_text())
    meta["pq_centroids"] = 128
    (cdir / "index" / "meta.json").write_text(json.dumps(meta))
    g.codes.tofile(cdir / "index" / "pq_codes.bin")
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    assert not eng.use_pq                    # 128 centroids: refused explicitly, not reshaped wrongly
    eng.close()


def test_request_batcher_returns_the_bits_of_direct_calls(tmp_path):
    """Concurrent one-query requests coalesced into search_batch calls (diskrag_amd/batching.py; the serving seam of
    app.py:84-130): every request gets exactly what its own _pq_accelerated_graph_search call returns -- the reference's
    golden output."""
    import threading
    from diskrag_amd.batching import RequestBatcher
    from diskrag_amd.search_engine import SearchEngineCorrect
    g = load_golden("sift128_R64_m32")
    write_collection(tmp_path, "c", g)
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    c = g.case(1)   # L=100, beam_width=8, k=10
    nq = len(g.queries)
    got = {}
    with RequestBatcher(eng, k_max=10, L=100, beam_width=8, max_batch=32, max_wait_ms=2) as rb:
        def client(lo, hi):
            for qi in range(lo, hi):
                got[qi] = rb.search(g.queries[qi], k=10 if qi % 2 else 7)
        th = [threading.Thread(target=client, args=(i * nq // 8, (i + 1) * nq // 8)) for i in range(8)]
        for t in th: t.start()
        for t in th: t.join()
        assert rb.queries_sent == nq and rb.batches_sent < nq
    for qi in range(nq):
        res, stats = got[qi]
        k = 10 if qi % 2 else 7
        n = min(int(c["count"][qi]), k)
        assert [int(i) for _, i in res] == [int(i) for i in c["ids"][qi][:n]]
        assert np.array_equal(np.array([d for d, _ in res], dtype=np.float32).view(np.uint32), c["dist"][qi][:n].view(np.uint32))
        assert [stats["search_steps"], stats["nodes_visited"], stats["exact_distance_computations"],
                stats["pq_distance_computations"]] == c["stats"][qi].tolist()
    eng.close()


def test_facade_request_threads_share_launches_and_keep_the_goldens(tmp_path):
    """Round 4: the facade's one-query searches go through the pipelined path (submit + wait), so the handler threads of a
    server (one query per request, app.py:84-130) share launches instead of queueing on the handle -- every request still gets
    the reference's golden answer (ids, distance bits, the four counters), whichever requests it rode with; mixed with blocking
    batched calls and M2 requests on the same engine."""
    import threading
    from diskrag_amd.search_engine import SearchEngineCorrect
    g = load_golden("sift128_R64_m32")
    write_collection(tmp_path, "c", g)
    eng = SearchEngineCorrect("c", base_dir=tmp_path)
    c = g.case(1)            # M1, L=100, beam_width=8, k=10
    m2 = eng._exact_graph_search(g.queries[3], k=8)
    errors = []

    def handler(t):
        try:
            for i in range(30):
                qi = (5 * t + i) % len(g.queries)
                res, stats = eng._pq_accelerated_graph_search(g.queries[qi], k=10, L=100, beam_width=8)
                n = int(c["count"][qi])
                ok = ([int(x) for _, x in res] == [int(x) for x in c["ids"][qi][:n]] and
                      np.array_equal(np.array([d for d, _ in res], dtype=np.float32).view(np.uint32), c["dist"][qi][:n].view(np.uint32)) and
                      [stats["search_steps"], stats["nodes_visited"], stats["exact_distance_computations"], stats["pq_distance_computations"]] == c["stats"][qi].tolist())
                if not ok:
                    errors.append((t, i, qi))
                if i % 10 == t % 10:
                    ids, _, _, _ = eng.search_batch(g.queries, k=10, L=100, beam_width=8)
                    if not np.array_equal(ids, c["ids"]):
                        errors.append((t, i, "batch"))
                    r2, _ = eng._exact_graph_search(g.queries[3], k=8)
                    if [int(x) for _, x in r2] != [int(x) for _, x in m2[0]]:
                        errors.append((t, i, "m2"))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=handler, args=(t,)) for t in range(12)]
    for t in th: t.start()
    for t in th: t.join(timeout=300)
    assert not errors, errors[:3]
    s = eng.index.pipeline_stats()
    assert s["tickets"] >= 12 * 30 and s["launches"] <= s["tickets"]
    eng.close()
