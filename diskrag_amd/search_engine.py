"""Drop-in search facade over the MI355X engine, mirroring the reference's `search_engine.py`.

Seams preserved (file:line into Jolara-ai/diskrag):
  B3  SearchEngineCorrect(collection_name, use_thread_safe_stats=True)            search_engine.py:18
  B1  _pq_accelerated_graph_search(q, k=10, L=100, beam_width=None)               search_engine.py:398-506
  B2  _exact_graph_search(q, k=10, L=100)                                         search_engine.py:508-528
  B4  search(...) / faq_search(...)                                               search_engine.py:530-614, 694-812
The graph traversal, PQ table build, ADC and exact distances run in libdiskrag_hip.so (HIP, gfx950); this
module is host glue: file discovery, argument checks with the reference's error behaviour, result shaping.
`search_batch` is the batched entry point the reference lacks.

Text/metadata join (CollectionManager.get_text_by_index, preprocessing/collection.py:445-510) is string work
outside the accelerated path: it is served from `metadata.parquet` read once with pyarrow, or from a
`text_lookup` callable.
"""
import json
import logging
import pickle
import threading
import time
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from . import _ffi

logger = logging.getLogger(__name__)

# preprocessing/config.py:88 (Q14: the facade rejects other dimensions; D=96 only through HipIndex directly)
SUPPORTED_DIMENSIONS = {128, 256, 768, 960, 1536}
RAW_CODEBOOK_NAME = "pq_codebook.f32"   # [m][256][D/m] float32 little-endian, written by export_codebook()


def read_codebook(pq_model_pkl):
    """The codebook [m][256][D/m] held by the reference's pq_model.pkl (a pickle of sklearn KMeans objects, T3,
    diskann_persist.py:33-105). Needs sklearn to unpickle; nothing is written."""
    with open(Path(pq_model_pkl), "rb") as f:
        data = pickle.load(f)
    kms = data["kmeans_list"] if isinstance(data, dict) else data.kmeans_list
    return np.stack([np.asarray(km.cluster_centers_, dtype=np.float32) for km in kms])


def export_codebook(pq_model_pkl, out_path=None):
    """Converts pq_model.pkl into the raw float file a box without sklearn can read (explicit, offline step)."""
    pq_model_pkl = Path(pq_model_pkl)
    cb = read_codebook(pq_model_pkl)
    out_path = Path(out_path) if out_path else pq_model_pkl.with_name(RAW_CODEBOOK_NAME)
    cb.tofile(out_path)
    return cb


class SearchEngineCorrect:
    def __init__(self, collection_name: str, use_thread_safe_stats: bool = True, base_dir: Optional[Path] = None,
                 device: int = 0, text_lookup: Optional[Callable[[int], Optional[Tuple[str, dict]]]] = None,
                 vector_tier: str = "hbm"):
        # vector_tier "host": the full-precision rows of index.dat stay in pinned host memory (the counterpart of the
        # reference's MMapNodeReader tier, diskann_persist.py:201-234); graph and PQ codes are resident in HBM either way
        if vector_tier not in ("hbm", "host"):
            raise ValueError("vector_tier must be 'hbm' or 'host'")
        self.collection_name = collection_name
        base = Path(base_dir) if base_dir else Path("collections")
        cdir = base / collection_name
        info_path = cdir / "collection_info.json"
        if not info_path.exists():
            raise ValueError(f"找不到集合: {collection_name}")                       # search_engine.py:22-23
        self.info = json.loads(info_path.read_text(encoding="utf-8"))
        self.dimension = int(self.info["dimension"])
        index_dir = cdir / "index"
        index_path, meta_path = index_dir / "index.dat", index_dir / "meta.json"
        if not index_path.exists() or not meta_path.exists():
            raise ValueError(f"集合 {collection_name} 的索引檔案不完整")              # search_engine.py:29-30
        self.meta = json.loads(meta_path.read_text())
        if self.dimension not in SUPPORTED_DIMENSIONS:
            raise ValueError(f"不支援的向量維度: {self.dimension}。請使用支援的維度重新建立索引")   # :81-85
        self.medoid_idx = int(self.meta.get("medoid_idx", 0))
        self.R = int(self.meta.get("R", 32))
        self.index = _ffi.HipIndex.open(index_path, int(self.meta["N"]), self.dimension, self.R, self.medoid_idx,
                                        device=device, vector_tier=_ffi.TIER_HOST if vector_tier == "host" else _ffi.TIER_HBM)
        self.use_pq = bool(self.meta.get("use_pq", True))
        self.n_subvectors = 0
        self.sub_dim = 0
        self.num_centroids = 0
        self._pq_codes_host = None
        if self.use_pq:
            self._load_pq(index_dir)
        self._text_lookup = text_lookup
        self._meta_rows = None            # vector_index -> (text, metadata), built on first use
        self._meta_path = cdir / "metadata.parquet"
        self.search_stats = {"total_searches": 0, "total_exact_computations": 0, "total_pq_computations": 0,
                             "total_search_time": 0.0}
        self._stats_lock = threading.Lock() if use_thread_safe_stats else None

    # PQ load failure downgrades the engine to exact mode, as the reference does (search_engine.py:49-51, 70-72)
    def _load_pq(self, index_dir: Path):
        codes_path = index_dir / "pq_codes.bin"
        raw_path, pkl_path = index_dir / RAW_CODEBOOK_NAME, index_dir / "pq_model.pkl"
        if not codes_path.exists() or not (raw_path.exists() or pkl_path.exists()):
            logger.warning("PQ 文件不完整，切換到暴力搜索模式")
            self.use_pq = False
            return
        # Only a model that cannot be READ downgrades the engine (the reference's behaviour for its pickle); file-system
        # and device errors surface. Nothing is written while loading: the raw codebook file is an offline export.
        try:
            m = int(self.meta["n_subvectors"])
            ncent = int(self.meta.get("pq_centroids", self.meta.get("n_centroids", 256)))
            if ncent != 256:
                raise ValueError(f"PQ codebook with {ncent} centroids (the engine serves 256: uint8 codes)")
            codes = np.fromfile(codes_path, dtype=np.uint8)
            if codes.size != int(self.meta["N"]) * m:
                raise ValueError(f"pq_codes.bin holds {codes.size} bytes, expected N*m = {int(self.meta['N']) * m}")
            codes = codes.reshape(int(self.meta["N"]), m)                                       # T2
            if raw_path.exists():
                cb = np.fromfile(raw_path, dtype=np.float32)
                if cb.size != m * 256 * (self.dimension // m):
                    raise ValueError(f"{raw_path.name} holds {cb.size} floats, expected {m * 256 * (self.dimension // m)}")
                cb = cb.reshape(m, 256, self.dimension // m)
            else:
                cb = read_codebook(pkl_path)                 # needs sklearn; raises on a box without it
                if cb.shape != (m, 256, self.dimension // m):
                    raise ValueError(f"pq_model.pkl codebook shape {cb.shape}")
        except (ValueError, KeyError, ImportError, ModuleNotFoundError, pickle.UnpicklingError, AttributeError, EOFError) as e:
            logger.warning("PQ 模型載入失敗: %s，切換到暴力搜索模式", e)
            self.use_pq = False
            return
        self.index.set_pq(cb, codes)
        self._pq_codes_host = codes
        self.n_subvectors, self.sub_dim, self.num_centroids = m, self.dimension // m, 256

    def close(self):
        self.index.close()

    # ------------------------------------------------------------------ stats bookkeeping (search_engine.py:118-140)
    def _bump(self, exact, pq, secs, n=1):
        def upd():
            self.search_stats["total_searches"] += n
            self.search_stats["total_exact_computations"] += int(exact)
            self.search_stats["total_pq_computations"] += int(pq)
            self.search_stats["total_search_time"] += secs
        if self._stats_lock:
            with self._stats_lock:
                upd()
        else:
            upd()

    def get_search_statistics(self) -> Dict[str, Any]:
        s = dict(self.search_stats)
        if s["total_searches"] == 0:
            return {"message": "尚未執行任何搜索"}
        ae, ap = s["total_exact_computations"] / s["total_searches"], s["total_pq_computations"] / s["total_searches"]
        return {"total_searches": s["total_searches"], "avg_exact_computations_per_search": ae,
                "avg_pq_computations_per_search": ap, "avg_search_time": s["total_search_time"] / s["total_searches"],
                "total_exact_computations": s["total_exact_computations"],
                "total_pq_computations": s["total_pq_computations"],
                "overall_computation_reduction_rate": 1 - (ae / max(1, ap))}

    # ------------------------------------------------------------------ B1 / B2
    @staticmethod
    def _check_status(stats):
        bits = int(np.bitwise_or.reduce(stats["status"])) if len(stats) else 0
        if bits & 16:       # dr_stats.status bit 4 (include/diskrag_hip.h): DR_F_IP on a query that is not unit-norm
            n = int(((stats["status"] & 16) != 0).sum())
            raise _ffi.DiskragHipError(_ffi.E_ARG, f"{n} queries are not unit-norm: the inner-product reading of the rerank (DR_F_IP) "
                                       "is defined on unit-norm queries and rows only")
        bad = stats["status"] != 0
        if bad.any():
            raise _ffi.DiskragHipError(_ffi.E_OVERFLOW, f"work-area overflow in {int(bad.sum())} queries (status bits {bits})")

    @staticmethod
    def _is_f64(query_vector) -> bool:
        """The CLI hands np.array(list) = float64 (diskrag.py:194), the API float32 (app.py:45); the reference's numpy
        arithmetic follows the query's dtype (quirk Q8), and so does the engine: float64 queries take the float64
        kernel and get np.float64 distances back."""
        return np.asarray(query_vector).dtype == np.float64

    def _one(self, query_vector, k, **kw):
        """One float32 query through the pipelined path (submit + wait): the handle is not held while the request is in flight,
        so concurrent request handlers (the reference serves one query per request from a thread pool, app.py:84-130) share
        launches -- 5x the requests/s of serialised blocking calls at 16+ threads, the same bits (tests/test_gpu_coalesce.py); a
        lone request is launched at once."""
        q = np.ascontiguousarray(query_vector, dtype=np.float32).reshape(1, -1)
        return self.index.search_submit(q, k, **kw).wait()

    def _pq_accelerated_graph_search(self, query_vector: np.ndarray, k: int = 10, L: int = 100,
                                     beam_width: Optional[int] = None, band_policy: int = 0
                                     ) -> Tuple[List[Tuple[float, int]], Dict]:
        t0 = time.time()
        f64 = self._is_f64(query_vector)
        run = self.index.search_batch_f64 if f64 else self._one
        if not f64 and (band_policy & 0xFF) == 2:
            # _ffi.POLICY_COIN(seed): the reference's coin flip itself (search_engine.py:393-395) -- a sequential walk, a blocking call
            run = lambda qv, kk, **kw: self.index.search_batch(np.ascontiguousarray(qv, dtype=np.float32).reshape(1, -1), kk, **kw)   # noqa: E731
        ids, dist, cnt, st = run(query_vector, k, L=L, beam_width=beam_width or 0, mode=_ffi.MODE_M1,
                                 band_policy=band_policy)
        self._check_status(st)
        secs = time.time() - t0
        n = int(cnt[0])
        ftype = np.float64 if f64 else np.float32
        results = [(ftype(dist[0, i]), np.uint32(ids[0, i])) for i in range(n)]
        exact, pq = int(st["exact"][0]), int(st["pq"][0])
        self._bump(exact, pq, secs)
        stats = {"search_time": secs, "nodes_visited": int(st["visited"][0]), "exact_distance_computations": exact,
                 "pq_distance_computations": pq, "computation_reduction_rate": 1 - (exact / max(1, pq)),
                 "search_steps": int(st["steps"][0])}
        return results, stats

    def _exact_graph_search(self, query_vector: np.ndarray, k: int = 10, L: int = 100
                            ) -> Tuple[List[Tuple[float, int]], Dict]:
        t0 = time.time()
        f64 = self._is_f64(query_vector)
        run = self.index.search_batch_f64 if f64 else self._one
        # the reference hard-codes beam_width=8 here and ignores L (search_engine.py:513-519, Q6)
        ids, dist, cnt, st = run(query_vector, k, L=L, beam_width=8, mode=_ffi.MODE_M2)
        self._check_status(st)
        secs = time.time() - t0
        n = int(cnt[0])
        ftype = np.float64 if f64 else np.float32
        results = [(ftype(dist[0, i]), np.uint32(ids[0, i])) for i in range(n)]
        return results, {"search_time": secs, "exact_distance_computations": len(results) * 2,
                         "search_type": "exact_beam_search"}

    def search_batch(self, query_vectors: np.ndarray, k: int = 10, L: Optional[int] = None,
                     beam_width: Optional[int] = 8, use_pq_search: bool = True, band_policy: int = 0):
        """Batched form of B1/B2: ids[nq,k], dist[nq,k], count[nq], stats (structured array)."""
        if L is None:
            L = max(k * 2, 20)
        q = np.asarray(query_vectors)
        if q.ndim != 2 or q.shape[1] != self.dimension:
            raise ValueError(f"查詢向量維度不匹配: 預期 {self.dimension}，實際 {q.shape[-1]}")
        t0 = time.time()
        if use_pq_search and self.use_pq:
            out = self.index.search_batch(q, k, L=L, beam_width=beam_width or 0, mode=_ffi.MODE_M1,
                                          band_policy=band_policy)
        else:
            out = self.index.search_batch(q, k, L=L, beam_width=8, mode=_ffi.MODE_M2)
        self._check_status(out[3])
        self._bump(out[3]["exact"].sum(), out[3]["pq"].sum(), time.time() - t0, n=q.shape[0])
        return out

    # ------------------------------------------------------------------ text join
    def _get_text_by_index(self, idx: int):
        """CollectionManager.get_text_by_index (preprocessing/collection.py:445-510): text and metadata of one vector.
        The reference re-reads metadata.parquet and filters it for every hit (Q17); here the file is read once into a
        vector_index -> row dict. Same shaping: a JSON-string metadata is parsed (fallback {"text", "id"}), a struct
        becomes a dict, and a nested "metadata" JSON string is merged in without overriding top-level keys."""
        if self._text_lookup is not None:
            return self._text_lookup(int(idx))
        if self._meta_rows is None:
            rows = {}
            if self._meta_path.exists():
                import pyarrow.parquet as pq
                for row in pq.read_table(self._meta_path).to_pylist():
                    vi = row.get("vector_index")
                    if vi is not None and int(vi) not in rows:          # the reference takes the first matching row
                        rows[int(vi)] = row
            self._meta_rows = rows
        row = self._meta_rows.get(int(idx))
        if row is None:
            return None
        text, raw = row.get("text"), row.get("metadata")
        if isinstance(raw, str):
            try:
                md = json.loads(raw)
            except json.JSONDecodeError:
                md = {"text": text, "id": int(idx)}
        elif isinstance(raw, dict):
            md = dict(raw)
        else:
            md = {"text": text, "id": int(idx)}
        if isinstance(md, dict) and isinstance(md.get("metadata"), str):
            try:
                for key, value in json.loads(md["metadata"]).items():
                    if key not in md:
                        md[key] = value
            except (json.JSONDecodeError, AttributeError):
                pass
        return text, md

    # ------------------------------------------------------------------ B5: the per-kernel seams
    def _compute_exact_distance(self, query_vector: np.ndarray, node_id: int):
        """search_engine.py:374-379: squared L2 of one stored vector, numpy's summation order (float32 queries)."""
        self._bump(1, 0, 0.0, n=0)
        return self.index.exact_distances(np.asarray(query_vector, dtype=np.float32)[None, :], [int(node_id)])[0, 0]

    def _build_pq_lut_fixed(self, query_vector: np.ndarray) -> np.ndarray:
        """search_engine.py:281-318 -> DiskANNPQ.compute_distance_table (fast_pq.py:294-318): T[m][256] float32."""
        if not self.use_pq:
            raise ValueError("PQ 模型缺少 kmeans_list，無法進行距離計算")
        return self.index.distance_table(np.asarray(query_vector, dtype=np.float32)[None, :])[0]

    _build_pq_lut = _build_pq_lut_fixed            # search_engine.py:262-279 (same table)

    def _get_pq_distance(self, lut: np.ndarray, pq_code: np.ndarray):
        """search_engine.py:365-372 -> asymmetric_distance (fast_pq.py:320-333): sqrt of the float32 sum of T[j, code_j]
        taken in sub-quantiser order. Pure host arithmetic on the caller's arrays (diagnostics only; the search kernels
        compute the same sum on the device from their own table)."""
        s = np.float32(0.0)
        for j, c in enumerate(np.asarray(pq_code).reshape(-1)):
            s = np.float32(s + np.float32(lut[j, int(c)]))
        return np.sqrt(s)

    def _run_diagnostic_check(self) -> bool:
        """search_engine.py:142-254: exact and PQ distances of a few stored vectors must be computable and correlated."""
        try:
            n = int(self.meta["N"])
            idx = np.random.choice(n, min(10, n), replace=False)
            q, _ = self.index.get_node(int(idx[0]))
            exact = [float(self._compute_exact_distance(q, int(i))) for i in idx[:5]]
            if not exact:
                return False
            if not self.use_pq:
                return True
            lut = self._build_pq_lut_fixed(q)
            if lut.shape != (self.n_subvectors, 256) or np.allclose(lut, 0):
                return False
            pqd = [float(self._get_pq_distance(lut, self._pq_codes_host[int(i)])) for i in idx[:5]]
            if len(exact) > 2 and np.std(exact) > 0 and np.std(pqd) > 0 and np.corrcoef(exact, pqd)[0, 1] < 0.5:
                logger.error("距離相關性過低")
                return False
            return True
        except Exception as e:  # noqa: BLE001 - the reference reports any failure as a failed diagnosis
            logger.error("診斷過程中發生錯誤: %s", e)
            return False

    def _debug_search_step_by_step(self, query_vector: np.ndarray, k: int = 5) -> Dict:
        """search_engine.py:319-362: the medoid's distance and exact / PQ distances of its first neighbours."""
        medoid_exact = self._compute_exact_distance(query_vector, self.medoid_idx)
        try:
            lut = self._build_pq_lut_fixed(query_vector)
        except Exception as e:  # noqa: BLE001
            return {"error": str(e)}
        _, nbrs = self.index.get_node(self.medoid_idx)
        info = []
        for nb in [int(v) for v in nbrs if int(v) < len(self._pq_codes_host)][:5]:
            e = self._compute_exact_distance(query_vector, nb)
            p = self._get_pq_distance(lut, self._pq_codes_host[nb])
            info.append({"id": nb, "exact_dist": e, "pq_dist": p, "ratio": p / e if e > 0 else float("inf")})
        return {"medoid_idx": self.medoid_idx, "medoid_exact_dist": medoid_exact, "neighbor_info": info}

    # ------------------------------------------------------------------ B4
    def _run(self, query_vector, k, L_search, beam_width, use_pq_search):
        if use_pq_search and not self.use_pq:
            use_pq_search = False
        if use_pq_search and self.use_pq:
            res, st = self._pq_accelerated_graph_search(query_vector, k, L_search, beam_width)
        else:
            res, st = self._exact_graph_search(query_vector, k, L_search)
        return res, st, ("pq_accelerated" if (use_pq_search and self.use_pq) else "exact")

    def search(self, query: str, k: int = 5, beam_width: int = 8, embedding_fn: Optional[Callable] = None,
               L_search: Optional[int] = None, use_pq_search: bool = True, use_simple_pq: bool = False
               ) -> Dict[str, Any]:
        if embedding_fn is None:
            raise ValueError("必須提供 embedding_fn 來產生查詢向量")
        if L_search is None:
            L_search = max(k * 2, 20)
        t_all = time.time()
        query_vector = np.asarray(embedding_fn(query))
        emb = time.time() - t_all
        if query_vector.shape[0] != self.dimension:
            raise ValueError(f"查詢向量維度不匹配: 預期 {self.dimension}，實際 {query_vector.shape[0]}")
        if use_simple_pq:
            use_pq_search = True
        top, st, kind = self._run(query_vector, k, L_search, beam_width, use_pq_search)
        results = []
        for dist, idx in top:
            td = self._get_text_by_index(int(idx))
            if td:
                text, md = td
                if not isinstance(md, dict):
                    md = {"id": int(idx), "text": text}
                results.append({"text": text, "distance": float(dist), "metadata": md})
        return {"results": results,
                "timing": {"embedding_time": emb, "search_time": st.get("search_time", 0),
                           "total_time": time.time() - t_all},
                "stats": {"search_type": kind, "nodes_visited": st.get("nodes_visited", 0), "k": k,
                          "L_search": L_search}}

    def search_with_debug(self, query: str, k: int = 5, beam_width: int = 8, embedding_fn: Optional[Callable] = None,
                          L_search: Optional[int] = None, use_pq_search: bool = True, debug_mode: bool = False
                          ) -> Dict[str, Any]:
        """search_engine.py:616-660: with debug_mode the diagnosis, the step-by-step view and both searches side by side;
        otherwise search()."""
        if embedding_fn is None:
            raise ValueError("必須提供 embedding_fn 來產生查詢向量")
        if L_search is None:
            L_search = max(k * 2, 20)
        if not debug_mode:
            return self.search(query, k, beam_width, embedding_fn, L_search, use_pq_search)
        query_vector = np.asarray(embedding_fn(query))
        diagnostic_result = self._run_diagnostic_check()
        debug_info = self._debug_search_step_by_step(query_vector, k)
        exact_results, _ = self._exact_graph_search(query_vector, k, L_search)
        pq_results = []
        if use_pq_search:
            try:
                pq_results, _ = self._pq_accelerated_graph_search(query_vector, k, L_search, beam_width)
            except Exception as e:  # noqa: BLE001 - as the reference: a failing PQ search is reported, not raised
                logger.error("PQ 搜索失敗: %s", e)
                pq_results = []
        return {"debug_info": debug_info, "exact_results": exact_results,
                "pq_results": pq_results if use_pq_search else [], "diagnostic_passed": diagnostic_result}

    def faq_search(self, query: str, k: int = 5, beam_width: int = 8, embedding_fn: Optional[Callable] = None,
                   L_search: Optional[int] = None, use_pq_search: bool = True) -> Dict[str, Any]:
        if embedding_fn is None:
            raise ValueError("必須提供 embedding_fn 來產生查詢向量")
        if L_search is None:
            L_search = max(k * 2, 20)
        t_all = time.time()
        query_vector = np.asarray(embedding_fn(query))
        emb = time.time() - t_all
        if query_vector.shape[0] != self.dimension:
            raise ValueError(f"查詢向量維度不匹配: 預期 {self.dimension}，實際 {query_vector.shape[0]}")
        top, st, kind = self._run(query_vector, k * 3, L_search, beam_width, use_pq_search)   # search_k = 3k (:724)
        final, seen = [], set()
        for dist, idx in top:
            td = self._get_text_by_index(int(idx))
            if not td:
                continue
            text, md = td
            if isinstance(md, str):
                try:
                    md = json.loads(md)
                except json.JSONDecodeError:
                    md = {"id": int(idx), "text": text}
            mtype = md.get("type")
            if not mtype:
                nested = md.get("metadata")
                if isinstance(nested, str):
                    try:
                        nested = json.loads(nested)
                    except json.JSONDecodeError:
                        nested = None
                if isinstance(nested, dict):
                    mtype = nested.get("type")
            if mtype != "faq":
                continue
            qa = md.get("qa_id")
            if not qa or qa in seen:
                continue
            seen.add(qa)
            final.append({"text": text, "distance": float(dist), "metadata": md})
            if len(final) >= k:
                break
        return {"results": final,
                "timing": {"embedding_time": emb, "search_time": st.get("search_time", 0),
                           "total_time": time.time() - t_all},
                "stats": {"search_type": kind, "nodes_visited": st.get("nodes_visited", 0), "k": k,
                          "L_search": L_search, "total_results_before_dedup": len(top),
                          "final_results_after_dedup": len(final)}}


SearchEngine = SearchEngineCorrect   # search_engine.py:816
