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


def export_codebook(pq_model_pkl, out_path=None):
    """Converts the reference's pq_model.pkl (a pickle holding sklearn KMeans objects, T3,
    diskann_persist.py:33-105) into the raw float file the GPU box can read without sklearn."""
    pq_model_pkl = Path(pq_model_pkl)
    with open(pq_model_pkl, "rb") as f:
        data = pickle.load(f)
    kms = data["kmeans_list"] if isinstance(data, dict) else data.kmeans_list
    cb = np.stack([np.asarray(km.cluster_centers_, dtype=np.float32) for km in kms])
    out_path = Path(out_path) if out_path else pq_model_pkl.with_name(RAW_CODEBOOK_NAME)
    cb.tofile(out_path)
    return cb


class SearchEngineCorrect:
    def __init__(self, collection_name: str, use_thread_safe_stats: bool = True, base_dir: Optional[Path] = None,
                 device: int = 0, text_lookup: Optional[Callable[[int], Optional[Tuple[str, dict]]]] = None):
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
                                        device=device)
        self.use_pq = bool(self.meta.get("use_pq", True))
        self.n_subvectors = 0
        self.sub_dim = 0
        self.num_centroids = 0
        if self.use_pq:
            self._load_pq(index_dir)
        self._text_lookup = text_lookup
        self._meta_table = None
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
        try:
            m = int(self.meta["n_subvectors"])
            codes = np.fromfile(codes_path, dtype=np.uint8).reshape(int(self.meta["N"]), m)   # T2
            if raw_path.exists():
                cb = np.fromfile(raw_path, dtype=np.float32).reshape(m, 256, self.dimension // m)
            else:
                cb = export_codebook(pkl_path, raw_path)        # needs sklearn; raises on a box without it
            self.index.set_pq(cb, codes)
            self.n_subvectors, self.sub_dim, self.num_centroids = m, self.dimension // m, 256
        except Exception as e:  # noqa: BLE001 - mirrors the reference's blanket downgrade
            logger.warning("PQ 模型載入失敗: %s，切換到暴力搜索模式", e)
            self.use_pq = False

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
        bad = stats["status"] != 0
        if bad.any():
            raise _ffi.DiskragHipError(_ffi.E_OVERFLOW, f"work-area overflow in {int(bad.sum())} queries "
                                       f"(status bits {int(np.bitwise_or.reduce(stats['status']))})")

    @staticmethod
    def _is_f64(query_vector) -> bool:
        """The CLI hands np.array(list) = float64 (diskrag.py:194), the API float32 (app.py:45); the reference's numpy
        arithmetic follows the query's dtype (quirk Q8), and so does the engine: float64 queries take the float64
        kernel and get np.float64 distances back."""
        return np.asarray(query_vector).dtype == np.float64

    def _pq_accelerated_graph_search(self, query_vector: np.ndarray, k: int = 10, L: int = 100,
                                     beam_width: Optional[int] = None, band_policy: int = 0
                                     ) -> Tuple[List[Tuple[float, int]], Dict]:
        t0 = time.time()
        f64 = self._is_f64(query_vector)
        run = self.index.search_batch_f64 if f64 else self.index.search_batch
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
        run = self.index.search_batch_f64 if f64 else self.index.search_batch
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
        if self._text_lookup is not None:
            return self._text_lookup(int(idx))
        if self._meta_table is None:
            if not self._meta_path.exists():
                return None
            import pyarrow.parquet as pq   # read once (the reference re-reads per hit, Q17)
            self._meta_table = pq.read_table(self._meta_path).to_pylist()
        for row in self._meta_table:
            if row.get("vector_index") == idx:
                md = row.get("metadata")
                if isinstance(md, str):
                    try:
                        md = json.loads(md)
                    except json.JSONDecodeError:
                        md = {"id": idx, "text": row.get("text")}
                return row.get("text"), md if md is not None else row
        return None

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
