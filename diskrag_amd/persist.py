"""Index directory writer/reader in the reference's on-disk formats, so that an index
built on the device can be opened by the reference's `SearchEngineCorrect` and the other way round.

  index.dat      T1  N records of 4*(D+R) bytes: D float32 then R uint32 neighbour ids; short neighbour lists are
                     padded with 0, long ones cut at R (pydiskann/io/diskann_persist.py:17-24); no header
  pq_codes.bin   T2  uint8[N][m], row-major, no header (diskann_persist.py:30-31, 205-206)
  meta.json      T4  the keys search reads: N, D, R, medoid_idx, use_pq, n_subvectors (+ the build parameters the
                     reference also records, scripts/tools/build_index.py:299-332)
  pq_model.pkl   T3  pickle of a dict holding m scikit-learn KMeans objects (diskann_persist.py:33-105); written only
                     where scikit-learn is importable, because that is what the reference's loader unpickles
  pq_codebook.f32    this package's raw companion of T3: float32[m][256][D/m]; what `SearchEngineCorrect` here reads

Host-side file plumbing only: nothing in this module computes distances.
"""
import json
import os
import pickle
from collections import namedtuple
from datetime import datetime
from pathlib import Path

import numpy as np

PAD = np.uint32(0xFFFFFFFF)

IndexFiles = namedtuple("IndexFiles", "meta vectors adjacency codes codebook")


def record_dtype(D, R):
    """One T1 record as a numpy structured type (little-endian, packed)."""
    return np.dtype([("vector", "<f4", (int(D),)), ("neighbors", "<u4", (int(R),))])


def pack_neighbor_lists(neighbors, degrees, R):
    """In-memory neighbour lists -> the R on-disk slots of every record, as `save_index` lays them out
    (diskann_persist.py:22-24): the list in its stored order, zeros after it, cut at R.
    `neighbors` is [N, W] uint32 (W >= max degree); `degrees` is [N] or None (then 0xFFFFFFFF marks unused slots)."""
    nb = np.asarray(neighbors, dtype=np.uint32)
    n, w = nb.shape
    if degrees is None:
        used = nb != PAD
        # lists are dense from slot 0: the degree is the number of used slots
        deg = used.sum(axis=1)
        if not np.array_equal(used, np.arange(w)[None, :] < deg[:, None]):
            raise ValueError("neighbour lists must be dense from slot 0 when no degree array is given")
    else:
        deg = np.asarray(degrees).astype(np.int64)
        if deg.shape != (n,) or (deg < 0).any() or (deg > w).any():
            raise ValueError("degrees must be [N] with 0 <= degree <= neighbors.shape[1]")
    out = np.zeros((n, int(R)), dtype=np.uint32)
    c = min(w, int(R))
    keep = np.arange(c)[None, :] < deg[:, None]
    out[:, :c] = np.where(keep, nb[:, :c], np.uint32(0))
    return out


def write_records(path, vectors, slots):
    """Writes index.dat: record i = vectors[i] (float32) followed by slots[i] (uint32[R])."""
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    s = np.ascontiguousarray(slots, dtype=np.uint32)
    if v.ndim != 2 or s.ndim != 2 or v.shape[0] != s.shape[0]:
        raise ValueError("vectors [N,D] and slots [N,R] must have the same N")
    rec = np.empty(v.shape[0], dtype=record_dtype(v.shape[1], s.shape[1]))
    rec["vector"] = v
    rec["neighbors"] = s
    tmp = str(path) + ".tmp"
    rec.tofile(tmp)
    os.replace(tmp, path)


def read_records(path, N, D, R, mmap=True):
    """index.dat -> (vectors [N,D] float32 view, neighbour slots [N,R] uint32 view). The file size must be exactly
    N*4*(D+R) bytes (the reference's reader never checks; a wrong R silently shears every record)."""
    dt = record_dtype(D, R)
    size = os.path.getsize(path)
    if size != int(N) * dt.itemsize:
        raise ValueError(f"{path}: {size} bytes, expected N*4*(D+R) = {int(N) * dt.itemsize}")
    rec = np.memmap(path, dtype=dt, mode="r", shape=(int(N),)) if mmap else np.fromfile(path, dtype=dt)
    return rec["vector"], rec["neighbors"]


def make_pq_model_dict(codebook):
    """The dict `save_pq_codebook` pickles (diskann_persist.py:43-54), with one fitted-looking scikit-learn KMeans per
    sub-quantiser holding our centroids. Needs scikit-learn."""
    from sklearn.cluster import KMeans   # only where the reference's loader could run at all

    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    m, ncent, sd = cb.shape
    kms = []
    for j in range(m):
        km = KMeans(n_clusters=ncent, n_init=1, random_state=42)
        km.cluster_centers_ = cb[j].copy()
        km.n_features_in_ = sd
        km._n_features_out = ncent
        km._n_threads = 1
        km.labels_ = np.zeros(0, dtype=np.int32)
        km.inertia_ = 0.0
        km.n_iter_ = 0
        kms.append(km)
    return {"n_subvectors": int(m), "n_centroids": int(ncent), "sub_dim": int(sd), "is_fitted": True,
            "kmeans_list": kms, "means_": None, "stds_": None, "epsilon": 1e-8, "model_type": "DiskANNPQ",
            "version": "2.0"}


def write_index(index_dir, vectors, neighbors, medoid, R=None, degrees=None, codes=None, codebook=None,
                build_params=None, pq_pickle="auto"):
    """Writes an index directory the reference's engine can open (search_engine.py:25-79).
    `neighbors`/`degrees`: in-memory lists (see pack_neighbor_lists), or already packed slots with degrees=None and
    no 0xFFFFFFFF in them. `pq_pickle`: "auto" (write pq_model.pkl if scikit-learn is importable), True, or False.
    Returns the meta dict written."""
    d = Path(index_dir)
    d.mkdir(parents=True, exist_ok=True)
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    nb = np.asarray(neighbors, dtype=np.uint32)
    R = int(R if R is not None else nb.shape[1])
    if degrees is None and not (nb == PAD).any() and nb.shape[1] == R:
        slots = nb
    else:
        slots = pack_neighbor_lists(nb, degrees, R)
    n, D = v.shape
    if not 0 <= int(medoid) < n:
        raise ValueError("medoid out of range")
    if slots.size and int(slots.max()) >= n:
        raise ValueError("neighbour id out of range")
    write_records(d / "index.dat", v, slots)
    use_pq = codes is not None and codebook is not None
    m = 0
    if use_pq:
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        cd = np.ascontiguousarray(codes, dtype=np.uint8)
        m = cb.shape[0]
        if cb.ndim != 3 or cb.shape[0] * cb.shape[2] != D or cd.shape != (n, m):
            raise ValueError("codebook must be [m][K][D/m] and codes [N][m]")
        cd.tofile(d / "pq_codes.bin")
        cb.tofile(d / "pq_codebook.f32")
        want = pq_pickle
        if want == "auto":
            try:
                import sklearn  # noqa: F401
                want = True
            except ImportError:
                want = False
        if want:
            tmp = d / "pq_model.pkl.tmp"
            with open(tmp, "wb") as f:
                pickle.dump(make_pq_model_dict(cb), f, protocol=pickle.HIGHEST_PROTOCOL)
            os.replace(tmp, d / "pq_model.pkl")
    bp = dict(build_params or {})
    meta = {"D": int(D), "R": R, "L": int(bp.pop("L", 0)), "alpha": float(bp.pop("alpha", 1.2)), "N": int(n),
            "medoid_idx": int(medoid), "n_subvectors": int(m), "pq_centroids": int(codebook.shape[1]) if use_pq else 0,
            "build_time": datetime.now().isoformat(), "use_pq": bool(use_pq),
            "vector_stats": {"dtype": str(v.dtype), "shape": list(v.shape), "min": float(v.min()), "max": float(v.max()),
                             "mean": float(v.mean()), "std": float(v.std())}}
    meta.update(bp)
    with open(d / "meta.json", "w") as f:
        json.dump(meta, f)
    return meta


def read_index(index_dir, mmap=True):
    """Reads an index directory written by the reference or by `write_index`. The codebook comes from
    pq_codebook.f32 when present, else from pq_model.pkl (needs scikit-learn). Returns IndexFiles."""
    d = Path(index_dir)
    with open(d / "meta.json") as f:
        meta = json.load(f)
    n, R = int(meta["N"]), int(meta["R"])
    rec_bytes = os.path.getsize(d / "index.dat")
    D = int(meta["D"]) if "D" in meta else rec_bytes // (4 * n) - R
    vec, adj = read_records(d / "index.dat", n, D, R, mmap=mmap)
    codes = codebook = None
    if meta.get("use_pq") and (d / "pq_codes.bin").exists():
        m = int(meta["n_subvectors"])
        codes = np.fromfile(d / "pq_codes.bin", dtype=np.uint8)
        if codes.size != n * m:
            raise ValueError(f"pq_codes.bin holds {codes.size} bytes, expected N*m = {n * m}")
        codes = codes.reshape(n, m)
        raw = d / "pq_codebook.f32"
        if raw.exists():
            codebook = np.fromfile(raw, dtype=np.float32).reshape(m, -1, D // m)
        elif (d / "pq_model.pkl").exists():
            with open(d / "pq_model.pkl", "rb") as f:
                model = pickle.load(f)
            kms = model["kmeans_list"] if isinstance(model, dict) else model.kmeans_list
            codebook = np.stack([np.asarray(km.cluster_centers_, dtype=np.float32) for km in kms])
    return IndexFiles(meta, vec, adj, codes, codebook)
