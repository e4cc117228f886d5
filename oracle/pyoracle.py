"""ctypes binding of the CPU oracle (oracle/diskrag_oracle.c).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (diskrag_amd) must never import this module.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libdiskrag_oracle.so"
PAD = 0xFFFFFFFF
M1, M2, M3, M4 = 1, 2, 3, 4
PQ = 5    # engine mode DR_MODE_PQ (no reference counterpart): M1's loop on squared ADC distances only
F_USE_PQ, F_CYTHON, F_QUERY_F64, F_PAIRWISE, F_RERANK, F_COSINE = 1, 2, 4, 8, 16, 32
PQB = 6   # engine mode DR_MODE_PQB (no reference counterpart): the batch-per-step ADC beam search (diskrag_oracle.c pqb_search_one)
def F_POPS(n): return (int(n) & 15) << 8     # PQB: frontier entries expanded per step
def F_RERANK_TOP(n): return (int(n) & 1023) << 12     # PQB + F_RERANK: rerank only the n entries with the smallest ADC (engine flag DR_F_RERANK_TOP)
F_IP = 128    # with F_RERANK: engine flag DR_F_IP (unit-norm data: distance = |q - v|^2 / 2 = 1 - <q, v>)
F_NO_VISITED_SET = 64     # PQ mode: the statement without a visited set (engine flag DR_F_NO_VISITED_SET): same ids / distances, evaluation counters

_lib = None


def build(force=False):
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < max(
            (HERE / "diskrag_oracle.c").stat().st_mtime, (HERE / "oracle_core.inc").stat().st_mtime):
        subprocess.check_call(["make", "-s", "-C", str(HERE), "-B"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        fp, dp, u8p, u32p = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                             C.POINTER(C.c_uint32))
        L.orc_sqdist_f32.restype = C.c_float
        L.orc_sqdist_f32.argtypes = [fp, fp, C.c_uint32]
        L.orc_sqdist_f64.restype = C.c_double
        L.orc_sqdist_f64.argtypes = [fp, dp, C.c_uint32]
        L.orc_build_lut_f32.restype = None
        L.orc_build_lut_f32.argtypes = [fp, fp, C.c_uint32, C.c_uint32, fp]
        L.orc_build_lut_f64.restype = None
        L.orc_build_lut_f64.argtypes = [fp, dp, C.c_uint32, C.c_uint32, fp]
        L.orc_adc.restype = None
        L.orc_adc.argtypes = [fp, u8p, C.c_uint64, C.c_uint32, fp, fp]
        L.orc_search_batch.restype = C.c_int
        L.orc_search_batch.argtypes = [fp, u32p, u8p, fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u32p, dp, u32p, u32p]
        L.orc_mt_doubles.restype = None
        L.orc_mt_doubles.argtypes = [C.c_uint32, C.c_uint32, dp]
        L.orc_bruteforce_topk.restype = None
        L.orc_bruteforce_topk.argtypes = [fp, C.c_uint64, C.c_uint32, fp, C.c_uint32, C.c_uint32, C.c_int, u32p]
        L.orc_l2_seq_f32.restype = C.c_float
        L.orc_l2_seq_f32.argtypes = [fp, fp, C.c_uint32]
        L.orc_cosine_dist_f32.restype = C.c_float
        L.orc_cosine_dist_f32.argtypes = [fp, fp, C.c_uint32]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def sqdist(v, q):
    v = np.ascontiguousarray(v, dtype=np.float32)
    if q.dtype == np.float64:
        q = np.ascontiguousarray(q)
        return lib().orc_sqdist_f64(_p(v, C.c_float), _p(q, C.c_double), v.size)
    q = np.ascontiguousarray(q, dtype=np.float32)
    return np.float32(lib().orc_sqdist_f32(_p(v, C.c_float), _p(q, C.c_float), v.size))


def build_lut(codebook, q):
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    m, kk, sd = cb.shape
    assert kk == 256
    lut = np.empty((m, 256), dtype=np.float32)
    if q.dtype == np.float64:
        q = np.ascontiguousarray(q)
        lib().orc_build_lut_f64(_p(cb, C.c_float), _p(q, C.c_double), m, sd, _p(lut, C.c_float))
    else:
        q = np.ascontiguousarray(q, dtype=np.float32)
        lib().orc_build_lut_f32(_p(cb, C.c_float), _p(q, C.c_float), m, sd, _p(lut, C.c_float))
    return lut


def adc(lut, codes):
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, m = codes.shape
    sq = np.empty(n, dtype=np.float32)
    rt = np.empty(n, dtype=np.float32)
    lib().orc_adc(_p(lut, C.c_float), _p(codes, C.c_uint8), n, m, _p(sq, C.c_float), _p(rt, C.c_float))
    return sq, rt


def search_batch(vectors, adj, queries, medoid, mode, k, L=100, bw=0, policy=0, flags=0, codes=None,
                 codebook=None, nthreads=1):
    """Returns ids[nq,k] u32, dist[nq,k] f64, count[nq] u32, stats[nq,4] u32."""
    vectors = np.ascontiguousarray(vectors, dtype=np.float32)
    adj = np.ascontiguousarray(adj, dtype=np.uint32)
    N, D = vectors.shape
    R = adj.shape[1]
    m = 0
    if codes is not None:
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        codebook = np.ascontiguousarray(codebook, dtype=np.float32)
        m = codes.shape[1]
    queries = np.ascontiguousarray(queries)
    if queries.dtype == np.float64:
        flags |= F_QUERY_F64
    else:
        queries = np.ascontiguousarray(queries, dtype=np.float32)
    nq = queries.shape[0]
    ids = np.empty((nq, k), dtype=np.uint32)
    dist = np.empty((nq, k), dtype=np.float64)
    cnt = np.empty(nq, dtype=np.uint32)
    stats = np.empty((nq, 4), dtype=np.uint32)
    rc = lib().orc_search_batch(_p(vectors, C.c_float), _p(adj, C.c_uint32), _p(codes, C.c_uint8),
                                _p(codebook, C.c_float), N, D, R, m, int(medoid),
                                queries.ctypes.data_as(C.c_void_p), nq, mode, k, L, bw, policy, flags, nthreads,
                                _p(ids, C.c_uint32), _p(dist, C.c_double), _p(cnt, C.c_uint32),
                                _p(stats, C.c_uint32))
    if rc != 0:
        raise RuntimeError(f"orc_search_batch failed rc={rc}")
    return ids, dist, cnt, stats


def POLICY_COIN(seed0):
    """band policy: the reference's coin flip itself (np.random.random() < 0.2), as if np.random.seed(seed0 + qi) ran before query qi"""
    return 2 | ((int(seed0) & 0xFFFFFF) << 8)


def mt_doubles(seed, n):
    out = np.empty(n, dtype=np.float64)
    lib().orc_mt_doubles(int(seed), int(n), _p(out, C.c_double))
    return out


def bruteforce_topk(vectors, queries, k, nthreads=1):
    vectors = np.ascontiguousarray(vectors, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    out = np.empty((queries.shape[0], k), dtype=np.uint32)
    lib().orc_bruteforce_topk(_p(vectors, C.c_float), vectors.shape[0], vectors.shape[1], _p(queries, C.c_float),
                              queries.shape[0], k, nthreads, _p(out, C.c_uint32))
    return out


def l2_seq(x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    return np.float32(lib().orc_l2_seq_f32(_p(x, C.c_float), _p(y, C.c_float), x.size))


def cosine_dist(x, y):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    return np.float32(lib().orc_cosine_dist_f32(_p(x, C.c_float), _p(y, C.c_float), x.size))
