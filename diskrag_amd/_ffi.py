"""ctypes binding of libdiskrag_hip.so (include/diskrag_hip.h).

No PyTorch, no CPU fallback: if the shared library is missing or no HIP device is visible every call raises.
"""
import ctypes as C
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
import os as _os
LIB_PATH = Path(_os.environ["DR_LIB"]) if _os.environ.get("DR_LIB") else PKG_DIR / "libdiskrag_hip.so"

PAD = 0xFFFFFFFF
MODE_M1, MODE_M2, MODE_M3, MODE_M4 = 1, 2, 3, 4
PIPE_DEPTH = 4          # DR_PIPE_DEPTH (csrc/engine.hip): LAUNCHES of the pipelined path in flight per handle
MAX_TICKETS = 128        # DR_MAX_TICKETS (include/diskrag_hip.h): dr_search_submit tickets in flight (small submits share launches)
MODE_PQ = 5      # engine mode without a reference counterpart: M1's loop on squared ADC distances only (diskrag_hip.h)
MODE_PQB = 6     # the engine's PQ-only traversal as a batch per step on a total (distance, id) order (diskrag_hip.h DR_MODE_PQB; round 5)
F_USE_PQ, F_SQDIST, F_RERANK, F_COSINE, F_NO_VISITED_SET = 1, 2, 4, 8, 16
F_IP = 32        # with F_RERANK: inner-product metric on unit-norm data (out_dist = 1 - <q, v>; DR_F_IP)


def POLICY_COIN(seed0):
    """band_policy: the reference's coin flip itself (np.random.random() < 0.2), as if np.random.seed(seed0 + i) ran before query i (DR_POLICY_COIN)"""
    return 2 | ((int(seed0) & 0xFFFFFF) << 8)


def F_POPS(n):
    """DR_MODE_PQB: frontier entries expanded per step (DR_F_POPS)."""
    return (int(n) & 15) << 8
def F_RERANK_TOP(n):
    """DR_MODE_PQB | DR_F_RERANK: rerank only the n list entries with the smallest ADC (DR_F_RERANK_TOP; 0 = the whole list)."""
    if not 0 <= int(n) <= 1023:
        raise ValueError("DR_F_RERANK_TOP takes 0 ... 1023")
    return int(n) << 12


TIER_HBM, TIER_HOST = 0, 1       # where the full-precision rows live (dr_index_*_tiered)
MAX_RESIDENT = 16
COMM_ID_BYTES = 128

E_ARG, E_NODEVICE, E_IO, E_NOPQ, E_OVERFLOW, E_UNSUPPORTED, E_REMOTE = -1, -2, -3, -4, -5, -6, -7


class DrStats(C.Structure):
    _fields_ = [("steps", C.c_uint32), ("visited", C.c_uint32), ("exact", C.c_uint32), ("pq", C.c_uint32),
                ("status", C.c_uint32), ("inserts", C.c_uint32), ("pq_evaluated", C.c_uint32), ("adj_prefetch_hits", C.c_uint32)]


class DrTiming(C.Structure):
    _fields_ = [("h2d_ms", C.c_float), ("search_kernel_ms", C.c_float), ("finalize_kernel_ms", C.c_float),
                ("d2h_ms", C.c_float), ("total_ms", C.c_float), ("grid", C.c_uint32), ("block", C.c_uint32),
                ("lds_bytes", C.c_uint32), ("waves_per_cu", C.c_uint32), ("variant", C.c_uint32), ("lut_kernel_ms", C.c_float)]


STATS_DTYPE = np.dtype([("steps", "<u4"), ("visited", "<u4"), ("exact", "<u4"), ("pq", "<u4"), ("status", "<u4"),
                        ("inserts", "<u4"), ("pq_evaluated", "<u4"), ("adj_prefetch_hits", "<u4")])

# every symbol include/diskrag_hip.h declares
EXPORTS = ["dr_device_count", "dr_last_error", "dr_index_open", "dr_index_create", "dr_index_set_pq",
           "dr_index_set_adjacency", "dr_index_open_tiered", "dr_index_create_tiered", "dr_index_create_empty_tiered", "dr_index_write_rows", "dr_search_batch", "dr_batch_upload", "dr_batch_run", "dr_batch_download",
           "dr_get_timing", "dr_exact_distances", "dr_distance_table", "dr_adc", "dr_pq_scan",
           "dr_bruteforce_topk", "dr_get_node", "dr_index_close", "dr_index_create_empty", "dr_build_vamana",
           "dr_get_adjacency", "dr_pq_train", "dr_pq_encode", "dr_debug_phase_cycles", "dr_batch_sync",
           "dr_debug_force_kind", "dr_search_batch_f64",
           "dr_index_create_codes", "dr_index_drop_vectors", "dr_index_attach_row_file", "dr_pq_scan_best",
           "dr_batch_select", "dr_search_submit", "dr_search_wait", "dr_search_flush", "dr_set_coalesce", "dr_pipeline_stats", "dr_debug_hold", "dr_host_alloc", "dr_host_free",
           "dr_comm_unique_id", "dr_comm_init", "dr_comm_rank", "dr_comm_destroy", "dr_sharded_search", "dr_sharded_submit", "dr_sharded_wait", "dr_sharded_set_group", "dr_sharded_flush", "dr_merge_topk",
           "dr_debug_prune", "dr_debug_prune_pq", "dr_pq_train_ex", "dr_index_create_codes_empty", "dr_pq_encode_rows", "dr_build_vamana_pq",
           "dr_scalar_kernels", "dr_index_inline_codes", "dr_pq_scan_topk", "dr_index_copy_codes"]

_lib = None


class DiskragHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdiskrag_hip error {code}: {msg}")
        self.code = code


def load_library():
    """Loads libdiskrag_hip.so. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C diskrag_amd/csrc` "
                          f"(or __graft_entry__.build()). There is no CPU fallback.")
    L = C.CDLL(str(LIB_PATH))
    vp, fp, u8p, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    L.dr_device_count.restype = C.c_int
    L.dr_device_count.argtypes = []
    L.dr_last_error.restype = C.c_char_p
    L.dr_last_error.argtypes = []
    L.dr_index_open.restype = C.c_int
    L.dr_index_open.argtypes = [C.POINTER(vp), C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.dr_index_create.restype = C.c_int
    L.dr_index_create.argtypes = [C.POINTER(vp), fp, u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.dr_index_open_tiered.restype = C.c_int
    L.dr_index_open_tiered.argtypes = [C.POINTER(vp), C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.dr_index_create_tiered.restype = C.c_int
    L.dr_index_create_tiered.argtypes = [C.POINTER(vp), fp, u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.dr_index_create_empty_tiered.restype = C.c_int
    L.dr_index_create_empty_tiered.argtypes = [C.POINTER(vp), fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32]
    L.dr_index_write_rows.restype = C.c_int
    L.dr_index_write_rows.argtypes = [vp, fp, C.c_uint64, C.c_uint64]
    L.dr_pq_scan_best.restype = C.c_int
    L.dr_pq_scan_best.argtypes = [vp, fp, C.c_uint32, fp, u32p, fp, fp]
    L.dr_index_create_codes.restype = C.c_int
    L.dr_index_create_codes.argtypes = [C.POINTER(vp), u32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, fp, u8p,
                                        C.c_uint32, C.c_int]
    L.dr_index_drop_vectors.restype = C.c_int
    L.dr_index_drop_vectors.argtypes = [vp]
    L.dr_index_attach_row_file.restype = C.c_int
    L.dr_index_attach_row_file.argtypes = [vp, C.c_char_p, C.c_uint64, C.c_uint64]
    L.dr_index_set_pq.restype = C.c_int
    L.dr_index_set_pq.argtypes = [vp, fp, u8p, C.c_uint32]
    L.dr_index_set_adjacency.restype = C.c_int
    L.dr_index_set_adjacency.argtypes = [vp, u32p]
    L.dr_search_batch.restype = C.c_int
    L.dr_search_batch.argtypes = [vp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, u32p, fp, u32p, C.POINTER(DrStats)]
    L.dr_batch_upload.restype = C.c_int
    L.dr_batch_upload.argtypes = [vp, fp, C.c_uint32]
    L.dr_batch_run.restype = C.c_int
    L.dr_batch_run.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.dr_batch_sync.restype = C.c_int
    L.dr_batch_sync.argtypes = [vp]
    L.dr_batch_download.restype = C.c_int
    L.dr_batch_download.argtypes = [vp, u32p, fp, u32p, C.POINTER(DrStats)]
    L.dr_get_timing.restype = C.c_int
    L.dr_get_timing.argtypes = [vp, C.POINTER(DrTiming)]
    L.dr_exact_distances.restype = C.c_int
    L.dr_exact_distances.argtypes = [vp, fp, C.c_uint32, u32p, C.c_uint32, fp]
    L.dr_distance_table.restype = C.c_int
    L.dr_distance_table.argtypes = [vp, fp, C.c_uint32, fp]
    L.dr_adc.restype = C.c_int
    L.dr_adc.argtypes = [vp, fp, C.c_uint32, u32p, C.c_uint32, fp, fp]
    L.dr_pq_scan.restype = C.c_int
    L.dr_pq_scan.argtypes = [vp, fp, C.c_uint32, fp, fp]
    L.dr_bruteforce_topk.restype = C.c_int
    L.dr_bruteforce_topk.argtypes = [vp, fp, C.c_uint32, C.c_uint32, u32p, fp]
    L.dr_get_node.restype = C.c_int
    L.dr_get_node.argtypes = [vp, C.c_uint64, fp, u32p]
    L.dr_index_create_empty.restype = C.c_int
    L.dr_index_create_empty.argtypes = [C.POINTER(vp), fp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    L.dr_build_vamana.restype = C.c_int
    L.dr_build_vamana.argtypes = [vp, C.c_uint32, C.c_float, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                  u32p, fp]
    L.dr_get_adjacency.restype = C.c_int
    L.dr_get_adjacency.argtypes = [vp, u32p]
    L.dr_pq_train.restype = C.c_int
    L.dr_pq_train.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, fp]
    L.dr_index_create_codes_empty.restype = C.c_int
    L.dr_index_create_codes_empty.argtypes = [C.POINTER(vp), C.c_uint64, C.c_uint32, C.c_uint32, fp, C.c_uint32, C.c_int]
    L.dr_pq_encode_rows.restype = C.c_int
    L.dr_pq_encode_rows.argtypes = [vp, fp, C.c_uint64, C.c_uint64]
    L.dr_build_vamana_pq.restype = C.c_int
    L.dr_build_vamana_pq.argtypes = [vp, C.c_uint32, C.c_float, C.c_uint32, C.c_uint64, C.c_uint32, u32p, fp]
    L.dr_scalar_kernels.restype = C.c_int
    L.dr_scalar_kernels.argtypes = [C.c_int, fp, fp, C.c_uint32, C.c_uint32, fp, fp]
    L.dr_pq_train_ex.restype = C.c_int
    L.dr_pq_train_ex.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, fp, C.POINTER(C.c_double)]
    L.dr_pq_encode.restype = C.c_int
    L.dr_pq_encode.argtypes = [vp, fp, C.c_uint32, u8p]
    L.dr_search_batch_f64.restype = C.c_int
    L.dr_search_batch_f64.argtypes = [vp, C.POINTER(C.c_double)] + [C.c_uint32] * 7 + [u32p, C.POINTER(C.c_double), u32p,
                                                                                      C.POINTER(DrStats)]
    L.dr_index_inline_codes.restype = C.c_int
    L.dr_index_inline_codes.argtypes = [vp, C.c_int]
    L.dr_debug_force_kind.restype = C.c_int
    L.dr_debug_force_kind.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
    L.dr_debug_phase_cycles.restype = C.c_int
    L.dr_debug_phase_cycles.argtypes = [vp, C.POINTER(C.c_double)]
    L.dr_index_close.restype = None
    L.dr_index_close.argtypes = [vp]
    L.dr_batch_select.restype = C.c_int
    L.dr_batch_select.argtypes = [vp, C.c_uint32]
    L.dr_search_submit.restype = C.c_int
    L.dr_search_submit.argtypes = [vp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_uint32, u32p, fp, u32p, C.POINTER(DrStats), C.POINTER(C.c_uint64)]
    L.dr_search_wait.restype = C.c_int
    L.dr_search_wait.argtypes = [vp, C.c_uint64]
    L.dr_search_flush.restype = C.c_int
    L.dr_search_flush.argtypes = [vp]
    L.dr_set_coalesce.restype = C.c_int
    L.dr_set_coalesce.argtypes = [vp, C.c_uint32]
    L.dr_pipeline_stats.restype = C.c_int
    L.dr_pipeline_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.dr_debug_hold.restype = C.c_int
    L.dr_debug_hold.argtypes = [vp, C.c_int]
    L.dr_host_alloc.restype = C.c_void_p
    L.dr_host_alloc.argtypes = [C.c_uint64]
    L.dr_host_free.restype = None
    L.dr_host_free.argtypes = [C.c_void_p]
    L.dr_comm_unique_id.restype = C.c_int
    L.dr_comm_unique_id.argtypes = [C.c_void_p]
    L.dr_comm_init.restype = C.c_int
    L.dr_comm_init.argtypes = [C.POINTER(vp), C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.dr_comm_rank.restype = C.c_int
    L.dr_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dr_comm_destroy.restype = None
    L.dr_comm_destroy.argtypes = [vp]
    L.dr_sharded_search.restype = C.c_int
    L.dr_sharded_search.argtypes = [C.POINTER(vp), u32p, C.c_uint32, vp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp, u32p, fp]
    L.dr_sharded_submit.restype = C.c_int
    L.dr_sharded_submit.argtypes = [C.POINTER(vp), u32p, C.c_uint32, vp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp, u32p, fp, C.POINTER(C.c_uint64)]
    L.dr_sharded_wait.restype = C.c_int
    L.dr_sharded_wait.argtypes = [vp, C.c_uint64]
    L.dr_sharded_set_group.restype = C.c_int
    L.dr_sharded_set_group.argtypes = [vp, C.c_uint32]
    L.dr_sharded_flush.restype = C.c_int
    L.dr_sharded_flush.argtypes = [vp]
    L.dr_debug_prune.restype = C.c_int
    L.dr_debug_prune.argtypes = [vp, C.c_uint32, u32p, C.c_uint32, C.c_float, C.c_uint32, u32p, u32p]
    L.dr_merge_topk.restype = C.c_int
    L.dr_merge_topk.argtypes = [C.c_int, u32p, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, fp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        msg = load_library().dr_last_error()
        raise DiskragHipError(rc, msg.decode("utf-8", "replace") if msg else "")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class HipIndex:
    """An index resident in HBM (handle of the C ABI)."""

    def __init__(self, handle, N, D, R, medoid):
        self._h = handle
        self.N, self.D, self.R, self.medoid = int(N), int(D), int(R), int(medoid)
        self.m = 0

    # -- construction
    @classmethod
    def open(cls, index_dat, N, D, R, medoid, device=0, vector_tier=TIER_HBM):
        """vector_tier=TIER_HOST keeps the full-precision rows in pinned host memory (graph and codes stay in HBM)."""
        L = load_library()
        h = C.c_void_p()
        _check(L.dr_index_open_tiered(C.byref(h), str(index_dat).encode(), int(N), int(D), int(R), int(medoid), int(device), int(vector_tier)))
        return cls(h, N, D, R, medoid)

    @classmethod
    def create(cls, vectors, adj, medoid, device=0, vector_tier=TIER_HBM):
        L = load_library()
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        adj = np.ascontiguousarray(adj, dtype=np.uint32)
        N, D = vectors.shape
        if adj.shape[0] != N:
            raise ValueError("adjacency rows != number of vectors")
        h = C.c_void_p()
        _check(L.dr_index_create_tiered(C.byref(h), _p(vectors, C.c_float), _p(adj, C.c_uint32), N, D, adj.shape[1],
                                        int(medoid), int(device), int(vector_tier)))
        return cls(h, N, D, adj.shape[1], medoid)

    @classmethod
    def create_empty(cls, vectors, R, device=0, vector_tier=TIER_HBM):
        """Vectors only; the graph is then built on the device with build_vamana()."""
        L = load_library()
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        N, D = vectors.shape
        h = C.c_void_p()
        _check(L.dr_index_create_empty_tiered(C.byref(h), _p(vectors, C.c_float), N, D, int(R), int(device), int(vector_tier)))
        return cls(h, N, D, R, 0)

    @classmethod
    def create_rows_empty(cls, N, D, R, device=0, vector_tier=TIER_HBM):
        """An index of N x D rows that arrive later through write_rows() (a stream: generated, or read chunk by chunk); the
        graph is then built on the device. With TIER_HOST the rows live in pinned host memory."""
        h = C.c_void_p()
        _check(load_library().dr_index_create_empty_tiered(C.byref(h), None, int(N), int(D), int(R), int(device), int(vector_tier)))
        return cls(h, int(N), int(D), int(R), 0)

    def write_rows(self, rows, row0):
        v = np.ascontiguousarray(rows, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.D:
            raise ValueError(f"rows must be [n, {self.D}]")
        _check(load_library().dr_index_write_rows(self._h, _p(v, C.c_float), int(row0), v.shape[0]))

    @classmethod
    def create_codes(cls, adj, medoid, D, codebook, codes, device=0):
        """PQ-only shard (config c5): adjacency + codes + codebook, no stored vectors (M3 with F_USE_PQ only)."""
        L = load_library()
        adj = np.ascontiguousarray(adj, dtype=np.uint32)
        codebook = np.ascontiguousarray(codebook, dtype=np.float32)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        N, R = adj.shape
        m = codes.shape[1]
        if codebook.shape != (m, 256, D // m) or codes.shape[0] != N:
            raise ValueError(f"PQ shapes do not match the index: codebook {codebook.shape}, codes {codes.shape}")
        h = C.c_void_p()
        _check(L.dr_index_create_codes(C.byref(h), _p(adj, C.c_uint32), N, int(D), R, int(medoid), _p(codebook, C.c_float),
                                       _p(codes, C.c_uint8), m, int(device)))
        ix = cls(h, N, int(D), R, int(medoid))
        ix.m = m
        return ix

    @classmethod
    def create_codes_empty(cls, N, D, R, codebook, device=0):
        """A PQ-only shard to be filled by encode_rows() and built by build_vamana_pq() (config c5: vectors never stored)."""
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        m = cb.shape[0]
        if cb.shape != (m, 256, D // m):
            raise ValueError(f"codebook shape {cb.shape} does not match D={D}")
        h = C.c_void_p()
        _check(load_library().dr_index_create_codes_empty(C.byref(h), int(N), int(D), int(R), _p(cb, C.c_float), m, int(device)))
        ix = cls(h, N, D, R, 0)
        ix.m = m
        return ix

    def copy_codes_from(self, other):
        """Code table + codebook of `other` (same N, D, device) copied on the device (dr_index_copy_codes)."""
        L = load_library()
        L.dr_index_copy_codes.restype = C.c_int
        L.dr_index_copy_codes.argtypes = [C.c_void_p, C.c_void_p]
        _check(L.dr_index_copy_codes(self._h, other._h))
        self.m = other.m

    def encode_rows(self, vectors, row0):
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.D:
            raise ValueError(f"vectors must be [rows, {self.D}]")
        _check(load_library().dr_pq_encode_rows(self._h, _p(v, C.c_float), int(row0), v.shape[0]))

    def build_vamana_pq(self, L_build=64, alpha=1.2, passes=2, seed=1, max_batch=0):
        med = C.c_uint32(0)
        secs = C.c_float(0)
        _check(load_library().dr_build_vamana_pq(self._h, int(L_build), float(alpha), int(passes), int(seed), int(max_batch),
                                                 C.byref(med), C.byref(secs)))
        self.medoid = int(med.value)
        return self.medoid, float(secs.value)

    def drop_vectors(self):
        """Frees the stored vectors: the index becomes a PQ-only shard (M3 with F_USE_PQ only)."""
        _check(load_library().dr_index_drop_vectors(self._h))

    def attach_row_file(self, index_dat, record_bytes=0, vector_offset=0):
        """Disk tier: the rows of a PQ-only index live in `index_dat` (the reference's record layout by default); DR_F_RERANK reads them from there."""
        _check(load_library().dr_index_attach_row_file(self._h, str(index_dat).encode(), int(record_bytes), int(vector_offset)))

    def build_vamana(self, L_build=100, alpha=1.2, passes=2, seed=1, pad_with_zero=True, max_batch=0):
        med = C.c_uint32(0)
        secs = C.c_float(0)
        _check(load_library().dr_build_vamana(self._h, int(L_build), float(alpha), int(passes), int(seed),
                                              1 if pad_with_zero else 0, int(max_batch), C.byref(med),
                                              C.byref(secs)))
        self.medoid = int(med.value)
        return self.medoid, float(secs.value)

    def get_adjacency(self):
        out = np.empty((self.N, self.R), dtype=np.uint32)
        _check(load_library().dr_get_adjacency(self._h, _p(out, C.c_uint32)))
        return out

    def pq_train(self, m, n_sample=100000, iters=10, seed=42):
        cb = np.empty((m, 256, self.D // m), dtype=np.float32)
        _check(load_library().dr_pq_train(self._h, int(m), int(n_sample), int(iters), int(seed), _p(cb, C.c_float)))
        return cb

    def pq_train_ex(self, m, n_sample=100000, max_iter=300, n_init=3, tol=1e-4, seed=42):
        """k-means++ seeding, n_init restarts, sklearn's stopping rule (DiskANNPQ.fit): (codebook, inertia on the sample)."""
        cb = np.empty((m, 256, self.D // m), dtype=np.float32)
        inertia = C.c_double(0)
        _check(load_library().dr_pq_train_ex(self._h, int(m), int(n_sample), int(max_iter), int(n_init), float(tol), int(seed),
                                             _p(cb, C.c_float), C.byref(inertia)))
        return cb, float(inertia.value)

    def pq_encode(self, codebook, want_codes=False):
        cb = np.ascontiguousarray(codebook, dtype=np.float32)
        m = cb.shape[0]
        codes = np.empty((self.N, m), dtype=np.uint8) if want_codes else None
        _check(load_library().dr_pq_encode(self._h, _p(cb, C.c_float), m, _p(codes, C.c_uint8)))
        self.m = m
        return codes

    def set_pq(self, codebook, codes):
        codebook = np.ascontiguousarray(codebook, dtype=np.float32)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        m = codes.shape[1]
        if codebook.shape != (m, 256, self.D // m) or codes.shape[0] != self.N:
            raise ValueError(f"PQ shapes do not match the index: codebook {codebook.shape}, codes {codes.shape}")
        _check(load_library().dr_index_set_pq(self._h, _p(codebook, C.c_float), _p(codes, C.c_uint8), m))
        self.m = m

    def set_adjacency(self, adj):
        adj = np.ascontiguousarray(adj, dtype=np.uint32)
        if adj.shape != (self.N, self.R):
            raise ValueError("adjacency shape mismatch")
        _check(load_library().dr_index_set_adjacency(self._h, _p(adj, C.c_uint32)))

    def close(self):
        if self._h:
            load_library().dr_index_close(self._h)       # drains every stream; unfinished jobs are dropped, not delivered
            self._h = None
            self.__dict__.pop("_inflight", None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- search
    def _queries(self, queries):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.D:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.D}")
        return q

    def search_batch(self, queries, k, L=100, beam_width=0, mode=MODE_M1, band_policy=0, flags=0):
        q = self._queries(queries)
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.uint32)
        dist = np.empty((nq, k), dtype=np.float32)
        cnt = np.empty(nq, dtype=np.uint32)
        stats = np.empty(nq, dtype=STATS_DTYPE)
        _check(load_library().dr_search_batch(self._h, _p(q, C.c_float), nq, int(k), int(L), int(beam_width or 0),
                                              int(mode), int(band_policy), int(flags), _p(ids, C.c_uint32),
                                              _p(dist, C.c_float), _p(cnt, C.c_uint32),
                                              stats.ctypes.data_as(C.POINTER(DrStats))))
        return ids, dist, cnt, stats

    def search_batch_f64(self, queries, k, L=100, beam_width=0, mode=MODE_M1, band_policy=0, flags=0):
        """float64 queries (the CLI path, quirk Q8): M1 / M2, float64 distances back."""
        q = np.ascontiguousarray(queries, dtype=np.float64)
        if q.ndim == 1:
            q = q[None, :]
        if q.ndim != 2 or q.shape[1] != self.D:
            raise ValueError(f"queries must be [nq, {self.D}], got {q.shape}")
        nq = q.shape[0]
        ids = np.empty((nq, k), dtype=np.uint32)
        dist = np.empty((nq, k), dtype=np.float64)
        cnt = np.empty(nq, dtype=np.uint32)
        stats = np.empty(nq, dtype=STATS_DTYPE)
        _check(load_library().dr_search_batch_f64(self._h, _p(q, C.c_double), nq, int(k), int(L), int(beam_width or 0),
                                                  int(mode), int(band_policy), int(flags), _p(ids, C.c_uint32),
                                                  _p(dist, C.c_double), _p(cnt, C.c_uint32),
                                                  stats.ctypes.data_as(C.POINTER(DrStats))))
        return ids, dist, cnt, stats

    def batch_select(self, slot):
        """Selects which of the MAX_RESIDENT resident batches batch_upload / batch_run refer to (slot 0 at creation)."""
        _check(load_library().dr_batch_select(self._h, int(slot)))

    def search_submit(self, queries, k, L=100, beam_width=0, mode=MODE_M1, band_policy=0, flags=0, reuse_outputs=False):
        """Pipelined dr_search_batch: queues upload, search, tie-order pass and download and returns a PendingSearch;
        its .wait() gives (ids, dist, count, stats). Up to MAX_TICKETS submits and PIPE_DEPTH launches are in flight per index;
        small submits that find the search stream busy are coalesced into one launch (same bits per ticket; set_coalesce,
        search_flush). With reuse_outputs the result arrays come from a ring of MAX_TICKETS + 1 sets (valid until that many
        submits after this one)."""
        q = self._queries(queries)
        nq = q.shape[0]
        if reuse_outputs:
            ring = self.__dict__.setdefault("_out_ring", {})
            key = (nq, int(k))
            if key not in ring:
                ring[key] = [[PendingSearch(self, None, nq, int(k)) for _ in range(MAX_TICKETS + 1)], 0]
            sets, pos = ring[key]
            job = sets[pos % (MAX_TICKETS + 1)]
            ring[key][1] = pos + 1
            job._q = q
        else:
            job = PendingSearch(self, q, nq, int(k))
        t = C.c_uint64(0)
        _check(load_library().dr_search_submit(self._h, _p(q, C.c_float), nq, int(k), int(L), int(beam_width or 0), int(mode),
                                               int(band_policy), int(flags), _p(job.ids, C.c_uint32), _p(job.dist, C.c_float),
                                               _p(job.cnt, C.c_uint32), job.stats.ctypes.data_as(C.POINTER(DrStats)),
                                               C.byref(t)))
        job.ticket = int(t.value)
        # The library writes into the job's arrays when the batch is FINISHED -- in wait(), or earlier, when a later submit
        # reuses its pipeline slot or a build / set_pq quiesces the handle. A caller that drops the job without wait()
        # (an exception between submit and wait) must not free them under the library: the index keeps every submitted
        # job until the library can no longer touch it (its ticket slot has been reused: ticket <= newest - MAX_TICKETS).
        live = self.__dict__.setdefault("_inflight", {})
        live[job.ticket] = job
        for tk in [tk for tk in list(live) if tk + MAX_TICKETS <= job.ticket]:
            live.pop(tk, None)          # (request threads share the index: another thread may have dropped it already)
        return job

    def search_flush(self):
        """Launches whatever search_submit is still holding back for coalescing (dr_search_flush)."""
        _check(load_library().dr_search_flush(self._h))

    def set_coalesce(self, max_queries):
        """Queries a coalesced launch of small submits may grow to (dr_set_coalesce); 0: every submit is its own launch."""
        _check(load_library().dr_set_coalesce(self._h, int(max_queries)))

    def pipeline_stats(self):
        """{launches, tickets, max_tickets_per_launch, queries} of the pipelined path since the index was created."""
        out = (C.c_uint64 * 4)()
        _check(load_library().dr_pipeline_stats(self._h, out))
        return {"launches": int(out[0]), "tickets": int(out[1]), "max_tickets_per_launch": int(out[2]), "queries": int(out[3])}

    def debug_hold(self, on):
        _check(load_library().dr_debug_hold(self._h, 1 if on else 0))

    def batch_upload(self, queries):
        q = self._queries(queries)
        _check(load_library().dr_batch_upload(self._h, _p(q, C.c_float), q.shape[0]))
        self._nq = q.shape[0]

    def batch_run(self, k, L=100, beam_width=0, mode=MODE_M1, band_policy=0, flags=0):
        _check(load_library().dr_batch_run(self._h, int(k), int(L), int(beam_width or 0), int(mode),
                                           int(band_policy), int(flags)))
        self._k = int(k)

    def batch_sync(self):
        _check(load_library().dr_batch_sync(self._h))

    def batch_download(self):
        nq, k = self._nq, self._k
        if nq is None:
            # (a sharded search uses the selected resident batch as scratch -- include/diskrag_hip.h: the library would write ITS query
            # count of rows into arrays sized for the batch this handle uploaded)
            raise RuntimeError("the resident batch was overwritten by a sharded search: batch_upload it again before batch_run / batch_download")
        ids = np.empty((nq, k), dtype=np.uint32)
        dist = np.empty((nq, k), dtype=np.float32)
        cnt = np.empty(nq, dtype=np.uint32)
        stats = np.empty(nq, dtype=STATS_DTYPE)
        _check(load_library().dr_batch_download(self._h, _p(ids, C.c_uint32), _p(dist, C.c_float),
                                                _p(cnt, C.c_uint32), stats.ctypes.data_as(C.POINTER(DrStats))))
        return ids, dist, cnt, stats

    def timing(self):
        t = DrTiming()
        _check(load_library().dr_get_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in DrTiming._fields_}

    # -- kernel-level seams (B5)
    def exact_distances(self, queries, node_ids):
        q = self._queries(queries)
        ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
        out = np.empty((q.shape[0], ids.size), dtype=np.float32)
        _check(load_library().dr_exact_distances(self._h, _p(q, C.c_float), q.shape[0], _p(ids, C.c_uint32),
                                                 ids.size, _p(out, C.c_float)))
        return out

    def distance_table(self, queries):
        q = self._queries(queries)
        out = np.empty((q.shape[0], self.m, 256), dtype=np.float32)
        _check(load_library().dr_distance_table(self._h, _p(q, C.c_float), q.shape[0], _p(out, C.c_float)))
        return out

    def adc(self, queries, node_ids):
        q = self._queries(queries)
        ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
        sq = np.empty((q.shape[0], ids.size), dtype=np.float32)
        rt = np.empty((q.shape[0], ids.size), dtype=np.float32)
        _check(load_library().dr_adc(self._h, _p(q, C.c_float), q.shape[0], _p(ids, C.c_uint32), ids.size,
                                     _p(sq, C.c_float), _p(rt, C.c_float)))
        return sq, rt

    def pq_scan(self, queries, want_output=True):
        q = self._queries(queries)
        out = np.empty((q.shape[0], self.N), dtype=np.float32) if want_output else None
        ms = C.c_float(0)
        _check(load_library().dr_pq_scan(self._h, _p(q, C.c_float), q.shape[0], _p(out, C.c_float), C.byref(ms)))
        return out, ms.value

    def pq_scan_best(self, queries, want_output=False):
        """Flat scan of every code word: (nearest id, its squared ADC distance, kernel ms[, all distances])."""
        q = self._queries(queries)
        nq = q.shape[0]
        out = np.empty((nq, self.N), dtype=np.float32) if want_output else None
        bid = np.empty(nq, dtype=np.uint32)
        bsq = np.empty(nq, dtype=np.float32)
        ms = C.c_float(0)
        _check(load_library().dr_pq_scan_best(self._h, _p(q, C.c_float), nq, _p(out, C.c_float), _p(bid, C.c_uint32),
                                              _p(bsq, C.c_float), C.byref(ms)))
        return (bid, bsq, ms.value, out) if want_output else (bid, bsq, ms.value)

    def pq_scan_topk(self, queries, k):
        """Brute-force ADC search (flat scan of every code word): (ids[nq, k], squared ADC distances, scan kernel ms)."""
        q = self._queries(queries)
        ids = np.empty((q.shape[0], k), dtype=np.uint32)
        sq = np.empty((q.shape[0], k), dtype=np.float32)
        ms = C.c_float(0)
        L = load_library()
        L.dr_pq_scan_topk.restype = C.c_int
        L.dr_pq_scan_topk.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                                      C.POINTER(C.c_float)]
        _check(L.dr_pq_scan_topk(self._h, _p(q, C.c_float), q.shape[0], int(k), _p(ids, C.c_uint32), _p(sq, C.c_float), C.byref(ms)))
        return ids, sq, ms.value

    def bruteforce_topk(self, queries, k):
        q = self._queries(queries)
        ids = np.empty((q.shape[0], k), dtype=np.uint32)
        dist = np.empty((q.shape[0], k), dtype=np.float32)
        _check(load_library().dr_bruteforce_topk(self._h, _p(q, C.c_float), q.shape[0], int(k), _p(ids, C.c_uint32),
                                                 _p(dist, C.c_float)))
        return ids, dist

    def debug_prune(self, point, candidates, alpha, R):
        """The builder's robust prune of one point over an explicit candidate list: picked ids in pick order."""
        c = np.ascontiguousarray(candidates, dtype=np.uint32)
        sel = np.empty(int(R), dtype=np.uint32)
        cnt = C.c_uint32(0)
        _check(load_library().dr_debug_prune(self._h, int(point), _p(c, C.c_uint32), c.size, float(alpha), int(R),
                                             _p(sel, C.c_uint32), C.byref(cnt)))
        return sel[:int(cnt.value)]

    def debug_prune_pq(self, point, candidates, alpha, R):
        """The PQ-only builder's prune of one point over an explicit candidate list (code-word distances): picked ids in pick order."""
        c = np.ascontiguousarray(candidates, dtype=np.uint32)
        sel = np.empty(int(R), dtype=np.uint32)
        cnt = C.c_uint32(0)
        L = load_library()
        L.dr_debug_prune_pq.restype = C.c_int
        L.dr_debug_prune_pq.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_float, C.c_uint32,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _check(L.dr_debug_prune_pq(self._h, int(point), _p(c, C.c_uint32), c.size, float(alpha), int(R), _p(sel, C.c_uint32), C.byref(cnt)))
        return sel[:int(cnt.value)]

    def inline_codes(self, enable=True):
        """Keeps the code words of every node's neighbours beside its adjacency row (N*R*m bytes of HBM): one coalesced
        read per expansion for the ADC of the rerank-policy-live M1 and of the PQ-only traversals. Results unchanged."""
        _check(load_library().dr_index_inline_codes(self._h, 1 if enable else 0))

    def debug_force_kind(self, kind):
        """Pins the search-kernel variant (-1: engine's choice); returns the handle's A4 regime (1 live, 0 not, -1 unknown)."""
        live = C.c_int(-1)
        _check(load_library().dr_debug_force_kind(self._h, int(kind), C.byref(live)))
        return int(live.value)

    def debug_phase_cycles(self):
        out = (C.c_double * 8)()
        _check(load_library().dr_debug_phase_cycles(self._h, out))
        return list(out)

    def get_node(self, node_id):
        vec = np.empty(self.D, dtype=np.float32)
        nbrs = np.empty(self.R, dtype=np.uint32)
        _check(load_library().dr_get_node(self._h, int(node_id), _p(vec, C.c_float), _p(nbrs, C.c_uint32)))
        return vec, nbrs


class PendingSearch:
    """A batch in flight (HipIndex.search_submit). The query and output arrays are kept alive until wait()."""

    def __init__(self, index, q, nq, k):
        self._index, self._q, self.ticket = index, q, 0
        self.ids = np.empty((nq, k), dtype=np.uint32)
        self.dist = np.empty((nq, k), dtype=np.float32)
        self.cnt = np.empty(nq, dtype=np.uint32)
        self.stats = np.empty(nq, dtype=STATS_DTYPE)

    def wait(self):
        _check(load_library().dr_search_wait(self._index._h, self.ticket))
        self._q = None
        self._index.__dict__.get("_inflight", {}).pop(self.ticket, None)
        return self.ids, self.dist, self.cnt, self.stats


class _PinnedBlock:
    def __init__(self, nbytes):
        self.ptr = load_library().dr_host_alloc(int(nbytes))
        if not self.ptr:
            raise MemoryError(f"dr_host_alloc({nbytes}) failed")

    def __del__(self):
        try:
            load_library().dr_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    """A numpy array over page-locked host memory (dr_host_alloc): the copy engine reads / writes it without staging."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    blk = _PinnedBlock(max(n, 1))
    buf = (C.c_char * max(n, 1)).from_address(blk.ptr)
    buf._block = blk      # the array keeps `buf` alive, `buf` keeps the pinned block alive
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


class Comm:
    """RCCL communicator of the graph-sharded search (one process per GPU). Rank 0 calls Comm.unique_id() and hands the
    128 bytes to the other ranks (file, pipe, ...); every rank then constructs Comm(id, nranks, rank, device)."""

    def __init__(self, unique_id, nranks, rank, device=0):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        h = C.c_void_p()
        _check(load_library().dr_comm_init(C.byref(h), buf, int(nranks), int(rank), int(device)))
        self._h, self.rank, self.nranks = h, int(rank), int(nranks)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(load_library().dr_comm_unique_id(buf))
        return buf.raw

    def close(self):
        if self._h:
            load_library().dr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sharded_search(shards, id_bases, queries, k, L=100, beam_width=8, mode=MODE_PQ, band_policy=0, flags=0, comm=None):
    """dr_sharded_search: every index of `shards` (this rank's) searches the batch, lists are merged on the device and, with a
    Comm, all-gathered over the ranks (RCCL) and merged again. Returns (global ids, distances, status per query, ms[3])."""
    n = len(shards)
    D = shards[0].D
    q = np.ascontiguousarray(queries, dtype=np.float32)
    if q.ndim != 2 or q.shape[1] != D:
        raise ValueError(f"queries must be [nq, {D}], got {q.shape}")
    nq = q.shape[0]
    hs = (C.c_void_p * n)(*[s._h for s in shards])
    bases = np.ascontiguousarray(id_bases, dtype=np.uint32)
    ids = np.empty((nq, k), dtype=np.uint32)
    dist = np.empty((nq, k), dtype=np.float32)
    status = np.empty(nq, dtype=np.uint32)
    ms = np.zeros(3, dtype=np.float32)
    for s in shards:
        s._nq = None            # (every shard's selected resident batch becomes the exchange's scratch)
    _check(load_library().dr_sharded_search(hs, _p(bases, C.c_uint32), n, comm._h if comm is not None else None,
                                            _p(q, C.c_float), nq, int(k), int(L), int(beam_width or 0), int(mode),
                                            int(band_policy), int(flags), _p(ids, C.c_uint32), _p(dist, C.c_float),
                                            _p(status, C.c_uint32), _p(ms, C.c_float)))
    return ids, dist, status, ms


class PendingSharded:
    """A sharded search in flight (sharded_submit): wait() gives (global ids, distances, status per query, ms[3])."""

    def __init__(self, first_shard, q, nq, k):
        self._first, self._q, self.ticket = first_shard, q, 0
        self.ids = np.empty((nq, k), dtype=np.uint32)
        self.dist = np.empty((nq, k), dtype=np.float32)
        self.status = np.empty(nq, dtype=np.uint32)
        self.ms = np.zeros(3, dtype=np.float32)

    def wait(self):
        _check(load_library().dr_sharded_wait(self._first._h, self.ticket))
        self._q = None
        self._first.__dict__.get("_sharded_inflight", {}).pop(self.ticket, None)
        return self.ids, self.dist, self.status, self.ms


def sharded_submit(shards, id_bases, queries, k, L=100, beam_width=8, mode=MODE_PQ, band_policy=0, flags=0, comm=None):
    """dr_sharded_submit: the first half of sharded_search; four exchanges may be in flight per first shard (batch i+1 is searched
    while batch i is exchanged and merged). Every rank must submit in the same order."""
    n = len(shards)
    D = shards[0].D
    q = np.ascontiguousarray(queries, dtype=np.float32)
    if q.ndim != 2 or q.shape[1] != D:
        raise ValueError(f"queries must be [nq, {D}], got {q.shape}")
    nq = q.shape[0]
    hs = (C.c_void_p * n)(*[s._h for s in shards])
    bases = np.ascontiguousarray(id_bases, dtype=np.uint32)
    job = PendingSharded(shards[0], q, nq, int(k))
    t = C.c_uint64(0)
    for s in shards:
        s._nq = None            # (every shard's selected resident batch becomes the exchange's scratch)
    _check(load_library().dr_sharded_submit(hs, _p(bases, C.c_uint32), n, comm._h if comm is not None else None,
                                            _p(q, C.c_float), nq, int(k), int(L), int(beam_width or 0), int(mode),
                                            int(band_policy), int(flags), _p(job.ids, C.c_uint32), _p(job.dist, C.c_float),
                                            _p(job.status, C.c_uint32), _p(job.ms, C.c_float), C.byref(t)))
    job.ticket = int(t.value)
    live = shards[0].__dict__.setdefault("_sharded_inflight", {})      # (the library writes into the job's arrays until it is finished)
    live[job.ticket] = job
    # (four exchanges of at most 16 submits are in flight: a ticket 64 submits old has been finished by the library)
    for tk in [tk for tk in list(live) if tk + 64 <= job.ticket]:
        live.pop(tk, None)
    return job


def sharded_set_group(first_shard, n):
    """dr_sharded_set_group: n consecutive sharded_submit calls share ONE exchange (one launch per shard, one all-gather); a count, not a
    timing, so that every rank forms the same exchanges. An exchange is launched when full, when one of its jobs is waited for, or by
    sharded_flush."""
    _check(load_library().dr_sharded_set_group(first_shard._h, int(n)))


def sharded_flush(first_shard):
    _check(load_library().dr_sharded_flush(first_shard._h))


def merge_topk_device(ids, dist, k_out, device=0):
    """The device merge kernel on host arrays: ids[S, nq, k] global ids (PAD = empty), dist[S, nq, k]."""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    S, nq, k = ids.shape
    out_ids = np.empty((nq, k_out), dtype=np.uint32)
    out_dist = np.empty((nq, k_out), dtype=np.float32)
    _check(load_library().dr_merge_topk(int(device), _p(ids, C.c_uint32), _p(dist, C.c_float), S, nq, k, int(k_out),
                                        _p(out_ids, C.c_uint32), _p(out_dist, C.c_float)))
    return out_ids, out_dist


def scalar_kernels(x, y, device=0):
    """C8 on the device: (squared L2, cosine distance) of the row pairs x[i], y[i]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    if x.shape != y.shape or x.ndim != 2:
        raise ValueError("x and y must be [n, D] arrays of one shape")
    l2 = np.empty(x.shape[0], dtype=np.float32)
    cs = np.empty(x.shape[0], dtype=np.float32)
    _check(load_library().dr_scalar_kernels(int(device), _p(x, C.c_float), _p(y, C.c_float), x.shape[0], x.shape[1],
                                            _p(l2, C.c_float), _p(cs, C.c_float)))
    return l2, cs


def device_count():
    return load_library().dr_device_count()
