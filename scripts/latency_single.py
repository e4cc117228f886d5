"""Single-query latency through the C ABI (the /search and /faq-search routes ask one query at a time): wall time of
dr_search_batch with nq = 1 on the bench index, host buffers in, results out (GPU box)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from diskrag_amd import HipIndex, _ffi
from diskrag_amd.synth import sift_like
x, q = sift_like(1000000, 128, n_queries=2000, n_clusters=1024, seed=2024, query_seed=9000)
ix = HipIndex.create_empty(x, R=64)
ix.build_vamana(L_build=100, alpha=1.2, passes=2, seed=7)
cb = ix.pq_train(32, n_sample=100000, iters=5); ix.pq_encode(cb)
out = {}
for tag, kw in (("api_default_k5_L20_bw8", dict(k=5, L=20, beam_width=8)), ("k10_L100_bw8", dict(k=10, L=100, beam_width=8))):
    k = kw.pop("k")
    for i in range(50): ix.search_batch(q[i], k, mode=_ffi.MODE_M1, **kw)
    ts = []
    for i in range(1000):
        t0 = time.perf_counter(); ix.search_batch(q[i], k, mode=_ffi.MODE_M1, **kw); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    out[tag] = {"p50_ms": float(np.percentile(ts, 50)), "p99_ms": float(np.percentile(ts, 99)), "mean_ms": float(ts.mean())}
    # where the time goes: the device-side spans of the last call (HIP events) against the wall time of the call
    tm = ix.timing()
    out[tag]["device_spans_ms_last_call"] = {kk: round(float(tm[kk]), 4) for kk in ("h2d_ms", "lut_kernel_ms", "search_kernel_ms", "finalize_kernel_ms", "d2h_ms")}
    # the C ABI alone (no numpy argument conversion, no result arrays allocated): the same call through preallocated buffers
    import ctypes as C
    L_ = _ffi.load_library()
    qq = np.ascontiguousarray(q[:1000], dtype=np.float32)
    oi = np.empty(k, np.uint32); od = np.empty(k, np.float32); oc = np.empty(1, np.uint32)
    pi, pd, pc = oi.ctypes.data_as(C.POINTER(C.c_uint32)), od.ctypes.data_as(C.POINTER(C.c_float)), oc.ctypes.data_as(C.POINTER(C.c_uint32))
    t1 = []
    for i in range(1000):
        pq_ = qq[i].ctypes.data_as(C.POINTER(C.c_float))
        t0 = time.perf_counter()
        L_.dr_search_batch(ix._h, pq_, 1, k, kw["L"], kw["beam_width"], _ffi.MODE_M1, 0, 0, pi, pd, pc, None)
        t1.append(time.perf_counter() - t0)
    out[tag]["c_abi_only_p50_ms"] = float(np.percentile(np.array(t1) * 1e3, 50))
q64 = q[:200].astype(np.float64)
ts = []
for i in range(200):
    t0 = time.perf_counter(); ix.search_batch_f64(q64[i], 5, L=20, beam_width=8); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
out["cli_float64_k5_L20_bw8"] = {"p50_ms": float(np.percentile(ts, 50)), "p99_ms": float(np.percentile(ts, 99)), "mean_ms": float(ts.mean())}
print(json.dumps(out))
