"""CPU-side checks: the C-ABI library builds/loads and exports every symbol the header declares; host logic."""
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_header_symbol():
    from diskrag_amd import _ffi
    lib = _ffi.load_library()
    header = (ROOT / "include" / "diskrag_hip.h").read_text()
    declared = set(re.findall(r"\b(dr_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirrors of dr_stats / dr_timing have the C layout: sizes and field offsets from a C program
    compiled against include/diskrag_hip.h (the header is plain C: a cgo / JNI binding includes it the same way)."""
    import shutil
    import subprocess
    from diskrag_amd import _ffi
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = (("dr_stats", _ffi.DrStats), ("dr_timing", _ffi.DrTiming))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "diskrag_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == __import__("ctypes").sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, f)]) == getattr(cls, f).offset, (cname, f)


def test_no_device_fails_loudly():
    """No CPU fallback: without a HIP device index creation must raise, not degrade."""
    import diskrag_amd
    if diskrag_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(diskrag_amd.DiskragHipError):
        diskrag_amd.HipIndex.create(np.zeros((4, 128), dtype=np.float32), np.zeros((4, 2), dtype=np.uint32), 0)


def test_product_never_imports_oracle():
    for p in (ROOT / "diskrag_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".hpp", ".inc", ".h") and p.is_file():
            txt = p.read_text()
            assert "pyoracle" not in txt and "diskrag_oracle" not in txt and "oracle/" not in txt, p


def test_chain_major_permutation_is_a_bijection():
    """numerics.hpp layout: restated here to check it is a permutation and keeps each pairwise leaf in place."""
    def perm_rec(off, n, out):
        if n <= 128:
            S = n // 8; G = S // 4; rem = S % 4
            for t in range(S):
                for j in range(8):
                    g, u = divmod(t, 4)
                    out[off + 8 * t + j] = off + g * 32 + j * 4 + u if g < G else off + G * 32 + j * rem + (t - 4 * G)
        else:
            n2 = n // 2; n2 -= n2 % 8
            perm_rec(off, n2, out); perm_rec(off + n2, n - n2, out)
    for D in (32, 64, 96, 128, 256, 768, 960, 1536):
        out = [-1] * D
        perm_rec(0, D, out)
        assert sorted(out) == list(range(D))


def test_facade_argument_errors(tmp_path):
    """B3/B4 error behaviour that does not need a device (search_engine.py:22-23, 29-30, 81-85)."""
    import json
    from diskrag_amd.search_engine import SearchEngineCorrect
    with pytest.raises(ValueError):
        SearchEngineCorrect("missing", base_dir=tmp_path)
    c = tmp_path / "c1"
    (c / "index").mkdir(parents=True)
    (c / "collection_info.json").write_text(json.dumps({"dimension": 128}))
    with pytest.raises(ValueError):          # index files incomplete
        SearchEngineCorrect("c1", base_dir=tmp_path)
    (c / "index" / "index.dat").write_bytes(b"")
    (c / "index" / "meta.json").write_text(json.dumps({"N": 1, "R": 2}))
    (c / "collection_info.json").write_text(json.dumps({"dimension": 96}))
    with pytest.raises(ValueError):          # unsupported dimension (Q14)
        SearchEngineCorrect("c1", base_dir=tmp_path)


def test_python_constants_match_the_header():
    """The ctypes layer restates the header's modes, flags and tiers as Python numbers: they must be the header's."""
    from diskrag_amd import _ffi
    hdr = (Path(__file__).resolve().parent.parent / "include" / "diskrag_hip.h").read_text()
    defs = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"^#define (DR_[A-Z0-9_]+) \(?(-?(?:0x)?[0-9A-Fa-f]+u?)\)?", hdr, flags=re.M)}
    want = {"DR_MODE_M1": _ffi.MODE_M1, "DR_MODE_M2": _ffi.MODE_M2, "DR_MODE_M3": _ffi.MODE_M3, "DR_MODE_M4": _ffi.MODE_M4,
            "DR_MODE_PQ": _ffi.MODE_PQ, "DR_MODE_PQB": _ffi.MODE_PQB, "DR_F_POPS_SHIFT": 8, "DR_F_POPS_MASK": _ffi.F_POPS(15), "DR_F_USE_PQ": _ffi.F_USE_PQ, "DR_F_SQDIST": _ffi.F_SQDIST, "DR_F_RERANK": _ffi.F_RERANK,
            "DR_F_COSINE": _ffi.F_COSINE, "DR_F_IP": _ffi.F_IP, "DR_F_NO_VISITED_SET": _ffi.F_NO_VISITED_SET, "DR_TIER_HBM": _ffi.TIER_HBM, "DR_TIER_HOST": _ffi.TIER_HOST,
            "DR_MAX_TICKETS": _ffi.MAX_TICKETS, "DR_E_REMOTE": _ffi.E_REMOTE}
    for name, val in want.items():
        assert defs[name] == val, name


def test_host_byte_query_check_agrees_with_numpy():
    """dr_host_all_u8 (csrc/host_simd.cpp; the check that routes a batch to the byte-query kernel variants): every component an
    integer in [0, 255] -- the AVX2 path against the plain statement, offending values in the vector body and in the tail."""
    import ctypes as C
    import diskrag_amd
    lib = diskrag_amd.load_library()
    lib.dr_host_all_u8.restype = C.c_bool
    lib.dr_host_all_u8.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    rs = np.random.RandomState(0)

    def ask(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return bool(lib.dr_host_all_u8(a.ctypes.data_as(C.POINTER(C.c_float)), a.size))

    def plain(a):
        a = np.asarray(a, dtype=np.float32)
        with np.errstate(invalid="ignore"):
            return bool(np.all((a >= 0) & (a <= 255) & (np.trunc(a) == a)))

    for n in (0, 1, 7, 31, 32, 33, 64, 100, 4096, 4097, 1250 * 128 + 5):
        base = rs.randint(0, 256, size=n).astype(np.float32)
        assert ask(base) and plain(base)
        if n:
            for bad in (0.5, -1.0, 256.0, np.nan, np.inf, -np.inf, 1e30, 254.99998):
                for pos in {0, n // 2, n - 1}:
                    a = base.copy(); a[pos] = bad
                    assert ask(a) == plain(a) == False, (n, bad, pos)
            a = base.copy(); a[n // 2] = -0.0          # converts to the byte 0
            assert ask(a) and plain(a)
