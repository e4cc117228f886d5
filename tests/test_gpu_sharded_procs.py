"""The graph-sharded search with TWO -- and EIGHT, the size of the node the driver measures on -- REAL PROCESSES (csrc/comm.inc with nranks = 2 / 8):
communicator set-up with the unique-id hand-off through
a file, blocking and grouped exchanges, the status-word failure protocol, and a failing exchange with the bounded wait -- on ONE GPU, through
tests/fake_rccl (a stand-in librccl whose all-gather travels through shared memory, loaded by the library's own DR_RCCL_LIB hook: RCCL itself
refuses two ranks on one device, and this pool hands out one GPU). Every rank is a fresh child process started before any GPU call."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
FAKE = ROOT / "tests" / "fake_rccl"


@pytest.fixture(scope="module")
def fake_lib():
    so = FAKE / "libfake_rccl.so"
    if not so.exists() or so.stat().st_mtime < (FAKE / "fake_rccl.cpp").stat().st_mtime:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(so), str(FAKE / "fake_rccl.cpp"), "-lrt"])
    return so


def _run(scenario, fake_lib, tmp_path, extra_env=None, nranks=2):
    env = dict(os.environ, DR_RCCL_LIB=str(fake_lib), **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, str(FAKE / "rank_main.py"), scenario, str(r), str(nranks), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for x in procs:
                x.kill()
            raise
        outs.append(o)
    res = []
    for r in range(nranks):
        f = tmp_path / ("result.%d.json" % r)
        assert f.exists(), outs[r][-3000:]
        res.append(json.loads(f.read_text()))
    for r, d in enumerate(res):
        assert "error" not in d, (d["error"], outs[r][-2000:])
    return res


def test_two_ranks_exchange_equals_the_all_local_merge(fake_lib, tmp_path):
    for d in _run("ok", fake_lib, tmp_path):
        assert len(d["checks"]) == 3 and all(ok for _, ok in d["checks"]), d


def test_a_failing_local_phase_fails_the_call_on_both_ranks(fake_lib, tmp_path):
    from diskrag_amd import _ffi
    res = _run("local_failure", fake_lib, tmp_path)
    assert all(ok for d in res for _, ok in d["checks"]), res
    assert res[0]["code"] == _ffi.E_REMOTE and res[1]["code"] != _ffi.E_REMOTE        # the local error where it happened, DR_E_REMOTE on the peer


# (the FIRST exchange over a communicator waits ten times the limit unless DR_EXCHANGE_FIRST_TIMEOUT_MS says otherwise: start-up skew)
_FAIL_ENV = {"FAKE_RCCL_FAIL_RANK": "1", "FAKE_RCCL_FAIL_AT": "0", "DR_EXCHANGE_TIMEOUT_MS": "3000", "DR_EXCHANGE_FIRST_TIMEOUT_MS": "3000"}


def test_a_failing_exchange_does_not_hang_the_peer(fake_lib, tmp_path):
    from diskrag_amd import _ffi
    res = _run("exchange_failure", fake_lib, tmp_path, _FAIL_ENV)
    assert all(ok for d in res for _, ok in d["checks"]), res
    assert res[0]["code"] == _ffi.E_REMOTE and 2.0 < res[0]["seconds"] < 60.0         # rank 0 waited its limit, then gave up
    assert res[1]["code"] == _ffi.E_NODEVICE                                           # rank 1 reports its own failure at once


def test_the_first_exchange_waits_longer_than_the_later_ones(fake_lib, tmp_path):
    """Start-up skew is not a dead peer: with only DR_EXCHANGE_TIMEOUT_MS set, the first exchange over a communicator gets ten times the limit."""
    from diskrag_amd import _ffi
    env = dict(_FAIL_ENV, DR_EXCHANGE_TIMEOUT_MS="600")
    del env["DR_EXCHANGE_FIRST_TIMEOUT_MS"]
    res = _run("exchange_failure", fake_lib, tmp_path, env)
    assert res[0]["code"] == _ffi.E_REMOTE and 5.5 < res[0]["seconds"] < 60.0


# ---- world = 8: the shape of the node the driver measures on (eight ranks sharing the one GPU of this pool)
def test_eight_ranks_exchange_equals_the_all_local_merge(fake_lib, tmp_path):
    res = _run("ok", fake_lib, tmp_path, nranks=8)
    assert len(res) == 8
    for d in res:
        assert len(d["checks"]) == 3 and all(ok for _, ok in d["checks"]), d


def test_one_failing_local_phase_among_eight_fails_the_call_on_every_rank(fake_lib, tmp_path):
    from diskrag_amd import _ffi
    res = _run("local_failure", fake_lib, tmp_path, nranks=8)
    assert all(ok for d in res for _, ok in d["checks"]), res
    assert [d["code"] == _ffi.E_REMOTE for d in res] == [r != 1 for r in range(8)]    # eight status words read on every rank


def test_one_failing_exchange_among_eight_releases_seven_peers(fake_lib, tmp_path):
    from diskrag_amd import _ffi
    res = _run("exchange_failure", fake_lib, tmp_path, _FAIL_ENV, nranks=8)
    assert all(ok for d in res for _, ok in d["checks"]), res
    for r, d in enumerate(res):
        if r == 1:
            assert d["code"] == _ffi.E_NODEVICE
        else:
            assert d["code"] == _ffi.E_REMOTE and 2.0 < d["seconds"] < 90.0, (r, d)
