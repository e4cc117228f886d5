"""bench.py's own N-process launcher and file barriers, on a CPU box: the stub engine of bench.py
(DR_BENCH_STUB=1) stands in for the GPU so that only the rank / barrier / JSON plumbing runs. No torch anywhere."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DR_BENCH_RANK")}
    env.update(DR_BENCH_STUB="1", **kw)
    return env


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batches-per-step", "4"], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert len(d["config"]["per_rank_seconds"]) == 2
    # whole-job value: both ranks' queries over the slowest rank's time
    assert abs(d["value"] - 2 * 10000 * 3 * 4 / max(d["config"]["per_rank_seconds"])) < 1e-6 * d["value"]
    assert "torch" not in p.stderr


def test_workers_also_run_under_a_torchrun_style_environment(tmp_path):
    procs = []
    for r in range(2):
        env = _env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29771",
                   TORCHELASTIC_RUN_ID="t%d" % os.getpid(), TMPDIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batches-per-step", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    d = json.loads(outs[0][0].strip())
    assert d["n_gpus"] == 2
    assert outs[1][0].strip() == ""          # only rank 0 prints


def test_bench_imports_no_torch_and_gpus_flag_is_used():
    src = (ROOT / "bench.py").read_text()
    assert "import torch" not in src
    assert "args.gpus" in src


def _torchrun_pair(tmp_path, run_id, extra=()):
    procs = []
    for r in range(2):
        env = _env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29500",
                   TORCHELASTIC_RUN_ID=run_id, TMPDIR=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batches-per-step", "3", *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1] for o in outs]
    return json.loads(outs[0][0].strip())


def test_two_runs_with_the_same_run_id_never_share_files(tmp_path):
    """plain torchrun gives every job port 29500 and run id 'none'. A second job (and one started over the
    leftovers of a crashed job) must not read the first one's barrier / time files."""
    base = tmp_path / "diskrag_bench_29500_none"
    base.mkdir()
    # leftovers of a crashed earlier job: a join / assign pair with a foreign nonce, old barrier and time files
    (base / "join.1").write_text("deadbeef")
    (base / "assign.1").write_text("deadbeef\n" + str(base / "job_stale"))
    (base / "job_stale").mkdir()
    for name in ("b1.0", "b1.1", "b2.0", "b2.1", "here.1"):
        (base / "job_stale" / name).write_text("x")
    (base / "job_stale" / "t.1").write_text("123.0")
    d1 = _torchrun_pair(tmp_path, "none")
    d2 = _torchrun_pair(tmp_path, "none")
    for d in (d1, d2):
        assert d["n_gpus"] == 2 and len(d["config"]["per_rank_seconds"]) == 2
        assert max(d["config"]["per_rank_seconds"]) < 5.0            # nobody read the stale 123 s
    # each job removed its own scratch directory; only the crashed job's leftovers remain
    assert sorted(p.name for p in base.iterdir() if p.name.startswith("job_")) == ["job_stale"]


def test_strong_scaling_slices_tile_the_batch(tmp_path):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batches-per-step", "3",
                        "--scaling", "strong", "--num-queries", "10001"], env=_env(), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2
    sl = sorted(d["config"]["per_rank_slice"])
    assert sl[0][0] == 0 and sl[-1][1] == 10001 and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))      # exactly once
    # whole-job value: ONE batch stream, not one per rank
    assert abs(d["value"] - 10001 * 2 * 3 / max(d["config"]["per_rank_seconds"])) < 1e-6 * d["value"]


def test_a_failing_rank_ends_the_job_at_once():
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=_env(DR_BENCH_STUB_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert time.time() - t0 < 60          # not the 1800 s barrier timeout


def test_bench_flags_of_round_4_parse():
    """--headline-only (the rocprofv3 run of scripts/profile_run.sh) and --c5-group (submits per exchange of the sharded path) exist"""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--headline-only", "--c5-group", "--scaling", "--config"):
        assert flag in r.stdout, flag


# ---- world = 8: the size of the node the driver measures on
def test_gpus_8_weak_and_strong_at_the_drivers_size():
    """eight ranks through the launcher: barriers, the gathered times, 1250-query slices that tile a 10 000-query batch, ONE JSON line"""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--batches-per-step", "4"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and len(d["config"]["per_rank_seconds"]) == 8
    assert abs(d["value"] - 8 * 10000 * 3 * 4 / max(d["config"]["per_rank_seconds"])) < 1e-6 * d["value"]
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batches-per-step", "3", "--scaling", "strong"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    sl = sorted(d["config"]["per_rank_slice"])
    assert d["scaling"] == "strong" and len(sl) == 8 and all(b - a == 1250 for a, b in sl) and sl[0][0] == 0 and sl[-1][1] == 10000
    assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
    assert abs(d["value"] - 10000 * 2 * 3 / max(d["config"]["per_rank_seconds"])) < 1e-6 * d["value"]


def test_one_failing_rank_among_eight_ends_the_job_at_once():
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       env=_env(DR_BENCH_STUB_FAIL_RANK="5"), capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and time.time() - t0 < 120


def _fake_sysfs(root, sockets):
    """eight GPUs' PCI functions under root/bus/pci/devices, `sockets[i]` = (cpulist, numa node) of device i"""
    bdfs = []
    for i, (cpulist, node) in enumerate(sockets):
        bdf = "0000:%02x:00.0" % (0x05 + 0x10 * i)
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpulist + "\n")
        (d / "numa_node").write_text("%d\n" % node)
        bdfs.append(bdf)
    return bdfs


def test_pin_to_gpu_socket_on_a_fake_tree_of_eight_devices_on_two_sockets(tmp_path, monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        import pytest
        pytest.skip("one CPU: nothing to split")
    half = len(have) // 2
    s0, s1 = have[:half], have[half:]

    def cpulist(cs):      # ranges and single CPUs, as the kernel prints them
        return ",".join("%d-%d" % (a, a) if i % 2 else str(a) for i, a in enumerate(cs))
    bdfs = _fake_sysfs(tmp_path, [(cpulist(s0), 0)] * 4 + [(cpulist(s1), 1)] * 4)

    class RK:
        world = 8
    got = []
    try:
        for dev in range(8):
            os.sched_setaffinity(0, have)
            r = bench.pin_to_gpu_socket(RK, dev, sysroot=str(tmp_path), bus_id=lambda d: bdfs[d])
            got.append((r, os.sched_getaffinity(0)))
        for dev, (r, aff) in enumerate(got):
            want = set(s0 if dev < 4 else s1)
            assert r == {"pci": bdfs[dev], "numa_node": 0 if dev < 4 else 1, "cpus": len(want)} and aff == want
        assert got[0][1].isdisjoint(got[7][1])                                    # the two sockets' ranks never share a CPU
        # graceful no-ops: the tree hides the device, the runtime has no bus id, one rank, the opt-out
        os.sched_setaffinity(0, have)
        assert bench.pin_to_gpu_socket(RK, 0, sysroot=str(tmp_path / "nothing"), bus_id=lambda d: bdfs[d]) is None
        assert bench.pin_to_gpu_socket(RK, 0, sysroot=str(tmp_path), bus_id=lambda d: None) is None
        RK.world = 1
        assert bench.pin_to_gpu_socket(RK, 0, sysroot=str(tmp_path), bus_id=lambda d: bdfs[d]) is None
        RK.world = 8
        monkeypatch.setenv("DR_BENCH_NO_PIN", "1")
        assert bench.pin_to_gpu_socket(RK, 0, sysroot=str(tmp_path), bus_id=lambda d: bdfs[d]) is None
        monkeypatch.delenv("DR_BENCH_NO_PIN")
        # a list of CPUs this process may not use (a cgroup narrower than the socket): nothing is changed
        _fake_sysfs(tmp_path / "other", [("100000-100003", 0)])
        assert bench.pin_to_gpu_socket(RK, 0, sysroot=str(tmp_path / "other"), bus_id=lambda d: "0000:05:00.0") is None
        assert os.sched_getaffinity(0) == set(have)
    finally:
        os.sched_setaffinity(0, have)


def test_c3_c4_default_to_the_recall_095_point_of_the_size_they_build():
    """`--config c3 | c4` run at the fastest measured recall >= 0.95 point of an index of THAT size (bench scale / --full-size)"""
    sys.path.insert(0, str(ROOT))
    import bench
    a = bench.parse_args(["--config", "c3"])
    assert (a.n, a.dim, a.L, a.bw, a.rerank_top) == (1_000_000, 1536, 100, 0, 72)
    a = bench.parse_args(["--config", "c3", "--full-size"])
    assert (a.n, a.L, a.bw, a.rerank_top) == (10_000_000, 272, 128, 0)
    a = bench.parse_args(["--config", "c4"])
    assert (a.n, a.dim, a.m, a.L, a.bw, a.rerank_top) == (4_000_000, 96, 16, 200, 8, 0)
    a = bench.parse_args(["--config", "c4", "--full-size"])
    assert (a.n, a.L, a.bw) == (100_000_000, 400, 32)
    a = bench.parse_args(["--config", "c3", "--L", "300"])          # an explicit list length: no rerank cut unless asked for
    assert (a.L, a.bw, a.rerank_top) == (300, 0, 0)
    a = bench.parse_args(["--config", "c5", "--full-size"])
    assert a.n == 125_000_000 and a.R == 128
    a = bench.parse_args([])
    assert (a.n, a.dim, a.R, a.L, a.bw, a.m) == (1_000_000, 128, 64, 100, 8, 32)
