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
