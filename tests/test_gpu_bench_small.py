"""bench.py itself on a GPU box, at toy size: the N = 1 line and the N = 2 job (two ranks sharing the one GPU) with the real
engine -- the weak-scaling line must carry the strong-scaling figure (`config.strong_scaling`), the explicit
`--scaling strong` mode must tile the batches, and the c5 pipeline must run on one rank. (Round 4: the strong-scaling block once
clobbered a resident batch of the weak-scaling measurement; nothing but a real run sees that.)"""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
TOY = ["--num-vectors", "30000", "--num-queries", "600", "--steps", "2", "--warmup", "1", "--batches-per-step", "3", "--distinct-batches", "4",
       "--no-cpu", "--min-recall", "0.5", "--blocking-calls", "3", "--pq-scan-codes", "1000000"]


def _json_line(stdout):
    """the bench's one JSON line (librccl prints its banner to the C stdout, which is flushed after Python's)"""
    return json.loads([l for l in stdout.strip().splitlines() if l.startswith("{")][-1])


def _run(extra):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + TOY + extra, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return _json_line(r.stdout)


def test_one_rank_line_has_roofline_and_small_batches():
    d = _run([])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and d["config"]["recall_at_10"] >= 0.5
    sb = d["config"]["small_batches_on_one_gpu"]
    assert len(sb) == 3 and all(v["qps"] > 0 and v["submits_per_launch"] >= 1 for v in sb.values())


def test_two_ranks_on_one_gpu_weak_line_carries_the_strong_figure():
    d = _run(["--gpus", "2", "--no-secondary"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["config"]["per_rank_seconds"]) == 2
    s = d["config"]["strong_scaling"]
    assert s["value"] > 0 and s["queries_per_batch_per_gpu"] == 300 and len(s["per_rank_seconds"]) == 2 and s["recall_at_10"] >= 0.5


def test_two_ranks_strong_mode_tiles_every_batch():
    d = _run(["--gpus", "2", "--scaling", "strong", "--no-secondary"])
    assert d["scaling"] == "strong" and d["config"]["per_rank_slice"] == [[0, 300], [300, 600]] and d["config"]["recall_at_10"] >= 0.5


def test_c5_pipeline_on_one_rank():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "c5", "--num-vectors", "65536", "--num-queries", "500", "--steps", "3", "--warmup", "1",
                        "--min-recall", "0.5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["value"] > 0 and d["config"]["rccl_ranks"] == 1 and d["config"]["recall_at_10_vs_bruteforce_adc"] >= 0.5


@pytest.mark.parametrize("cfg", ["c3", "c4"])
def test_c3_c4_configs_at_toy_size(cfg):
    """--config c3 (d = 1536) and --config c4 (d = 96; the reference-faithful M1 beside it): PQ traversal + exact rerank, roofline + cpu_baseline blocks,
    the CPU restatement bit-identical on the sample (the bench raises otherwise)."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", cfg, "--num-vectors", "20000", "--num-queries", "300", "--steps", "2", "--warmup", "1",
                        "--min-recall", "0.5", "--cpu-sample", "100"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["config"]["recall_at_10"] >= 0.5
    assert "bit-identical on the sample: True" in d["cpu_baseline"]["sample"]
    if cfg == "c4":
        assert d["config"]["m1_reference_faithful"]["qps"] > 0
    one = _run(["--rows", "f32", "--no-secondary"]) if cfg == "c3" else None       # (the float-row headline of c2 rides along once)
    if one:
        assert one["config"]["rows"] == "f32" and one["config"]["launch"]["variant"] == 9 and one["roofline"]["row_bytes"] == 512


# ---- world = 8 on one GPU: the job the driver launches on an 8-GPU node, with eight ranks sharing this pool's one device
def test_eight_ranks_on_one_gpu_weak_and_strong():
    d = _run(["--gpus", "8", "--no-secondary", "--num-queries", "800"])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and len(d["config"]["per_rank_seconds"]) == 8 and d["config"]["recall_at_10"] >= 0.5
    assert abs(d["value"] - 8 * 800 * 2 * 3 / max(d["config"]["per_rank_seconds"])) < 1e-6 * d["value"]
    s = d["config"]["strong_scaling"]
    assert s["queries_per_batch_per_gpu"] == 100 and len(s["per_rank_seconds"]) == 8
    d = _run(["--gpus", "8", "--scaling", "strong", "--no-secondary", "--num-queries", "800"])
    assert d["scaling"] == "strong" and d["config"]["per_rank_slice"] == [[100 * r, 100 * (r + 1)] for r in range(8)] and d["config"]["recall_at_10"] >= 0.5
    # per-rank host memory and set-up time: every rank builds the graph and the ground truth itself (DESIGN.md section 7 states them at full size)
    assert len(d["config"]["per_rank_setup"]) == 8 and all(x["max_rss_mb"] > 0 and x["setup_seconds"] > 0 for x in d["config"]["per_rank_setup"])


def test_c5_pipeline_with_eight_ranks_through_the_stand_in_exchange():
    """bench.py --config c5 --gpus 8: eight real processes, eight shards, grouped exchanges with eight status words -- csrc/comm.inc at nranks = 8
    through tests/fake_rccl (RCCL refuses several ranks on one device)."""
    import os
    fake = ROOT / "tests" / "fake_rccl"
    so = fake / "libfake_rccl.so"
    if not so.exists() or so.stat().st_mtime < (fake / "fake_rccl.cpp").stat().st_mtime:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(so), str(fake / "fake_rccl.cpp"), "-lrt"])
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "c5", "--gpus", "8", "--num-vectors", "16384", "--num-queries", "400", "--steps", "6", "--warmup", "2",
                        "--min-recall", "0.5"], capture_output=True, text=True, timeout=1200, env=dict(os.environ, DR_RCCL_LIB=str(so)))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["rccl_ranks"] == 8 and len(d["config"]["per_rank_seconds"]) == 8
    assert d["value"] > 0 and d["config"]["recall_at_10_vs_bruteforce_adc"] >= 0.5 and d["config"]["submits_per_exchange"] == 3
