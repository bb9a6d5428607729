"""`python bench.py --gpus 2` launches its own ranks (no torchrun): rendezvous, barrier +
max-over-ranks timing and the record all_gather, run here over gloo with the device work left
out (`--launch-selftest`; the line says so).  The GPU path differs only in backend="nccl"."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_two_ranks():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--launch-selftest"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2
    assert rec["value"] is None and "NO device work" in rec["selftest"]
    assert rec["seconds"] >= 0.02  # the slower rank's time (max over ranks)


def test_bench_self_launch_eight_ranks():
    """The shape the driver's scaling run has (one node, eight ranks): rendezvous at 127.0.0.1, barrier + max-over-ranks
    timing and the record all_gather at world size 8 (keep the 8-rank path warm)."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8",
                          "--launch-selftest"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rccl_ranks"] == 8
    assert rec["value"] is None and "NO device work" in rec["selftest"]


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "WORLD_SIZE" in (out.stderr + out.stdout)
