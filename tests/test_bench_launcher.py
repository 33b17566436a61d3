"""`python bench.py --gpus N` must start N ranks itself when no launcher did (verdict r02: a plain --gpus 8 silently
benchmarked one GPU).  CPU part: the launcher / rendezvous / reporting path with --dry-run over gloo, world 2.  GPU part: the
real step through the same launcher, two gloo ranks sharing the one GPU of the test box (RCCL refuses two ranks per device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout            # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_n_without_launcher_starts_n_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run"],
                       env=_env(FACENET_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["dist_world_size"] == 2 and out["steps"] == 3 and out["dry_run"] is True


def test_world_size_mismatch_is_an_error_also_for_world_1():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run"], env=_env(WORLD_SIZE="2", RANK="0"), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_child_failure_is_the_parents_exit_code():
    # backend nccl (= RCCL) cannot initialise without GPUs: the ranks fail, the launcher reports it, bench.py exits non-zero
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("needs a box without GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(FACENET_DIST_BACKEND="nccl"), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_real_step_through_the_launcher_two_gloo_ranks_on_one_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       env=_env(FACENET_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["dist_world_size"] == 2 and out["config"]["global_batch"] == 180 and out["value"] > 0
    xp = out["gradient_exchange"]
    assert xp["n_buckets"] == len(xp["buckets"]) >= 3 and xp["allreduce_ms"] > 0 and 0.0 <= xp["overlapped_frac"] <= 1.0
    assert abs(sum(b["elements"] for b in xp["buckets"]) * 4e-6 - sum(b["mbytes"] for b in xp["buckets"])) < 0.1
