"""`bench.py --gpus N` starts its own N ranks (it used to run ONE rank and print n_gpus: 1).
Covered here on the CPU: the launcher + rendezvous with 2 gloo ranks, the refusal to shrink the job when fewer
devices are visible, and the refusal of a --gpus / WORLD_SIZE mismatch.  No kernels run (no CPU path exists)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_launcher_starts_two_ranks_and_relays_rank0():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-selftest", "--rows", "40000", "--dim", "8"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["gpus_arg"] == 2  # the world size the ranks SAW, not the flag
    assert rec["union_matches_single"] is True


def test_launcher_refuses_to_shrink_the_job():
    import torch

    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 2
    assert "refusing to run a smaller job" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2
    assert "WORLD_SIZE=2" in p.stderr
