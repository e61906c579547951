"""GPU: bench.py's N > 1 route end to end on one GPU (ANRAG_FORCE_SHARDED=1: RCCL at world size 1) at a small size:
the sharded pipeline, the block-seeded corpus, and rank 0's check of the sharded answers against a single index
(`sharded_matches_single`) -- the line the driver's multi-GPU runs will be read by."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--rows", "70000", "--dim", "256", "--vocab", "5000",
                        "--queries", "16", "--steps", "64", "--warmup", "8", "--no-also", "--cpu-queries", "3"] + extra,
                       env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_sharded_route_matches_single_index():
    p, rec = _run([], {"ANRAG_FORCE_SHARDED": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert rec["n_gpus"] == 1 and rec["sharded_matches_single"] is True
    assert rec["sharded_check"]["queries"] == 8 and rec["sharded_check"]["max_abs_fused_score_diff"] == 0.0
    assert rec["scaling"] == "strong" and rec["config"]["sharding"].startswith("rows/1")
    assert rec["roofline"]["launches"] >= 8 and 0.0 < rec["roofline"]["frac"] < 1.0


def test_single_route_matches_cpu_port_and_weak_scaling_label():
    p, rec = _run(["--rows-per-gpu", "70000"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert rec["scaling"] == "weak" and rec["config"]["rows"] == 70000
    assert rec["cpu_baseline"]["gpu_results_match_cpu"] is True and rec["cpu_baseline"]["identical_id_lists"] == 3
    assert rec["recall_at_10"]["equal"] is True
    assert rec["roofline"]["traffic"] is None  # no PMC collection for this shape: nothing is pasted in
