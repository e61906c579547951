"""The host-slot pipeline's bookkeeping (a-nice-rag_amd/csrc/host_slots.hpp: slot = sequence number mod N, busy
flags, condition variable, staging re-size) under ThreadSanitizer with a fake device -- no GPU.  `anrag_hybrid_search`
(api.hip) runs exactly this template with HIP streams and events behind the same Backend contract; the GPU stress tests
(tests/test_gpu_pipeline_stress.py) cover that side.  Sanitizers belong on the CPU build: this is it."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "native", "host_slots_tsan.cpp")
INC = os.path.join(REPO, "a-nice-rag_amd", "csrc")


def _build(tmp_path, name, extra=()):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", INC, *extra, SRC, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def _run(exe, *args):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, env=env, timeout=300)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_slot_ring_is_race_free_under_tsan(tmp_path):
    exe = _build(tmp_path, "host_slots_tsan")
    for threads, per_thread in ((12, 400), (16, 250), (2, 600)):
        r = _run(exe, threads, per_thread)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
        assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
        assert " 0 wrong" in r.stdout and "ring idle at the end: 1" in r.stdout, r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_the_harness_catches_a_broken_ring(tmp_path):
    """The same harness over a deliberately wrong pipeline (slot released before its results are copied out) must be
    caught -- by ThreadSanitizer, by the value checks, or both: the test above is not vacuous."""
    exe = _build(tmp_path, "host_slots_broken", ["-DBREAK_RING"])
    r = _run(exe, 12, 400)
    assert "ThreadSanitizer" in r.stderr or r.returncode != 0, (r.stdout, r.stderr[-500:])
