#!/usr/bin/env python3
"""Repeat the pipeline / concurrency stress tests in ONE process (races show up rarely).
usage: python scripts/stress_loop.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_pipeline_stress as t

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.time()
for i in range(rounds):
    t.test_concurrent_host_callers_match_serial()
    t.test_mixed_bursts_match_serial()
    print("round", i, "ok", round(time.time() - t0, 1), flush=True)
