#!/usr/bin/env python3
"""Print the headline fields of a bench.py JSON line read from stdin (helper for sweeps)."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(tag, d["config"]["rows"], "rows:", round(d["value"]), "q/s", round(d["ms_per_step"] * 1e3, 1), "us/step; main kernel",
      round(r["avg_launch_ms"] * 1e3, 1), "us", round(r["achieved"], 1), r["unit"], f"({r['frac']*100:.1f}%)")
