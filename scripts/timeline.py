#!/usr/bin/env python3
"""Print the last N anrag kernel dispatches of a rocprofv3 kernel_trace.csv as a timeline."""
import csv, glob, sys
path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(path + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "anrag" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-n:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f} us q={r['Queue_Id']} {r['Kernel_Name'][12:44]}")
