#!/usr/bin/env python3
"""gpurun_out/r02k2/ (scripts/refresh_profiles_k2.sh) -> profiles/r02_k2_kernel_stats_{mode}.csv, profiles/r02_pmc_k2_{mode}.json"""
import csv, json, os, statistics, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r02k2"), os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
FULL = {"f32": "dense_batched_kernel<256, false, false>", "bf16x3": "dense_batched_split_dma_kernel<false, 2>"}
FLOP = 2.0 * 256 * 1_000_000 * 768

for p in ("f32", "bf16x3"):
    rows = [r for r in csv.DictReader(open(os.path.join(SRC, f"k2_kernel_stats_{p}.csv"))) if "anrag" in r["Name"]]
    with open(os.path.join(DST, f"r02_k2_kernel_stats_{p}.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    per_pass = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e3 * int(r["Calls"]) / 101 for r in rows}
    pm = {}
    for r in csv.DictReader(open(os.path.join(SRC, f"pmc_k2_{p}.csv"))):
        if FULL[p] in r["Kernel_Name"]:
            pm.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    out = {"kernel": FULL[p] + " (K2 full pass, 256 queries x 1M x 768)", "commit": commit,
           "steady_state_us_per_pass_by_kernel_rocprofv3_kernel_trace": per_pass,
           "full_pass_TFLOPs_unprofiled": (3 if p == "bf16x3" else 1) * FLOP / ([float(r["AverageNs"]) for r in rows if FULL[p] in r["Name"]][0] * 1e-9) / 1e12,
           "command": "ITERS=100 rocprofv3 --kernel-trace --stats -- python3 scripts/microbench_batched.py 1000000 768 256 10 "
                      f"{p}; ITERS=60 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- (same)"}
    for c, v in pm.items():
        out[c] = statistics.median(x for x, _ in v)
        out["kernel_us_under_pmc"] = statistics.median(t for _, t in v) / 1e3
        out["dispatches"] = len(v)
    clk = out["GRBM_GUI_ACTIVE"] / 8 / (out["kernel_us_under_pmc"] * 1e-6) / 1e9
    out["derived"] = {"effective_clock_GHz": clk,
                      "mfma_pipe_utilisation": out["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * out["GRBM_GUI_ACTIVE"] / 8),
                      "note": "busy cycles summed over the 1,024 SIMDs / (SIMDs x kernel cycles); GRBM_GUI_ACTIVE is the sum over the 8 XCDs"}
    out["microbench_line_unprofiled_run"] = open(os.path.join(SRC, f"trace_{p}.txt")).read().strip().splitlines()[-1]
    json.dump(out, open(os.path.join(DST, f"r02_pmc_k2_{p}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
