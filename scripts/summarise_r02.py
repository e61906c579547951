#!/usr/bin/env python3
"""gpurun_out/r02/ (scripts/refresh_profiles_r02.sh, run on the GPU box) -> profiles/r02_*: the bench lines as they
were printed, the rocprofv3 kernel-stats CSVs trimmed to this library's kernels, and small JSON summaries of the PMC
passes (FETCH_SIZE / WRITE_SIZE per K1 launch with the gfx950 correction; MFMA pipe counters of K2)."""
import csv, json, os, shutil, statistics, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "r02")
DST = os.path.join(REPO, "profiles")
import subprocess
commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
# (the GPU box has no .git: run this right after the refresh, on the commit that was sent)

for name in ("bench_hybrid_1Mx768", "bench_hybrid_filter_1Mx768", "bench_c5_one_rank_1Mx1024",
             "bench_shard_rehearsal_500000_rows", "bench_shard_rehearsal_250000_rows",
             "bench_shard_rehearsal_125000_rows", "bench_hybrid_under_rocprofv3"):
    line = open(os.path.join(SRC, name + ".json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(DST, "r02_" + name + ".json"), "w").write(line + "\n")


def trim_stats(src, dst):
    rows = list(csv.reader(open(src)))
    keep = [rows[0]] + [r for r in rows[1:] if "anrag::" in r[0]]  # (the rest is torch generating the synthetic corpus)
    csv.writer(open(dst, "w")).writerows(keep)
    return {r[0].split("(")[0].replace("void ", ""): (int(r[1]), float(r[3])) for r in keep[1:]}


bench_stats = trim_stats(os.path.join(SRC, "bench_hybrid_1Mx768_kernel_stats.csv"),
                         os.path.join(DST, "r02_bench_hybrid_1Mx768_kernel_stats.csv"))
k3_stats = trim_stats(os.path.join(SRC, "k3_kernel_stats.csv"), os.path.join(DST, "r02_k3_kernel_stats.csv"))


def pmc(path, kernel_substr, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return vals


scan = "dense_scan_kernel<64, 3, 2, false, false, 256>"
fetch = pmc(os.path.join(SRC, "pmc_FETCH_SIZE_dense_1Mx768.csv"), scan, "FETCH_SIZE")
write = pmc(os.path.join(SRC, "pmc_WRITE_SIZE_dense_1Mx768.csv"), scan, "WRITE_SIZE")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy(os.path.join(SRC, f"pmc_{c}_dense_1Mx768.csv"), os.path.join(DST, f"r02_pmc_{c}_dense_1Mx768.csv"))
    rows = list(csv.reader(open(os.path.join(DST, f"r02_pmc_{c}_dense_1Mx768.csv"))))
    csv.writer(open(os.path.join(DST, f"r02_pmc_{c}_dense_1Mx768.csv"), "w")).writerows(
        [rows[0]] + [r for r in rows[1:] if "anrag::" in r[8]])
f_kb = statistics.median(v for v, _ in fetch)
w_kb = statistics.median(v for v, _ in write)
rec = {
    "kernel": "dense_scan_kernel<64,3,2,false,false,256>", "rows": 1000000, "dim": 768, "commit": commit,
    "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB_raw": w_kb, "launches": len(fetch),
    "correction": "gfx950: FETCH_SIZE reports exactly 1/2 of a 16 B/lane coalesced streaming read "
                  "(guides/MI355X_MICROARCH.md, HBM): x2; WRITE_SIZE exact",
    "hbm_bytes_per_launch": int(f_kb * 2 * 1024 + w_kb * 1024), "algorithmic_bytes_per_launch": 3072000000,
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 scripts/microbench_dense.py 1000000 768 10 20 "
               "(WRITE_SIZE in its own pass); median over the one-query launches (the script's first call carries 4 "
               "queries in one launch)",
}
rec["ratio_to_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
json.dump(rec, open(os.path.join(DST, "r02_pmc_dense_scan.json"), "w"), indent=1)

# K3: algorithmic bytes from the microbench's own line
line = open(os.path.join(SRC, "k3_microbench.txt")).read().strip().splitlines()[-1]
name = [k for k in k3_stats if "bm25_kernel" in k][0]
calls, avg_ns = k3_stats[name]
mb = float(line.split("algorithmic ")[1].split(" MB")[0])
k3 = {"kernel": name, "commit": commit, "docs": 1000000, "calls": calls, "avg_us_rocprofv3_kernel_trace": avg_ns / 1e3,
      "algorithmic_MB_per_query": mb, "what": "sum over the query's 9 terms of df(t) x 12 B (int32 doc id + fp64 impact)",
      "achieved_GBps": mb * 1e6 / (avg_ns * 1e-9) / 1e9, "frac_of_8TBps": mb * 1e6 / (avg_ns * 1e-9) / 8e12,
      "under_the_scan_avg_us": [v[1] / 1e3 for k, v in bench_stats.items() if "bm25_kernel" in k][0],
      "microbench_line_under_the_profiler": line,
      "command": "rocprofv3 --kernel-trace --stats -- python3 scripts/microbench_bm25.py 1000000 400 (idle GPU, one launch per query)"}
json.dump(k3, open(os.path.join(DST, "r02_k3_bm25_1M.json"), "w"), indent=1)

# (K2: scripts/summarise_k2.py)
print(json.dumps(rec, indent=1)); print(json.dumps(k3, indent=1))
for k, v in bench_stats.items():
    print(f"{v[0]:6d} x {v[1] / 1e3:9.1f} us  {k[:90]}")
