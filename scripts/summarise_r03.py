#!/usr/bin/env python3
"""gpurun_out/r03/ (scripts/refresh_profiles_r03.sh, run on the GPU box) -> profiles/r03_*: the bench lines as they were
printed, the rocprofv3 kernel-stats CSVs trimmed to this library's kernels, small JSON summaries of the PMC passes
(FETCH_SIZE / WRITE_SIZE per K1 launch with the gfx950 correction; FETCH_SIZE of the select + sort kernel over a 1M-row
score tile) and the side measurements' text lines."""
import csv, json, os, shutil, statistics, subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "r03")
DST = os.path.join(REPO, "profiles")
commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"

for name in ("bench_hybrid_1Mx768", "bench_hybrid_filter_1Mx768", "bench_c5_one_rank_1Mx1024",
             "bench_shard_rehearsal_500000_rows", "bench_shard_rehearsal_250000_rows",
             "bench_shard_rehearsal_125000_rows", "bench_hybrid_under_rocprofv3"):
    line = open(os.path.join(SRC, name + ".json")).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(DST, "r03_" + name + ".json"), "w").write(line + "\n")
for name in ("c2_batch1_lanes", "launch_gap", "k3_terms_forms", "eval_route", "dropin"):
    text = open(os.path.join(SRC, name + ".txt")).read()
    if name == "dropin":
        text = "\n".join(l for l in text.splitlines() if "q/s" in l or l.startswith("built")) + "\n"
    open(os.path.join(DST, "r03_" + name + ".txt"), "w").write(text)


def trim_stats(src, dst):
    rows = list(csv.reader(open(src)))
    keep = [rows[0]] + [r for r in rows[1:] if "anrag::" in r[0]]  # (the rest is torch generating the synthetic corpus)
    csv.writer(open(dst, "w")).writerows(keep)
    return {r[0].split("(")[0].replace("void ", ""): (int(r[1]), float(r[3])) for r in keep[1:]}


stats = {}
for src, dst in (("bench_hybrid_1Mx768_kernel_stats", "r03_bench_hybrid_1Mx768_kernel_stats"),
                 ("k3_kernel_stats_1_per_launch", "r03_k3_kernel_stats_1_per_launch"),
                 ("k3_kernel_stats_8_per_launch", "r03_k3_kernel_stats_8_per_launch"),
                 ("k3_kernel_stats_16_per_launch", "r03_k3_kernel_stats_16_per_launch"),
                 ("rank_9609x384_kernel_stats", "r03_rank_9609x384_kernel_stats"),
                 ("rank_1Mx768_kernel_stats", "r03_rank_1Mx768_kernel_stats")):
    stats[src] = trim_stats(os.path.join(SRC, src + ".csv"), os.path.join(DST, dst + ".csv"))
for name in ("rank_9609x384", "rank_1Mx768", "k3_microbench_1", "k3_microbench_8", "k3_microbench_16"):
    line = open(os.path.join(SRC, name + ".txt")).read().strip().splitlines()[-1]
    open(os.path.join(DST, "r03_" + name + "_under_rocprofv3.txt"), "w").write(line + "\n")


def pmc(path, kernel_substr, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return vals


def keep_ours(src, dst):
    rows = list(csv.reader(open(src)))
    csv.writer(open(dst, "w")).writerows([rows[0]] + [r for r in rows[1:] if "anrag::" in r[8]])


scan = "dense_scan_kernel<64, 3, 2, false, false, 256>"
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    keep_ours(os.path.join(SRC, f"pmc_{c}_dense_1Mx768.csv"), os.path.join(DST, f"r03_pmc_{c}_dense_1Mx768.csv"))
fetch = pmc(os.path.join(SRC, "pmc_FETCH_SIZE_dense_1Mx768.csv"), scan, "FETCH_SIZE")
write = pmc(os.path.join(SRC, "pmc_WRITE_SIZE_dense_1Mx768.csv"), scan, "WRITE_SIZE")
f_kb, w_kb = statistics.median(fetch), statistics.median(write)
rec = {
    "kernel": "dense_scan_kernel<64,3,2,false,false,256>", "rows": 1000000, "dim": 768, "commit": commit,
    "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB_raw": w_kb, "launches": len(fetch),
    "correction": "gfx950: FETCH_SIZE reports exactly 1/2 of a 16 B/lane coalesced streaming read "
                  "(guides/MI355X_MICROARCH.md, HBM): x2; WRITE_SIZE exact",
    "hbm_bytes_per_launch": int(f_kb * 2 * 1024 + w_kb * 1024), "algorithmic_bytes_per_launch": 3072000000,
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 scripts/microbench_dense.py 1000000 768 10 20 "
               "(WRITE_SIZE in its own pass); median over the one-query launches (single queries now ride the scan lanes: "
               "every launch but the first also carries the previous query's list merge as one extra workgroup)",
}
rec["ratio_to_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
json.dump(rec, open(os.path.join(DST, "r03_pmc_dense_scan.json"), "w"), indent=1)

# the select + sort kernel over 1M-row score tiles: bytes fetched per SEGMENT (query) against one read of the segment
keep_ours(os.path.join(SRC, "pmc_FETCH_SIZE_rank_1Mx768.csv"), os.path.join(DST, "r03_pmc_FETCH_SIZE_rank_1Mx768.csv"))
sel = {}
for key, name, seg_bytes in (("f32", "seg_topk_sort_kernel<float, false>", 4e6), ("f64", "seg_topk_sort_kernel<double, false>", 8e6),
                             ("fused", "seg_topk_sort_kernel<double, true>", 12e6)):
    v = pmc(os.path.join(SRC, "pmc_FETCH_SIZE_rank_1Mx768.csv"), name, "FETCH_SIZE")
    if v:
        per_launch = statistics.median(v) * 1024
        sel[key] = {"kernel": name, "launches": len(v), "FETCH_SIZE_bytes_raw_per_launch": per_launch, "segments_per_launch": 256,
                    "segment_bytes": seg_bytes, "raw_fetch_over_one_read_of_the_segments": per_launch / (256 * seg_bytes)}
sel["note"] = ("raw FETCH_SIZE (x2 would apply if these were 16 B/lane streaming reads; the passes here are 4 B and 8 B per lane: "
               "uncorrected); a radix-select pass reads the whole segment, the compaction pass reads it once more")
sel["commit"] = commit
json.dump(sel, open(os.path.join(DST, "r03_pmc_rank_select.json"), "w"), indent=1)

# K1T (dense_tile.hip) on the 1M x 768 corpus: time per launch from the kernel trace, HBM bytes per launch from the FETCH_SIZE
# pass of the same command (x2: 16 B / lane streaming reads, as for K1)
tile_name = "dense_tile_mfma_kernel<3, false>"
tile = {"kernel": tile_name, "rows": 1000000, "dim": 768, "commit": commit}
for k, v in stats["rank_1Mx768_kernel_stats"].items():
    if k.startswith("anrag::" + tile_name) or tile_name in k:
        tile["launches"], tile["avg_launch_ns"] = v
tf = pmc(os.path.join(SRC, "pmc_FETCH_SIZE_rank_1Mx768.csv"), tile_name, "FETCH_SIZE")
if tf and "avg_launch_ns" in tile:
    q_per_launch = 128
    tile["queries_per_launch"] = q_per_launch
    tile["queries_in_lds_at_a_time"] = 32
    tile["FETCH_SIZE_KB_raw_per_launch"] = statistics.median(tf)
    tile["hbm_bytes_per_launch"] = statistics.median(tf) * 2 * 1024
    tile["one_pass_over_the_corpus_bytes"] = 3072000000
    tile["passes_over_the_corpus_per_launch"] = tile["hbm_bytes_per_launch"] / 3072000000
    tile["hbm_GBps"] = tile["hbm_bytes_per_launch"] / tile["avg_launch_ns"]
    tile["hbm_frac_of_8TBps"] = tile["hbm_GBps"] / 8000.0
    tile["flop_per_launch"] = 2.0 * 1000000 * 768 * q_per_launch
    tile["TFLOPs"] = tile["flop_per_launch"] / tile["avg_launch_ns"] / 1e3
    tile["frac_of_157.3_TF_fp32_matrix_peak"] = tile["TFLOPs"] / 157.3
    tile["mfma_pipe_cycles_per_launch_per_simd"] = (1000000 / 16) * (q_per_launch / 16) * 192 * 32 / 1024
    tile["us_per_query"] = tile["avg_launch_ns"] / 1e3 / q_per_launch
json.dump(tile, open(os.path.join(DST, "r03_k1t_tile.json"), "w"), indent=1)

print(json.dumps(rec, indent=1))
print(json.dumps(sel, indent=1))
print(json.dumps(tile, indent=1))
for src, st in stats.items():
    print(src)
    for k, v in st.items():
        print(f"  {v[0]:6d} x {v[1] / 1e3:9.1f} us  {k[:100]}")
