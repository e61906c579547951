#!/usr/bin/env python3
"""gpurun_out/r02traffic (scripts/refresh_pmc_traffic.sh) -> profiles/r02_pmc_traffic_k2_k3.json: HBM bytes fetched per
launch (FETCH_SIZE is in KB and, on gfx950, counts half of a 16 B/lane streaming read: x2, as the guide prescribes and as
K1's own collection confirms) against the algorithmic bytes."""
import csv, json, os, statistics, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r02traffic"), os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
def fetch(path, kernel):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    return statistics.median(v) * 1024 * 2, len(v)
out = {"commit": commit, "unit": "bytes per launch", "correction": "FETCH_SIZE (KB) x 1024 x 2 (gfx950, guides/MI355X_MICROARCH.md)"}
b, n = fetch(os.path.join(SRC, "pmc_FETCH_SIZE_k3.csv"), "bm25_kernel")
line = open(os.path.join(SRC, "k3.txt")).read().strip().splitlines()[-1]
alg = float(line.split("algorithmic ")[1].split(" MB")[0]) * 1e6
out["K3 bm25_kernel<false,false,1024> (1M docs, 9 terms)"] = {"hbm_bytes": b, "algorithmic_bytes_mean": alg, "ratio": b / alg, "launches": n,
    "note": "algorithmic = sum df(t) x 12 B; the kernel also reads indptr / partition pointers / idf per term and workgroup"}
for p, kern, alg in (("f32", "dense_batched_kernel<256, false, false>", 1_000_000 * 768 * 4.0),
                     ("bf16x3", "dense_batched_split_dma_kernel<false, 2>", 3907 * 256 * 768 * 4.0)):
    b, n = fetch(os.path.join(SRC, f"pmc_FETCH_SIZE_k2_{p}.csv"), kern)
    out[f"K2 {kern} (256 x 1M x 768)"] = {"hbm_bytes": b, "algorithmic_bytes": alg, "ratio": b / alg, "launches": n,
        "note": "algorithmic = the corpus (images) once; the 768 KB query block is re-read per tile from L2"}
json.dump(out, open(os.path.join(DST, "r02_pmc_traffic_k2_k3.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
