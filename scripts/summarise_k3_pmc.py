#!/usr/bin/env python3
"""gpurun_out/r03k3/ (scripts/refresh_pmc_k3.sh) -> profiles/r03_pmc_k3_forms.json + the counter rows of bm25_kernel."""
import collections, csv, glob, json, os, re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for form in ("tall", "wide"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in sorted(glob.glob(os.path.join(REPO, "gpurun_out", "r03k3", f"pmc_{form}_*.csv"))):
        rows = list(csv.reader(open(f)))
        keep = [rows[0][8:]] + [[re.sub(r"\(.*\)", "", r[8].replace("void anrag::", ""))] + r[9:] for r in rows[1:] if "bm25_kernel" in r[8]]
        csv.writer(open(os.path.join(REPO, "profiles", "r03_k3_" + os.path.basename(f)), "w")).writerows(keep)
        for r in csv.DictReader(open(f)):
            if "bm25_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    waves = 1960 * (4 if form == "tall" else 16)
    d = {k: v / n for k, (v, n) in agg.items()}
    d["launches_sampled"] = max(n for _, (v, n) in agg.items())
    d["waves_per_launch"] = waves
    for k, c in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD")):
        d[k + "_per_wave"] = d[c] / waves
    for k, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY")):
        d["share_" + k] = d[c] / d["SQ_WAVE_CYCLES"]
    d["kernel_cycles_per_xcd"] = d["GRBM_GUI_ACTIVE"] / 8
    d["valu_pipe_busy"] = d["SQ_INSTS_VALU"] * 4 / 1024 / d["kernel_cycles_per_xcd"]
    out[form] = d
out["what"] = ("bm25_kernel at 8 queries per launch, 1M documents, 9-term queries (scripts/microbench_bm25.py 1000000 512 8), "
               "rocprofv3 --pmc in separate passes (scripts/refresh_pmc_k3.sh); tall = <256 threads x 16 documents> (three "
               "workgroups per CU, 24 slots per gather round), wide = <1024 x 4>; counter values are means per launch; "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles")
json.dump(out, open(os.path.join(REPO, "profiles", "r03_pmc_k3_forms.json"), "w"), indent=1)
for form in ("tall", "wide"):
    print(form, {k: round(v, 3) for k, v in out[form].items() if "share" in k or "per_wave" in k or "busy" in k or k == "kernel_cycles_per_xcd"})
