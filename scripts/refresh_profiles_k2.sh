#!/bin/bash
# K2 profiles (run on the GPU box: gpurun -- bash scripts/refresh_profiles_k2.sh), then locally: python scripts/summarise_k2.py
# 1. rocprofv3 kernel trace of 100 steady-state passes per arithmetic mode  2. MFMA pipe counters (own pass, --pmc only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02k2
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in f32 bf16x3; do
  ITERS=100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$p -o k2 -- \
    python3 $R/scripts/microbench_batched.py 1000000 768 256 10 $p > $O/trace_$p.txt 2> $O/trace_$p.err || exit 1
  find $O/trace_$p -name "*kernel_stats.csv" -exec cp {} $O/k2_kernel_stats_$p.csv \;
  ITERS=60 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$p -o k2 -- \
    python3 $R/scripts/microbench_batched.py 1000000 768 256 10 $p > $O/pmc_k2_$p.txt 2> $O/pmc_k2_$p.err || exit 1
  find $O/pmc_$p -name "*counter_collection.csv" -exec cp {} $O/pmc_k2_$p.csv \;
  rm -rf $O/trace_$p $O/pmc_$p
done
ls -la $O
