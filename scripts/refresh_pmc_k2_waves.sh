#!/bin/bash
# where the waves of the two K2 full-pass kernels spend their cycles (SQ counters, two --pmc passes per mode)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02k2waves
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in f32 bf16x3; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    ITERS=20 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${p}_$i -o k2 -- python3 $R/scripts/microbench_batched.py 1000000 768 256 10 $p > $O/${p}_$i.txt 2> $O/${p}_$i.err || exit 1
    find $O/${p}_$i -name "*counter_collection.csv" -exec cp {} $O/pmc_${p}_$i.csv \;
    rm -rf $O/${p}_$i
  done
done
ls $O
