#!/bin/bash
# HBM traffic (FETCH_SIZE, own --pmc pass, kernel trace only beside it) of K3 and of the two K2 full-pass kernels
#   gpurun -- bash scripts/refresh_pmc_traffic.sh ; then python scripts/summarise_traffic.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02traffic
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k3 -o k3 -- python3 $R/scripts/microbench_bm25.py 1000000 200 > $O/k3.txt 2> $O/k3.err || exit 1
find $O/k3 -name "*counter_collection.csv" -exec cp {} $O/pmc_FETCH_SIZE_k3.csv \;
for p in f32 bf16x3; do
  ITERS=20 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/k2_$p -o k2 -- python3 $R/scripts/microbench_batched.py 1000000 768 256 10 $p > $O/k2_$p.txt 2> $O/k2_$p.err || exit 1
  find $O/k2_$p -name "*counter_collection.csv" -exec cp {} $O/pmc_FETCH_SIZE_k2_$p.csv \;
done
rm -rf $O/k3 $O/k2_f32 $O/k2_bf16x3
ls -la $O
