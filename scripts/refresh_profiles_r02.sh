#!/bin/bash
# Round 2: regenerates the measurements kept under profiles/r02_* (run on the GPU box through gpurun; outputs land in
# gpurun_out/r02/ and scripts/summarise_r02.py turns them into the committed summaries).
#   gpurun --timeout 1100 -- 'bash scripts/refresh_profiles_r02.sh'
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd $R
git rev-parse --short HEAD > $O/commit.txt 2>/dev/null || echo unknown > $O/commit.txt
python bench.py > $O/bench_hybrid_1Mx768.json 2> $O/bench_hybrid.err
echo "bench done"
python bench.py --filter --no-cpu-baseline > $O/bench_hybrid_filter_1Mx768.json 2>/dev/null
python bench.py --rows 1000000 --dim 1024 --no-cpu-baseline --no-also > $O/bench_c5_one_rank_1Mx1024.json 2>/dev/null
for rows in 500000 250000 125000; do
  ANRAG_FORCE_SHARDED=1 python bench.py --rows $rows --steps 2000 --warmup 200 > $O/bench_shard_rehearsal_${rows}_rows.json 2>/dev/null
done
echo "rehearsals done"
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (no counters in this run)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o hybrid -- python3 $R/bench.py --no-cpu-baseline --no-also \
  > $O/bench_hybrid_under_rocprofv3.json 2> $O/rocprof_bench.err
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_hybrid_1Mx768_kernel_stats.csv \;
echo "trace done"
# 2. K3 alone: kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k3 -o k3 -- python3 $R/scripts/microbench_bm25.py 1000000 400 \
  > $O/k3_microbench.txt 2> $O/rocprof_k3.err
find $O/prof_k3 -name "*kernel_stats.csv" -exec cp {} $O/k3_kernel_stats.csv \;
# 3. HBM traffic of K1: separate --pmc passes (kernel trace only beside them)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o dense -- python3 $R/scripts/microbench_dense.py 1000000 768 10 20 \
    > $O/pmc_$c.txt 2> $O/pmc_$c.err
  find $O/pmc_$c -name "*counter_collection.csv" -exec cp {} $O/pmc_${c}_dense_1Mx768.csv \;
done
echo "pmc K1 done"
# 4. K2 has its own scripts: scripts/refresh_profiles_k2.sh + scripts/summarise_k2.py
# the raw traces are big: keep the summaries only
rm -rf $O/prof_bench $O/prof_k3 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
