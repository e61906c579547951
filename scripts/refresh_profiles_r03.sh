#!/bin/bash
# Round 3: regenerates the measurements kept under profiles/r03_* (run on the GPU box through gpurun; outputs land in
# gpurun_out/r03/ and scripts/summarise_r03.py turns them into the committed summaries).
#   gpurun --timeout 1200 -- 'PART=a bash scripts/refresh_profiles_r03.sh'   then   PART=b (the profiler runs)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
PART=${PART:-all}
if [ "$PART" = "all" ] || [ "$PART" = "a" ]; then
python bench.py > $O/bench_hybrid_1Mx768.json 2> $O/bench_hybrid.err; echo "bench done"
python bench.py --filter --no-cpu-baseline --no-also > $O/bench_hybrid_filter_1Mx768.json 2>/dev/null
python bench.py --rows 1000000 --dim 1024 --no-cpu-baseline --no-also > $O/bench_c5_one_rank_1Mx1024.json 2>/dev/null
for rows in 500000 250000 125000; do
  ANRAG_FORCE_SHARDED=1 python bench.py --rows $rows --steps 2000 --warmup 200 --no-cpu-baseline --no-also > $O/bench_shard_rehearsal_${rows}_rows.json 2>/dev/null
done
echo "rehearsals done"
for l in 0 1 2 4; do ANRAG_SCAN_LANES=$l python scripts/microbench_c2.py 100000 768 2>&1 | tail -1; done > $O/c2_batch1_lanes.txt
for l in 0 4; do ANRAG_SCAN_LANES=$l python scripts/microbench_c2.py 9609 384 2>&1 | tail -1; done >> $O/c2_batch1_lanes.txt
./scripts/exp/launch_gap > $O/launch_gap.txt 2>&1
for t in 2 4 9 16; do for f in tall wide; do N_TERMS=$t ANRAG_BM25_FORM=$f python scripts/microbench_bm25.py 1000000 1024 8 2>&1 | tail -1; done; done > $O/k3_terms_forms.txt
python scripts/microbench_bm25.py 1000000 2048 1 2>&1 | tail -1 >> $O/k3_terms_forms.txt
python scripts/measure_eval_route.py 40 > $O/eval_route.txt 2>&1
python scripts/measure_dropin.py 1000000 768 300 > $O/dropin.txt 2>&1
echo "side measurements done"
fi
if [ "$PART" = "all" ] || [ "$PART" = "b" ]; then
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (no counters in this run)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o hybrid -- python3 $R/bench.py --no-cpu-baseline --no-also \
  > $O/bench_hybrid_under_rocprofv3.json 2> $O/rocprof_bench.err
find $O/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_hybrid_1Mx768_kernel_stats.csv \;
# 2. K3: one query per launch and 8 per launch
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k3 -o k3 -- python3 $R/scripts/microbench_bm25.py 1000000 400 1 > $O/k3_microbench_1.txt 2> $O/rocprof_k3.err
find $O/prof_k3 -name "*kernel_stats.csv" -exec cp {} $O/k3_kernel_stats_1_per_launch.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k3g -o k3 -- python3 $R/scripts/microbench_bm25.py 1000000 1024 8 > $O/k3_microbench_8.txt 2> $O/rocprof_k3g.err
find $O/prof_k3g -name "*kernel_stats.csv" -exec cp {} $O/k3_kernel_stats_8_per_launch.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k3h -o k3 -- python3 $R/scripts/microbench_bm25.py 1000000 2048 16 > $O/k3_microbench_16.txt 2> $O/rocprof_k3h.err
find $O/prof_k3h -name "*kernel_stats.csv" -exec cp {} $O/k3_kernel_stats_16_per_launch.csv \;
# 3. full-ranking mode: the reference's corpus shape and the 1M corpus
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rank1 -o rank -- python3 $R/scripts/microbench_rank.py 9609 384 12000 2048 8 > $O/rank_9609x384.txt 2> $O/rocprof_rank1.err
find $O/prof_rank1 -name "*kernel_stats.csv" -exec cp {} $O/rank_9609x384_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rank2 -o rank -- python3 $R/scripts/microbench_rank.py 1000000 768 12000 256 4 > $O/rank_1Mx768.txt 2> $O/rocprof_rank2.err
find $O/prof_rank2 -name "*kernel_stats.csv" -exec cp {} $O/rank_1Mx768_kernel_stats.csv \;
echo "traces done"
# 4. HBM traffic of K1: separate --pmc passes (kernel trace only beside them)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o dense -- python3 $R/scripts/microbench_dense.py 1000000 768 10 20 \
    > $O/pmc_$c.txt 2> $O/pmc_$c.err
  find $O/pmc_$c -name "*counter_collection.csv" -exec cp {} $O/pmc_${c}_dense_1Mx768.csv \;
done
# ... and of the select + sort kernel over the 1M-row score tile (reads of the tile per pass of the radix select)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rank -o rank -- python3 $R/scripts/microbench_rank.py 1000000 768 12000 256 2 \
  > $O/pmc_rank.txt 2> $O/pmc_rank.err
find $O/pmc_rank -name "*counter_collection.csv" -exec cp {} $O/pmc_FETCH_SIZE_rank_1Mx768.csv \;
echo "pmc done"
fi
rm -rf $O/prof_bench $O/prof_k3 $O/prof_k3g $O/prof_k3h $O/prof_rank1 $O/prof_rank2 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_rank
ls -la $O
