#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box through gpurun; outputs land in
# gpurun_out/refresh/ and are copied into profiles/ by hand after a look).
#   gpurun --timeout 1100 -- 'bash scripts/refresh_profiles.sh'
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py > $O/bench_hybrid_1Mx768.json 2> $O/bench_hybrid.err
python bench.py --workload dense --rows 100000 --no-cpu-baseline > $O/bench_c2_dense_100kx768.json 2>/dev/null
python bench.py --rows 1000000 --dim 1024 --no-cpu-baseline > $O/bench_c5_one_rank_1Mx1024.json 2>/dev/null
for rows in 500000 250000 125000; do
  ANRAG_FORCE_SHARDED=1 python bench.py --rows $rows --no-cpu-baseline > $O/bench_shard_rehearsal_$rows.json 2>/dev/null
done
# the same default command under rocprofv3 (kernel trace + stats only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o hybrid -- python $R/bench.py --no-cpu-baseline \
  > $O/bench_hybrid_under_rocprofv3.json 2> $O/rocprof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cd $R
for f in bench_hybrid_1Mx768 bench_c2_dense_100kx768 bench_c5_one_rank_1Mx1024 bench_shard_rehearsal_500000 \
         bench_shard_rehearsal_250000 bench_shard_rehearsal_125000 bench_hybrid_under_rocprofv3; do
  python scripts/summ.py $f < $O/$f.json
done
head -4 $O/kernel_stats.csv | cut -c1-60,150-260
