#!/bin/bash
# per-kernel times of one K2 pass (both arithmetic modes) under rocprofv3: bash scripts/k2_breakdown.sh  (on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in bf16x3 f32; do
  [ -n "$K2LIB" ] && export ANRAG_LIB=$K2LIB
  ITERS=100 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k2prof_$m -o k2 -- \
    python3 $R/scripts/microbench_batched.py 1000000 768 256 10 $m > /dev/null 2>&1 || exit 1
done
python3 - <<PY
import csv,glob
for m in ('bf16x3','f32'):
    f=glob.glob('$R/gpurun_out/k2prof_%s/**/*kernel_stats.csv'%m,recursive=True)[0]
    print(m)
    for r in list(csv.DictReader(open(f)))[:12]:
        if 'anrag' in r['Name']: print('  %-84s %4s %10.1f us'%(r['Name'][:84], r['Calls'], float(r['AverageNs'])/1e3))
PY
