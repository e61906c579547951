#!/bin/bash
# Where K3's waves spend their cycles (SQ counters, separate --pmc passes), per form (ANRAG_BM25_FORM tall / wide) at
# 8 queries per launch.  gpurun --timeout 900 -- 'bash scripts/refresh_pmc_k3.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03k3
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for form in tall wide; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    ANRAG_BM25_FORM=$form rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${form}_$i -o k3 -- python3 $R/scripts/microbench_bm25.py 1000000 512 8 > $O/${form}_$i.txt 2> $O/${form}_$i.err
    find $O/${form}_$i -name "*counter_collection.csv" -exec cp {} $O/pmc_${form}_$i.csv \;
    rm -rf $O/${form}_$i
    echo "$form pass $i done"
  done
done
ls $O
