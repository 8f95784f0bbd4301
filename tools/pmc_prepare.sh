#!/bin/bash
# What stalls the input-preparation kernels: two separate PMC passes (--kernel-trace only) over tools/prepare_microbench.py
#   tools/pmc_prepare.sh <out-subdir-of-gpurun_out>   -> gpurun_out/<dir>/pmc_prepare_set<i>_ds<factor>.csv
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for ds in 16 0; do
    d=/tmp/pmcp_${i}_$ds; rm -rf $d
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python $R/tools/prepare_microbench.py $ds 5 > /dev/null 2>&1
    python $R/tools/pmc_table.py $(find $d -name "x_counter_collection.csv" | head -1) prepare_crop > $OUT/pmc_prepare_set${i}_ds$ds.csv
  done
done
cat $OUT/pmc_prepare_set*_ds16.csv
