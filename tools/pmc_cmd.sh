#!/bin/bash
# Matrix-pipe busy / VALU mix / LDS conflicts / wait split of every kernel of an arbitrary command: separate --pmc passes with
# --kernel-trace only, one table per pass (tools/pmc_table.py) and the merged derived table (tools/pmc_merge.py).
#   tools/pmc_cmd.sh <out-subdir-of-gpurun_out> <command ...>      -> gpurun_out/<dir>/pmc_cmd.csv, pmc_cmd_set{1,2,3}.csv
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  d=/tmp/pmc_cmd_$i; rm -rf $d
  (cd $R && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- "$@" > /dev/null 2>&1)
  python $R/tools/pmc_table.py $(find $d -name "x_counter_collection.csv" | head -1) > $OUT/pmc_cmd_set$i.csv
done
python $R/tools/pmc_merge.py $OUT/pmc_cmd_set1.csv $OUT/pmc_cmd_set2.csv > $OUT/pmc_cmd.csv
cut -c1-260 $OUT/pmc_cmd.csv | head -12
grep -i "p3\|kernel" $OUT/pmc_cmd_set3.csv | cut -c1-260 | head
