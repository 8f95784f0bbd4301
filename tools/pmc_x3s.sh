#!/bin/bash
# PMC passes (separate, --kernel-trace only) over the standalone harness of the pre-split convolutions, per layer shape:
#   tools/pmc_x3s.sh <out-subdir-of-gpurun_out>     -> gpurun_out/<dir>/pmc_<set>_<layer>.csv (tools/pmc_table.py tables)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for sh in "56 56 64 64" "28 28 128 128" "14 14 256 256" "7 7 512 512"; do
    tag=$(echo $sh | tr ' ' 'x')
    d=/tmp/pmc_$i_$tag
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- $R/tools/ubench/bin/conv_x3s_bench 120 $sh 3 7 > /dev/null 2>&1
    python $R/tools/pmc_table.py $(find $d -name "x_counter_collection.csv" | head -1) x3s_conv x3s_wgrad_kernel conv3_kernel conv_wgrad3 > $OUT/pmc_set${i}_$tag.csv
  done
done
head -3 $OUT/pmc_set1_56x56x64x64.csv
