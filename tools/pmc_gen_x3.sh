#!/bin/bash
# PMC rows of the generator forward kernels (tools/gen_microbench.py 120 <options>): separate --pmc passes, --kernel-trace only
#   tools/pmc_gen_x3.sh <out-subdir> [option=value ...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  d=/tmp/pmc_gx_$i; rm -rf $d
  DMC_MB_ITERS=3 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python $R/tools/gen_microbench.py 120 "$@" > /dev/null 2>$OUT/pmc_err_$i.txt
  python $R/tools/pmc_table.py $(find $d -name "x_counter_collection.csv" | head -1) | grep -i "^kernel\|gen_x3\|gen_layer_mfma_kernel\|gen_layer_gather_kernel\|l45\|gen_wino" > $OUT/pmc_gx_set$i.csv
  cat $OUT/pmc_gx_set$i.csv | cut -c1-260
done
