#!/bin/bash
# PMC passes (separate, --kernel-trace only) over a short I3D bench run: matrix-pipe busy / VALU mix / LDS conflicts of the 3-D
# kernels (the row-ring weight gradient against the tap-stepping one: option conv3d_wgrad 2 / 0).
#   tools/pmc_i3d.sh <out-subdir-of-gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  for opt in 2 0; do
    d=/tmp/pmc_i3d_${i}_$opt
    DMC_I3D_BRANCH_STREAMS=0 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python $R/tools/ab/opt_bench.py conv3d_wgrad $opt --config i3d --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
    python $R/tools/pmc_table.py $(find $d -name "x_counter_collection.csv" | head -1) conv3d_wgrad conv3d_bf16_kernel stem_dgrad > $OUT/pmc_set${i}_wgrad$opt.csv
  done
done
head -4 $OUT/pmc_set1_wgrad2.csv
