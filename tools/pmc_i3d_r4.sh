#!/bin/bash
# Round-4 refresh of the I3D kernels' PMC rows (two separate --pmc passes, --kernel-trace only, one stream):
#   tools/pmc_i3d_r4.sh <out-subdir-of-gpurun_out>   -> gpurun_out/<dir>/pmc_i3d.csv  (merged by tools/pmc_merge.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  d=/tmp/pmc_i3d4_$i; rm -rf $d
  DMC_I3D_BRANCH_STREAMS=0 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python $R/bench.py --config i3d --graph 0 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/pmc_table.py $(find $d -name "x_counter_collection.csv" | head -1) > $OUT/pmc_i3d_set$i.csv
done
python $R/tools/pmc_merge.py $OUT/pmc_i3d_set1.csv $OUT/pmc_i3d_set2.csv > $OUT/pmc_i3d.csv
grep -i "conv3d\|stem3d\|pool3d\|bn3d" $OUT/pmc_i3d.csv | head -20 | cut -c1-200
