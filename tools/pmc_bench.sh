#!/bin/bash
# Matrix-pipe busy / VALU mix / LDS conflicts of EVERY kernel of a short bench.py run (config 2, one stream): two separate
# --pmc passes with --kernel-trace only, merged per kernel by tools/pmc_merge.py.
#   tools/pmc_bench.sh <out-subdir-of-gpurun_out> [bench args]      -> gpurun_out/<dir>/pmc_bench.csv
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  d=/tmp/pmc_bench_$i; rm -rf $d
  DMC_WGRAD_STREAM=0 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o x -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
  python $R/tools/pmc_table.py $(find $d -name "x_counter_collection.csv" | head -1) > $OUT/pmc_bench_set$i.csv
done
python $R/tools/pmc_merge.py $OUT/pmc_bench_set1.csv $OUT/pmc_bench_set2.csv > $OUT/pmc_bench.csv
head -5 $OUT/pmc_bench.csv | cut -c1-220
