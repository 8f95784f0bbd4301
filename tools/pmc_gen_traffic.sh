#!/bin/bash
# Two separate PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over tools/gen_microbench.py, then tools/pmc_traffic.py:
#   tools/pmc_gen_traffic.sh <out-subdir-of-gpurun_out> [round-tag]   -> gpurun_out/<dir>/gen_traffic.csv / .json
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o x -- python $R/tools/gen_microbench.py 120 > $OUT/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "x_counter_collection.csv" | head -1) \
       $(find /tmp/pmc_WRITE_SIZE -name "x_counter_collection.csv" | head -1) 120 $OUT/gen_traffic.json > $OUT/gen_traffic.csv
tail -4 $OUT/gen_traffic.csv
