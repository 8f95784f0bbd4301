#!/bin/bash
# Round-2 verdict, generator variant (ii): forward layers 2 and 3 on 4-row tiles with two workgroups per CU (option
# gen_layer_path = 3) against the default (1): per-kernel durations (rocprofv3 --kernel-trace --stats) and one PMC pass.
#   tools/gen_variant_ii.sh <out-subdir-of-gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for p in 5 1 3 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gv_$p -o x -- python $R/tools/gen_microbench.py 120 gen_layer_path=$p > $OUT/microbench_path$p.txt 2>/dev/null
  python - $(find /tmp/gv_$p -name "x_kernel_stats.csv" | head -1) > $OUT/kernels_path$p.csv <<'PY'
import csv, sys
print("kernel,calls,avg_us")
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if k.startswith("gen_layer_mfma_kernel<0") or k.startswith("gen_layer_gather_kernel<0") or k.startswith("gen_l45"):
        print('"%s",%s,%.1f' % (k, r["Calls"], float(r["AverageNs"]) / 1000))
PY
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/gp_$p -o x -- python $R/tools/gen_microbench.py 120 gen_layer_path=$p > /dev/null 2>&1
  python $R/tools/pmc_table.py $(find /tmp/gp_$p -name "x_counter_collection.csv" | head -1) "gen_layer_mfma_kernel<0" > $OUT/pmc_path$p.csv
done
cat $OUT/kernels_path5.csv $OUT/kernels_path1.csv $OUT/kernels_path3.csv $OUT/kernels_path4.csv $OUT/pmc_path5.csv $OUT/pmc_path1.csv $OUT/pmc_path3.csv $OUT/pmc_path4.csv
