#!/bin/bash
# Tile weight gradient (gen_wgrad_path 4) against the row-sliding kernel (5, the default) on ONE box: bench.py under rocprofv3
# --kernel-trace for both, the weight-gradient kernel's mean duration inside the timed region and the step time.
#   tools/gen_wgrad_ab.sh <out-subdir-of-gpurun_out>   -> gpurun_out/<dir>/gen_wgrad_ab.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
{
for rep in 1 2; do
for p in 4 5; do
  D=/tmp/wab_$p; rm -rf $D
  rocprofv3 --kernel-trace --output-format csv -d $D -o x -- python $R/tools/ab/opt_bench.py gen_wgrad_path $p --steps 20 --warmup 5 --no-cpu-baseline > $OUT/wab_$p.json 2>/dev/null
  python $R/tools/rocprof_region.py $(find $D -name "x_kernel_trace.csv" | head -1) 20 > $OUT/wab_${p}_region.csv
  python - <<PY
import json
d = json.loads(open("$OUT/wab_$p.json").read().strip().splitlines()[-1])
k = [l for l in open("$OUT/wab_${p}_region.csv") if "gen_wgrad_rs_kernel" in l or "gen_bwd_weight_pc_kernel" in l][0].split(",")
print("gen_wgrad_path $p: step %.3f ms (median %.3f); %s %.1f us in the timed region" % (d["ms_per_step"], d["ms_per_step_median"], "gen_wgrad_rs_kernel" if $p == 5 else "gen_bwd_weight_pc_kernel<3>", float(k[-5]) / 1e3))
PY
done
done
} | tee $OUT/gen_wgrad_ab.txt
