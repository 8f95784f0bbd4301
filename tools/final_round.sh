#!/bin/bash
# the round's closing measurements in one gpurun call:  tools/final_round.sh <out-subdir-of-gpurun_out>
O=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$O; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# HBM traffic of the generator kernels first: bench.py reports roofline.traffic only from a measurement of the sources in the tree
tools/pmc_gen_traffic.sh $O > /dev/null 2>&1
cp $OUT/gen_traffic.json profiles/r5_gen_traffic.json; cp $OUT/gen_traffic.csv profiles/r5_gen_traffic.csv
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --config gan --no-cpu-baseline > $OUT/bench_gan.json 2>/dev/null
python bench.py --config i3d --no-cpu-baseline > $OUT/bench_i3d.json 2>/dev/null
tools/profile_region.sh $O r5_bench > /dev/null 2>&1
tools/profile_region.sh $O r5_bench_one_stream DMC_WGRAD_STREAM=0 > /dev/null 2>&1
BENCH_ARGS="--config gan" tools/profile_region.sh $O r5_gan > /dev/null 2>&1
BENCH_ARGS="--config i3d" tools/profile_region.sh $O r5_i3d_one_stream DMC_I3D_BRANCH_STREAMS=0 > /dev/null 2>&1
tools/pmc_bench.sh $O > /dev/null 2>&1
{ python tools/x3q_microbench.py 120; python tools/x3q_microbench.py 120 conv_cfg=201; } > $OUT/x3q_microbench.txt 2>/dev/null
python - <<PY
import json
for f in ("default", "gan", "i3d"):
    d = json.load(open("$OUT/bench_%s.json" % f))
    print(f, d["ms_per_step"], d.get("ms_per_step_median"), d["value"], d.get("host_clean_ms_per_step"), d["roofline"]["frac"], (d.get("roofline_classifier_convs") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
PY
