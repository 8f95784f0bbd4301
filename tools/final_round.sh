#!/bin/bash
# the round's closing measurements in one gpurun call:  tools/final_round.sh <out-subdir-of-gpurun_out> [round tag, default r6]
# copies the tables the documents cite into profiles/ (tracked); everything else stays in gpurun_out/<dir>
O=$1; T=${2:-r6}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$O; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/gpurun_out/$O/profiles; mkdir -p $P
# HBM traffic of the generator kernels first: bench.py reports roofline.traffic only from a measurement of the sources in the tree
tools/pmc_gen_traffic.sh $O > /dev/null 2>&1
cp $OUT/gen_traffic.json profiles/${T}_gen_traffic.json; cp $OUT/gen_traffic.csv profiles/${T}_gen_traffic.csv
cp $OUT/gen_traffic.json $P/${T}_gen_traffic.json; cp $OUT/gen_traffic.csv $P/${T}_gen_traffic.csv
cd $GRAFT_REPO_ROOT
python bench.py > $P/${T}_bench_default.json 2> $OUT/bench_default.err
python bench.py --config gan --no-cpu-baseline > $P/${T}_bench_gan.json 2>/dev/null
python bench.py --config i3d --no-cpu-baseline > $P/${T}_bench_i3d.json 2>/dev/null
python bench.py --config i3d --graph 0 --no-cpu-baseline > $P/${T}_bench_i3d_eager.json 2>/dev/null
tools/profile_region.sh $O ${T}_bench > /dev/null 2>&1
tools/profile_region.sh $O ${T}_bench_one_stream DMC_WGRAD_STREAM=0 > /dev/null 2>&1
BENCH_ARGS="--config gan" tools/profile_region.sh $O ${T}_bench_gan > /dev/null 2>&1
BENCH_ARGS="--config i3d --graph 0" tools/profile_region.sh $O ${T}_i3d > /dev/null 2>&1
BENCH_ARGS="--config i3d --graph 0" tools/profile_region.sh $O ${T}_i3d_one_stream DMC_I3D_BRANCH_STREAMS=0 DMC_WGRAD_STREAM=0 > /dev/null 2>&1
for f in ${T}_bench_timed_region ${T}_bench_one_stream_timed_region ${T}_bench_gan_timed_region ${T}_i3d_timed_region ${T}_i3d_one_stream_timed_region ${T}_bench_kernel_stats; do cp $OUT/$f.csv $P/ 2>/dev/null; done
tools/pmc_bench.sh $O > /dev/null 2>&1; cp $OUT/pmc_bench.csv $P/${T}_pmc_mfma_busy.csv
tools/pmc_i3d_r4.sh $O > /dev/null 2>&1; cp $OUT/pmc_i3d.csv $P/${T}_pmc_i3d_kernels.csv
python tools/coviar_post_bench.py > $P/${T}_coviar_post_bench.json 2>/dev/null
python - <<PY
import json
for f in ("default", "gan", "i3d", "i3d_eager"):
    d = json.load(open("$P/${T}_bench_%s.json" % f))
    print(f, d["ms_per_step"], d.get("ms_per_step_median"), d["value"], d.get("host_clean_ms_per_step"), d["roofline"]["frac"], d["roofline"].get("traffic"), (d.get("roofline_classifier_convs") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
PY
