#!/bin/bash
# rocprofv3 kernel trace of bench.py cut to the timed region:  tools/profile_region.sh <out-subdir> <tag> [env ...] [-- bench args]
#   -> gpurun_out/<dir>/<tag>_timed_region.csv, <tag>_kernel_stats.csv, <tag>_under_rocprof.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; TAG=$2; shift 2
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
D=/tmp/prof_$TAG; rm -rf $D
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $D -o x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $BENCH_ARGS > $OUT/${TAG}_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
python $R/tools/rocprof_region.py $(find $D -name "x_kernel_trace.csv" | head -1) 20 > $OUT/${TAG}_timed_region.csv
cp $(find $D -name "x_kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
head -3 $OUT/${TAG}_timed_region.csv | cut -c1-200
