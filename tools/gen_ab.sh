#!/bin/bash
# per-kernel times of the generator micro-benchmark under option sets:  tools/gen_ab.sh <out-subdir> "opt=v opt=v" "..." ...
O=$1; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$O; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
# option sets that name gen_ablate / conv_ablate / gen_stagger need the measurement build (python dmc-net_amd/build.py --measure):
# ... and so do the values that select variants which lost their measurement (gen_fused 2 / 3, gen_layer_path 3 .. 5, gen_wgrad_path 0 .. 3)
case "$*" in *ablate*|*stagger*|*gen_fused=[23]*|*gen_layer_path=[345]*|*gen_wgrad_path=[0123]*) export DMC_HIP_LIB=$R/dmc-net_amd/libdmcnet_hip_measure.so;; esac
cd /tmp && export TMPDIR=/tmp
i=0
for m in "$@"; do
  D=/tmp/gab_$i; rm -rf $D
  DMC_MB_ITERS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o x -- python $R/tools/gen_microbench.py 120 $m > $OUT/mb_$i.txt 2>/dev/null
  python - $D "$m" <<'PY' | tee $OUT/kernels_$i.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/x_kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gen_" in r["Name"][:70]]
print("# options:", sys.argv[2])
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print("%-62s calls %4s avg %9.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  grep -h "fwd\|bwd" $OUT/mb_$i.txt
  i=$((i+1))
done
