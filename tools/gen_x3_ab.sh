#!/bin/bash
# per-kernel times of the generator forward with the hidden layers on gen_x3.hip (option gen_x3) vs the fp32 kernels:
#   tools/gen_x3_ab.sh <out-subdir> [mask ...]
O=$1; shift; OUT=$GRAFT_REPO_ROOT/gpurun_out/$O; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in ${@:-0 7}; do
  D=/tmp/gx3_$m; rm -rf $D
  DMC_MB_ITERS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o x -- python $R/tools/gen_microbench.py 120 gen_x3=$m > $OUT/mb_$m.txt 2>/dev/null
  python - $D $m <<'PY' | tee $OUT/kernels_$m.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/x_kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Name"].startswith(("void (anonymous namespace)::gen_", "(anonymous namespace)::gen_", "gen_")) or "gen_" in r["Name"][:60]]
print("# gen_x3 =", sys.argv[2])
tot = 0.0
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "")[:70]
    print("%-72s calls %4s avg %9.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  grep fwd $OUT/mb_$m.txt
done
