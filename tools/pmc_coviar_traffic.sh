#!/bin/bash
# HBM-side counters of the post-decode extraction kernels: two separate PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over
# tools/coviar_post_bench.py.  Raw KiB counters x 1024 per launch; the guide's gfx950 caveat applies to FETCH_SIZE (it reports half the
# bytes of a wide coalesced stream; gathers uncalibrated), WRITE_SIZE is exact on the calibration kernels of tools/pmc_traffic.py.
#   tools/pmc_coviar_traffic.sh <out-subdir-of-gpurun_out>   -> gpurun_out/<dir>/coviar_traffic.csv
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcc_$c -o x -- python $R/tools/coviar_post_bench.py --cpu-chains 1 > $OUT/pmcc_$c.log 2>&1
done
python - $(find /tmp/pmcc_FETCH_SIZE -name "x_counter_collection.csv" | head -1) $(find /tmp/pmcc_WRITE_SIZE -name "x_counter_collection.csv" | head -1) > $OUT/coviar_traffic.csv <<'PY'
import csv, sys
from collections import defaultdict
def per_kernel(path, counter):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return agg
f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
print("kernel,launches,FETCH_SIZE_bytes_per_launch_raw,FETCH_x2_bytes,WRITE_SIZE_bytes_per_launch,min_write,max_write")
for k in sorted(f):
    if "owner_tile" in k or "gop_trace" in k:
        fv, wv = f[k], w.get(k, [0.0])
        print('"%s",%d,%.0f,%.0f,%.0f,%.0f,%.0f' % (("gop_trace_kernel" if "gop_trace" in k else "mv_owner_tile_kernel"), len(fv), sum(fv) / len(fv), 2 * sum(fv) / len(fv), sum(wv) / len(wv), min(wv), max(wv)))
PY
cat $OUT/coviar_traffic.csv
