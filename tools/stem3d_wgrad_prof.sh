#!/bin/bash
# per-kernel times of tools/stem3d_wgrad_bench.py under rocprofv3 (bounded):  tools/stem3d_wgrad_prof.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_w2
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w2 -o w2 -- python $R/tools/stem3d_wgrad_bench.py > /tmp/w2_bench.json 2>/dev/null
cat /tmp/w2_bench.json
f=$(find /tmp/prof_w2 -name "*kernel_stats.csv" | head -1)
test -n "$f" && grep stem3d $f | cut -d, -f1-4 | cut -c1-140
