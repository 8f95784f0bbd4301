#!/usr/bin/env python3
"""Idle gaps between consecutive kernels inside bench.py's timed region (kernel trace CSV):
where the GPU waits for the host.  python tools/rocprof_gaps.py <trace.csv> [steps]"""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
marks = [i for i, r in enumerate(rows) if "dmc_profile_mark_kernel" in r[2]]
region = rows[marks[0] + 1:marks[-1]]
gaps = defaultdict(lambda: [0, 0])
end = region[0][1]
prev = region[0][2]
total = 0
big = []
for s, e, name in region[1:]:
    g = s - end
    if g > 2000:
        key = name[:70]
        gaps[key][0] += g; gaps[key][1] += 1
        total += g
        if g > 500000:
            big.append((g, prev[:60], name[:60]))
    if e > end:
        end, prev = e, name
print("total idle (gaps > 2us): %.3f ms/step" % (total / steps / 1e6))
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%-72s %8.1f us/step in %5.1f gaps/step (before this kernel)" % (k, g / steps / 1e3, n / steps))
for g, a, b in sorted(big, reverse=True)[:8]:
    print("gap %.2f ms between [%s] and [%s]" % (g / 1e6, a, b))
