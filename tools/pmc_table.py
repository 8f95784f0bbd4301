#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 counter_collection CSV (one --pmc pass):
    python tools/pmc_table.py <dir>/x_counter_collection.csv [name-filter ...]"""
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if len(sys.argv) > 2 and not any(f in k for f in sys.argv[2:]):
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in agg.values() for c in v})
out = csv.writer(sys.stdout)
out.writerow(["kernel"] + names)
for k in sorted(agg):
    out.writerow([k] + ["%.4g" % (sum(agg[k][c]) / max(1, len(agg[k][c]))) for c in names])
