#!/usr/bin/env python3
"""Cut a rocprofv3 kernel trace (``--kernel-trace --output-format csv``) to the region bench.py
brackets with ``dmc_profile_mark_kernel`` and print per-kernel statistics for that region.

    python tools/rocprof_region.py <dir>/<prefix>_kernel_trace.csv [steps] > profiles/...csv
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "dmc_profile_mark_kernel" in r[2]]
    if len(marks) < 2:
        raise SystemExit("need two dmc_profile_mark_kernel launches, found %d" % len(marks))
    region = rows[marks[0] + 1:marks[-1]]
    span_ns = rows[marks[-1]][0] - rows[marks[0]][1]
    agg = defaultdict(list)
    for s, e, name in region:
        agg[name].append(e - s)
    busy = sum(sum(v) for v in agg.values())
    w = csv.writer(sys.stdout)
    w.writerow(["# region between dmc_profile_mark_kernel launches: %d kernels, span %.3f ms, "
                "kernel-busy %.3f ms, %d steps" % (len(region), span_ns / 1e6, busy / 1e6, steps)])
    w.writerow(["Name", "Calls", "CallsPerStep", "TotalDurationNs", "AverageNs", "MsPerStep",
                "Percentage", "MinNs", "MaxNs"])
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name[:160], len(v), "%.2f" % (len(v) / steps), sum(v), "%.1f" % (sum(v) / len(v)),
                    "%.4f" % (sum(v) / steps / 1e6), "%.3f" % (100.0 * sum(v) / busy), min(v), max(v)])


if __name__ == "__main__":
    main()
