#!/usr/bin/env python3
"""Merge the two per-kernel PMC tables of tools/pmc_bench.sh (pmc_table.py output) into one row per kernel with the derived
columns: kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs), mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs),
VALU instructions per MFMA, LDS bank-conflict share.  Kernels without MFMA instructions or below 20 us are dropped.
    python tools/pmc_merge.py set1.csv set2.csv"""
import csv
import sys

a = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r["kernel"]: r for r in csv.DictReader(open(sys.argv[2]))}
w = csv.writer(sys.stdout)
w.writerow(["kernel", "kernel_cycles", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "VALU_per_MFMA", "mfma_busy", "SQ_INSTS_LDS",
            "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "bank_conflict_share"])
rows = []
for k in a:
    if k not in b:
        continue
    mf, va, busy = float(a[k]["SQ_INSTS_MFMA"]), float(a[k]["SQ_INSTS_VALU"]), float(a[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
    cyc = float(b[k]["GRBM_GUI_ACTIVE"]) / 8.0
    if mf <= 0 or cyc < 40000:
        continue
    lds, conf, idx = float(b[k]["SQ_INSTS_LDS"]), float(b[k]["SQ_LDS_BANK_CONFLICT"]), float(b[k]["SQ_LDS_IDX_ACTIVE"])
    rows.append((cyc, [k, "%.0f" % cyc, "%.3g" % mf, "%.3g" % va, "%.2f" % (va / mf), "%.3f" % (busy / (cyc * 1024.0)), "%.3g" % lds,
                       "%.3g" % conf, "%.3g" % idx, "%.3f" % (conf / idx if idx else 0.0)]))
for _, r in sorted(rows, key=lambda t: -t[0]):
    w.writerow(r)
