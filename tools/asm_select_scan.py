#!/usr/bin/env python3
"""Heuristic scan of hipcc's gfx950 assembly (--save-temps *.s) for code the compiler lowered badly: a large share of
v_cndmask / s_cselect usually means a run-time index into a register array (a select chain per access) -- the pattern that
cost the input-preparation kernels 2/3 of their instructions (profiles/r4_pmc_prepare.csv).
    python tools/asm_select_scan.py file.s [...]"""
import re
import sys

for path in sys.argv[1:]:
    txt = open(path).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ins = [l.split()[0] for l in body.splitlines() if re.match(r"\s+[vsdgb]\w*_", l)]
        n = len(ins)
        sel = sum(1 for i in ins if i.startswith("v_cndmask") or i.startswith("s_cselect"))
        mfma = sum(1 for i in ins if i.startswith("v_mfma"))
        valu = sum(1 for i in ins if i.startswith("v_")) - mfma
        scr = sum(1 for i in ins if i.startswith("scratch_"))
        if n > 200:
            print("%-78s %6d instr  valu %5d  select %5d (%4.1f %%)  mfma %5d  scratch %d" % (name[14:92], n, valu, sel, 100.0 * sel / n, mfma, scr))
