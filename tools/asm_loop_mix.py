#!/usr/bin/env python3
"""Instruction mix of the innermost loops of a kernel in hipcc's -S output.

    python tools/asm_loop_mix.py file.s <kernel-name-substring> [...]
For every backward branch in the kernel, prints the instruction classes between the label and the branch.
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load_lds") :
        return "dma"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    for pat in sys.argv[2:]:
        starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l]
        for s in starts:
            e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
            body = lines[s:e]
            labels = {}
            insts = []
            for l in body:
                m = re.match(r"^(\.LBB\S+):", l)
                if m:
                    labels[m.group(1)] = len(insts)
                    continue
                t = l.strip()
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                insts.append(t)
            print("==", lines[s][:100], "insts", len(insts))
            tot = Counter(classify(t.split()[0]) for t in insts)
            print("   whole:", dict(tot))
            for i, t in enumerate(insts):
                m = re.match(r"s_cbranch\S*\s+(\.LBB\S+)", t)
                if m and m.group(1) in labels and labels[m.group(1)] <= i:
                    lo = labels[m.group(1)]
                    c = Counter(classify(x.split()[0]) for x in insts[lo:i + 1])
                    print("   loop %s [%d insts]: %s" % (m.group(1), i + 1 - lo, dict(c)))
                    if "-v" in sys.argv:
                        vc = Counter(x.split()[0] for x in insts[lo:i + 1] if classify(x.split()[0]) in ("valu", "salu"))
                        print("      ", vc.most_common(40))


if __name__ == "__main__":
    main()
