#!/usr/bin/env python3
"""LDS bank model (MI355X_MICROARCH.md, LDS table) for the reads of the row-sliding generator weight gradient
(gen_wgrad.hip): picks the plane / row pitches of the bf16-slice rings.  ds_read_b128: four groups of 16 lanes, bank =
dword address mod 64, a lane covers 4 banks; ds_read_b32: two groups of 32 lanes, bank = dword address mod 32."""
import itertools

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G32 = [list(range(0, 32)), list(range(32, 64))]


def cycles_b128(addr):
    tot = 0
    for grp in G128:
        banks = {}
        for l in grp:
            for b in range(4):
                banks.setdefault((addr[l] + b) % 64, set()).add(addr[l])
        tot += max(len(v) for v in banks.values())
    return tot


def cycles_b32(addr):
    tot = 0
    for grp in G32:
        banks = {}
        for l in grp:
            banks.setdefault(addr[l] % 32, set()).add(addr[l])
        tot += max(len(v) for v in banks.values())
    return tot


def window_cost(XP, XROW):
    c128 = c32 = n = 0
    for gt in range(7):
        for r0 in range(6):
            a = []
            for lane in range(64):
                j, kq = lane & 15, lane >> 4
                g = min(16 * gt + j, 99)
                ci, dy = g // 3, g % 3
                a.append(((r0 + dy) % 6) * XROW + ci * XP + 4 * kq + 4)
            c128 += cycles_b128(a)
            c32 += cycles_b32([x - 1 for x in a]) + cycles_b32([x + 4 for x in a])
            n += 1
    return c128 / n, c32 / n / 2


def afrag_cost(GP):
    a = [(lane & 15) * GP + 4 * (lane >> 4) for lane in range(64)]
    return cycles_b128(a)


if __name__ == "__main__":
    best = []
    for XP in range(24, 44, 4):
        slab = 3 * 34 * XP
        for pad in range(0, 64, 4):
            XROW = slab + pad
            c128, c32 = window_cost(XP, XROW)
            best.append((c128 + 2 * c32, c128, c32, XP, XROW, pad))
    best.sort()
    for b in best[:12]:
        print("XP %d XROW %d (pad %d): b128 %.2f cycles (4 ideal), b32 %.2f (2 ideal)" % (b[3], b[4], b[5], b[1], b[2]))
    for GP in range(16, 40, 4):
        print("GP", GP, "A-fragment b128 cycles", afrag_cost(GP))
