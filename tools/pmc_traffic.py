#!/usr/bin/env python3
"""HBM traffic per launch of the generator kernels from two rocprofv3 PMC passes
(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, collected separately as MI355X_MICROARCH.md prescribes).

    python tools/pmc_traffic.py <fetch_dir>/x_counter_collection.csv <write_dir>/x_counter_collection.csv N

FETCH_SIZE / WRITE_SIZE are in KiB.  The guide's gfx950 caveat (FETCH_SIZE reports half the bytes
of a wide coalesced stream; other patterns uncalibrated) is handled by CALIBRATING on two kernels
of the same run whose traffic is known exactly: flow_mse_partial reads 2 x N*2*224*224*4 bytes
with 16 B/lane loads, flow_mse_bwd writes N*2*224*224*4 bytes with 16 B/lane stores.
"""
import csv
import sys
from collections import defaultdict


def per_kernel(path, counter):
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}       # bytes per launch


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    n = int(sys.argv[3])
    px = n * 224 * 224
    cal_r = next(v for k, v in fetch.items() if "flow_mse_partial" in k)
    cal_w = next(v for k, v in write.items() if "flow_mse_bwd" in k)
    kr, kw = (2 * px * 2 * 4) / cal_r, (px * 2 * 4) / cal_w
    print("# N=%d frames; calibration: FETCH_SIZE x %.3f (flow_mse_partial reads %.1f MB, counter %.1f MB); "
          "WRITE_SIZE x %.3f (flow_mse_bwd writes %.1f MB, counter %.1f MB)"
          % (n, kr, 2 * px * 8 / 1e6, cal_r / 1e6, kw, px * 8 / 1e6, cal_w / 1e6))
    print("kernel,fetch_MB_raw,write_MB_raw,fetch_MB_cal,write_MB_cal,total_B_per_px_cal")
    tot = defaultdict(float)
    for k in sorted(set(fetch) | set(write)):
        if not any(t in k for t in ("gen_", "flow_mse", "pack_params")):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print("%s,%.2f,%.2f,%.2f,%.2f,%.1f" % (name, f / 1e6, w / 1e6, f * kr / 1e6, w * kw / 1e6,
                                             (f * kr + w * kw) / px))
        layer = "gen_layer" in k or "gen_wino" in k          # (template arguments: <MODE, K ...>, MODE 2 = data gradient)
        grp = "fwd" if (("_kernel<0" in k or "_kernel<1" in k) and layer) or "gen_l45" in k or "gen_x3_kernel" in k or "gen_fused_kernel" in k else \
              "bwd" if ("_kernel<2" in k and layer) or "gen_bwd" in k or "gen_wgrad" in k else None
        if grp:
            tot[grp] += f * kr + w * kw
    for g, v in tot.items():
        print("# generator %s total: %.1f MB calibrated = %.1f B/px (algorithmic: fwd 28 B/px fused, "
              "140 B/px with saved features)" % (g, v / 1e6, v / px))
    if len(sys.argv) > 4:
        import hashlib
        import json
        import os
        csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmc-net_amd", "csrc")
        both = b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("gen_tiny.hip", "gen_x3.hip", "gen_fused.hip"))   # as bench.py hashes them
        json.dump({"frames": n, "fetch_scale": kr, "write_scale": kw,
                   "kernel_source_sha16": hashlib.sha256(both).hexdigest()[:16],
                   "gen_fwd_bytes_per_px": tot["fwd"] / px, "gen_bwd_bytes_per_px": tot["bwd"] / px,
                   "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                             "tools/gen_microbench.py, calibrated on flow_mse kernels of known traffic"},
                  open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
