#!/usr/bin/env python3
"""Per-shape timings of the stride-2 block pair (conv_x3q.hip) at N frames: forward (conv1 + shortcut + both statistics),
data gradient (both branches), weight gradients (both + their reductions), HIP-event timed through the C ABI wrappers.
    python tools/x3q_microbench.py [frames=120] [name=value ...]      e.g. conv_cfg=201 (full-size workgroups only)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd                       # noqa: E402
from dmcnet_amd import ops              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
lib = dmcnet_amd._lib.load()
for arg in sys.argv[2:]:
    k, v = arg.split("=")
    dmcnet_amd._lib.check(lib.dmc_set_option(k.encode(), int(v)), "dmc_set_option")
CL = torch.channels_last


def timed(f, it=20):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


print("# %d frames; GFLOP per pass = 2 x pixels x Cin x Cout x (9 + 1); TFLOP/s fp32-equivalent (bf16x3 bound: 416.7)" % n)
for (cin, h, cout) in ((64, 56, 128), (128, 28, 256), (256, 14, 512)):
    oh = h // 2
    x = torch.randn(n, cin, h, h, device="cuda").contiguous(memory_format=CL)
    xq = ops.x3q_split(x)
    w3 = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.1).contiguous(memory_format=CL)
    w1 = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.1).contiguous(memory_format=CL)
    wf, wt = ops.x3q_pack_weights(w3, w1)
    g = torch.randn(n, cout, oh, oh, device="cuda").contiguous(memory_format=CL)
    d3, d1 = ops.x3s_split(g), ops.x3s_split(g * 0.5)
    tf = timed(lambda: ops.x3q_conv_fwd(xq, wf, n, oh, oh, cin, cout, want_stats=True))
    td = timed(lambda: ops.x3q_conv_dgrad(d3, d1, wt, n, oh, oh, cin, cout))
    tw = timed(lambda: ops.x3q_conv_wgrad(xq, d3, d1, n, oh, oh, cin, cout))
    gf = 2.0 * n * oh * oh * cin * cout * 10 / 1e9
    print("%3d -> %3d @ %2d x %2d (%5.2f GFLOP): forward %6.1f us (%5.1f)   data gradient %6.1f us (%5.1f)   weight gradients %6.1f us (%5.1f)"
          % (cin, cout, oh, oh, gf, tf, gf / tf * 1e3, td, gf / td * 1e3, tw, gf / tw * 1e3))
