#!/usr/bin/env python3
"""HIP-event timing of the I3D trunk's bf16 3-D convolutions (csrc/conv3d_bf16.hip) against PyTorch-ROCm (MIOpen,
bf16) on the shapes of a 3-clip x 64-frame micro-step: forward, data gradient, weight gradient.  TFLOP/s per kernel.
    python tools/conv3d_microbench.py [name-substring ...] [option=value ...]     (DMC_MB_MIOPEN=0: own kernels only)"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from dmcnet_amd import ops

dev = "cuda:0"
CL3 = torch.channels_last_3d
# name, N, Cin, D, HW, Cout, k
SHAPES = [("2b_1x1", 3, 64, 32, 56, 64, 1), ("2c_3x3", 3, 64, 32, 56, 192, 3),
          ("3b.b0", 3, 192, 32, 28, 64, 1), ("3b.b1", 3, 96, 32, 28, 128, 3), ("3b.b2", 3, 16, 32, 28, 32, 3),
          ("3c.b1", 3, 128, 32, 28, 192, 3), ("3c.b0", 3, 256, 32, 28, 128, 1),
          ("4b.b0", 3, 480, 16, 14, 192, 1), ("4b.b1", 3, 96, 16, 14, 208, 3), ("4c.b1", 3, 112, 16, 14, 224, 3),
          ("4f.b1", 3, 160, 16, 14, 320, 3), ("4f.b0", 3, 528, 16, 14, 256, 1),
          ("5b.b0", 3, 832, 8, 7, 256, 1), ("5c.b1", 3, 192, 8, 7, 384, 3), ("5c.b0", 3, 832, 8, 7, 384, 1)]
names = [a for a in sys.argv[1:] if "=" not in a]
for arg in sys.argv[1:]:
    if "=" in arg:
        k, v = arg.split("=")
        dmcnet_amd._lib.check(dmcnet_amd._lib.load().dmc_set_option(k.encode(), int(v)), "dmc_set_option")
if names:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in names)]
MIO = os.environ.get("DMC_MB_MIOPEN", "1") == "1"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print("%-8s %8s | %21s | %21s | %21s" % ("shape", "GFLOP", "fwd ms (TF) hip/mio", "dgrad hip/mio", "wgrad hip/mio"))
tot = [0.0] * 6
for name, n, cin, d, hw, cout, k in SHAPES:
    x = torch.randn(n, cin, d, hw, hw, device=dev).bfloat16().contiguous(memory_format=CL3)
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    dy = torch.randn(n, cout, d, hw, hw, device=dev).bfloat16().contiguous(memory_format=CL3)
    gf = 2.0 * n * d * hw * hw * cout * cin * k ** 3 / 1e9
    xr = x.detach().requires_grad_(True)
    wr = w.detach().requires_grad_(True)
    res = []

    def own(which):
        xr.grad = wr.grad = None
        xi = xr if which != 2 else x
        wi = wr if which != 1 else w
        if which == 0:
            with torch.no_grad():
                return ops.conv3d_bf16(x, w)
        y = ops.conv3d_bf16(xi, wi)
        y.backward(dy)

    fwd = timeit(lambda: own(0))
    res.append(fwd)
    wb, xb = w.bfloat16(), x.contiguous()
    dyb = dy.contiguous()
    res.append(timeit(lambda: F.conv3d(xb, wb, None, 1, k // 2)) if MIO else 0.0)
    res.append(timeit(lambda: own(1)) - fwd)
    bw = lambda m: torch.ops.aten.convolution_backward(dyb, xb, wb, None, (1, 1, 1), (k // 2,) * 3, (1, 1, 1), False, (0, 0, 0), 1, m)
    res.append(timeit(lambda: bw((True, False, False))) if MIO else 0.0)
    res.append(timeit(lambda: own(2)) - fwd)
    res.append(timeit(lambda: bw((False, True, False))) if MIO else 0.0)
    f = lambda ms: "%.3f(%5.0f)" % (ms, gf / ms if ms > 0 else 0.0)
    for i in range(6):
        tot[i] += res[i]
    print("%-8s %8.1f | %s %s | %s %s | %s %s" % (name, gf, f(res[0]), f(res[1]), f(res[2]), f(res[3]), f(res[4]), f(res[5])))
print("%-8s %8s | %.3f %.3f | %.3f %.3f | %.3f %.3f   (ms, sums over the listed shapes)" % ("total", "", *tot))
