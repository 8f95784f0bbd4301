#!/usr/bin/env python3
"""HIP-event timing of the NHWC matrix-core convolutions against PyTorch-ROCm (MIOpen) on the same
channels_last tensors: forward, data gradient, weight gradient of the ResNet-18 (120 frames) and
discriminator (240 frames) shapes.  Prints TFLOP/s per kernel."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from dmcnet_amd import ops

dev = "cuda:0"
SHAPES = [("rn.layer1", 120, 64, 56, 64, 3, 1), ("rn.layer2.0.c1", 120, 64, 56, 128, 3, 2),
          ("rn.layer2.0.ds", 120, 64, 56, 128, 1, 2), ("rn.layer2", 120, 128, 28, 128, 3, 1),
          ("rn.layer3", 120, 256, 14, 256, 3, 1), ("rn.layer4", 120, 512, 7, 512, 3, 1),
          ("d.1_2", 240, 16, 112, 16, 3, 1), ("d.2", 240, 16, 112, 32, 3, 2), ("d.2_2", 240, 32, 56, 32, 3, 1),
          ("d.3_2", 240, 64, 28, 64, 3, 1), ("d.4_2", 240, 128, 14, 128, 3, 1)]
names = [a for a in sys.argv[1:] if "=" not in a]
for arg in sys.argv[1:]:                      # kernel-selection options for A/B runs, e.g. conv_cfg=3 conv_path=0
    if "=" in arg:
        k, v = arg.split("=")
        dmcnet_amd._lib.check(dmcnet_amd._lib.load().dmc_set_option(k.encode(), int(v)), "dmc_set_option")
if names:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in names)]
MIO = os.environ.get("DMC_MB_MIOPEN", "1") == "1"
torch.backends.cudnn.benchmark = True


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print("%-16s %8s | %21s | %21s | %21s" % ("shape", "GFLOP", "fwd ms (TF) hip/mio", "dgrad hip/mio", "wgrad hip/mio"))
for name, n, cin, hw, cout, k, s in SHAPES:
    pad = k // 2
    x = torch.randn(n, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    oh = (hw + 2 * pad - k) // s + 1
    dy = torch.randn(n, cout, oh, oh, device=dev).contiguous(memory_format=torch.channels_last)
    gf = 2.0 * n * oh * oh * cout * cin * k * k / 1e9
    res = []
    res.append(timeit(lambda: ops._conv_fwd(x, w, None, None, s, pad, 0, False)))
    res.append(timeit(lambda: F.conv2d(x, w, None, s, pad)) if MIO else 0.0)
    res.append(timeit(lambda: ops._conv_dgrad(dy, w, x.shape, s, pad)))
    res.append(timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (pad, pad), (1, 1), False, (0, 0), 1, (True, False, False))) if MIO else 0.0)
    res.append(timeit(lambda: ops._conv_wgrad(x, dy, w, s, pad)))
    res.append(timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (pad, pad), (1, 1), False, (0, 0), 1, (False, True, False))) if MIO else 0.0)
    f = lambda ms: "%.3f(%5.1f)" % (ms, gf / ms if ms else 0.0)
    print("%-16s %8.1f | %s %s | %s %s | %s %s" % (name, gf, f(res[0]), f(res[1]), f(res[2]), f(res[3]), f(res[4]), f(res[5])))
