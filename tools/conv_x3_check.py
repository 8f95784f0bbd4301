#!/usr/bin/env python3
"""Accuracy of the NHWC convolutions' two arithmetics against an fp64 evaluation: conv_arith=0 (fp32 MFMA),
conv_arith=1 (fp32 products from three bf16 slices, six bf16 MFMAs per k-block) and PyTorch-ROCm (MIOpen fp32).
Prints max |err| / max |ref| and rms err / rms ref for forward, data gradient and weight gradient."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from dmcnet_amd import ops

dev = "cuda:0"
lib = dmcnet_amd._lib.load()
SHAPES = [("layer1", 8, 64, 56, 64, 3, 1), ("layer2.0", 8, 64, 56, 128, 3, 2), ("layer2.ds", 8, 64, 56, 128, 1, 2),
          ("layer3", 8, 256, 14, 256, 3, 1), ("layer4", 16, 512, 7, 512, 3, 1)]


def ref64(x, w, dy, s, pad):
    x = x.double().requires_grad_(True)
    w = w.double().requires_grad_(True)
    k = w.shape[2]
    xp = F.pad(x, (pad, pad, pad, pad))
    oh = dy.shape[2]
    y = 0
    for ky in range(k):
        for kx in range(k):
            xs = xp[:, :, ky:ky + (oh - 1) * s + 1:s, kx:kx + (oh - 1) * s + 1:s]
            y = y + torch.einsum("nchw,oc->nohw", xs, w[:, :, ky, kx])
    y.backward(dy.double())
    return y.detach(), x.grad, w.grad


def err(a, r):
    a = a.double()
    return "%.2e/%.2e" % (float((a - r).abs().max() / r.abs().max()), float(((a - r) ** 2).mean().sqrt() / (r ** 2).mean().sqrt()))


torch.manual_seed(0)
print("%-10s %-8s %-20s %-20s %-20s" % ("shape", "arith", "fwd max/rms", "dgrad", "wgrad"))
for name, n, cin, hw, cout, k, s in SHAPES:
    pad = k // 2
    x = torch.randn(n, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    oh = (hw + 2 * pad - k) // s + 1
    dy = torch.randn(n, cout, oh, oh, device=dev).contiguous(memory_format=torch.channels_last)
    ry, rdx, rdw = ref64(x, w, dy, s, pad)
    for arith in (0, 1):
        lib.dmc_set_option(b"conv_arith", arith)
        y = ops._conv_fwd(x, w, None, None, s, pad, 0, False)
        y = y[0] if isinstance(y, (tuple, list)) else y
        dx = ops._conv_dgrad(dy, w, x.shape, s, pad)
        dw = ops._conv_wgrad(x, dy, w, s, pad)
        print("%-10s %-8s %-20s %-20s %-20s" % (name, ("f32mfma", "bf16x3")[arith], err(y, ry), err(dx, rdx), err(dw, rdw)))
    y = F.conv2d(x, w, None, s, pad)
    gi, gw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (pad, pad), (1, 1), False, (0, 0), 1, (True, True, False))
    print("%-10s %-8s %-20s %-20s %-20s" % (name, "miopen", err(y, ry), err(gi, rdx), err(gw, rdw)))
