#!/usr/bin/env python3
"""Generator forward / backward (120 frames) on random and on all-zero inputs AND weights: the same instruction stream;
a difference is the clock the chip sustains (power), not the kernel structure (profiles/r3_gen_power.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd

dev = "cuda:0"
torch.manual_seed(0)


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


for fill in ("random", "zero", "random", "zero"):
    m = dmcnet_amd.model.EstimatorDenseNetTiny(5).to(dev)
    mv, res = torch.randn(120, 2, 224, 224, device=dev), torch.randn(120, 3, 224, 224, device=dev)
    if fill == "zero":
        mv.zero_(); res.zero_()
        with torch.no_grad():
            for p in m.parameters():
                p.zero_()
    with torch.no_grad():
        f = timeit(lambda: m.forward_mv_res(mv, res, add_mv=True))
    g = torch.randn(120, 2, 224, 224, device=dev) if fill == "random" else torch.zeros(120, 2, 224, 224, device=dev)

    def fb():
        for p in m.parameters():
            p.grad = None
        m.forward_mv_res(mv, res, add_mv=True).backward(g)
    fbt = timeit(fb)
    px = 120 * 224 * 224
    print("%-6s forward %.3f ms (%.1f TFLOP/s, %.3f of the fp32 peak)   forward + backward %.3f ms" % (fill, f, px * 9108 / f / 1e9, px * 9108 / f / 1e9 / 157.3, fbt))
