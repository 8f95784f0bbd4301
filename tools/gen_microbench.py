#!/usr/bin/env python3
"""Micro-benchmark of the generator kernels alone (N frames of 224x224), HIP-event timed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from dmcnet_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
# kernel-selection options for A/B runs: name=value arguments, e.g. gen_fuse_fwd=0 gen_gather=0
for arg in sys.argv[2:]:
    k, v = arg.split("=")
    dmcnet_amd._lib.check(dmcnet_amd._lib.load().dmc_set_option(k.encode(), int(v)), "dmc_set_option")
dev = "cuda:0"
torch.manual_seed(0)
m = dmcnet_amd.model.EstimatorDenseNetTiny(5).to(dev)
mv, res = torch.randn(N, 2, 224, 224, device=dev), torch.randn(N, 3, 224, 224, device=dev)
r = torch.randn(N, 2, 224, 224, device=dev)
flow = torch.randn(N, 2, 224, 224, device=dev)
for _ in range(3):
    m.zero_grad(); y = m.forward_mv_res(mv, res, True); y.backward(r)
    yy = y.detach().requires_grad_(True); ops.flow_mse(yy, flow).backward()   # HBM calibration kernels
probe = ops.EventProbe(); ops.PROBE = probe
for _ in range(int(os.environ.get("DMC_MB_ITERS", "10"))):
    m.zero_grad(); y = m.forward_mv_res(mv, res, True); y.backward(r)
    yy = y.detach().requires_grad_(True); ops.flow_mse(yy, flow).backward()
for k, (ms, n) in probe.summary().items():
    px = N * 224 * 224
    fl = 9108 if "fwd" in k else 2 * 7758
    print("%s: %.3f ms  (%.1f TFLOP/s)" % (k, ms, px * fl / ms / 1e9))
