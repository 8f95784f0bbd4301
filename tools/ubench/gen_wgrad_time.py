#!/usr/bin/env python3
"""Generator backward at the BASELINE size (120 frames of 224 x 224) on the tile weight-gradient kernel (gen_wgrad_path = 4) and the
row-sliding one (5): run under `rocprofv3 --kernel-trace --stats` and read gen_bwd_weight_pc_kernel / gen_wgrad_rs_kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dmcnet_amd  # noqa: E402

lib = dmcnet_amd._lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
m = dmcnet_amd.model.EstimatorDenseNetTiny(5).to("cuda:0")
g = torch.Generator().manual_seed(1)
mv, res = torch.randn(N, 2, 224, 224, generator=g).cuda(), torch.randn(N, 3, 224, 224, generator=g).cuda()
r = torch.randn(N, 2, 224, 224, generator=g).cuda()
for path in (4, 5, 4, 5):
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_wgrad_path", path), "set")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(6):
        if it == 1:
            ev[0].record()
        m.zero_grad()
        (m.forward_mv_res(mv, res, add_mv=True) * r).sum().backward()
    ev[1].record()
    torch.cuda.synchronize()
    print("path", path, "fwd+bwd ms", ev[0].elapsed_time(ev[1]) / 5, flush=True)
