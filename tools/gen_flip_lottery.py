#!/usr/bin/env python3
"""The metric of tests/test_hip_parity_full.py::test_generator_full_batch_generic_weights over several seeds and kernel selections:
worst ratio (HIP gradient's distance from an fp64 evaluation) / (fp32 CPU oracle's distance), 120 frames of 224 x 224, generic weights.
    python tools/gen_flip_lottery.py [seed ...]
The ratio is decided by single LeakyReLU branch flips at pre-activations within rounding of zero."""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from oracle import dmc_oracle as O
DEV = "cuda:0"
lib = dmcnet_amd._lib.load()
CONFIGS = [("direct fp32", 0, 0), ("gen_x3 layer 1 (default fwd)", 0, 2), ("winograd layer 1", 2, 0), ("winograd every layer", 0x1F0F, 0),
           ("default (x3 layer 1, winograd groups 0/1)", 0x300, 2)]
seeds = [int(a) for a in sys.argv[1:]] or [23, 24, 25]
print("# worst gradient ratio vs fp64 (parameter), per seed; columns:", " | ".join(c[0] for c in CONFIGS))
for seed in seeds:
    o = O.seeded_state_fill(O.build_estimator("DenseNetTiny"), 15)
    m = dmcnet_amd.model.EstimatorDenseNetTiny(5); m.load_state_dict(o.state_dict()); m.to(DEV)
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((120, 5, 224, 224)).astype(np.float32))
    r = torch.from_numpy(rs.standard_normal((120, 2, 224, 224)).astype(np.float32))
    yo = o(x) + x[:, :2]; (yo * r).sum().backward()
    o64 = copy.deepcopy(o).double()
    for p in o64.parameters(): p.grad = None
    y64 = o64(x.double()) + x[:, :2].double(); (y64 * r.double()).sum().backward()
    row = []
    for name, wino, x3 in CONFIGS:
        lib.dmc_set_option(b"gen_wino", wino); lib.dmc_set_option(b"gen_x3", x3)
        m.zero_grad()
        y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=True)
        (y * r.to(DEV)).sum().backward()
        worst = (0.0, "")
        for (k, po), (_, pm), (_, p64) in zip(o.named_parameters(), m.named_parameters(), o64.named_parameters()):
            scale = float(p64.grad.abs().max())
            e_hip = float((pm.grad.double().cpu() - p64.grad).abs().max()) / scale
            e_ref = float((po.grad.double() - p64.grad).abs().max()) / scale
            worst = max(worst, (e_hip / max(e_ref, 5e-6), k))
        row.append("%.2f (%s)" % (worst[0], worst[1].replace("conv_", "c").replace(".0.", ".")))
    print("seed %d: %s" % (seed, " | ".join(row)), flush=True)
