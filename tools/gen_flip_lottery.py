#!/usr/bin/env python3
"""Generator gradients against fp64 over several data seeds and kernel selections, 120 frames (default) of 224 x 224, generic weights.
    python tools/gen_flip_lottery.py [--frames N] [seed ...]
Per seed and selection two numbers:
  * conditioned (what tests/test_hip_parity_full.py::test_generator_full_batch_generic_weights asserts, tests/gen_conditioned.py):
    worst ratio (device gradient's distance from the fp64 backward FORCED to the device run's own LeakyReLU branches) /
    (fp32 CPU oracle's distance from the fp64 backward forced to ITS branches), and the number of sign disagreements with the
    plain fp64 forward (device / oracle);
  * unconditioned (the round-4 metric): the same ratio against the PLAIN fp64 backward -- decided by single branch flips at
    pre-activations within rounding of zero: 1x .. 200x for every selection."""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from oracle import dmc_oracle as O
from tests.gen_conditioned import conditioned_report
DEV = "cuda:0"
lib = dmcnet_amd._lib.load()
# (name, gen_fused, gen_wino, gen_x3)
CONFIGS = [("fused forward, default backward", 1, 0x300, 2), ("fused forward, winograd every group", 1, 0x1F00, 2),
           ("layerwise direct fp32", 0, 0, 0), ("layerwise round-4 default", 0, 0x300, 2), ("layerwise winograd everywhere", 0, 0x1F0F, 0)]
args = sys.argv[1:]
frames = 120
if args and args[0] == "--frames":
    frames = int(args[1]); args = args[2:]
seeds = [int(a) for a in args] or [23, 24, 25, 26, 27]
print("# %d frames; per selection: conditioned ratio (flips device/oracle) | unconditioned ratio" % frames)
print("# selections:", " ; ".join(c[0] for c in CONFIGS))
for seed in seeds:
    o = O.seeded_state_fill(O.build_estimator("DenseNetTiny"), 15)
    m = dmcnet_amd.model.EstimatorDenseNetTiny(5); m.load_state_dict(o.state_dict()); m.to(DEV)
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((frames, 5, 224, 224)).astype(np.float32))
    r = torch.from_numpy(rs.standard_normal((frames, 2, 224, 224)).astype(np.float32))
    o64 = copy.deepcopy(o).double()
    for p in o64.parameters(): p.grad = None
    y64 = o64(x.double()) + x[:, :2].double(); (y64 * r.double()).sum().backward()
    plain = [p.grad.clone() for p in o64.parameters()]
    for p in o.parameters(): p.grad = None
    yo = o(x) + x[:, :2]; (yo * r).sum().backward()
    ref = [p.grad.clone() for p in o.parameters()]
    row = []
    for name, fused, wino, x3 in CONFIGS:
        lib.dmc_set_option(b"gen_fused", fused); lib.dmc_set_option(b"gen_wino", wino); lib.dmc_set_option(b"gen_x3", x3)
        m.zero_grad()
        y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=True)
        saved = y.grad_fn.saved_tensors[2].view(frames, 28, 224, 224)
        (y * r.to(DEV)).sum().backward()
        grads = [p.grad.clone() for p in m.parameters()]
        rep = conditioned_report(o, o64, x, r, y.detach(), grads, saved)
        cond = max(e_h / max(e_r, 1e-6) for e_h, e_r in rep["params"].values())
        unc = 0.0
        for gh, gr, g64 in zip(grads, ref, plain):
            s = float(g64.abs().max())
            unc = max(unc, (float((gh.double().cpu() - g64).abs().max()) / s) / max(float((gr.double() - g64).abs().max()) / s, 5e-6))
        row.append("%.2f (%d/%d) | %.1f" % (cond, rep["flips_hip"], rep["flips_ref"], unc))
    print("seed %d: %s" % (seed, " ; ".join(row)), flush=True)
