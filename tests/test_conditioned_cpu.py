"""The branch-conditioned gradient comparison (tests/gen_conditioned.py) on the CPU: with the fp32 oracle standing in for the
device run the report is symmetric, and a deliberately flipped LeakyReLU branch is COUNTED while the conditioned error stays at
rounding level (the unconditioned one does not).  code/dmcnet/model.py:111-119, :172-194."""
import copy

import numpy as np
import torch

from oracle import dmc_oracle as O
from tests.gen_conditioned import conditioned_report, forced_fp64, oracle_features


def test_conditioned_report_is_symmetric_and_counts_flips():
    o = O.seeded_state_fill(O.build_estimator("DenseNetTiny"), 15)
    o64 = copy.deepcopy(o).double()
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((2, 5, 24, 28)).astype(np.float32))
    r = torch.from_numpy(rs.standard_normal((2, 2, 24, 28)).astype(np.float32))
    yo = o(x) + x[:, :2]
    (yo * r).sum().backward()
    grads = [p.grad.clone() for p in o.parameters()]
    saved = torch.cat(oracle_features(o, x), 1)
    rep = conditioned_report(o, o64, x, r, yo.detach(), grads, saved)
    assert rep["flips_hip"] == rep["flips_ref"]
    for k, (e_hip, e_ref) in rep["params"].items():
        assert e_hip == e_ref and e_hip < 2e-6, (k, e_hip, e_ref)
    # flip the branches of 40 values of y3 in the "device" features: counted, and the forced fp64 backward follows them
    flipped = saved.clone()
    flipped[0, 22:24, 3:13, 7:9] *= -1.0
    rep2 = conditioned_report(o, o64, x, r, yo.detach(), grads, flipped)
    assert rep2["flips_hip"] == rep["flips_hip"] + 40
    m = [f > 0 for f in oracle_features(o64, x.double())]
    _, g_plain = forced_fp64(o64, x.double(), r.double(), m)
    worst_plain = max(float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(grads, g_plain))
    worst_forced = max(e for e, _ in rep2["params"].values())
    assert worst_plain < 2e-6 < worst_forced          # the run did NOT take those branches: forcing them moves the fp64 answer away
