"""EstimatorDenseNetTiny on the Winograd F(2x2, 3x3) ring kernel (csrc/gen_tiny.hip gen_wino_kernel; option gen_wino = bit K:
forward hidden layer K, bit 8 + K: data-gradient group K): the checks of test_hip_parity.py's generator tests with that path
switched on -- forward against the CPU oracle (code/dmcnet/model.py:172-194 restated in oracle/dmc_oracle.py) and the golden
vectors, parameter gradients, ragged shapes, distance from an fp64 evaluation next to the direct kernels', and bitwise
determinism."""
import copy

import numpy as np
import pytest
import torch

import dmcnet_amd
from tests.test_hip_parity import CASES, DEV, checksum, rel_err, rnd, tiny_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def layerwise_forward():
    """These tests select kernels of the layer-by-layer forward: the fused one-launch forward (option gen_fused, the default) off."""
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(b"gen_fused")
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_fused", 0), "dmc_set_option")
    yield
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_fused", before), "dmc_set_option")
FWD, BWD, ALL = 0x0F, 0x1F00, 0x1F0F


@pytest.fixture
def gen_wino(request):
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(b"gen_wino")
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_wino", int(getattr(request, "param", ALL))), "dmc_set_option")
    yield int(getattr(request, "param", ALL))
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_wino", before), "dmc_set_option")


@pytest.mark.parametrize("gen_wino", [0x1, 0x2, 0x4, 0x8, FWD, 0x100, 0x200, 0x400, 0x800, 0x1000, ALL], indirect=True)
@pytest.mark.parametrize("tag,shape,sd", CASES)
def test_generator_wino_vs_oracle_and_golden(golden, gen_wino, tag, shape, sd):
    g = golden("g1_generator")
    o, m = tiny_pair()
    x, r = rnd(sd, shape), rnd(sd + 50, (shape[0], 2) + shape[2:])
    yo = o(x)
    (yo * r).sum().backward()
    y = m(x.to(DEV))
    (y * r.to(DEV)).sum().backward()
    assert rel_err(y, yo) < 1e-5
    if tag == "frame":
        np.testing.assert_allclose(checksum(y.cpu()), g["frame_out_checksum"], rtol=1e-5)
        assert rel_err(y[0, :, 100:108, 0:16], g["frame_out_slice"]) < 1e-5
    else:
        assert rel_err(y, g[tag + "_out"]) < 1e-5
    for (k, po), (_, pm) in zip(o.named_parameters(), m.named_parameters()):
        assert rel_err(pm.grad, po.grad) < 1e-4, k


@pytest.mark.parametrize("shape", [(1, 5, 1, 1), (1, 5, 3, 5), (2, 5, 8, 64), (1, 5, 9, 68), (2, 5, 64, 260), (1, 5, 8, 4),
                                   (3, 5, 17, 220), (1, 5, 33, 224), (2, 5, 16, 64), (2, 5, 23, 100), (1, 5, 2, 224),
                                   (1, 5, 40, 228), (2, 5, 65, 128), (1, 5, 1, 224), (1, 5, 7, 112), (1, 5, 96, 96),
                                   (2, 5, 31, 188)])
def test_generator_wino_edge_shapes(gen_wino, shape):
    """Tiles that end inside the image (odd and even row counts, one row), widths below 224 (blocks of idle lanes), widths the
    path does not take (no multiple of 4, above 224, below 64: the other kernels run)."""
    o, m = tiny_pair(12)
    x = rnd(7, shape)
    yo = o(x) + x[:, :2]
    y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=True)
    assert rel_err(y, yo) < 1e-5
    r = rnd(8, tuple(yo.shape))
    (yo * r).sum().backward()
    (y * r.to(DEV)).sum().backward()
    for (k, po), (_, pm) in zip(o.named_parameters(), m.named_parameters()):
        assert rel_err(pm.grad, po.grad) < 1e-4, k


def test_generator_wino_vs_fp64_and_determinism():
    """Output and parameter gradients against an fp64 evaluation of the oracle, next to the direct kernels' distance from it
    (Winograd rounding: the transforms add and subtract values of like magnitude), and two runs bit-identical."""
    o, m = tiny_pair(13)
    o64 = copy.deepcopy(o).double()
    mv, res = rnd(1, (3, 2, 72, 224)), rnd(2, (3, 3, 72, 224))
    r = rnd(3, (3, 2, 72, 224))
    x64 = torch.cat([mv, res], 1).double()
    y64 = o64(x64) + mv.double()
    (y64 * r.double()).sum().backward()
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(b"gen_wino")
    outs, grads = {}, {}
    try:
        for mask in (0, ALL):
            dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_wino", mask), "dmc_set_option")
            outs[mask], grads[mask] = [], []
            for _ in range(2):
                m.zero_grad()
                y = m.forward_mv_res(mv.to(DEV), res.to(DEV), add_mv=True)
                (y * r.to(DEV)).sum().backward()
                outs[mask].append(y.detach().clone())
                grads[mask].append([p.grad.clone() for p in m.parameters()])
    finally:
        dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_wino", before), "dmc_set_option")
    assert torch.equal(outs[ALL][0], outs[ALL][1])
    for a, b in zip(grads[ALL][0], grads[ALL][1]):
        assert torch.equal(a, b)
    e_w, e_d = rel_err(outs[ALL][0], y64), rel_err(outs[0][0], y64)
    print("forward vs fp64: winograd %.3g, direct %.3g" % (e_w, e_d))
    assert e_w <= max(4 * e_d, 2e-6), (e_w, e_d)
    for (k, p64), gw, gd in zip(o64.named_parameters(), grads[ALL][0], grads[0][0]):
        ew, ed = rel_err(gw, p64.grad), rel_err(gd, p64.grad)
        assert ew <= max(4 * ed, 2e-5), (k, ew, ed)
