"""EstimatorDenseNetTiny hidden layers on the bf16x3 16x16x32 kernel (csrc/gen_x3.hip; option gen_x3 = bit mask of layers):
the checks of test_hip_parity.py's generator tests with that path switched on -- forward against the CPU oracle
(code/dmcnet/model.py:172-194 restated in oracle/dmc_oracle.py) and the golden vectors, parameter gradients through the
unchanged backward kernels (which read the features this path writes), ragged shapes, and bitwise determinism."""
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


@pytest.fixture
def gen_x3(request):
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(b"gen_x3")
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_x3", int(getattr(request, "param", 7))), "dmc_set_option")
    yield int(getattr(request, "param", 7))
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_x3", before), "dmc_set_option")


@pytest.mark.parametrize("gen_x3", [1, 2, 4, 7], indirect=True)
@pytest.mark.parametrize("tag,shape,sd", CASES)
def test_generator_x3_layers_vs_oracle_and_golden(golden, gen_x3, tag, shape, sd):
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


@pytest.mark.parametrize("shape", [(1, 5, 1, 1), (1, 5, 3, 5), (2, 5, 8, 32), (1, 5, 9, 33), (2, 5, 64, 260), (1, 5, 8, 4),
                                   (3, 5, 17, 220), (1, 5, 33, 224), (2, 5, 16, 64), (2, 5, 23, 100), (1, 5, 2, 224),
                                   (1, 5, 40, 228), (2, 5, 65, 31), (1, 5, 32, 32), (1, 5, 96, 95)])
def test_generator_x3_edge_shapes(gen_x3, shape):
    """Strips that end inside the image (rows and columns), images smaller than one strip, widths that are no multiple of 4
    (the other layers then take the VALU kernels), row counts on both sides of the 32-row strip."""
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


def test_generator_x3_hidden_features_vs_fp64_and_determinism(gen_x3):
    """The features the path writes (y0, y1, y2: what the backward kernels read) against an fp64 evaluation of the oracle's
    layers -- no worse than the fp32 kernels' -- and two runs bit-identical."""
    import copy
    o, m = tiny_pair(13)
    o64 = copy.deepcopy(o).double()
    mv, res = rnd(1, (3, 2, 72, 224)), rnd(2, (3, 3, 72, 224))
    lib = dmcnet_amd._lib.load()
    outs = {}
    for mask in (0, 7):
        dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_x3", mask), "dmc_set_option")
        with torch.no_grad():
            outs[mask] = [m.forward_mv_res(mv.to(DEV), res.to(DEV), add_mv=True).clone() for _ in range(2)]
    assert torch.equal(outs[7][0], outs[7][1])
    x64 = torch.cat([mv, res], 1).double()
    y64 = o64(x64) + mv.double()
    e_x3 = rel_err(outs[7][0], y64)
    e_f32 = rel_err(outs[0][0], y64)
    assert e_x3 <= max(2 * e_f32, 2e-6), (e_x3, e_f32)
