"""GPU tests of the stride-2 block pair on space-to-depth slice tensors (conv_x3q.hip): the 3x3 / stride-2 / padding-1
convolution and the 1x1 / stride-2 shortcut convolution of torchvision's BasicBlock (behind code/dmcnet/model.py:305,352),
fused per direction.  Bars as for conv_x3s.hip: <= 1e-5 of an fp64 evaluation (bf16x3 is fp32-level arithmetic)."""
import pytest
import torch
import torch.nn.functional as F

import dmcnet_amd
from dmcnet_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CL = torch.channels_last


def rnd(seed, shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# (n, cin, h, w, cout): the classifier's three shapes at a few frames + ragged ones (tiles ending mid-row / mid-image,
# a tile count that is not a multiple of anything, a single image smaller than a tile)
SHAPES = [(6, 64, 56, 56, 128), (7, 128, 28, 28, 256), (9, 256, 14, 14, 512), (1, 64, 8, 8, 64), (3, 64, 20, 12, 64),
          (5, 128, 6, 10, 192), (2, 64, 2, 2, 64)]


@pytest.mark.parametrize("shape", SHAPES)
def test_s2d_slices_round_trip(shape):
    n, cin, h, w, _ = shape
    x = rnd(601, (n, cin, h, w)).to(DEV).contiguous(memory_format=CL)
    xq = ops.x3q_split(x)
    assert torch.equal(ops.x3q_merge(xq, x.shape), x)
    # class (py, px) of the s2d tensor is the ordinary slice tensor of x[:, :, py::2, px::2]
    nchunk = cin // 16
    mq = n * (h // 2) * (w // 2)
    q = xq.view(3, 4, nchunk * mq * 32)
    for py in (0, 1):
        for px in (0, 1):
            sub = x[:, :, py::2, px::2].contiguous(memory_format=CL)
            assert torch.equal(q[:, 2 * py + px].reshape(-1), ops.x3s_split(sub).view(3, -1).reshape(-1))


@pytest.mark.parametrize("shape", SHAPES)
def test_pair_forward_vs_fp64(shape):
    n, cin, h, w, cout = shape
    oh, ow = h // 2, w // 2
    assert dmcnet_amd._lib.load().dmc_x3q_supported(n, oh, ow, cin, cout)
    x, w3, w1 = rnd(611, (n, cin, h, w)), rnd(612, (cout, cin, 3, 3)) * 0.1, rnd(613, (cout, cin, 1, 1)) * 0.2
    xq = ops.x3q_split(x.to(DEV).contiguous(memory_format=CL))
    wf, _ = ops.x3q_pack_weights(w3.to(DEV).contiguous(memory_format=CL), w1.to(DEV).contiguous(memory_format=CL), transposed=False)
    y3, y1, p3, p1 = ops.x3q_conv_fwd(xq, wf, n, oh, ow, cin, cout, want_stats=True)
    r3 = F.conv2d(x.double(), w3.double(), None, 2, 1)
    r1 = F.conv2d(x.double(), w1.double(), None, 2, 0)
    assert rel_err(y3, r3) < 1e-5 and rel_err(y1, r1) < 1e-5, (rel_err(y3, r3), rel_err(y1, r1))
    # the statistics partials are those of the STORED values
    for y, p in ((y3, p3), (y1, p1)):
        s = p.sum(0).cpu()
        yd = y.double().cpu()
        assert rel_err(s[:, 0], yd.sum((0, 2, 3))) < 1e-12 + 1e-9 and rel_err(s[:, 1], (yd * yd).sum((0, 2, 3))) < 1e-12
    # run-to-run bitwise
    y3b, y1b, p3b, _ = ops.x3q_conv_fwd(xq, wf, n, oh, ow, cin, cout, want_stats=True)
    assert torch.equal(y3, y3b) and torch.equal(y1, y1b) and torch.equal(p3, p3b)


@pytest.mark.parametrize("shape", SHAPES)
def test_pair_data_gradient_vs_fp64(shape):
    n, cin, h, w, cout = shape
    oh, ow = h // 2, w // 2
    w3, w1 = rnd(622, (cout, cin, 3, 3)) * 0.1, rnd(623, (cout, cin, 1, 1)) * 0.2
    g3, g1 = rnd(624, (n, cout, oh, ow)), rnd(625, (n, cout, oh, ow))
    _, wt = ops.x3q_pack_weights(w3.to(DEV).contiguous(memory_format=CL), w1.to(DEV).contiguous(memory_format=CL), forward=False)
    dys3 = ops.x3s_split(g3.to(DEV).contiguous(memory_format=CL))
    dys1 = ops.x3s_split(g1.to(DEV).contiguous(memory_format=CL))
    dx = ops.x3q_conv_dgrad(dys3, dys1, wt, n, oh, ow, cin, cout)
    xr = torch.zeros((n, cin, h, w), dtype=torch.float64, requires_grad=True)
    (F.conv2d(xr, w3.double(), None, 2, 1) * g3.double()).sum().backward()
    ref = xr.grad.clone()
    xr.grad = None
    (F.conv2d(xr, w1.double(), None, 2, 0) * g1.double()).sum().backward()
    ref = ref + xr.grad
    assert rel_err(dx, ref) < 1e-5, rel_err(dx, ref)
    assert torch.equal(dx, ops.x3q_conv_dgrad(dys3, dys1, wt, n, oh, ow, cin, cout))


# the weight gradient walks whole padded rows of the classifier's three output grids
WG_SHAPES = [(6, 64, 56, 56, 128), (7, 128, 28, 28, 256), (9, 256, 14, 14, 512), (1, 64, 56, 56, 64), (2, 128, 14, 14, 64),
             (43, 64, 28, 28, 128)]


@pytest.mark.parametrize("shape", WG_SHAPES)
def test_pair_weight_gradient_vs_fp64(shape):
    n, cin, h, w, cout = shape
    oh, ow = h // 2, w // 2
    lib = dmcnet_amd._lib.load()
    assert lib.dmc_x3q_conv_wgrad_supported(n, oh, ow, cin, cout)
    x = rnd(631, (n, cin, h, w))
    g3, g1 = rnd(634, (n, cout, oh, ow)), rnd(635, (n, cout, oh, ow))
    xq = ops.x3q_split(x.to(DEV).contiguous(memory_format=CL))
    dys3 = ops.x3s_split(g3.to(DEV).contiguous(memory_format=CL))
    dys1 = ops.x3s_split(g1.to(DEV).contiguous(memory_format=CL))
    dw3, dw1 = ops.x3q_conv_wgrad(xq, dys3, dys1, n, oh, ow, cin, cout)
    w3 = torch.zeros((cout, cin, 3, 3), dtype=torch.float64, requires_grad=True)
    w1 = torch.zeros((cout, cin, 1, 1), dtype=torch.float64, requires_grad=True)
    (F.conv2d(x.double(), w3, None, 2, 1) * g3.double()).sum().backward()
    (F.conv2d(x.double(), w1, None, 2, 0) * g1.double()).sum().backward()
    assert rel_err(dw3, w3.grad) < 1e-5 and rel_err(dw1, w1.grad) < 1e-5, (rel_err(dw3, w3.grad), rel_err(dw1, w1.grad))
    dw3b, dw1b = ops.x3q_conv_wgrad(xq, dys3, dys1, n, oh, ow, cin, cout)
    assert torch.equal(dw3, dw3b) and torch.equal(dw1, dw1b)
    assert not lib.dmc_x3q_conv_wgrad_supported(n, 10, 10, cin, cout)      # a grid without a configuration


@pytest.mark.parametrize("cin,hw,n,seed", [(64, 56, 3, 655), (128, 28, 5, 651), (256, 14, 7, 680)])
def test_stride2_block_after_identity_block_pair_path_vs_separate_launches_and_fp64(cin, hw, n, seed, monkeypatch):
    """Two chained BasicBlocks as in layerN-1.1 -> layerN.0 (training mode): the identity block writes its result ONLY as a
    space-to-depth slice tensor, the stride-2 block runs conv1 + downsample as one launch per direction (ops.conv_bn_s2_pair).
    Against (b) the separate in-loop-split launches (DMC_X3Q=0) and (c) the stock modules in fp64: outputs 1e-5, input
    gradient 2e-5, parameter gradients / BatchNorm statistics 2e-4 (the BatchNorm chain's conditioning, as the other block
    tests).  Seeds chosen so that no ReLU pre-activation lies within fp32 rounding of zero."""
    from dmcnet_amd import resnet
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(11)
    u0 = resnet.ResidualUnit("basic", cin, cin, 1)
    u1 = resnet.ResidualUnit("basic", cin, 2 * cin, 2)
    u0.next_conv[0], u0.next_identity[0], u0.next_unit[0] = u1.conv1, False, u1
    net = torch.nn.Sequential(u0, u1).to(DEV).to(memory_format=CL).train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    x0 = rnd(seed, (n, cin, hw, hw)).to(DEV).contiguous(memory_format=CL)
    go = rnd(seed + 50, (n, 2 * cin, hw // 2, hw // 2)).to(DEV).contiguous(memory_format=CL)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "X3Q", mode)
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        x = (x0 * 1.0).requires_grad_(True)
        mid = u0(x)
        assert (ops.x3q_of(mid) is not None and not ops.f32_valid(mid)) == mode       # s2d slices only / fp32 as before
        out = u1(mid)
        (out * go).sum().backward()
        res[mode] = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()] + \
                    [u1.bn1.running_var.clone(), u1.downsample[1].running_mean.clone(), u1.downsample[1].running_var.clone()]
        assert int(u1.downsample[1].num_batches_tracked) == 1 and int(u1.bn1.num_batches_tracked) == 1
    monkeypatch.setattr(resnet, "OWN_CONV", False)
    r0, r1 = resnet.ResidualUnit("basic", cin, cin, 1), resnet.ResidualUnit("basic", cin, 2 * cin, 2)
    ref = torch.nn.Sequential(r0, r1).double().train()
    ref.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in state.items()})
    xr = x0.cpu().double().requires_grad_(True)
    with torch.no_grad():                          # conditioning of the test data: ReLU margins of the fp64 evaluation
        rc = torch.nn.Sequential(resnet.ResidualUnit("basic", cin, cin, 1), resnet.ResidualUnit("basic", cin, 2 * cin, 2)).double().train()
        rc.load_state_dict(ref.state_dict())
        c0, c1 = rc[0], rc[1]
        p1 = c0.bn1(c0.conv1(xr)); p2 = c0.bn2(c0.conv2(torch.relu(p1))) + xr
        mid64 = torch.relu(p2)
        q1 = c1.bn1(c1.conv1(mid64)); q2 = c1.bn2(c1.conv2(torch.relu(q1))) + c1.downsample(mid64)
        assert min(float(t.abs().min()) for t in (p1, p2, q1, q2)) > 4e-6, "pick another input seed"
    outr = ref(xr)
    (outr * go.cpu().double()).sum().backward()
    want = [outr, xr.grad] + [p.grad for p in ref.parameters()] + [r1.bn1.running_var, r1.downsample[1].running_mean,
                                                                   r1.downsample[1].running_var]
    for mode in (True, False):
        assert rel_err(res[mode][0], want[0]) < 1e-5, (mode, rel_err(res[mode][0], want[0]))
        assert rel_err(res[mode][1], want[1]) < 2e-5, (mode, rel_err(res[mode][1], want[1]))
        for i, (a, b) in enumerate(zip(res[mode][2:], want[2:])):
            assert rel_err(a, b) < 2e-4, (mode, i, rel_err(a, b))
    # bitwise repeatable
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    monkeypatch.setattr(ops, "X3Q", True)
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    x = (x0 * 1.0).requires_grad_(True)
    out = net(x)
    (out * go).sum().backward()
    assert torch.equal(out, res[True][0]) and torch.equal(x.grad, res[True][1])
    for p, g in zip(net.parameters(), res[True][2:]):
        assert torch.equal(p.grad, g)


def test_stride2_block_input_without_slices_is_split_on_the_fly(monkeypatch):
    """A stride-2 block whose input carries no s2d slices (a stand-alone call): the pair path splits it itself."""
    from dmcnet_amd import resnet
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(12)
    u1 = resnet.ResidualUnit("basic", 64, 128, 2).to(DEV).to(memory_format=CL).train()
    x = rnd(661, (4, 64, 28, 28)).to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    assert ops.s2_pair_usable(tuple(x.shape), u1)
    out = u1(x)
    out.square().mean().backward()
    ref = resnet.ResidualUnit("basic", 64, 128, 2).double().train()
    ref.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in u1.state_dict().items()})
    monkeypatch.setattr(resnet, "OWN_CONV", False)
    # (the statistics buffers were updated by the first call: reload the pre-call values is not needed for the output)
    xr = x.detach().cpu().double().requires_grad_(True)
    ref.bn1.reset_running_stats(); ref.bn2.reset_running_stats(); ref.downsample[1].reset_running_stats()
    outr = ref(xr)
    outr.square().mean().backward()
    assert rel_err(out, outr) < 1e-5 and rel_err(x.grad, xr.grad) < 2e-5
    # eval mode / no_grad: the pair path is a training op; the block falls back to the separate forward-only kernels
    u1.eval()
    with torch.no_grad():
        assert not ops.s2_pair_usable(tuple(x.shape), u1)
        ev = u1(x.detach())
    assert torch.isfinite(ev).all()


@pytest.mark.parametrize("shape", [(6, 64, 56, 56, 128), (9, 256, 14, 14, 512), (3, 64, 20, 12, 64), (2, 64, 2, 2, 64)])
def test_half_size_workgroups_are_bitwise_the_full_size_ones(shape):
    """The TN = 1 variant (128-pixel workgroups of 32 x 32 wave tiles, chosen where 256-pixel tiles would leave a partly
    filled round of workgroups: option conv_cfg 202 forces it, 201 the full-size one) and the 4-wave variant that runs two
    workgroups per CU (203) sum the same products in the same order: forward, both statistics and the data gradient are
    bitwise equal."""
    n, cin, h, w, cout = shape
    oh, ow = h // 2, w // 2
    L, lib = dmcnet_amd._lib, dmcnet_amd._lib.load()
    x, w3, w1 = rnd(671, (n, cin, h, w)), rnd(672, (cout, cin, 3, 3)) * 0.1, rnd(673, (cout, cin, 1, 1)) * 0.2
    g3, g1 = rnd(674, (n, cout, oh, ow)), rnd(675, (n, cout, oh, ow))
    xq = ops.x3q_split(x.to(DEV).contiguous(memory_format=CL))
    wf, wt = ops.x3q_pack_weights(w3.to(DEV).contiguous(memory_format=CL), w1.to(DEV).contiguous(memory_format=CL))
    dys3 = ops.x3s_split(g3.to(DEV).contiguous(memory_format=CL))
    dys1 = ops.x3s_split(g1.to(DEV).contiguous(memory_format=CL))
    before = lib.dmc_get_option(b"conv_cfg")
    out = {}
    try:
        for cfg in (201, 202, 203):
            L.check(lib.dmc_set_option(b"conv_cfg", cfg), "dmc_set_option")
            y3, y1, p3, p1 = ops.x3q_conv_fwd(xq, wf, n, oh, ow, cin, cout, want_stats=True)
            dx = ops.x3q_conv_dgrad(dys3, dys1, wt, n, oh, ow, cin, cout)
            out[cfg] = (y3, y1, dx, p3.sum(0), p1.sum(0), p3.shape[0])
    finally:
        L.check(lib.dmc_set_option(b"conv_cfg", before), "dmc_set_option")
    a, b, c = out[201], out[202], out[203]      # 203: the 4-wave, two-workgroups-per-CU variant (128-pixel tiles, two patch buffers)
    for o in (b, c):
        assert torch.equal(a[0], o[0]) and torch.equal(a[1], o[1]) and torch.equal(a[2], o[2])
        for i in (3, 4):
            assert float((a[i] - o[i]).abs().max()) <= 1e-12 * float(a[i].abs().max())
    assert b[5] == (n * oh * ow + 127) // 128 and a[5] == (n * oh * ow + 255) // 256 and c[5] == b[5]
