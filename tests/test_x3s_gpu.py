"""GPU tests of the pre-split bf16x3 convolution path (dmc-net_amd/csrc/conv_x3s.hip and its producers in bn_act.hip),
through the C ABI: slice tensors, the 3x3 / stride-1 convolution and both gradients against an fp64 evaluation of
F.conv2d (the arithmetic of the torchvision BasicBlock convolutions behind code/dmcnet/model.py:305,352), the slice-writing
BatchNorm / pool kernels bit for bit against their fp32 forms, and the fused conv -> bn op in this mode against the stock
modules.  Bars: 1e-5 relative on convolution results (fp32-accurate products, as the in-loop-split kernels), bit equality
wherever the same fp32 arithmetic is only stored differently."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dmcnet_amd
from dmcnet_amd import ops, resnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CL = torch.channels_last


def rnd(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def rel_err(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_split_merge_round_trip_is_exact():
    """s0 + s1 + s2 reproduces every finite fp32 value exactly (normal, tiny, huge, exact bf16 values); the only bit pattern
    that changes is -0.0, which comes back as +0.0 (its remainder slices are +0)."""
    rs = np.random.RandomState(3)
    v = rs.standard_normal((3, 64, 9, 7)).astype(np.float32)
    v *= np.exp(rs.uniform(-40, 40, v.shape)).astype(np.float32)
    v.flat[:8] = [0.0, -0.0, 1.0, -1.5, 3.0e38, -3.0e38, 1.0e-30, 2.0 ** -100]
    x = torch.from_numpy(v).to(DEV).contiguous(memory_format=CL)
    xs = ops.x3s_split(x)
    back = ops.x3s_merge(xs, tuple(x.shape))
    assert torch.equal(back, x)
    nz = x != 0
    assert torch.equal(back.view(torch.int32)[nz], x.view(torch.int32)[nz])
    assert xs.numel() == x.numel() * 6


X3S_CASES = [
    (2, 64, 56, 56, 64),         # layer1: 256-pixel tiles, tiles crossing image rows
    (3, 128, 28, 28, 128),       # layer2
    (2, 256, 14, 14, 256),       # layer3
    (5, 512, 7, 7, 512),         # layer4: whole padded images per weight-gradient step
    (3, 64, 28, 28, 128),        # Cin != Cout
    (1, 128, 14, 14, 64),
    (7, 64, 7, 7, 64),           # tiles spanning several images
    (2, 64, 10, 14, 64),         # H != W
    (2, 128, 30, 28, 64),
    (37, 64, 14, 14, 64),        # enough pixels for the 256-pixel configuration on 14 x 14 images
]


@pytest.mark.parametrize("case", X3S_CASES)
def test_x3s_conv_fwd_dgrad_wgrad_vs_fp64(case):
    n, cin, h, w, cout = case
    lib = dmcnet_amd._lib.load()
    assert lib.dmc_x3s_conv_supported(n, h, w, cin, cout) and lib.dmc_x3s_conv_wgrad_supported(n, h, w, cin, cout)
    x, wt = rnd(301, (n, cin, h, w)), rnd(302, (cout, cin, 3, 3)) * 0.1
    xo, wo = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yo = F.conv2d(xo, wo, None, 1, 1)
    go = rnd(303, tuple(yo.shape))
    (yo * go.double()).sum().backward()
    xg = x.to(DEV).contiguous(memory_format=CL)
    wg = wt.to(DEV).contiguous(memory_format=CL)
    gg = go.to(DEV).contiguous(memory_format=CL)
    xs, dys = ops.x3s_split(xg), ops.x3s_split(gg)
    wf, wtr = ops.x3s_pack_weights(wg)
    y, part = ops.x3s_conv_fwd(xs, wf, n, h, w, cin, cout, want_stats=True)
    dx = ops.x3s_conv_dgrad(dys, wtr, n, h, w, cin, cout)
    dw = ops.x3s_conv_wgrad(xs, dys, n, h, w, cin, cout)
    assert rel_err(y, yo) < 1e-5
    assert rel_err(dx, xo.grad) < 1e-5
    assert rel_err(dw, wo.grad) < 1e-5
    assert y.is_contiguous(memory_format=CL) and dw.is_contiguous(memory_format=CL)
    # BatchNorm partials of the forward's epilogue = the sums of the STORED outputs
    s = part.sum(0).cpu()
    yd = y.double().permute(1, 0, 2, 3).reshape(cout, -1).cpu()
    assert float((s[:, 0] - yd.sum(1)).abs().max()) <= 1e-9 * float(yd.abs().sum(1).max())
    assert float((s[:, 1] - (yd * yd).sum(1)).abs().max()) <= 1e-9 * float((yd * yd).sum(1).max())
    # the addend of the data gradient (the residual branch's gradient) is added exactly once, in fp32
    add = rnd(304, (n, cin, h, w)).to(DEV).contiguous(memory_format=CL)
    assert torch.equal(ops.x3s_conv_dgrad(dys, wtr, n, h, w, cin, cout, addend=add), dx + add)
    # deterministic
    y2, part2 = ops.x3s_conv_fwd(xs, wf, n, h, w, cin, cout, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(part, part2)
    assert torch.equal(dx, ops.x3s_conv_dgrad(dys, wtr, n, h, w, cin, cout))
    assert torch.equal(dw, ops.x3s_conv_wgrad(xs, dys, n, h, w, cin, cout))


def test_x3s_conv_unsupported_shapes_are_refused():
    lib = dmcnet_amd._lib.load()
    assert not lib.dmc_x3s_conv_supported(2, 14, 14, 48, 64)          # Cin % 64
    assert not lib.dmc_x3s_conv_wgrad_supported(2, 13, 13, 64, 64)    # width without a weight-gradient configuration
    xs = torch.zeros(16, dtype=torch.uint8, device=DEV)
    y = torch.zeros(16, device=DEV)
    assert lib.dmc_x3s_conv_fwd(dmcnet_amd._lib.ptr(xs), dmcnet_amd._lib.ptr(xs), dmcnet_amd._lib.ptr(y), None, 0, 2, 14, 14, 48, 64,
                                None) != 0


@pytest.mark.parametrize("m,c,relu,res", [(3 * 28 * 28, 128, 1, 1), (2 * 56 * 56, 64, 1, 0), (5 * 7 * 7, 512, 0, 1), (977, 64, 0, 0)])
def test_bn_producers_bit_identical_to_fp32_forms(m, c, relu, res):
    """dmc_bn_apply_act_x3s / dmc_bn_act_bwd_x3s: fp32 outputs, ReLU mask, dgamma / dbeta bit-identical to
    dmc_bn_apply_act_nhwc / dmc_bn_act_bwd, and the slice tensors are exactly the split of those fp32 outputs."""
    L, lib = dmcnet_amd._lib, dmcnet_amd._lib.load()
    st = ops._stream()
    x = rnd(311, (m, c)).to(DEV)
    r = rnd(312, (m, c)).to(DEV) if res else None
    gamma, beta = (rnd(313, (c,)) * 0.5 + 1).to(DEV), rnd(314, (c,)).to(DEV)
    stats = torch.cat([rnd(315, (c,)) * 0.1, rnd(316, (c,)).abs() + 0.5]).to(DEV)
    mask_a = torch.zeros(m * c // 4, dtype=torch.uint8, device=DEV) if (relu and res) else None
    mask_b = torch.zeros(m * c // 4, dtype=torch.uint8, device=DEV) if (relu and res) else None
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    ys = torch.empty(lib.dmc_x3s_slices_bytes(m, c), dtype=torch.uint8, device=DEV)
    L.check(lib.dmc_bn_apply_act_nhwc(L.ptr(x), L.ptr(r), L.ptr(gamma), L.ptr(beta), L.ptr(stats), L.ptr(ya), L.ptr(mask_a),
                                      m, c, relu, st), "a")
    L.check(lib.dmc_bn_apply_act_x3s(L.ptr(x), L.ptr(r), L.ptr(gamma), L.ptr(beta), L.ptr(stats), L.ptr(yb), L.ptr(ys),
                                     L.ptr(mask_b), m, c, relu, st), "b")
    assert torch.equal(ya, yb)
    if mask_a is not None:
        assert torch.equal(mask_a, mask_b)
    ref = torch.empty_like(ys)
    L.check(lib.dmc_x3s_split(L.ptr(ya), L.ptr(ref), m, c, st), "split")
    assert torch.equal(ys, ref)
    # slices only (no fp32 output)
    ys2 = torch.empty_like(ys)
    L.check(lib.dmc_bn_apply_act_x3s(L.ptr(x), L.ptr(r), L.ptr(gamma), L.ptr(beta), L.ptr(stats), None, L.ptr(ys2),
                                     L.ptr(mask_b), m, c, relu, st), "c")
    assert torch.equal(ys, ys2)
    # backward
    dout = rnd(317, (m, c)).to(DEV)
    scratch = ops._floats(lib.dmc_bn_act_scratch_bytes(c), DEV)
    outs = []
    for form in (0, 1):
        dx, dres = torch.empty_like(x), (torch.empty_like(x) if (res and relu) else None)
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        dxs = torch.empty_like(ys)
        rr = r if (relu and mask_a is None) else None
        if form == 0:
            L.check(lib.dmc_bn_act_bwd(L.ptr(x), L.ptr(rr), L.ptr(gamma), L.ptr(beta), L.ptr(stats), L.ptr(scratch), L.ptr(dout),
                                       L.ptr(dx), L.ptr(dres), L.ptr(dg), L.ptr(db), L.ptr(mask_a), m, c, relu, st), "d")
            L.check(lib.dmc_x3s_split(L.ptr(dx), L.ptr(dxs), m, c, st), "split")
        else:
            L.check(lib.dmc_bn_act_bwd_x3s(L.ptr(x), L.ptr(rr), L.ptr(gamma), L.ptr(beta), L.ptr(stats), L.ptr(scratch), L.ptr(dout),
                                           L.ptr(dx), L.ptr(dxs), L.ptr(dres), L.ptr(dg), L.ptr(db), L.ptr(mask_a), m, c, relu, st), "e")
        outs.append((dx, dres, dg, db, dxs))
    for a, b in zip(*outs):
        assert (a is None and b is None) or torch.equal(a, b)


@pytest.mark.parametrize("shape", [(3, 64, 112, 112), (2, 64, 17, 23), (2, 128, 8, 8)])
def test_pool_slices_match_fp32_form(shape):
    n, c, h, w = shape
    x = rnd(321, shape).to(DEV).contiguous(memory_format=CL)
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    bn2 = torch.nn.BatchNorm2d(c).to(DEV).train()
    y0 = ops.bn_relu_pool(x, bn)
    y1 = ops.bn_relu_pool(x, bn2, want_slices=True)
    assert torch.equal(y0, y1) and ops.x3s_of(y0) is None
    assert torch.equal(ops.x3s_of(y1), ops.x3s_split(y1))
    assert torch.equal(bn.running_var, bn2.running_var)


@pytest.mark.parametrize("shape", [(3, 64, 112, 112), (2, 64, 17, 23), (2, 128, 8, 8), (1, 64, 1, 5), (2, 64, 30, 31)])
def test_pool_argmax_record_backward(shape, monkeypatch):
    """The stem tail with the arg-max record (forward writes each window's arg-max position and the raw input there;
    the backward's BatchNorm sums stream over that pooled-size record, dmc_bn_relu_pool_bwd_arg) against the stock
    modules and against the recomputing backward (dmc_bn_relu_pool_bwd); quantised inputs: exact ties inside windows
    pin PyTorch's arg-max rule in the FORWARD's scan."""
    n, c, h, w = shape
    x = torch.round(rnd(81, shape) * 4) / 4
    go = rnd(82, (n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1))
    bn_o = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn_o.weight.copy_(rnd(83, (c,)) * 0.5 + 1.0)       # both signs of gamma: the maximum of relu(bn(x)) is not the maximum of x
        bn_o.weight[::3] *= -1
        bn_o.bias.copy_(rnd(84, (c,)) * 0.3)
    mp = torch.nn.MaxPool2d(3, 2, 1)
    xo = x.clone().double().requires_grad_(True)
    bn_d = torch.nn.BatchNorm2d(c).double()
    bn_d.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn_o.state_dict().items()})
    yo = mp(torch.relu(bn_d(xo)))
    (yo * go.double()).sum().backward()
    res = {}
    for rec in (True, False):
        monkeypatch.setattr(ops, "POOL_ARGMAX", rec)
        bn_m = torch.nn.BatchNorm2d(c)
        bn_m.load_state_dict(bn_o.state_dict())
        bn_m.to(DEV)
        xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_(True)
        y = ops.bn_relu_pool(xg, bn_m, want_slices=True)
        (y * go.to(DEV)).sum().backward()
        res[rec] = (y.detach(), xg.grad, bn_m.weight.grad, bn_m.bias.grad)
        assert rel_err(y, yo.float()) < 1e-5
        assert rel_err(xg.grad, xo.grad.float()) < 1e-4
        assert rel_err(bn_m.weight.grad, bn_d.weight.grad.float()) < 1e-4
        assert rel_err(bn_m.bias.grad, bn_d.bias.grad.float()) < 1e-4
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):      # same arg-max, same sums up to the summation order
        assert rel_err(a, b) < 1e-5
    # deterministic
    monkeypatch.setattr(ops, "POOL_ARGMAX", True)
    bn_m = torch.nn.BatchNorm2d(c)
    bn_m.load_state_dict(bn_o.state_dict())
    bn_m.to(DEV)
    xg = x.to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    (ops.bn_relu_pool(xg, bn_m, want_slices=True) * go.to(DEV)).sum().backward()
    assert torch.equal(xg.grad, res[True][1]) and torch.equal(bn_m.weight.grad, res[True][2])


@pytest.mark.parametrize("n,h,w", [(3, 224, 224), (2, 40, 36), (1, 8, 8), (2, 37, 44)])
def test_stem_statistics_from_the_convolution_epilogue(n, h, w, monkeypatch):
    """conv1 -> bn1 -> relu -> maxpool of the classifier: bn1's batch statistics reduced in conv1's epilogue
    (dmc_stem_fwd_x3_stats -> stat_split of dmc_bn_relu_pool_fwd_arg) against the separate statistics pass: the same
    convolution output (bitwise), statistics / pooled map / gradients equal to fp32 rounding, and against fp64."""
    x = rnd(401, (n, 2, h, w)).to(DEV)
    wt = (rnd(402, (64, 2, 7, 7)) * 0.1).to(DEV)
    go = rnd(403, (n, 64, ((h + 1) // 2 - 1) // 2 + 1, ((w + 1) // 2 - 1) // 2 + 1)).to(DEV)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "STEM_STATS", fused)
        bn = torch.nn.BatchNorm2d(64).to(DEV).train()
        wg = wt.clone().requires_grad_(True)
        y = ops.stem_conv(x, wg, want_stats=True)
        assert (getattr(y, "_dmc_stat_partials", None) is not None) == fused
        out = ops.bn_relu_pool(y, bn, want_slices=True)
        (out * go).sum().backward()
        res[fused] = (y.detach(), out.detach(), bn.running_mean.clone(), bn.running_var.clone(), wg.grad.clone(),
                      bn.weight.grad.clone(), bn.bias.grad.clone())
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert rel_err(a, b) < 2e-6
    yd = F.conv2d(x.double().cpu(), wt.double().cpu(), None, 2, 3)
    mean, var = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=True)
    assert rel_err(res[True][2].cpu().double(), 0.1 * mean) < 1e-5
    assert rel_err(res[True][3].cpu().double(), 0.9 + 0.1 * var) < 1e-5


def test_stem_statistics_with_a_large_mean_offset():
    """|mean| >> std in conv1's output (a constant-ish cue through weights with a common sign): the epilogue's statistics are
    pivoted sums (sum of v - K and (v - K)^2 around the lane's first value, the pivot removed in fp64), so the variance
    E[x^2] - E[x]^2 keeps its digits.  Checked against the fp64 statistics of the SAME stored output (isolates the reduction
    from the convolution's own rounding): plain fp32 sums of v^2 would lose ~|mean|^2 / var x 1e-6 here."""
    n, h, w = 6, 224, 224
    x = (4.0 + 0.02 * rnd(431, (n, 2, h, w))).to(DEV)
    wt = (0.05 + 0.01 * rnd(432, (64, 2, 7, 7))).to(DEV)
    bn = torch.nn.BatchNorm2d(64, momentum=1.0).to(DEV).train()       # momentum 1: the running statistics ARE the batch's
    y = ops.stem_conv(x, wt, want_stats=True)
    assert getattr(y, "_dmc_stat_partials", None) is not None
    ops.bn_relu_pool(y, bn)
    yd = y.detach().double()
    mean, var = yd.mean((0, 2, 3)), yd.var((0, 2, 3), unbiased=True)
    assert float((mean.abs() / var.sqrt()).min()) > 3.0                # the regime the test is about
    assert rel_err(bn.running_mean.double(), mean) < 1e-6
    assert float(((bn.running_var.double() - var).abs() / var).max()) < 5e-6


@pytest.mark.parametrize("cin,planes,hw,n,seed", [(64, 64, 14, 6, 331), (128, 128, 28, 3, 338), (64, 64, 56, 2, 342), (512, 512, 7, 5, 339)])
def test_basic_block_presplit_vs_in_loop_split_and_stock(cin, planes, hw, n, seed, monkeypatch):
    """An identity-shortcut BasicBlock in training mode: (a) pre-split path (conv1 reads the input's slices, writes ONLY
    slices for conv2; both convolutions, their gradients and the residual link on conv_x3s.hip), (b) the in-loop-split kernels,
    (c) the stock modules in fp64.  (a) and (b) use the same bf16x3 products in a different summation order: both within
    the usual bars of (c), BatchNorm statistics included.  The input seeds are chosen so that no ReLU pre-activation of the
    block lies within fp32 rounding of zero (asserted below on the fp64 evaluation): a branch that flips with the
    summation order moves a whole channel's dbeta by one pixel's gradient, in ANY fp32 implementation."""
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(7)
    unit = resnet.ResidualUnit("basic", cin, planes, 1).to(DEV).train()
    state = {k: v.clone() for k, v in unit.state_dict().items()}
    x0 = rnd(seed, (n, cin, hw, hw)).to(DEV).contiguous(memory_format=CL)
    go = rnd(332, (n, cin, hw, hw)).to(DEV).contiguous(memory_format=CL)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "X3S", mode)
        unit.load_state_dict(state)
        unit.zero_grad(set_to_none=True)
        x = (x0 * 1.0).requires_grad_(True)
        x.retain_grad()
        if mode:
            assert ops.x3s_usable(n, hw, hw, unit.conv1) and ops.x3s_usable(n, hw, hw, unit.conv2)
            ops._attach_x3s(x, ops.x3s_split(x.detach()))
        out = unit(x)
        (out * go).sum().backward()
        res[mode] = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in unit.parameters()] + \
                    [unit.bn1.running_var.clone(), unit.bn2.running_mean.clone()]
    monkeypatch.setattr(resnet, "OWN_CONV", False)
    ref = resnet.ResidualUnit("basic", cin, planes, 1).double().train()
    ref.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in state.items()})
    xr = x0.cpu().double().requires_grad_(True)
    with torch.no_grad():                          # conditioning of the test data: ReLU margins of the fp64 evaluation
        refc = resnet.ResidualUnit("basic", cin, planes, 1).double().train()
        refc.load_state_dict(ref.state_dict())
        p1 = refc.bn1(refc.conv1(xr))
        p2 = refc.bn2(refc.conv2(torch.relu(p1))) + xr
        assert float(p1.abs().min()) > 4e-6 and float(p2.abs().min()) > 4e-6, "pick another input seed"
    outr = ref(xr)
    (outr * go.cpu().double()).sum().backward()
    want = [outr, xr.grad] + [p.grad for p in ref.parameters()] + [ref.bn1.running_var, ref.bn2.running_mean]
    for mode in (True, False):
        assert rel_err(res[mode][0], want[0]) < 1e-5
        assert rel_err(res[mode][1], want[1]) < 2e-5
        for a, b in zip(res[mode][2:], want[2:]):
            assert rel_err(a, b) < 2e-4
    for a, b in zip(res[True], res[False]):
        assert rel_err(a, b) < 2e-4


def test_fp32_memory_of_a_slices_only_result_is_guarded(monkeypatch):
    """conv_bn_act(want_f32=False) does not write the fp32 result: a consumer that needs it must fail loudly."""
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(9)
    conv = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(DEV).to(memory_format=CL)
    bn = torch.nn.BatchNorm2d(64).to(DEV).train()
    x = rnd(341, (2, 64, 14, 14)).to(DEV).contiguous(memory_format=CL).requires_grad_(True)
    y = ops.conv_bn_act(x, conv, bn, want_f32=False, want_slices=True)
    assert not ops.f32_valid(y) and ops.x3s_of(y) is not None
    conv1x1 = torch.nn.Conv2d(64, 64, 1, 1, 0, bias=False).to(DEV).to(memory_format=CL)
    with pytest.raises(RuntimeError):
        ops.conv_bn_act(y, conv1x1, bn)
    z = ops.conv_bn_act(y, conv, bn)                 # a pre-split consumer is fine
    full = ops.conv_bn_act(x, conv, bn, want_f32=True, want_slices=True)
    assert torch.equal(ops.x3s_of(full), ops.x3s_of(y)) and torch.equal(ops.x3s_merge(ops.x3s_of(y), tuple(y.shape)), full)
    assert torch.isfinite(z).all()


@pytest.mark.parametrize("cin,cout,hw,n,seed", [(64, 128, 56, 2, 351), (128, 256, 28, 3, 357), (256, 512, 14, 5, 351)])
def test_stride2_unit_data_gradient_in_one_presplit_launch(cin, cout, hw, n, seed, monkeypatch):
    """layerN.0.conv1 (3x3, stride 2) -> bn -> relu: with DMC_X3S the data gradient is ONE dmc_x3s_conv_dgrad_s2 launch on
    the slices of dy (four input-parity classes, nine (tap, class) pairs) instead of four parity-class launches; forward and
    weight gradient stay on the in-loop-split kernels.  Both modes against the stock modules in fp64 (input seeds without
    ReLU near-ties, asserted)."""
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(cin, cout, 3, 2, 1, bias=False)
    bn = torch.nn.BatchNorm2d(cout)
    state = ({k: v.clone() for k, v in conv.state_dict().items()}, {k: v.clone() for k, v in bn.state_dict().items()})
    x0 = rnd(seed, (n, cin, hw, hw))
    go = rnd(361, (n, cout, hw // 2, hw // 2))
    cd, bd = torch.nn.Conv2d(cin, cout, 3, 2, 1, bias=False).double(), torch.nn.BatchNorm2d(cout).double().train()
    cd.load_state_dict({k: v.double() for k, v in state[0].items()})
    xr = x0.double().requires_grad_(True)
    pre = bd(cd(xr))
    assert float(pre.detach().abs().min()) > 4e-6, "pick another input seed"
    (torch.relu(pre) * go.double()).sum().backward()
    lib = dmcnet_amd._lib.load()
    assert lib.dmc_x3s_conv_dgrad_s2_supported(n, hw // 2, hw // 2, cin, cout)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "X3S", mode)
        cm = torch.nn.Conv2d(cin, cout, 3, 2, 1, bias=False).to(DEV).to(memory_format=CL)
        bm = torch.nn.BatchNorm2d(cout).to(DEV).train()
        cm.load_state_dict(state[0]); bm.load_state_dict(state[1])
        x = (x0.to(DEV).contiguous(memory_format=CL) * 1.0).requires_grad_(True)
        x.retain_grad()
        assert ops.conv_bn_act_supported(x, cm, bm)
        out = ops.conv_bn_act(x, cm, bm)
        (out * go.to(DEV)).sum().backward()
        res[mode] = (out.detach(), x.grad, cm.weight.grad, bm.weight.grad, bm.bias.grad)
    for mode in (True, False):
        assert rel_err(res[mode][0], torch.relu(pre)) < 1e-5
        assert rel_err(res[mode][1], xr.grad) < 2e-5
        assert rel_err(res[mode][2], cd.weight.grad) < 2e-5
        assert rel_err(res[mode][3], bd.weight.grad) < 2e-4 and rel_err(res[mode][4], bd.bias.grad) < 2e-4
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][2], res[False][2])   # same forward / weight-gradient kernels
    assert not torch.equal(res[True][1], res[False][1])          # the data gradient really took the other kernel


def test_bn_backward_sums_from_the_data_gradient_epilogue(monkeypatch):
    """ops.BnBwdLink: in layer1 (two identity-shortcut BasicBlocks) the BatchNorm-backward sums of three of the four
    conv -> bn units come out of the epilogue of the data-gradient launch that writes their output gradient
    (dmc_x3s_conv_dgrad_bnb: conv1 -> conv2 inside a block, and the first block's output through the second block's
    conv1 + residual-gradient addend); the last unit has no pre-split consumer and keeps its own reduction pass.  Same
    arithmetic in another summation order: every gradient within 1e-5 of the unlinked run."""
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(13)
    net = resnet.ResNet("basic", (2, 2, 2, 2)).to(DEV).train()
    layer = net.layer1
    state = {k: v.clone() for k, v in layer.state_dict().items()}
    x0 = rnd(371, (4, 64, 56, 56)).to(DEV).contiguous(memory_format=CL)
    go = rnd(372, (4, 64, 56, 56)).to(DEV).contiguous(memory_format=CL)
    links = []
    orig_init = ops.BnBwdLink.__init__
    monkeypatch.setattr(ops.BnBwdLink, "__init__", lambda self, kind: (orig_init(self, kind), links.append(self))[0])
    res = {}
    for linked in (True, False):
        monkeypatch.setattr(ops, "BN_BWD_LINK", linked)
        layer.load_state_dict(state)
        layer.zero_grad(set_to_none=True)
        links.clear()
        x = (x0 * 1.0).requires_grad_(True)
        x.retain_grad()
        ops._attach_x3s(x, ops.x3s_split(x.detach()))
        # the last unit of the stage normally feeds layer2's stride-2 convolution: no pre-split consumer here either
        (layer(x) * go).sum().backward()
        res[linked] = [x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
        if linked:
            assert [(l.kind, l.ready) for l in links] == [("inner", True), ("block", True), ("inner", True)], \
                [(l.kind, l.ready) for l in links]
        else:
            assert links == []
    for a, b in zip(res[True], res[False]):
        assert rel_err(a, b) < 1e-5


def test_weight_gradients_on_the_side_stream_are_transparent(monkeypatch):
    """ops.WGRAD_STREAM inside ``with ops.wgrad_side_stream()`` (as DmcnetTrainStep wraps its backward): the classifier's
    weight gradients run on a side HIP stream and the caller's stream re-joins when backward() returns.  A ResNet-18 training pass with and without it: every gradient bitwise equal, read IMMEDIATELY
    after backward() (no synchronize: the engine callback must already have ordered the streams), over three steps, and with
    gradient accumulation (second backward without zero_grad: the side stream must not be used when .grad exists)."""
    import copy
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(11)
    net = resnet.build("resnet18")
    net.conv1 = torch.nn.Conv2d(2, 64, 7, 2, 3, bias=False)
    net = net.to(DEV).to(memory_format=CL).train()
    ref = copy.deepcopy(net)
    for it in range(3):
        x = rnd(500 + it, (6, 2, 224, 224)).to(DEV)
        outs = []
        for model, side in ((net, True), (ref, False)):
            monkeypatch.setattr(ops, "WGRAD_STREAM", side)
            model.zero_grad(set_to_none=True)
            with ops.wgrad_side_stream():
                model(x).square().mean().backward()
                if it == 2:
                    model(x).square().mean().backward()      # accumulation into existing .grad
            outs.append([p.grad.clone() for p in model.parameters()])      # cloned on the caller's stream, no sync
        assert not ops._WGRAD_PENDING[0]
        for (name, _), a, b in zip(net.named_parameters(), *outs):
            assert torch.equal(a, b), (it, name)


@pytest.mark.parametrize("n,h,w", [(3, 224, 224), (2, 40, 36), (2, 37, 44), (1, 8, 8), (5, 112, 64), (1, 9, 255)])
def test_stem_data_gradient_kernel(n, h, w, monkeypatch):
    """conv1's data gradient (GAN variant: the classifier's loss reaches the generator, code/dmcnet_GAN/model.py:557-561)
    on dmc_stem_dgrad -- one wave per input row, row GEMM in bf16x3 arithmetic + 1-D fold -- against fp64 autograd
    (fp32-level bar) and against the library GEMM + col2im path it replaces; contiguous and channels_last weights;
    odd sizes (border windows); deterministic."""
    x = rnd(601, (n, 2, h, w))
    wt = rnd(602, (64, 2, 7, 7)) * 0.1
    go = rnd(603, (n, 64, (h + 1) // 2, (w + 1) // 2))
    xo = x.double().requires_grad_(True)
    (F.conv2d(xo, wt.double(), None, 2, 3) * go.double()).sum().backward()
    if w % 4 != 0:       # the stem op wants W % 4 == 0 (its weight gradient): call the C entry point directly
        L, lib = dmcnet_amd._lib, dmcnet_amd._lib.load()
        assert lib.dmc_stem_dgrad_supported(h, w)
        dy = go.to(DEV).contiguous(memory_format=CL)
        wg = wt.to(DEV)
        dx = torch.empty((n, 2, h, w), device=DEV)
        work = torch.empty(lib.dmc_stem_dgrad_workspace_bytes(), dtype=torch.uint8, device=DEV)
        so, si, sy, sx = wg.stride()
        L.check(lib.dmc_stem_dgrad(L.ptr(dy), L.ptr(wg), so, si, sy, sx, L.ptr(work), L.ptr(dx), n, h, w, L._P(0)), "dmc_stem_dgrad")
        torch.cuda.synchronize()
        assert rel_err(dx, xo.grad.float()) < 2e-6
        return
    res = {}
    for own in (True, False):
        monkeypatch.setattr(ops, "STEM_DGRAD_HIP", own)
        for cl in (False, True):
            wg = wt.to(DEV)
            if cl:
                wg = wg.contiguous(memory_format=CL)
            xg = x.to(DEV).requires_grad_(True)
            (ops.stem_conv(xg, wg) * go.to(DEV)).sum().backward()
            res[(own, cl)] = xg.grad
            assert rel_err(xg.grad, xo.grad.float()) < (2e-6 if own else 2e-5)
    assert torch.equal(res[(True, False)], res[(True, True)])
    assert rel_err(res[(True, False)], res[(False, False)]) < 2e-5
    monkeypatch.setattr(ops, "STEM_DGRAD_HIP", True)
    xg = x.to(DEV).requires_grad_(True)
    (ops.stem_conv(xg, wt.to(DEV)) * go.to(DEV)).sum().backward()
    assert torch.equal(xg.grad, res[(True, False)])
    assert not dmcnet_amd._lib.load().dmc_stem_dgrad_supported(224, 260)


def test_side_stream_weight_gradients_and_gradient_hooks(monkeypatch):
    """A hook that READS a gradient the moment it is accumulated (here: clones it, as the two-rank test's capture hooks do)
    runs on the main stream: parameters with such a hook must not have their weight gradient on the side stream; the
    gradient exchange's own hook (ddp.GradBucketReducer: it only counts, the bucket copy waits for the side stream) must
    keep it.  Values equal to the one-stream run either way."""
    import copy
    from dmcnet_amd import ddp
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(12)
    net = resnet.build("resnet18")
    net.conv1 = torch.nn.Conv2d(2, 64, 7, 2, 3, bias=False)
    net = net.to(DEV).to(memory_format=CL).train()
    x = rnd(510, (6, 2, 224, 224)).to(DEV)
    monkeypatch.setattr(ops, "WGRAD_STREAM", False)
    ref = copy.deepcopy(net)
    ref(x).square().mean().backward()
    want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    monkeypatch.setattr(ops, "WGRAD_STREAM", True)
    # (a) foreign hooks that read at accumulation time
    a = copy.deepcopy(net)
    seen = {}
    for k, p in a.named_parameters():
        p.register_post_accumulate_grad_hook(lambda q, k=k: seen.__setitem__(k, q.grad.clone()))
    n0 = ops._WGRAD_COUNT[0]
    with ops.wgrad_side_stream():
        a(x).square().mean().backward()
    assert ops._WGRAD_COUNT[0] == n0                     # no launch left the main stream
    for k in want:
        assert torch.equal(seen[k], want[k]), k
    # (b) the gradient exchange's hooks (single process: the bucket copies still run)
    b = copy.deepcopy(net)
    red = ddp.GradBucketReducer([("base_model", list(b.parameters()))])
    n0 = ops._WGRAD_COUNT[0]
    red.begin()
    with ops.wgrad_side_stream():
        b(x).square().mean().backward()
    red.finish()
    assert ops._WGRAD_COUNT[0] > n0                      # the side stream was used
    for k, p in b.named_parameters():
        assert torch.equal(p.grad, want[k]), k


def test_side_stream_with_a_weight_used_twice_in_one_backward(monkeypatch):
    """A convolution applied TWICE in one forward (shared weight): autograd sums the two weight gradients on the main stream;
    the first one may be on the side stream, so the second sighting re-joins and stays on the main stream.  Bitwise the
    one-stream result, and repeatedly so."""
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(13)
    conv = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(DEV).to(memory_format=CL)
    bn1, bn2 = torch.nn.BatchNorm2d(64).to(DEV).train(), torch.nn.BatchNorm2d(64).to(DEV).train()
    x = rnd(520, (24, 64, 56, 56)).to(DEV).contiguous(memory_format=CL)

    def run(side):
        monkeypatch.setattr(ops, "WGRAD_STREAM", side)
        for m in (conv, bn1, bn2):
            m.zero_grad(set_to_none=True)
        xin = x.clone().requires_grad_(True)
        y = resnet._conv_bn_act(conv, bn1, xin, next_conv=conv)
        z = resnet._conv_bn_act(conv, bn2, y)
        n0 = ops._WGRAD_COUNT[0]
        with ops.wgrad_side_stream():
            z.square().mean().backward()
        torch.cuda.synchronize()
        return conv.weight.grad.clone(), xin.grad.clone(), ops._WGRAD_COUNT[0] - n0

    w0, x0, n_main = run(False)
    assert n_main == 0
    for _ in range(3):
        w1, x1, n_side = run(True)
        assert n_side == 1                                   # the first sighting only
        assert torch.equal(w1, w0) and torch.equal(x1, x0)


def test_side_stream_with_a_weight_used_three_times_in_one_backward(monkeypatch):
    """The same convolution applied THREE times: the second sighting re-joins the streams (which forgets what was in flight);
    the third must still stay on the main stream -- its gradient is summed into the buffer the first two share.  Bitwise the
    one-stream result, repeatedly, and exactly one launch leaves the main stream."""
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(14)
    conv = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(DEV).to(memory_format=CL)
    bns = [torch.nn.BatchNorm2d(64).to(DEV).train() for _ in range(3)]
    x = rnd(521, (24, 64, 56, 56)).to(DEV).contiguous(memory_format=CL)

    def run(side):
        monkeypatch.setattr(ops, "WGRAD_STREAM", side)
        for m in [conv] + bns:
            m.zero_grad(set_to_none=True)
        xin = x.clone().requires_grad_(True)
        y = resnet._conv_bn_act(conv, bns[0], xin, next_conv=conv)
        y = resnet._conv_bn_act(conv, bns[1], y, next_conv=conv)
        z = resnet._conv_bn_act(conv, bns[2], y)
        n0 = ops._WGRAD_COUNT[0]
        with ops.wgrad_side_stream():
            z.square().mean().backward()
        torch.cuda.synchronize()
        return conv.weight.grad.clone(), xin.grad.clone(), ops._WGRAD_COUNT[0] - n0

    w0, x0, n_main = run(False)
    assert n_main == 0
    for _ in range(3):
        w1, x1, n_side = run(True)
        assert n_side == 1                                   # the first sighting only
        assert torch.equal(w1, w0) and torch.equal(x1, x0)
    assert not ops._WGRAD_MAIN                               # cleared when the pass ends


def test_side_stream_with_two_shared_weights_used_alternately(monkeypatch):
    """Two shared convolutions A and B applied A, B, A, B: backward meets B, A, B, A -- B's second use joins the streams
    mid-pass; A's first gradient went to the side stream BEFORE that join and A's second use comes after it.  The set of
    weights seen in this pass survives the join, so A's second gradient is produced on the main stream, where the engine
    sums the two.  Bitwise the one-stream result, repeatedly; exactly two launches (the first sighting of each) leave
    the main stream."""
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(15)
    ca = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(DEV).to(memory_format=CL)
    cb = torch.nn.Conv2d(64, 64, 3, 1, 1, bias=False).to(DEV).to(memory_format=CL)
    bns = [torch.nn.BatchNorm2d(64).to(DEV).train() for _ in range(4)]
    x = rnd(522, (24, 64, 56, 56)).to(DEV).contiguous(memory_format=CL)

    def run(side):
        monkeypatch.setattr(ops, "WGRAD_STREAM", side)
        for m in [ca, cb] + bns:
            m.zero_grad(set_to_none=True)
        xin = x.clone().requires_grad_(True)
        y = resnet._conv_bn_act(ca, bns[0], xin, next_conv=cb)
        y = resnet._conv_bn_act(cb, bns[1], y, next_conv=ca)
        y = resnet._conv_bn_act(ca, bns[2], y, next_conv=cb)
        z = resnet._conv_bn_act(cb, bns[3], y)
        n0 = ops._WGRAD_COUNT[0]
        with ops.wgrad_side_stream():
            z.square().mean().backward()
        torch.cuda.synchronize()
        return ca.weight.grad.clone(), cb.weight.grad.clone(), xin.grad.clone(), ops._WGRAD_COUNT[0] - n0

    a0, b0, x0, n_main = run(False)
    assert n_main == 0
    for _ in range(3):
        a1, b1, x1, n_side = run(True)
        assert n_side == 2
        assert torch.equal(a1, a0) and torch.equal(b1, b0) and torch.equal(x1, x0)
    assert not ops._WGRAD_MAIN and not ops._WGRAD_SEEN


@pytest.mark.parametrize("case", [(6, 64, 14, 14, 128), (37, 64, 14, 14, 64), (5, 128, 7, 7, 64), (42, 64, 14, 14, 512)])
def test_x3s_two_workgroups_per_cu_configuration(case):
    """Tile configuration 4 of the pre-split convolutions (128-pixel tiles, 4 waves, a patch of <= 224 pixels: 79,872 B of
    LDS, two workgroups per CU; chosen by default where a 14- / 7-wide map yields >= 512 workgroups, here also forced through
    option conv_cfg = 105) against the 8-wave configuration (conv_cfg = 106 excludes it): every output element sums the same
    products in the same order, so forward and data gradient are bitwise equal; the BatchNorm partials group pixels differently
    and agree to fp64 rounding.  The last case takes configuration 4 by the default rule."""
    n, cin, h, w, cout = case
    L, lib = dmcnet_amd._lib, dmcnet_amd._lib.load()
    x, wt, go = rnd(311, (n, cin, h, w)), rnd(312, (cout, cin, 3, 3)) * 0.1, rnd(313, (n, cout, h, w))
    xs = ops.x3s_split(x.to(DEV).contiguous(memory_format=CL))
    dys = ops.x3s_split(go.to(DEV).contiguous(memory_format=CL))
    wf, wtr = ops.x3s_pack_weights(wt.to(DEV).contiguous(memory_format=CL))
    before = lib.dmc_get_option(b"conv_cfg")
    out = {}
    try:
        for cfg in (105, 106, 0):
            L.check(lib.dmc_set_option(b"conv_cfg", cfg), "dmc_set_option")
            y, part = ops.x3s_conv_fwd(xs, wf, n, h, w, cin, cout, want_stats=True)
            dx = ops.x3s_conv_dgrad(dys, wtr, n, h, w, cin, cout)
            out[cfg] = (y, dx, part.sum(0))
    finally:
        L.check(lib.dmc_set_option(b"conv_cfg", before), "dmc_set_option")
    for cfg in (106, 0):
        assert torch.equal(out[105][0], out[cfg][0]) and torch.equal(out[105][1], out[cfg][1])
        assert float((out[105][2] - out[cfg][2]).abs().max()) <= 1e-12 * float(out[cfg][2].abs().max())
    assert lib.dmc_x3s_conv_stat_blocks(n, h, w, cout) == (n * h * w + 127) // 128 if case[0] == 42 else True
    yo = F.conv2d(x.double(), wt.double(), None, 1, 1)
    assert rel_err(out[105][0], yo) < 1e-5


def test_statistics_partials_sized_under_another_option_are_refused():
    """dmc_x3s_conv_stat_blocks / dmc_conv_nhwc_stat_blocks and the forward each read the tile-configuration option: a
    buffer sized before an option change must be REFUSED by the launch (it would be overrun), not written."""
    L, lib = dmcnet_amd._lib, dmcnet_amd._lib.load()
    n, cin, h, w, cout = 6, 64, 14, 14, 128
    x = rnd(411, (n, cin, h, w)).to(DEV).contiguous(memory_format=CL)
    xs = ops.x3s_split(x)
    wt = (rnd(412, (cout, cin, 3, 3)) * 0.1).to(DEV).contiguous(memory_format=CL)
    wf, _ = ops.x3s_pack_weights(wt)
    y = torch.empty((n, cout, h, w), device=DEV).contiguous(memory_format=CL)
    before = lib.dmc_get_option(b"conv_cfg")
    try:
        L.check(lib.dmc_set_option(b"conv_cfg", 106), "dmc_set_option")          # 8-wave tiles: 256 pixels per row of partials
        nblk = lib.dmc_x3s_conv_stat_blocks(n, h, w, cout)
        part = torch.empty((nblk, cout, 2), dtype=torch.float64, device=DEV)
        L.check(lib.dmc_set_option(b"conv_cfg", 105), "dmc_set_option")          # 128-pixel tiles: twice as many rows
        assert lib.dmc_x3s_conv_stat_blocks(n, h, w, cout) != nblk
        rc = lib.dmc_x3s_conv_fwd(L.ptr(xs), L.ptr(wf), L.ptr(y), L.ptr(part), nblk, n, h, w, cin, cout, None)
        assert rc != 0 and b"rows" in lib.dmc_last_error()
        L.check(lib.dmc_set_option(b"conv_cfg", 106), "dmc_set_option")
        L.check(lib.dmc_x3s_conv_fwd(L.ptr(xs), L.ptr(wf), L.ptr(y), L.ptr(part), nblk, n, h, w, cin, cout, None), "fwd")
        # the in-loop-split kernels: a wrong row count is refused as well
        nb2 = lib.dmc_conv_nhwc_stat_blocks(n, h, w, cin, cout, 3, 1, 1)
        p2 = torch.empty((nb2 + 1, cout, 2), dtype=torch.float64, device=DEV)
        wp = torch.empty(lib.dmc_conv_nhwc_wt_bytes(cin, cout, 3, 3), dtype=torch.uint8, device=DEV)
        rc = lib.dmc_conv_nhwc_fwd(L.ptr(x), L.ptr(wt), L.ptr(wp), None, None, L.ptr(y), L.ptr(p2), nb2 + 1, n, h, w, cin, cout,
                                   3, 3, 1, 1, 0, None)
        assert rc != 0 and b"rows" in lib.dmc_last_error()
        L.check(lib.dmc_conv_nhwc_fwd(L.ptr(x), L.ptr(wt), L.ptr(wp), None, None, L.ptr(y), L.ptr(p2), nb2, n, h, w, cin, cout,
                                      3, 3, 1, 1, 0, None), "conv_nhwc_fwd")
        torch.cuda.synchronize()
    finally:
        L.check(lib.dmc_set_option(b"conv_cfg", before), "dmc_set_option")
