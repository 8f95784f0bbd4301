"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs
and against the golden vectors the reference produced.  fp32 tolerance per BASELINE.json:
1e-4 relative on logits and losses (summation order differs from the CPU's)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dmcnet_amd
from dmcnet_amd import ops
from oracle import dmc_oracle as O
from tests.golden.make_golden import checksum

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(seed, shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def rel_err(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def tiny_pair(seed=11):
    o = O.seeded_state_fill(O.build_estimator("DenseNetTiny"), seed)
    m = dmcnet_amd.model.EstimatorDenseNetTiny(5)
    m.load_state_dict(o.state_dict())
    return o, m.to(DEV)


CASES = [("small", (2, 5, 40, 40), 101), ("ragged", (3, 5, 19, 37), 102), ("frame", (1, 5, 224, 224), 103)]


@pytest.mark.parametrize("tag,shape,sd", CASES)
def test_generator_forward_backward_vs_oracle_and_golden(golden, tag, shape, sd):
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
        assert rel_err(pm.grad, g["%s_grad_%s" % (tag, k)]) < 1e-4, k


# W % 4 == 0 and W <= 224 take the matrix-core kernels (push, gather, fused layers 4+5, producer /
# consumer weight gradient): tiles that end inside the image, images shorter than one tile, a
# single 4-pixel column, widths where the 64-pixel segments straddle rows differently; the other
# shapes take the VALU fallback kernels
@pytest.mark.parametrize("shape", [(1, 5, 1, 1), (1, 5, 3, 5), (2, 5, 8, 32), (1, 5, 9, 33), (2, 5, 64, 260),
                                   (1, 5, 8, 4), (2, 5, 9, 8), (3, 5, 17, 220), (1, 5, 33, 224), (2, 5, 16, 64),
                                   (1, 5, 7, 12), (2, 5, 23, 100), (1, 5, 2, 224), (1, 5, 40, 228)])
@pytest.mark.parametrize("delta", [False, True])
def test_generator_edge_shapes(shape, delta):
    o, m = tiny_pair(12)
    x = rnd(7, shape)
    yo = o(x) + (x[:, :2] if delta else 0)
    y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=delta)
    assert rel_err(y, yo) < 1e-5
    r = rnd(8, tuple(yo.shape))
    (yo * r).sum().backward()
    (y * r.to(DEV)).sum().backward()
    for (k, po), (_, pm) in zip(o.named_parameters(), m.named_parameters()):
        assert rel_err(pm.grad, po.grad) < 1e-4, k


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_generator_full_batch_vs_oracle(sign):
    """BASELINE size, 120 frames of 224x224 (3,360 tiles: every persistent workgroup walks ~13
    tiles through its LDS ring, which the smaller cases never do): forward and all twelve parameter
    gradients against the CPU oracle.
    With generic weights this comparison is ill-conditioned at this size: among 170 million hidden
    pre-activations some lie within rounding of zero, a different fp32 summation order flips their
    LeakyReLU branch, and each flip moves a gradient entry by ~1e-3 of its size (measured: every
    implementation, including the independent VALU kernels, sits 2e-4 .. 1.5e-3 from an fp64
    evaluation).  The test therefore pins the branches: small hidden weights and biases of +1 (all
    slopes 1) or -1 (all slopes 0.1) -- every tile, halo row, channel mapping and reduction is
    still exercised, and the result must match an fp64 evaluation as well as the fp32 oracle does."""
    import copy
    o, m = tiny_pair(14)
    with torch.no_grad():
        for name, p in o.named_parameters():
            if name.startswith("conv_"):
                if name.endswith("weight"):
                    p.mul_(0.02)
                else:
                    p.fill_(sign)
    m.load_state_dict(o.state_dict())
    x = rnd(21, (120, 5, 224, 224))
    r = rnd(22, (120, 2, 224, 224))
    yo = o(x) + x[:, :2]
    (yo * r).sum().backward()
    o64 = copy.deepcopy(o).double()
    for p in o64.parameters():
        p.grad = None
    ((o64(x.double()) + x[:, :2].double()) * r.double()).sum().backward()
    y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=True)
    (y * r.to(DEV)).sum().backward()
    assert rel_err(y, yo) < 1e-5
    for (k, po), (_, pm), (_, p64) in zip(o.named_parameters(), m.named_parameters(), o64.named_parameters()):
        scale = float(p64.grad.abs().max())
        e_hip = float((pm.grad.double().cpu() - p64.grad).abs().max()) / scale
        e_ref = float((po.grad.double() - p64.grad).abs().max()) / scale
        assert e_hip <= max(4 * e_ref, 2e-5), (k, e_hip, e_ref)


def test_generator_linearity_in_last_layer_and_determinism():
    """Size-independent properties at the full 224x224 size: the output is affine in
    predict_flow's parameters, and two runs are bit-identical (fixed-order reductions)."""
    _, m = tiny_pair(13)
    mv, res = rnd(1, (4, 2, 224, 224)).to(DEV), rnd(2, (4, 3, 224, 224)).to(DEV)
    with torch.no_grad():
        y1 = m.forward_mv_res(mv, res)
        b0 = m.predict_flow.bias.clone()
        m.predict_flow.bias.add_(torch.tensor([0.5, -0.25], device=DEV))
        y2 = m.forward_mv_res(mv, res)
        m.predict_flow.bias.copy_(b0)
        y3 = m.forward_mv_res(mv, res)
    assert torch.equal(y1, y3)
    d = (y2 - y1)
    assert float((d[:, 0] - 0.5).abs().max()) < 1e-5 and float((d[:, 1] + 0.25).abs().max()) < 1e-5
    gs = []
    for _ in range(2):
        m.zero_grad()
        m.forward_mv_res(mv, res).square().sum().backward()
        gs.append([p.grad.clone() for p in m.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*gs))


@pytest.mark.parametrize("numel", [1, 7, 4096, 2 * 2 * 224 * 224 + 3])
def test_flow_mse(numel):
    a, b = rnd(3, (numel,)), rnd(4, (numel,))
    ao = a.clone().requires_grad_(True)
    lo = F.mse_loss(ao, b) * 10.0
    lo.backward()
    ag = a.to(DEV).requires_grad_(True)
    l = ops.flow_mse(ag, b.to(DEV)) * 10.0
    l.backward()
    assert rel_err(l, lo) < 1e-5
    assert rel_err(ag.grad, ao.grad) < 1e-5


@pytest.mark.parametrize("shape", [(6, 224, 224), (3, 40, 40), (2, 19, 37), (120, 224, 224)])
def test_generator_fused_flow_mse_equals_separate_ops(shape):
    """ops.gen_tiny_mse (loss reduced in the forward's last kernel) against gen_tiny + flow_mse: identical
    gen_flow, the same loss to fp32 rounding of a different (both fixed) summation order, the same parameter
    gradients for loss-only, output-only and mixed upstream gradients; W % 4 != 0 takes the internal fallback."""
    n, h, w = shape
    torch.manual_seed(5)
    gen = dmcnet_amd.model.EstimatorDenseNetTiny(5).to(DEV)
    mv, res, flow = rnd(301, (n, 2, h, w)).to(DEV), rnd(302, (n, 3, h, w)).to(DEV), rnd(303, (n, 2, h, w)).to(DEV)
    r = rnd(304, (n, 2, h, w)).to(DEV)
    params = list(gen.parameters())
    for lw, ow in ((10.0, 0.0), (0.0, 1.0), (10.0, 0.5)):
        gen.zero_grad()
        y0 = gen.forward_mv_res(mv, res, True)
        l0 = ops.flow_mse(y0, flow)
        (lw * l0 + ow * (y0 * r).sum()).backward()
        g0 = [p.grad.clone() for p in params]
        gen.zero_grad()
        y1, l1 = gen.forward_mv_res_mse(mv, res, flow, True)
        (lw * l1 + ow * (y1 * r).sum()).backward()
        assert torch.equal(y0, y1)
        assert abs(float(l0) - float(l1)) <= 2e-6 * abs(float(l0))
        for a, b in zip([p.grad for p in params], g0):
            assert rel_err(a, b) < 1e-6
    y2, l2 = gen.forward_mv_res_mse(mv, res, flow, True)
    assert torch.equal(y1, y2) and float(l1) == float(l2)          # deterministic


@pytest.mark.parametrize("B,S,C", [(40, 3, 51), (2, 3, 51), (1, 25, 101), (5, 1, 2), (17, 3, 400)])
def test_consensus_ce(B, S, C):
    x = rnd(5, (B * S, C)) * 3
    t = torch.from_numpy(np.random.RandomState(6).randint(0, C, B)).long()
    xo = x.clone().requires_grad_(True)
    co = O.consensus(xo, S)
    lo = F.cross_entropy(co, t)
    (lo * 0.7).backward()
    xg = x.to(DEV).requires_grad_(True)
    l, c = ops.consensus_ce(xg, t.to(DEV), S)
    (l * 0.7).backward()
    assert rel_err(l, lo) < 1e-5 and rel_err(c, co) < 1e-6
    assert rel_err(xg.grad, xo.grad) < 1e-5


@pytest.mark.parametrize("arch", ["Discriminator3", "Discriminator", "Discriminator4"])
def test_discriminator_train_mode_vs_oracle(golden, arch):
    o = O.seeded_state_fill(O.OracleDiscriminator(arch), seed=31).train()
    m = dmcnet_amd.model._DISCRIMINATORS[arch](2)
    m.load_state_dict(o.state_dict())
    m.to(DEV).train()
    xin = rnd(32, (4, 2, 224, 224))
    masks = O.seeded_dropout_masks(33, o, 4)
    o.forced_masks, m.forced_masks = masks, masks
    xo = xin.clone().requires_grad_(True)
    vo = o(xo)
    tgt = torch.tensor([0, 0, 1, 1])
    F.cross_entropy(vo, tgt).backward()
    xg = xin.to(DEV).requires_grad_(True)
    ops.DEBUG_DISC_Z = []
    try:
        v = m(xg)
        zs = [z.detach().cpu() for z in ops.DEBUG_DISC_Z]
    finally:
        ops.DEBUG_DISC_Z = None
    F.cross_entropy(v, tgt.to(DEV)).backward()
    assert rel_err(v, vo) < 1e-4
    so, sm = o.state_dict(), m.state_dict()
    for k in so:
        assert rel_err(sm[k].float(), so[k].float()) < 1e-4, k
    # Gradients.  Two things make a plain comparison with the fp32 CPU oracle meaningless here:
    # (1) dy - mean(dy) - zhat*mean(dy*zhat) of the BatchNorm(eps=0.8) backward cancels, and its sums are
    #     over up to 50k values: fp32 reductions (torch's GPU kernels, any fp32 code) lose digits;
    # (2) LeakyReLU'(pre) is discontinuous: each block has ~10 pre-activations within 1e-5 of zero and
    #     typically one within 1e-8 -- closer than fp32 rounding of the convolution -- whose branch an
    #     fp32 evaluation takes at random; one flipped pixel moves the (cancelling) bias / weight sums by
    #     percents.  Both CPU and GPU fp32 paths show it, at different blocks.
    # So the reference is an fp64 evaluation of the same graph WITH THE BRANCH PATTERN OF THE RUN UNDER TEST
    # (the z tensors of the HIP blocks give it): against that, the HIP gradients must be fp32-accurate.
    o64 = O.seeded_state_fill(O.OracleDiscriminator(arch), seed=31).double().train()
    x64 = xin.double().requires_grad_(True)
    pinned = len(zs) == len(o64.names)
    if pinned:
        cur = x64
        for name, zk in zip(o64.names, zs):
            blk = getattr(o64, name)
            slope = torch.where(zk.double() > 0, 1.0, 0.2)
            cur = blk[0](cur) * slope * masks[name].double()[:, :, None, None]
            if len(blk) == 4:
                cur = blk[3](cur)
        v64 = o64.adv_layer(cur.reshape(cur.shape[0], -1))
    else:                       # blocks on the stock-op path (Discriminator4): no pattern to pin
        o64.forced_masks = {k: v_.double() for k, v_ in masks.items()}
        v64 = o64(x64)
    F.cross_entropy(v64, tgt).backward()
    po, pm, p64 = dict(o.named_parameters()), dict(m.named_parameters()), dict(o64.named_parameters())
    worst = 0.0
    for k in list(po) + ["input"]:
        t64 = x64.grad if k == "input" else p64[k].grad
        e_hip = rel_err(xg.grad if k == "input" else pm[k].grad, t64)
        e_cpu = rel_err(xo.grad if k == "input" else po[k].grad, t64)
        worst = max(worst, e_hip)
        if pinned:
            assert e_hip < 5e-5, (k, e_hip)
        else:
            assert e_hip <= max(4 * e_cpu, 1e-1), (k, e_hip, e_cpu)
    print("worst HIP-vs-fp64 gradient error %.1e (branch pattern pinned: %s)" % (worst, pinned))
    if arch == "Discriminator3":
        g = golden("g3_disc_train")
        assert rel_err(v, g["validity"]) < 1e-4


def test_discriminator_eval_mode_vs_golden(golden):
    g = golden("g2_model_eval")
    xd = rnd(27, (2, 2, 224, 224)).to(DEV)
    for arch in O.DISC_PLANS:
        o = O.seeded_state_fill(O.OracleDiscriminator(arch), seed=28)
        m = dmcnet_amd.model._DISCRIMINATORS[arch](2)
        m.load_state_dict(o.state_dict())
        m.to(DEV).eval()
        with torch.no_grad():
            assert rel_err(m(xd), g["disc_" + arch]) < 1e-4, arch


def _product(gan, seed):
    o = O.OracleModel(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1,
                      arch_estimator="DenseNetTiny", arch_d="Discriminator3" if gan else None)
    O.seeded_state_fill(o, seed)
    m = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1,
                         arch_estimator="DenseNetTiny", arch_d="Discriminator3" if gan else None)
    m.load_state_dict(o.state_dict())
    return o, m.to(DEV)


def test_model_forward_eval_vs_golden(golden):
    g = golden("g2_model_eval")
    flow, mv, res, _ = O.synthetic_batch(seed=21, batch=2, num_segments=3, num_class=51, flow_ds_factor=16)
    _, m = _product(False, 22)
    m.eval()
    with torch.no_grad():
        logits, gen_flow = m(mv.to(DEV), res.to(DEV))
    assert rel_err(logits, g["dmcnet_logits"]) < 1e-4
    np.testing.assert_allclose(checksum(gen_flow.cpu()), g["dmcnet_genflow_checksum"], rtol=1e-5)
    assert rel_err(gen_flow[:, :, 64:72, 200:224], g["dmcnet_genflow_slice"]) < 1e-5
    _, mg = _product(True, 23)
    mg.eval()
    with torch.no_grad():
        lo, va, gf = mg(mv.to(DEV), res.to(DEV), flow.to(DEV))
        _, va2, _ = mg(mv.to(DEV), res.to(DEV))
    assert rel_err(lo, g["gan_logits"]) < 1e-4
    assert rel_err(va, g["gan_validity_fake_real"]) < 1e-4
    assert rel_err(va2, g["gan_validity_fake"]) < 1e-4


@pytest.mark.parametrize("shape,use_bn", [((4, 16, 112, 112), True), ((3, 5, 7, 9), True),
                                          ((4, 16, 112, 112), False), ((12, 128, 14, 14), True)])
def test_disc_tail_unit(shape, use_bn):
    """LeakyReLU(0.2) -> keep-mask -> BatchNorm2d(eps=0.8) against the stock modules."""
    n, c = shape[:2]
    x = rnd(41, shape)
    keep = (torch.from_numpy(np.random.RandomState(42).rand(n, c)) < 0.75).float() / 0.75
    r = rnd(43, shape)
    bn_o = torch.nn.BatchNorm2d(c, 0.8)
    O.seeded_state_fill(bn_o, 44)
    bn_m = torch.nn.BatchNorm2d(c, 0.8)
    bn_m.load_state_dict(bn_o.state_dict())
    bn_m.to(DEV)
    xo = x.clone().requires_grad_(True)
    zo = F.leaky_relu(xo, 0.2) * keep[:, :, None, None]
    yo = bn_o(zo) if use_bn else zo
    (yo * r).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    y = ops.disc_tail(xg, keep.to(DEV), bn_m if use_bn else None, True)
    (y * r.to(DEV)).sum().backward()
    assert rel_err(y, yo) < 1e-5
    assert rel_err(xg.grad, xo.grad) < 1e-4
    if use_bn:
        assert rel_err(bn_m.weight.grad, bn_o.weight.grad) < 1e-4
        assert rel_err(bn_m.bias.grad, bn_o.bias.grad) < 1e-4
        assert rel_err(bn_m.running_mean, bn_o.running_mean) < 1e-5
        assert rel_err(bn_m.running_var, bn_o.running_var) < 1e-5
        assert int(bn_m.num_batches_tracked) == 1
        # eval mode uses the running statistics
        bn_o.eval(); bn_m.eval()
        with torch.no_grad():
            ye = ops.disc_tail(x.to(DEV), None, bn_m, False)
            assert rel_err(ye, bn_o(F.leaky_relu(x, 0.2))) < 1e-5


# ------------------------------------------------------------------ whole training steps
from tests.golden.make_golden import HP, WATCH, WATCH_D   # noqa: E402
from dmcnet_amd import train as T                          # noqa: E402


def _watch(model, keys):
    sd = model.state_dict()
    return {k: (sd[k] if sd[k].numel() <= 4096 else sd[k].reshape(-1)[:4096]) for k in keys}


@pytest.fixture
def conv_mode(request, monkeypatch):
    """Classifier convolution path for one test: "miopen", "own_f32" (this package's kernels, fp32 MFMA) or
    "own_x3" (the same in bf16x3 arithmetic)."""
    from dmcnet_amd import resnet
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(b"conv_arith")
    monkeypatch.setattr(resnet, "OWN_CONV", request.param != "miopen")
    dmcnet_amd._lib.check(lib.dmc_set_option(b"conv_arith", int(request.param == "own_x3")), "dmc_set_option")
    yield request.param
    dmcnet_amd._lib.check(lib.dmc_set_option(b"conv_arith", before), "dmc_set_option")


@pytest.mark.parametrize("conv_mode", ["miopen", "own_f32", "own_x3"], indirect=True)
@pytest.mark.parametrize("tag,freeze", [("dmcnet", False), ("dmcnet_frozen", True)])
def test_dmcnet_train_step_vs_reference_golden(golden, tag, freeze, conv_mode):
    """One iteration of the reference's own train() (golden G4) reproduced by the HIP path:
    losses / consensus logits within 1e-4 relative, post-step weights close.  conv_mode: the classifier's
    3x3 / 1x1 convolutions on MIOpen or on this package's matrix-core kernels (fused conv -> bn op) in
    either arithmetic."""
    g = golden("g4_train_steps")
    _, m = _product(False, 41)
    m.train()
    batch = O.synthetic_batch(seed=42, batch=2, num_segments=3, num_class=51, flow_ds_factor=16)
    step = T.DmcnetTrainStep(m, 3, HP["lr_cls"], HP["lr_mse"], HP["lr"], HP["weight_decay"],
                             HP["lr_cls_mult"], HP["lr_mse_mult"])
    T.adjust_learning_rate(step.optimizer_cls, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"],
                           freeze=True, epoch_thre=1 if freeze else 0)
    T.adjust_learning_rate(step.optimizer_gf, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"])
    r = step.step(tuple(t.to(DEV) for t in batch), freeze=freeze)
    for k in ("loss", "loss_cls", "loss_mse", "output"):
        assert rel_err(r[k], g["%s_%s" % (tag, k)]) < 1e-4, k
    np.testing.assert_allclose(checksum(r["gen_flow"].cpu()), g[tag + "_genflow_checksum"], rtol=1e-5)
    for k, v in _watch(m, WATCH).items():
        # Adam normalises the update: a weight moves by ~lr whatever its gradient's size, so the
        # comparison is absolute, at a small fraction of the step (lr * lr_mult <= 1e-2)
        ref = torch.as_tensor(g["%s_post_%s" % (tag, k)])
        assert float((v.cpu() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max())), k


def test_gan_train_steps_vs_reference_golden(golden):
    """D step then G step of the reference's own train() (golden G4).  Losses, logits and
    validity within 2e-4 relative.  Post-step weights: the gradients that reach them pass the
    ill-conditioned BatchNorm(eps=0.8) chain (see the discriminator test) and Adam(eps=1e-3)
    normalises small gradients, so the criterion is accuracy against an fp64 run of the same two
    steps: the HIP path may be at most 4x further from it than the reference's fp32 CPU result
    (the golden) is, or within 2e-5 absolute."""
    g = golden("g4_train_steps")
    o, m = _product(True, 43)
    m.train()
    o64 = o.double().train()
    b0 = O.synthetic_batch(seed=44, batch=2, num_segments=3, num_class=51)
    b1 = O.synthetic_batch(seed=45, batch=2, num_segments=3, num_class=51)
    md = O.seeded_dropout_masks(46, o.discriminator, 12)
    mg = O.seeded_dropout_masks(47, o.discriminator, 6)
    step = T.GanTrainStep(m, 3, HP["lr_cls"], HP["lr_adv_g"], HP["lr_adv_d"], HP["lr_mse"], HP["lr"],
                          HP["weight_decay"], HP["lr_cls_mult"], HP["lr_mse_mult"], HP["lr_d_mult"])
    opts64 = O.make_optimizers(o64, HP["lr"], HP["weight_decay"], HP["lr_cls_mult"], HP["lr_mse_mult"],
                               HP["lr_d_mult"])
    for o_ in opts64:
        O.adjust_learning_rate(o_, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"])
    for i, (b, masks, tag) in enumerate(((b0, md, "gan_D"), (b1, mg, "gan_G"))):
        m.discriminator.forced_masks = masks
        o64.discriminator.forced_masks = {k: v.double() for k, v in masks.items()}
        r = step.step(tuple(t.to(DEV) for t in b), i)
        b64 = tuple(t.double() if t.is_floating_point() else t for t in b)
        O.gan_train_step(o64, opts64[0], opts64[1], opts64[2], b64, i, 3, HP["lr_cls"], HP["lr_adv_g"],
                         HP["lr_adv_d"], HP["lr_mse"])
        for k in ("loss", "loss_cls", "loss_adv", "output", "validity") + (("loss_mse",) if i else ()):
            e = rel_err(r[k], g["%s_%s" % (tag, k)])
            # D step: forward of identical weights -> the 2e-4 bar of the north star.  G step: its forward
            # runs on the weights the D step just updated; Adam(eps 1e-3) turns the (legitimately
            # branch-dependent, see test_discriminator_train_mode_vs_oracle) differences of the
            # discriminator's gradients into weight differences of up to ~1e-4, which the G step's
            # discriminator output inherits at the 1e-3 level.
            assert e < (2e-4 if (i == 0 or k in ("loss_mse",)) else 5e-3), (tag, k, e)
        w64 = _watch(o64, WATCH + WATCH_D)
        for k, v in _watch(m, WATCH + WATCH_D).items():
            ref32 = torch.as_tensor(g["%s_post_%s" % (tag, k)]).double()
            e_hip = float((v.double().cpu() - w64[k]).abs().max())
            e_ref = float((ref32 - w64[k]).abs().max())
            assert e_hip <= max(4 * e_ref, 2e-5), (tag, k, e_hip, e_ref)


def test_reducer_single_rank_nccl_is_transparent():
    """world_size 1 over RCCL: the bucketed path must leave gradients untouched."""
    import os
    import torch.distributed as dist
    from dmcnet_amd import ddp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        _, m = _product(False, 41)
        m.train()
        batch = tuple(t.to(DEV) for t in O.synthetic_batch(seed=42, batch=2, num_segments=3, num_class=51))
        kw = dict(lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
        a = T.DmcnetTrainStep(m, 3, 1.0, 10.0, **kw)
        ra = a.step(batch)
        _, m2 = _product(False, 41)
        m2.train()
        b = T.DmcnetTrainStep(m2, 3, 1.0, 10.0, reducer=ddp.for_model(m2), **kw)
        rb = b.step(batch)
        assert rel_err(ra["loss"], rb["loss"]) < 1e-5
        # With the package's own convolutions (the default) every kernel of the step has a fixed reduction order:
        # two model instances stepping on the same batch must end BITWISE equal, with or without the reducer.
        # Only the stock path (DMC_OWN_CONV=0: MIOpen's NHWC weight gradients split K with atomics and its
        # algorithm choice may differ between model instances) is allowed last-bit noise, bounded by one
        # Adam step = lr * lr_mult = 1e-4.
        from dmcnet_amd import resnet
        for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
            if resnet.OWN_CONV or k.startswith("gen_flow_model"):
                assert torch.equal(p, q), k
            else:
                assert float((p - q).abs().max()) < 1e-4, k
        if resnet.OWN_CONV:
            assert torch.equal(ra["output"], rb["output"]) and torch.equal(ra["loss"], rb["loss"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", [(3, 64, 112, 112), (2, 64, 17, 23), (2, 128, 8, 8), (1, 64, 1, 5), (2, 64, 30, 31)])
def test_bn_relu_pool_unit(shape):
    """Fused BatchNorm + ReLU + MaxPool(3,2,1) of the stem against the stock modules, forward and
    backward; inputs contain exact ties (zeros after ReLU, duplicated maxima) to pin the arg-max rule."""
    n, c, h, w = shape
    x, go = rnd(81, shape), rnd(82, (n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1))
    x = torch.round(x * 4) / 4                       # quantised: many equal values inside windows
    bn_o = torch.nn.BatchNorm2d(c)
    O.seeded_state_fill(bn_o, 83)
    bn_m = torch.nn.BatchNorm2d(c)
    bn_m.load_state_dict(bn_o.state_dict())
    bn_m.to(DEV)
    mp = torch.nn.MaxPool2d(3, 2, 1)
    xo = x.clone().requires_grad_(True)
    yo = mp(torch.relu(bn_o(xo)))
    (yo * go).sum().backward()
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert ops.bn_relu_pool_supported(xg)
    y = ops.bn_relu_pool(xg, bn_m)
    (y * go.to(DEV)).sum().backward()
    assert y.shape == yo.shape
    assert rel_err(y, yo) < 1e-5
    assert rel_err(xg.grad, xo.grad) < 1e-4
    assert rel_err(bn_m.weight.grad, bn_o.weight.grad) < 1e-4
    assert rel_err(bn_m.bias.grad, bn_o.bias.grad) < 1e-4
    assert rel_err(bn_m.running_mean, bn_o.running_mean) < 1e-5
    assert rel_err(bn_m.running_var, bn_o.running_var) < 1e-5
    assert int(bn_m.num_batches_tracked) == 1
    g1 = xg.grad.clone()
    xg.grad = None
    (ops.bn_relu_pool(xg, bn_m) * go.to(DEV)).sum().backward()
    assert torch.equal(g1, xg.grad)                  # deterministic
    bn_m.load_state_dict(bn_o.state_dict())          # (the second training pass moved the running stats)
    bn_o.eval(); bn_m.eval()
    with torch.no_grad():
        assert rel_err(ops.bn_relu_pool(xg, bn_m), mp(torch.relu(bn_o(x)))) < 1e-5


@pytest.mark.parametrize("n,h,w", [(3, 224, 224), (2, 40, 36), (2, 37, 44), (1, 8, 8), (5, 112, 64)])
def test_stem_wgrad_unit(n, h, w):
    """conv1 (2->64, 7x7, stride 2, pad 3) weight gradient through dmc_stem_wgrad against torch's CPU
    autograd (fp64 accumulate, the oracle's arithmetic); the forward is the same convolution."""
    x, wt = rnd(71, (n, 2, h, w)), rnd(72, (64, 2, 7, 7)) * 0.1
    oh, ow = (h + 1) // 2, (w + 1) // 2
    go = rnd(73, (n, 64, oh, ow))
    wo = wt.double().requires_grad_(True)
    yo = F.conv2d(x.double(), wo, None, 2, 3)
    (yo * go.double()).sum().backward()
    for cl in (False, True):
        wg = wt.to(DEV)
        if cl:
            wg = wg.contiguous(memory_format=torch.channels_last)
        wg.requires_grad_(True)
        xg = x.to(DEV)
        assert ops.stem_conv_supported(xg, wg)
        y = ops.stem_conv(xg, wg)
        (y * go.to(DEV)).sum().backward()
        assert rel_err(y, yo.float()) < 1e-5
        assert wg.grad.shape == wg.shape
        assert rel_err(wg.grad, wo.grad.float()) < 2e-5
        # deterministic: a second evaluation is bit-identical
        g1 = wg.grad.clone()
        wg.grad = None
        (ops.stem_conv(xg, wg) * go.to(DEV)).sum().backward()
        assert torch.equal(g1, wg.grad)
    # data gradient (GAN variant): GEMM + fold against autograd
    xo = x.double().requires_grad_(True)
    (F.conv2d(xo, wt.double(), None, 2, 3) * go.double()).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    wg2 = wt.to(DEV)                                   # no weight gradient wanted: only dx is computed
    assert ops.stem_conv_supported(xg, wg2)
    (ops.stem_conv(xg, wg2) * go.to(DEV)).sum().backward()
    assert rel_err(xg.grad, xo.grad.float()) < 2e-5
    assert not ops.stem_conv_supported(rnd(74, (1, 2, 10, 10)).to(DEV), wg)                 # W % 4 != 0


def test_stem_kernels_full_size_properties():
    """BASELINE size (120 frames): size-independent properties of the stem kernels.
    conv1 weight gradient: linear in dy and additive over a split of the batch; fused
    bn1+ReLU+maxpool: agrees with the stock modules run on the same device, and its input gradient
    sums to ~0 per channel (a BatchNorm backward property: sum_x dx = 0)."""
    n, h, w = 120, 224, 224
    g = torch.Generator(device=DEV).manual_seed(91)
    x = torch.randn((n, 2, h, w), device=DEV, generator=g)
    wt = (torch.randn((64, 2, 7, 7), device=DEV, generator=g) * 0.1).contiguous(memory_format=torch.channels_last)
    go = torch.randn((n, 64, 112, 112), device=DEV, generator=g).contiguous(memory_format=torch.channels_last)

    def wgrad(xs, gs, scale=1.0):
        wp = wt.clone().requires_grad_(True)
        (ops.stem_conv(xs, wp) * (gs * scale)).sum().backward()
        return wp.grad
    full = wgrad(x, go)
    assert rel_err(wgrad(x, go, 2.0), 2.0 * full) < 1e-6
    halves = wgrad(x[:60], go[:60]) + wgrad(x[60:], go[60:])
    assert rel_err(halves, full) < 2e-5
    assert torch.equal(full, wgrad(x, go))                                    # deterministic

    act = torch.randn((n, 64, 112, 112), device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    gp = torch.randn((n, 64, 56, 56), device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    bn_a, bn_b = torch.nn.BatchNorm2d(64).to(DEV), torch.nn.BatchNorm2d(64).to(DEV)
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5, generator=g); bn_a.bias.uniform_(-0.5, 0.5, generator=g)
    bn_b.load_state_dict(bn_a.state_dict())
    xa = act.clone().requires_grad_(True)
    ya = ops.bn_relu_pool(xa, bn_a)
    (ya * gp).sum().backward()
    xb = act.clone().requires_grad_(True)
    yb = torch.nn.functional.max_pool2d(torch.relu(bn_b(xb)), 3, 2, 1)
    (yb * gp).sum().backward()
    assert rel_err(ya, yb) < 1e-5
    assert rel_err(xa.grad, xb.grad) < 1e-4
    assert rel_err(bn_a.weight.grad, bn_b.weight.grad) < 1e-4 and rel_err(bn_a.bias.grad, bn_b.bias.grad) < 1e-4
    per_channel = xa.grad.sum(dim=(0, 2, 3)).abs().max() / xa.grad.abs().sum(dim=(0, 2, 3)).max()
    assert float(per_channel) < 1e-5


@pytest.mark.parametrize("shape,res,relu", [((4, 64, 56, 56), False, True), ((4, 64, 56, 56), True, True),
                                             ((3, 128, 7, 9), True, True), ((2, 512, 7, 7), False, False),
                                             ((6, 256, 14, 14), True, False)])
def test_bn_act_unit(shape, res, relu):
    """Fused NHWC BatchNorm(+residual)(+ReLU) against the stock modules, forward and backward."""
    c = shape[1]
    x, r, go = rnd(51, shape), rnd(52, shape), rnd(53, shape)
    bn_o = torch.nn.BatchNorm2d(c)
    O.seeded_state_fill(bn_o, 54)
    bn_m = torch.nn.BatchNorm2d(c)
    bn_m.load_state_dict(bn_o.state_dict())
    bn_m.to(DEV)
    xo, ro = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    yo = bn_o(xo) + (ro if res else 0)
    yo = torch.relu(yo) if relu else yo
    (yo * go).sum().backward()
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = r.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert ops.bn_act_supported(xg)
    y = ops.bn_act(xg, bn_m, rg if res else None, relu)
    (y * go.to(DEV)).sum().backward()
    assert rel_err(y, yo) < 1e-5
    assert rel_err(xg.grad, xo.grad) < 1e-4
    if res:
        assert rel_err(rg.grad, ro.grad) < 1e-5
    assert rel_err(bn_m.weight.grad, bn_o.weight.grad) < 1e-4
    assert rel_err(bn_m.bias.grad, bn_o.bias.grad) < 1e-4
    assert rel_err(bn_m.running_mean, bn_o.running_mean) < 1e-5
    assert rel_err(bn_m.running_var, bn_o.running_var) < 1e-5
    assert int(bn_m.num_batches_tracked) == 1
    bn_o.eval(); bn_m.eval()
    with torch.no_grad():
        ye = ops.bn_act(xg, bn_m, rg if res else None, relu)
        yr = bn_o(x) + (r if res else 0)
        assert rel_err(ye, torch.relu(yr) if relu else yr) < 1e-5


def test_eval_path_25_segments_vs_oracle():
    """test.py-style scoring: 25 segments per video, consensus mean, eval mode."""
    from dmcnet_amd import evaluate
    o, m = _product(False, 61)
    o.eval(); m.eval()
    flow, mv, res, _ = O.synthetic_batch(seed=62, batch=1, num_segments=25, num_class=51)
    with torch.no_grad():
        ref = o(mv, res)[0].view(1, 25, 51).mean(1).numpy()
    got = evaluate.forward_video(m, mv.to(DEV), res.to(DEV), 25, 1)
    assert got.shape == (1, 51) and rel_err(torch.from_numpy(got), ref) < 1e-4
    loader = [(flow, mv, res, torch.tensor([7]))]
    out, acc = evaluate.evaluate(m, loader, 25, 1, DEV)
    assert len(out) == 1 and out[0][1] == 7 and acc in (0.0, 100.0)


def test_eval_path_ten_crops_vs_oracle():
    """The ``--test-crops 10`` leg of test.py (code/dmcnet/test.py:96-99,139-150): frames through the 10-crop transform
    (transforms.GroupOverSample: 5 offsets x (crop, mirrored crop)), 3 segments x 10 crops per video scored and averaged."""
    from dmcnet_amd import dataset, evaluate, transforms
    o, m = _product(False, 65)
    o.eval(); m.eval()
    rs = np.random.RandomState(66)
    over = transforms.GroupOverSample(224, None)
    mvs, ress, flows = [], [], []
    for _ in range(2):                                            # two videos, 3 segments each, decoded at 256 x 256
        clip = dataset.synthetic_clip_u8(rs, 3, size=256)                              # [S,7,H,W] uint8
        frames = over([np.transpose(f, (1, 2, 0)) for f in clip])                    # 30 HWC crops, segment order inside each offset
        assert len(frames) == 30
        flow, mv, res = dataset.to_tensors(np.transpose(np.array(frames), (0, 3, 1, 2)), 0)
        mvs.append(mv); ress.append(res); flows.append(flow)
    mv, res = torch.stack(mvs), torch.stack(ress)                # [2, 30, C, 224, 224]
    with torch.no_grad():
        ref = o(mv, res)[0].view(2, 30, 51).mean(1).numpy()
    got = evaluate.forward_video(m, mv.to(DEV), res.to(DEV), 3, 10)
    assert got.shape == (2, 51) and rel_err(torch.from_numpy(got), ref) < 1e-4
    out, acc = evaluate.evaluate(m, [(torch.stack(flows), mv, res, torch.tensor([4, 9]))], 3, 10, DEV)
    assert len(out) == 2 and [x[1] for x in out] == [4, 9] and 0.0 <= acc <= 100.0


def test_eval_and_validate_run_no_stock_convolution(monkeypatch):
    """Forward-only paths (evaluate.forward_video = code/dmcnet/test.py:139-198, driver.validate = validate() of
    code/dmcnet/train.py): every convolution of the classifier runs on this package's kernels with the BatchNorm's running
    statistics -- nn.Conv2d's stock forward (MIOpen) is never entered -- and the scores equal the oracle's."""
    from dmcnet_amd import driver, evaluate, resnet
    if not resnet.OWN_CONV:
        pytest.skip("stock convolutions requested")
    o, m = _product(False, 63)
    o.eval(); m.eval()
    flow, mv, res, tgt = O.synthetic_batch(seed=64, batch=2, num_segments=25, num_class=51)
    calls = []
    orig = torch.nn.Conv2d._conv_forward
    monkeypatch.setattr(torch.nn.Conv2d, "_conv_forward", lambda self, x, w, b: (calls.append(tuple(w.shape)), orig(self, x, w, b))[1])
    got = evaluate.forward_video(m, mv.to(DEV), res.to(DEV), 25, 1)
    val = driver.validate([(flow, mv, res, tgt)], m, 25, 1.0, 10.0, DEV, log=None)
    assert calls == [], calls
    with torch.no_grad():
        ref = o(mv, res)[0].view(2, 25, 51).mean(1).numpy()
    assert rel_err(torch.from_numpy(got), ref) < 1e-4
    assert np.isfinite(val["loss"]) and 0.0 <= val["top1"] <= 100.0


@pytest.mark.parametrize("size,factor", [(224, 0), (224, 16), (50, 16), (37, 0)])
def test_prepare_inputs_bit_exact(size, factor):
    """GPU input preparation == the reference dataset's tensors, bit for bit, incl. the flip."""
    from dmcnet_amd import transforms
    frames = O.synthetic_frames_u8(71, 2, 3, size=size)                # [B,S,7,H,W] u8
    hwc = np.ascontiguousarray(frames.reshape(6, 7, size, size).transpose(0, 2, 3, 1))
    flip = np.array([0, 0, 0, 1, 1, 1], dtype=np.uint8)                # second clip mirrored
    ref = []
    for i in range(6):
        img = hwc[i]
        if flip[i]:
            img = transforms.flip_with_x_negation(img)                 # int32, as the reference
        ref.append(np.transpose(np.asarray(img), (2, 0, 1)))
    ref = np.stack(ref)                                                # [6,7,H,W]
    f0, m0, r0 = O.normalize_sample(ref, factor)
    f, m, r = ops.prepare_inputs(torch.from_numpy(hwc).to(DEV), torch.from_numpy(flip), factor)
    assert torch.equal(m.cpu(), m0) and torch.equal(r.cpu(), r0)
    assert torch.equal(f.cpu(), f0)


@pytest.mark.parametrize("factor", [0, 16])
def test_prepare_inputs_crop_resize_flip_bit_exact(factor):
    """Crop + bilinear resize + flip + blockify + normalise on the GPU == the CPU transforms followed by
    the dataset's tensor conversion, bit for bit, for the reference's training pipeline
    (GroupMultiScaleCrop + GroupRandomHorizontalFlip: every scale pair, both flip outcomes) and its
    validation pipeline (GroupScale + GroupCenterCrop); code/dmcnet/transforms.py:36-139,
    dataset.py:215-263."""
    import random
    from dmcnet_amd import dataset, transforms
    rs = np.random.RandomState(31)
    pipes = [transforms.Compose([transforms.GroupMultiScaleCrop(224, [1, .875, .75]),
                                 transforms.GroupRandomHorizontalFlip()]),
             transforms.Compose([transforms.GroupScale(256), transforms.GroupCenterCrop(224)]),
             transforms.Compose([transforms.GroupCenterCrop(224), transforms.GroupRandomHorizontalFlip()])]
    frames, plans, flips, want = [], [], [], []
    for k in range(9):
        f = rs.randint(0, 256, (256, 340, 7)).astype(np.uint8)
        random.seed(100 + k)
        plan, out, flip = transforms.geometry_plan(pipes[k % 3], f.shape)
        assert out == (224, 224)
        frames.append(f); plans.append(plan); flips.append(int(flip))
        want.append(np.transpose(np.asarray(transforms.apply_plan(f, plan, out, flip)), (2, 0, 1)))
    assert 0 < sum(flips) < len(flips) and any(p[2] != 224 for p in plans)
    ref = dataset.to_tensors(np.stack(want), factor)
    got = ops.prepare_inputs(torch.from_numpy(np.stack(frames)).to(DEV), torch.tensor(flips, dtype=torch.uint8),
                             factor, boxes=torch.tensor(plans, dtype=torch.int32), out_size=(224, 224))
    for g, r, name in zip(got, ref, ("flow", "mv", "res")):
        assert torch.equal(g.cpu(), r), name


def test_device_prep_matches_reference_getitem(golden, tmp_path, monkeypatch):
    """a19 end to end on the GPU: raw uint8 item -> DevicePrep (dmc_prepare_inputs_crop) == the 4-tuple
    the REFERENCE's CoviarDataSet.__getitem__ returned for the same files (golden G9)."""
    import random
    import sys
    from dmcnet_amd import dataset, transforms
    from tests.golden import coviar_fixture as CF
    monkeypatch.setitem(sys.modules, "coviar", CF.coviar_module())
    data_root, flow_root, lst = CF.write_dataset(str(tmp_path))
    g = golden("g9_dataset_item")
    prep = dataset.DevicePrep(DEV, flow_ds_factor=0)
    for tag, is_train, minmax, seed, index, with_flip in CF.CASES:
        ts = [transforms.GroupCenterCrop(CF.CROP)] + ([transforms.GroupRandomHorizontalFlip()] if with_flip else [])
        ds = dataset.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 0, False, transforms.Compose(ts),
                                   3, is_train, True, 12, mv_minmaxnorm=minmax)
        random.seed(seed)
        flow, mv, res, label = prep(dataset.collate_raw([ds.raw_item(index)]))
        assert int(label[0]) == int(g[tag + "_label"])
        for got, key in ((flow, "_flow"), (mv, "_mv"), (res, "_res")):
            assert torch.equal(got[0].cpu(), torch.from_numpy(g[tag + key])), (tag, key)


def test_device_prep_flow_ds_factor_16_matches_reference_getitem(golden, tmp_path, monkeypatch):
    """The same with flow_ds_factor = 16 (BASELINE config 2's setting): DevicePrep's one-pass blockify on the device against the
    reference's __getitem__ (golden g9_dataset_item_ds16; code/dmcnet/dataset.py:229-246), whole and ragged blocks, bit for bit."""
    import random
    import sys
    from dmcnet_amd import dataset, transforms
    from tests.golden import coviar_fixture as CF
    monkeypatch.setitem(sys.modules, "coviar", CF.coviar_module())
    data_root, flow_root, lst = CF.write_dataset(str(tmp_path))
    g = golden("g9_dataset_item_ds16")
    prep = dataset.DevicePrep(DEV, flow_ds_factor=16)
    for tag, is_train, minmax, seed, index, with_flip in CF.CASES:
        for crop in CF.DS16_CROPS:
            ts = [transforms.GroupCenterCrop(crop)] + ([transforms.GroupRandomHorizontalFlip()] if with_flip else [])
            ds = dataset.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 16, False, transforms.Compose(ts),
                                       3, is_train, True, 12, mv_minmaxnorm=minmax)
            key = "%s_c%d" % (tag, crop)
            random.seed(seed)
            flow, mv, res, label = prep(dataset.collate_raw([ds.raw_item(index)]))
            assert int(label[0]) == int(g[key + "_label"])
            for got, k in ((flow, "_flow"), (mv, "_mv"), (res, "_res")):
                assert torch.equal(got[0].cpu(), torch.from_numpy(g[key + k])), (key, k)


def test_i3d_forward_vs_reference_golden(golden):
    """BASELINE config 5: I3D trunk over the per-frame HIP generator, eval mode, against the
    reference's own i3d.py (golden G8); bf16-autocast trunk sanity."""
    from dmcnet_amd import i3d
    g = golden("g8_i3d_eval")
    ref = O.build_estimator("DenseNetTiny")      # only to reuse seeded_state_fill's key-based fill
    net = i3d.I3D(51, modality="flow+mp4", dropout_prob=0, arch_estimator="DenseNetTiny",
                  arch_d="Discriminator")
    assert list(net.state_dict().keys()) == g["keys"].tolist()
    O.seeded_state_fill(net, seed=81)
    net.to(DEV).eval()
    data = rnd(82, (1, 7, 16, 224, 224)).to(DEV)
    with torch.no_grad():
        logits, flow = net(data[:, :5], node="flow+logit")
        validity = net(flow.transpose(1, 2).reshape(-1, 2, 224, 224)[:4], node="D")
    assert rel_err(logits, g["logits"]) < 1e-4
    np.testing.assert_allclose(checksum(flow.cpu()), g["flow_checksum"], rtol=1e-5)
    assert rel_err(flow[0, :, 3, 100:104, 50:66], g["flow_slice"]) < 1e-5
    assert rel_err(validity, g["validity"]) < 1e-4
    net.trunk_dtype = torch.bfloat16
    with torch.no_grad():
        lb = net(data[:, :5])
    assert rel_err(lb, g["logits"]) < 5e-2            # bf16 trunk: loose sanity only


def _g11_run(g, detach, trunk_dtype):
    from dmcnet_amd import i3d
    c = eval(str(g["cfg"]))
    net = i3d.I3D(c["num_classes"], modality="flow+mp4", dropout_prob=0, arch_estimator="DenseNetTiny", arch_d="Discriminator")
    assert list(net.state_dict().keys()) == g["keys"].tolist()
    O.seeded_state_fill(net, seed=c["seed_net"])
    net.to(DEV).train()
    net.trunk_dtype = trunk_dtype
    data = rnd(c["seed_data"], (1, 7, c["frames"], 224, 224)).to(DEV)
    net.discriminator.forced_masks = {k: v.to(DEV) for k, v in O.seeded_dropout_masks(c["seed_masks"], net.discriminator, 2 * c["frames"]).items()}
    logits, losses = i3d.i3d_losses(net, data, torch.tensor([c["label"]], device=DEV), stage=1, detach=detach)
    sum(losses).backward()
    return net, logits, losses


@pytest.mark.parametrize("tag,detach", [("nodetach", False), ("detach", True)])
def test_i3d_training_losses_vs_reference_golden(golden, tag, detach):
    """G11 (the reference's own static_model.forward in TRAINING mode, code/dmcnet_I3D/train/model.py:135-188, around its own
    I3D): ``i3d.i3d_losses`` -- per-frame HIP generator, the three reductions on csrc/losses.hip, HIP discriminator --
    with an fp32 trunk: losses and logits to 1e-4; the twelve named gradients CONDITIONED -- the trunk's training-mode
    BatchNorm chain at batch 1 is ill-conditioned (the reference's own fp32 run is 2-3 % from an fp64 evaluation of the same
    graph on trunk / generator gradients; G11 stores both), so the device run must be within 3 x the reference's own
    distance to the fp64 gradients; then the bf16 trunk (BASELINE config 5's setting)
    against this fp32 run: generator / discriminator losses and every gradient that does not pass the trunk unchanged to
    1e-5, the classification loss / logits at bf16's resolution."""
    from tests.test_oracle_golden import g11_compare
    g = golden("g11_i3d_train")
    net, logits, losses = _g11_run(g, detach, None)
    worst = g11_compare(g, tag, logits, losses, dict(net.named_parameters()), 1e-4, 1e-4, cond=3.0)
    print("G11 fp32 trunk", tag, {k: ("%.1e" % v if isinstance(v, float) else "%.1e (ref %.1e)" % v) for k, v in worst.items()})
    sd = net.state_dict()
    assert rel_err(sd["conv3d_1a_7x7.batch3d.running_mean"], g[tag + "_stem_running_mean"]) < 1e-4
    assert rel_err(sd["conv3d_1a_7x7.batch3d.running_var"], g[tag + "_stem_running_var"]) < 1e-4
    net16, logits16, losses16 = _g11_run(g, detach, torch.bfloat16)
    assert abs(float(losses16[1]) - float(losses[1])) <= 1e-5 * float(losses[1])
    assert abs(float(losses16[2]) - float(losses[2])) <= 1e-5 * float(losses[2])
    assert abs(float(losses16[0]) - float(losses[0])) <= 2e-2 * float(losses[0])
    assert rel_err(logits16, logits) < 5e-2
    p32, p16 = dict(net.named_parameters()), dict(net16.named_parameters())
    rep = {}
    for k in g["grad_names"].tolist():
        rep[k] = float((p16[k].grad - p32[k].grad).norm() / p32[k].grad.norm())
        # What does not pass the trunk is unchanged.  What does is REPORTED, not bounded: on this seeded random-weight network
        # the gradient through 58 training-mode BatchNorms at batch 1 amplifies fp32 rounding (6e-8) to 2-3 % (G11's fp64 leg), so
        # bf16 rounding (4e-3) leaves no common digits -- recorded 1.2 on the generator's first layer; the bf16 kernels themselves
        # are held against fp64 / the stock bf16 trunk in tests/test_conv3d_gpu.py.
        if k.startswith("discriminator") or (detach and k.startswith("gen_flow_model")):
            assert rep[k] <= 1e-5, (k, rep[k])
    print("G11 bf16 trunk vs fp32", tag, {k: "%.1e" % v for k, v in rep.items()})


@pytest.mark.parametrize("iter_size", [1, 2])
def test_i3d_trainer_graph_mode_equals_eager(iter_size):
    """I3DTrainer.enable_graphs(): forward + losses + backward replayed from a hipGraph per phase kind, the policy (learning
    rates, which optimizer steps, Adam) eager -- against the eager trainer from the same state on the same micro-batches,
    eight micro-steps (D and G phases, gradient accumulation over ``iter_size`` and ACROSS phases, epoch 0 and epoch 1 G
    kinds): every loss and every parameter after every micro-step.  Dropout off / masks fixed (a replay draws from the
    graph's own Philox offsets, so random masks differ between the modes by construction); bf16 trunk as in BASELINE
    config 5.  Reference: model.fit, code/dmcnet_I3D/train/model.py:345-491."""
    from dmcnet_amd import i3d, i3d_train

    def build():
        torch.manual_seed(5)
        net = i3d.I3D(51, modality="flow+mp4", dropout_prob=0, arch_estimator="DenseNetTiny", arch_d="Discriminator")
        O.seeded_state_fill(net, seed=87)
        net.to(DEV).train()
        net.trunk_dtype = torch.bfloat16
        net.discriminator.forced_masks = {k: v.to(DEV) for k, v in O.seeded_dropout_masks(88, net.discriminator, 32).items()}
        return net, i3d_train.recipe_trainer(net, batch_size=1, iter_size=iter_size, epoch_thre=1)

    (ne, te), (ng, tg) = build(), build()
    tg.enable_graphs(warmup=1)
    batches = [(rnd(900 + i, (1, 7, 16, 224, 224)).to(DEV), torch.tensor([i % 51], device=DEV)) for i in range(3)]
    worst, replays = 0.0, 0
    for k in range(12):
        epoch = 0 if k < 8 else 1                      # the G kind changes at epoch 1 (classification loss no longer x 0)
        data, tgt = batches[k % 3]
        oe, le, pe, se = te.step(data, tgt, epoch, k)
        og, lg, pg, sg = tg.step(data, tgt, epoch, k)
        assert (pe, se) == (pg, sg)
        for a, b in zip(le, lg):
            worst = max(worst, abs(float(a) - float(b)) / max(abs(float(a)), 1e-12))
        for (ka, a), (_, b) in zip(ne.named_parameters(), ng.named_parameters()):
            d = float((a - b).abs().max())
            if d > 0:
                worst = max(worst, d / max(float(a.abs().max()), 1e-12))
        for (ka, a), (_, b) in zip(ne.named_buffers(), ng.named_buffers()):
            assert torch.equal(a, b) or rel_err(b.float(), a.float()) < 1e-5, ka
    assert len(tg._graphs) >= 2 and tg.static_batch() is not None
    print("graph vs eager, iter_size %d: worst relative difference %.2e over 12 micro-steps, %d graphs" % (iter_size, worst, len(tg._graphs)))
    assert worst <= 1e-5, worst


def test_i3d_train_step_phases():
    from dmcnet_amd import i3d
    torch.manual_seed(0)
    net = i3d.I3D(51, modality="flow+mp4", dropout_prob=0, arch_estimator="DenseNetTiny",
                  arch_d="Discriminator").to(DEV).train()
    from dmcnet_amd import i3d_train
    step = i3d_train.recipe_trainer(net, batch_size=1, iter_size=1)
    data, tgt = rnd(83, (1, 7, 16, 224, 224)).to(DEV), torch.tensor([3], device=DEV)
    w_gen = net.gen_flow_model.predict_flow.weight.detach().clone()
    w_d = net.discriminator.adv_layer.weight.detach().clone()
    w_trunk = net.conv3d_1a_7x7.conv3d.weight.detach().clone() if hasattr(net.conv3d_1a_7x7, "conv3d") else None
    _, losses, phase, stepped = step.step(data, tgt, 0, 0)
    assert phase == "D" and stepped and len(losses) == 3 and all(torch.isfinite(l) for l in losses)
    assert torch.equal(net.gen_flow_model.predict_flow.weight, w_gen)      # D phase: G untouched
    assert not torch.equal(net.discriminator.adv_layer.weight, w_d)
    if w_trunk is not None:                                                # stage 1 + detach: pretrained trunk frozen (lr 0)
        assert torch.equal(net.conv3d_1a_7x7.conv3d.weight, w_trunk)
    w_d = net.discriminator.adv_layer.weight.detach().clone()
    _, _, phase, stepped = step.step(data, tgt, 0, 1)
    assert phase == "G" and stepped
    assert not torch.equal(net.gen_flow_model.predict_flow.weight, w_gen)
    assert torch.equal(net.discriminator.adv_layer.weight, w_d)


@pytest.mark.parametrize("kw", [dict(arch_estimator="DenseNetTiny", gen_flow_ds_factor=16, gen_flow_or_delta=1),
                                dict(arch_estimator="ContextNetwork", att=1),
                                dict(arch_estimator="DenseNetSmall", gen_flow_or_delta=1, arch_d="Discriminator4"),
                                dict(arch_estimator="DenseNetTinyEarlyFusionSum", arch_d="Discriminator")])
def test_model_variants_forward_vs_oracle(kw):
    """API-only variants (other estimators, down-sampled generator, attention head, other
    discriminators): same outputs as the oracle model in eval mode."""
    gan = kw.get("arch_d") is not None
    o = O.OracleModel(51, 3, "mv", base_model="resnet18", use_databn=0, **kw)
    O.seeded_state_fill(o, 91)
    m = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, **kw)
    m.load_state_dict(o.state_dict())
    o.eval(); m.to(DEV).eval()
    flow, mv, res, _ = O.synthetic_batch(seed=92, batch=1, num_segments=3, num_class=51)
    with torch.no_grad():
        ro = o(mv, res, flow) if gan else o(mv, res)
        rm = m(mv.to(DEV), res.to(DEV), flow.to(DEV)) if gan else m(mv.to(DEV), res.to(DEV))
    assert len(ro) == len(rm)
    for a, b in zip(rm, ro):
        assert a.shape == b.shape and rel_err(a, b) < 2e-4


def test_driver_fit_epochs_and_checkpoint(tmp_path):
    """Two tiny epochs through the epoch driver: freeze schedule, validation, checkpoint in the
    reference's layout; the MSE loss must go down on a fixed synthetic set."""
    import os
    from dmcnet_amd import dataset, driver
    torch.manual_seed(0)
    m = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1,
                         arch_estimator="DenseNetTiny").to(DEV)
    ds = dataset.SyntheticCoviarDataSet(8, 51, num_segments=3, flow_ds_factor=16, size=224)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)
    step = T.DmcnetTrainStep(m, 3, 1.0, 10.0, lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
    w_cls = m.base_model.fc.weight.detach().clone()
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        hist, best = driver.fit(m, step, loader, loader, epochs=3, lr=0.01, weight_decay=1e-4,
                                lr_steps=[20, 35, 45], epoch_thre=1, eval_freq=1, log=None,
                                model_prefix="t", device=DEV, miopen_find=False)
    finally:
        os.chdir(cwd)
    assert len(hist) == 3 and all("val" in h for h in hist)
    assert hist[2]["train"]["loss_mse"] < hist[0]["train"]["loss_mse"]
    assert os.path.exists(os.path.join(tmp_path, "t_mv_checkpoint.pth.tar"))
    ck = torch.load(os.path.join(tmp_path, "t_mv_checkpoint.pth.tar"), map_location="cpu")
    assert set(ck) >= {"epoch", "arch", "state_dict", "best_prec1", "optimizer_cls", "optimizer_gf"}
    assert all(k.startswith("module.") for k in ck["state_dict"])
    assert not torch.equal(m.base_model.fc.weight.detach(), w_cls)   # unfrozen after epoch_thre


def test_driver_epoch_from_raw_uint8_batches_matches_float_loader():
    """The wired GPU-side input path: RawView + collate_raw + DevicePrep feed the same epoch as the
    float loader (dataset tensors made on the CPU): identical inputs, so the deterministic generator /
    MSE side is bit-identical; the classifier side is compared at 1e-4 (its kernels are deterministic too -- the bar
    dates from the MIOpen-convolution days and is kept as the north_star's tolerance)."""
    from dmcnet_amd import dataset, driver
    ds = dataset.SyntheticCoviarDataSet(4, 51, num_segments=3, flow_ds_factor=16, size=224)
    float_loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)
    raw_loader = torch.utils.data.DataLoader(dataset.RawView(ds), batch_size=2, shuffle=False,
                                             collate_fn=dataset.collate_raw)
    prep = dataset.DevicePrep(DEV, flow_ds_factor=16)
    a = next(iter(float_loader))
    b = prep(next(iter(raw_loader)))
    for x, y in zip(a, b):
        assert torch.equal(x.to(DEV), y)
    res = []
    for loader, p in ((float_loader, None), (raw_loader, prep)):
        _, m = _product(False, 41)
        m.train()
        step = T.DmcnetTrainStep(m, 3, 1.0, 10.0, lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
        res.append(driver.train_epoch(loader, step, 0, DEV, log=None, prep=p))
    assert res[0]["loss_mse"] == res[1]["loss_mse"]
    assert abs(res[0]["loss"] - res[1]["loss"]) <= 1e-5 * abs(res[0]["loss"])


CONV_CASES = [  # (N, Cin, H, W, Cout, k, stride)
    (2, 64, 56, 56, 64, 3, 1),       # ResNet layer1
    (2, 64, 56, 56, 128, 3, 2),      # layer2.0.conv1
    (2, 64, 56, 56, 128, 1, 2),      # layer2.0.downsample
    (3, 128, 28, 28, 128, 3, 1),
    (2, 256, 14, 14, 256, 3, 1),
    (5, 512, 7, 7, 512, 3, 1),       # layer4: the 64-pixel tile configuration
    (2, 256, 14, 14, 512, 3, 2),
    (2, 16, 112, 112, 16, 3, 1),     # discriminator blocks
    (2, 16, 112, 112, 32, 3, 2),
    (3, 32, 56, 56, 32, 3, 1),
    (1, 32, 13, 11, 64, 3, 2),       # ragged: odd sizes, tiles ending inside the image
    (3, 48, 9, 7, 80, 3, 1),         # channel counts that are multiples of 16 only
    (3, 64, 28, 28, 64, 3, 1),       # discriminator 3_2 / 3_3
    (2, 128, 28, 28, 256, 1, 2),     # layer3.0.downsample
    (3, 64, 13, 9, 128, 3, 2),       # second-generation kernels on ragged sizes (odd rows / columns, partial tiles)
    (2, 64, 5, 3, 64, 3, 1),
    (1, 128, 6, 2, 64, 1, 1),
    (3, 16, 9, 7, 16, 3, 1),         # the small-channel bf16x3 kernel (conv_small.hip) on ragged sizes: 16-pixel tiles that
    (1, 32, 5, 3, 32, 3, 1),         # straddle rows and images, a last tile with invalid pixels, a 1 x 1 image
    (2, 16, 1, 1, 16, 3, 1),
    (5, 32, 4, 30, 32, 3, 1),
    (3, 16, 10, 38, 32, 3, 2),       # the stride-2 small-channel weight gradient (even / odd column staging): a row of 16 + 3 output
    (1, 32, 6, 4, 64, 3, 2),         # pixels, images smaller than a tile pair, an odd number of tiles
    (2, 32, 56, 56, 64, 3, 2),
    (1, 16, 2, 2, 32, 3, 2),
]


@pytest.fixture
def conv_arith(request):
    """Option conv_arith for one test: 0 = fp32 MFMA, 1 = bf16x3 (fp32 products from three bf16 slices)."""
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(b"conv_arith")
    dmcnet_amd._lib.check(lib.dmc_set_option(b"conv_arith", int(request.param)), "dmc_set_option")
    yield int(request.param)
    dmcnet_amd._lib.check(lib.dmc_set_option(b"conv_arith", before), "dmc_set_option")


@pytest.mark.parametrize("conv_arith", [0, 1], indirect=True)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_nhwc_fwd_dgrad_wgrad_vs_fp64(case, conv_arith):
    """The matrix-core NHWC convolution and both gradients against an fp64 evaluation of
    F.conv2d (the arithmetic the reference's nn.Conv2d performs), for the ResNet-18 and discriminator
    shapes, in both arithmetics (the bar is the same: bf16x3 is an fp32-accurate product); results are
    deterministic (two runs bit-identical)."""
    n, cin, h, w, cout, k, stride = case
    pad = k // 2
    x, wt = rnd(201, (n, cin, h, w)), rnd(202, (cout, cin, k, k)) * 0.1
    xo, wo = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yo = F.conv2d(xo, wo, None, stride, pad)
    go = rnd(203, tuple(yo.shape))
    (yo * go.double()).sum().backward()
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = wt.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert ops.conv_nhwc_supported(xg, wg, stride, pad)
    y = ops.conv_nhwc(xg, wg, stride, pad)
    (y * go.to(DEV)).sum().backward()
    assert y.shape == yo.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert rel_err(y, yo) < 1e-5                 # fp32 fmaf chains over K = 9 Cin up to 4608
    assert rel_err(xg.grad, xo.grad) < 1e-5
    assert rel_err(wg.grad, wo.grad) < 1e-5
    assert wg.grad.shape == wg.shape and wg.grad.is_contiguous(memory_format=torch.channels_last)
    g1, d1 = wg.grad.clone(), xg.grad.clone()
    xg.grad = wg.grad = None
    (ops.conv_nhwc(xg, wg, stride, pad) * go.to(DEV)).sum().backward()
    assert torch.equal(g1, wg.grad) and torch.equal(d1, xg.grad)


@pytest.mark.parametrize("case", [(3, 64, 14, 14, 64, 3, 1), (2, 64, 15, 15, 128, 3, 2), (2, 128, 9, 9, 256, 1, 2)])
def test_conv_nhwc_presplit_weights_bit_identical(case):
    """dmc_conv_nhwc_split fills the forward's and the data gradient's bf16x3 weight slices in one launch; the
    convolution called with w == NULL on them is bit-identical to the call that splits by itself, and the
    fused conv + BatchNorm op (which takes that route) matches the stock modules' gradients."""
    n, cin, h, w, cout, k, stride = case
    pad = k // 2
    lib = dmcnet_amd._lib.load()
    if not lib.dmc_conv_nhwc_presplit_supported(cin, cout):
        pytest.skip("the bf16x3 arithmetic is switched off")
    x = rnd(211, (n, cin, h, w)).to(DEV).contiguous(memory_format=torch.channels_last)
    wt = (rnd(212, (cout, cin, k, k)) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last)
    y0, _, _ = ops._conv_fwd(x, wt, None, None, stride, pad, 0, False)
    go = rnd(213, tuple(y0.shape)).to(DEV).contiguous(memory_format=torch.channels_last)
    d0 = ops._conv_dgrad(go, wt, x.shape, stride, pad)
    nb = lib.dmc_conv_nhwc_wt_bytes(cin, cout, k, k)
    wf, wtr = ops._floats(nb, x.device), ops._floats(nb, x.device)
    dmcnet_amd._lib.check(lib.dmc_conv_nhwc_split(dmcnet_amd._lib.ptr(wt), dmcnet_amd._lib.ptr(wf), dmcnet_amd._lib.ptr(wtr),
                                                  cin, cout, k, k, ops._stream()), "dmc_conv_nhwc_split")
    y1, _, _ = ops._conv_fwd(x, wt, None, None, stride, pad, 0, False, presplit=wf)
    d1 = ops._conv_dgrad(go, wt, x.shape, stride, pad, presplit=wtr)
    assert torch.equal(y0, y1) and torch.equal(d0, d1)


@pytest.mark.parametrize("kind,cin,planes,hw", [("basic", 64, 64, 14), ("basic", 128, 128, 9), ("bottleneck", 256, 64, 8)])
def test_residual_grad_link_bit_identical(kind, cin, planes, hw, monkeypatch):
    """Identity-shortcut blocks (torchvision BasicBlock / Bottleneck, `out += identity`): the residual branch's gradient joins
    the main branch's in the epilogue of the first convolution's data-gradient launch (ops.ResidualGradLink,
    dmc_conv_nhwc_dgrad_add) instead of autograd's separate add -- same two fp32 addends, so every gradient is bit-identical
    to the unlinked run, and both match the stock modules in fp64."""
    from dmcnet_amd import resnet
    if not dmcnet_amd._lib.load().dmc_conv_nhwc_presplit_supported(cin, planes):
        pytest.skip("the bf16x3 arithmetic is switched off")
    monkeypatch.setattr(resnet, "OWN_CONV", True)
    torch.manual_seed(5)
    unit = resnet.ResidualUnit(kind, cin, planes, 1).to(DEV).train()
    assert unit.downsample is None
    x0 = rnd(221, (6, cin, hw, hw)).to(DEV).contiguous(memory_format=torch.channels_last)
    go = rnd(222, (6, cin, hw, hw)).to(DEV).contiguous(memory_format=torch.channels_last)
    state = {k: v.clone() for k, v in unit.state_dict().items()}
    res = {}
    for linked in (True, False):
        monkeypatch.setattr(resnet, "RESIDUAL_GRAD_LINK", linked)
        unit.load_state_dict(state)
        unit.zero_grad(set_to_none=True)
        x = (x0 * 1.0).requires_grad_(True)          # a non-leaf producer, as the block input is in the trunk
        x.retain_grad()
        (unit(x) * go).sum().backward()
        res[linked] = [x.grad.clone()] + [p.grad.clone() for p in unit.parameters()]
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    # the unlinked run against the stock modules in fp64
    monkeypatch.setattr(resnet, "OWN_CONV", False)
    ref = resnet.ResidualUnit(kind, cin, planes, 1).double().train()
    ref.load_state_dict({k: v.cpu().double() if v.is_floating_point() else v.cpu() for k, v in state.items()})
    xr = x0.cpu().double().requires_grad_(True)
    (ref(xr) * go.cpu().double()).sum().backward()
    assert rel_err(res[True][0], xr.grad) < 2e-5
    for a, p in zip(res[True][1:], ref.parameters()):
        assert rel_err(a, p.grad) < 2e-4


@pytest.mark.parametrize("first,cin,cout,stride,use_bn,hw", [(True, 2, 16, 2, False, 40), (False, 16, 16, 1, True, 20),
                                                             (False, 16, 32, 2, True, 21), (False, 64, 128, 2, True, 9),
                                                             (False, 64, 64, 1, True, 28), (False, 32, 64, 2, True, 56),
                                                             (False, 32, 32, 1, True, 56)])
def test_disc_block_unit(first, cin, cout, stride, use_bn, hw):
    """One fused discriminator block (conv + bias + LeakyReLU(0.2) + keep mask [+ BatchNorm eps 0.8])
    against the stock modules in fp64: output, every gradient, running statistics; eval mode."""
    n = 5
    x = rnd(211, (n, cin, hw, hw + 3))
    conv_o = torch.nn.Conv2d(cin, cout, 3, stride, 1)
    bn_o = torch.nn.BatchNorm2d(cout, 0.8) if use_bn else None
    O.seeded_state_fill(conv_o, 212)
    if use_bn:
        O.seeded_state_fill(bn_o, 213)
    keep = (torch.from_numpy(np.random.RandomState(214).rand(n, cout)) < 0.75).float() / 0.75
    conv_m = torch.nn.Conv2d(cin, cout, 3, stride, 1)
    conv_m.load_state_dict(conv_o.state_dict())
    conv_m = conv_m.to(DEV).to(memory_format=torch.channels_last)
    bn_m = None
    if use_bn:
        bn_m = torch.nn.BatchNorm2d(cout, 0.8)
        bn_m.load_state_dict(bn_o.state_dict())
        bn_m.to(DEV)
    conv_d = conv_o.double()
    bn_d = bn_o.double() if use_bn else None
    xo = x.double().requires_grad_(True)
    zo = F.leaky_relu(conv_d(xo), 0.2) * keep.double()[:, :, None, None]
    yo = bn_d(zo) if use_bn else zo
    r = rnd(215, tuple(yo.shape))
    (yo * r.double()).sum().backward()
    xg = x.to(DEV)
    if not first:
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg.requires_grad_(True)
    assert ops.disc_block_supported(xg, conv_m, first)
    y = ops.disc_block(xg, conv_m, keep.to(DEV), bn_m, True, first=first)
    (y * r.to(DEV)).sum().backward()
    assert rel_err(y, yo) < 1e-5
    assert rel_err(xg.grad, xo.grad) < 2e-5
    assert rel_err(conv_m.weight.grad, conv_d.weight.grad) < 2e-5
    assert rel_err(conv_m.bias.grad, conv_d.bias.grad) < 2e-5
    if use_bn:
        assert rel_err(bn_m.weight.grad, bn_d.weight.grad) < 2e-5 and rel_err(bn_m.bias.grad, bn_d.bias.grad) < 2e-5
        assert rel_err(bn_m.running_mean, bn_d.running_mean) < 1e-5 and rel_err(bn_m.running_var, bn_d.running_var) < 1e-5
        assert int(bn_m.num_batches_tracked) == 1
        bn_d.eval(); bn_m.eval()
    with torch.no_grad():
        ye = ops.disc_block(xg.detach(), conv_m, None, bn_m, False, first=first)
        ze = F.leaky_relu(conv_d(x.double()), 0.2)
        assert rel_err(ye, bn_d(ze) if use_bn else ze) < 1e-5


@pytest.mark.gpu
def test_generator_half_tile_variant_is_bit_identical():
    """Option gen_layer_path: 5 = every ring layer on the three-stage single-workgroup kernel, 1 (default) = forward layer 2 on
    the two-stage ring with two workgroups per CU, 4 = layers 2 and 3 so, 3 = layers 2 and 3 on 4-row tiles with two pixels
    per lane.  The same products in the same order: the forward output and the backward's gradients are bitwise equal;
    224 x 224 and an edge shape."""
    lib = dmcnet_amd._lib.load()
    if lib.dmc_get_option(b"measure_build") != 1:
        pytest.skip("gen_layer_path 3 / 4 / 5 select variants that lost their measurement: -DDMC_MEASURE build only (DMC_HIP_LIB)")
    before = lib.dmc_get_option(b"gen_layer_path")
    fused_before = lib.dmc_get_option(b"gen_fused")
    dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_fused", 0), "dmc_set_option")       # (the layer-by-layer forward is what the option selects within)
    try:
        for (n, h, w) in ((5, 224, 224), (2, 70, 92)):
            torch.manual_seed(3)
            m = dmcnet_amd.model.EstimatorDenseNetTiny(5).to(DEV)
            mv, res = torch.randn(n, 2, h, w, device=DEV), torch.randn(n, 3, h, w, device=DEV)
            go = torch.randn(n, 2, h, w, device=DEV)
            outs = []
            for path in (5, 1, 3, 4):
                dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_layer_path", path), "dmc_set_option")
                m.zero_grad(set_to_none=True)
                y = m.forward_mv_res(mv, res, True)
                y.backward(go)
                outs.append([y.detach()] + [p.grad.clone() for p in m.parameters()])
            for other in outs[1:]:
                for a, b in zip(outs[0], other):
                    assert torch.equal(a, b)
    finally:
        dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_layer_path", before), "dmc_set_option")
        dmcnet_amd._lib.check(lib.dmc_set_option(b"gen_fused", fused_before), "dmc_set_option")
