"""CPU oracle for the DMC-Net training hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file.  The product package (``dmc-net_amd/``) never does: its hot path is the HIP extension
and it raises when the extension is missing.

What this is: a restatement, on stock ``torch`` CPU fp32 ops and numpy, of the reference's
algorithm for the path named in BASELINE.json -- DMC generator, ResNet classifier, discriminator,
flow-MSE / adversarial / consensus losses, optimiser policy and the integer segment-index
sampling.  The reference is Python on ``torch.nn``; its own arithmetic lives in two third-party
packages that are not in ``/root/reference``:

* **PyTorch** (README pins "pytorch 0.31"; no lock file).  Call sites: every ``nn.*`` in
  ``code/dmcnet/model.py``, ``torch.optim.Adam`` ``code/dmcnet/train.py:134-142``, the losses
  ``code/dmcnet/train.py:166-172``.  torch 2.10 CPU ops are the arithmetic oracle.
* **torchvision** ``models.resnet*`` (unpinned, era 0.2): ``code/dmcnet/model.py:305``.  Absent
  from this image, so the published architecture (He et al. 2015; BasicBlock/Bottleneck,
  ``AvgPool2d(7)``, ``fc``) is restated in :func:`build_resnet`; its ``pretrained=True``
  ImageNet weights cannot be obtained offline -> **for pretrained-weight-dependent numbers
  parity is unpinned** (DESIGN.md says the same).

Pinning: ``tests/golden/make_golden.py`` imports the *reference's own* ``model.py`` /
``dataset.py`` (with stub ``cv2`` / ``torchvision`` / ``coviar`` / ``skimage``) in the build
container, runs it on seeded inputs and commits the outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against those vectors.

Each function cites the reference file:line it follows (paths relative to ``/root/reference``).
"""
from __future__ import annotations

import math
import random
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

# --------------------------------------------------------------------------------------------
# 1. DMC generator zoo                                   code/dmcnet/model.py:31-250
# --------------------------------------------------------------------------------------------

#: growth widths of the dense estimators (code/dmcnet/model.py:122-194)
DENSE_WIDTHS = {
    "DenseNet": (128, 128, 96, 64, 32),        # :122-144
    "DenseNetSmall": (32, 32, 24, 16, 8),      # :147-169
    "DenseNetTiny": (8, 8, 6, 4, 2),           # :172-194  <- the one every run.sh uses
}


def _conv_lrelu(cin, cout):
    # code/dmcnet/model.py:111-115 : Conv2d(k3,s1,p1,bias) + LeakyReLU(0.1), NOT in place
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=True), nn.LeakyReLU(0.1))


class DenseEstimator(nn.Module):
    """EstimatorDenseNet / Small / Tiny: x <- cat(conv_i(x), x) five times, then a bare conv.

    code/dmcnet/model.py:122-194.  New features are PREPENDED, so the input-channel order of
    layer k's weight is [y_{k-1}, ..., y_0, mv(2), residual(3)].
    """

    def __init__(self, ch_in, widths):
        super().__init__()
        c = ch_in
        for i, w in enumerate(widths):
            setattr(self, "conv_%d" % i, _conv_lrelu(c, w))
            c += w
        self.predict_flow = nn.Conv2d(c, 2, 3, 1, 1, bias=True)   # :118-119
        self._n = len(widths)

    def forward(self, x):
        for i in range(self._n):
            x = torch.cat((getattr(self, "conv_%d" % i)(x), x), 1)
        return self.predict_flow(x)


class EarlyFusionEstimator(nn.Module):
    """EstimatorDenseNetTinyEarlyFusionSum / Stack, code/dmcnet/model.py:197-250."""

    def __init__(self, stack):
        super().__init__()
        self.conv_0_mv = _conv_lrelu(2, 8)
        self.conv_0_r = _conv_lrelu(3, 8)
        c = 16 if stack else 8
        for i, w in zip((1, 2, 3, 4), (8, 6, 4, 2)):
            setattr(self, "conv_%d" % i, _conv_lrelu(c, w))
            c += w
        self.predict_flow = nn.Conv2d(c, 2, 3, 1, 1, bias=True)
        self._stack = stack

    def forward(self, x):
        a, b = self.conv_0_mv(x[:, :2]), self.conv_0_r(x[:, 2:])
        x = torch.cat((a, b), 1) if self._stack else a + b
        for i in (1, 2, 3, 4):
            x = torch.cat((getattr(self, "conv_%d" % i)(x), x), 1)
        return self.predict_flow(x)


def _dilated(cin, cout, dil):
    # code/dmcnet/model.py:31-43, batch_norm=True branch (the only one Model uses, :313-315)
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, padding=dil, dilation=dil, bias=False),
                         nn.BatchNorm2d(cout), nn.LeakyReLU(0.1, inplace=True))


class ContextEstimator(nn.Module):
    """ContextNetwork / ContextNetworkAtt, code/dmcnet/model.py:45-104."""

    def __init__(self, ch_in, ds_factor, att):
        super().__init__()
        fifth = 16 if ds_factor == 0 else 1                     # :48-68
        spec = [(ch_in, 32, 1), (32, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, fifth),
                (64, 32, 1)]
        if not att:
            spec.append((32, 2, 1))
        self.conv_context = nn.Sequential(*[_dilated(*s) for s in spec])
        self._att = att
        if att:                                                  # :95-99
            self.predict_flow = _dilated(32, 2, 1)
            self.predict_att = nn.Sequential(_dilated(32, 2, 1), nn.ReLU(inplace=True))

    def forward(self, x):
        x = self.conv_context(x)
        if self._att:
            return self.predict_flow(x), self.predict_att(x)
        return x


def build_estimator(arch, ds_factor=0, att=0):
    """Dispatch of code/dmcnet/model.py:311-325 (ch_in is always 5 = MV 2 + residual 3)."""
    if arch == "ContextNetwork":
        return ContextEstimator(5, ds_factor, bool(att))
    if arch in DENSE_WIDTHS:
        return DenseEstimator(5, DENSE_WIDTHS[arch])
    if arch == "DenseNetTinyEarlyFusionSum":
        return EarlyFusionEstimator(stack=False)
    if arch == "DenseNetTinyEarlyFusionStack":
        return EarlyFusionEstimator(stack=True)
    raise ValueError("unknown estimator %r" % (arch,))


# --------------------------------------------------------------------------------------------
# 2. Discriminators                                      code/dmcnet_GAN/model.py:250-438
# --------------------------------------------------------------------------------------------

#: per variant: list of (attribute suffix, cout, stride); first block has no BatchNorm.
def _disc_plan(widths, extra):
    plan = []
    for stage, w in enumerate(widths, start=1):
        plan.append(("%d" % stage, w, 2))
        for j in range(2, 2 + extra):
            plan.append(("%d_%d" % (stage, j), w, 1))
    return plan


DISC_PLANS = {
    "Discriminator": (_disc_plan((16, 32, 64, 128), 0), 128 * 14 * 14),    # :282-302
    "Discriminator2": (_disc_plan((16, 32, 64, 128), 1), 128 * 14 * 14),   # :305-329
    "Discriminator3": (_disc_plan((16, 32, 64, 128), 2), 128 * 14 * 14),   # :332-366
    "Discriminator4": (_disc_plan((8, 16, 32), 0), 32 * 28 * 28),          # :369-384
    "Discriminator5": (_disc_plan((16, 32, 64, 128), 4), 128 * 14 * 14),   # :387-438
}


class OracleDiscriminator(nn.Module):
    """Conv3x3(s2|s1,p1,bias) -> LeakyReLU(0.2) -> Dropout2d(0.25) -> BatchNorm2d(eps=0.8).

    code/dmcnet_GAN/model.py:254-279: ``nn.BatchNorm2d(out_filters, 0.8)`` binds the second
    positional argument, eps, to 0.8; momentum stays 0.1.  ``forced_masks`` (a dict
    ``block name -> [N,C] float keep-mask already divided by 0.75``) replaces the Bernoulli
    draw so that CPU and GPU runs can be compared.
    """

    def __init__(self, arch, ch_in=2):
        super().__init__()
        plan, flat = DISC_PLANS[arch]
        self.names = []
        c = ch_in
        for idx, (suffix, w, stride) in enumerate(plan):
            layers = [nn.Conv2d(c, w, 3, stride, 1), nn.LeakyReLU(0.2, inplace=True),
                      nn.Dropout2d(0.25)]
            if idx > 0:
                layers.append(nn.BatchNorm2d(w, 0.8))
            name = "discriminator_block_" + suffix
            setattr(self, name, nn.Sequential(*layers))
            self.names.append(name)
            c = w
        self.adv_layer = nn.Linear(flat, 2)
        self.forced_masks = None

    def forward(self, x):
        for name in self.names:
            blk = getattr(self, name)
            if self.forced_masks is None:
                x = blk(x)
            else:
                x = F.leaky_relu(blk[0](x), 0.2)
                x = x * self.forced_masks[name][:, :, None, None]
                if len(blk) == 4:
                    x = blk[3](x)
        return self.adv_layer(x.reshape(x.shape[0], -1))


# --------------------------------------------------------------------------------------------
# 3. ResNet (torchvision.models.resnet*, third-party, restated)   code/dmcnet/model.py:305
# --------------------------------------------------------------------------------------------

class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = down

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _Bottle(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = down

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


RESNET_DEPTHS = {"resnet18": (_Basic, (2, 2, 2, 2)), "resnet34": (_Basic, (3, 4, 6, 3)),
                 "resnet50": (_Bottle, (3, 4, 6, 3)), "resnet101": (_Bottle, (3, 4, 23, 3)),
                 "resnet152": (_Bottle, (3, 8, 36, 3))}


class OracleResNet(nn.Module):
    def __init__(self, block, depths, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), depths), start=1):
            stride = 1 if i == 1 else 2
            blocks = []
            for j in range(n):
                s = stride if j == 0 else 1
                down = None
                if s != 1 or cin != planes * block.expansion:
                    down = nn.Sequential(nn.Conv2d(cin, planes * block.expansion, 1, s, bias=False),
                                         nn.BatchNorm2d(planes * block.expansion))
                blocks.append(block(cin, planes, s, down))
                cin = planes * block.expansion
            setattr(self, "layer%d" % i, nn.Sequential(*blocks))
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def build_resnet(name, pretrained=False):
    """Architecture of ``torchvision.models.<name>``; ``pretrained`` is accepted and ignored
    (no network) -- random init, see the module docstring on unpinned parity."""
    block, depths = RESNET_DEPTHS[name]
    return OracleResNet(block, depths)


# --------------------------------------------------------------------------------------------
# 4. Model                     code/dmcnet/model.py:253-378, code/dmcnet_GAN/model.py:442-585
# --------------------------------------------------------------------------------------------

class OracleModel(nn.Module):
    """TSN wrapper.  ``arch_d=None`` gives the dmcnet variant (classifier sees a DETACHED cue,
    code/dmcnet/model.py:352); a discriminator name gives the GAN variant (no detach,
    code/dmcnet_GAN/model.py:560)."""

    def __init__(self, num_class, num_segments, representation, base_model="resnet152",
                 new_length=1, use_databn=1, gen_flow_or_delta=0, gen_flow_ds_factor=0,
                 arch_estimator="ContextNetwork", arch_d=None, att=0):
        super().__init__()
        self.num_segments, self.att = num_segments, att
        self.gen_flow_or_delta, self.gen_flow_ds_factor = gen_flow_or_delta, gen_flow_ds_factor
        if "resnet" not in base_model:
            raise ValueError("Unknown base model: {}".format(base_model))
        self.base_model = build_resnet(base_model)                       # model.py:305
        self.gen_flow_model = build_estimator(arch_estimator, gen_flow_ds_factor, att)
        if gen_flow_ds_factor != 0:                                       # model.py:326-327
            self.downsample = nn.AvgPool2d(gen_flow_ds_factor, stride=gen_flow_ds_factor)
        if arch_d is not None:                                            # GAN model.py:520-530
            self.discriminator = OracleDiscriminator(arch_d, 2)
        self.is_gan = arch_d is not None
        # _prepare_tsn, model.py:283-299
        self.base_model.fc = nn.Linear(self.base_model.fc.in_features, num_class)
        if representation in ("mv", "flow"):
            self.base_model.conv1 = nn.Conv2d(2 * new_length, 64, 7, 2, 3, bias=False)
            if use_databn == 1:
                self.data_bn = nn.BatchNorm2d(2)
        if representation == "residual" and use_databn == 1:
            self.data_bn = nn.BatchNorm2d(3)

    def forward(self, input_mv, input_residual, input_flow=None):
        mv = input_mv.reshape((-1,) + tuple(input_mv.shape[-3:]))                 # :333
        res = input_residual.reshape((-1,) + tuple(input_residual.shape[-3:]))    # :334
        if self.gen_flow_ds_factor != 0:
            mv, res = self.downsample(mv), self.downsample(res)
        g = self.gen_flow_model(torch.cat((mv, res), 1))                          # :341
        att_flow = None
        if self.att == 1:
            g, att_flow = g
        if self.gen_flow_or_delta == 1:
            g = g + mv                                                            # :345-346
        if self.gen_flow_ds_factor != 0:
            g = g.repeat(1, 1, self.gen_flow_ds_factor, self.gen_flow_ds_factor)  # :347-348
        if not self.is_gan:
            out = (self.base_model(g.detach()), g)                                # :352
        else:
            d_in = g
            if input_flow is not None:       # "first fake then real", GAN model.py:555-557
                d_in = torch.cat((g, input_flow.reshape((-1,) + tuple(input_flow.shape[-3:]))), 0)
            out = (self.base_model(g), self.discriminator(d_in), g)               # :560-561
        return out + ((att_flow,) if self.att == 1 else ())


# --------------------------------------------------------------------------------------------
# 5. Optimiser policy, losses, steps     code/dmcnet/train.py:121-142,205-266,398-424
#                                         code/dmcnet_GAN/train.py:122-153,219-397
# --------------------------------------------------------------------------------------------

def make_optimizers(model, lr, weight_decay, lr_cls_mult, lr_mse_mult, lr_d_mult=None):
    """One param group PER TENSOR, routed by key substring; biases get decay_mult 0;
    Adam(eps=1e-3) with coupled L2.  code/dmcnet/train.py:121-142, GAN :122-153."""
    routes = [("base_model", lr_cls_mult), ("gen_flow_model", lr_mse_mult)]
    if lr_d_mult is not None:
        routes.append(("discriminator", lr_d_mult))
    groups = [[] for _ in routes]
    for key, value in model.named_parameters():
        for slot, (tag, mult) in enumerate(routes):
            if tag in key:
                groups[slot].append({"params": value, "lr": lr, "lr_mult": mult,
                                     "decay_mult": 0.0 if "bias" in key else 1.0})
    return [torch.optim.Adam(g, weight_decay=weight_decay, eps=0.001) for g in groups]


def adjust_learning_rate(optimizer, epoch, lr_steps, lr_decay, base_lr, weight_decay,
                         freeze=False, epoch_thre=500):
    """code/dmcnet/train.py:398-408."""
    decay = lr_decay ** int(sum(epoch >= np.array(lr_steps)))
    lr, wd = base_lr * decay, weight_decay
    if epoch < epoch_thre and freeze:
        lr = wd = 0
    for g in optimizer.param_groups:
        g["lr"] = lr * g["lr_mult"]
        g["weight_decay"] = wd * g["decay_mult"]
    return lr


def accuracy(output, target, topk=(1,)):
    """Top-k precision in percent, code/dmcnet/train.py:411-424."""
    _, pred = output.topk(max(topk), 1, True, True)
    hit = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [hit[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


def consensus(output, num_segments):
    """code/dmcnet/train.py:239-240: view(-1,S,C).mean(1)."""
    return output.view((-1, num_segments) + tuple(output.shape[1:])).mean(dim=1)


def dmcnet_train_step(model, opt_cls, opt_gf, batch, num_segments, lr_cls, lr_mse, freeze=False):
    """One iteration of code/dmcnet/train.py:221-266 on an already-normalised batch."""
    input_flow, input_mv, input_residual, target = batch
    flow = input_flow.reshape((-1,) + tuple(input_mv.shape[-3:]))               # :230
    output, gen_flow = model(input_mv, input_residual)                          # :236
    output = consensus(output, num_segments)
    loss_cls = F.cross_entropy(output, target)                                  # :241
    loss_mse = F.mse_loss(gen_flow, flow)                                       # :245
    loss = loss_cls * lr_cls + loss_mse * lr_mse                                # :248
    opt_cls.zero_grad()
    opt_gf.zero_grad()
    if freeze:                                                                  # :260-262
        (loss_mse * lr_mse).backward()
    else:
        loss.backward()
        opt_cls.step()
    opt_gf.step()                                                               # :266
    return {"loss": loss.detach(), "loss_cls": loss_cls.detach(), "loss_mse": loss_mse.detach(),
            "output": output.detach(), "gen_flow": gen_flow.detach()}


def gan_train_step(model, opt_cls, opt_gf, opt_d, batch, i, num_segments, lr_cls, lr_adv_g,
                   lr_adv_d, lr_mse):
    """Iteration ``i`` of code/dmcnet_GAN/train.py:236-371: even i trains D (+classifier),
    odd i trains G."""
    input_flow, input_mv, input_residual, target = batch
    flow = input_flow.reshape((-1,) + tuple(input_mv.shape[-3:]))
    valid = torch.cat([torch.ones_like(target)] * num_segments, 0)              # :253-256
    fake = torch.cat([torch.zeros_like(target)] * num_segments, 0)
    out = {}
    if i % 2 == 0:                                                              # :261
        output, validity, gen_flow = model(input_mv, input_residual, flow)
        output = consensus(output, num_segments)
        loss_cls = F.cross_entropy(output, target)
        loss_adv = F.cross_entropy(validity, torch.cat((fake, valid), 0))       # :274
        loss = loss_cls * lr_cls + loss_adv * lr_adv_d                          # :278
        for o in (opt_cls, opt_gf, opt_d):
            o.zero_grad()
        loss.backward()
        opt_cls.step()                                                          # :301-302
        opt_d.step()
    else:                                                                       # :331
        output, validity, gen_flow = model(input_mv, input_residual)
        output = consensus(output, num_segments)
        loss_cls = F.cross_entropy(output, target)
        loss_adv = F.cross_entropy(validity, valid)                             # :346
        loss_mse = F.mse_loss(gen_flow, flow)                                   # :350
        loss = loss_cls * lr_cls + loss_adv * lr_adv_g + loss_mse * lr_mse      # :355
        for o in (opt_cls, opt_gf, opt_d):
            o.zero_grad()
        loss.backward()
        opt_gf.step()                                                           # :371
        out["loss_mse"] = loss_mse.detach()
    out.update(loss=loss.detach(), loss_cls=loss_cls.detach(), loss_adv=loss_adv.detach(),
               output=output.detach(), validity=validity.detach(), gen_flow=gen_flow.detach())
    return out


# --------------------------------------------------------------------------------------------
# 6. Integer segment-index sampling (bit-exact)        code/dmcnet/dataset.py:46-73,130-149
# --------------------------------------------------------------------------------------------

_PFRAME = ("residual", "mv", "flow")


def get_seg_range(n, num_segments, seg, representation):
    """code/dmcnet/dataset.py:46-60 (np.round = half-to-even on float64)."""
    if representation in _PFRAME:
        n -= 1
    seg_size = float(n - 1) / num_segments
    b = int(np.round(seg_size * seg))
    e = int(np.round(seg_size * (seg + 1)))
    if e == b:
        e = b + 1
    if representation in _PFRAME:
        return b + 1, e + 1            # frame 0 is the I-frame
    return b, e


def get_gop_pos(frame_idx, representation, gop_size=12):
    """code/dmcnet/dataset.py:63-73."""
    g, p = frame_idx // gop_size, frame_idx % gop_size
    if representation in _PFRAME:
        if p == 0:
            g, p = g - 1, gop_size - 1
    else:
        p = 0
    return g, p


def train_frame_index(num_frames, seg, num_segments, representation, rng=random, gop_size=12):
    """code/dmcnet/dataset.py:130-137: one ``randint(begin, end-1)`` (inclusive) per segment."""
    b, e = get_seg_range(num_frames, num_segments, seg, representation)
    return get_gop_pos(rng.randint(b, e - 1), representation, gop_size)


def test_frame_index(num_frames, seg, num_segments, representation, gop_size=12):
    """code/dmcnet/dataset.py:139-149."""
    if representation in _PFRAME:
        num_frames -= 1
    v = int(np.round(float(num_frames - 1) / num_segments * (seg + 0.5)))
    if representation in _PFRAME:
        v += 1
    return get_gop_pos(v, representation, gop_size)


test_frame_index.__test__ = False   # not a pytest test


def flow_frame_number(gop_index, gop_pos, gop_size=12):
    """code/dmcnet/dataset.py:178: index of the TV-L1 flow JPEG that pairs with (gop, pos)."""
    return gop_index * gop_size + gop_pos + 1


# --------------------------------------------------------------------------------------------
# 7. Tensor contract of the dataset                     code/dmcnet/dataset.py:215-263
# --------------------------------------------------------------------------------------------

_STD = np.array([0.229, 0.224, 0.225])


def blockify_flow(flow_u8, factor):
    """code/dmcnet/dataset.py:229-246 with ``upsample_interp=False``: mean over factor x factor
    blocks (skimage ``block_reduce`` zero-pads ragged edges), repeat, crop back."""
    s, c, h, w = flow_u8.shape
    ph, pw = (-h) % factor, (-w) % factor
    x = np.pad(flow_u8.astype(np.float64), ((0, 0), (0, 0), (0, ph), (0, pw)))
    x = x.reshape(s, c, (h + ph) // factor, factor, (w + pw) // factor, factor).mean(axis=(3, 5))
    return x.repeat(factor, axis=2).repeat(factor, axis=3)[:, :, :h, :w]


def normalize_sample(frames_u8, flow_ds_factor=0):
    """frames ``[S,7,H,W]`` uint8 = [flow2, mv2, res3] -> (input_flow, input_mv, input_residual)
    fp32, exactly as code/dmcnet/dataset.py:224-263 (representation 'mv')."""
    flow, mv, res = frames_u8[:, 0:2], frames_u8[:, 2:4], frames_u8[:, 4:]
    if flow_ds_factor != 0:
        flow = blockify_flow(flow, flow_ds_factor)
    std = torch.from_numpy(_STD.reshape((1, 3, 1, 1))).float()
    t_flow = torch.from_numpy(np.ascontiguousarray(flow)).float() / 255.0
    t_mv = torch.from_numpy(np.ascontiguousarray(mv)).float() / 255.0
    t_res = torch.from_numpy(np.ascontiguousarray(res)).float() / 255.0
    t_mv = (t_mv - 0.5) / torch.mean(std)
    t_flow = (t_flow - 0.5) / torch.mean(std)
    t_res = (t_res - 0.5) / std
    return t_flow, t_mv, t_res


# --------------------------------------------------------------------------------------------
# 8. Seeded fills shared by the golden generator and the tests (not reference behaviour)
# --------------------------------------------------------------------------------------------

def seeded_state_fill(module, seed):
    """Overwrite every parameter/buffer with values that depend only on (seed, key, shape), so the
    reference model and any same-keyed model get identical weights without shipping them."""
    sd = module.state_dict()
    out = OrderedDict()
    for key in sorted(sd.keys()):
        t = sd[key]
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
        if key.endswith("num_batches_tracked"):
            v = torch.zeros_like(t)
        elif key.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif key.endswith("running_mean"):
            v = (torch.rand(t.shape, generator=g) - 0.5) * 0.2
        elif t.dim() == 1 and key.endswith("weight"):          # norm scale
            v = torch.rand(t.shape, generator=g) + 0.5
        elif t.dim() == 1:                                       # biases
            v = (torch.rand(t.shape, generator=g) - 0.5) * 0.2
        else:
            fan_in = int(np.prod(t.shape[1:]))
            bound = math.sqrt(3.0 / fan_in)
            v = (torch.rand(t.shape, generator=g) * 2 - 1) * bound
        out[key] = v.to(t.dtype)
    module.load_state_dict(out)
    return module


def synthetic_frames_u8(seed, batch, num_segments, size=224):
    """uint8 [B,S,7,H,W] = [flow2, mv2, res3] in the spirit of SURVEY.md section 8(d): MV constant
    per 16x16 macroblock, residual and flow per pixel."""
    rs = np.random.RandomState(seed)
    mb = (size + 15) // 16
    mv = np.clip(128 + np.round(rs.normal(0, 6, (batch, num_segments, 2, mb, mb)) * 127.5 / 20),
                 0, 255)
    mv = mv.repeat(16, axis=3).repeat(16, axis=4)[..., :size, :size]
    res = np.clip(128 + np.round(rs.normal(0, 12, (batch, num_segments, 3, size, size))), 0, 255)
    flow = np.clip(128 + np.round(rs.normal(0, 10, (batch, num_segments, 2, size, size))), 0, 255)
    return np.concatenate((flow, mv, res), axis=2).astype(np.uint8)


def synthetic_batch(seed, batch, num_segments, num_class, size=224, flow_ds_factor=0):
    """(input_flow, input_mv, input_residual, target) with the dataset's normalisation."""
    frames = synthetic_frames_u8(seed, batch, num_segments, size)
    parts = [normalize_sample(frames[b], flow_ds_factor) for b in range(batch)]
    flow = torch.stack([p[0] for p in parts]).float()
    mv = torch.stack([p[1] for p in parts]).float()
    res = torch.stack([p[2] for p in parts]).float()
    target = torch.from_numpy(np.random.RandomState(seed + 7).randint(0, num_class, batch)).long()
    return flow, mv, res, target


def seeded_dropout_masks(seed, disc, n):
    """Dropout2d keep-masks ``[n, C] in {0, 1/0.75}`` per ``discriminator_block_*`` child of
    ``disc`` (reference or oracle discriminator), a function of (seed, block name) only."""
    masks = {}
    for name, blk in disc.named_children():
        if not name.startswith("discriminator_block"):
            continue
        c = blk[0].out_channels
        g = torch.Generator().manual_seed((seed * 7919 + zlib.crc32(name.encode())) % (2 ** 31))
        masks[name] = (torch.rand((n, c), generator=g) < 0.75).float() / 0.75
    return masks


def i3d_losses(net, data, target, stage=None, detach=False):
    """Training-mode loss assembly of the I3D variant, code/dmcnet_I3D/train/model.py:135-188 (``static_model.forward``,
    'flow+logit' node) on stock torch CPU ops; pinned by golden G11 (tests/golden/make_golden_i3d_train.py runs the
    reference's own method).  ``net(x, node=..., detach=...)`` is any network with the reference I3D's call contract;
    ``data`` [b,7,T,H,W]: channels [:5] feed the generator, [5:7] are the flow target (:145,:176); with a stage the T axis
    is folded into the batch for the discriminator, generated frames first then the real ones, targets 0 then 1 (:146-156).
    Returns (logits, [loss, mse] or [loss, mse, loss_adv])."""
    output, flow = net(data[:, :5], node="flow+logit", detach=detach)
    losses = [F.cross_entropy(output, target), F.mse_loss(flow, data[:, 5:7])]
    if stage is not None:
        t = flow.size(2)
        h, w = 224, 224                                               # "manually set", :151-152
        valid = torch.cat([torch.ones_like(target)] * t, 0)
        fake = torch.cat([torch.zeros_like(target)] * t, 0)
        d_in = torch.cat((torch.reshape(torch.transpose(flow, 1, 2), (-1, 2, h, w)),
                          torch.reshape(torch.transpose(data[:, 5:7], 1, 2), (-1, 2, h, w))), 0)
        losses.append(F.cross_entropy(net(d_in, node="D"), torch.cat((fake, valid), 0)))
    return output, losses

