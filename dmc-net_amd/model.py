"""``Model`` -- drop-in for the reference's ``from model import Model``.

Same constructor, attribute names (``base_model``, ``gen_flow_model``, ``discriminator``,
``data_bn``, ``downsample``), state-dict keys, ``forward`` contracts, ``crop_size`` /
``scale_size`` / ``get_augmentation`` as code/dmcnet/model.py:253-378 and
code/dmcnet_GAN/model.py:442-585 (paths relative to the reference root).  What differs is
where the arithmetic runs: the DenseNetTiny generator (+ cat + delta add), the ResNet's
convolutions / BatchNorms / stem (``resnet.py`` -> ``ops.conv_bn_act``, ``ops.stem_conv``, ...) and
the Discriminator3 blocks (``ops.disc_block``) are hand-written HIP kernels behind the C ABI; only
the API-only estimator / discriminator variants and the two ``nn.Linear`` heads stay on
PyTorch-ROCm ops.  Passing ``arch_d`` selects the GAN variant.
"""
import torch
from torch import nn
import torch.nn.functional as F

from . import ops, resnet, transforms

# ---------------------------------------------------------------------------- generators

_DENSE = {"DenseNet": (128, 128, 96, 64, 32), "DenseNetSmall": (32, 32, 24, 16, 8),
          "DenseNetTiny": (8, 8, 6, 4, 2)}


def _unit(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True),
                         nn.LeakyReLU(0.1))


class DenseEstimatorBase(nn.Module):
    """Five dense units + a linear 3x3 head (code/dmcnet/model.py:122-194)."""

    def __init__(self, widths, ch_in=5):
        super().__init__()
        total = ch_in
        for i, w in enumerate(widths):
            self.add_module("conv_%d" % i, _unit(total, w))
            total += w
        self.predict_flow = nn.Conv2d(total, 2, kernel_size=3, stride=1, padding=1, bias=True)
        self.depth = len(widths)

    def forward(self, x):
        for i in range(self.depth):
            x = torch.cat((getattr(self, "conv_%d" % i)(x), x), 1)
        return self.predict_flow(x)


class EstimatorDenseNet(DenseEstimatorBase):
    def __init__(self, ch_in=5):
        super().__init__(_DENSE["DenseNet"], ch_in)


class EstimatorDenseNetSmall(DenseEstimatorBase):
    def __init__(self, ch_in=5):
        super().__init__(_DENSE["DenseNetSmall"], ch_in)


class EstimatorDenseNetTiny(DenseEstimatorBase):
    """The generator of every shipped recipe (code/dmcnet/model.py:172-194), evaluated by the
    fused HIP path.  ``forward(x)`` keeps the reference signature (x = cat(mv, residual));
    ``forward_mv_res`` skips the concat and folds the optional ``+ input_mv``."""

    def __init__(self, ch_in=5):
        if ch_in != 5:
            raise ValueError("EstimatorDenseNetTiny is defined for MV(2)+residual(3) input")
        super().__init__(_DENSE["DenseNetTiny"], ch_in)

    def _params(self):
        ws = [getattr(self, "conv_%d" % i)[0].weight for i in range(5)] + [self.predict_flow.weight]
        bs = [getattr(self, "conv_%d" % i)[0].bias for i in range(5)] + [self.predict_flow.bias]
        return ws, bs

    def forward_mv_res(self, mv, res, add_mv=False):
        ws, bs = self._params()
        return ops.gen_tiny(mv, res, ws, bs, add_mv)

    def forward_mv_res_mse(self, mv, res, flow, add_mv=False):
        """(gen_flow, nn.MSELoss()(gen_flow, flow)): the loss is reduced in the forward's last kernel."""
        ws, bs = self._params()
        return ops.gen_tiny_mse(mv, res, flow, ws, bs, add_mv)

    def forward(self, x):
        return self.forward_mv_res(x[:, :2].contiguous(), x[:, 2:].contiguous(), False)


class _EarlyFusion(nn.Module):
    def __init__(self, stack):
        super().__init__()
        self.conv_0_mv, self.conv_0_r = _unit(2, 8), _unit(3, 8)
        total = 16 if stack else 8
        for i, w in ((1, 8), (2, 6), (3, 4), (4, 2)):
            self.add_module("conv_%d" % i, _unit(total, w))
            total += w
        self.predict_flow = nn.Conv2d(total, 2, kernel_size=3, stride=1, padding=1, bias=True)
        self.stack = stack

    def forward(self, x):
        m, r = self.conv_0_mv(x[:, :2]), self.conv_0_r(x[:, 2:])
        x = torch.cat((m, r), 1) if self.stack else m + r
        for i in (1, 2, 3, 4):
            x = torch.cat((getattr(self, "conv_%d" % i)(x), x), 1)
        return self.predict_flow(x)


class EstimatorDenseNetTinyEarlyFusionSum(_EarlyFusion):
    def __init__(self, ch_in=5):
        super().__init__(False)


class EstimatorDenseNetTinyEarlyFusionStack(_EarlyFusion):
    def __init__(self, ch_in=5):
        super().__init__(True)


def _dilated_unit(cin, cout, dilation, batch_norm=True):
    pad = dilation
    if batch_norm:
        return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, pad, dilation, bias=False),
                             nn.BatchNorm2d(cout), nn.LeakyReLU(0.1, inplace=True))
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, pad, dilation, bias=True),
                         nn.LeakyReLU(0.1, inplace=True))


class ContextNetwork(nn.Module):
    """Seven dilated 3x3 units (code/dmcnet/model.py:45-71)."""

    def __init__(self, ch_in, batch_norm=True, gen_flow_ds_factor=0):
        super().__init__()
        d5 = 16 if gen_flow_ds_factor == 0 else 1
        plan = ((ch_in, 32, 1), (32, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, d5),
                (64, 32, 1), (32, 2, 1))
        self.conv_context = nn.Sequential(*[_dilated_unit(a, b, d, batch_norm) for a, b, d in plan])

    def forward(self, x):
        return self.conv_context(x)


class ContextNetworkAtt(nn.Module):
    """code/dmcnet/model.py:74-104: shared trunk, flow head and ReLU attention head."""

    def __init__(self, ch_in, batch_norm=True, gen_flow_ds_factor=0):
        super().__init__()
        d5 = 16 if gen_flow_ds_factor == 0 else 1
        plan = ((ch_in, 32, 1), (32, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, d5), (64, 32, 1))
        self.conv_context = nn.Sequential(*[_dilated_unit(a, b, d, batch_norm) for a, b, d in plan])
        self.predict_flow = _dilated_unit(32, 2, 1, batch_norm)
        self.predict_att = nn.Sequential(_dilated_unit(32, 2, 1, batch_norm), nn.ReLU(inplace=True))

    def forward(self, x):
        x = self.conv_context(x)
        return self.predict_flow(x), self.predict_att(x)


# ------------------------------------------------------------------------- discriminators

def discriminator_block(in_filters, out_filters, bn=True, stride=2):
    """Conv3x3(stride 2) -> LeakyReLU(0.2) -> Dropout2d(0.25) [-> BatchNorm2d(eps=0.8)]
    (code/dmcnet_GAN/model.py:254-265; the reference passes 0.8 positionally, i.e. as eps)."""
    layers = [nn.Conv2d(in_filters, out_filters, 3, stride, 1), nn.LeakyReLU(0.2, inplace=True),
              nn.Dropout2d(0.25)]
    if bn:
        layers.append(nn.BatchNorm2d(out_filters, 0.8))
    return nn.Sequential(*layers)


def discriminator_block2(in_filters, out_filters, bn=True):
    """Stride-1 variant (code/dmcnet_GAN/model.py:268-279)."""
    return discriminator_block(in_filters, out_filters, bn, stride=1)


class _DiscriminatorBase(nn.Module):
    """A chain of blocks and ``adv_layer``.  ``forced_masks`` (dict block name -> [N,C] keep mask
    already divided by 0.75) replaces the on-device Bernoulli draw; parity tests use it because
    CPU and GPU random streams differ."""

    WIDTHS = (16, 32, 64, 128)
    EXTRA = 0
    FLAT = 128 * 14 * 14

    def __init__(self, ch_in):
        super().__init__()
        self.block_names = []
        c = ch_in
        for stage, w in enumerate(self.WIDTHS, start=1):
            self._add("discriminator_block_%d" % stage, discriminator_block(c, w, bn=(stage > 1)))
            for j in range(2, 2 + self.EXTRA):
                self._add("discriminator_block_%d_%d" % (stage, j), discriminator_block2(w, w))
            c = w
        self.adv_layer = nn.Linear(self.FLAT, 2)
        self.forced_masks = None
        # OHWI weights (channels_last memory) are what the NHWC matrix-core kernels read; values and
        # state-dict keys are unchanged
        self.to(memory_format=torch.channels_last)

    def _add(self, name, block):
        self.add_module(name, block)
        self.block_names.append(name)

    def _keep(self, name, x, drop):
        if not self.training:
            return None
        if self.forced_masks is not None:
            return self.forced_masks[name].to(x.device, torch.float32)
        return (torch.rand((x.shape[0], getattr(self, name)[0].out_channels), device=x.device) >= drop.p).float() \
            / (1.0 - drop.p)

    def forward(self, x):
        """Blocks run on the fused HIP path (``ops.disc_block``: NHWC activations, convolution on the
        fp32 matrix cores with bias / LeakyReLU / mask / BatchNorm statistics in its epilogue) whenever
        the shape qualifies -- every block of Discriminator..Discriminator3 and Discriminator5 does;
        otherwise (Discriminator4's 8-channel blocks, CPU tensors) the convolution goes to
        PyTorch-ROCm and the tail to ``ops.disc_tail``."""
        nhwc = False
        # the Dropout2d keep masks of ALL blocks from one draw (same p everywhere, as the reference's blocks have): 4 launches for the
        # pass instead of 4 per block -- they are ~5 us kernels in front of every block of every discriminator pass
        keeps = None
        drops = [getattr(self, nm)[2] for nm in self.block_names]
        if self.training and self.forced_masks is None and all(d.p == drops[0].p for d in drops) and __import__("os").environ.get("DMC_DISC_ONE_DRAW", "1") != "0":
            widths = [getattr(self, nm)[0].out_channels for nm in self.block_names]
            n = x.shape[0]
            allk = (torch.rand(n * sum(widths), device=x.device) >= drops[0].p).float() / (1.0 - drops[0].p)
            keeps, c0 = {}, 0
            for nm, wd in zip(self.block_names, widths):
                keeps[nm] = allk[c0:c0 + n * wd].view(n, wd)       # (contiguous pieces of the one draw: no copies)
                c0 += n * wd
        for i, name in enumerate(self.block_names):
            blk = getattr(self, name)
            conv, drop = blk[0], blk[2]
            bn = blk[3] if len(blk) == 4 else None
            keep = keeps[name] if keeps is not None else self._keep(name, x, drop)
            first = i == 0 and not nhwc
            if ops.disc_block_supported(x, conv, first) and (bn is None or not first):
                x = ops.disc_block(x, conv, keep, bn, self.training, first=first)
                nhwc = True
                continue
            x = F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding)
            x = ops.disc_tail(x.contiguous(), keep, bn, self.training)
            nhwc = False
        # flatten in the reference's NCHW order (the Linear's weight is laid out for it).  An NHWC (channels_last) activation is
        # NOT copied into that order: the weight's columns are permuted to (h, w, c) instead -- 50 k values against 4.7 M (GAN) /
        # 9.6 M (I3D) per step each way; autograd carries the weight gradient back through the same permutation.
        if x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
            n, c, h, w = x.shape
            wp = self.adv_layer.weight.view(-1, c, h, w).permute(0, 2, 3, 1).reshape(self.adv_layer.out_features, -1)
            return F.linear(x.permute(0, 2, 3, 1).reshape(n, -1), wp, self.adv_layer.bias)
        return self.adv_layer(x.contiguous().reshape(x.shape[0], -1))


class Discriminator(_DiscriminatorBase):
    pass


class Discriminator2(_DiscriminatorBase):
    EXTRA = 1


class Discriminator3(_DiscriminatorBase):
    EXTRA = 2


class Discriminator4(_DiscriminatorBase):
    WIDTHS = (8, 16, 32)
    FLAT = 32 * 28 * 28


class Discriminator5(_DiscriminatorBase):
    EXTRA = 4


_DISCRIMINATORS = {c.__name__: c for c in (Discriminator, Discriminator2, Discriminator3,
                                           Discriminator4, Discriminator5)}
_ESTIMATORS = {"DenseNet": EstimatorDenseNet, "DenseNetSmall": EstimatorDenseNetSmall,
               "DenseNetTiny": EstimatorDenseNetTiny,
               "DenseNetTinyEarlyFusionSum": EstimatorDenseNetTinyEarlyFusionSum,
               "DenseNetTinyEarlyFusionStack": EstimatorDenseNetTinyEarlyFusionStack}


# ---------------------------------------------------------------------------------- Model

class Model(nn.Module):
    """TSN model over the DMC cue.

    ``arch_d=None``: the dmcnet variant -- ``forward(input_mv, input_residual)`` returns
    ``(base_out, gen_flow[, att_flow])`` and the classifier sees ``gen_flow.detach()``
    (code/dmcnet/model.py:330-357).  With ``arch_d`` set: the dmcnet_GAN variant --
    ``forward(input_mv, input_residual, input_flow=None)`` returns
    ``(base_out, validity, gen_flow[, att_flow])`` with no detach
    (code/dmcnet_GAN/model.py:533-566).
    """

    def __init__(self, num_class, num_segments, representation, base_model="resnet152",
                 new_length=1, use_databn=1, gen_flow_or_delta=0, gen_flow_ds_factor=0,
                 arch_estimator="ContextNetwork", arch_d=None, att=0, verbose=False,
                 channels_last=True):
        super().__init__()
        self._representation = representation
        self.num_segments = num_segments
        self.new_length = new_length
        self.use_databn = use_databn
        self.gen_flow_or_delta = gen_flow_or_delta
        self.gen_flow_ds_factor = gen_flow_ds_factor
        self.arch_estimator = arch_estimator
        self.arch_d = arch_d
        self.att = att
        if verbose:
            print("Initializing model: base model {}, representation {}, num_class {}, "
                  "num_segments {}, new_length {}".format(base_model, representation, num_class,
                                                          num_segments, new_length))
        self._prepare_base_model(base_model)
        self._prepare_tsn(num_class)
        if channels_last:
            # NHWC weights let MIOpen run its implicit-GEMM / Winograd kernels on the classifier
            # without per-call NCHW<->NHWC transposes (measured +12 % clips/s); values and
            # state-dict keys are unchanged
            self.base_model.to(memory_format=torch.channels_last)

    def _prepare_tsn(self, num_class):
        self.base_model.fc = nn.Linear(self.base_model.fc.in_features, num_class)
        if self._representation in ("mv", "flow"):
            self.base_model.conv1 = nn.Conv2d(2 * self.new_length, 64, kernel_size=(7, 7),
                                              stride=(2, 2), padding=(3, 3), bias=False)
            if self.use_databn == 1:
                self.data_bn = nn.BatchNorm2d(2)      # constructed but unused, as in the reference
        if self._representation == "residual" and self.use_databn == 1:
            self.data_bn = nn.BatchNorm2d(3)

    def _prepare_base_model(self, base_model):
        if "resnet" not in base_model:
            raise ValueError("Unknown base model: {}".format(base_model))
        self.base_model = resnet.build(base_model, pretrained=True)
        self._input_size = 224
        if self.arch_estimator == "ContextNetwork":
            cls = ContextNetworkAtt if self.att == 1 else ContextNetwork
            self.gen_flow_model = cls(5, True, self.gen_flow_ds_factor)
        elif self.arch_estimator in _ESTIMATORS:
            self.gen_flow_model = _ESTIMATORS[self.arch_estimator](5)
        else:
            raise ValueError("Unknown estimator: {}".format(self.arch_estimator))
        if self.gen_flow_ds_factor != 0:
            self.downsample = nn.AvgPool2d(self.gen_flow_ds_factor, stride=self.gen_flow_ds_factor)
        if self.arch_d is not None:
            if self.arch_d not in _DISCRIMINATORS:
                raise ValueError("Unknown discriminator: {}".format(self.arch_d))
            self.discriminator = _DISCRIMINATORS[self.arch_d](2)

    def forward(self, input_mv, input_residual, input_flow=None):
        if input_flow is not None and self.arch_d is None:
            raise TypeError("input_flow is only accepted by the GAN variant (arch_d set)")
        input_mv = input_mv.reshape((-1,) + tuple(input_mv.shape[-3:]))
        input_residual = input_residual.reshape((-1,) + tuple(input_residual.shape[-3:]))
        if self.gen_flow_ds_factor != 0:
            input_mv = self.downsample(input_mv)
            input_residual = self.downsample(input_residual)

        att_flow = None
        delta = self.gen_flow_or_delta == 1
        if isinstance(self.gen_flow_model, EstimatorDenseNetTiny):
            gen_flow = self.gen_flow_model.forward_mv_res(input_mv, input_residual, add_mv=delta)
        else:
            gen_flow = self.gen_flow_model(torch.cat((input_mv, input_residual), 1))
            if self.att == 1:
                gen_flow, att_flow = gen_flow
            if delta:
                gen_flow = torch.add(gen_flow, input_mv)
        if self.gen_flow_ds_factor != 0:
            gen_flow = gen_flow.repeat(1, 1, self.gen_flow_ds_factor, self.gen_flow_ds_factor)

        if self.arch_d is None:
            outputs = (self.base_model(gen_flow.detach()), gen_flow)
        else:
            d_in = gen_flow
            if input_flow is not None:          # first fake then real
                input_flow = input_flow.reshape((-1,) + tuple(input_flow.shape[-3:]))
                d_in = torch.cat((gen_flow, input_flow), 0)
            outputs = (self.base_model(gen_flow), self.discriminator(d_in), gen_flow)
        return outputs + ((att_flow,) if self.att == 1 else ())

    def forward_with_flow_mse(self, input_mv, input_residual, input_flow):
        """dmcnet variant only: ``forward(input_mv, input_residual)`` plus the reconstruction loss the reference
        computes right after it (``criterion_mse(gen_flow, input_flow)``, code/dmcnet/train.py:236,245) as a third
        output, reduced inside the generator's last kernel when the generator is the HIP DenseNetTiny at full
        resolution; any other configuration computes it with ops.flow_mse afterwards.  Same values either way."""
        if self.arch_d is not None or self.att == 1:
            raise TypeError("forward_with_flow_mse serves the dmcnet variant without attention")
        flow = input_flow.reshape((-1,) + tuple(input_flow.shape[-3:]))
        if isinstance(self.gen_flow_model, EstimatorDenseNetTiny) and self.gen_flow_ds_factor == 0:
            mv = input_mv.reshape((-1,) + tuple(input_mv.shape[-3:]))
            res = input_residual.reshape((-1,) + tuple(input_residual.shape[-3:]))
            gen_flow, loss_mse = self.gen_flow_model.forward_mv_res_mse(mv, res, flow, add_mv=self.gen_flow_or_delta == 1)
            return self.base_model(gen_flow.detach()), gen_flow, loss_mse
        base_out, gen_flow = self.forward(input_mv, input_residual)
        return base_out, gen_flow, ops.flow_mse(gen_flow, flow)

    @property
    def crop_size(self):
        return self._input_size

    @property
    def scale_size(self):
        return self._input_size * 256 // 224

    def get_augmentation(self):
        scales = [1, .875, .75] if self._representation in ("mv", "residual", "flow") \
            else [1, .875, .75, .66]
        return transforms.Compose([transforms.GroupMultiScaleCrop(self._input_size, scales),
                                   transforms.GroupRandomHorizontalFlip()])
