"""ResNet backbones for the classifier over the DMC cue.

The reference takes them from torchvision (``getattr(torchvision.models, base_model)``,
code/dmcnet/model.py:305), which is not available here; the published architecture is restated
with the same attribute names, so ``base_model.*`` state-dict keys match torchvision's.  The 3x3 /
1x1 convolutions run, with ``OWN_CONV`` (the default), together with their
BatchNorms on this package's NHWC matrix-core kernels (ops.conv_bn_act); BatchNorm / ReLU / add, the
stem tail and the stem's weight gradient are HIP kernels either way.
"""
import torch
from torch import nn

from . import ops


def _bn_act(bn, x, residual=None, relu=True):
    """relu?(bn(x) [+ residual]): the fused NHWC HIP kernel when the activation qualifies
    (channels_last fp32 on the GPU, train or eval), the stock modules otherwise."""
    if x.is_cuda and bn.track_running_stats and (bn.training or not torch.is_grad_enabled()) \
            and ops.bn_act_supported(x):
        return ops.bn_act(x, bn, residual, relu)
    y = bn(x)
    if residual is not None:
        y = y + residual
    return torch.relu(y) if relu else y

# True (default): the 3x3 / 1x1 convolutions run on this package's matrix-core NHWC kernels (fused conv -> bn op,
# deterministic, no MIOpen) whenever the activation qualifies -- in bf16x3 arithmetic (library option conv_arith = 1,
# the default: fp32 products from three bf16 slices, error <= the fp32 MFMA's) they are the faster path on MI355X
# (round 2, N = 120: 14.96 ms per step vs 15.97 with MIOpen's searched fp32 solvers; with conv_arith = 0, the fp32
# MFMA, 19.1 ms).  False = PyTorch-ROCm (MIOpen), which BASELINE config 2 allows.  bench.py --own-conv 0 /
# DMC_OWN_CONV=0 switch it off.
import os as _os
OWN_CONV = _os.environ.get("DMC_OWN_CONV", "1") != "0"
#: True: the last unit of layer1 / layer2 / layer3 also writes its fp32 result (by default the stride-2 block that follows reads
#: bf16x3 slices only and the fp32 memory of that tensor stays unwritten: nothing but this package's ops may read it)
KEEP_F32_OUTPUTS = _os.environ.get("DMC_KEEP_F32_OUTPUTS", "0") == "1"
# identity-shortcut blocks: residual gradient added in the first convolution's data-gradient epilogue (ops.ResidualGradLink)
RESIDUAL_GRAD_LINK = _os.environ.get("DMC_RESIDUAL_GRAD_LINK", "1") != "0"


def _same_bn_mode(bn, other):
    """True if ``other`` (the consumer's BatchNorm) will qualify for the same fused op as ``bn`` does now: the producer may
    then leave the fp32 form of its result unwritten.  A block whose BatchNorms are in mixed states (one in eval mode,
    momentum=None, affine=False ...) keeps the fp32 form and the consumer falls back to the stock modules."""
    return (other is not None and other.training == bn.training and other.track_running_stats and other.affine
            and other.momentum is not None)


def _conv_bn_act(conv, bn, x, residual=None, relu=True, link=None, next_conv=None, only_consumer=False, bn_link=None,
                 next_bn=None, next_unit=None):
    """relu?(bn(conv(x)) [+ residual]).  ``next_conv``: a convolution that reads the result -- when it takes the
    pre-split bf16x3 path (ops.x3s_usable) the result's slice tensor is written alongside; ``only_consumer``: nothing
    else reads the result, so its fp32 form is not written at all (only when ``next_bn``, the consumer's BatchNorm, is
    in the same mode as ``bn``: the consumer then takes the fused op that reads slices)."""
    train_op = OWN_CONV and x.is_cuda and ops.conv_bn_act_supported(x, conv, bn)
    eval_op = OWN_CONV and x.is_cuda and not train_op and ops.conv_bn_eval_supported(x, conv, bn)
    if train_op and next_unit is not None:
        # the result is the input of a stride-2 block: its conv1 and its shortcut convolution both read the space-to-depth
        # slice tensor (ops.conv_bn_s2_pair) and nothing reads the fp32 form
        n, _, h, w = x.shape
        k, s_, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        oshape = (n, conv.out_channels, (h + 2 * p - k) // s_ + 1, (w + 2 * p - k) // s_ + 1)
        if ops.s2_pair_usable(oshape, next_unit):
            return ops.conv_bn_act(x, conv, bn, residual, relu, link, want_f32=False, want_slices=True, s2d=True)
    if train_op or eval_op:
        slices = False
        if next_conv is not None:
            n, _, h, w = x.shape
            k, s_, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            slices = ops.x3s_usable(n, (h + 2 * p - k) // s_ + 1, (w + 2 * p - k) // s_ + 1, next_conv)
            only_consumer = only_consumer and _same_bn_mode(bn, next_bn)
        if eval_op:     # evaluation / validation: running statistics, forward only, the same convolution kernels
            return ops.conv_bn_act_eval(x, conv, bn, residual, relu, want_f32=not (slices and only_consumer), want_slices=slices)
        return ops.conv_bn_act(x, conv, bn, residual, relu, link, want_f32=not (slices and only_consumer), want_slices=slices,
                               bn_link=bn_link if slices else None)
    if not ops.f32_valid(x):
        raise RuntimeError("the activation's fp32 memory was not written (slices only) but the stock path needs it")
    return _bn_act(bn, conv(x), residual, relu)


_CFG = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
    "resnet152": ("bottleneck", (3, 8, 36, 3)),
}


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


class ResidualUnit(nn.Module):
    """BasicBlock (two 3x3) or Bottleneck (1x1, 3x3, 1x1 with 4x expansion)."""

    def __init__(self, kind, cin, planes, stride):
        super().__init__()
        self.kind = kind
        cout = planes * (4 if kind == "bottleneck" else 1)
        if kind == "basic":
            self.conv1, self.bn1 = _conv(cin, planes, 3, stride), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        else:
            self.conv1, self.bn1 = _conv(cin, planes, 1), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _conv(planes, planes, 3, stride), nn.BatchNorm2d(planes)
            self.conv3, self.bn3 = _conv(planes, cout, 1), nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))
        self.out_channels = cout
        self.next_conv = [None]      # the next unit's first convolution (set by ResNet; in a list: not a sub-module)
        self.next_identity = [False] # ... and whether that unit has an identity shortcut
        self.next_unit = [None]      # ... and the unit itself when it has a downsample branch (stride-2 pair path)

    def _next_unit(self):
        """The next unit when ITS input may exist as slices only (stride-2 pair path): then this unit's fp32 result is never
        written.  Not when someone else may read that result -- a forward hook on this unit, or KEEP_F32_OUTPUTS (feature taps,
        hooks on the enclosing Sequential, callers that do not go through ResNet.forward)."""
        return None if (KEEP_F32_OUTPUTS or self._forward_hooks) else self.next_unit[0]

    def forward(self, x):
        link = None
        if self.downsample is None:
            shortcut = x
            # identity shortcut: x feeds conv1 and the add; the two gradients are summed in conv1's data-gradient epilogue
            link = ops.ResidualGradLink() if RESIDUAL_GRAD_LINK and x.requires_grad else None
        elif OWN_CONV and x.is_cuda and ops.s2_pair_usable(tuple(x.shape), self) and (ops.x3q_of(x) is not None or ops.f32_valid(x)):
            # stride-2 block: conv1 and the shortcut convolution read the same input -- one launch per direction for the pair
            n, _, h, w = x.shape
            y, shortcut = ops.conv_bn_s2_pair(x, self, want_slices=ops.x3s_usable(n, h // 2, w // 2, self.conv2) and
                                              _same_bn_mode(self.bn1, self.bn2))
            return _conv_bn_act(self.conv2, self.bn2, y, residual=shortcut, next_conv=self.next_conv[0],
                                bn_link="block" if (self.next_identity[0] and RESIDUAL_GRAD_LINK) else None,
                                next_unit=self._next_unit())
        else:
            shortcut = _conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
        # conv1's result is read by conv2 alone: slices only when conv2 takes the pre-split path
        # (bn_link: conv2's data gradient is conv1-unit's whole output gradient; the block's output gradient is the next
        # block's conv1 data gradient + its residual gradient when that block has an identity shortcut -- ops.BnBwdLink)
        y = _conv_bn_act(self.conv1, self.bn1, x, link=link, next_conv=self.conv2, only_consumer=True, bn_link="inner",
                         next_bn=self.bn2)
        if self.kind == "basic":
            return _conv_bn_act(self.conv2, self.bn2, y, residual=shortcut, link=link, next_conv=self.next_conv[0],
                                bn_link="block" if (self.next_identity[0] and RESIDUAL_GRAD_LINK) else None,
                                next_unit=self._next_unit())
        y = _conv_bn_act(self.conv2, self.bn2, y, next_conv=self.conv3, only_consumer=True, next_bn=self.bn3)
        return _conv_bn_act(self.conv3, self.bn3, y, residual=shortcut, link=link, next_conv=self.next_conv[0])


class ResNet(nn.Module):
    def __init__(self, kind, depths, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        width = 64
        for stage, (planes, count) in enumerate(zip((64, 128, 256, 512), depths)):
            units = []
            for j in range(count):
                unit = ResidualUnit(kind, width, planes, 2 if (stage > 0 and j == 0) else 1)
                width = unit.out_channels
                units.append(unit)
            setattr(self, "layer%d" % (stage + 1), nn.Sequential(*units))
        chain = [u for st in (self.layer1, self.layer2, self.layer3, self.layer4) for u in st]
        for u, nxt in zip(chain, chain[1:]):
            u.next_conv[0] = nxt.conv1
            u.next_identity[0] = nxt.downsample is None
            u.next_unit[0] = nxt if (nxt.downsample is not None and kind == "basic") else None
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(width, num_classes)

    def _stem(self, x):
        # 2-channel flow input: own gradients for conv1 (MIOpen's degenerate with 2 input channels)
        c = self.conv1
        if (c.in_channels == 2 and c.bias is None and c.stride == (2, 2) and c.padding == (3, 3)
                and c.dilation == (1, 1) and c.groups == 1
                and (not torch.is_grad_enabled() or c.weight.requires_grad or x.requires_grad)
                and ops.stem_conv_supported(x, c.weight)):
            # training: bn1's batch statistics come out of the convolution's epilogue (consumed by _stem_tail's fused op)
            return ops.stem_conv(x, c.weight, want_stats=self.bn1.training and self.bn1.track_running_stats)
        return c(x)

    def _stem_tail(self, x):
        # bn1 -> relu -> maxpool(3, 2, 1): one fused HIP pass each way when the activation qualifies
        bn, mp = self.bn1, self.maxpool
        if (x.is_cuda and bn.track_running_stats and (bn.training or not torch.is_grad_enabled())
                and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1 and mp.dilation == 1
                and not mp.ceil_mode and not mp.return_indices and ops.bn_relu_pool_supported(x)):
            n, _, h, w = x.shape
            first = self.layer1[0].conv1
            return ops.bn_relu_pool(x, bn, want_slices=OWN_CONV and (bn.training or not torch.is_grad_enabled()) and
                                    ops.x3s_usable(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, first))
        return mp(_bn_act(bn, x))

    def forward(self, x):
        with ops.batched_bn_counters():
            x = self._stem_tail(self._stem(x))
            for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
                x = stage(x)
            return self.fc(torch.flatten(self.avgpool(x), 1))


def _torchvision_init(net):
    """torchvision's ResNet initialisation: kaiming_normal_(fan_out, relu) for every convolution,
    BatchNorm weight 1 / bias 0 (the Linear keeps nn.Linear's default), so that a from-scratch run
    starts from the same distribution as the upstream architecture."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0.0)


def build(name, pretrained=False, weights=None):
    """``pretrained`` mirrors the reference's call (code/dmcnet/model.py:305 always passes True):
    torchvision's ImageNet weights cannot be downloaded here, so ``pretrained=True`` without
    ``weights`` WARNS that the classifier starts from torchvision's random initialisation -- a
    different recipe from the reference's.  ``weights`` (a path to, or a dict of, a torchvision
    ``resnetNN`` state dict; also taken from $DMC_RESNET_WEIGHTS) loads them as
    ``pretrained=True`` would; ``fc`` / ``conv1`` mismatches are left to ``Model._prepare_tsn``,
    which replaces those layers anyway."""
    import os
    import warnings
    if name not in _CFG:
        raise ValueError("Unknown base model: {}".format(name))
    kind, depths = _CFG[name]
    net = ResNet(kind, depths)
    _torchvision_init(net)
    if weights is None:
        weights = os.environ.get("DMC_RESNET_WEIGHTS")
    if weights is not None:
        sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, bytes, os.PathLike)) else weights
        net.load_state_dict(sd, strict=True)
    elif pretrained:
        warnings.warn("%s: pretrained=True was requested (as the reference does) but no ImageNet weights "
                      "are available offline; the classifier starts from torchvision's random "
                      "initialisation. Pass weights=<torchvision state dict>, set DMC_RESNET_WEIGHTS, or "
                      "load a checkpoint with train.load_reference_weights." % name, stacklevel=2)
    return net
