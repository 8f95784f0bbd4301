"""I3D variant (BASELINE config 5): the DMC generator applied per frame, an Inception-v1 I3D trunk
over the resulting cue, optional discriminator node.

Same module/attribute names and state-dict keys as the reference's
``code/dmcnet_I3D/network/i3d.py`` (``Unit3Dpy`` :328-403, ``MaxPool3dTFPadding`` :406-418,
``Mixed`` :421-455, ``I3D`` :458-601) and the loss assembly of
``code/dmcnet_I3D/train/model.py:135-188``.  The per-frame generator is the HIP path
(``EstimatorDenseNetTiny``); with a bf16 trunk (``trunk_dtype``) the stride-1 1x1x1 / 3x3x3 convolutions run on this
package's bf16 matrix-core kernels (``ops.conv3d_bf16``, NDHWC), the rest on PyTorch-ROCm (the generator stays fp32).
"""
import torch
from torch import nn
import torch.nn.functional as F

from . import model as _m
from . import ops

#: True (default): inside a bf16 trunk the stride-1 1x1x1 / 3x3x3 Unit3Dpy convolutions run on this package's
#: matrix-core kernels (csrc/conv3d_bf16.hip: NDHWC bf16, fp32 accumulate, deterministic weight gradient) instead of
#: MIOpen; the 2-channel 7x7x7 stem and the biased classifier head stay on PyTorch-ROCm.  DMC_OWN_CONV3D=0 switches off.
import os as _os
OWN_CONV3D = _os.environ.get("DMC_OWN_CONV3D", "1") != "0"


#: True (default): the branches of an Inception block run on concurrent HIP streams (Mixed.forward); DMC_I3D_BRANCH_STREAMS=0: one stream
BRANCH_STREAMS = _os.environ.get("DMC_I3D_BRANCH_STREAMS", "1") != "0"
#: an Inception block's branches write their channels straight into the block's output (DMC_I3D_JOIN_IN_PLACE=0: torch.cat)
JOIN_IN_PLACE = _os.environ.get("DMC_I3D_JOIN_IN_PLACE", "1") != "0"
#: the four data gradients of a block's input summed by one kernel (DMC_I3D_FANOUT_ADD=0: the autograd engine's three additions)
FANOUT_ADD = _os.environ.get("DMC_I3D_FANOUT_ADD", "1") != "0"
_SIDE_STREAMS = {}


def _side_streams(device):
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=device) for _ in range(3)]
    return _SIDE_STREAMS[key]


def _same_pad(kernel, stride):
    """TF 'SAME' padding per dimension (d, h, w) -> (front, back) pairs."""
    out = []
    for k, s in zip(kernel, stride):
        total = max(k - s, 0)
        out.append((total // 2, total - total // 2))
    return out


class Unit3Dpy(nn.Module):
    """Conv3d [+ BatchNorm3d] [+ ReLU] with TF-SAME padding; optional squeeze(H,W) + mean(T)."""

    def __init__(self, in_channels, out_channels, kernel_size=(1, 1, 1), stride=(1, 1, 1),
                 activation="relu", padding="SAME", use_bias=False, use_bn=True, squeeze=False,
                 mean=False):
        super().__init__()
        if padding not in ("SAME", "VALID"):
            raise ValueError("padding should be in [VALID|SAME] but got {}".format(padding))
        self.squeeze, self.mean, self.relu = squeeze, mean, activation is not None
        pad = 0
        self.pad = None
        if padding == "SAME":
            pads = _same_pad(kernel_size, stride)
            if all(p == pads[0] and p[0] == p[1] for p in pads):
                pad = pads[0][0]
            else:   # ConstantPad3d order: (w_l, w_r, h_t, h_b, d_f, d_b)
                self.pad = nn.ConstantPad3d(pads[2] + pads[1] + pads[0], 0)
        self.conv3d = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad,
                                bias=use_bias)
        if use_bn:
            self.batch3d = nn.BatchNorm3d(out_channels)
        self.use_bn = use_bn

    def forward(self, x, into=None):
        """``into`` (optional): a channel slice of a wider NDHWC tensor that may receive the result in place; whether it did is
        ``result.data_ptr() == into.data_ptr()`` (only the fused training-mode op does, Mixed.forward)."""
        c = self.conv3d
        if (OWN_CONV3D and self.pad is not None and self.use_bn and not self.squeeze and c.kernel_size == (7, 7, 7)
                and ops.stem3d_supported(x, c, self.batch3d)):
            # the 2-channel stem: forward on the bf16 matrix cores with the BatchNorm3d statistics in its epilogue
            return ops.stem3d_bn_relu(x, c, self.batch3d, self.relu)
        if self.pad is not None:
            x = self.pad(x)
        own = (OWN_CONV3D and self.pad is None and c.bias is None and x.is_cuda and x.dtype == torch.bfloat16
               and ops.conv3d_bf16_supported(x, c.weight, c.stride, c.padding))
        if own and self.use_bn and not self.squeeze and ops.conv_bn_relu3d_supported(x, c, self.batch3d):
            # conv (batch statistics in its epilogue) -> BatchNorm3d -> ReLU as one op on the bf16 kernels
            return ops.conv_bn_relu3d(x, c, self.batch3d, self.relu, into)
        if own:
            x = ops.conv3d_bf16(x, c.weight)            # bf16 NDHWC implicit GEMM on the matrix cores
        else:
            x = c(x)
        if self.use_bn:
            x = self.batch3d(x)
        if self.relu:
            x = F.relu(x)
        if self.squeeze:
            x = x.squeeze(3).squeeze(3)
            if self.mean:
                x = x.mean(2)
        return x


class MaxPool3dTFPadding(nn.Module):
    def __init__(self, kernel_size, stride=None, padding="SAME"):
        super().__init__()
        pads = _same_pad(kernel_size, stride)
        self.pad = nn.ConstantPad3d(pads[2] + pads[1] + pads[0], 0)
        self.pool = nn.MaxPool3d(kernel_size, stride, ceil_mode=True)
        self.kernel_size, self.stride = tuple(kernel_size), tuple(stride)

    def forward(self, x):
        if OWN_CONV3D and x.is_cuda and x.dtype == torch.bfloat16 and ops.maxpool3d_tf_supported(x, self.kernel_size, self.stride):
            return ops.maxpool3d_tf(x, self.kernel_size, self.stride)   # pad + pool in one NDHWC pass (csrc/pool3d_bf16.hip)
        return self.pool(self.pad(x))


class Mixed(nn.Module):
    """Inception block: 1x1 | 1x1-3x3 | 1x1-3x3 | pool-1x1, concatenated."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        o = out_channels
        self.branch_0 = Unit3Dpy(in_channels, o[0])
        self.branch_1 = nn.Sequential(Unit3Dpy(in_channels, o[1]), Unit3Dpy(o[1], o[2], (3, 3, 3)))
        self.branch_2 = nn.Sequential(Unit3Dpy(in_channels, o[3]), Unit3Dpy(o[3], o[4], (3, 3, 3)))
        self.branch_3 = nn.Sequential(MaxPool3dTFPadding((3, 3, 3), (1, 1, 1)),
                                      Unit3Dpy(in_channels, o[5]))

    def _branches(self):
        """(modules before the last unit, last unit) per branch"""
        return ((None, self.branch_0), (self.branch_1[0], self.branch_1[1]), (self.branch_2[0], self.branch_2[1]),
                (self.branch_3[0], self.branch_3[1]))

    def forward(self, x):
        own = OWN_CONV3D and x.is_cuda and x.dtype == torch.bfloat16
        # Each branch's last unit writes its channels straight into the block's output (ops.conv_bn_relu3d(..., into=slice)): no
        # torch.cat -- 9 concatenations of 4 strided copies each, 0.34 ms per micro-step on the chain between two blocks.
        # (Only the fused training-mode op writes in place; anything else falls back to the concatenation.)
        buf, slices = None, [None] * 4
        if own and JOIN_IN_PLACE and torch.is_grad_enabled() and x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d):
            widths = [u.conv3d.out_channels for _, u in self._branches()]
            if all(wd % 8 == 0 for wd in widths):
                buf = torch.empty((x.shape[0], sum(widths)) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device,
                                  memory_format=torch.channels_last_3d)
                c0 = 0
                for k, wd in enumerate(widths):
                    slices[k] = buf[:, c0:c0 + wd]
                    c0 += wd

        # the four consumers of x: their data gradients are summed in one pass (ops.fanout4) instead of three additions
        xs = ops.fanout4(x) if (own and FANOUT_ADD and torch.is_grad_enabled() and x.requires_grad) else (x, x, x, x)

        def run(k):
            head, last = self._branches()[k]
            return last(xs[k] if head is None else head(xs[k]), into=slices[k])

        if BRANCH_STREAMS and own:
            # The four branches are independent and, from mixed_4b on, small (9,408 or 1,176 pixels: every kernel a
            # fraction of the chip, bounded by its own latency): branches 1-3 run on side streams next to branch 0 and
            # join before the concatenation.  Autograd replays each node on its forward stream, so the backward
            # overlaps the same way.  Same kernels, same arithmetic, same results.
            main = torch.cuda.current_stream(x.device)
            sides = _side_streams(x.device)
            outs = [None] * 4
            fork = main.record_event()               # x is ready; the side pools' blocks are free of earlier main-stream readers
            outs[0] = run(0)                         # (autograd nodes are created in the one-stream order: the same gradient sums)
            for k, st in enumerate(sides):
                st.wait_event(fork)
                if buf is not None:
                    buf.record_stream(st)
                with torch.cuda.stream(st):
                    outs[k + 1] = run(k + 1)
            for st in sides:
                main.wait_stream(st)
        else:
            outs = [run(k) for k in range(4)]
        if buf is not None and all(o.data_ptr() == sl.data_ptr() and o.shape == sl.shape for o, sl in zip(outs, slices)):
            return ops.join_slices(buf, outs)
        return torch.cat(outs, 1)


_MIXED = (("mixed_3b", 192, (64, 96, 128, 16, 32, 32)), ("mixed_3c", 256, (128, 128, 192, 32, 96, 64)),
          ("mixed_4b", 480, (192, 96, 208, 16, 48, 64)), ("mixed_4c", 512, (160, 112, 224, 24, 64, 64)),
          ("mixed_4d", 512, (128, 128, 256, 24, 64, 64)), ("mixed_4e", 512, (112, 144, 288, 32, 64, 64)),
          ("mixed_4f", 528, (256, 160, 320, 32, 128, 128)), ("mixed_5b", 832, (256, 160, 320, 32, 128, 128)),
          ("mixed_5c", 832, (384, 192, 384, 48, 128, 128)))


class I3D(nn.Module):
    def __init__(self, num_classes, modality="rgb", dropout_prob=0, arch_estimator=None, arch_d=None,
                 name="inception", **kwargs):
        super().__init__()
        self.name, self.num_classes, self.modality = name, num_classes, modality
        in_channels = 2 if modality in ("flow", "mv", "flow+mp4") else 3
        self.arch_estimator, self.arch_d = arch_estimator, arch_d
        if arch_estimator in ("DenseNet", "DenseNetSmall", "DenseNetTiny"):
            self.gen_flow_model = _m._ESTIMATORS[arch_estimator](5)
        if arch_d is not None:
            if arch_d not in _m._DISCRIMINATORS:
                raise ValueError("Unknown discriminator: {}".format(arch_d))
            self.discriminator = _m._DISCRIMINATORS[arch_d](2)
        self.conv3d_1a_7x7 = Unit3Dpy(in_channels, 64, (7, 7, 7), (2, 2, 2))
        self.maxPool3d_2a_3x3 = MaxPool3dTFPadding((1, 3, 3), (1, 2, 2))
        self.conv3d_2b_1x1 = Unit3Dpy(64, 64)
        self.conv3d_2c_3x3 = Unit3Dpy(64, 192, (3, 3, 3))
        self.maxPool3d_3a_3x3 = MaxPool3dTFPadding((1, 3, 3), (1, 2, 2))
        for nm, cin, outs in _MIXED[:2]:
            setattr(self, nm, Mixed(cin, outs))
        self.maxPool3d_4a_3x3 = MaxPool3dTFPadding((3, 3, 3), (2, 2, 2))
        for nm, cin, outs in _MIXED[2:7]:
            setattr(self, nm, Mixed(cin, outs))
        self.maxPool3d_5a_2x2 = MaxPool3dTFPadding((2, 2, 2), (2, 2, 2))
        for nm, cin, outs in _MIXED[7:]:
            setattr(self, nm, Mixed(cin, outs))
        self.avg_pool = nn.AvgPool3d((2, 7, 7), (1, 1, 1))
        self.dropout = nn.Dropout(dropout_prob)
        self.conv3d_0c_1x1 = Unit3Dpy(1024, 400, activation=None, use_bias=True, use_bn=False,
                                      squeeze=True, mean=True)
        self.classifier = nn.Linear(400, num_classes)
        self.softmax = nn.Softmax(1)
        #: dtype the trunk runs in (torch.bfloat16 = autocast, as BASELINE config 5 asks)
        self.trunk_dtype = None

    _ORDER = ("conv3d_1a_7x7", "maxPool3d_2a_3x3", "conv3d_2b_1x1", "conv3d_2c_3x3",
              "maxPool3d_3a_3x3", "mixed_3b", "mixed_3c", "maxPool3d_4a_3x3", "mixed_4b", "mixed_4c",
              "mixed_4d", "mixed_4e", "mixed_4f", "maxPool3d_5a_2x2", "mixed_5b", "mixed_5c",
              "avg_pool", "conv3d_0c_1x1")

    def generate(self, inp):
        """[b,5,T,H,W] -> the DMC cue [b,2,T,H,W] (generator applied to every frame)."""
        b, c, t, h, w = inp.shape
        if isinstance(self.gen_flow_model, _m.EstimatorDenseNetTiny):
            # frame-major MV / residual planes straight from the clip-major input: two strided copies (5 channels once),
            # not a 5-channel frame-major copy that is then sliced and copied again
            ft = inp.transpose(1, 2)                                      # [b, T, 5, H, W] view
            g = self.gen_flow_model.forward_mv_res(ft[:, :, :2].reshape(-1, 2, h, w), ft[:, :, 2:].reshape(-1, c - 2, h, w))
        else:
            g = self.gen_flow_model(inp.transpose(1, 2).reshape(-1, c, h, w))
        return g.reshape(b, t, 2, h, w).transpose(1, 2)

    def forward(self, inp, node="logit", detach=False):
        if node == "D":
            return self.discriminator(inp)
        if self.arch_estimator in ("DenseNet", "DenseNetSmall", "DenseNetTiny"):
            inp = self.generate(inp)
        x = inp.detach() if detach else inp
        if self.trunk_dtype is not None and x.is_cuda:
            with torch.autocast("cuda", dtype=self.trunk_dtype), ops.batched_bn_counters():
                for nm in self._ORDER:
                    x = getattr(self, nm)(x)
            x = x.float()
        else:
            for nm in self._ORDER:
                x = getattr(self, nm)(x)
        out = self.classifier(self.dropout(x))
        if node == "flow+logit":
            return out, inp
        if node == "gen_flow":
            return inp
        return out


def i3d_losses(net, data, target, stage=None, detach=False):
    """Loss assembly of the reference's ``static_model.forward`` in training mode
    (code/dmcnet_I3D/train/model.py:135-188), 'flow+logit' node.  ``data`` [b,7,T,H,W].

    As written in the reference: channels [:5] feed the generator and channels [5:7] are the flow
    target, although its loader packs [flow2, mv2, res3] -- kept, not "fixed".
    Returns (logits, [loss_cls, mse] or [loss_cls, mse, loss_adv]).

    The three reductions run on this package's kernels (csrc/losses.hip: ``ops.consensus_ce`` with one segment =
    ``CrossEntropyLoss``, ``ops.flow_mse`` = ``MSELoss``), the MSE over the frame-major memory the generator wrote (the
    mean does not depend on the order; no [b,2,T,H,W] copy of the cue is made).  CUDA tensors only -- pinned by golden G11."""
    output, flow = net(data[:, :5], node="flow+logit", detach=detach)
    t = flow.size(2)
    h, w = flow.shape[-2:]
    fake = flow.transpose(1, 2).reshape(-1, 2, h, w)             # the generator's own [b*T,2,H,W] output, no copy
    real = data[:, 5:7].transpose(1, 2).reshape(-1, 2, h, w)
    losses = [ops.consensus_ce(output, target, 1)[0], ops.flow_mse(fake, real)]
    if stage is not None:
        valid = torch.ones(target.numel() * t, dtype=torch.int64, device=target.device)
        fake_lbl = torch.zeros_like(valid)
        validity = net(torch.cat((fake, real), 0), node="D")        # fake first, then real (:153-155)
        losses.append(ops.consensus_ce(validity, torch.cat((fake_lbl, valid), 0), 1)[0])
    return output, losses
