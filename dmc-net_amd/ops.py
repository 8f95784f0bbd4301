"""torch.autograd bindings of the HIP kernels (device memory and streams are PyTorch's; the
arithmetic is libdmcnet_hip.so's).  Every op requires contiguous fp32 CUDA tensors and raises
otherwise -- there is no CPU path."""
import torch

from . import _lib

__all__ = ["gen_tiny", "flow_mse", "consensus_ce", "disc_tail", "bn_act", "bn_act_supported", "prepare_inputs",
           "u8_frames_buffer", "conv_nhwc", "conv_nhwc_supported", "disc_block", "disc_block_supported"]


class EventProbe(object):
    """HIP-event timing of individual C-ABI calls on the stream they are launched on (PyTorch's
    current stream).  bench.py installs one over the timed region to obtain the generator
    kernels' average duration; ``summary()`` synchronises."""

    def __init__(self, only=None):
        self.pairs = {}
        self.only = None if only is None else frozenset(only)   # restrict to these span names (fewer event records)

    def span(self, name):
        return _Span(self, name)

    def summary(self):
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v))
                for k, v in self.pairs.items() if v}


class _Span(object):
    def __init__(self, probe, name):
        self.probe, self.name = probe, name

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.probe.pairs.setdefault(self.name, []).append((self.a, self.b))


class _NoSpan(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


PROBE = None          # set to an EventProbe to time the calls below
#: test hook: a list that receives every discriminator block's z = keep * LeakyReLU(conv + bias) (the
#: tensor whose sign pattern decides the LeakyReLU branches), in call order; None = off
DEBUG_DISC_Z = None


def _span(name):
    if PROBE is None or (PROBE.only is not None and name not in PROBE.only):
        return _NoSpan()
    return PROBE.span(name)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The HIP stream PyTorch is launching on (every C-ABI call takes it).  Through the raw getter: 0.3 us instead of the
    9 us of ``torch.cuda.current_stream()`` (a Stream object per call) -- at 2-3 calls per op that was ~2 ms of host time
    per I3D micro-step, which is launch-bound."""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _lib._P(_RAW_STREAM(_CUR_DEVICE()))
    return _lib._P(torch.cuda.current_stream().cuda_stream)


def profile_mark():
    """Bracket a region in a rocprofv3 kernel trace (see tools/rocprof_region.py)."""
    _lib.check(_lib.load().dmc_profile_mark(_stream()), "dmc_profile_mark")


def _need_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.DmcHipError("the DMC-Net hot path runs on the HIP extension only: got a %s "
                                   "tensor (no CPU fallback)" % t.device)
        if t.dtype not in (torch.float32, torch.int64):
            raise _lib.DmcHipError("expected fp32 tensors, got %s" % t.dtype)


def _floats(nbytes, device):
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


class _GenTiny(torch.autograd.Function):
    """EstimatorDenseNetTiny(cat(mv, res)) [+ mv]; reference code/dmcnet/model.py:187-194,341-346."""

    @staticmethod
    def forward(ctx, mv, res, add_mv, grad_mode, *params):
        lib = _lib.load()
        _need_cuda(mv, res, *params)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            raise NotImplementedError("gradients w.r.t. the MV/residual inputs are not produced "
                                      "(the reference never requests them)")
        mv, res = mv.contiguous(), res.contiguous()
        ws, bs = [p.contiguous() for p in params[:6]], [p.contiguous() for p in params[6:]]
        n, _, h, w = mv.shape
        out = torch.empty((n, 2, h, w), dtype=torch.float32, device=mv.device)
        # inference (no parameter wants a gradient, or torch.no_grad()): the one-launch forward keeps no features -- 8 instead
        # of 120 B/px written.  needs_input_grad reflects the tensors' requires_grad only, NOT the grad mode (it reads True
        # under no_grad, and torch.is_grad_enabled() is always False in here): the caller passes the mode, `grad_mode`.
        keep = (grad_mode and any(ctx.needs_input_grad[4:])) or not ((lib.dmc_get_option(b"gen_fused") & 1) and w <= 224)
        saved = _floats(lib.dmc_gen_tiny_saved_bytes(n, h, w), mv.device) if keep else None
        work = _floats(lib.dmc_gen_tiny_workspace_bytes(), mv.device)
        with _span("gen_tiny_fwd"):
            _lib.check(lib.dmc_gen_tiny_fwd(_lib.ptr(mv), _lib.ptr(res), _lib.ptr_array(ws),
                                            _lib.ptr_array(bs), _lib.ptr(out), _lib.ptr(saved) if keep else None,
                                            _lib.ptr(work), n, h, w, int(add_mv), _stream()),
                       "dmc_gen_tiny_fwd")
        if not keep:
            return out
        ctx.save_for_backward(mv, res, saved, *ws)
        ctx.bias_like = [(b.shape, b.dtype) for b in bs]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        mv, res, saved = ctx.saved_tensors[:3]
        ws = list(ctx.saved_tensors[3:])
        grad_out = grad_out.contiguous()
        n, _, h, w = mv.shape
        dws = [torch.empty_like(x) for x in ws]
        dbs = [torch.empty(s, dtype=d, device=mv.device) for s, d in ctx.bias_like]
        gbuf = _floats(lib.dmc_gen_tiny_gbuf_bytes(n, h, w), mv.device)
        partials = _floats(lib.dmc_gen_tiny_partials_bytes(n, h, w), mv.device)
        work = _floats(lib.dmc_gen_tiny_workspace_bytes(), mv.device)
        with _span("gen_tiny_bwd"):
            _lib.check(lib.dmc_gen_tiny_bwd(_lib.ptr(mv), _lib.ptr(res), _lib.ptr_array(ws),
                                            _lib.ptr(saved), _lib.ptr(grad_out), _lib.ptr_array(dws),
                                            _lib.ptr_array(dbs), _lib.ptr(gbuf), _lib.ptr(partials),
                                            _lib.ptr(work), n, h, w, _stream()),
                       "dmc_gen_tiny_bwd")
        return (None, None, None, None) + tuple(dws) + tuple(dbs)


class _GenTinyMSE(torch.autograd.Function):
    """(gen_flow, MSELoss(gen_flow, flow)) with the loss reduced in the epilogue of the kernel that writes
    gen_flow: EstimatorDenseNetTiny(cat(mv, res)) [+ mv] (code/dmcnet/model.py:187-194,341-346) followed by
    ``criterion_mse(gen_flow, input_flow)`` (code/dmcnet/train.py:167,245).  Backward = the flow-MSE gradient
    (plus whatever else reaches gen_flow) through the generator's backward, as with the separate ops."""

    @staticmethod
    def forward(ctx, mv, res, flow, add_mv, *params):
        lib = _lib.load()
        _need_cuda(mv, res, flow, *params)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise NotImplementedError("no gradients w.r.t. the MV / residual / flow inputs")
        mv, res, flow = mv.contiguous(), res.contiguous(), flow.contiguous()
        ws, bs = [p.contiguous() for p in params[:6]], [p.contiguous() for p in params[6:]]
        n, _, h, w = mv.shape
        if tuple(flow.shape) != (n, 2, h, w):
            raise ValueError("flow target %s does not match the generator output %s" % (tuple(flow.shape), (n, 2, h, w)))
        out = torch.empty((n, 2, h, w), dtype=torch.float32, device=mv.device)
        loss = torch.empty((), dtype=torch.float32, device=mv.device)
        saved = _floats(lib.dmc_gen_tiny_saved_bytes(n, h, w), mv.device)
        work = _floats(lib.dmc_gen_tiny_workspace_bytes(), mv.device)
        part = _floats(lib.dmc_gen_tiny_mse_partials_bytes(), mv.device)
        with _span("gen_tiny_fwd"):
            _lib.check(lib.dmc_gen_tiny_fwd_mse(_lib.ptr(mv), _lib.ptr(res), _lib.ptr_array(ws), _lib.ptr_array(bs),
                                                _lib.ptr(flow), _lib.ptr(out), _lib.ptr(saved), _lib.ptr(work),
                                                _lib.ptr(loss), _lib.ptr(part), n, h, w, int(add_mv), _stream()),
                       "dmc_gen_tiny_fwd_mse")
        ctx.save_for_backward(mv, res, saved, out, flow, *ws)
        ctx.bias_like = [(b.shape, b.dtype) for b in bs]
        ctx.set_materialize_grads(False)
        return out, loss

    @staticmethod
    def backward(ctx, grad_out, grad_loss):
        lib = _lib.load()
        mv, res, saved, out, flow = ctx.saved_tensors[:5]
        ws = list(ctx.saved_tensors[5:])
        n, _, h, w = mv.shape
        g = None
        if grad_loss is not None:
            g = torch.empty_like(out)
            with _span("flow_mse_bwd"):
                _lib.check(lib.dmc_flow_mse_bwd(_lib.ptr(out), _lib.ptr(flow), _lib.ptr(grad_loss.contiguous().float()),
                                                _lib.ptr(g), out.numel(), _stream()), "dmc_flow_mse_bwd")
        if grad_out is not None:
            g = grad_out.contiguous() if g is None else g.add_(grad_out)
        if g is None:
            return (None,) * (4 + 12)
        dws = [torch.empty_like(x) for x in ws]
        dbs = [torch.empty(s, dtype=d, device=mv.device) for s, d in ctx.bias_like]
        gbuf = _floats(lib.dmc_gen_tiny_gbuf_bytes(n, h, w), mv.device)
        partials = _floats(lib.dmc_gen_tiny_partials_bytes(n, h, w), mv.device)
        work = _floats(lib.dmc_gen_tiny_workspace_bytes(), mv.device)
        with _span("gen_tiny_bwd"):
            _lib.check(lib.dmc_gen_tiny_bwd(_lib.ptr(mv), _lib.ptr(res), _lib.ptr_array(ws), _lib.ptr(saved),
                                            _lib.ptr(g), _lib.ptr_array(dws), _lib.ptr_array(dbs), _lib.ptr(gbuf),
                                            _lib.ptr(partials), _lib.ptr(work), n, h, w, _stream()),
                       "dmc_gen_tiny_bwd")
        return (None, None, None, None) + tuple(dws) + tuple(dbs)


def gen_tiny_mse(mv, res, flow, weights, biases, add_mv=False):
    """(gen_flow [N,2,H,W], mean((gen_flow - flow)^2)) in one forward launch sequence (see _GenTinyMSE)."""
    return _GenTinyMSE.apply(mv, res, flow, bool(add_mv), *weights, *biases)


def gen_tiny(mv, res, weights, biases, add_mv=False):
    """mv [N,2,H,W], res [N,3,H,W], 6 weights + 6 biases (reference layout) -> [N,2,H,W]."""
    return _GenTiny.apply(mv, res, bool(add_mv), torch.is_grad_enabled(), *weights, *biases)


class _FlowMSE(torch.autograd.Function):
    """nn.MSELoss()(gen_flow, input_flow); reference code/dmcnet/train.py:167,245."""

    @staticmethod
    def forward(ctx, gen_flow, flow):
        lib = _lib.load()
        _need_cuda(gen_flow, flow)
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("no gradient w.r.t. the flow target")
        if gen_flow.shape != flow.shape:
            raise ValueError("shape mismatch %s vs %s" % (tuple(gen_flow.shape), tuple(flow.shape)))
        gen_flow, flow = gen_flow.contiguous(), flow.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=gen_flow.device)
        partials = _floats(lib.dmc_flow_mse_partials_bytes(), gen_flow.device)
        with _span("flow_mse_fwd"):
            _lib.check(lib.dmc_flow_mse_fwd(_lib.ptr(gen_flow), _lib.ptr(flow), _lib.ptr(loss),
                                            _lib.ptr(partials), gen_flow.numel(), _stream()),
                       "dmc_flow_mse_fwd")
        ctx.save_for_backward(gen_flow, flow)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        gen_flow, flow = ctx.saved_tensors
        grad_loss = grad_loss.contiguous().float()
        grad = torch.empty_like(gen_flow)
        with _span("flow_mse_bwd"):
            _lib.check(lib.dmc_flow_mse_bwd(_lib.ptr(gen_flow), _lib.ptr(flow), _lib.ptr(grad_loss),
                                            _lib.ptr(grad), gen_flow.numel(), _stream()),
                       "dmc_flow_mse_bwd")
        return grad, None


def flow_mse(gen_flow, flow):
    return _FlowMSE.apply(gen_flow, flow)


class _ConsensusCE(torch.autograd.Function):
    """view(-1,S,C).mean(1) + CrossEntropyLoss; reference code/dmcnet/train.py:239-241."""

    @staticmethod
    def forward(ctx, logits, target, num_segments):
        lib = _lib.load()
        _need_cuda(logits, target)
        if logits.dtype != torch.float32 or target.dtype != torch.int64:
            raise _lib.DmcHipError("consensus_ce expects fp32 logits and int64 class labels, got %s / %s"
                                   % (logits.dtype, target.dtype))
        logits, target = logits.contiguous(), target.contiguous()
        n, c = logits.shape
        if n % num_segments != 0 or target.numel() * num_segments != n:
            raise ValueError("logits %s do not match %d targets x %d segments"
                             % (tuple(logits.shape), target.numel(), num_segments))
        b = n // num_segments
        consensus = torch.empty((b, c), dtype=torch.float32, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        grad = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        _lib.check(lib.dmc_consensus_ce_fwd_bwd(_lib.ptr(logits), _lib.ptr(target),
                                                _lib.ptr(consensus), _lib.ptr(loss), _lib.ptr(grad),
                                                b, num_segments, c, _stream()),
                   "dmc_consensus_ce_fwd_bwd")
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(consensus)
        return loss, consensus

    @staticmethod
    def backward(ctx, grad_loss, _grad_consensus):
        (grad,) = ctx.saved_tensors
        return grad * grad_loss, None, None


def consensus_ce(logits, target, num_segments):
    """Returns (loss, consensus logits [B, C])."""
    return _ConsensusCE.apply(logits, target, num_segments)


class _DiscTail(torch.autograd.Function):
    """LeakyReLU(0.2) -> Dropout2d keep-mask -> BatchNorm2d; reference
    code/dmcnet_GAN/model.py:254-279."""

    @staticmethod
    def forward(ctx, x, keep, gamma, beta, running_mean, running_var, training, eps, momentum):
        lib = _lib.load()
        _need_cuda(x, keep, gamma, beta)
        x = x.contiguous()
        n, c, h, w = x.shape
        use_bn = gamma is not None
        keep = keep.contiguous() if keep is not None else None
        y = torch.empty_like(x)
        stats = _floats(lib.dmc_disc_tail_stats_bytes(c), x.device) if use_bn else None
        _lib.check(lib.dmc_disc_tail_fwd(_lib.ptr(x), _lib.ptr(keep), _lib.ptr(gamma), _lib.ptr(beta),
                                         _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(y),
                                         _lib.ptr(stats), n, c, h, w, int(use_bn), int(training),
                                         float(eps), float(momentum), _stream()),
                   "dmc_disc_tail_fwd")
        ctx.save_for_backward(x, keep, gamma, stats)
        ctx.use_bn = use_bn
        ctx.training = bool(training)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, keep, gamma, stats = ctx.saved_tensors
        if ctx.use_bn and not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented")
        dy = dy.contiguous()
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma) if ctx.use_bn else None
        dbeta = torch.empty_like(gamma) if ctx.use_bn else None
        _lib.check(lib.dmc_disc_tail_bwd(_lib.ptr(x), _lib.ptr(keep), _lib.ptr(gamma), _lib.ptr(stats),
                                         _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta),
                                         n, c, h, w, int(ctx.use_bn), _stream()),
                   "dmc_disc_tail_bwd")
        return dx, None, dgamma, dbeta, None, None, None, None, None


def disc_tail(x, keep, bn, training):
    """x: conv output; keep: [N,C] mask/(1-p) or None; bn: nn.BatchNorm2d or None."""
    if bn is None:
        return _DiscTail.apply(x, keep, None, None, None, None, training, 0.0, 0.0)
    if training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _DiscTail.apply(x, keep, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                           bn.eps, bn.momentum)


class _BnAct(torch.autograd.Function):
    """act(BatchNorm2d(x) [+ residual]) on channels_last tensors; reference: the BatchNorm /
    ReLU / residual-add modules of torchvision's ResNet (code/dmcnet/model.py:305,352)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, relu, training, eps,
                momentum):
        lib = _lib.load()
        _need_cuda(x, residual, gamma, beta)
        n, c, h, w = x.shape
        m = n * h * w
        y = torch.empty_like(x)           # keeps the channels_last strides
        stats = _floats(lib.dmc_bn_act_stats_bytes(c), x.device)      # (mean, invstd): kept for backward
        # reduction workspace (16 KB x C): transient -- training forwards only, never saved
        scratch = _floats(lib.dmc_bn_act_scratch_bytes(c), x.device) if training else None
        # with a residual the ReLU sign depends on it: keep 4 sign bits per float4 (1/16 of the
        # residual's size) so that the backward neither re-reads nor even keeps the residual
        mask = None
        if relu and training and residual is not None:
            mask = torch.empty(m * (c // 4), dtype=torch.uint8, device=x.device)
        with _span("bn_act_fwd"):
            _lib.check(lib.dmc_bn_act_fwd(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(gamma),
                                          _lib.ptr(beta), _lib.ptr(running_mean),
                                          _lib.ptr(running_var), _lib.ptr(y), _lib.ptr(stats),
                                          _lib.ptr(scratch), _lib.ptr(mask), m, c, int(relu), int(training), float(eps),
                                          float(momentum), _stream()), "dmc_bn_act_fwd")
        ctx.save_for_backward(x, residual if (relu and mask is None) else None, gamma, beta, stats, mask)
        ctx.relu, ctx.training, ctx.has_res = bool(relu), bool(training), residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, residual, gamma, beta, stats, mask = ctx.saved_tensors
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented")
        n, c, h, w = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        want_dres = ctx.has_res and ctx.needs_input_grad[1]
        # without ReLU the residual's gradient is dy itself
        dres = torch.empty_like(x) if (want_dres and ctx.relu) else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        scratch = _floats(lib.dmc_bn_act_scratch_bytes(c), x.device)
        with _span("bn_act_bwd"):
            _lib.check(lib.dmc_bn_act_bwd(_lib.ptr(x), _lib.ptr(residual), _lib.ptr(gamma),
                                          _lib.ptr(beta), _lib.ptr(stats), _lib.ptr(scratch), _lib.ptr(dy), _lib.ptr(dx),
                                          _lib.ptr(dres), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(mask),
                                          n * h * w, c, int(ctx.relu), _stream()), "dmc_bn_act_bwd")
        if want_dres and not ctx.relu:
            dres = dy
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None


def bn_act_supported(x):
    """True if the fused NHWC kernel handles this activation (else use the stock modules)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if not x.is_contiguous(memory_format=torch.channels_last):
        return False
    n, c, h, w = x.shape
    return bool(_lib.load().dmc_bn_act_supported(n * h * w, c))


_PENDING_COUNTERS = None


class batched_bn_counters:
    """Context manager: inside it ``bn_act`` defers ``num_batches_tracked += 1`` and one
    multi-tensor add bumps every counter on exit (20 launches per ResNet-18 forward otherwise)."""

    def __enter__(self):
        global _PENDING_COUNTERS
        self._outer = _PENDING_COUNTERS
        _PENDING_COUNTERS = []
        return self

    def __exit__(self, *exc):
        global _PENDING_COUNTERS
        pending, _PENDING_COUNTERS = _PENDING_COUNTERS, self._outer
        if pending and exc[0] is None:
            torch._foreach_add_(pending, 1)
        return False


def bn_act(x, bn, residual=None, relu=True):
    """relu?(bn(x) [+ residual]) for a channels_last ``x`` and an ``nn.BatchNorm2d`` ``bn``."""
    training = bn.training
    if training and bn.num_batches_tracked is not None:
        if _PENDING_COUNTERS is not None:
            _PENDING_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    if residual is not None:
        residual = residual.contiguous(memory_format=torch.channels_last)
    return _BnAct.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, relu,
                        training, bn.eps, bn.momentum if bn.momentum is not None else 0.1)


#: True (default): the stem tail's forward records each pooling window's arg-max (position + raw input value, +120 MB at
#: 120 frames) and the backward's BatchNorm sums stream over that record (dmc_bn_relu_pool_bwd_arg); DMC_POOL_ARGMAX=0:
#: the backward recomputes the arg-max from the input (dmc_bn_relu_pool_bwd)
POOL_ARGMAX = __import__("os").environ.get("DMC_POOL_ARGMAX", "1") != "0"


class _BnReluPool(torch.autograd.Function):
    """maxpool3x3s2p1(relu(bn(x))) of the ResNet stem (torchvision's bn1 / relu / maxpool behind
    code/dmcnet/model.py:305,352) on a channels_last ``x``; the rectified tensor is never stored."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, eps, momentum, want_slices=False, stat_partials=None):
        lib = _lib.load()
        _need_cuda(x, gamma, beta)
        ctx.set_materialize_grads(False)
        n, c, h, w = x.shape
        ph, pw = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, c, ph, pw), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last)
        stats = _floats(lib.dmc_bn_act_stats_bytes(c), x.device)
        stat_split = 0
        if training and stat_partials is not None and want_slices and c % 16 == 0:
            scratch, stat_split = stat_partials     # reduced by x's producer (dmc_stem_fwd_x3_stats)
        else:
            scratch = _floats(lib.dmc_bn_act_scratch_bytes(c), x.device) if training else None
        ys = codes = xmax = None
        with _span("bn_relu_pool_fwd"):
            if want_slices and c % 16 == 0:     # the pooled map also as a bf16x3 slice tensor (layer1's first convolution)
                ys = torch.empty(lib.dmc_x3s_slices_bytes(n * ph * pw, c), dtype=torch.uint8, device=x.device)
                if training and POOL_ARGMAX and any(ctx.needs_input_grad[:3]):
                    # arg-max record for the backward: its BatchNorm sums then stream over pooled-size tensors
                    codes = _floats(lib.dmc_bn_relu_pool_codes_bytes(n, h, w, c), x.device)
                    xmax = torch.empty_like(y)
                _lib.check(lib.dmc_bn_relu_pool_fwd_arg(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                                        _lib.ptr(running_var), _lib.ptr(y), _lib.ptr(ys), _lib.ptr(codes),
                                                        _lib.ptr(xmax), _lib.ptr(stats), _lib.ptr(scratch), int(stat_split), n, h, w, c,
                                                        int(training), float(eps), float(momentum), _stream()),
                           "dmc_bn_relu_pool_fwd_arg")
            else:
                _lib.check(lib.dmc_bn_relu_pool_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta),
                                                    _lib.ptr(running_mean), _lib.ptr(running_var),
                                                    _lib.ptr(y), _lib.ptr(stats), _lib.ptr(scratch), n, h, w, c, int(training),
                                                    float(eps), float(momentum), _stream()),
                           "dmc_bn_relu_pool_fwd")
        if codes is not None:
            ctx.save_for_backward(x, gamma, beta, stats, codes, xmax)
        else:
            ctx.save_for_backward(x, gamma, beta, stats)
        ctx.training = bool(training)
        if ys is None:
            ys = torch.empty(0, dtype=torch.uint8, device=x.device)
        ctx.mark_non_differentiable(ys)
        return y, ys

    @staticmethod
    def backward(ctx, dy, _dys):
        lib = _lib.load()
        x, gamma, beta, stats = ctx.saved_tensors[:4]
        if dy is None:
            return (None,) * 10
        if not ctx.training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented")
        n, c, h, w = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        scratch = _floats(lib.dmc_bn_act_scratch_bytes(c), x.device)
        with _span("bn_relu_pool_bwd"):
            if len(ctx.saved_tensors) == 6:
                codes, xmax = ctx.saved_tensors[4:]
                _lib.check(lib.dmc_bn_relu_pool_bwd_arg(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                                        _lib.ptr(scratch), _lib.ptr(dy), _lib.ptr(codes), _lib.ptr(xmax),
                                                        _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), n, h, w, c, _stream()),
                           "dmc_bn_relu_pool_bwd_arg")
            else:
                codes = _floats(lib.dmc_bn_relu_pool_codes_bytes(n, h, w, c), x.device)
                _lib.check(lib.dmc_bn_relu_pool_bwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta),
                                                    _lib.ptr(stats), _lib.ptr(scratch), _lib.ptr(dy), _lib.ptr(dx),
                                                    _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(codes),
                                                    n, h, w, c, _stream()), "dmc_bn_relu_pool_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def bn_relu_pool_supported(x):
    """True if the fused stem kernel handles this activation (channels_last fp32 on the GPU)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if not x.is_contiguous(memory_format=torch.channels_last):
        return False
    n, c, h, w = x.shape
    return bool(_lib.load().dmc_bn_relu_pool_supported(n, h, w, c))


def bn_relu_pool(x, bn, want_slices=False):
    """maxpool(3, 2, 1)(relu(bn(x))) for a channels_last ``x`` and an ``nn.BatchNorm2d`` ``bn``; ``want_slices``: the
    pooled map's bf16x3 slice tensor is attached to the result (x3s_of) for a pre-split convolution."""
    training = bn.training
    if training and bn.num_batches_tracked is not None:
        if _PENDING_COUNTERS is not None:
            _PENDING_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    y, ys = _BnReluPool.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps,
                              bn.momentum if bn.momentum is not None else 0.1, bool(want_slices),
                              getattr(x, "_dmc_stat_partials", None))
    if ys.numel():
        y._dmc_x3s = ys
        y._dmc_f32 = True
    return y


#: True (default): conv1's forward on dmc_stem_fwd (exact fp32 MFMA, no LDS); False = F.conv2d (MIOpen)
STEM_FWD_HIP = __import__("os").environ.get("DMC_STEM_FWD", "1") != "0"


#: True (default): conv1's data gradient (GAN variant) on dmc_stem_dgrad; DMC_STEM_DGRAD=0: library GEMM + col2im
STEM_DGRAD_HIP = __import__("os").environ.get("DMC_STEM_DGRAD", "1") != "0"


class _StemConv(torch.autograd.Function):
    """conv1 of the classifier for the 2-channel flow input (code/dmcnet/model.py:285-294): the
    forward convolution is dmc_stem_fwd (fp32 MFMA with the weights resident in registers); with 2 input
    channels MIOpen's implicit-GEMM *gradients* degenerate (weight gradient 0.80 ms, data gradient 2.1 ms at
    120 frames), so
      * the weight gradient is dmc_stem_wgrad (0.24 ms, deterministic);
      * the data gradient (needed only by the GAN variant, whose classifier loss reaches the
        generator) is dmc_stem_dgrad (one wave per input row: a [OW x 14] row GEMM over the 3-4 window rows that
        reach it, in bf16x3 arithmetic, and a 1-D fold); frames wider than 256 fall back to a batched GEMM
        W^T[98,64] x dy[64, OH*OW] followed by ``fold`` (col2im)."""

    @staticmethod
    def forward(ctx, x, weight, want_stats=False):
        xc = x.detach().contiguous()
        stat = None
        if STEM_FWD_HIP:
            n, _, h, w = xc.shape
            y = torch.empty((n, 64, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device,
                            memory_format=torch.channels_last)
            so, si, sy, sx = weight.stride()
            lib = _lib.load()
            with _span("stem_fwd"):
                if lib.dmc_get_option(b"conv_arith") == 1:         # bf16x3 arithmetic, as the other classifier convolutions
                    work = _floats(lib.dmc_stem_fwd_x3_workspace_bytes(n, h, w), x.device)
                    if want_stats:      # bn1's batch statistics from the accumulators: partial sums for the stem tail
                        stat = _floats(lib.dmc_bn_act_scratch_bytes(64), x.device)
                    _lib.check(lib.dmc_stem_fwd_x3_stats(_lib.ptr(xc), _lib.ptr(weight), so, si, sy, sx, _lib.ptr(work), _lib.ptr(y),
                                                         _lib.ptr(stat), n, h, w, _stream()), "dmc_stem_fwd_x3_stats")
                else:
                    _lib.check(lib.dmc_stem_fwd(_lib.ptr(xc), _lib.ptr(weight), so, si, sy, sx, _lib.ptr(y), n, h, w,
                                                _stream()), "dmc_stem_fwd")
        else:
            y = torch.nn.functional.conv2d(x, weight, None, 2, 3)
        ctx.save_for_backward(xc, weight)
        if stat is None:
            stat = torch.empty(0, dtype=torch.float32, device=x.device)
        ctx.mark_non_differentiable(stat)
        return y, stat

    @staticmethod
    def backward(ctx, dy, _dstat=None):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        n, _, h, w = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dw = dx = None
        if ctx.needs_input_grad[1]:
            def launch():
                dw = torch.empty((64, 2, 7, 7), dtype=torch.float32, device=x.device)
                partials = _floats(lib.dmc_stem_wgrad_partials_bytes(n, h, w), x.device)
                _lib.check(lib.dmc_stem_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(partials),
                                              n, h, w, _stream()), "dmc_stem_wgrad")
                if weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous():
                    dw = dw.contiguous(memory_format=torch.channels_last)
                return dw
            with _span("stem_wgrad"):
                dw = _on_wgrad_stream(weight, (x, dy), launch)
        if ctx.needs_input_grad[0]:
            if STEM_DGRAD_HIP and dy.dtype == torch.float32 and lib.dmc_stem_dgrad_supported(h, w):
                # row GEMM + 1-D fold in one kernel, bf16x3 arithmetic (dmc_stem_dgrad): no [pixels][98] column matrix
                dx = torch.empty((n, 2, h, w), dtype=torch.float32, device=x.device)
                work = _floats(lib.dmc_stem_dgrad_workspace_bytes(), x.device)
                so, si, sy, sx = weight.stride()
                with _span("stem_dgrad"):
                    _lib.check(lib.dmc_stem_dgrad(_lib.ptr(dy), _lib.ptr(weight), so, si, sy, sx, _lib.ptr(work), _lib.ptr(dx),
                                                  n, h, w, _stream()), "dmc_stem_dgrad")
            else:
                oh, ow = dy.shape[2], dy.shape[3]
                g = dy.permute(0, 2, 3, 1).reshape(n, oh * ow, 64)            # a view of the NHWC storage
                cols = torch.matmul(weight.reshape(64, 98).t(), g.transpose(1, 2))   # [N, 98, OH*OW]
                dx = torch.nn.functional.fold(cols, (h, w), kernel_size=7, padding=3, stride=2)
        return dx, dw, None


def stem_conv_supported(x, weight):
    """True when ``conv2d(x, weight, stride 2, padding 3)`` can take the HIP weight gradient."""
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
            and tuple(weight.shape) == (64, 2, 7, 7) and weight.dtype == torch.float32
            and x.shape[1] == 2
            and bool(_lib.load().dmc_stem_wgrad_supported(int(x.shape[2]), int(x.shape[3]))))


#: True (default): conv1's forward also reduces bn1's batch statistics from its accumulators (dmc_stem_fwd_x3_stats)
STEM_STATS = __import__("os").environ.get("DMC_STEM_STATS", "1") != "0"


def stem_conv(x, weight, want_stats=False):
    """conv1(x) for ``x`` [N,2,H,W] and ``weight`` [64,2,7,7].  ``want_stats``: the batch statistics' partial sums of the
    result are attached to it for ``bn_relu_pool`` (training mode: the BatchNorm pass over the convolution output goes)."""
    _need_cuda(x, weight)
    y, stat = _StemConv.apply(x, weight, bool(want_stats) and STEM_STATS)
    if stat.numel():
        lib = _lib.load()
        y._dmc_stat_partials = (stat, lib.dmc_stem_fwd_x3_stat_blocks(int(x.shape[0]), int(x.shape[2]), int(x.shape[3])))
    return y


# ------------------------------------------------------------------ NHWC convolutions (conv_nhwc.hip)
_CL = torch.channels_last


def _is_cl(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=_CL)


def conv_nhwc_supported(x, weight, stride, padding):
    """True if ``conv2d(x, weight, stride=stride, padding=padding)`` can run on the HIP implicit-GEMM
    kernels (fp32, channels_last x, 3x3/pad 1 or 1x1/pad 0, stride 1 or 2, channels multiples of 16)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dtype == torch.float32):
        return False
    n, cin, h, w = x.shape
    cout, cin2, kh, kw = weight.shape
    if cin2 != cin:
        return False
    return bool(_lib.load().dmc_conv_nhwc_supported(n, h, w, cin, cout, kh, kw, int(stride), int(padding)))


def _conv_geom(x, weight, stride, padding):
    n, cin, h, w = x.shape
    cout, _, kh, kw = weight.shape
    oh, ow = (h + 2 * padding - kh) // stride + 1, (w + 2 * padding - kw) // stride + 1
    return n, h, w, cin, cout, kh, kw, oh, ow


def _conv_fwd(x, weight, bias, keep, stride, padding, act, want_stats, presplit=None):
    lib = _lib.load()
    n, h, w, cin, cout, kh, kw, oh, ow = _conv_geom(x, weight, stride, padding)
    y = torch.empty((n, cout, oh, ow), dtype=torch.float32, device=x.device, memory_format=_CL)
    part, nblk = None, 0
    if want_stats:
        nblk = lib.dmc_conv_nhwc_stat_blocks(n, h, w, cin, cout, kh, stride, padding)
        part = torch.empty((nblk, cout, 2), dtype=torch.float64, device=x.device)
    wpack = presplit if presplit is not None else _floats(lib.dmc_conv_nhwc_wt_bytes(cin, cout, kh, kw), x.device)
    wptr = None if presplit is not None else _lib.ptr(weight)      # NULL: wpack already holds the slices
    _lib.check(lib.dmc_conv_nhwc_fwd(_lib.ptr(x), wptr, _lib.ptr(wpack), _lib.ptr(bias), _lib.ptr(keep), _lib.ptr(y),
                                     _lib.ptr(part), nblk, n, h, w, cin, cout, kh, kw, stride, padding, int(act),
                                     _stream()), "dmc_conv_nhwc_fwd")
    return y, part, nblk


def _conv_dgrad(dy, weight, x_shape, stride, padding, presplit=None, addend=None):
    lib = _lib.load()
    n, cin, h, w = x_shape
    cout, _, kh, kw = weight.shape
    dx = torch.empty((n, cin, h, w), dtype=torch.float32, device=dy.device, memory_format=_CL)
    wt = presplit if presplit is not None else _floats(lib.dmc_conv_nhwc_wt_bytes(cin, cout, kh, kw), dy.device)
    wptr = None if presplit is not None else _lib.ptr(weight)      # NULL: wt already holds the transposed slices
    if addend is not None:                      # dx = data gradient + addend, in the convolution's epilogue
        addend = _as_cl(addend)
        _lib.check(lib.dmc_conv_nhwc_dgrad_add(_lib.ptr(dy), wptr, _lib.ptr(wt), _lib.ptr(addend), _lib.ptr(dx), n, h, w,
                                               cin, cout, kh, kw, stride, padding, _stream()), "dmc_conv_nhwc_dgrad_add")
        return dx
    _lib.check(lib.dmc_conv_nhwc_dgrad(_lib.ptr(dy), wptr, _lib.ptr(wt), _lib.ptr(dx), n, h, w, cin,
                                       cout, kh, kw, stride, padding, _stream()), "dmc_conv_nhwc_dgrad")
    return dx


def _conv_wgrad(x, dy, weight, stride, padding):
    lib = _lib.load()
    n, cin, h, w = x.shape
    cout, _, kh, kw = weight.shape
    dw = torch.empty((cout, cin, kh, kw), dtype=torch.float32, device=x.device, memory_format=_CL)
    work = _floats(lib.dmc_conv_nhwc_wgrad_bytes(n, h, w, cin, cout, kh, kw, stride, padding), x.device)
    _lib.check(lib.dmc_conv_nhwc_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(work), n, h, w, cin, cout,
                                       kh, kw, stride, padding, _stream()), "dmc_conv_nhwc_wgrad")
    return dw


def _as_cl(t):
    return t if _is_cl(t) else t.contiguous(memory_format=_CL)


def _grad_like(dw, weight):
    """dw (channels_last memory) in the strides of ``weight`` (no copy when weight is channels_last)."""
    return dw if _is_cl(weight) and weight.stride() == dw.stride() else dw.contiguous().view_as(weight).contiguous()


class _ConvNHWC(torch.autograd.Function):
    """nn.Conv2d(bias=False) on the fp32 matrix cores: the ResNet's 3x3 / 1x1 convolutions
    (torchvision BasicBlock behind code/dmcnet/model.py:305,352)."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        _need_cuda(x, weight)
        x, wcl = _as_cl(x), _as_cl(weight)
        with _span("conv_nhwc_fwd"):
            y, _, _ = _conv_fwd(x, wcl, None, None, stride, padding, 0, False)
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, padding)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, padding = ctx.geom
        dy, wcl = _as_cl(dy), _as_cl(weight)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            with _span("conv_nhwc_dgrad"):
                dx = _conv_dgrad(dy, wcl, x.shape, stride, padding)
        if ctx.needs_input_grad[1]:
            with _span("conv_nhwc_wgrad"):
                dw = _grad_like(_conv_wgrad(x, dy, wcl, stride, padding), weight)
        return dx, dw, None, None


def conv_nhwc(x, weight, stride=1, padding=1):
    """conv2d(x, weight, None, stride, padding) for a channels_last ``x`` (see conv_nhwc_supported)."""
    return _ConvNHWC.apply(x, weight, int(stride), int(padding))



# ------------------------------------------------------------------ pre-split bf16x3 operands (conv_x3s.hip)
#: True (default): stride-1 3x3 convolutions of the fused conv -> bn op run on conv_x3s.hip -- both operands arrive as
#: bf16x3 slice tensors written by the PRODUCER of the activation (BatchNorm apply / stem pool / BatchNorm backward), the
#: convolution's main loop is LDS reads + MFMAs only.  DMC_X3S=0: the in-loop-split kernels of conv_nhwc.hip.
X3S = __import__("os").environ.get("DMC_X3S", "1") != "0"
#: DMC_BN_BWD_LINK=1: a pre-split data gradient also reduces the BatchNorm-backward sums of the unit it feeds (BnBwdLink).
#: Off by default: measured neutral (13.28 vs 13.27 ms per step, two A/B pairs on one box) -- the 0.33 ms of bn_partial
#: launches it removes come back as un-overlapped epilogue time of the data-gradient kernels.
BN_BWD_LINK = __import__("os").environ.get("DMC_BN_BWD_LINK", "0") == "1"


def x3s_of(t):
    """The slice tensor attached to activation ``t`` by its producer (or None)."""
    return getattr(t, "_dmc_x3s", None)


def f32_valid(t):
    """False when ``t``'s fp32 memory was not written (its producer emitted slices only)."""
    return getattr(t, "_dmc_f32", True)


def _attach_x3s(t, xs, has_f32=True):
    t._dmc_x3s = xs
    t._dmc_f32 = has_f32
    return t


def _x3s_buffer(m, c, device):
    return torch.empty(_lib.load().dmc_x3s_slices_bytes(m, c), dtype=torch.uint8, device=device)


def x3s_split(x):
    """Slice tensor of a channels_last fp32 activation (the stand-alone producer: tests, inputs nobody split)."""
    n, c, h, w = x.shape
    xs = _x3s_buffer(n * h * w, c, x.device)
    _lib.check(_lib.load().dmc_x3s_split(_lib.ptr(x), _lib.ptr(xs), n * h * w, c, _stream()), "dmc_x3s_split")
    return xs


def x3s_merge(xs, shape):
    """fp32 channels_last tensor of ``shape`` from its slice tensor (exact: the three slices sum to the value)."""
    n, c, h, w = shape
    x = torch.empty(shape, dtype=torch.float32, device=xs.device, memory_format=torch.channels_last)
    _lib.check(_lib.load().dmc_x3s_merge(_lib.ptr(xs), _lib.ptr(x), n * h * w, c, _stream()), "dmc_x3s_merge")
    return x


def x3s_pack_weights(weight, forward=True, transposed=True):
    """(wpack_f, wpack_t) of a channels_last [Cout, Cin, 3, 3] weight for dmc_x3s_conv_fwd / dmc_x3s_conv_dgrad."""
    lib = _lib.load()
    cout, cin = weight.shape[0], weight.shape[1]
    nb = lib.dmc_x3s_wpack_bytes(cin, cout)
    wf = torch.empty(nb, dtype=torch.uint8, device=weight.device) if forward else None
    wt = torch.empty(nb, dtype=torch.uint8, device=weight.device) if transposed else None
    _lib.check(lib.dmc_x3s_pack_weights(_lib.ptr(_as_cl(weight)), _lib.ptr(wf), _lib.ptr(wt), cin, cout, _stream()),
               "dmc_x3s_pack_weights")
    return wf, wt


def x3s_conv_fwd(xs, wpack_f, n, h, w, cin, cout, want_stats=False):
    """conv3x3 (stride 1, padding 1) of the activation whose slice tensor is ``xs`` -> (y fp32 channels_last, statistics
    partials [blocks, Cout, 2] float64 or None)."""
    lib = _lib.load()
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=xs.device, memory_format=torch.channels_last)
    part, nblk = None, 0
    if want_stats:
        nblk = lib.dmc_x3s_conv_stat_blocks(n, h, w, cout)
        part = torch.empty((nblk, cout, 2), dtype=torch.float64, device=xs.device)
    _lib.check(lib.dmc_x3s_conv_fwd(_lib.ptr(xs), _lib.ptr(wpack_f), _lib.ptr(y), _lib.ptr(part), nblk, n, h, w, cin, cout, _stream()),
               "dmc_x3s_conv_fwd")
    return y, part


def x3s_conv_dgrad(dys, wpack_t, n, h, w, cin, cout, addend=None):
    dx = torch.empty((n, cin, h, w), dtype=torch.float32, device=dys.device, memory_format=torch.channels_last)
    _lib.check(_lib.load().dmc_x3s_conv_dgrad(_lib.ptr(dys), _lib.ptr(wpack_t), _lib.ptr(addend), _lib.ptr(dx), n, h, w, cin, cout,
                                              _stream()), "dmc_x3s_conv_dgrad")
    return dx


def x3s_conv_wgrad(xs, dys, n, h, w, cin, cout):
    lib = _lib.load()
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=xs.device, memory_format=torch.channels_last)
    work = _floats(lib.dmc_x3s_conv_wgrad_bytes(n, h, w, cin, cout), xs.device)
    _lib.check(lib.dmc_x3s_conv_wgrad(_lib.ptr(xs), _lib.ptr(dys), _lib.ptr(dw), _lib.ptr(work), n, h, w, cin, cout, _stream()),
               "dmc_x3s_conv_wgrad")
    return dw


def x3s_usable(n, h, w, conv):
    """True if ``conv`` on an [n, Cin, h, w] activation takes the pre-split path (3x3, stride 1, padding 1, no bias)."""
    if not X3S or conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.padding != (1, 1) or conv.bias is not None:
        return False
    if conv.dilation != (1, 1) or conv.groups != 1:
        return False
    lib = _lib.load()
    if lib.dmc_get_option(b"conv_arith") != 1 or lib.dmc_get_option(b"conv_path") != 1:
        return False
    cin, cout = conv.in_channels, conv.out_channels
    return bool(lib.dmc_x3s_conv_supported(n, h, w, cin, cout)) and bool(lib.dmc_x3s_conv_wgrad_supported(n, h, w, cin, cout))


# ------------------------------------------------------------------ stride-2 block pair on s2d slice tensors (conv_x3q.hip)
def x3q_split(x):
    """Space-to-depth slice tensor of a channels_last fp32 activation with even H, W (stand-alone producer: tests)."""
    n, c, h, w = x.shape
    xq = _x3s_buffer(n * h * w, c, x.device)
    _lib.check(_lib.load().dmc_x3q_split(_lib.ptr(x), _lib.ptr(xq), n, h, w, c, _stream()), "dmc_x3q_split")
    return xq


def x3q_merge(xq, shape):
    n, c, h, w = shape
    x = torch.empty(shape, dtype=torch.float32, device=xq.device, memory_format=torch.channels_last)
    _lib.check(_lib.load().dmc_x3q_merge(_lib.ptr(xq), _lib.ptr(x), n, h, w, c, _stream()), "dmc_x3q_merge")
    return x


def x3q_pack_weights(w3, w1, forward=True, transposed=True):
    """(wpack_f, wpack_t) of the channels_last [Cout, Cin, 3, 3] / [Cout, Cin, 1, 1] weight pair of a stride-2 block."""
    lib = _lib.load()
    cout, cin = w3.shape[0], w3.shape[1]
    nb = lib.dmc_x3q_wpack_bytes(cin, cout)
    wf = torch.empty(nb, dtype=torch.uint8, device=w3.device) if forward else None
    wt = torch.empty(nb, dtype=torch.uint8, device=w3.device) if transposed else None
    _lib.check(lib.dmc_x3q_pack_weights(_lib.ptr(_as_cl(w3)), _lib.ptr(_as_cl(w1)), _lib.ptr(wf), _lib.ptr(wt), cin, cout, _stream()),
               "dmc_x3q_pack_weights")
    return wf, wt


def x3q_conv_fwd(xq, wpack_f, n, oh, ow, cin, cout, want_stats=False):
    """(y3, y1, partials3, partials1): the 3x3 / stride-2 convolution and the 1x1 / stride-2 shortcut of the activation whose
    s2d slice tensor is ``xq``, both [n, cout, oh, ow] channels_last fp32, from ONE launch."""
    lib = _lib.load()
    y3 = torch.empty((n, cout, oh, ow), dtype=torch.float32, device=xq.device, memory_format=torch.channels_last)
    y1 = torch.empty_like(y3)
    p3 = p1 = None
    nblk = 0
    if want_stats:
        nblk = lib.dmc_x3q_stat_blocks(n, oh, ow, cout)
        p3 = torch.empty((nblk, cout, 2), dtype=torch.float64, device=xq.device)
        p1 = torch.empty_like(p3)
    _lib.check(lib.dmc_x3q_conv_fwd(_lib.ptr(xq), _lib.ptr(wpack_f), _lib.ptr(y3), _lib.ptr(y1), _lib.ptr(p3), _lib.ptr(p1), nblk,
                                    n, oh, ow, cin, cout, _stream()), "dmc_x3q_conv_fwd")
    return y3, y1, p3, p1


def x3q_conv_dgrad(dys3, dys1, wpack_t, n, oh, ow, cin, cout):
    """dx [n, cin, 2 oh, 2 ow] of the pair from the slice tensors of the two output gradients."""
    dx = torch.empty((n, cin, 2 * oh, 2 * ow), dtype=torch.float32, device=dys3.device, memory_format=torch.channels_last)
    _lib.check(_lib.load().dmc_x3q_conv_dgrad(_lib.ptr(dys3), _lib.ptr(dys1), _lib.ptr(wpack_t), _lib.ptr(dx), n, oh, ow, cin, cout,
                                              _stream()), "dmc_x3q_conv_dgrad")
    return dx


def x3q_conv_wgrad(xq, dys3, dys1, n, oh, ow, cin, cout):
    """(dw3 [cout, cin, 3, 3], dw1 [cout, cin, 1, 1]), channels_last memory, of the stride-2 pair."""
    lib = _lib.load()
    dw3 = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=xq.device, memory_format=torch.channels_last)
    dw1 = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=xq.device, memory_format=torch.channels_last)
    work = _floats(lib.dmc_x3q_conv_wgrad_bytes(n, oh, ow, cin, cout), xq.device)
    _lib.check(lib.dmc_x3q_conv_wgrad(_lib.ptr(xq), _lib.ptr(dys3), _lib.ptr(dys1), _lib.ptr(dw3), _lib.ptr(dw1), _lib.ptr(work),
                                      n, oh, ow, cin, cout, _stream()), "dmc_x3q_conv_wgrad")
    return dw3, dw1


class ResidualGradLink:
    """Couples the two fused ops of an identity-shortcut residual block (torchvision BasicBlock: `out += identity`, behind
    code/dmcnet/model.py:305): the block input feeds the first convolution AND the residual add, so autograd would sum the
    two gradients in a separate elementwise pass over the block input (96 MB per tensor in layer1 at 120 frames).  With a link
    the block's LAST op (the one given ``residual``) parks its residual gradient here instead of returning it, and the
    block's FIRST op (which autograd always runs later: its output feeds the last op) adds it in the epilogue of its data-
    gradient launch.  The sum is the same two fp32 addends; only the pass is saved."""
    __slots__ = ("armed", "grad")

    def __init__(self):
        self.armed = False
        self.grad = None


#: True (default): the classifier's weight gradients run on a side HIP stream.  Nothing reads a weight gradient before the
#: optimizer / the gradient exchange, while everything else in the backward pass is one dependent chain; on its own stream
#: a (matrix-bound) weight-gradient launch overlaps the (memory-bound) BatchNorm passes of the layers below it: 12.43 ->
#: 12.25 ms per step.  The main stream re-joins when backward() returns (an autograd-engine callback), in the gradient
#: exchange before a bucket is read, and in join_wgrad_stream().  DMC_WGRAD_STREAM=0: one stream.
#: Used where a backward pass is wrapped in ``with ops.wgrad_side_stream():`` -- DmcnetTrainStep does (config 2: -0.32 ms);
#: the GAN step does not (measured 0.1 ms slower there: its D step is closer to launch-bound and gains nothing back).
WGRAD_STREAM = __import__("os").environ.get("DMC_WGRAD_STREAM", "1") != "0"
_WGRAD_STREAMS = {}
_WGRAD_PENDING = [False]
_WGRAD_SCOPE = [0]
_WGRAD_COUNT = [0]            # launches that went to the side stream (diagnostics / tests)
_WGRAD_SEEN = set()           # id() of EVERY weight whose gradient went to the side stream in this backward pass (cleared per pass, not per join)
_WGRAD_MAIN = set()           # id() of the weights forced to the main stream for the REST of this backward pass (seen twice)


class wgrad_side_stream(object):
    """Context manager around ``loss.backward()``: weight gradients of the classifier convolutions inside it may run on
    the side stream (see WGRAD_STREAM).  A plain global counter: backward functions run on the autograd engine's thread."""

    def __enter__(self):
        _WGRAD_SCOPE[0] += 1
        _WGRAD_SEEN.clear()
        _WGRAD_MAIN.clear()
        return self

    def __exit__(self, *exc):
        _WGRAD_SCOPE[0] -= 1
        join_wgrad_stream()
        _WGRAD_SEEN.clear()
        _WGRAD_MAIN.clear()
        return False


def _wgrad_stream(device):
    key = torch.device(device).index
    if key not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return _WGRAD_STREAMS[key]


def join_wgrad_stream():
    """The current stream waits for the weight gradients launched on the side stream (no-op when none are pending).
    _WGRAD_SEEN survives: a mid-pass join (a shared weight's second use, a gradient bucket being read) completes the gradients
    in flight but a weight B that produced one before it may still be used AGAIN later in the pass -- its second gradient
    must then be produced on the main stream, where the engine sums the two."""
    if _WGRAD_PENDING[0]:
        cur = torch.cuda.current_stream()
        for st in _WGRAD_STREAMS.values():
            cur.wait_stream(st)
        _WGRAD_PENDING[0] = False


def _wgrad_side_ok(weight):
    """True if ``weight``'s gradient may be produced on the side stream (see _on_wgrad_stream)."""
    if weight.grad is not None:                 # ``weight.grad += dw`` would run on the main stream at once
        return False
    # a weight used TWICE in one backward pass (a shared convolution, a module called twice): the engine sums the two
    # gradients on the main stream -- in its input buffer or in AccumulateGrad -- believing the main stream produced the
    # first one.  The second sighting therefore re-joins (the first gradient is complete on the main stream) and stays there.
    # _WGRAD_SEEN is per PASS (cleared by the scope, not by a join): with two shared weights A and B used alternately, A's
    # second use joins mid-pass; B's second use afterwards must still be recognised.  Any further use (a module called three
    # times, a discriminator applied to real, fake and an interpolate) stays on the main stream too -- its gradient is
    # summed into the same buffer.
    if id(weight) in _WGRAD_MAIN:
        return False
    if id(weight) in _WGRAD_SEEN:
        join_wgrad_stream()                     # no-op when an earlier join already completed the first gradient
        _WGRAD_MAIN.add(id(weight))
        return False
    # C++-level gradient hooks (torch's DistributedDataParallel reducer) read the gradient on the main stream the moment it
    # is accumulated and are not visible from Python: with a process group up, only parameters that carry THIS package's
    # deferring hook (ddp.GradBucketReducer) may use the side stream
    post = getattr(weight, "_post_accumulate_grad_hooks", None)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        if not post or not all(getattr(h, "_dmc_defers_read", False) for h in post.values()):
            return False
    # a hook that reads the gradient the moment it is accumulated would read it on the main stream, before the side stream
    # has written it: only hooks that declare they defer the read (the gradient exchange's, ddp.py) are compatible
    if getattr(weight, "_backward_hooks", None):
        return False
    if post and not all(getattr(h, "_dmc_defers_read", False) for h in post.values()):
        return False
    return True


def _on_wgrad_stream(weight, reads, launch):
    """``launch()`` (allocates, launches, returns the weight gradient -- a tuple of them when ``weight`` is a tuple of
    parameters served by one call) on the side stream when that is safe: inside an autograd backward pass (the join is
    queued as an engine callback: it runs on the caller's stream before backward() returns) and with no gradient to
    accumulate into."""
    weights = weight if isinstance(weight, tuple) else (weight,)
    if not WGRAD_STREAM or _WGRAD_SCOPE[0] <= 0 or not all(_wgrad_side_ok(w) for w in weights):
        return launch()
    if PROBE is not None and (PROBE.only is None or "conv_nhwc_wgrad" in PROBE.only or "stem_wgrad" in PROBE.only):
        return launch()                         # HIP-event spans around this call time the launch stream: stay on it
    if not _WGRAD_PENDING[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_stream)
        except RuntimeError:                    # not inside a backward pass (a direct call): stay on this stream
            return launch()
    main, side = torch.cuda.current_stream(), _wgrad_stream(weights[0].device)
    side.wait_stream(main)                      # the operands were written on the main stream
    for t in reads:
        t.record_stream(side)                   # their memory is not reused before the side stream has read it
    _WGRAD_PENDING[0] = True
    _WGRAD_COUNT[0] += 1
    for w in weights:
        _WGRAD_SEEN.add(id(w))
    with torch.cuda.stream(side):
        dw = launch()
    for t in (dw if isinstance(dw, tuple) else (dw,)):
        t.record_stream(main)
    return dw


class BnBwdLink:
    """Couples a fused conv -> bn op U with the ONE pre-split convolution V that reads its result: V's data-gradient launch
    writes U's output gradient, so it also reduces U's BatchNorm-backward sums (dbeta, dgamma) from the tile it holds in LDS
    (dmc_x3s_conv_dgrad_bnb) and U's backward skips its reduction pass (bn_partial: a read of dout and y).  ``kind``:
    "inner" = U's result feeds V alone (conv1 -> conv2 of a block); "block" = U closes a block whose output feeds an
    identity-shortcut block -- V's launch is then U's complete output gradient only when it also adds the residual branch's
    gradient (ResidualGradLink), which V checks at backward time.  Unused links cost nothing: U falls back to its own pass."""
    __slots__ = ("kind", "y", "stats", "gamma", "beta", "mask", "relu", "dgamma", "dbeta", "ready")

    def __init__(self, kind):
        self.kind, self.ready = kind, False
        self.y = self.stats = self.gamma = self.beta = self.mask = self.dgamma = self.dbeta = None
        self.relu = False


class _ConvBnAct(torch.autograd.Function):
    """relu?(BatchNorm2d(conv2d(x, w)) [+ residual]) in training mode: the ResNet's conv -> bn [-> add] [-> relu]
    chains (torchvision BasicBlock / Bottleneck / downsample behind code/dmcnet/model.py:305,352) on the
    matrix-core NHWC kernels.  The convolution's epilogue reduces the batch statistics (no separate pass
    over its output), one streaming pass normalises / adds / rectifies; the backward runs the BatchNorm
    backward (two passes) and feeds the data- and weight-gradient kernels (deterministic).

    ``xs``: the slice tensor of ``x`` (or None); ``mode`` bit 0: write the fp32 result, bit 1: write its slice tensor
    (returned second, non-differentiable), bit 2: take the pre-split convolution path (conv_x3s.hip), bit 3: the slice
    tensor is written SPACE-TO-DEPTH (its consumer is a stride-2 block pair, conv_x3q.hip)."""

    @staticmethod
    def forward(ctx, x, weight, residual, gamma, beta, running_mean, running_var, stride, padding, relu, eps,
                momentum, link=None, xs=None, mode=1, bn_out=None, bn_in=None):
        lib = _lib.load()
        _need_cuda(x, weight, residual, gamma, beta)
        ctx.set_materialize_grads(False)
        ctx.bn_out, ctx.bn_in = bn_out, None    # else autograd zero-fills a gradient for the slice output every backward
        want_f32, want_xs, use_x3s = bool(mode & 1), bool(mode & 2), bool(mode & 4)
        wcl = _as_cl(weight)
        cout, cin = weight.shape[0], weight.shape[1]
        n, _, h, w = x.shape
        ctx.wsplit_t = None
        ctx.wsplit_t2 = None
        ctx.link = None
        if use_x3s:
            if xs is None:
                xs = x3s_split(_as_cl(x))
            with _span("conv_nhwc_fwd"):
                nb = lib.dmc_x3s_wpack_bytes(cin, cout)
                wf = torch.empty(nb, dtype=torch.uint8, device=x.device)
                ctx.wsplit_t = torch.empty(nb, dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
                _lib.check(lib.dmc_x3s_pack_weights(_lib.ptr(wcl), _lib.ptr(wf), _lib.ptr(ctx.wsplit_t), cin, cout, _stream()),
                           "dmc_x3s_pack_weights")
                y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device, memory_format=_CL)
                nblk = lib.dmc_x3s_conv_stat_blocks(n, h, w, cout)
                part = torch.empty((nblk, cout, 2), dtype=torch.float64, device=x.device)
                _lib.check(lib.dmc_x3s_conv_fwd(_lib.ptr(xs), _lib.ptr(wf), _lib.ptr(y), _lib.ptr(part), nblk, n, h, w, cin, cout,
                                                _stream()), "dmc_x3s_conv_fwd")
            wf_ok = True
            if bn_in is not None and stride == 1 and ctx.needs_input_grad[0]:
                ctx.bn_in = bn_in                           # this op's data gradient is the producer's output gradient
        else:
            if not f32_valid(x):
                raise RuntimeError("conv_bn_act: the input's fp32 memory was not written by its producer (slices only)")
            x = _as_cl(x)
            with _span("conv_nhwc_fwd"):
                wf = None
                if ctx.needs_input_grad[0] and lib.dmc_conv_nhwc_presplit_supported(cin, cout):
                    # bf16x3 kernels: the forward's and the data gradient's weight slices from ONE launch
                    nb = lib.dmc_conv_nhwc_wt_bytes(cin, cout, weight.shape[2], weight.shape[3])
                    wf, ctx.wsplit_t = _floats(nb, x.device), _floats(nb, x.device)
                    _lib.check(lib.dmc_conv_nhwc_split(_lib.ptr(wcl), _lib.ptr(wf), _lib.ptr(ctx.wsplit_t), cin, cout,
                                                       weight.shape[2], weight.shape[3], _stream()), "dmc_conv_nhwc_split")
                y, part, nblk = _conv_fwd(x, wcl, None, None, stride, padding, 0, True, presplit=wf)
                # 3x3 / stride 2 (layerN.0.conv1): the data gradient in ONE pre-split launch instead of four parity-class launches
                ctx.wsplit_t2 = None
                if (X3S and ctx.needs_input_grad[0] and stride == 2 and padding == 1 and tuple(weight.shape[2:]) == (3, 3)
                        and h == 2 * y.shape[2] and w == 2 * y.shape[3] and lib.dmc_get_option(b"conv_arith") == 1
                        and lib.dmc_x3s_conv_dgrad_s2_supported(n, y.shape[2], y.shape[3], cin, cout)):
                    ctx.wsplit_t2 = torch.empty(lib.dmc_x3s_wpack_bytes(cin, cout), dtype=torch.uint8, device=x.device)
                    _lib.check(lib.dmc_x3s_pack_weights_s2(_lib.ptr(wcl), _lib.ptr(ctx.wsplit_t2), cin, cout, _stream()),
                               "dmc_x3s_pack_weights_s2")
            wf_ok = wf is not None
        if link is not None:
            if residual is None:
                if wf_ok and stride == 1:                   # first op of the block: will add the parked gradient
                    link.armed, ctx.link = True, link
            elif link.armed and ctx.needs_input_grad[2]:    # last op of the block: will park its residual gradient
                ctx.link = link
        n, _, oh, ow = y.shape
        m = n * oh * ow
        stats = _floats(lib.dmc_bn_act_stats_bytes(cout), x.device)
        out = torch.empty_like(y)                           # not written when the consumer reads slices only
        out_xs = _x3s_buffer(m, cout, x.device) if want_xs else None
        mask = None
        if residual is not None:
            if not f32_valid(residual):
                raise RuntimeError("conv_bn_act: the residual's fp32 memory was not written by its producer")
            residual = _as_cl(residual)
            if relu:
                mask = torch.empty(m * (cout // 4), dtype=torch.uint8, device=x.device)
        with _span("bn_apply_fwd"):
            _lib.check(lib.dmc_conv_nhwc_stats_final(_lib.ptr(part), nblk, cout, m, _lib.ptr(stats),
                                                     _lib.ptr(running_mean), _lib.ptr(running_var), float(eps),
                                                     float(momentum), _stream()), "dmc_conv_nhwc_stats_final")
            if want_xs and (mode & 8):
                _lib.check(lib.dmc_bn_apply_act_x3q(_lib.ptr(y), _lib.ptr(residual), _lib.ptr(gamma), _lib.ptr(beta),
                                                    _lib.ptr(stats), _lib.ptr(out) if want_f32 else None, _lib.ptr(out_xs),
                                                    _lib.ptr(mask), n, oh, ow, cout, int(relu), _stream()), "dmc_bn_apply_act_x3q")
            elif want_xs or not want_f32:
                _lib.check(lib.dmc_bn_apply_act_x3s(_lib.ptr(y), _lib.ptr(residual), _lib.ptr(gamma), _lib.ptr(beta),
                                                    _lib.ptr(stats), _lib.ptr(out) if want_f32 else None, _lib.ptr(out_xs),
                                                    _lib.ptr(mask), m, cout, int(relu), _stream()), "dmc_bn_apply_act_x3s")
            else:
                _lib.check(lib.dmc_bn_apply_act_nhwc(_lib.ptr(y), _lib.ptr(residual), _lib.ptr(gamma), _lib.ptr(beta),
                                                     _lib.ptr(stats), _lib.ptr(out), _lib.ptr(mask), m, cout, int(relu),
                                                     _stream()), "dmc_bn_apply_act_nhwc")
        if bn_out is not None:                              # what the consumer's data gradient needs for this op's BatchNorm sums
            bn_out.y, bn_out.stats, bn_out.gamma, bn_out.beta, bn_out.mask, bn_out.relu = y, stats, gamma, beta, mask, bool(relu)
        # the pre-split path keeps the input's slices for its weight gradient, not the fp32 input
        ctx.save_for_backward(xs if use_x3s else x, weight, y, gamma, beta, stats, mask)
        ctx.cfg = (int(stride), int(padding), bool(relu), residual is not None, use_x3s, tuple(x.shape))
        if out_xs is None:
            out_xs = torch.empty(0, dtype=torch.uint8, device=x.device)
        ctx.mark_non_differentiable(out_xs)
        return out, out_xs

    @staticmethod
    def backward(ctx, dout, _dxs):
        lib = _lib.load()
        x, weight, y, gamma, beta, stats, mask = ctx.saved_tensors
        if dout is None:                    # the result did not reach the loss
            return (None,) * 17
        stride, padding, relu, has_res, use_x3s, x_shape = ctx.cfg
        n, cout, oh, ow = y.shape
        m = n * oh * ow
        cin = weight.shape[1]
        dout = _as_cl(dout)
        want_dres = has_res and ctx.needs_input_grad[2]
        dres = torch.empty_like(y) if (want_dres and relu) else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        scratch = _floats(lib.dmc_bn_act_scratch_bytes(cout), y.device)
        dy = dys = None
        pre = ctx.bn_out is not None and ctx.bn_out.ready   # the consumer's data gradient already reduced dgamma / dbeta
        with _span("bn_act_bwd"):
            if pre:
                dgamma, dbeta = ctx.bn_out.dgamma, ctx.bn_out.dbeta
                s2 = ctx.wsplit_t2 is not None and ctx.needs_input_grad[0]
                if use_x3s or s2:
                    dys = _x3s_buffer(m, cout, y.device)
                if not use_x3s:
                    dy = torch.empty_like(y)
                _lib.check(lib.dmc_bn_act_bwd_x3s_apply(_lib.ptr(y), None, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                                        _lib.ptr(dout), _lib.ptr(dy), _lib.ptr(dys), _lib.ptr(dres),
                                                        _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(mask), m, cout,
                                                        int(relu), _stream()), "dmc_bn_act_bwd_x3s_apply")
            elif use_x3s:                                   # the convolution's output gradient: slices only
                dys = _x3s_buffer(m, cout, y.device)
                _lib.check(lib.dmc_bn_act_bwd_x3s(_lib.ptr(y), None, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                                  _lib.ptr(scratch), _lib.ptr(dout), None, _lib.ptr(dys), _lib.ptr(dres),
                                                  _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(mask), m, cout,
                                                  int(relu), _stream()), "dmc_bn_act_bwd_x3s")
            elif ctx.wsplit_t2 is not None and ctx.needs_input_grad[0]:   # fp32 for the weight gradient, slices for the data gradient
                dy = torch.empty_like(y)
                dys = _x3s_buffer(m, cout, y.device)
                _lib.check(lib.dmc_bn_act_bwd_x3s(_lib.ptr(y), None, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                                  _lib.ptr(scratch), _lib.ptr(dout), _lib.ptr(dy), _lib.ptr(dys), _lib.ptr(dres),
                                                  _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(mask), m, cout,
                                                  int(relu), _stream()), "dmc_bn_act_bwd_x3s")
            else:
                dy = torch.empty_like(y)
                _lib.check(lib.dmc_bn_act_bwd(_lib.ptr(y), None, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                              _lib.ptr(scratch), _lib.ptr(dout), _lib.ptr(dy), _lib.ptr(dres),
                                              _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(mask), m, cout,
                                              int(relu), _stream()), "dmc_bn_act_bwd")
        if want_dres and not relu:
            dres = dout
        addend = None
        if ctx.link is not None:
            if has_res:
                ctx.link.grad, dres = dres, None            # parked for the block's first op (ResidualGradLink)
            else:
                addend, ctx.link.grad = ctx.link.grad, None
        wcl = _as_cl(weight)
        dx = dw = None
        if use_x3s:
            nn_, _, h, w = x_shape
            if ctx.needs_input_grad[0]:
                with _span("conv_nhwc_dgrad"):
                    dx = torch.empty(x_shape, dtype=torch.float32, device=y.device, memory_format=_CL)
                    if addend is not None:
                        addend = _as_cl(addend)
                    bl = ctx.bn_in
                    if bl is not None and bl.y is not None and (bl.kind == "inner" or addend is not None):
                        # dx is the producer's COMPLETE output gradient: reduce its BatchNorm sums in this launch's epilogue
                        nblk = lib.dmc_x3s_conv_stat_blocks(nn_, h, w, cin)
                        part = torch.empty((nblk, cin, 2), dtype=torch.float64, device=y.device)
                        bl.dgamma, bl.dbeta = torch.empty_like(bl.gamma), torch.empty_like(bl.gamma)
                        _lib.check(lib.dmc_x3s_conv_dgrad_bnb(_lib.ptr(dys), _lib.ptr(ctx.wsplit_t), _lib.ptr(addend), _lib.ptr(dx),
                                                              _lib.ptr(bl.y), _lib.ptr(bl.stats), _lib.ptr(bl.gamma), _lib.ptr(bl.beta),
                                                              _lib.ptr(bl.mask), int(bl.relu), _lib.ptr(part), nblk, _lib.ptr(bl.dgamma),
                                                              _lib.ptr(bl.dbeta), nn_, h, w, cin, cout, _stream()),
                                   "dmc_x3s_conv_dgrad_bnb")
                        bl.ready = True
                    else:
                        _lib.check(lib.dmc_x3s_conv_dgrad(_lib.ptr(dys), _lib.ptr(ctx.wsplit_t), _lib.ptr(addend), _lib.ptr(dx),
                                                          nn_, h, w, cin, cout, _stream()), "dmc_x3s_conv_dgrad")
            if ctx.needs_input_grad[1]:
                with _span("conv_nhwc_wgrad"):
                    def launch():
                        dwc = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=y.device, memory_format=_CL)
                        work = _floats(lib.dmc_x3s_conv_wgrad_bytes(nn_, h, w, cin, cout), y.device)
                        _lib.check(lib.dmc_x3s_conv_wgrad(_lib.ptr(x), _lib.ptr(dys), _lib.ptr(dwc), _lib.ptr(work), nn_, h, w, cin,
                                                          cout, _stream()), "dmc_x3s_conv_wgrad")
                        return _grad_like(dwc, weight)
                    dw = _on_wgrad_stream(weight, (x, dys), launch)
        else:
            if ctx.needs_input_grad[0]:
                with _span("conv_nhwc_dgrad"):
                    if dys is not None:
                        dx = torch.empty(x_shape, dtype=torch.float32, device=y.device, memory_format=_CL)
                        _lib.check(lib.dmc_x3s_conv_dgrad_s2(_lib.ptr(dys), _lib.ptr(ctx.wsplit_t2), _lib.ptr(dx), n, oh, ow, cin,
                                                             cout, _stream()), "dmc_x3s_conv_dgrad_s2")
                    else:
                        dx = _conv_dgrad(dy, wcl, x.shape, stride, padding, presplit=ctx.wsplit_t, addend=addend)
            if ctx.needs_input_grad[1]:
                with _span("conv_nhwc_wgrad"):
                    dw = _on_wgrad_stream(weight, (x, dy), lambda: _grad_like(_conv_wgrad(x, dy, wcl, stride, padding), weight))
        return dx, dw, dres, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None, None


def conv_bn_act_supported(x, conv, bn):
    """True if conv -> bn [-> add] [-> relu] can run as the fused NHWC training op."""
    if not (torch.is_grad_enabled() and bn.training and bn.track_running_stats and bn.affine):
        return False
    if bn.momentum is None:                       # cumulative moving average: the stock module handles it
        return False
    if conv.bias is not None or conv.dilation != (1, 1) or conv.groups != 1 or conv.padding_mode != "zeros":
        return False
    if conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1]:
        return False
    if not _is_cl(x) or not conv_nhwc_supported(x, conv.weight, conv.stride[0], conv.padding[0]):
        return False
    n, _, h, w = x.shape
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    return bool(_lib.load().dmc_bn_act_supported(n * oh * ow, conv.out_channels))


def conv_bn_eval_supported(x, conv, bn):
    """True if conv -> bn(running statistics) [-> add] [-> relu] can run forward-only on this package's kernels
    (evaluation / validation: code/dmcnet/test.py:139-198, validate() of code/dmcnet/train.py)."""
    if torch.is_grad_enabled() or bn.training or not bn.track_running_stats or not bn.affine:
        return False
    if conv.bias is not None or conv.dilation != (1, 1) or conv.groups != 1 or conv.padding_mode != "zeros":
        return False
    if conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1]:
        return False
    if not _is_cl(x) or not conv_nhwc_supported(x, conv.weight, conv.stride[0], conv.padding[0]):
        return False
    n, _, h, w = x.shape
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    return bool(_lib.load().dmc_bn_act_supported(n * oh * ow, conv.out_channels)) and conv.out_channels % 16 == 0


def conv_bn_act_eval(x, conv, bn, residual=None, relu=True, want_f32=True, want_slices=False):
    """Forward-only relu?(bn(conv(x)) [+ residual]) with the BatchNorm's RUNNING statistics (eval mode, no autograd): the
    same convolution kernels as the training op (pre-split path for stride-1 3x3), no statistics epilogue, and the
    normalisation as one streaming pass with (mean, 1 / sqrt(var + eps)) taken from the module's buffers."""
    lib = _lib.load()
    _need_cuda(x, conv.weight, residual)
    n, cin, h, w = x.shape
    cout = conv.out_channels
    wcl = _as_cl(conv.weight.detach())
    stride, padding = conv.stride[0], conv.padding[0]
    if x3s_usable(n, h, w, conv):
        xs = x3s_of(x)
        if xs is None:
            xs = x3s_split(_as_cl(x))
        wf, _ = x3s_pack_weights(wcl, forward=True, transposed=False)
        y, _ = x3s_conv_fwd(xs, wf, n, h, w, cin, cout)
    else:
        if not f32_valid(x):
            raise RuntimeError("conv_bn_act_eval: the input's fp32 memory was not written by its producer (slices only)")
        y, _, _ = _conv_fwd(_as_cl(x), wcl, None, None, stride, padding, 0, False)
    m = y.shape[0] * y.shape[2] * y.shape[3]
    stats = torch.cat([bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)]).contiguous()
    if not want_f32 and not want_slices:
        want_f32 = True
    out = torch.empty_like(y)
    out_xs = _x3s_buffer(m, cout, x.device) if want_slices else None
    if residual is not None:
        if not f32_valid(residual):
            raise RuntimeError("conv_bn_act_eval: the residual's fp32 memory was not written by its producer")
        residual = _as_cl(residual)
    _lib.check(lib.dmc_bn_apply_act_x3s(_lib.ptr(y), _lib.ptr(residual), _lib.ptr(bn.weight), _lib.ptr(bn.bias), _lib.ptr(stats),
                                        _lib.ptr(out) if want_f32 else None, _lib.ptr(out_xs), None, m, cout, int(relu),
                                        _stream()), "dmc_bn_apply_act_x3s")
    if want_slices:
        _attach_x3s(out, out_xs, want_f32)
    return out


def conv_bn_act(x, conv, bn, residual=None, relu=True, link=None, want_f32=True, want_slices=False, bn_link=None, s2d=False):
    """relu?(bn(conv(x)) [+ residual]) for a channels_last ``x`` (see conv_bn_act_supported); ``link``: the
    ResidualGradLink shared by the first and the last op of an identity-shortcut block.  ``want_slices``: also write
    the result's bf16x3 slice tensor (attached to the returned tensor, see x3s_of) for a pre-split consumer;
    ``want_f32=False``: ONLY the slices -- the returned tensor's fp32 memory is then not written (f32_valid).
    ``bn_link``: "inner" / "block" -- the result has ONE pre-split consumer (see BnBwdLink); the link travels on the
    returned tensor and that consumer picks it up from its input."""
    if bn.num_batches_tracked is not None:
        if _PENDING_COUNTERS is not None:
            _PENDING_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    if not want_f32 and not want_slices:
        want_f32 = True
    n, _, h, w = x.shape
    use_x3s = x3s_usable(n, h, w, conv)
    mode = int(want_f32) | (int(want_slices) << 1) | (int(use_x3s) << 2) | (int(bool(s2d and want_slices)) << 3)
    bn_out = BnBwdLink(bn_link) if (bn_link and BN_BWD_LINK and want_slices and not s2d) else None
    bn_in = getattr(x, "_dmc_bnlink", None) if use_x3s else None
    out, out_xs = _ConvBnAct.apply(x, conv.weight, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   conv.stride[0], conv.padding[0], relu, bn.eps,
                                   bn.momentum if bn.momentum is not None else 0.1, link,
                                   x3s_of(x) if use_x3s else None, mode, bn_out, bn_in)
    if want_slices and s2d:
        out._dmc_x3q, out._dmc_f32 = out_xs, want_f32
    elif want_slices:
        _attach_x3s(out, out_xs, want_f32)
    if bn_out is not None:
        out._dmc_bnlink = bn_out
    return out


# ------------------------------------------------------------------ the stride-2 block pair (conv_x3q.hip)
#: True (default): a BasicBlock with a stride-2 conv1 and a 1x1 / stride-2 downsample runs both convolutions -- which read
#: the same block input -- as ONE launch per direction on the input's space-to-depth slice tensor.  DMC_X3Q=0: the separate
#: in-loop-split launches of conv_nhwc.hip (+ the stride-2 data gradient of conv_x3s.hip).
X3Q = __import__("os").environ.get("DMC_X3Q", "1") != "0"


def x3q_of(t):
    """The space-to-depth slice tensor attached to activation ``t`` by its producer (or None)."""
    return getattr(t, "_dmc_x3q", None)


def _bn_trains(bn):
    return bn.training and bn.track_running_stats and bn.affine and bn.momentum is not None


def s2_pair_usable(x_shape, unit):
    """True if ``unit`` (a resnet.ResidualUnit of kind "basic" with a downsample branch) on an input of ``x_shape`` takes the
    fused stride-2 pair path: conv1 3x3 / stride 2 / padding 1 and downsample = (1x1 / stride 2, BatchNorm2d), both bias-free,
    BatchNorms in training mode, autograd on, shapes the kernels cover.  The producer of the block input asks the same
    question to decide the layout (and the absence of an fp32 copy) of what it writes."""
    if not (X3Q and X3S and torch.is_grad_enabled()):
        return False
    ds = getattr(unit, "downsample", None)
    if getattr(unit, "kind", None) != "basic" or ds is None or len(ds) != 2:
        return False
    c3, c1 = unit.conv1, ds[0]
    for c, k, p in ((c3, (3, 3), (1, 1)), (c1, (1, 1), (0, 0))):
        if not isinstance(c, torch.nn.Conv2d) or c.kernel_size != k or c.stride != (2, 2) or c.padding != p or c.bias is not None:
            return False
        if c.dilation != (1, 1) or c.groups != 1 or c.padding_mode != "zeros" or c.weight.dtype != torch.float32:
            return False
    if not isinstance(ds[1], torch.nn.BatchNorm2d) or not _bn_trains(unit.bn1) or not _bn_trains(ds[1]):
        return False
    n, cin, h, w = x_shape
    cout = c3.out_channels
    if c3.in_channels != cin or c1.in_channels != cin or c1.out_channels != cout or (h & 1) or (w & 1):
        return False
    lib = _lib.load()
    if lib.dmc_get_option(b"conv_arith") != 1 or lib.dmc_get_option(b"conv_path") != 1:
        return False
    return bool(lib.dmc_x3q_supported(n, h // 2, w // 2, cin, cout)) and bool(lib.dmc_x3q_conv_wgrad_supported(n, h // 2, w // 2, cin, cout)) \
        and bool(lib.dmc_bn_act_supported(n * (h // 2) * (w // 2), cout))


class _ConvBnS2Pair(torch.autograd.Function):
    """(relu(bn1(conv1(x))), bn_d(downsample_conv(x))) of a stride-2 BasicBlock (torchvision, behind code/dmcnet/model.py:305,352)
    in training mode: ONE convolution launch for both branches (conv_x3q.hip: the shortcut reads the centre tap's operand), both
    BatchNorms' statistics from its epilogue; the backward runs the two BatchNorm backwards (slice outputs), ONE data-gradient
    launch for the sum of both branches and the two weight gradients from one call.  ``xq``: the space-to-depth slice tensor
    of ``x``; ``x`` itself only carries the autograd edge (its fp32 memory may be unwritten).  ``want_xs``: the main branch's
    result is written as a slice tensor ONLY (its consumer, conv2, takes the pre-split path)."""

    @staticmethod
    def forward(ctx, x, xq, w3, w1, g3, b3, rm3, rv3, eps3, mom3, g1, b1, rm1, rv1, eps1, mom1, want_xs):
        lib = _lib.load()
        _need_cuda(x, w3, w1, g3, b3, g1, b1)
        ctx.set_materialize_grads(False)
        n, cin, h, w = x.shape
        cout, oh, ow = w3.shape[0], h // 2, w // 2
        m = n * oh * ow
        dev = x.device
        with _span("conv_nhwc_fwd"):
            nb = lib.dmc_x3q_wpack_bytes(cin, cout)
            wf = torch.empty(nb, dtype=torch.uint8, device=dev)
            ctx.wt = torch.empty(nb, dtype=torch.uint8, device=dev) if ctx.needs_input_grad[0] else None
            _lib.check(lib.dmc_x3q_pack_weights(_lib.ptr(_as_cl(w3)), _lib.ptr(_as_cl(w1)), _lib.ptr(wf), _lib.ptr(ctx.wt), cin, cout,
                                                _stream()), "dmc_x3q_pack_weights")
            y3 = torch.empty((n, cout, oh, ow), dtype=torch.float32, device=dev, memory_format=_CL)
            y1 = torch.empty_like(y3)
            nblk = lib.dmc_x3q_stat_blocks(n, oh, ow, cout)
            part = torch.empty((2, nblk, cout, 2), dtype=torch.float64, device=dev)
            _lib.check(lib.dmc_x3q_conv_fwd(_lib.ptr(xq), _lib.ptr(wf), _lib.ptr(y3), _lib.ptr(y1), _lib.ptr(part[0]), _lib.ptr(part[1]),
                                            nblk, n, oh, ow, cin, cout, _stream()), "dmc_x3q_conv_fwd")
        stats3 = _floats(lib.dmc_bn_act_stats_bytes(cout), dev)
        stats1 = _floats(lib.dmc_bn_act_stats_bytes(cout), dev)
        out3 = torch.empty_like(y3)                         # not written when conv2 reads slices only
        out3_xs = _x3s_buffer(m, cout, dev) if want_xs else None
        out1 = torch.empty_like(y1)
        with _span("bn_apply_fwd"):
            _lib.check(lib.dmc_conv_nhwc_stats_final(_lib.ptr(part[0]), nblk, cout, m, _lib.ptr(stats3), _lib.ptr(rm3), _lib.ptr(rv3),
                                                     float(eps3), float(mom3), _stream()), "dmc_conv_nhwc_stats_final")
            _lib.check(lib.dmc_conv_nhwc_stats_final(_lib.ptr(part[1]), nblk, cout, m, _lib.ptr(stats1), _lib.ptr(rm1), _lib.ptr(rv1),
                                                     float(eps1), float(mom1), _stream()), "dmc_conv_nhwc_stats_final")
            _lib.check(lib.dmc_bn_apply_act_x3s(_lib.ptr(y3), None, _lib.ptr(g3), _lib.ptr(b3), _lib.ptr(stats3),
                                                None if want_xs else _lib.ptr(out3), _lib.ptr(out3_xs), None, m, cout, 1, _stream()),
                       "dmc_bn_apply_act_x3s")
            _lib.check(lib.dmc_bn_apply_act_nhwc(_lib.ptr(y1), None, _lib.ptr(g1), _lib.ptr(b1), _lib.ptr(stats1), _lib.ptr(out1), None,
                                                 m, cout, 0, _stream()), "dmc_bn_apply_act_nhwc")
        ctx.save_for_backward(xq, w3, w1, y3, y1, g3, b3, stats3, g1, b1, stats1)
        ctx.shape = (n, cin, h, w, cout)
        if out3_xs is None:
            out3_xs = torch.empty(0, dtype=torch.uint8, device=dev)
        ctx.mark_non_differentiable(out3_xs)
        return out3, out3_xs, out1

    @staticmethod
    def backward(ctx, dout3, _dxs, dout1):
        lib = _lib.load()
        xq, w3, w1, y3, y1, g3, b3, stats3, g1, b1, stats1 = ctx.saved_tensors
        n, cin, h, w, cout = ctx.shape
        oh, ow = h // 2, w // 2
        m = n * oh * ow
        if dout3 is None and dout1 is None:
            return (None,) * 17
        dout3 = torch.zeros_like(y3) if dout3 is None else _as_cl(dout3)
        dout1 = torch.zeros_like(y1) if dout1 is None else _as_cl(dout1)
        dg3, db3, dg1, db1 = torch.empty_like(g3), torch.empty_like(g3), torch.empty_like(g1), torch.empty_like(g1)
        dys3, dys1 = _x3s_buffer(m, cout, y3.device), _x3s_buffer(m, cout, y3.device)
        with _span("bn_act_bwd"):
            for y, g, b, st, dout, dys, dg, db, relu in ((y3, g3, b3, stats3, dout3, dys3, dg3, db3, 1),
                                                          (y1, g1, b1, stats1, dout1, dys1, dg1, db1, 0)):
                scratch = _floats(lib.dmc_bn_act_scratch_bytes(cout), y.device)
                _lib.check(lib.dmc_bn_act_bwd_x3s(_lib.ptr(y), None, _lib.ptr(g), _lib.ptr(b), _lib.ptr(st), _lib.ptr(scratch),
                                                  _lib.ptr(dout), None, _lib.ptr(dys), None, _lib.ptr(dg), _lib.ptr(db), None,
                                                  m, cout, relu, _stream()), "dmc_bn_act_bwd_x3s")
        dx = dw3 = dw1 = None
        if ctx.needs_input_grad[0]:
            with _span("conv_nhwc_dgrad"):
                dx = torch.empty((n, cin, h, w), dtype=torch.float32, device=y3.device, memory_format=_CL)
                _lib.check(lib.dmc_x3q_conv_dgrad(_lib.ptr(dys3), _lib.ptr(dys1), _lib.ptr(ctx.wt), _lib.ptr(dx), n, oh, ow, cin, cout,
                                                  _stream()), "dmc_x3q_conv_dgrad")
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            with _span("conv_nhwc_wgrad"):
                def launch():
                    d3 = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=y3.device, memory_format=_CL)
                    d1 = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=y3.device, memory_format=_CL)
                    work = _floats(lib.dmc_x3q_conv_wgrad_bytes(n, oh, ow, cin, cout), y3.device)
                    _lib.check(lib.dmc_x3q_conv_wgrad(_lib.ptr(xq), _lib.ptr(dys3), _lib.ptr(dys1), _lib.ptr(d3), _lib.ptr(d1),
                                                      _lib.ptr(work), n, oh, ow, cin, cout, _stream()), "dmc_x3q_conv_wgrad")
                    return _grad_like(d3, w3), _grad_like(d1, w1)
                dw3, dw1 = _on_wgrad_stream((w3, w1), (xq, dys3, dys1), launch)
        return dx, None, dw3, dw1, dg3, db3, None, None, None, None, dg1, db1, None, None, None, None, None


def conv_bn_s2_pair(x, unit, want_slices):
    """(relu(bn1(conv1(x))), downsample(x)) of a stride-2 BasicBlock ``unit`` (see s2_pair_usable); ``want_slices``: conv2 of
    the unit takes the pre-split path, the first result is returned as slices only (x3s_of)."""
    bn3, bn1 = unit.bn1, unit.downsample[1]
    for bn in (bn3, bn1):
        if bn.num_batches_tracked is not None:
            if _PENDING_COUNTERS is not None:
                _PENDING_COUNTERS.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
    xq = x3q_of(x)
    if xq is None:
        if not f32_valid(x):
            raise RuntimeError("conv_bn_s2_pair: the input carries neither s2d slices nor valid fp32 memory")
        xq = x3q_split(_as_cl(x))
    out3, out3_xs, out1 = _ConvBnS2Pair.apply(x, xq, unit.conv1.weight, unit.downsample[0].weight, bn3.weight, bn3.bias,
                                              bn3.running_mean, bn3.running_var, bn3.eps, bn3.momentum, bn1.weight, bn1.bias,
                                              bn1.running_mean, bn1.running_var, bn1.eps, bn1.momentum, bool(want_slices))
    if want_slices:
        _attach_x3s(out3, out3_xs, False)
    return out3, out1


class _DiscBlock(torch.autograd.Function):
    """One discriminator block, Conv2d(3x3, stride s, padding 1, bias) -> LeakyReLU(0.2) ->
    Dropout2d keep mask -> [BatchNorm2d(eps 0.8)], code/dmcnet_GAN/model.py:254-279, on NHWC tensors:
    the convolution epilogue applies bias / LeakyReLU / mask and reduces the BatchNorm statistics, one
    streaming pass normalises; the backward folds the BatchNorm backward, the mask and the LeakyReLU
    derivative into one pass and feeds the matrix-core data / weight gradients.
    ``first`` = the 2-channel NCHW input block (direct kernels, disc_first.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, keep, gamma, beta, running_mean, running_var, stride, training, eps,
                momentum, first):
        lib = _lib.load()
        _need_cuda(x, weight, bias, keep, gamma, beta)
        use_bn = gamma is not None
        cout = weight.shape[0]
        if first:
            x = x.contiguous()
            m, _, h, w = x.shape
            oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            wq = weight.contiguous()
            z = torch.empty((m, cout, oh, ow), dtype=torch.float32, device=x.device, memory_format=_CL)
            with _span("disc_first_fwd"):
                _lib.check(lib.dmc_disc_first_fwd(_lib.ptr(x), _lib.ptr(wq), _lib.ptr(bias), _lib.ptr(keep), _lib.ptr(z),
                                                  m, h, w, cout, 1, _stream()), "dmc_disc_first_fwd")
            part, nblk = None, 0
            if use_bn:
                raise NotImplementedError("the 2-channel first block has no BatchNorm in any discriminator")
        else:
            x, wq = _as_cl(x), _as_cl(weight)
            with _span("disc_conv_fwd"):
                z, part, nblk = _conv_fwd(x, wq, bias, keep, stride, 1, 1, use_bn and training)
        if DEBUG_DISC_Z is not None:
            DEBUG_DISC_Z.append(z)
        y, stats = z, None
        if use_bn:
            m_rows = z.shape[0] * z.shape[2] * z.shape[3]
            stats = _floats(lib.dmc_bn_act_stats_bytes(cout), x.device)
            y = torch.empty_like(z)
            with _span("disc_bn_fwd"):
                if training:
                    _lib.check(lib.dmc_conv_nhwc_stats_final(_lib.ptr(part), nblk, cout, m_rows, _lib.ptr(stats),
                                                             _lib.ptr(running_mean), _lib.ptr(running_var),
                                                             float(eps), float(momentum), _stream()),
                               "dmc_conv_nhwc_stats_final")
                    _lib.check(lib.dmc_bn_apply_nhwc(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                                     _lib.ptr(y), m_rows, cout, _stream()), "dmc_bn_apply_nhwc")
                else:
                    _lib.check(lib.dmc_bn_act_fwd(_lib.ptr(z), None, _lib.ptr(gamma), _lib.ptr(beta),
                                                  _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(y),
                                                  _lib.ptr(stats), None, None, m_rows, cout, 0, 0, float(eps),
                                                  float(momentum), _stream()), "dmc_bn_act_fwd")
        ctx.save_for_backward(x, weight, z, keep, gamma, beta, stats)
        ctx.cfg = (int(stride), bool(training), bool(first), use_bn, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight, z, keep, gamma, beta, stats = ctx.saved_tensors
        stride, training, first, use_bn, has_bias = ctx.cfg
        if use_bn and not training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not implemented")
        dy = _as_cl(dy)
        m, cout, oh, ow = z.shape
        rows = m * oh * ow
        g = torch.empty_like(z)
        dgamma = dbeta = None
        scratch = _floats(lib.dmc_bn_act_scratch_bytes(cout), z.device)
        if use_bn:
            dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        with _span("disc_bn_bwd"):
            _lib.check(lib.dmc_bn_bwd_act_nhwc(_lib.ptr(z), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                               _lib.ptr(scratch), _lib.ptr(dy), _lib.ptr(g), _lib.ptr(dgamma),
                                               _lib.ptr(dbeta), _lib.ptr(keep), oh * ow, 0.2, rows, cout, _stream()),
                       "dmc_bn_bwd_act_nhwc")
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1]
        if first:
            wq = weight.contiguous()
            mm, _, h, w = x.shape
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                with _span("disc_first_dgrad"):
                    _lib.check(lib.dmc_disc_first_dgrad(_lib.ptr(g), _lib.ptr(wq), _lib.ptr(dx), mm, h, w, cout,
                                                        _stream()), "dmc_disc_first_dgrad")
            if want_w:
                dw = torch.empty_like(wq)
                db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
                work = _floats(lib.dmc_disc_first_wgrad_bytes(cout), x.device)
                with _span("disc_first_wgrad"):
                    _lib.check(lib.dmc_disc_first_wgrad(_lib.ptr(x), _lib.ptr(g), _lib.ptr(dw), _lib.ptr(db),
                                                        _lib.ptr(work), mm, h, w, cout, _stream()),
                               "dmc_disc_first_wgrad")
                dw = dw.view_as(weight)
        else:
            wq = _as_cl(weight)
            if ctx.needs_input_grad[0]:
                with _span("disc_conv_dgrad"):
                    dx = _conv_dgrad(g, wq, x.shape, stride, 1)
            if want_w:
                with _span("disc_conv_wgrad"):
                    dw = _grad_like(_conv_wgrad(x, g, wq, stride, 1), weight)
                if has_bias:
                    db = torch.empty(cout, dtype=torch.float32, device=x.device)
                    _lib.check(lib.dmc_channel_sum_nhwc(_lib.ptr(g), _lib.ptr(scratch), _lib.ptr(db), rows, cout,
                                                        _stream()), "dmc_channel_sum_nhwc")
        if not want_w:
            db = None
        return dx, dw, db, None, dgamma, dbeta, None, None, None, None, None, None, None


def disc_block_supported(x, conv, first):
    """True if the HIP discriminator-block path handles this convolution (else the stock ops run)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if conv.kernel_size != (3, 3) or conv.padding != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1:
        return False
    if first:
        return conv.in_channels == 2 and conv.stride == (2, 2) and \
            bool(_lib.load().dmc_disc_first_supported(conv.out_channels))
    if conv.stride not in ((1, 1), (2, 2)):
        return False
    return conv_nhwc_supported(x, conv.weight, conv.stride[0], 1) and \
        bool(_lib.load().dmc_bn_act_supported(1, conv.out_channels))


def disc_block(x, conv, keep, bn, training, first=False):
    """x -> BN(keep * LeakyReLU_0.2(conv(x))) (BN optional) through the fused NHWC path.  ``x`` is the
    NCHW 2-channel cue for ``first`` blocks, a channels_last activation otherwise."""
    if bn is not None and training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    if keep is not None:
        keep = keep.contiguous()
    if bn is None:
        return _DiscBlock.apply(x, conv.weight, conv.bias, keep, None, None, None, None, conv.stride[0], training,
                                0.0, 0.0, first)
    return _DiscBlock.apply(x, conv.weight, conv.bias, keep, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                            conv.stride[0], training, bn.eps, bn.momentum if bn.momentum is not None else 0.1, first)


_STD = (0.229, 0.224, 0.225)


def u8_frames_buffer(shape, device):
    """A contiguous uint8 tensor of ``shape`` whose storage has 16 spare bytes behind it (what
    ``prepare_inputs`` wants for its aligned 4-pixel reads): the H2D copy target of a loader."""
    n = 1
    for d in shape:
        n *= int(d)
    return torch.empty(n + 16, dtype=torch.uint8, device=device)[:n].view(tuple(shape))


def check_geometry_plans(boxes, h0, w0, oh, ow):
    """Raise ValueError unless every plan (y0, x0, h, w, rh, rw, cy, cx) crops inside the H0 x W0 frame, resizes to a
    positive size and takes its OH x OW output window inside the resized box."""
    b = boxes.to(torch.int64)
    y0, x0, h, w, rh, rw, cy, cx = (b[:, i] for i in range(8))
    ok = (y0 >= 0) & (x0 >= 0) & (h > 0) & (w > 0) & (y0 + h <= h0) & (x0 + w <= w0) & (rh > 0) & (rw > 0) & \
         (cy >= 0) & (cx >= 0) & (cy + oh <= rh) & (cx + ow <= rw)
    if not bool(ok.all()):
        bad = int((~ok).nonzero()[0])
        raise ValueError("geometry plan %d = %s leaves the %d x %d frame or its resized box (output %d x %d)"
                         % (bad, [int(v) for v in boxes[bad]], h0, w0, oh, ow))


def prepare_inputs(frames_u8, flip=None, flow_ds_factor=0, boxes=None, out_size=None):
    """uint8 [N,H0,W0,7] HWC frames (+ optional [N] flip flags, [N,8] int32 geometry plans
    (y0,x0,h,w,rh,rw,cy,cx -- see transforms.geometry_plan) and an output size) on the GPU ->
    (input_flow [N,2,H,W], input_mv [N,2,H,W], input_residual
    [N,3,H,W]) exactly as the reference's loader produces them: crop + bilinear resize
    (code/dmcnet/transforms.py:36-139), flip (:47-58), blockify / normalise
    (code/dmcnet/dataset.py:215-263).  Without ``boxes`` / ``out_size`` the frames are taken whole."""
    import ctypes
    lib = _lib.load()
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 \
            or frames_u8.shape[-1] != 7:
        raise _lib.DmcHipError("prepare_inputs expects a CUDA uint8 tensor [N,H,W,7]")
    n, h0, w0, _ = frames_u8.shape
    dev = frames_u8.device
    oh, ow = (h0, w0) if out_size is None else (int(out_size[0]), int(out_size[1]))
    # the 4-pixel path reads whole aligned dwords: it needs 16 spare bytes behind the last frame.
    # ``u8_frames_buffer`` allocates such tensors; anything else is copied once into one.
    spare = frames_u8.untyped_storage().nbytes() - frames_u8.storage_offset() - frames_u8.numel()
    if frames_u8.is_contiguous() and spare >= 16:
        padded = frames_u8
    else:
        padded = u8_frames_buffer(frames_u8.shape, dev)
        padded.copy_(frames_u8)
    if flip is not None:
        flip = flip.to(dev, torch.uint8).contiguous()
    if boxes is not None:
        if tuple(boxes.shape) != (n, 8):
            raise ValueError("boxes must be [N,8] (y0, x0, h, w, rh, rw, cy, cx)")
        if not boxes.is_cuda:
            # the kernel indexes the frames with these numbers and cannot truncate the way numpy slicing does: a plan that
            # leaves the frame or the resized box is refused while it is still a host tensor (plans already on the device
            # are the caller's responsibility: checking them would cost a synchronisation per batch)
            check_geometry_plans(boxes, h0, w0, oh, ow)
        boxes = boxes.to(dev, torch.int32).contiguous()
    elif (oh, ow) != (h0, w0):
        boxes = torch.tensor([[0, 0, h0, w0, oh, ow, 0, 0]] * n, dtype=torch.int32, device=dev)
    flow = torch.empty((n, 2, oh, ow), dtype=torch.float32, device=dev)
    mv = torch.empty((n, 2, oh, ow), dtype=torch.float32, device=dev)
    res = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=dev)
    work = _floats(lib.dmc_prepare_crop_workspace_bytes(n, oh, ow, int(flow_ds_factor)), dev)
    std = torch.tensor(_STD, dtype=torch.float64).float()
    std4 = (ctypes.c_float * 4)(float(torch.mean(std)), float(std[0]), float(std[1]), float(std[2]))
    with _span("prepare_inputs"):
        _lib.check(lib.dmc_prepare_inputs_crop(_lib.ptr(padded), _lib.ptr(boxes), _lib.ptr(flip), _lib.ptr(flow),
                                               _lib.ptr(mv), _lib.ptr(res), _lib.ptr(work), n, h0, w0, oh, ow,
                                               int(flow_ds_factor), ctypes.cast(std4, ctypes.c_void_p),
                                               _stream()), "dmc_prepare_inputs_crop")
    return flow, mv, res


# ------------------------------------------------------------------ I3D trunk: bf16 Conv3d (conv3d_bf16.hip)
_CL3 = torch.channels_last_3d


def _as_cl3(t, dtype=torch.bfloat16):
    if t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous(memory_format=_CL3) else t.contiguous(memory_format=_CL3)


def conv3d_bf16_supported(x, weight, stride=(1, 1, 1), padding=None):
    """True if ``conv3d(x, weight)`` (bf16 ``x`` [N,Cin,D,H,W], fp32 ``weight``, stride 1, padding k // 2, kernel
    extents 1 or 3, channels multiples of 8) can run on the HIP implicit-GEMM kernels; the weight gradient
    additionally needs a 1x1x1 or 3x3x3 kernel."""
    if not (x.is_cuda and x.dim() == 5 and x.dtype == torch.bfloat16 and weight.dtype == torch.float32):
        return False
    if tuple(stride) != (1, 1, 1) or x.shape[1] != weight.shape[1]:
        return False
    kd, kh, kw = weight.shape[2:]
    if padding is not None and tuple(padding) != (kd // 2, kh // 2, kw // 2):
        return False
    if not ((kd, kh, kw) == (1, 1, 1) or (kd, kh, kw) == (3, 3, 3)):
        return False
    n, cin, d, h, w = x.shape
    return bool(_lib.load().dmc_conv3d_bf16_supported(n, d, h, w, cin, weight.shape[0], kd, kh, kw))


class _Conv3dBf16(torch.autograd.Function):
    """nn.Conv3d(bias=False, stride 1, TF-"SAME") of the I3D trunk's Unit3Dpy
    (code/dmcnet_I3D/network/i3d.py:372-393) on the bf16 matrix cores: bf16 NDHWC activations, fp32 master
    weights rounded to bf16 per call, fp32 accumulation; deterministic weight gradient in fp32."""

    @staticmethod
    def forward(ctx, x, weight):
        lib = _lib.load()
        if not (x.is_cuda and weight.is_cuda):
            raise _lib.DmcHipError("conv3d_bf16 runs on the HIP extension only (no CPU fallback)")
        ctx.x_was_cl3 = x.is_contiguous(memory_format=_CL3)
        x = _as_cl3(x)
        wc = weight.detach().contiguous()
        n, cin, d, h, w = x.shape
        cout, _, kd, kh, kw = wc.shape
        t = kd * kh * kw
        y = torch.empty((n, cout, d, h, w), dtype=torch.bfloat16, device=x.device, memory_format=_CL3)
        wpack = _floats(lib.dmc_conv3d_bf16_wpack_bytes(cin, cout, kd, kh, kw), x.device)
        with _span("conv3d_bf16_fwd"):
            _lib.check(lib.dmc_conv3d_bf16_fwd(_lib.ptr(x), _lib.ptr(wc), cin * t, t, 1, _lib.ptr(wpack), _lib.ptr(y),
                                               None, n, d, h, w, cin, cout, kd, kh, kw, _stream()),
                       "dmc_conv3d_bf16_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy, _dstat=None):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        dy = _as_cl3(dy)
        wc = weight.detach().contiguous()
        n, cin, d, h, w = x.shape
        cout, _, kd, kh, kw = wc.shape
        t = kd * kh * kw
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wpack = _floats(lib.dmc_conv3d_bf16_wpack_bytes(cin, cout, kd, kh, kw), x.device)
            with _span("conv3d_bf16_dgrad"):
                _lib.check(lib.dmc_conv3d_bf16_dgrad(_lib.ptr(dy), _lib.ptr(wc), cin * t, t, 1, _lib.ptr(wpack),
                                                     _lib.ptr(dx), n, d, h, w, cin, cout, kd, kh, kw, _stream()),
                           "dmc_conv3d_bf16_dgrad")
            if not ctx.x_was_cl3:          # hand the gradient back in the producer's layout (a stock op's backward --
                dx = dx.contiguous()       # e.g. MIOpen's stem convolution -- then sees the problem it saw in the forward)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(wc)
            work = _floats(lib.dmc_conv3d_bf16_wgrad_bytes(n, d, h, w, cin, cout, kd, kh, kw), x.device)
            with _span("conv3d_bf16_wgrad"):
                _lib.check(lib.dmc_conv3d_bf16_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(work), n, d, h, w,
                                                     cin, cout, kd, kh, kw, _stream()), "dmc_conv3d_bf16_wgrad")
            dw = dw.view_as(weight)
        return dx, dw


def conv3d_bf16(x, weight):
    """conv3d(x, weight, None, 1, k // 2) for a bf16 ``x`` (see conv3d_bf16_supported); returns a bf16
    channels_last_3d tensor."""
    return _Conv3dBf16.apply(x, weight)


def maxpool3d_tf_supported(x, kernel, stride):
    """True if MaxPool3dTFPadding(kernel, stride)(x) can run on the HIP kernels (bf16 CUDA [N,C,D,H,W], C % 8 == 0)."""
    if not (x.is_cuda and x.dim() == 5 and x.dtype == torch.bfloat16):
        return False
    n, c, d, h, w = x.shape
    return bool(_lib.load().dmc_maxpool3d_tf_out_shape(d, h, w, c, *[int(k) for k in kernel], *[int(s) for s in stride],
                                                       None, None, None))


class _MaxPool3dTF(torch.autograd.Function):
    """MaxPool3dTFPadding (code/dmcnet_I3D/network/i3d.py:406-418) on bf16 NDHWC tensors: zero padding to the
    TF-"SAME" extent and MaxPool3d(ceil_mode=True) in one pass; the backward gathers from the stored winning taps."""

    @staticmethod
    def forward(ctx, x, kernel, stride):
        import ctypes
        lib = _lib.load()
        if not (x.is_cuda and x.dtype == torch.bfloat16):
            raise _lib.DmcHipError("maxpool3d_tf runs on the HIP extension only: bf16 CUDA tensors (no CPU fallback)")
        ctx.x_was_cl3 = x.is_contiguous(memory_format=_CL3)
        x = _as_cl3(x)
        n, c, d, h, w = x.shape
        od, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if not lib.dmc_maxpool3d_tf_out_shape(d, h, w, c, *kernel, *stride, ctypes.byref(od), ctypes.byref(oh), ctypes.byref(ow)):
            raise _lib.DmcHipError("maxpool3d_tf: unsupported shape %s kernel %s stride %s" % (tuple(x.shape), kernel, stride))
        y = torch.empty((n, c, od.value, oh.value, ow.value), dtype=torch.bfloat16, device=x.device, memory_format=_CL3)
        code = torch.empty(y.numel(), dtype=torch.uint8, device=x.device)
        with _span("maxpool3d_fwd"):
            _lib.check(lib.dmc_maxpool3d_tf_bf16_fwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(code), n, d, h, w, c, *kernel, *stride,
                                                     _stream()), "dmc_maxpool3d_tf_bf16_fwd")
        ctx.save_for_backward(code)
        ctx.geom = (tuple(x.shape), kernel, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (code,) = ctx.saved_tensors
        (n, c, d, h, w), kernel, stride = ctx.geom
        dy = _as_cl3(dy)
        dx = torch.empty((n, c, d, h, w), dtype=torch.bfloat16, device=dy.device, memory_format=_CL3)
        with _span("maxpool3d_bwd"):
            _lib.check(lib.dmc_maxpool3d_tf_bf16_bwd(_lib.ptr(dy), _lib.ptr(code), _lib.ptr(dx), n, d, h, w, c, *kernel, *stride,
                                                     _stream()), "dmc_maxpool3d_tf_bf16_bwd")
        if not ctx.x_was_cl3:
            dx = dx.contiguous()
        return dx, None, None


def maxpool3d_tf(x, kernel, stride):
    """MaxPool3dTFPadding(kernel, stride)(x) for a bf16 ``x`` (see maxpool3d_tf_supported); channels_last_3d result."""
    return _MaxPool3dTF.apply(x, tuple(int(k) for k in kernel), tuple(int(s) for s in stride))


#: weight gradients of the I3D trunk's serial units on the side stream (DMC_UNIT3D_WGRAD_SIDE=0: inside the unit's one call)
UNIT3D_WGRAD_SIDE = __import__("os").environ.get("DMC_UNIT3D_WGRAD_SIDE", "1") != "0"
UNIT3D_WGRAD_SIDE_MINW = int(__import__("os").environ.get("DMC_UNIT3D_WGRAD_SIDE_MINW", "56"))


class _ConvBnRelu3d(torch.autograd.Function):
    """relu?(BatchNorm3d(conv3d(x, w))) of a Unit3Dpy in training mode (code/dmcnet_I3D/network/i3d.py:390-398) on the
    bf16 kernels: the convolution's epilogue reduces the batch statistics, one streaming pass normalises and rectifies;
    the backward runs the ReLU + BatchNorm backward in two passes and feeds the data / weight gradient kernels."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, eps, momentum, relu, into=None):
        lib = _lib.load()
        if not (x.is_cuda and weight.is_cuda):
            raise _lib.DmcHipError("conv_bn_relu3d runs on the HIP extension only (no CPU fallback)")
        ctx.x_was_cl3 = x.is_contiguous(memory_format=_CL3)
        x = _as_cl3(x)
        wc = weight.detach()
        if not wc.is_contiguous():
            wc = wc.contiguous()
        n, cin, d, h, w = x.shape
        cout, _, kd, kh, kw = wc.shape
        geom = (n, d, h, w, cin, cout, kd, kh, kw)
        # ONE foreign call (pack + convolution with the statistics in its epilogue + BatchNorm / ReLU pass) and one
        # workspace: the trunk is bound by the host that issues its ~940 launches per micro-step (csrc/unit3d.hip)
        y = torch.empty((n, cout, d, h, w), dtype=torch.bfloat16, device=x.device, memory_format=_CL3)
        # `into`: a channel slice of a wider NDHWC tensor (an Inception block's output) that receives the result in place --
        # the block then needs no torch.cat; autograd sees a view of it as this op's output (the buffer itself has no history)
        if into is not None:
            ld = into.stride(4)
            if not (into.dtype == torch.bfloat16 and tuple(into.shape) == (n, cout, d, h, w) and ld >= cout and ld % 8 == 0
                    and into.storage_offset() % 8 == 0 and into.stride() == (d * h * w * ld, 1, h * w * ld, w * ld, ld)):
                raise ValueError("conv_bn_relu3d: `into` must be a channel slice of a channels_last_3d bf16 tensor of this op's shape")
            out = into
        else:
            out, ld = torch.empty_like(y), cout
        ws = torch.empty(lib.dmc_unit3d_bf16_fwd_workspace_bytes(*geom), dtype=torch.uint8, device=x.device)
        with _span("conv3d_bf16_fwd"):
            _lib.check(lib.dmc_unit3d_bf16_fwd_into(_lib.ptr(x), _lib.ptr(wc), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                                                    _lib.ptr(running_var), _lib.ptr(ws), _lib.ptr(y), _lib.ptr(out), ld, *geom, int(relu),
                                                    float(eps), float(momentum), _stream()), "dmc_unit3d_bf16_fwd_into")
        ctx.save_for_backward(x, weight, y, gamma, beta, ws)
        ctx.relu, ctx.geom = bool(relu), geom
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, weight, y, gamma, beta, ws = ctx.saved_tensors
        geom = ctx.geom
        n, d, h, w, cin, cout, kd, kh, kw = geom
        # a channel slice of a concatenated (Inception) gradient is read in place: NDHWC memory with a wider pixel stride
        ld = dout.stride(4) if dout.dim() == 5 else 0
        if not (dout.dtype == torch.bfloat16 and ld >= cout and ld % 8 == 0 and dout.storage_offset() % 8 == 0
                and dout.stride() == (d * h * w * ld, 1, h * w * ld, w * ld, ld)):
            dout, ld = _as_cl3(dout), cout
        dy = torch.empty_like(y)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        # the weight gradient on the side stream (see _on_wgrad_stream) for the units of the trunk's SERIAL part (the stem's
        # successors, 56-wide maps): their input gradient heads a dependent chain, nothing reads dw.  Inside the Inception
        # blocks the branches already overlap on their own streams.
        side = ctx.needs_input_grad[1] and UNIT3D_WGRAD_SIDE and w >= UNIT3D_WGRAD_SIDE_MINW and WGRAD_STREAM and _WGRAD_SCOPE[0] > 0
        dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] and not side else None
        bws = torch.empty(lib.dmc_unit3d_bf16_bwd_workspace_bytes(*geom), dtype=torch.uint8, device=x.device)
        with _span("conv3d_bf16_bwd"):
            _lib.check(lib.dmc_unit3d_bf16_bwd(_lib.ptr(dout), ld, _lib.ptr(x), _lib.ptr(y), _lib.ptr(ws), _lib.ptr(gamma),
                                               _lib.ptr(beta), _lib.ptr(bws), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dw),
                                               _lib.ptr(dgamma), _lib.ptr(dbeta), *geom, int(ctx.relu), _stream()),
                       "dmc_unit3d_bf16_bwd")
        if side:
            def launch():
                dw_ = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
                work_ = torch.empty(lib.dmc_conv3d_bf16_wgrad_bytes(*geom), dtype=torch.uint8, device=x.device)
                _lib.check(lib.dmc_conv3d_bf16_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw_), _lib.ptr(work_), *geom, _stream()),
                           "dmc_conv3d_bf16_wgrad")
                return dw_
            with _span("conv3d_bf16_wgrad"):
                dw = _on_wgrad_stream(weight, (x, dy), launch)
        if dx is not None and not ctx.x_was_cl3:
            dx = dx.contiguous()
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None


def conv_bn_relu3d_supported(x, conv, bn):
    """True if conv -> BatchNorm3d (training) [-> ReLU] of a Unit3Dpy can run as the fused bf16 op."""
    if not (torch.is_grad_enabled() and bn.training and bn.track_running_stats and bn.affine and bn.momentum is not None):
        return False
    if bn.weight.dtype != torch.float32 or not conv3d_bf16_supported(x, conv.weight, conv.stride, conv.padding):
        return False
    n, _, d, h, w = x.shape
    return bool(_lib.load().dmc_bn3d_bf16_supported(n * d * h * w, conv.out_channels))


def conv_bn_relu3d(x, conv, bn, relu=True, into=None):
    """relu?(bn(conv(x))) for a bf16 ``x`` (see conv_bn_relu3d_supported); updates the running statistics and
    ``num_batches_tracked`` as nn.BatchNorm3d does.  ``into``: see _ConvBnRelu3d.forward."""
    if bn.num_batches_tracked is not None:
        if _PENDING_COUNTERS is not None:
            _PENDING_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    return _ConvBnRelu3d.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                               bool(relu), into)


class _JoinSlices(torch.autograd.Function):
    """The concatenation that already happened: ``parts`` are channel slices of ``buf`` that their producers wrote in place
    (conv_bn_relu3d(..., into=...)).  Forward hands out ``buf``; backward hands each producer its slice of the gradient -- what
    torch.cat's backward does, without the forward copies."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[1] for p in parts]
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        outs, c = [], 0
        for wd in ctx.widths:
            outs.append(g[:, c:c + wd])
            c += wd
        return (None,) + tuple(outs)


def join_slices(buf, parts):
    return _JoinSlices.apply(buf, *parts)


class _Fanout4(torch.autograd.Function):
    """x handed to four consumers; backward adds their four gradients in ONE pass (dmc_add4_bf16) instead of the engine's three
    additions -- in the engine's order (the consumer created last delivers first) and with its bf16 rounding after every sum, so
    the result is bit for bit the engine's."""

    @staticmethod
    def forward(ctx, x):
        return tuple(x.view(x.shape) for _ in range(4))

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        gs = [g for g in (g3, g2, g1, g0) if g is not None]
        if (len(gs) == 4 and all(g.is_cuda and g.dtype == torch.bfloat16 and g.shape == gs[0].shape and g.stride() == gs[0].stride()
                                 and g.storage_offset() % 8 == 0 for g in gs)
                and gs[0].numel() % 8 == 0 and (gs[0].is_contiguous() or gs[0].is_contiguous(memory_format=_CL3))):
            out = torch.empty_like(gs[0])
            _lib.check(_lib.load().dmc_add4_bf16(_lib.ptr(gs[0]), _lib.ptr(gs[1]), _lib.ptr(gs[2]), _lib.ptr(gs[3]), _lib.ptr(out),
                                                 out.numel(), _stream()), "dmc_add4_bf16")
            return out
        r = None
        for g in gs:
            r = g if r is None else r + g
        return r


def fanout4(x):
    return _Fanout4.apply(x)


class _Stem3dBnRelu(torch.autograd.Function):
    """relu(BatchNorm3d(conv3d_1a_7x7(x))) of the I3D stem in a bf16 trunk (code/dmcnet_I3D/network/i3d.py:480-481,
    :390-398): forward on dmc_stem3d_bf16_fwd (statistics in its epilogue) + the fused BatchNorm3d / ReLU pass; the
    BatchNorm / ReLU backward on bn3d_bf16.hip, the weight gradient on dmc_stem3d_bf16_wgrad, the 2-channel data gradient on
    dmc_stem3d_bf16_dgrad (PyTorch-ROCm / MIOpen only for frames wider than 256)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, eps, momentum, relu):
        lib = _lib.load()
        if not (x.is_cuda and weight.is_cuda):
            raise _lib.DmcHipError("stem3d runs on the HIP extension only (no CPU fallback)")
        xc = x.detach().float().contiguous()
        wc = weight.detach().float().contiguous()
        n, _, t, h, w = xc.shape
        od, oh, ow = (t - 2) // 2 + 1, (h - 2) // 2 + 1, (w - 2) // 2 + 1
        y = torch.empty((n, 64, od, oh, ow), dtype=torch.bfloat16, device=x.device, memory_format=_CL3)
        work = _floats(lib.dmc_stem3d_bf16_workspace_bytes(n, t, h, w), x.device)
        nblk = lib.dmc_stem3d_bf16_stat_blocks(n, t, h, w)
        part = torch.empty((nblk, 64, 2), dtype=torch.float32, device=x.device)
        with _span("stem3d_fwd"):
            _lib.check(lib.dmc_stem3d_bf16_fwd(_lib.ptr(xc), _lib.ptr(wc), _lib.ptr(work), _lib.ptr(y), _lib.ptr(part), n, t, h, w,
                                               _stream()), "dmc_stem3d_bf16_fwd")
        stats = torch.empty(128, dtype=torch.float32, device=x.device)
        out = torch.empty_like(y)
        m = n * od * oh * ow
        with _span("bn3d_fwd"):
            _lib.check(lib.dmc_bn3d_bf16_fwd(_lib.ptr(y), _lib.ptr(part), nblk, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(stats),
                                             _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(out), m, 64, int(relu),
                                             float(eps), float(momentum), _stream()), "dmc_bn3d_bf16_fwd")
        ctx.save_for_backward(xc, weight, y, gamma, beta, stats)
        ctx.relu = bool(relu)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        xc, weight, y, gamma, beta, stats = ctx.saved_tensors
        n, _, od, oh, ow = y.shape
        m = n * od * oh * ow
        dout = _as_cl3(dout)
        dy = torch.empty_like(y)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        scratch = _floats(lib.dmc_bn3d_bf16_scratch_bytes(64), y.device)
        with _span("bn3d_bwd"):
            _lib.check(lib.dmc_bn3d_bf16_bwd(_lib.ptr(dout), 64, _lib.ptr(y), _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta),
                                             _lib.ptr(scratch), _lib.ptr(dy), _lib.ptr(dgamma), _lib.ptr(dbeta), m, 64,
                                             int(ctx.relu), _stream()), "dmc_bn3d_bf16_bwd")
        t, h, w = xc.shape[2:]
        dx = dw = None
        if ctx.needs_input_grad[1]:
            def launch():       # nothing below reads dw: on the side stream it runs next to the data gradient and the generator's backward
                dw_ = torch.empty((64, 2, 7, 7, 7), dtype=torch.float32, device=y.device)
                work_ = _floats(lib.dmc_stem3d_bf16_wgrad_workspace_bytes(n, t, h, w), y.device)
                _lib.check(lib.dmc_stem3d_bf16_wgrad(_lib.ptr(xc), _lib.ptr(dy), _lib.ptr(dw_), _lib.ptr(work_), n, t, h, w, _stream()),
                           "dmc_stem3d_bf16_wgrad")
                return dw_.to(weight.dtype)
            with _span("stem3d_wgrad"):
                dw = _on_wgrad_stream(weight, (xc, dy), launch)
        if ctx.needs_input_grad[0] and w <= 256:           # data gradient (2 channels): row GEMM + fold on the matrix cores
            dx = torch.empty((n, 2, t, h, w), dtype=torch.float32, device=y.device)
            work = _floats(lib.dmc_stem3d_bf16_dgrad_workspace_bytes(), y.device)
            with _span("stem3d_dgrad"):
                _lib.check(lib.dmc_stem3d_bf16_dgrad(_lib.ptr(dy), _lib.ptr(weight.detach().float().contiguous()), _lib.ptr(dx),
                                                     _lib.ptr(work), n, t, h, w, _stream()), "dmc_stem3d_bf16_dgrad")
        elif ctx.needs_input_grad[0]:                      # wider than the kernel's row tile: MIOpen, on the stock path's tensors
            xpad = torch.nn.functional.pad(xc.bfloat16(), (2, 3, 2, 3, 2, 3))
            with _span("stem3d_dgrad"):
                dxp, _, _ = torch.ops.aten.convolution_backward(
                    dy.contiguous(), xpad, weight.detach().bfloat16(), None, (2, 2, 2), (0, 0, 0), (1, 1, 1), False, (0, 0, 0), 1,
                    (True, False, False))
            dx = dxp[:, :, 2:2 + t, 2:2 + h, 2:2 + w].float()
        return dx, dw, dgamma, dbeta, None, None, None, None, None


def stem3d_supported(x, conv, bn):
    """True if the I3D stem unit (2 -> 64 channels, 7x7x7, stride 2, TF-"SAME") can take the bf16 HIP forward: a CUDA
    fp32 / bf16 cue inside a bf16 autocast region, BatchNorm3d in training mode."""
    if not (x.is_cuda and x.dim() == 5 and x.shape[1] == 2 and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return False
    if tuple(conv.weight.shape) != (64, 2, 7, 7, 7) or tuple(conv.stride) != (2, 2, 2) or conv.bias is not None:
        return False
    if not (torch.is_grad_enabled() and bn.training and bn.track_running_stats and bn.affine and bn.momentum is not None):
        return False
    return all(int(s) >= 2 and int(s) % 2 == 0 for s in x.shape[2:])


def stem3d_bn_relu(x, conv, bn, relu=True):
    if bn.num_batches_tracked is not None:
        if _PENDING_COUNTERS is not None:
            _PENDING_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    return _Stem3dBnRelu.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                               bool(relu))
