"""Training-step drivers: the inner loops of the reference's ``train()`` functions
(code/dmcnet/train.py:205-266, code/dmcnet_GAN/train.py:219-397) with the same optimiser policy
(:121-142 / :122-153), LR schedule (:398-408), meters (:380-395) and top-k accuracy (:411-424).

Differences that do not change results: losses and meters stay on the device (the reference
reads ``loss.data[0]`` 5-7 times per iteration, each a host sync); Adam parameter groups keep the
reference's one-group-per-tensor layout (so optimiser state dicts are interchangeable) but the
update is issued as a few multi-tensor launches instead of one per tensor; backward segments
whose results the reference discards at the next ``zero_grad`` are skipped.
"""
import contextlib
import importlib
import shutil

import numpy as np
import torch
import torch.nn.functional as F

from . import ops

_adam_kernel = importlib.import_module("torch.optim.adam").adam   # functional multi-tensor Adam


class GroupedAdam(torch.optim.Adam):
    """torch.optim.Adam semantics (coupled L2, bias-corrected), executed per distinct
    (lr, weight_decay, betas, eps) set with the multi-tensor kernels instead of per group:
    torch's fused Adam kernel on the GPU, the foreach kernels on the CPU."""

    @torch.no_grad()
    def step(self, closure=None):
        buckets = {}
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize"):
                raise NotImplementedError("amsgrad / maximize are not used by the reference")
            key = (float(g["lr"]), float(g["weight_decay"]), tuple(g["betas"]), float(g["eps"]))
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    # the fused kernel wants the step counter next to the parameter
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device if p.is_cuda else "cpu")
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif p.is_cuda and st["step"].device != p.device:      # e.g. a loaded state dict
                    st["step"] = st["step"].to(p.device)
                b = buckets.setdefault(key + (p.is_cuda,), ([], [], [], [], []))
                b[0].append(p); b[1].append(p.grad); b[2].append(st["exp_avg"])
                b[3].append(st["exp_avg_sq"]); b[4].append(st["step"])
        for (lr, wd, (b1, b2), eps, on_gpu), (ps, gs, ms, vs, steps) in buckets.items():
            # GPU: one fused multi-tensor kernel per bucket (the foreach path is ~10 launches)
            _adam_kernel(ps, gs, ms, vs, [], steps, foreach=not on_gpu, fused=on_gpu, amsgrad=False,
                         beta1=b1, beta2=b2, lr=lr, weight_decay=wd, eps=eps, maximize=False,
                         grad_scale=None, found_inf=None)
        return None


def make_optimizers(model, lr, weight_decay, lr_cls_mult, lr_mse_mult, lr_d_mult=None):
    """optimizer_cls, optimizer_gf[, optimizer_d]: one group per tensor, selected by key substring
    ('base_model' / 'gen_flow_model' / 'discriminator'), ``decay_mult`` 0 for biases,
    Adam(eps=1e-3) -- code/dmcnet/train.py:121-142, code/dmcnet_GAN/train.py:122-153."""
    plan = [("base_model", lr_cls_mult), ("gen_flow_model", lr_mse_mult)]
    if lr_d_mult is not None:
        plan.append(("discriminator", lr_d_mult))
    params = dict(model.named_parameters())
    opts = []
    for tag, mult in plan:
        groups = [{"params": v, "lr": lr, "lr_mult": mult,
                   "decay_mult": 0.0 if "bias" in k else 1.0}
                  for k, v in params.items() if tag in k]
        opts.append(GroupedAdam(groups, weight_decay=weight_decay, eps=0.001))
    for o in opts:   # what adjust_learning_rate does at epoch 0 with no decay step passed
        for g in o.param_groups:
            g["lr"] = lr * g["lr_mult"]
            g["weight_decay"] = weight_decay * g["decay_mult"]
    return opts


def adjust_learning_rate(optimizer, epoch, lr_steps, lr_decay, base_lr, weight_decay,
                         freeze=False, epoch_thre=500):
    """code/dmcnet/train.py:398-408 (``args.lr`` / ``args.weight_decay`` passed explicitly)."""
    lr = base_lr * lr_decay ** int(sum(epoch >= np.array(lr_steps)))
    wd = weight_decay
    if epoch < epoch_thre and freeze:
        lr, wd = 0, 0
    for g in optimizer.param_groups:
        g["lr"] = lr * g["lr_mult"]
        g["weight_decay"] = wd * g["decay_mult"]
    return lr


def accuracy(output, target, topk=(1,)):
    """Precision@k in percent, as device tensors (code/dmcnet/train.py:411-424)."""
    _, pred = output.topk(max(topk), 1, True, True)
    correct = pred.t().eq(target.view(1, -1))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


class AverageMeter(object):
    """code/dmcnet/train.py:380-395; ``val`` may be a device tensor (no sync until read)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count += n
        self.avg = self.sum / self.count


def save_checkpoint(state, is_best, model_prefix, representation, filename="checkpoint.pth.tar"):
    """code/dmcnet/train.py:372-377.  ``state`` carries epoch, arch, state_dict, best_prec1,
    optimizer_cls, optimizer_gf[, optimizer_d]; state_dict keys carry the 'module.' prefix the
    reference's DataParallel wrapper adds (use :func:`reference_state_dict`)."""
    name = "_".join((model_prefix, representation.lower(), filename))
    torch.save(state, name)
    if is_best:
        shutil.copyfile(name, "_".join((model_prefix, representation.lower(), "model_best.pth.tar")))
    return name


def reference_state_dict(model):
    return {"module." + k: v for k, v in model.state_dict().items()}


def load_reference_weights(model, state_dict, strict=False):
    """--weights semantics: drop the first key component ('module.'), strict=False
    (code/dmcnet/train.py:64-68)."""
    stripped = {".".join(k.split(".")[1:]): v for k, v in state_dict.items()}
    return model.load_state_dict(stripped, strict=strict)


@contextlib.contextmanager
def _without_param_grads(*modules):
    """Build the graph with these modules' parameters as constants: their weight gradients --
    which the reference computes and then discards at the next zero_grad -- are never computed;
    gradients still flow THROUGH the modules to their inputs."""
    ps = [p for m in modules for p in m.parameters() if p.requires_grad]
    for p in ps:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in ps:
            p.requires_grad_(True)


FLOW_LOSSES = ("MSELoss", "SmoothL1Loss", "L1")


def flow_loss(kind, gen_flow, flow, att_flow=None):
    """``criterion_mse`` of the reference (``--loss_mse``, code/dmcnet/train.py:166-172): MSELoss is
    the HIP kernel (every shipped recipe); SmoothL1Loss / L1 and the attention-weighted form
    ``criterion(att * gen, att * flow)`` (code/dmcnet/train.py:335, code/dmcnet_GAN/train.py:352),
    whose gradient also reaches ``att`` through the target, run on the stock PyTorch-ROCm ops."""
    if kind not in FLOW_LOSSES:
        raise ValueError("loss_mse must be one of %s, got %r" % (FLOW_LOSSES, kind))
    if att_flow is not None:
        gen_flow, flow = att_flow * gen_flow, att_flow * flow
    elif kind == "MSELoss":
        return ops.flow_mse(gen_flow, flow)
    fn = {"MSELoss": F.mse_loss, "SmoothL1Loss": F.smooth_l1_loss, "L1": F.l1_loss}[kind]
    return fn(gen_flow, flow)


class DmcnetTrainStep(object):
    """One iteration of code/dmcnet/train.py:221-266.  ``att=1`` (a model built with att=1 returns
    (base_out, gen_flow, att_flow)) uses the attention-weighted reconstruction loss the reference's
    validate() computes (:320-335; its train() cannot unpack that model's 3-tuple at all)."""

    def __init__(self, model, num_segments, lr_cls, lr_mse, lr, weight_decay, lr_cls_mult,
                 lr_mse_mult, reducer=None, loss_mse="MSELoss", att=0):
        if loss_mse not in FLOW_LOSSES:
            raise ValueError("loss_mse must be one of %s, got %r" % (FLOW_LOSSES, loss_mse))
        if bool(att) != bool(getattr(model, "att", 0)):
            raise ValueError("att=%r but the model was built with att=%r" % (att, getattr(model, "att", 0)))
        self.model, self.num_segments = model, num_segments
        self.lr_cls, self.lr_mse = lr_cls, lr_mse
        self.loss_mse, self.att = loss_mse, int(att)
        self.optimizer_cls, self.optimizer_gf = make_optimizers(model, lr, weight_decay,
                                                                lr_cls_mult, lr_mse_mult)
        self.reducer = reducer

    def step(self, batch, freeze=False):
        input_flow, input_mv, input_residual, target = batch
        flow = input_flow.reshape((-1,) + tuple(input_mv.shape[-3:]))
        self.optimizer_cls.zero_grad(set_to_none=True)
        self.optimizer_gf.zero_grad(set_to_none=True)
        if self.loss_mse == "MSELoss" and self.att == 0 and hasattr(self.model, "forward_with_flow_mse"):
            # the MSE is reduced in the epilogue of the kernel that writes gen_flow (no re-read of it)
            output, gen_flow, loss_mse = self.model.forward_with_flow_mse(input_mv, input_residual, flow)
            loss_cls, consensus = ops.consensus_ce(output, target, self.num_segments)
        else:
            outs = self.model(input_mv, input_residual)
            output, gen_flow = outs[0], outs[1]
            att_flow = outs[2] if self.att == 1 else None
            loss_cls, consensus = ops.consensus_ce(output, target, self.num_segments)
            loss_mse = flow_loss(self.loss_mse, gen_flow, flow, att_flow)
        loss = loss_cls * self.lr_cls + loss_mse * self.lr_mse
        if self.reducer is not None:
            self.reducer.begin()
        with ops.wgrad_side_stream():            # the classifier's weight gradients overlap the rest of the backward pass
            if freeze:
                (loss_mse * self.lr_mse).backward()
            else:
                loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        if not freeze:
            self.optimizer_cls.step()
        self.optimizer_gf.step()
        return {"loss": loss.detach(), "loss_cls": loss_cls.detach(), "loss_mse": loss_mse.detach(),
                "output": consensus, "gen_flow": gen_flow.detach()}


class GanTrainStep(object):
    """Iterations of code/dmcnet_GAN/train.py:236-371: even ``i`` trains the discriminator and
    the classifier, odd ``i`` trains the generator."""

    def __init__(self, model, num_segments, lr_cls, lr_adv_g, lr_adv_d, lr_mse, lr, weight_decay,
                 lr_cls_mult, lr_mse_mult, lr_d_mult, reducer=None, loss_mse="MSELoss", att=0):
        if loss_mse not in FLOW_LOSSES:
            raise ValueError("loss_mse must be one of %s, got %r" % (FLOW_LOSSES, loss_mse))
        if bool(att) != bool(getattr(model, "att", 0)):
            raise ValueError("att=%r but the model was built with att=%r" % (att, getattr(model, "att", 0)))
        self.model, self.num_segments = model, num_segments
        self.loss_mse, self.att = loss_mse, int(att)
        self.lr_cls, self.lr_adv_g, self.lr_adv_d, self.lr_mse = lr_cls, lr_adv_g, lr_adv_d, lr_mse
        self.optimizer_cls, self.optimizer_gf, self.optimizer_d = make_optimizers(
            model, lr, weight_decay, lr_cls_mult, lr_mse_mult, lr_d_mult)
        self.reducer = reducer

    def step(self, batch, i):
        input_flow, input_mv, input_residual, target = batch
        flow = input_flow.reshape((-1,) + tuple(input_mv.shape[-3:]))
        n = target.numel() * self.num_segments
        valid = torch.ones(n, dtype=torch.int64, device=target.device)
        fake = torch.zeros(n, dtype=torch.int64, device=target.device)
        for o in (self.optimizer_cls, self.optimizer_gf, self.optimizer_d):
            o.zero_grad(set_to_none=True)
        out = {}
        if i % 2 == 0:
            # only optimizer_cls and optimizer_d step (GAN train.py:301-302): the generator's
            # gradients would be thrown away
            with _without_param_grads(self.model.gen_flow_model):
                output, validity, gen_flow = self.model(input_mv, input_residual, flow)[:3]   # att=1: att_flow unused (GAN train.py:266-267)
            loss_cls, consensus = ops.consensus_ce(output, target, self.num_segments)
            loss_adv, _ = ops.consensus_ce(validity, torch.cat((fake, valid), 0), 1)
            loss = loss_cls * self.lr_cls + loss_adv * self.lr_adv_d
            self._backward(loss)
            self.optimizer_cls.step()
            self.optimizer_d.step()
        else:
            # only optimizer_gf steps (GAN train.py:371): classifier / discriminator weight
            # gradients would be thrown away; their BatchNorm running statistics still update
            with _without_param_grads(self.model.base_model, self.model.discriminator):
                outs = self.model(input_mv, input_residual)
            output, validity, gen_flow = outs[:3]
            loss_cls, consensus = ops.consensus_ce(output, target, self.num_segments)
            loss_adv, _ = ops.consensus_ce(validity, valid, 1)
            loss_mse = flow_loss(self.loss_mse, gen_flow, flow, outs[3] if self.att == 1 else None)   # GAN train.py:349-352
            loss = loss_cls * self.lr_cls + loss_adv * self.lr_adv_g + loss_mse * self.lr_mse
            self._backward(loss)
            self.optimizer_gf.step()
            out["loss_mse"] = loss_mse.detach()
        out.update(loss=loss.detach(), loss_cls=loss_cls.detach(), loss_adv=loss_adv.detach(),
                   output=consensus, validity=validity.detach(), gen_flow=gen_flow.detach())
        return out

    def _backward(self, loss):
        if self.reducer is not None:
            self.reducer.begin()
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()


_SETTLED = [False]


def settle_host():
    """Once per process, after the first steps: ``gc.collect(); gc.freeze()``.  The interpreter then holds ~10^6 long-lived
    objects (torch, this package, the model, autograd closures); a full (generation-2) collection walks all of them --
    40-100 ms of host time during which nothing is launched (measured: a 13 ms GAN step became 82 ms, the GPU idle for the
    difference).  Frozen objects are skipped by later collections, which then cost well under a millisecond.  Host-side
    only; no effect on results."""
    if _SETTLED[0]:
        return
    import gc
    gc.collect()
    gc.freeze()
    _SETTLED[0] = True
