"""Trainer policy of the I3D variant (BASELINE config 5): optimizer set, step-count LR schedule,
two-stage switch, D / G alternation with gradient accumulation, and the data-parallel gradient exchange.

Restates, with citations, what code/dmcnet_I3D/train_model.py:68-179,219-238 builds and what
``model.fit`` (code/dmcnet_I3D/train/model.py:286-491) does per micro-batch; pinned bit-for-bit on the
CPU by tests/golden/g10_i3d_trainer.npz, which the reference's own ``fit`` loop produced
(tests/golden/make_golden_i3d.py).  Quirks are kept, not fixed -- they are listed where they occur.
"""
import torch
import torch.distributed as dist

from . import ops
from .i3d import i3d_losses


class MultiFactorScheduler(object):
    """Step-count LR schedule (code/dmcnet_I3D/train/lr_scheduler.py:22-61): ``lr *= factor`` each
    time the step counter passes an entry of ``steps``; the first 99 updates return HALF the current
    lr (warm-up, :59-60)."""

    def __init__(self, steps, base_lr=0.01, factor=0.1, step_counter=0):
        steps = list(steps)
        if not steps:
            raise ValueError("steps must be a non-empty increasing list")
        for i, s in enumerate(steps):
            if i != 0 and steps[i] <= steps[i - 1]:
                raise ValueError("Schedule step must be an increasing integer list")
            if s < 1:
                raise ValueError("Schedule step must be greater or equal than 1 round")
        if factor > 1.0:
            raise ValueError("Factor must be no more than 1 to make lr reduce")
        self.steps, self.factor = steps, factor
        self.step_counter, self.base_lr = step_counter, base_lr
        self.lr, self.cursor = base_lr, 0

    def get_lr(self):
        return self.lr

    def update(self):
        self.step_counter += 1
        if self.cursor >= len(self.steps):
            return self.lr
        while self.steps[self.cursor] < self.step_counter:
            self.lr *= self.factor
            self.cursor += 1
            if self.cursor >= len(self.steps):
                return self.lr
        if self.step_counter < 100:
            return self.lr / 2.0
        return self.lr


def split_parameters(net, modality="flow+mp4", fine_tune=False):
    """(base, new, generator, discriminator, lr_mul) by name prefix, code/dmcnet_I3D/train_model.py:68-106:
    'flow+mp4': ``gen_flow_model*`` / ``discriminator*`` get their own optimizers, ``conv3d_0c_1x1*`` and
    ``classifier*`` are the new layers (lr_mult 1), the rest of the trunk is base (lr_mult 0.5, or 0.2
    when fine-tuning); other modalities: everything new unless fine-tuning."""
    base, new, gf, d = [], [], [], []
    for name, p in net.named_parameters():
        head = name.startswith("conv3d_0c_1x1") or name.startswith("classifier")
        if modality == "flow+mp4":
            if name.startswith("gen_flow_model"):
                gf.append(p)
            elif name.startswith("discriminator"):
                d.append(p)
            elif head:
                new.append(p)
            else:
                base.append(p)
        elif fine_tune:
            (new if head else base).append(p)
        else:
            new.append(p)
    lr_mul = (0.2 if fine_tune else 0.5) if modality == "flow+mp4" else 0.2
    return base, new, gf, d, lr_mul


def make_optimizers(net, lr_base, lr_base2, optim="adam", adv=1.0, modality="flow+mp4", fine_tune=False,
                    weight_decay=1e-4):
    """The five optimizers of code/dmcnet_I3D/train_model.py:122-179 as a dict:
    ``optimizer`` / ``optimizer_2`` (trunk, stage 1 / stage 2; groups base lr_mult, new 1.0),
    ``optimizer_3`` (discriminator, Adam eps 1e-3, when adv > 0), ``optimizer_mse`` (generator, stage 1:
    Adam eps 1e-8) / ``optimizer_mse_2`` (stage 2: Adam eps 1e-3) for 'flow+mp4'."""
    base, new, gf, d, lr_mul = split_parameters(net, modality, fine_tune)
    # on the GPU torch's FUSED multi-tensor Adam (train.GroupedAdam: the same update, one kernel per distinct hyper-parameter set
    # instead of the ~10 foreach launches per optimizer -- 0.4 ms of a 15 ms micro-step with iter_size 1); on the CPU torch's own
    # class, whose arithmetic the policy golden G10 pins bit for bit
    from . import train as _train
    on_gpu = all(p.is_cuda for p in list(base) + list(new) + list(gf) + list(d)) and __import__("os").environ.get("DMC_I3D_FUSED_ADAM", "1") != "0"
    Adam = _train.GroupedAdam if on_gpu else torch.optim.Adam

    def trunk(lr):
        groups = [{"params": base, "lr_mult": lr_mul}, {"params": new, "lr_mult": 1.0}]
        if optim == "adam":
            return Adam(groups, lr=lr, weight_decay=weight_decay)
        return torch.optim.SGD(groups, lr=lr, momentum=0.9, weight_decay=weight_decay, nesterov=True)

    def gen(lr, eps):
        if optim == "adam":
            return Adam(gf, lr=lr, weight_decay=weight_decay, eps=eps)
        return torch.optim.SGD(gf, lr=lr, momentum=0.9, weight_decay=weight_decay, nesterov=True)

    out = {"optimizer": trunk(lr_base), "optimizer_2": trunk(lr_base2), "optimizer_3": None,
           "optimizer_mse": None, "optimizer_mse_2": None}
    if adv > 0.0:
        out["optimizer_3"] = Adam(d, lr=lr_base, weight_decay=weight_decay, eps=0.001)
    if modality == "flow+mp4":
        out["optimizer_mse"] = gen(lr_base, 1e-08)
        out["optimizer_mse_2"] = gen(lr_base2, 0.001)
    return out


def adjust_learning_rate(optimizer, lr, epoch=0, epoch_thre=0):
    """code/dmcnet_I3D/train/model.py:267-281: groups with lr_mult 0.2 / 0.5 (the pretrained trunk) are
    frozen (lr 0) while ``epoch + 1 <= epoch_thre``; afterwards 0.5 becomes 1.0."""
    for g in optimizer.param_groups:
        m = g.get("lr_mult", 1.0)
        if m == 0.2 or m == 0.5:
            if epoch_thre > 0 and epoch + 1 <= epoch_thre:
                m = 0.0
            elif m == 0.5:
                m = 1.0
        g["lr"] = lr * m


class I3DTrainer(object):
    """``model.fit``'s training half, one micro-batch per :meth:`step`
    (code/dmcnet_I3D/train/model.py:345-491).

    Kept as written in the reference:
      * the accumulation counter ``i`` is shared by the D and G phases and only stepping resets it;
      * a phase divides by ``iter_size`` and steps only ITS optimizers and zeroes only THEIR gradients, so the
        generator's gradients of a D phase are still there when the G phase steps it, and the trunk's /
        discriminator's gradients of a G phase flow into the next D-phase step;
      * with a discriminator, the G phase of epoch 0 multiplies the classification loss by 0 (:437);
      * with a discriminator, the G phase re-uses the ``lr`` of the preceding D phase during stage 1 and
        ``lr_d`` is only refreshed during stage 1 (:366-382,444-459);
      * ``detach`` does not detach anything: ``fit`` never passes it to the forward, it only sets the trunk's
        stage-1 learning rate to 0 (:371-374,449-452).
    ``world_size > 1``: the gradients of the optimizers about to step are averaged over the ranks (one
    flat all-reduce per optimizer) between the last backward of the window and the step -- the RCCL
    counterpart of the nn.DataParallel wrap at code/dmcnet_I3D/train_model.py:120."""

    def __init__(self, net, optimizers, lr_scheduler, lr_scheduler2=None, lr_scheduler3=None, adv=1.0,
                 iter_size=1, epoch_thre=1, detach=False, group=None, losses_fn=None):
        #: the loss assembly (``i3d.i3d_losses``: this package's HIP reductions, CUDA tensors only); the CPU policy tests
        #: pass the oracle's restatement of the same reference lines
        self.losses_fn = i3d_losses if losses_fn is None else losses_fn
        self.net, self.adv, self.iter_size = net, adv, iter_size
        self.epoch_thre, self.detach, self.group = epoch_thre, detach, group
        self.optimizer = optimizers["optimizer"]
        self.optimizer_2 = optimizers.get("optimizer_2")
        self.optimizer_3 = optimizers.get("optimizer_3")
        self.optimizer_mse = optimizers.get("optimizer_mse")
        self.optimizer_mse_2 = optimizers.get("optimizer_mse_2")
        self.lr_scheduler, self.lr_scheduler2, self.lr_scheduler3 = lr_scheduler, lr_scheduler2, lr_scheduler3
        self.i = 0
        self.note = True
        self.lr = self.lr_d = None
        for o in (self.optimizer, self.optimizer_2, self.optimizer_mse, self.optimizer_mse_2, self.optimizer_3):
            if o is not None:
                o.zero_grad()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if self.world > 1:       # parameters AND buffers (BatchNorm running statistics) start out as rank 0's
            from .ddp import broadcast_coalesced
            broadcast_coalesced([p.data for p in net.parameters()] + [b for b in net.buffers() if b.numel()], 0, group)
        self.exchanged = []          # [(optimizer attribute, bytes)] of the last stepping micro-batch
        # hipGraph mode (enable_graphs): forward + losses + backward of a micro-batch replayed from a captured graph
        self.graph_warmup = None
        self._graphs, self._graph_seen, self._graph_pool = {}, {}, None
        self._static = None          # (data, target): the graphs' input buffers, shared by every captured phase
        self._acc = {}               # parameter -> persistent gradient buffer (never graph memory)

    # --------------------------------------------------------------------------------------- hipGraph mode
    def enable_graphs(self, warmup=1):
        """Replay forward + losses + backward of a micro-batch from a hipGraph, one graph per phase kind (D; G with the
        classification loss x 0; G) captured the first time a kind comes up after ``warmup`` eager occurrences.  The I3D
        micro-step is ~840 kernel launches of ~20 us: eager, the host needs 13-15 ms to issue them (and more when eight ranks
        share a host); a replay costs it microseconds.  What stays eager is the POLICY -- which optimizer steps, with which
        learning rate, the gradient exchange, Adam -- so the trainer's behaviour (golden G10) does not depend on the mode.
        Gradients: a replay writes the micro-batch's gradients into the graph's own memory; they are copied / added into
        persistent buffers exactly as autograd's AccumulateGrad would (``old + new``, same addends), so accumulation over
        ``iter_size`` micro-batches and over phases (a phase zeroes only ITS optimizers' gradients) is unchanged.
        Inputs are copied into static buffers (``static_batch()``: a loader can fill them directly and pass them in)."""
        self.graph_warmup = int(warmup)

    def static_batch(self):
        """(data, target) input buffers of the captured graphs, or None before the first capture; passing these very tensors
        to ``step`` skips the copy."""
        return self._static

    def _capture(self, key, data, target, stage, combine):
        net = self.net
        if self._static is None:
            self._static = (data.clone(), target.clone())
            self._graph_pool = torch.cuda.graph_pool_handle()
        sdata, starget = self._static
        if sdata.shape != data.shape or starget.shape != target.shape:
            raise ValueError("hipGraph mode: the micro-batch shape changed (%s -> %s); graphs are captured for one shape"
                             % (tuple(sdata.shape), tuple(data.shape)))
        params = [p for p in net.parameters() if p.requires_grad]
        stash = [p.grad for p in params]
        for p in params:
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(graph, pool=self._graph_pool):
                out, losses = self.losses_fn(net, sdata, starget, stage=stage, detach=False)
                with ops.wgrad_side_stream():
                    combine(losses).backward()
            produced = [(p, p.grad) for p in params if p.grad is not None]
        finally:
            for p, t in zip(params, stash):
                p.grad = t
        ent = {"graph": graph, "out": out, "losses": losses, "params": [p for p, _ in produced], "grads": [g for _, g in produced]}
        self._graphs[key] = ent
        return ent

    def _fwd_bwd(self, key, data, target, stage, combine):
        """Forward + losses + backward of one micro-batch; returns (logits, losses).  Eager, or -- graph mode, CUDA, after
        the warm-up occurrences of this phase kind -- one replay."""
        use_graph = self.graph_warmup is not None and data.is_cuda and ops.PROBE is None    # (HIP-event spans need eager launches)
        ent = self._graphs.get(key) if use_graph else None
        if ent is None:
            seen = self._graph_seen.get(key, 0)
            if not use_graph or seen < self.graph_warmup:
                self._graph_seen[key] = seen + 1
                out, losses = self.losses_fn(self.net, data, target, stage=stage, detach=False)
                self._backward(combine(losses))
                return out, losses
            ent = self._capture(key, data, target, stage, combine)
        sdata, starget = self._static
        if data is not sdata:
            sdata.copy_(data, non_blocking=True)
        if target is not starget:
            starget.copy_(target, non_blocking=True)
        ent["graph"].replay()
        # AccumulateGrad, outside the graph: fresh gradients are copied into persistent buffers, or added to what is there
        fresh_p, fresh_g, add_old, add_new = [], [], [], []
        for p, g in zip(ent["params"], ent["grads"]):
            if p.grad is None:
                buf = self._acc.get(p)
                if buf is None or buf.shape != g.shape or buf.stride() != g.stride():
                    buf = self._acc[p] = torch.empty_like(g)
                fresh_p.append(buf); fresh_g.append(g)
                p.grad = buf
            else:
                add_old.append(p.grad); add_new.append(g)
        if fresh_p:
            torch._foreach_copy_(fresh_p, fresh_g)
        if add_old:
            torch._foreach_add_(add_old, add_new)
        return ent["out"], ent["losses"]

    # ---------------------------------------------------------------------------------------
    def _exchange(self, name, opt):
        grads = [p.grad for g in opt.param_groups for p in g["params"] if p.grad is not None]
        nbytes = sum(t.numel() * t.element_size() for t in grads)
        self.exchanged.append((name, nbytes))
        if self.world == 1 or not grads:
            return
        flat = torch.cat([t.reshape(-1) for t in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)
        # one multi-tensor copy back instead of one small launch per gradient (102 for the trunk)
        views, off = [], 0
        for t in grads:
            views.append(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        torch._foreach_copy_(grads, views)

    def comm_summary(self):
        """The ``comm`` object of ``bench.py --gpus N --config i3d`` (same keys as ddp.GradBucketReducer.comm_summary where
        they apply): which communicator, and what the LAST stepping micro-batch exchanged per optimizer.  The exchange is one
        blocking all-reduce per stepping optimizer after the window's last backward (the recipe's iter_size 32 amortises it
        over 32 micro-batches), so all of it is exposed."""
        by_set = {}
        for name, nbytes in self.exchanged:
            by_set[name] = by_set.get(name, 0) + nbytes
        return {"backend": dist.get_backend(self.group) if dist.is_initialized() else None, "world_size": self.world,
                "reduce_op": "sum, then / world", "last_step_bytes_by_set": by_set,
                "last_step_launched_from": ["after backward (blocking)"] if self.exchanged else [],
                "exposed_wait_ms_per_step": None}

    def _scale(self, opt):
        if self.iter_size != 1:
            grads = [p.grad for g in opt.param_groups for p in g["params"]]
            if all(t is not None and t.is_cuda for t in grads):
                # ``t /= python_scalar`` on a CUDA tensor is lowered to a multiplication by the reciprocal (the division
                # kernel's CPU-scalar path); the multi-tensor form reproduces exactly that for every iter_size, not only
                # for powers of two
                torch._foreach_mul_(grads, 1.0 / self.iter_size)
            else:
                for t in grads:
                    t /= self.iter_size

    def _backward(self, loss):
        """``loss.backward()`` with the gradient ACCUMULATION of micro-batches 2 .. iter_size done by one multi-tensor add
        instead of one small ``grad += new`` launch per parameter (autograd's AccumulateGrad: 102 launches per micro-batch
        for the I3D trunk): the gradients accumulated so far are set aside, backward() fills fresh ones, and
        ``old + new`` is formed per parameter exactly as AccumulateGrad would (fp32 addition commutes): bit-identical."""
        params = [p for p in self.net.parameters() if p.requires_grad]
        stash = [p.grad for p in params]
        if not all(t is None or t.is_cuda for t in stash):
            loss.backward()
            return
        # (a post-accumulate hook -- ddp.GradBucketReducer -- would see the fresh micro-batch gradient instead of the window's sum)
        assert not any(getattr(p, "_post_accumulate_grad_hooks", None) for p in params), \
            "I3DTrainer accumulates outside autograd: post-accumulate gradient hooks would see partial gradients"
        for p in params:
            p.grad = None
        # no gradient to accumulate into during this pass: weight gradients may run on the side stream (ops._on_wgrad_stream:
        # the stem's and the serial units' overlap the data-gradient chain down to the generator); joined on exit
        try:
            with ops.wgrad_side_stream():
                loss.backward()
        except BaseException:
            # a failed pass (out of memory, a DMC_E_* from a kernel) must not lose the window's accumulated gradients
            for p, t in zip(params, stash):
                p.grad = t
            raise
        old, new = [], []
        for p, t in zip(params, stash):
            if t is None:
                continue
            if p.grad is None:
                p.grad = t
            elif p.grad.shape == t.shape and p.grad.dtype == t.dtype and p.grad.layout == t.layout:
                old.append(t); new.append(p.grad); p.grad = t
            else:
                t += p.grad
                p.grad = t
        if old:
            torch._foreach_add_(old, new)

    def step(self, data, target, i_epoch, i_batch):
        """One micro-batch of epoch ``i_epoch``; returns (logits, losses, phase, stepped)."""
        net, gan, joint = self.net, self.optimizer_3 is not None, self.optimizer_mse is not None
        if joint and i_epoch == self.epoch_thre and self.note:     # :348-352: second-stage optimizers
            self.optimizer, self.optimizer_mse, self.note = self.optimizer_2, self.optimizer_mse_2, False
        d_phase = gan and i_batch % (2 * self.iter_size) < self.iter_size
        stage1 = i_epoch + 1 <= self.epoch_thre
        self.exchanged = []
        stepped = False
        adv = self.adv
        if not joint:
            out = net(data)
            losses = [torch.nn.functional.cross_entropy(out, target)]
            self._backward(losses[0])
        elif d_phase:
            # (fit() never forwards its ``detach`` flag to the network, :355,:414-416: the classifier always
            # sees the undetached cue; ``detach`` only zeroes the trunk's stage-1 learning rate below)
            out, losses = self._fwd_bwd(("D", adv), data, target, "D", lambda l: l[0] + adv * l[2])
        elif not gan:
            out, losses = self._fwd_bwd(("G2",), data, target, None, lambda l: l[0] + l[1])
        elif i_epoch < 1:
            out, losses = self._fwd_bwd(("G0", adv), data, target, "D", lambda l: 0.0 * l[0] + l[1] + adv * l[2])
        else:
            out, losses = self._fwd_bwd(("G", adv), data, target, "D", lambda l: l[0] + l[1] + adv * l[2])
        if d_phase:
            if joint:
                if stage1:
                    self.lr = self.lr_scheduler.update()
                    self.lr_scheduler2.update()
                    self.lr_d = self.lr_scheduler3.update()
                    lr1 = 0.0 if self.detach else self.lr
                else:
                    self.lr = self.lr_scheduler2.update()
                    lr1 = self.lr
                adjust_learning_rate(self.optimizer, lr1, i_epoch, self.epoch_thre)
                adjust_learning_rate(self.optimizer_3, self.lr_d)
            else:
                adjust_learning_rate(self.optimizer, self.lr_scheduler.update())
            self.i += 1
            if self.i % self.iter_size == 0:
                self._exchange("optimizer", self.optimizer)
                self._exchange("optimizer_3", self.optimizer_3)
                self._scale(self.optimizer)
                self._scale(self.optimizer_3)
                self.optimizer.step(); self.optimizer.zero_grad()
                self.optimizer_3.step(); self.optimizer_3.zero_grad()
                self.i, stepped = 0, True
            return out.detach(), [l.detach() for l in losses], "D", stepped
        # ---- G phase (or the only phase without a discriminator): its backward ran above ----
        if joint:
            if stage1:
                if not gan:
                    self.lr = self.lr_scheduler.update()
                self.lr_scheduler2.update()
                lr1 = 0.0 if self.detach else self.lr
            else:
                self.lr = self.lr_scheduler2.update()
                lr1 = self.lr
            if not gan:
                adjust_learning_rate(self.optimizer, lr1, i_epoch, self.epoch_thre)
            adjust_learning_rate(self.optimizer_mse, self.lr)
        else:
            adjust_learning_rate(self.optimizer, self.lr_scheduler.update())
        self.i += 1
        if self.i % self.iter_size == 0:
            if not gan:
                self._exchange("optimizer", self.optimizer)
            if joint:
                self._exchange("optimizer_mse", self.optimizer_mse)
            if not gan:
                self._scale(self.optimizer)
            if joint:
                self._scale(self.optimizer_mse)
            if not gan:
                self.optimizer.step(); self.optimizer.zero_grad()
            if joint:
                self.optimizer_mse.step(); self.optimizer_mse.zero_grad()
            self.i, stepped = 0, True
        return out.detach(), [l.detach() for l in losses], "G", stepped


def recipe_trainer(net, batch_size=3, world_size=1, lr_base=4e-4, lr_base2=4e-4, lr_d=2e-3, lr_factor=0.2,
                   lr_steps=(400000, 800000), iter_size=32, adv=1.0, epoch_thre=6, detach=True, optim="adam",
                   group=None):
    """Trainer with the shipped recipe's settings (code/dmcnet_I3D/train.sh: batch 3, iter-size 32, Adam,
    lr-base / lr-base2 4e-4, lr-d 2e-3, lr-factor 0.2, adv 1, epoch-thre 6, detach 1) and the schedulers as
    code/dmcnet_I3D/train_model.py:219-238 builds them (steps in samples / (batch * workers))."""
    opts = make_optimizers(net, lr_base, lr_base2, optim=optim, adv=adv)
    steps = [int(x / (batch_size * world_size)) for x in lr_steps]
    mk = lambda base: MultiFactorScheduler(steps, base_lr=base, factor=lr_factor)
    return I3DTrainer(net, opts, mk(lr_base), mk(lr_base2), mk(lr_d), adv=adv, iter_size=iter_size,
                      epoch_thre=epoch_thre, detach=detach, group=group)
