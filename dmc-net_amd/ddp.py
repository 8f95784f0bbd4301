"""Data-parallel gradient exchange: one process per GPU, one RCCL all-reduce per gradient bucket,
launched from autograd hooks so that it overlaps the rest of the backward pass.

Replaces ``torch.nn.DataParallel(model, device_ids=args.gpus)`` (code/dmcnet/train.py:117,
code/dmcnet_GAN/train.py:118): no per-iteration parameter broadcast, no activation gather -- each
rank computes its own mean loss over its shard of the batch and the averaged gradients equal the
gathered-batch gradients when shards are equal.  BatchNorm statistics stay per rank, as they are
per replica under DataParallel.

Bucket plan: the reference steps up to three optimizers whose parameters are
selected by key substring (``base_model`` / ``gen_flow_model`` / ``discriminator``,
code/dmcnet/train.py:125-129, code/dmcnet_GAN/train.py:135), and each training phase produces
gradients for a subset of them only:

  dmcnet step      classifier graph -> base_model (44.8 MB);  independent MSE graph -> gen_flow_model (18 KB)
  GAN D step       base_model + discriminator (2.16 MB); the generator's gradients are discarded
                   (code/dmcnet_GAN/train.py:301-302)
  GAN G step       gen_flow_model only (code/dmcnet_GAN/train.py:371)

so buckets are cut PER PARAMETER SET and never span two sets: the generator's 18 KB travel alone
(they are ready as soon as the short MSE graph is done and, in the G step, nothing else is
reduced), a set that produced no gradient costs nothing, and ``finish()`` only waits.

xGMI is point to point, so a ring all-reduce is bound by one link (~153 GB/s): the 44.8 MB of
ResNet-18 gradients cost ~0.5 ms; a few large buckets in reverse execution order are enough.
"""
import time

import torch
import torch.distributed as dist

from . import ops

SET_TAGS = ("base_model", "gen_flow_model", "discriminator")


def param_sets_of(model, tags=SET_TAGS):
    """[(tag, [parameters])] by key substring, the reference's optimizer routing; parameters that
    match no tag form a trailing ``"other"`` set."""
    sets = [(t, []) for t in tags]
    other = []
    for k, p in model.named_parameters():
        for t, lst in sets:
            if t in k:
                lst.append(p)
                break
        else:
            other.append(p)
    sets = [(t, lst) for t, lst in sets if lst]
    if other:
        sets.append(("other", other))
    return sets


class GradBucketReducer(object):
    def __init__(self, params, bucket_bytes=16 << 20, group=None, broadcast_from=0, state=None):
        """``params``: either parameters in forward order (one set), or a list of
        ``(name, [parameters])`` sets (see :func:`param_sets_of`).  Within a set buckets are filled
        in reverse, the order in which backward produces gradients; no bucket spans two sets.
        ``state``: further tensors that must start out equal on every rank and are broadcast with the
        parameters -- frozen parameters and BUFFERS (BatchNorm running statistics, ``num_batches_tracked``):
        a ``--resume`` on rank 0 alone, or any per-rank difference at construction, would otherwise diverge
        silently (:func:`for_model` passes them)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        params = list(params)
        if params and isinstance(params[0], (tuple, list)):
            sets = [(name, [p for p in ps if p.requires_grad]) for name, ps in params]
        else:
            sets = [("all", [p for p in params if p.requires_grad])]
        self.sets = [(name, ps) for name, ps in sets if ps]
        self.params = [p for _, ps in self.sets for p in ps]
        self._use_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.buckets = []          # (flat buffer, [(param, offset, numel)])
        self.bucket_set = []       # set name of each bucket
        for name, ps in self.sets:
            cur, cur_bytes = [], 0
            for p in reversed(ps):
                cur.append(p)
                cur_bytes += p.numel() * p.element_size()
                if cur_bytes >= bucket_bytes:
                    self._close(cur, name)
                    cur, cur_bytes = [], 0
            if cur:
                self._close(cur, name)
        self._slot = {}
        for b, (_, entries) in enumerate(self.buckets):
            for p, off, n in entries:
                self._slot[p] = (b, off, n)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._active = False
        #: what the last begin()..finish() reduced: [(set name, bucket index, bytes, where)] with
        #: where = "hook" (launched from backward) or "finish" (a partially filled bucket)
        self.last_reduced = []
        #: exposed communication: time spent in finish() waiting for exchanges launched from the hooks (device time
        #: from HIP events on the compute stream when the buckets live on the GPU, host time otherwise)
        self.time_waits = False
        self._wait_events = []
        self._wait_ms_folded = 0.0       # elapsed time of event pairs already folded away (bounded memory in long runs)
        self._wait_host_s = 0.0
        self._finished_steps = 0
        if self.world > 1 and broadcast_from is not None:
            seen, todo = set(), []
            for t in list(self.params) + [p for _, ps in sets for p in ps] + list(state or []):
                t = t.data if isinstance(t, torch.nn.Parameter) else t
                if id(t) not in seen and t.numel():
                    seen.add(id(t))
                    todo.append(t)
            broadcast_coalesced(todo, broadcast_from, group)

    def _close(self, plist, name):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        entries, off = [], 0
        for p in plist:
            entries.append((p, off, p.numel()))
            off += p.numel()
        self.buckets.append((flat, entries))
        self.bucket_set.append(name)

    # -- per step ------------------------------------------------------------------------
    def begin(self):
        """Call after zero_grad, before backward."""
        self._pending = [len(e) for _, e in self.buckets]
        self._ready = [0] * len(self.buckets)
        self._seen = set()
        self._moved = set()
        self._works = []
        self.last_reduced = []
        self._active = True

    def _view(self, p):
        # the bucket slice takes the PARAMETER's memory order (channels_last weights stay
        # channels_last), so the gradient keeps the strides fused optimizer kernels expect
        b, off, n = self._slot[p]
        flat = self.buckets[b][0]
        dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)) \
            or (p.dim() == 5 and p.is_contiguous(memory_format=torch.channels_last_3d))
        return flat[off:off + n].as_strided(p.size(), p.stride()) if dense else flat[off:off + n].view_as(p)

    def _gather(self, b):
        """Move the gradients produced so far into bucket ``b`` with ONE multi-tensor copy (a copy
        per parameter from its hook cost ~75 tiny launches per step) and re-point ``p.grad``."""
        todo = [p for p, _, _ in self.buckets[b][1] if p in self._seen and p not in self._moved]
        if not todo:
            return
        ops.join_wgrad_stream()      # weight gradients launched on the side stream (ops.WGRAD_STREAM) are read from here on
        views = [self._view(p) for p in todo]
        torch._foreach_copy_(views, [p.grad for p in todo])
        for p, v in zip(todo, views):
            p.grad = v
            self._moved.add(p)

    def _on_grad(self, p):
        if not self._active or p in self._seen:
            return
        self._seen.add(p)
        b = self._slot[p][0]
        self._ready[b] += 1
        if self._ready[b] == self._pending[b]:
            self._gather(b)
            self._launch(b, "hook")

    # ops.WGRAD_STREAM: this hook only counts; the gradient is read in _gather, behind ops.join_wgrad_stream()
    _on_grad._dmc_defers_read = True

    def _launch(self, b, where):
        flat = self.buckets[b][0]
        self.last_reduced.append((self.bucket_set[b], b, flat.numel() * flat.element_size(), where))
        self._ready[b] = -1        # launched
        if self.world == 1:
            return
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        self._works.append((b, dist.all_reduce(flat, op=op, group=self.group, async_op=True)))

    def finish(self):
        """Call after backward, before the optimizers step: waits for the exchanges launched from
        the hooks.  Buckets none of whose members produced a gradient are not touched (their
        parameters keep ``grad=None`` and Adam skips them, as in the reference); a bucket only
        SOME of whose members produced one -- not the case in any shipped recipe -- is completed
        with zeros and reduced here."""
        self._active = False
        for b, (flat, entries) in enumerate(self.buckets):
            if self._ready[b] > 0:              # some but not all members produced a gradient
                self._gather(b)
                for p, off, n in entries:
                    if p not in self._seen:
                        flat[off:off + n].zero_()
                self._launch(b, "finish")
        timed = self.time_waits and bool(self._works)
        on_gpu = timed and self.buckets[0][0].is_cuda
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t0 = time.perf_counter()
        for b, work in self._works:
            work.wait()
            if not self._use_avg:
                self.buckets[b][0].div_(self.world)
        if on_gpu:
            e1.record()
            self._wait_events.append((e0, e1))
            if len(self._wait_events) >= 256:
                # fold the pairs that have completed into a running sum: a long run that never calls comm_summary() keeps
                # at most a few hundred events (query() does not block; the newest pairs may still be in flight)
                keep = []
                for a, b in self._wait_events:
                    if b.query():
                        self._wait_ms_folded += a.elapsed_time(b)
                    else:
                        keep.append((a, b))
                self._wait_events = keep
        if timed:
            self._wait_host_s += time.perf_counter() - t0
        if self.time_waits:
            self._finished_steps += 1
        self._works = []

    def comm_summary(self, reset=True):
        """What a multi-GPU run should look at first: the communicator this reducer really uses (backend, world size),
        the bucket plan, what the LAST step exchanged per parameter set, and the exposed communication per step (the wait
        in finish(); everything else overlapped the backward pass) accumulated since ``time_waits`` was set."""
        exposed_ms = None
        steps = self._finished_steps
        if self._wait_events or self._wait_ms_folded:
            if self._wait_events:
                self._wait_events[-1][1].synchronize()       # the newest pair's end: every earlier pair has completed too
            exposed_ms = self._wait_ms_folded + sum(a.elapsed_time(b) for a, b in self._wait_events)
        out = {
            "backend": dist.get_backend(self.group) if dist.is_initialized() else None,
            "world_size": self.world,
            "buckets": [{"set": name, "bytes": flat.numel() * flat.element_size()} for (flat, _), name in zip(self.buckets, self.bucket_set)],
            "reduce_op": "avg" if self._use_avg else "sum, then / world",
            "last_step_bytes_by_set": self.reduced_bytes(by_set=True),
            "last_step_launched_from": sorted(set(w for _, _, _, w in self.last_reduced)),
            "timed_steps": steps,
            "exposed_wait_ms_per_step": None if (exposed_ms is None or not steps) else round(exposed_ms / steps, 4),
            "exposed_wait_host_ms_per_step": round(1e3 * self._wait_host_s / steps, 4) if steps else None,
        }
        if reset:
            self._wait_events, self._wait_host_s, self._finished_steps, self._wait_ms_folded = [], 0.0, 0, 0.0
        return out

    def reduced_bytes(self, by_set=False):
        """Bytes all-reduced by the last step (optionally per parameter set)."""
        if not by_set:
            return sum(n for _, _, n, _ in self.last_reduced)
        out = {}
        for name, _, n, _ in self.last_reduced:
            out[name] = out.get(name, 0) + n
        return out

    def remove(self):
        for h in self._handles:
            h.remove()


def broadcast_coalesced(tensors, src=0, group=None, chunk_bytes=64 << 20):
    """Broadcast ``tensors`` from ``src`` in place with one collective per (dtype, <= chunk_bytes) flat buffer instead of
    one per tensor (a ResNet-18 + generator + discriminator has ~250 parameters and buffers)."""
    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for (_dtype, _dev), ts in by_type.items():
        i = 0
        while i < len(ts):
            j, size = i, 0
            while j < len(ts) and (j == i or size + ts[j].numel() * ts[j].element_size() <= chunk_bytes):
                size += ts[j].numel() * ts[j].element_size()
                j += 1
            part = ts[i:j]
            flat = torch.cat([t.contiguous().reshape(-1) for t in part])        # logical (row-major) order
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in part:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))      # copy_ keeps t's own strides (channels_last stays channels_last)
                off += n
            i = j


def for_model(model, bucket_bytes=16 << 20, group=None, broadcast_from=0):
    """The reducer every driver uses: one bucket set per optimizer of the reference; every parameter (trainable or
    frozen) and every buffer of ``model`` starts out as rank ``broadcast_from``'s."""
    state = [p for p in model.parameters()] + [b for b in model.buffers()]
    return GradBucketReducer(param_sets_of(model), bucket_bytes=bucket_bytes, group=group,
                             broadcast_from=broadcast_from, state=state)
