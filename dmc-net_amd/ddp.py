"""Data-parallel gradient exchange: one process per GPU, one RCCL all-reduce per gradient bucket,
launched from autograd hooks so that it overlaps the rest of the backward pass.

Replaces ``torch.nn.DataParallel(model, device_ids=args.gpus)`` (code/dmcnet/train.py:117,
code/dmcnet_GAN/train.py:118): no per-iteration parameter broadcast, no activation gather -- each
rank computes its own mean loss over its shard of the batch and the averaged gradients equal the
gathered-batch gradients when shards are equal.  BatchNorm statistics stay per rank, as they are
per replica under DataParallel.

xGMI is point to point, so a ring all-reduce is bound by one link (~153 GB/s): the 44.8 MB of
ResNet-18 gradients cost ~0.5 ms; a few large buckets in reverse execution order are enough.
"""
import torch
import torch.distributed as dist


class GradBucketReducer(object):
    def __init__(self, params, bucket_bytes=16 << 20, group=None, broadcast_from=0):
        """``params``: parameters in forward order (``model.parameters()``); buckets are filled in
        reverse, the order in which backward produces gradients."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self._use_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.buckets = []          # (flat buffer, [(param, offset, numel)])
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._slot = {}
        for b, (_, entries) in enumerate(self.buckets):
            for p, off, n in entries:
                self._slot[p] = (b, off, n)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._active = False
        if self.world > 1 and broadcast_from is not None:
            for p in self.params:
                dist.broadcast(p.data, src=broadcast_from, group=group)

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        entries, off = [], 0
        for p in plist:
            entries.append((p, off, p.numel()))
            off += p.numel()
        self.buckets.append((flat, entries))

    # -- per step ------------------------------------------------------------------------
    def begin(self):
        """Call after zero_grad, before backward."""
        self._pending = [len(e) for _, e in self.buckets]
        self._ready = [0] * len(self.buckets)
        self._seen = set()
        self._moved = set()
        self._works = []
        self._active = True

    def _view(self, p):
        # the bucket slice takes the PARAMETER's memory order (channels_last weights stay
        # channels_last), so the gradient keeps the strides fused optimizer kernels expect
        b, off, n = self._slot[p]
        flat = self.buckets[b][0]
        dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
        return flat[off:off + n].as_strided(p.size(), p.stride()) if dense else flat[off:off + n].view_as(p)

    def _gather(self, b):
        """Move the gradients produced so far into bucket ``b`` with ONE multi-tensor copy (a copy
        per parameter from its hook cost ~75 tiny launches per step) and re-point ``p.grad``."""
        todo = [p for p, _, _ in self.buckets[b][1] if p in self._seen and p not in self._moved]
        if not todo:
            return
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
            self._launch(b)

    def _launch(self, b):
        if self.world == 1:
            return
        flat = self.buckets[b][0]
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        self._works.append((b, dist.all_reduce(flat, op=op, group=self.group, async_op=True)))
        self._ready[b] = -1        # launched

    def finish(self):
        """Call after backward: reduces partially filled buckets, waits, rescales."""
        self._active = False
        for b, (flat, entries) in enumerate(self.buckets):
            if self._ready[b] > 0:              # some but not all members produced a gradient
                self._gather(b)
                for p, off, n in entries:
                    if p not in self._seen:
                        flat[off:off + n].zero_()
                self._launch(b)
        for b, work in self._works:
            work.wait()
            if not self._use_avg:
                self.buckets[b][0].div_(self.world)
        self._works = []

    def remove(self):
        for h in self._handles:
            h.remove()
