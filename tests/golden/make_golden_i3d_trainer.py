#!/usr/bin/env python3
"""G10: the I3D trainer's policy, produced by the REFERENCE's own code run in this container:

* optimizers: the parameter routing + optimizer construction lines of
  code/dmcnet_I3D/train_model.py (68-106, 122-179) are read and executed as they are (only the
  surrounding function is absent: the names they use are supplied);
* schedule: code/dmcnet_I3D/train/lr_scheduler.py ``MultiFactorScheduler`` (imported);
* loop: ``model.fit`` of code/dmcnet_I3D/train/model.py (imported), 3 epochs x 8 micro-batches,
  iter_size 2, epoch_thre 1, adv 1, detach 1 on tests/golden/tiny_i3d.TinyI3D.  ``.cuda()`` is a no-op
  (no GPU here) and ``torch.distributed._initialized`` (removed from torch long ago) reads False.  After every micro-batch the data iterator records every parameter and the learning
  rate of every optimizer group.

Run in the build container: ``python tests/golden/make_golden_i3d_trainer.py``."""
import logging
import os
import sys
import tempfile
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code/dmcnet_I3D"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
from tests.golden import tiny_i3d  # noqa: E402

torch.set_num_threads(1)        # one thread: the CPU kernels' summation order is then reproducible (the test does the same)
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.cuda.is_available = lambda: True            # model.fit asserts it
torch.cuda.is_current_stream_capturing = lambda: False   # (the optimizers ask once is_available says yes)
torch.distributed._initialized = False                   # pre-1.0 attribute read by the checkpoint-path helper

from train import lr_scheduler as ref_sched, metric, model as ref_model  # noqa: E402

CFG = dict(seed_net=91, seed_data=92, epochs=3, per_epoch=8, iter_size=2, epoch_thre=1, adv=1.0, detach=True,
           lr_base=4e-4, lr_base2=2e-4, lr_d=2e-3, lr_factor=0.2, sched_steps=[5, 9])


def reference_optimizers(net_wrapper, lr_base, lr_base2, adv):
    lines = open(os.path.join(REF, "train_model.py")).read().split("\n")
    # 1-based line ranges: parameter routing + lr_mul; weight decay; optimizer construction
    src = lines[67:106] + lines[111:115] + lines[121:179]
    code = textwrap.dedent("\n".join(src))
    ns = dict(net=net_wrapper, modality="flow+mp4", fine_tune=False, optim="adam", lr_base=lr_base,
              lr_base2=lr_base2, adv=adv, net_name="I3D", torch=torch, logging=logging)
    exec(compile(code, "train_model.py[68-179]", "exec"), ns)
    return {k: ns[k] for k in ("optimizer", "optimizer_2", "optimizer_3", "optimizer_mse", "optimizer_mse_2")}


class Recorder(object):
    """train_iter: yields the epoch's batches and snapshots the model / learning rates after each one."""

    def __init__(self, net, opts, data):
        self.net, self.opts, self.data, self.epoch = net, opts, data, 0
        self.params, self.lrs = [], []

    def snap(self):
        self.params.append(tiny_i3d.snapshot(self.net))
        self.lrs.append([g["lr"] for k in sorted(self.opts) for g in self.opts[k].param_groups])

    def __len__(self):
        return len(self.data[0])

    def __iter__(self):
        for i, item in enumerate(self.data[self.epoch]):
            if i:
                self.snap()
            yield item
        self.snap()
        self.epoch += 1


def main():
    c = CFG
    net = tiny_i3d.build(c["seed_net"])
    tmp = tempfile.mkdtemp()
    wrapper = ref_model.model(net=net, criterion=torch.nn.CrossEntropyLoss(), model_prefix=os.path.join(tmp, "g10"),
                              step_callback_freq=50, save_checkpoint_freq=1000, opt_batch_size=1,
                              criterion2=torch.nn.MSELoss(), criterion3=torch.nn.CrossEntropyLoss(), adv=c["adv"])
    opts = reference_optimizers(wrapper, c["lr_base"], c["lr_base2"], c["adv"])
    mk = lambda base: ref_sched.MultiFactorScheduler(base_lr=base, steps=list(c["sched_steps"]), factor=c["lr_factor"],
                                                     step_counter=0)
    metrics = metric.MetricList(metric.Loss(name="loss-ce"), metric.Loss(name="loss-mse"),
                                metric.Accuracy(name="top1", topk=1), metric.Accuracy(name="top5", topk=5))
    metrics_d = metric.MetricList(metric.Loss(name="classi_D"), metric.Loss(name="adv_D"))
    rec = Recorder(wrapper.net, opts, tiny_i3d.batches(c["seed_data"], c["epochs"], c["per_epoch"]))
    init = tiny_i3d.snapshot(wrapper.net)
    wrapper.fit(train_iter=rec, eval_iter=None, optimizer=opts["optimizer"], lr_scheduler=mk(c["lr_base"]),
                metrics=metrics, epoch_start=0, epoch_end=c["epochs"], iter_size=c["iter_size"],
                optimizer_mse=opts["optimizer_mse"], optimizer_2=opts["optimizer_2"], optimizer_3=opts["optimizer_3"],
                optimizer_mse_2=opts["optimizer_mse_2"], lr_scheduler2=mk(c["lr_base2"]), lr_scheduler3=mk(c["lr_d"]),
                metrics_D=metrics_d, epoch_thre=c["epoch_thre"], score_dir=None, detach=c["detach"])
    groups = [(k, len(opts[k].param_groups), [len(g["params"]) for g in opts[k].param_groups],
               [g.get("lr_mult", 1.0) for g in opts[k].param_groups]) for k in sorted(opts)]
    sched = ref_sched.MultiFactorScheduler(base_lr=0.1, steps=[2, 14, 18], factor=0.1, step_counter=2)
    np.savez_compressed(os.path.join(HERE, "g10_i3d_trainer.npz"), init=init, params=np.stack(rec.params),
                        lrs=np.array(rec.lrs, dtype=np.float64),
                        group_layout=np.array(repr(groups)), cfg=np.array(repr(c)),
                        sched_table=np.array([sched.update() for _ in range(130)], dtype=np.float64))
    print("G10 done: %d snapshots of %d parameters; groups %s" % (len(rec.params), init.size, groups))


if __name__ == "__main__":
    main()
