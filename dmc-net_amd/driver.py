"""Epoch-level driver: what ``main()`` / ``train()`` / ``validate()`` of the reference do around the
training step (code/dmcnet/train.py:31-201, :205-288, :292-369; GAN: code/dmcnet_GAN/train.py).

* ``train_epoch``  -- iterate a loader of (input_flow, input_mv, input_residual, target), run the
  HIP-backed step, keep the reference's meters (loss, loss_cls, loss_mse, Prec@1/5), optional
  freeze of the classifier (epochs below ``epoch_thre``);
* ``validate``     -- eval mode, same losses / meters, returns Prec@1 (``top1.avg``);
* ``fit``          -- epochs with the reference's LR policy, eval every ``eval_freq`` epochs and on
  the last, checkpoint when best or every 40 epochs, in the reference's checkpoint layout.

Meters accumulate device tensors; the host reads them once per ``print_freq`` iterations (the
reference syncs 5-7 times per iteration).
"""
import time

import torch

from . import ops, train

SAVE_FREQ = 40
PRINT_FREQ = 20


def _to(batch, device):
    return tuple(t.to(device, non_blocking=True) for t in batch)


def train_epoch(loader, stepper, epoch, device="cuda:0", freeze=False, print_freq=PRINT_FREQ,
                log=print, prep=None):
    """One epoch of ``train()``.  ``stepper`` is a DmcnetTrainStep or a GanTrainStep.  ``prep`` (a
    ``dataset.DevicePrep``) turns the loader's raw uint8 batches (``dataset.RawView`` +
    ``dataset.collate_raw``) into the float batch on the GPU instead of in the loader workers."""
    meters = {k: train.AverageMeter() for k in ("loss", "loss_cls", "loss_mse", "top1", "top5")}
    gan = isinstance(stepper, train.GanTrainStep)
    stepper.model.train()
    t0 = time.time()
    for i, batch in enumerate(loader):
        batch = prep(batch) if prep is not None else _to(batch, device)
        out = stepper.step(batch, i) if gan else stepper.step(batch, freeze=freeze)
        if i == 1:
            train.settle_host()      # both step kinds have run once: park the long-lived objects outside the collector
        n = batch[0].shape[0] * stepper.num_segments
        prec1, prec5 = train.accuracy(out["output"], batch[3], topk=(1, 5))
        meters["loss"].update(out["loss"], n)
        meters["loss_cls"].update(out["loss_cls"], n)
        if "loss_mse" in out:
            meters["loss_mse"].update(out["loss_mse"], n)
        meters["top1"].update(prec1, n)
        meters["top5"].update(prec5, n)
        if log is not None and i % print_freq == 0:
            log("Epoch: [%d][%d/%d]\tTime %.3f\tLoss %.4f (%.4f)\tLoss_cls %.4f\tloss_mse %.4f\t"
                "Prec@1 %.3f (%.3f)\tPrec@5 %.3f" % (
                    epoch, i, len(loader), time.time() - t0, float(meters["loss"].val),
                    float(meters["loss"].avg), float(meters["loss_cls"].avg),
                    float(meters["loss_mse"].avg), float(meters["top1"].val),
                    float(meters["top1"].avg), float(meters["top5"].avg)))
    return {k: float(m.avg) for k, m in meters.items()}


@torch.no_grad()
def validate(loader, model, num_segments, lr_cls, lr_mse, device="cuda:0", log=print, loss_mse="MSELoss",
             prep=None):
    """``validate()``: eval mode, CE on the segment consensus + flow MSE, Prec@1/5; returns the
    dict of averages (``top1`` is what the reference returns)."""
    meters = {k: train.AverageMeter() for k in ("loss", "loss_cls", "loss_mse", "top1", "top5")}
    model.eval()
    for batch in loader:
        input_flow, input_mv, input_residual, target = prep(batch) if prep is not None else _to(batch, device)
        flow = input_flow.reshape((-1,) + tuple(input_mv.shape[-3:]))
        out = model(input_mv, input_residual)
        att = getattr(model, "att", 0) == 1
        output, gen_flow = out[0], (out[-2] if att else out[-1])
        loss_cls, consensus = ops.consensus_ce(output, target, num_segments)
        # att == 1: criterion_mse(att_flow * gen_flow, att_flow * input_flow), code/dmcnet/train.py:332-335
        loss_mse_v = train.flow_loss(loss_mse, gen_flow, flow, out[-1] if att else None)
        n = flow.shape[0]
        prec1, prec5 = train.accuracy(consensus, target, topk=(1, 5))
        meters["loss"].update(loss_cls * lr_cls + loss_mse_v * lr_mse, n)
        meters["loss_cls"].update(loss_cls, n)
        meters["loss_mse"].update(loss_mse_v, n)
        meters["top1"].update(prec1, n)
        meters["top5"].update(prec5, n)
    res = {k: float(m.avg) for k, m in meters.items()}
    if log is not None:
        log("Testing Results: Prec@1 %.3f Prec@5 %.3f Loss %.5f" % (res["top1"], res["top5"], res["loss"]))
    return res


def fit(model, stepper, train_loader, val_loader, epochs, lr, weight_decay, lr_steps, lr_decay=0.1,
        epoch_thre=0, eval_freq=5, lr_cls=1.0, lr_mse=10.0, arch="resnet18", model_prefix="model",
        representation="mv", device="cuda:0", start_epoch=0, best_prec1=0.0, log=print, save=True,
        miopen_find=True, prep=None):
    """The epoch loop of ``main()`` (code/dmcnet/train.py:175-201).  ``miopen_find`` mirrors the
    reference's ``cudnn.benchmark = True`` (:118), answered from the shipped MIOpen find-db."""
    if miopen_find and str(device).startswith("cuda"):
        from . import miopen
        miopen.enable_find()
    gan = isinstance(stepper, train.GanTrainStep)
    history = []
    for epoch in range(start_epoch, epochs):
        train.adjust_learning_rate(stepper.optimizer_cls, epoch, lr_steps, lr_decay, lr, weight_decay,
                                   freeze=not gan, epoch_thre=epoch_thre)
        train.adjust_learning_rate(stepper.optimizer_gf, epoch, lr_steps, lr_decay, lr, weight_decay)
        if gan:
            train.adjust_learning_rate(stepper.optimizer_d, epoch, lr_steps, lr_decay, lr, weight_decay)
        tr = train_epoch(train_loader, stepper, epoch, device, freeze=(epoch < epoch_thre and not gan),
                         log=log, prep=prep)
        entry = {"epoch": epoch, "train": tr}
        if epoch % eval_freq == 0 or epoch == epochs - 1:
            va = validate(val_loader, model, stepper.num_segments, lr_cls, lr_mse, device, log=log,
                          loss_mse=getattr(stepper, "loss_mse", "MSELoss"), prep=prep)
            entry["val"] = va
            is_best = va["top1"] > best_prec1
            best_prec1 = max(va["top1"], best_prec1)
            if save and (is_best or epoch % SAVE_FREQ == 0):
                state = {"epoch": epoch + 1, "arch": arch, "state_dict": train.reference_state_dict(model),
                         "best_prec1": best_prec1, "optimizer_cls": stepper.optimizer_cls.state_dict(),
                         "optimizer_gf": stepper.optimizer_gf.state_dict()}
                if gan:
                    state["optimizer_d"] = stepper.optimizer_d.state_dict()
                train.save_checkpoint(state, is_best, model_prefix, representation)
        history.append(entry)
    return history, best_prec1
