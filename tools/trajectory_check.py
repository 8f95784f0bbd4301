#!/usr/bin/env python3
"""Trajectory parity -- the "matched accuracy" evidence available offline (no datasets, no ImageNet weights): the HIP path
and the CPU oracle (the reference's train loop restated, pinned bit-exact by tests/golden G4) start from ONE seeded state
and step over the SAME seeded batches (and, for the GAN variant, the same forced Dropout2d masks); printed per step: both
loss triples, their relative gaps, and the cosine of the consensus logits.

    python tools/trajectory_check.py [dmcnet|gan] [steps] [batch] [fp64]

Reference lines: code/dmcnet/train.py:221-266, code/dmcnet_GAN/train.py:261-371; published accuracies README.md:73-76
(not reproducible offline).  What "parity" can mean over many steps: the two runs compute the same function with different
fp32 summation orders (and bf16x3 convolutions); Adam(eps = 1e-3) divides rounding noise of tiny gradients by ~eps, so the
weights drift apart chaotically while the LOSSES and LOGITS stay together -- the envelope this prints is what
tests/test_trajectory_gpu.py bounds."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dmcnet_amd                                    # noqa: E402
from dmcnet_amd import train as T                    # noqa: E402
from oracle import dmc_oracle as O                   # noqa: E402  (a checker: this tool is test infrastructure)

KW = dict(base_model="resnet18", use_databn=0, gen_flow_or_delta=1, arch_estimator="DenseNetTiny")
HP = dict(lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
DEV = "cuda:0"


def cosine(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))


def rel(a, b):
    a, b = float(a), float(b)
    return abs(a - b) / max(abs(b), 1e-30)


def _oracle(num_class, gan, seed, dtype):
    o = O.OracleModel(num_class, 3, "mv", arch_d="Discriminator3" if gan else None, **KW)
    O.seeded_state_fill(o, seed)
    o = o.to(dtype).train()
    opts = O.make_optimizers(o, lr_d_mult=1.0, **HP) if gan else O.make_optimizers(o, **HP)
    for opt in opts:      # what the reference's main() does before every epoch (code/dmcnet/train.py:398-408): lr * lr_mult per group
        O.adjust_learning_rate(opt, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"])
    return o, opts


def run(kind="dmcnet", steps=20, batch=8, num_class=51, seed=700, log=None, fp64=False, resync=False):
    """``resync``: before every step the device model is given the ORACLE's current weights and buffers, so every step is a
    one-step comparison from an identical state along the trajectory the oracle takes (losses and logits depend on the
    weights only; the optimizer update itself is pinned by the golden G4 steps) -- no chaotic amplification in it.
    Returns a list of per-step dicts: {"ref": {...}, "got": {...}, "rel": {...}, "cos": float}; with ``fp64`` also the
    same trajectory of the oracle in DOUBLE precision ("f64") and the gaps of both fp32 runs to it ("rel64_hip", "rel64_ref"):
    how far two fp32 evaluations of the same recipe drift apart by themselves."""
    gan = kind == "gan"
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
    o, oopts = _oracle(num_class, gan, seed, torch.float32)
    o64, oopts64 = _oracle(num_class, gan, seed, torch.float64) if fp64 else (None, None)
    m = dmcnet_amd.Model(num_class, 3, "mv", arch_d="Discriminator3" if gan else None, **KW)
    m.load_state_dict(o.state_dict())
    m.to(DEV).train()
    if gan:
        stepper = T.GanTrainStep(m, 3, 1.0, 1.0, 0.01, 10.0, lr_d_mult=1.0, **HP)
    else:
        stepper = T.DmcnetTrainStep(m, 3, 1.0, 10.0, **HP)
    out = []
    for i in range(steps):
        b = O.synthetic_batch(seed=seed + 1 + i, batch=batch, num_segments=3, num_class=num_class,
                              flow_ds_factor=0 if gan else 16)
        b64 = tuple(t.double() if t.is_floating_point() else t for t in b)
        if resync and i > 0:
            m.load_state_dict(o.state_dict())
        t0 = time.time()
        r64 = None
        if gan:
            frames = batch * 3 * (2 if i % 2 == 0 else 1)
            masks = O.seeded_dropout_masks(seed + 1000 + i, o.discriminator, frames)
            o.discriminator.forced_masks = masks
            m.discriminator.forced_masks = masks
            ref = O.gan_train_step(o, oopts[0], oopts[1], oopts[2], b, i, 3, 1.0, 1.0, 0.01, 10.0)
            if fp64:
                o64.discriminator.forced_masks = {k: v.double() for k, v in masks.items()}
                r64 = O.gan_train_step(o64, oopts64[0], oopts64[1], oopts64[2], b64, i, 3, 1.0, 1.0, 0.01, 10.0)
            got = stepper.step(tuple(t.to(DEV) for t in b), i)
            keys = ("loss", "loss_cls", "loss_adv") + (("loss_mse",) if i % 2 else ())
        else:
            ref = O.dmcnet_train_step(o, oopts[0], oopts[1], b, 3, 1.0, 10.0)
            if fp64:
                r64 = O.dmcnet_train_step(o64, oopts64[0], oopts64[1], b64, 3, 1.0, 10.0)
            got = stepper.step(tuple(t.to(DEV) for t in b))
            keys = ("loss", "loss_cls", "loss_mse")
        rec = {"ref": {k: float(ref[k]) for k in keys}, "got": {k: float(got[k]) for k in keys},
               "rel": {k: rel(got[k], ref[k]) for k in keys}, "cos": cosine(got["output"], ref["output"]),
               "logit_rel": float((got["output"].cpu().double() - ref["output"].double()).abs().max() /
                                  ref["output"].double().abs().max()),
               "cpu_s": time.time() - t0}
        if gan:
            rec["cos_validity"] = cosine(got["validity"], ref["validity"])
        extra = ""
        if fp64:
            rec["f64"] = {k: float(r64[k]) for k in keys}
            rec["rel64_hip"] = {k: rel(got[k], r64[k]) for k in keys}
            rec["rel64_ref"] = {k: rel(ref[k], r64[k]) for k in keys}
            rec["cos64_hip"], rec["cos64_ref"] = cosine(got["output"], r64["output"]), cosine(ref["output"], r64["output"])
            extra = " | vs fp64: hip %s cos %.7f ; oracle-fp32 %s cos %.7f" % (
                " ".join("%.1e" % rec["rel64_hip"][k] for k in keys), rec["cos64_hip"],
                " ".join("%.1e" % rec["rel64_ref"][k] for k in keys), rec["cos64_ref"])
        out.append(rec)
        if log:
            log("%3d | ref %s | hip %s | rel %s | logits cos %.7f rel %.1e%s%s" % (
                i, " ".join("%9.6f" % rec["ref"][k] for k in keys), " ".join("%9.6f" % rec["got"][k] for k in keys),
                " ".join("%.1e" % rec["rel"][k] for k in keys), rec["cos"], rec["logit_rel"],
                "  validity cos %.7f" % rec["cos_validity"] if gan else "", extra))
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "dmcnet"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    print("# %s: %d steps at B = %d clips x 3 segments, one seeded state, identical batches%s; columns: loss, loss_cls, %s"
          % (kind, steps, batch, " and Dropout2d masks" if kind == "gan" else "",
             "loss_adv [, loss_mse on G steps]" if kind == "gan" else "loss_mse"))
    recs = run(kind, steps, batch, log=print, fp64="fp64" in sys.argv[4:])
    worst = {}
    for r in recs:
        for k, v in r["rel"].items():
            worst[k] = max(worst.get(k, 0.0), v)
    print("# largest relative gaps over %d steps: %s; smallest logit cosine %.7f"
          % (steps, ", ".join("%s %.2e" % kv for kv in sorted(worst.items())), min(r["cos"] for r in recs)))


if __name__ == "__main__":
    main()
