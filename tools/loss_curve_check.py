#!/usr/bin/env python3
"""Training-equivalence check of the classifier convolution paths: the same seeded model and the same fixed synthetic
batches trained for a few dozen steps with (a) PyTorch-ROCm / MIOpen fp32 convolutions, (b) this package's fp32-MFMA
convolutions, (c) the bf16x3 default -- prints the loss trajectories side by side and their largest relative gap.
    python tools/loss_curve_check.py [steps] [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from dmcnet_amd import dataset, resnet, train

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
HP = dict(lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
lib = dmcnet_amd._lib.load()
curves = {}
for mode in ("miopen", "own_f32", "own_x3"):
    resnet.OWN_CONV = mode != "miopen"
    lib.dmc_set_option(b"conv_arith", int(mode == "own_x3"))
    torch.manual_seed(0)
    model = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1,
                             arch_estimator="DenseNetTiny").to(dev).train()
    stepper = train.DmcnetTrainStep(model, 3, 1.0, 10.0, **HP)
    batches = [dataset.synthetic_batch_on_device(100 + i, B, 3, 51, dev, flow_ds_factor=16) for i in range(4)]
    out = []
    for i in range(steps):
        r = stepper.step(batches[i % 4])
        out.append((float(r["loss"]), float(r["loss_cls"]), float(r["loss_mse"])))
    curves[mode] = out
print("%4s | %-28s | %-28s | %-28s" % ("step", "miopen  loss / cls / mse", "own fp32-MFMA", "own bf16x3 (default)"))
worst = {"own_f32": 0.0, "own_x3": 0.0}
for i in range(steps):
    a = curves["miopen"][i]
    for m in worst:
        worst[m] = max(worst[m], abs(curves[m][i][0] - a[0]) / abs(a[0]))
    if i < 8 or i % 4 == 3:
        print("%4d | %s | %s | %s" % (i, " ".join("%8.5f" % v for v in a), " ".join("%8.5f" % v for v in curves["own_f32"][i]),
                                     " ".join("%8.5f" % v for v in curves["own_x3"][i])))
print("largest relative gap of the total loss to the MIOpen run over %d steps: fp32-MFMA %.2e, bf16x3 %.2e" % (steps, worst["own_f32"], worst["own_x3"]))
