#!/usr/bin/env python3
"""Where the HOST time of a training step goes: cProfile over a few steps of bench.py's config-2 step, each started on an
empty launch queue (synchronize() in front, outside the profile), top functions by own time.
    python tools/host_profile.py [steps] [batch] [gan]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd                                          # noqa: E402
from dmcnet_amd import dataset, train                      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 40
gan = "gan" in sys.argv[3:]
dev = torch.device("cuda", 0)
HP = dict(lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
torch.manual_seed(0)
model = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1, arch_estimator="DenseNetTiny",
                         arch_d="Discriminator3" if gan else None).to(dev).train()
stepper = train.GanTrainStep(model, 3, 1.0, 1.0, 0.01, 10.0, lr_d_mult=1.0, **HP) if gan else train.DmcnetTrainStep(model, 3, 1.0, 10.0, **HP)
batch = dataset.synthetic_batch_on_device(1234, B, 3, 51, dev, flow_ds_factor=0 if gan else 16)
one = (lambda i: stepper.step(batch, i)) if gan else (lambda i: stepper.step(batch))
for i in range(6):
    one(i)
train.settle_host()
clean = []
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one(i)
    clean.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("clean host ms per step (empty queue): median %.3f  min %.3f  max %.3f" % (sorted(clean)[len(clean) // 2], min(clean), max(clean)))
pr = cProfile.Profile()
for i in range(steps):
    torch.cuda.synchronize()
    pr.enable()
    one(i)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
