#!/usr/bin/env python3
"""One test.py-style evaluation batch (4 videos x 25 segments x 224x224, eval mode, no_grad) through
evaluate.forward_video -- run under `rocprofv3 --kernel-trace --stats` to list the kernels of the forward-only path
(profiles/r3_eval_kernel_stats.csv: no MIOpen kernel in it)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dmcnet_amd
from dmcnet_amd import dataset, evaluate

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = dmcnet_amd.Model(51, 25, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1, arch_estimator="DenseNetTiny").to(dev).eval()
flow, mv, res, _ = dataset.synthetic_batch_on_device(7, 4, 25, 51, dev, flow_ds_factor=0)
for _ in range(3):
    s = evaluate.forward_video(m, mv, res, 25, 1)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    s = evaluate.forward_video(m, mv, res, 25, 1)
b.record(); torch.cuda.synchronize()
print("eval forward of 100 frames: %.3f ms per batch, scores %s" % (a.elapsed_time(b) / 10, tuple(s.shape)))
