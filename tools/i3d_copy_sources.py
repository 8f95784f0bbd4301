"""Which Python call sites issue the torch-native copy / add kernels of an eager I3D micro-step (torch.profiler with stacks).
    python tools/i3d_copy_sources.py"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
from dmcnet_amd import i3d, i3d_train
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = i3d.I3D(101, modality="flow+mp4", dropout_prob=0.85, arch_estimator="DenseNetTiny", arch_d="Discriminator").to(dev).train()
net.trunk_dtype = torch.bfloat16
trainer = i3d_train.recipe_trainer(net, batch_size=3, world_size=1, iter_size=1)
data = torch.randn((3, 7, 64, 224, 224), device=dev)
target = torch.randint(0, 101, (3,), device=dev)
for k in range(4):
    trainer.step(data, target, 0, k)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    trainer.step(data, target, 0, 4)
    trainer.step(data, target, 0, 5)
    torch.cuda.synchronize()
agg = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::cat") and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if "dmc-net_amd" in s or "dmcnet_amd" in s or "bench" in s]
        key = (e.name, str(e.input_shapes)[:70], st[0][-70:] if st else "(autograd engine / no package frame)")
        agg[key] += 1; tim[key] += e.device_time_total
for key, t in tim.most_common(14):
    print("%8.1f us  x%3d  %s" % (t / 2, agg[key] // 2 if agg[key] > 1 else agg[key], key))
