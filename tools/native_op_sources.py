"""torch-native operators (aten::*) of a training step that launch device work, by device time, with shapes: what is NOT on this
package's kernels.  python tools/native_op_sources.py [dmcnet|gan|i3d]"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
import dmcnet_amd
from dmcnet_amd import train, dataset, i3d, i3d_train
cfg = sys.argv[1] if len(sys.argv) > 1 else "dmcnet"
dev = torch.device("cuda:0")
torch.manual_seed(0)
if cfg == "i3d":
    net = i3d.I3D(101, modality="flow+mp4", dropout_prob=0.85, arch_estimator="DenseNetTiny", arch_d="Discriminator").to(dev).train()
    net.trunk_dtype = torch.bfloat16
    trainer = i3d_train.recipe_trainer(net, batch_size=3, world_size=1, iter_size=1)
    data = torch.randn((3, 7, 64, 224, 224), device=dev)
    target = torch.randint(0, 101, (3,), device=dev)
    one = lambda i: trainer.step(data, target, 0, i)
else:
    import bench
    gan = cfg == "gan"
    model = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1, arch_estimator="DenseNetTiny",
                             arch_d="Discriminator3" if gan else None).to(dev).train()
    stepper = (train.GanTrainStep(model, 3, 1.0, 1.0, 0.01, 10.0, lr_d_mult=1.0, **bench.HP) if gan
               else train.DmcnetTrainStep(model, 3, 1.0, 10.0, **bench.HP))
    batch = dataset.synthetic_batch_on_device(1234, 40, 3, 51, dev, flow_ds_factor=0 if gan else 16)
    one = (lambda i: stepper.step(batch, i)) if gan else (lambda i: stepper.step(batch))
for k in range(6):
    one(k)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for k in range(6, 6 + N):
        one(k)
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.self_device_time_total > 0:
        key = (e.name, str(e.input_shapes)[:90])
        cnt[key] += 1; tim[key] += e.self_device_time_total
tot = sum(tim.values())
print("aten:: self device time per step: %.1f us" % (tot / N))
for key, t in tim.most_common(22):
    print("%8.1f us/step  x%5.1f  %s %s" % (t / N, cnt[key] / N, key[0], key[1]))
