import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import dmcnet_amd
from dmcnet_amd import i3d, i3d_train, ops, train
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = i3d.I3D(51, modality="flow+mp4", dropout_prob=0.85, arch_estimator="DenseNetTiny", arch_d="Discriminator").to(dev).train()
net.trunk_dtype = torch.bfloat16
trainer = i3d_train.recipe_trainer(net, batch_size=3, world_size=1, iter_size=1)
data = torch.randn((3, 7, 64, 224, 224), device=dev); target = torch.randint(0, 51, (3,), device=dev)
T = {"bwd": 0.0, "fwd": 0.0, "n": 0}
orig_bwd = torch.Tensor.backward
def timed_bwd(self, *a, **k):
    t = time.perf_counter(); r = orig_bwd(self, *a, **k); T["bwd"] += time.perf_counter() - t; return r
torch.Tensor.backward = timed_bwd
orig_losses = i3d_train.i3d_losses
def timed_losses(*a, **k):
    t = time.perf_counter(); r = orig_losses(*a, **k); T["fwd"] += time.perf_counter() - t; return r
i3d_train.i3d_losses = timed_losses
cnt = [0]
def one():
    trainer.step(data, target, 0, cnt[0]); cnt[0] += 1
for _ in range(8): one()
train.settle_host()
torch.cuda.synchronize()
T["bwd"] = T["fwd"] = 0.0
t0 = time.perf_counter()
for _ in range(20): one()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("per step: total %.2f ms, host enqueue %.2f ms: forward %.2f, backward %.2f, rest (optimizers, schedulers) %.2f" % (
    tot / 20 * 1e3, host / 20 * 1e3, T["fwd"] / 20 * 1e3, T["bwd"] / 20 * 1e3, (host - T["fwd"] - T["bwd"]) / 20 * 1e3))
