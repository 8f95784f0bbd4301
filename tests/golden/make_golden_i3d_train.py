#!/usr/bin/env python3
"""G11: TRAINING-mode loss assembly of the I3D variant, produced by the REFERENCE's own classes run in this container:
``static_model.forward`` of code/dmcnet_I3D/train/model.py:135-188 (imported; the `[:5]` / `[5:7]` channel slicing, the T
axis folded into the batch for the discriminator, fake-then-real order and targets) around the reference's own
``network/i3d.py`` I3D (imported as it is) with the DenseNetTiny generator and the `Discriminator` node, on a seeded
1 x 7 x 16 x 224 x 224 clip, dropout_prob 0, BatchNorm in training mode.  The discriminator's Dropout2d layers use seeded keep
masks (forward hooks, as G3 does) so that the run is reproducible.

Stored: the three losses, the logits, the generated cue's checksum / a slice, gradients of (loss + mse + loss_adv) for named
parameters (whole when <= 8192 values, else every stride-th value + a checksum) of the generator, the trunk, the classifier head and the discriminator, and the stem BatchNorm's running
statistics after the forward -- for `detach` False and True -- and the same gradients from an fp64 evaluation of the same
graph (see the comment in main()).  ``.cuda()`` is a no-op here (no GPU in the build container).

Run in the build container: ``python tests/golden/make_golden_i3d_train.py``."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code/dmcnet_I3D"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "network"))          # i3d.py does `import initializer`
from oracle import dmc_oracle as O                                   # noqa: E402
from tests.golden.make_golden import checksum, hook_dropout, npy     # noqa: E402

torch.set_num_threads(8)
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
torch.distributed._initialized = False                               # pre-1.0 attribute read at import time

spec = importlib.util.spec_from_file_location("ref_i3d", os.path.join(REF, "network", "i3d.py"))
ref_i3d = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_i3d)
from train import model as ref_model                                 # noqa: E402

CFG = dict(seed_net=84, seed_data=85, seed_masks=86, frames=16, label=3, num_classes=51)
GRADS = ("gen_flow_model.conv_0.0.weight", "gen_flow_model.conv_3.0.bias", "gen_flow_model.predict_flow.weight",
         "conv3d_1a_7x7.conv3d.weight", "conv3d_2c_3x3.batch3d.weight", "mixed_4d.branch_1.1.conv3d.weight",
         "mixed_5c.branch_3.1.conv3d.weight", "conv3d_0c_1x1.conv3d.bias", "classifier.weight",
         "discriminator.discriminator_block_1.0.weight", "discriminator.discriminator_block_3.3.weight",
         "discriminator.adv_layer.weight")


def sample_stride(numel):
    return max(1, numel // 4096) | 1          # odd: walks through every tap / channel phase


def main():
    c = CFG
    out = {"cfg": np.array(repr(c)), "grad_names": np.array(GRADS)}
    for tag, detach in (("nodetach", False), ("detach", True)):
        net = ref_i3d.I3D(c["num_classes"], modality="flow+mp4", dropout_prob=0, arch_estimator="DenseNetTiny",
                          arch_d="Discriminator")
        O.seeded_state_fill(net, seed=c["seed_net"]).train()
        wrapper = ref_model.static_model(net=net, criterion=torch.nn.CrossEntropyLoss(), criterion2=torch.nn.MSELoss(),
                                         criterion3=torch.nn.CrossEntropyLoss())
        data = torch.from_numpy(np.random.RandomState(c["seed_data"]).standard_normal(
            (1, 7, c["frames"], 224, 224)).astype(np.float32))
        target = torch.tensor([c["label"]])
        masks = O.seeded_dropout_masks(c["seed_masks"], net.discriminator, 2 * c["frames"])
        hooks = hook_dropout(net.discriminator, masks)
        flows = []
        h2 = net.gen_flow_model.register_forward_hook(lambda m, i, o: flows.append(o.detach()))
        (logits,), (loss, mse, loss_adv) = wrapper.forward(data, target, node="flow+logit", detach=detach, stage=1)
        (loss + mse + loss_adv).backward()
        for h in hooks + [h2]:
            h.remove()
        params = dict(net.named_parameters())
        out[tag + "_logits"] = npy(logits)
        out[tag + "_losses"] = np.array([float(loss), float(mse), float(loss_adv)], dtype=np.float64)
        out[tag + "_flow_checksum"] = checksum(flows[0])
        out[tag + "_flow_slice"] = npy(flows[0][5, :, 100:104, 50:66])
        for k in GRADS:
            g = params[k].grad
            assert g is not None, k
            if g.numel() <= 8192:
                out[tag + "_grad_" + k] = npy(g)
            else:                    # big tensors: every stride-th value (about 4096 of them) + checksum
                out[tag + "_gradsample_" + k] = npy(g.reshape(-1)[::sample_stride(g.numel())])
                out[tag + "_gradsum_" + k] = checksum(g)
        sd = net.state_dict()
        out[tag + "_stem_running_mean"] = npy(sd["conv3d_1a_7x7.batch3d.running_mean"])
        out[tag + "_stem_running_var"] = npy(sd["conv3d_1a_7x7.batch3d.running_var"])
        # The same graph in fp64 (the reference's own network classes cast to double; the loss assembly is the oracle's
        # restatement, bit-identical to static_model.forward in fp32 -- static_model.forward itself forces .float()):
        # the trunk's training-mode BatchNorm chain at batch 1 is ill-conditioned, the fp32 reference run above sits 2-3 %
        # from this on the trunk / generator gradients, so device runs are judged by their distance to THIS, relative to
        # the reference's own distance.
        net64 = ref_i3d.I3D(c["num_classes"], modality="flow+mp4", dropout_prob=0, arch_estimator="DenseNetTiny",
                            arch_d="Discriminator")
        O.seeded_state_fill(net64, seed=c["seed_net"]).train().double()
        hooks = hook_dropout(net64.discriminator, {k: v.double() for k, v in masks.items()})
        logits64, losses64 = O.i3d_losses(net64, data.double(), target, stage=1, detach=detach)
        sum(losses64).backward()
        for h in hooks:
            h.remove()
        p64 = dict(net64.named_parameters())
        out[tag + "_losses64"] = np.array([float(l) for l in losses64], dtype=np.float64)
        out[tag + "_logits64"] = npy(logits64)
        for k in GRADS:
            g = p64[k].grad
            out[tag + "_grad64_" + k] = npy(g) if g.numel() <= 8192 else npy(g.reshape(-1)[::sample_stride(g.numel())])
        print(tag, [float(loss), float(mse), float(loss_adv)], float(logits.abs().max()),
              {k: (None if params[k].grad is None else float(params[k].grad.abs().max())) for k in GRADS[:4]})
    out["keys"] = np.array(list(net.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "g11_i3d_train.npz"), **out)
    print("G11 done")


if __name__ == "__main__":
    main()
