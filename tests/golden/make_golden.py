#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python in this container.

Run once, here (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

What is imported from the reference (nothing is copied into the repo; only inputs' seeds and the
reference's OUTPUTS are stored):

* ``code/dmcnet/model.py`` and ``code/dmcnet_GAN/model.py`` -- imported as-is, with stub modules
  for what this image lacks: ``cv2`` (only referenced by ``transforms.py``), ``torchvision``
  (``models.resnet*`` -> the architecture restated in ``oracle/dmc_oracle.py``;
  ``transforms.Compose``), ``coviar`` and ``skimage.measure`` (only referenced by ``dataset.py``).
* ``code/dmcnet/dataset.py`` -- for the integer index-sampling functions.
* ``code/dmcnet/train.py`` and ``code/dmcnet_GAN/train.py`` -- do not parse on Python >= 3.7
  (``.cuda(dev, async=True)``).  Their source is read, patched IN MEMORY for interpreter/torch
  compatibility only (``async=`` -> ``non_blocking=``; ``x.data[0]`` -> ``x.item()``;
  ``accuracy`` -- meters only -- replaced by the oracle's, returning 1-element tensors as
  torch 0.3 did, because the reference's ``.view(-1)`` on a non-contiguous slice now raises;
  ``Tensor.cuda``/``Module.cuda`` made no-ops because this container has no GPU) and executed, so
  that the reference's own ``train()`` loop and ``adjust_learning_rate()`` produce the post-step
  weights and the LR table.

Weights come from ``seeded_state_fill`` (function of seed, key and shape only) so that a model
with the same state-dict keys can be given identical weights elsewhere without shipping them.
"""
import contextlib
import importlib.util
import io
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import dmc_oracle as O  # noqa: E402

torch.set_num_threads(8)


# ------------------------------------------------------------------ stubs for absent packages
def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.COLOR_BGR2HLS = cv2.COLOR_HLS2BGR = 0
    sys.modules["cv2"] = cv2

    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    for name in O.RESNET_DEPTHS:
        setattr(tv.models, name, (lambda n: (lambda pretrained=False: O.build_resnet(n)))(name))
    tv.transforms = types.ModuleType("torchvision.transforms")

    class Compose(object):
        def __init__(self, ts):
            self.transforms = ts

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    tv.transforms.Compose = Compose
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    sys.modules["torchvision.transforms"] = tv.transforms

    cov = types.ModuleType("coviar")
    cov.get_num_frames = lambda p: 0
    cov.load = lambda *a: None
    sys.modules["coviar"] = cov
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.measure")
    skm.block_reduce = None
    sk.measure = skm
    sys.modules["skimage"] = sk
    sys.modules["skimage.measure"] = skm


def import_ref(variant, module):
    """Import /root/reference/code/<variant>/<module>.py under a private name."""
    d = os.path.join(REF, "code", variant)
    for m in ("transforms", "model", "dataset", "train_options"):
        sys.modules.pop(m, None)
    sys.path.insert(0, d)
    try:
        spec = importlib.util.spec_from_file_location("ref_%s_%s" % (variant, module),
                                                      os.path.join(d, module + ".py"))
        mod = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
    finally:
        sys.path.remove(d)
    return mod


def exec_ref_train(variant, args):
    """Execute the reference's train.py (patched in memory, see module docstring)."""
    d = os.path.join(REF, "code", variant)
    src = open(os.path.join(d, "train.py")).read()
    src = src.replace("async=True", "non_blocking=True").replace(".data[0]", ".item()")
    for m in ("transforms", "model", "dataset", "train_options"):
        sys.modules.pop(m, None)
    sys.path.insert(0, d)
    ns = {"__name__": "ref_train_" + variant}
    argv = sys.argv
    sys.argv = ["train.py"]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            exec(compile(src, os.path.join(d, "train.py"), "exec"), ns)
    finally:
        sys.path.remove(d)
        sys.argv = argv
    ns["args"] = args
    # the reference's accuracy() (train.py:411-424) calls .view(-1) on a non-contiguous slice,
    # which modern torch rejects; it only feeds the printed meters, so the oracle's restatement
    # stands in for it (results as 1-element tensors, as torch 0.3 returned them).
    ns["accuracy"] = lambda output, target, topk=(1,): [
        x.reshape(1) for x in O.accuracy(output, target, topk)]
    return ns


@contextlib.contextmanager
def cuda_is_a_noop():
    t_cuda, m_cuda = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = t_cuda, m_cuda


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npy(t):
    return t.detach().cpu().numpy().copy()


def checksum(t):
    """Position-weighted checksum + plain moments of a big tensor (float64)."""
    x = t.detach().double().reshape(-1)
    w = torch.cos(torch.arange(x.numel(), dtype=torch.float64) * 0.37) + 1.5
    return np.array([float(x.sum()), float((x * x).sum()), float((x * w).sum()),
                     float(x.abs().max())])


# ------------------------------------------------------------------ dropout determinism
def dropout_masks(seed, disc, n):
    """Keep-masks (already divided by 0.75) per discriminator block, function of (seed, name)."""
    return O.seeded_dropout_masks(seed, disc, n)


def hook_dropout(disc, masks):
    """Make the reference's nn.Dropout2d use the seeded masks (output = input * mask/0.75, the
    same arithmetic torch's feature dropout performs)."""
    handles = []
    for name, blk in disc.named_children():
        if not name.startswith("discriminator_block"):
            continue
        drop = blk[2]
        assert isinstance(drop, torch.nn.Dropout2d)
        m = masks[name]
        handles.append(drop.register_forward_hook(
            lambda mod, inp, out, m=m: inp[0] * m[:, :, None, None] if mod.training else out))
    return handles


HP = dict(lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0, lr_d_mult=1.0,
          lr_cls=1.0, lr_mse=10.0, lr_adv_g=1.0, lr_adv_d=0.01)
WATCH = ["gen_flow_model.predict_flow.weight", "gen_flow_model.predict_flow.bias",
         "gen_flow_model.conv_0.0.weight", "gen_flow_model.conv_2.0.bias",
         "base_model.fc.bias", "base_model.fc.weight", "base_model.bn1.running_mean",
         "base_model.layer4.1.bn2.running_var", "base_model.layer1.0.conv1.weight"]
WATCH_D = ["discriminator.adv_layer.bias", "discriminator.discriminator_block_1.0.bias",
           "discriminator.discriminator_block_2.3.running_var",
           "discriminator.discriminator_block_4_3.3.weight",
           "discriminator.discriminator_block_3_2.0.bias"]


def watch(model, keys):
    sd = model.state_dict()
    out = {}
    for k in keys:
        v = sd[k]
        out[k] = npy(v if v.numel() <= 4096 else v.reshape(-1)[:4096])
    return out


def main():
    install_stubs()
    ref_model = import_ref("dmcnet", "model")
    ref_gan = import_ref("dmcnet_GAN", "model")
    ref_data = import_ref("dmcnet", "dataset")
    out = {}

    # ---------------- G1: EstimatorDenseNetTiny forward + grads --------------------------
    g1 = {}
    est = O.seeded_state_fill(ref_model.EstimatorDenseNetTiny(5), seed=11)
    for tag, shape, sd in (("small", (2, 5, 40, 40), 101), ("ragged", (3, 5, 19, 37), 102),
                           ("frame", (1, 5, 224, 224), 103)):
        x = torch.from_numpy(np.random.RandomState(sd).standard_normal(shape).astype(np.float32))
        r = torch.from_numpy(np.random.RandomState(sd + 50).standard_normal(
            (shape[0], 2) + shape[2:]).astype(np.float32))
        est.zero_grad()
        y = est(x)
        (y * r).sum().backward()
        if tag == "frame":
            g1[tag + "_out_checksum"] = checksum(y)
            g1[tag + "_out_slice"] = npy(y[0, :, 100:108, 0:16])
        else:
            g1[tag + "_out"] = npy(y)
        for k, p in est.named_parameters():
            g1["%s_grad_%s" % (tag, k)] = npy(p.grad)
    np.savez_compressed(os.path.join(HERE, "g1_generator.npz"), **g1)
    print("G1 done")

    # ---------------- G2: Model.forward, eval mode --------------------------------------
    g2 = {}
    batch = O.synthetic_batch(seed=21, batch=2, num_segments=3, num_class=51, flow_ds_factor=16)
    flow, mv, res, target = batch
    m = quiet(ref_model.Model, 51, 3, "mv", base_model="resnet18", use_databn=0,
              gen_flow_or_delta=1, arch_estimator="DenseNetTiny")
    O.seeded_state_fill(m, seed=22).eval()
    with torch.no_grad():
        logits, gen_flow = m(mv, res)
    g2["dmcnet_logits"] = npy(logits)
    g2["dmcnet_genflow_checksum"] = checksum(gen_flow)
    g2["dmcnet_genflow_slice"] = npy(gen_flow[:, :, 64:72, 200:224])
    mg = quiet(ref_gan.Model, 51, 3, "mv", base_model="resnet18", use_databn=0,
               gen_flow_or_delta=1, arch_estimator="DenseNetTiny", arch_d="Discriminator3")
    O.seeded_state_fill(mg, seed=23).eval()
    with torch.no_grad():
        lo, va, gf = mg(mv, res, flow)
        lo2, va2, _ = mg(mv, res)
    g2["gan_logits"], g2["gan_validity_fake_real"], g2["gan_validity_fake"] = npy(lo), npy(va), npy(va2)
    g2["gan_genflow_checksum"] = checksum(gf)
    # other estimator / discriminator variants, eval mode, one segment of 2 clips at 64x64 is
    # not possible for D (Linear is sized for 224) -> estimators at 48x48, discriminators at 224.
    xs = torch.from_numpy(np.random.RandomState(24).standard_normal((2, 5, 48, 48)).astype(np.float32))
    for arch, cls in (("DenseNet", "EstimatorDenseNet"), ("DenseNetSmall", "EstimatorDenseNetSmall"),
                      ("DenseNetTinyEarlyFusionSum", "EstimatorDenseNetTinyEarlyFusionSum"),
                      ("DenseNetTinyEarlyFusionStack", "EstimatorDenseNetTinyEarlyFusionStack")):
        e = O.seeded_state_fill(getattr(ref_model, cls)(5), seed=25).eval()
        with torch.no_grad():
            g2["est_" + arch] = npy(e(xs))
    for att in (0, 1):
        cls = ref_model.ContextNetworkAtt if att else ref_model.ContextNetwork
        e = O.seeded_state_fill(cls(5, True, 0), seed=26).eval()
        with torch.no_grad():
            y = e(xs)
        if att:
            g2["est_ContextNetworkAtt_flow"], g2["est_ContextNetworkAtt_att"] = npy(y[0]), npy(y[1])
        else:
            g2["est_ContextNetwork"] = npy(y)
    xd = torch.from_numpy(np.random.RandomState(27).standard_normal((2, 2, 224, 224)).astype(np.float32))
    for arch in ("Discriminator", "Discriminator2", "Discriminator3", "Discriminator4", "Discriminator5"):
        d = O.seeded_state_fill(getattr(ref_gan, arch)(2), seed=28).eval()
        with torch.no_grad():
            g2["disc_" + arch] = npy(d(xd))
    np.savez_compressed(os.path.join(HERE, "g2_model_eval.npz"), **g2)
    print("G2 done")

    # ---------------- G3: train-mode Discriminator3 with seeded dropout masks ------------
    g3 = {}
    d3 = O.seeded_state_fill(ref_gan.Discriminator3(2), seed=31).train()
    xin = torch.from_numpy(np.random.RandomState(32).standard_normal((4, 2, 224, 224)).astype(np.float32))
    xin.requires_grad_(True)
    masks = dropout_masks(33, d3, 4)
    hs = hook_dropout(d3, masks)
    v = d3(xin)
    tgt = torch.tensor([0, 0, 1, 1])
    loss = torch.nn.functional.cross_entropy(v, tgt)
    loss.backward()
    for h in hs:
        h.remove()
    g3["validity"], g3["loss"] = npy(v), npy(loss)
    g3["grad_in_checksum"] = checksum(xin.grad)
    g3["grad_in_slice"] = npy(xin.grad[:, :, 100:104, 100:116])
    sd = d3.state_dict()
    for k in ("discriminator_block_1_2.3.running_mean", "discriminator_block_1_2.3.running_var",
              "discriminator_block_4_3.3.running_mean", "discriminator_block_4_3.3.running_var"):
        g3["post_" + k] = npy(sd[k])
    for k in ("discriminator_block_1.0.weight", "discriminator_block_1_2.0.bias",
              "discriminator_block_2.3.weight", "discriminator_block_4_3.3.bias", "adv_layer.bias"):
        g3["grad_" + k] = npy(dict(d3.named_parameters())[k].grad)
    np.savez_compressed(os.path.join(HERE, "g3_disc_train.npz"), **g3)
    print("G3 done")

    # ---------------- G4: full training steps through the reference's own train() --------
    g4 = {}
    args = types.SimpleNamespace(gpus=[0], num_segments=3, lr=HP["lr"],
                                 weight_decay=HP["weight_decay"])
    # (a) dmcnet: one unfrozen step and one frozen step
    tr = exec_ref_train("dmcnet", args)
    for tag, freeze in (("dmcnet", False), ("dmcnet_frozen", True)):
        torch.manual_seed(0)
        m = quiet(ref_model.Model, 51, 3, "mv", base_model="resnet18", use_databn=0,
                  gen_flow_or_delta=1, arch_estimator="DenseNetTiny")
        O.seeded_state_fill(m, seed=41)
        m2 = quiet(ref_model.Model, 51, 3, "mv", base_model="resnet18", use_databn=0,
                   gen_flow_or_delta=1, arch_estimator="DenseNetTiny")
        m2.load_state_dict(m.state_dict())
        batch = O.synthetic_batch(seed=42, batch=2, num_segments=3, num_class=51, flow_ds_factor=16)
        oc, og = O.make_optimizers(m, HP["lr"], HP["weight_decay"], HP["lr_cls_mult"], HP["lr_mse_mult"])
        for o in (oc, og):    # what main() does before each epoch, code/dmcnet/train.py:177-178
            quiet(tr["adjust_learning_rate"], o, 0, [20, 35, 45], 0.1,
                  **({"freeze": True, "epoch_thre": 1 if freeze else 0} if o is oc else {}))
        with cuda_is_a_noop():
            quiet(tr["train"], [batch], m, torch.nn.CrossEntropyLoss(), torch.nn.MSELoss(), oc, og,
                  0, 0.0, 0.0, HP["lr_cls"], HP["lr_mse"], 0, freeze=freeze)
        # the same step through the oracle's restated step on an identical copy
        oc2, og2 = O.make_optimizers(m2, HP["lr"], HP["weight_decay"], HP["lr_cls_mult"], HP["lr_mse_mult"])
        O.adjust_learning_rate(oc2, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"],
                               freeze=True, epoch_thre=1 if freeze else 0)
        O.adjust_learning_rate(og2, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"])
        m2.train()
        r = O.dmcnet_train_step(m2, oc2, og2, batch, 3, HP["lr_cls"], HP["lr_mse"], freeze=freeze)
        for k in m.state_dict():
            a, b = m.state_dict()[k], m2.state_dict()[k]
            assert torch.equal(a, b), "restated step != reference train() at " + k
        for k, v_ in watch(m, WATCH).items():
            g4["%s_post_%s" % (tag, k)] = v_
        for k in ("loss", "loss_cls", "loss_mse", "output"):
            g4["%s_%s" % (tag, k)] = npy(r[k])
        g4[tag + "_genflow_checksum"] = checksum(r["gen_flow"])
    print("G4 dmcnet done (reference train() == restated step, bit-exact)")

    # (b) GAN: D step (i=0) then G step (i=1) in one call of the reference's train()
    trg = exec_ref_train("dmcnet_GAN", args)
    mg = quiet(ref_gan.Model, 51, 3, "mv", base_model="resnet18", use_databn=0,
               gen_flow_or_delta=1, arch_estimator="DenseNetTiny", arch_d="Discriminator3")
    O.seeded_state_fill(mg, seed=43)
    mg2 = quiet(ref_gan.Model, 51, 3, "mv", base_model="resnet18", use_databn=0,
                gen_flow_or_delta=1, arch_estimator="DenseNetTiny", arch_d="Discriminator3")
    mg2.load_state_dict(mg.state_dict())
    b0 = O.synthetic_batch(seed=44, batch=2, num_segments=3, num_class=51, flow_ds_factor=0)
    b1 = O.synthetic_batch(seed=45, batch=2, num_segments=3, num_class=51, flow_ds_factor=0)
    masks_d = dropout_masks(46, mg.discriminator, 12)    # D step sees 2*B*S = 12 frames
    masks_g = dropout_masks(47, mg.discriminator, 6)     # G step sees B*S = 6 frames

    class TwoStepMasks(object):
        """Dropout hooks whose masks switch from the D-step set to the G-step set."""
        def __init__(self, disc):
            self.step = 0
            for name, blk in disc.named_children():
                if name.startswith("discriminator_block"):
                    blk[2].register_forward_hook(self._mk(name))

        def _mk(self, name):
            def hook(mod, inp, out):
                m_ = (masks_d if inp[0].shape[0] == 12 else masks_g)[name]
                return inp[0] * m_[:, :, None, None]
            return hook

    TwoStepMasks(mg.discriminator)
    TwoStepMasks(mg2.discriminator)
    opts = O.make_optimizers(mg, HP["lr"], HP["weight_decay"], HP["lr_cls_mult"], HP["lr_mse_mult"], HP["lr_d_mult"])
    for o in opts:
        quiet(trg["adjust_learning_rate"], o, 0, [20, 35, 45], 0.1)
    with cuda_is_a_noop():
        quiet(trg["train"], [b0, b1], mg, torch.nn.CrossEntropyLoss(), torch.nn.MSELoss(),
              opts[0], opts[1], opts[2], 0, 0.0, 0.0, 0.0, HP["lr_cls"], HP["lr_adv_g"],
              HP["lr_adv_d"], HP["lr_mse"], 0)
    opts2 = O.make_optimizers(mg2, HP["lr"], HP["weight_decay"], HP["lr_cls_mult"], HP["lr_mse_mult"], HP["lr_d_mult"])
    for o in opts2:
        O.adjust_learning_rate(o, 0, [20, 35, 45], 0.1, HP["lr"], HP["weight_decay"])
    mg2.train()
    for i, b in enumerate((b0, b1)):
        r = O.gan_train_step(mg2, opts2[0], opts2[1], opts2[2], b, i, 3, HP["lr_cls"],
                             HP["lr_adv_g"], HP["lr_adv_d"], HP["lr_mse"])
        tag = "gan_D" if i == 0 else "gan_G"
        for k in ("loss", "loss_cls", "loss_adv", "output", "validity") + (("loss_mse",) if i else ()):
            g4["%s_%s" % (tag, k)] = npy(r[k])
        g4[tag + "_genflow_checksum"] = checksum(r["gen_flow"])
        if i == 0:
            for k, v_ in watch(mg2, WATCH + WATCH_D).items():
                g4["gan_D_post_" + k] = v_
    for k in mg.state_dict():
        assert torch.equal(mg.state_dict()[k], mg2.state_dict()[k]), \
            "restated GAN steps != reference train() at " + k
    for k, v_ in watch(mg, WATCH + WATCH_D).items():
        g4["gan_G_post_" + k] = v_
    print("G4 GAN done (reference train() == restated D+G steps, bit-exact)")
    np.savez_compressed(os.path.join(HERE, "g4_train_steps.npz"), **g4)

    # ---------------- G5: integer index sampling -----------------------------------------
    g5 = {}
    ds = object.__new__(ref_data.CoviarDataSet)
    ds._representation = "mv"
    for n in (13, 14, 25, 50, 121, 300, 1000):
        for S in (3, 25):
            ds._num_segments = S
            g5["range_n%d_s%d" % (n, S)] = np.array(
                [ref_data.get_seg_range(n, S, seg, "mv") for seg in range(S)], dtype=np.int64)
            g5["test_n%d_s%d" % (n, S)] = np.array(
                [ds._get_test_frame_index(n, seg) for seg in range(S)], dtype=np.int64)
            for seed in (0, 1, 2):
                random.seed(seed)
                g5["train_n%d_s%d_seed%d" % (n, S, seed)] = np.array(
                    [ds._get_train_frame_index(n, seg) for seg in range(S)], dtype=np.int64)
    g5["gop_pos_mv"] = np.array([ref_data.get_gop_pos(v, "mv") for v in range(1, 61)], dtype=np.int64)
    g5["gop_pos_iframe"] = np.array([ref_data.get_gop_pos(v, "iframe") for v in range(0, 60)], dtype=np.int64)
    g5["range_iframe_n121_s3"] = np.array(
        [ref_data.get_seg_range(121, 3, seg, "iframe") for seg in range(3)], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "g5_index_sampling.npz"), **g5)
    print("G5 done")

    # ---------------- G6: adjust_learning_rate table -------------------------------------
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([{"params": p, "lr": 0.01, "lr_mult": 0.01, "decay_mult": 1.0}],
                           weight_decay=1e-4, eps=1e-3)
    rows = []
    for epoch in range(50):
        for freeze, thre in ((False, 500), (True, 1), (True, 0)):
            lr = tr["adjust_learning_rate"](opt, epoch, [20, 35, 45], 0.1, freeze=freeze, epoch_thre=thre)
            rows.append([epoch, int(freeze), thre, lr, opt.param_groups[0]["lr"],
                         opt.param_groups[0]["weight_decay"]])
    np.savez_compressed(os.path.join(HERE, "g6_lr_schedule.npz"), table=np.array(rows, dtype=np.float64))
    print("G6 done")


if __name__ == "__main__":
    main()
