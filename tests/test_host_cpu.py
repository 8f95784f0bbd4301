"""CPU tests of the host logic: index sampling (bit-exact vs the reference's vectors), tensor
contract, LR policy / optimiser grouping, transforms, and the gradient-bucket reducer on a
2-rank gloo group."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

import dmcnet_amd
from dmcnet_amd import dataset, ddp, train, transforms
from oracle import dmc_oracle as O


def test_index_sampling_bit_exact_vs_reference_vectors(golden):
    g = golden("g5_index_sampling")
    for n in (13, 14, 25, 50, 121, 300, 1000):
        for S in (3, 25):
            got = np.array([dataset.get_seg_range(n, S, s, "mv") for s in range(S)], dtype=np.int64)
            assert np.array_equal(got, g["range_n%d_s%d" % (n, S)])
            got = np.array([dataset.test_frame_index(n, s, S, "mv") for s in range(S)], dtype=np.int64)
            assert np.array_equal(got, g["test_n%d_s%d" % (n, S)])
            for seed in (0, 1, 2):
                random.seed(seed)
                got = np.array([dataset.train_frame_index(n, s, S, "mv") for s in range(S)], dtype=np.int64)
                assert np.array_equal(got, g["train_n%d_s%d_seed%d" % (n, S, seed)])
    assert np.array_equal(np.array([dataset.get_gop_pos(v, "mv") for v in range(1, 61)]), g["gop_pos_mv"])
    assert np.array_equal(np.array([dataset.get_gop_pos(v, "iframe") for v in range(60)]), g["gop_pos_iframe"])
    assert np.array_equal(np.array([dataset.get_seg_range(121, 3, s, "iframe") for s in range(3)]),
                          g["range_iframe_n121_s3"])
    assert dataset.flow_frame_number(2, 5) == 30


def test_tensor_contract_matches_oracle():
    frames = O.synthetic_frames_u8(3, 2, 3, size=50)       # ragged: 50 is not a multiple of 16
    for f in (0, 16):
        a = dataset.to_tensors(frames[0], f)
        b = O.normalize_sample(frames[0], f)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    ds = dataset.SyntheticCoviarDataSet(4, 51, num_segments=3, flow_ds_factor=16, size=32)
    flow, mv, res, label = ds[1]
    assert flow.shape == (3, 2, 32, 32) and mv.shape == (3, 2, 32, 32) and res.shape == (3, 3, 32, 32)
    assert 0 <= label < 51 and len(ds) == 4
    blocks = flow.reshape(3, 2, 2, 16, 2, 16)
    assert torch.equal(blocks, blocks[..., :1, :, :1].expand_as(blocks))    # blockified flow


def test_lr_schedule_and_param_groups(golden):
    table = golden("g6_lr_schedule")["table"]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = train.GroupedAdam([{"params": p, "lr": 0.01, "lr_mult": 0.01, "decay_mult": 1.0}],
                            weight_decay=1e-4, eps=1e-3)
    for epoch, freeze, thre, lr, glr, gwd in table:
        got = train.adjust_learning_rate(opt, int(epoch), [20, 35, 45], 0.1, 0.01, 1e-4,
                                         freeze=bool(freeze), epoch_thre=int(thre))
        assert got == lr and opt.param_groups[0]["lr"] == glr and opt.param_groups[0]["weight_decay"] == gwd
    m = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1,
                         arch_estimator="DenseNetTiny", arch_d="Discriminator3")
    oc, og, od = train.make_optimizers(m, 0.01, 1e-4, 0.01, 1.0, 1.0)
    assert (len(oc.param_groups), len(og.param_groups), len(od.param_groups)) == (62, 12, 48)
    ro = O.make_optimizers(O.OracleModel(51, 3, "mv", base_model="resnet18", use_databn=0,
                                         gen_flow_or_delta=1, arch_estimator="DenseNetTiny",
                                         arch_d="Discriminator3"), 0.01, 1e-4, 0.01, 1.0, 1.0)
    for mine, ref in zip((oc, og, od), ro):
        for a, b in zip(mine.param_groups, ref.param_groups):
            assert a["lr_mult"] == b["lr_mult"] and a["decay_mult"] == b["decay_mult"]
            assert a["eps"] == b["eps"] == 1e-3 and a["params"][0].shape == b["params"][0].shape


def test_grouped_adam_equals_torch_adam():
    torch.manual_seed(0)
    ws = [torch.randn(5, 3), torch.randn(7), torch.randn(2, 2, 3, 3)]
    a = [torch.nn.Parameter(w.clone()) for w in ws]
    b = [torch.nn.Parameter(w.clone()) for w in ws]
    mk = lambda ps: [{"params": p, "lr": 0.01 * (1 + i), "weight_decay": 1e-4 * (i % 2)} for i, p in enumerate(ps)]
    oa, ob = train.GroupedAdam(mk(a), eps=1e-3), torch.optim.Adam(mk(b), eps=1e-3)
    for step in range(4):
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa)
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa, pb, rtol=1e-6, atol=1e-7)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert len(sa["param_groups"]) == len(sb["param_groups"]) == 3
    assert set(sa["state"][0].keys()) == set(sb["state"][0].keys())


def test_accuracy_and_meters():
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5]])
    tgt = torch.tensor([1, 2, 2])
    p1, p2 = train.accuracy(out, tgt, topk=(1, 2))
    q1, q2 = O.accuracy(out, tgt, topk=(1, 2))
    assert float(p1) == float(q1) and float(p2) == float(q2)
    m = train.AverageMeter()
    m.update(2.0, 2); m.update(4.0, 2)
    assert m.avg == 3.0 and m.val == 4.0


def test_transforms_contract():
    random.seed(0)
    frames = [np.random.RandomState(i).randint(0, 256, (256, 340, 7)).astype(np.uint8) for i in range(3)]
    aug = transforms.Compose([transforms.GroupMultiScaleCrop(224, [1, .875, .75]),
                              transforms.GroupRandomHorizontalFlip()])
    out = aug(frames)
    assert len(out) == 3 and all(o.shape == (224, 224, 7) for o in out)
    f = transforms.flip_with_x_negation(frames[0])
    assert np.array_equal(f[:, :, 1], frames[0][:, ::-1, 1].astype(np.int32))          # flow y kept
    assert np.array_equal(f[:, :, 0], 256 - frames[0][:, ::-1, 0].astype(np.int32))    # flow x negated
    assert np.array_equal(f[:, :, 2], 256 - frames[0][:, ::-1, 2].astype(np.int32))    # mv x negated
    val = transforms.Compose([transforms.GroupScale(256), transforms.GroupCenterCrop(224)])(frames)
    assert all(o.shape == (224, 224, 7) for o in val)
    ident = transforms.resize_bilinear(frames[0], 256, 340)
    assert ident is frames[0]


def test_checkpoint_key_layout(tmp_path):
    m = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, arch_estimator="DenseNetTiny")
    sd = train.reference_state_dict(m)
    assert all(k.startswith("module.") for k in sd)
    assert "module.gen_flow_model.conv_0.0.weight" in sd and "module.base_model.fc.bias" in sd
    m2 = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, arch_estimator="DenseNetTiny")
    res = train.load_reference_weights(m2, sd)
    assert not res.missing_keys and not res.unexpected_keys
    os.chdir(tmp_path)
    name = train.save_checkpoint({"epoch": 1, "arch": "resnet18", "state_dict": sd, "best_prec1": 0.0},
                                 True, "model", "mv")
    assert os.path.exists(name) and os.path.exists("model_mv_model_best.pth.tar")


# ----------------------------------------------------------------------------- 2-rank gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)          # different init per rank: the reducer must broadcast
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4),
                              torch.nn.ReLU(), torch.nn.Linear(4, 3))
    red = ddp.GradBucketReducer(list(net.parameters()), bucket_bytes=128)    # several buckets
    red.time_waits = True
    torch.manual_seed(7)
    x, t = torch.randn(8, 6), torch.randint(0, 3, (8,))
    shard = slice(rank * 4, rank * 4 + 4)
    for p in net.parameters():
        p.grad = None
    red.begin()
    torch.nn.functional.cross_entropy(net(x[shard]), t[shard]).backward()
    red.finish()
    # second step with one layer frozen out of the graph (unused parameters keep grad None)
    for p in net.parameters():
        p.grad = None
    net[4].weight.requires_grad_(False)
    red.begin()
    torch.nn.functional.cross_entropy(net(x[shard]), t[shard]).backward()
    red.finish()
    torch.save({"params": [p.detach().clone() for p in net.parameters()],
                "grads": [None if p.grad is None else p.grad.clone() for p in net.parameters()],
                "x": x, "t": t, "nbuckets": len(red.buckets), "comm": red.comm_summary()},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_grad_bucket_reducer_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "r%d.pt" % r)) for r in (0, 1))
    assert r0["nbuckets"] > 1
    # the "comm" object of bench.py --gpus N (the first thing to read after a multi-GPU run): schema and plausibility
    c = r0["comm"]
    assert set(c) >= {"backend", "world_size", "buckets", "reduce_op", "last_step_bytes_by_set", "last_step_launched_from",
                      "timed_steps", "exposed_wait_ms_per_step", "exposed_wait_host_ms_per_step"}
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["timed_steps"] == 2
    assert len(c["buckets"]) == r0["nbuckets"] and all(b["set"] == "all" and b["bytes"] > 0 for b in c["buckets"])
    # the second step froze one weight: its bucket travels whole (completed with zeros in finish())
    assert sum(g.numel() * 4 for g in r0["grads"] if g is not None) <= sum(c["last_step_bytes_by_set"].values()) \
        <= sum(b["bytes"] for b in c["buckets"])
    assert "finish" in c["last_step_launched_from"]
    assert c["exposed_wait_ms_per_step"] is None and c["exposed_wait_host_ms_per_step"] >= 0.0   # CPU buckets: host time only
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                      # broadcast from rank 0
    for a, b in zip(r0["grads"], r1["grads"]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)                  # identical after the all-reduce
    # equal to the single-process gradient of the mean loss over the whole batch
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4),
                              torch.nn.ReLU(), torch.nn.Linear(4, 3))
    with torch.no_grad():
        for p, v in zip(net.parameters(), r0["params"]):
            p.copy_(v)
    net[4].weight.requires_grad_(False)
    torch.nn.functional.cross_entropy(net(r0["x"]), r0["t"]).backward()
    for p, g in zip(net.parameters(), r0["grads"]):
        if p.grad is None:
            assert g is None
        else:
            torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-7)


class _ToyGan(torch.nn.Module):
    """Three sub-modules with the reference's attribute names (what the optimizer routing and the
    bucket plan key on); the generator is a conv so that the phases of the real steps can be
    emulated on the CPU (the real generator has no CPU path)."""

    def __init__(self):
        super().__init__()
        self.gen_flow_model = torch.nn.Conv2d(5, 2, 3, padding=1)
        self.base_model = torch.nn.Sequential(torch.nn.Conv2d(2, 8, 3, padding=1), torch.nn.ReLU(),
                                              torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(),
                                              torch.nn.Linear(8, 4))
        self.discriminator = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3, stride=2, padding=1),
                                                 torch.nn.Flatten(), torch.nn.Linear(4 * 4 * 4, 2))


def _phase_losses(net, x, flow, t, phase):
    g = net.gen_flow_model(x)
    if phase == "dmcnet":          # classifier sees a detached cue; MSE graph independent
        return F.cross_entropy(net.base_model(g.detach()), t) + 10.0 * F.mse_loss(g, flow)
    if phase == "D":               # generator weights are constants in the D step
        with train._without_param_grads(net.gen_flow_model):
            g = net.gen_flow_model(x)
            v = net.discriminator(torch.cat((g, flow), 0))
            tv = torch.cat((torch.zeros(len(x)), torch.ones(len(x)))).long()
            return F.cross_entropy(net.base_model(g), t) + 0.01 * F.cross_entropy(v, tv)
    with train._without_param_grads(net.base_model, net.discriminator):      # G step
        g = net.gen_flow_model(x)
        return (F.cross_entropy(net.base_model(g), t) + 10.0 * F.mse_loss(g, flow)
                + F.cross_entropy(net.discriminator(g), torch.ones(len(x)).long()))


def _ddp_buffers_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(500 + rank)
    net = _ToyGan()
    net.base_model.add_module("bn", torch.nn.BatchNorm1d(4))
    net.gen_flow_model.weight.data = net.gen_flow_model.weight.data.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():                         # rank 1 resumed from somewhere else: other statistics, a frozen bias
        net.base_model.bn.running_mean.fill_(1.0 + rank)
        net.base_model.bn.running_var.fill_(2.0 + rank)
        net.base_model.bn.num_batches_tracked.fill_(10 * (rank + 1))
    net.discriminator[2].bias.requires_grad_(False)
    red = ddp.for_model(net)
    torch.save({"state": {k: v.clone() for k, v in net.state_dict().items()},
                "cl": net.gen_flow_model.weight.is_contiguous(memory_format=torch.channels_last),
                "in_buckets": sum(len(e) for _, e in red.buckets)}, os.path.join(out_dir, "buf_r%d.pt" % rank))
    dist.destroy_process_group()


def test_reducer_broadcasts_buffers_and_frozen_parameters(tmp_path):
    """ddp.for_model: every rank starts from rank 0's parameters (trainable or frozen) AND buffers -- BatchNorm running
    statistics and num_batches_tracked that differ on rank 1 (a resume on one rank only) are overwritten; memory formats
    survive; frozen parameters are broadcast although they are in no bucket."""
    port = _free_port()
    mp.spawn(_ddp_buffers_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "buf_r%d.pt" % r)) for r in (0, 1))
    assert list(r0["state"]) == list(r1["state"])
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    assert float(r1["state"]["base_model.bn.running_mean"][0]) == 1.0 and int(r1["state"]["base_model.bn.num_batches_tracked"]) == 10
    assert r0["cl"] and r1["cl"]
    assert r0["in_buckets"] == len(r0["state"]) - 3 - 1          # 3 buffers and the frozen bias are in no bucket


def _ddp_plan_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(200 + rank)
    net = _ToyGan()
    red = ddp.for_model(net, bucket_bytes=256)       # base_model / discriminator split in several buckets
    torch.manual_seed(11)
    x, flow, t = torch.randn(4, 5, 8, 8), torch.randn(4, 2, 8, 8), torch.randint(0, 4, (4,))
    sh = slice(rank * 2, rank * 2 + 2)
    rec = {"sets": [n for n, _ in red.sets], "bucket_set": list(red.bucket_set), "x": x, "flow": flow, "t": t,
           "state": {k: v.clone() for k, v in net.state_dict().items()}}
    for phase in ("dmcnet", "D", "G"):
        for p in net.parameters():
            p.grad = None
        red.begin()
        _phase_losses(net, x[sh], flow[sh], t[sh], phase).backward()
        red.finish()
        rec[phase] = {"reduced": list(red.last_reduced), "bytes": red.reduced_bytes(by_set=True),
                      "grads": {k: (None if p.grad is None else p.grad.clone()) for k, p in net.named_parameters()}}
    torch.save(rec, os.path.join(out_dir, "p%d.pt" % rank))
    dist.destroy_process_group()


def test_bucket_plan_per_optimizer_two_ranks_gloo(tmp_path):
    """one bucket set per optimizer; the dmcnet step reduces classifier + generator,
    the GAN D step classifier + discriminator only, the G step the generator's few bytes only --
    and never anything from finish()."""
    port = _free_port()
    mp.spawn(_ddp_plan_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "p%d.pt" % r)) for r in (0, 1))
    assert r0["sets"] == ["base_model", "gen_flow_model", "discriminator"]
    assert r0["bucket_set"].count("gen_flow_model") == 1 and r0["bucket_set"].count("discriminator") > 1
    net = _ToyGan()
    net.load_state_dict(r0["state"])
    nbytes = {tag: sum(p.numel() * 4 for k, p in net.named_parameters() if tag in k) for tag in r0["sets"]}
    expect = {"dmcnet": {"base_model", "gen_flow_model"}, "D": {"base_model", "discriminator"},
              "G": {"gen_flow_model"}}
    for phase, sets in expect.items():
        rec = r0[phase]
        assert set(rec["bytes"]) == sets, (phase, rec["bytes"])
        assert all(where == "hook" for _, _, _, where in rec["reduced"])         # nothing left to finish()
        for tag in sets:
            assert rec["bytes"][tag] == nbytes[tag]
        if phase == "G":
            assert sum(rec["bytes"].values()) == nbytes["gen_flow_model"]         # 372 B here, 18 KB in the real model
        # equal on both ranks, and equal to the single-process gradient over the whole batch
        for p in net.parameters():
            p.grad = None
        _phase_losses(net, r0["x"], r0["flow"], r0["t"], phase).backward()
        for k, p in net.named_parameters():
            g0, g1 = rec["grads"][k], r1[phase]["grads"][k]
            assert (g0 is None) == (p.grad is None), (phase, k)
            if g0 is not None:
                assert torch.equal(g0, g1)
                torch.testing.assert_close(g0, p.grad, rtol=1e-5, atol=1e-7)


def _ddp_cl_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 4, 3, padding=1)).to(memory_format=torch.channels_last)
    red = ddp.GradBucketReducer(list(net.parameters()), bucket_bytes=256)
    torch.manual_seed(9)
    x = torch.randn(4, 3, 6, 6)
    shard = slice(rank * 2, rank * 2 + 2)
    red.begin()
    net(x[shard]).square().mean().backward()
    red.finish()
    ok = all(p.grad.stride() == p.stride() for p in net.parameters())      # bucket views keep the parameter's strides
    torch.save({"grads": [p.grad.clone() for p in net.parameters()], "x": x, "strides_ok": ok,
                "state": net.state_dict()}, os.path.join(out_dir, "c%d.pt" % rank))
    dist.destroy_process_group()


def test_grad_bucket_reducer_keeps_channels_last_strides(tmp_path):
    """channels_last weights: the gradient views into the flat buckets keep the parameters' memory
    order (what torch's fused Adam kernel requires) and still average to the full-batch gradient."""
    port = _free_port()
    mp.spawn(_ddp_cl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "c%d.pt" % r)) for r in (0, 1))
    assert r0["strides_ok"] and r1["strides_ok"]
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(8, 4, 3, padding=1)).to(memory_format=torch.channels_last)
    net.load_state_dict(r0["state"])
    net(r0["x"]).square().mean().backward()
    for p, g0, g1 in zip(net.parameters(), r0["grads"], r1["grads"]):
        assert torch.equal(g0, g1)
        torch.testing.assert_close(g0, p.grad, rtol=1e-5, atol=1e-7)


def test_score_fusion_matches_reference_combine(golden, tmp_path):
    """combine() on the reference's shipped HMDB-51 split-1 score files reproduces the accuracy the
    reference's own combine.py prints for them (fixture g7)."""
    from dmcnet_amd import evaluate
    g = golden("g7_score_fusion")
    exp = dict(zip(g["expected_names"].tolist(), g["expected_acc"].tolist()))
    part = lambda t: (g["hmdb51_split1_" + t].astype(np.float64), g["hmdb51_split1_labels_" + t])
    for tag, name in (("dmc", "hmdb51/gen_flow/split1"), ("dmc_gan", "hmdb51/gan/split1")):
        acc, comb = evaluate.combine(part("iframe"), part("mv"), part("residual"), part(tag))
        assert abs(acc - exp[name]) < 1e-6 and comb.shape == (1530, 51)
    assert abs(exp["hmdb51/gen_flow/split1"] - 0.639216) < 1e-6       # 63.92 %
    # weights matter: the I-frame stream carries weight 2
    acc_eq, _ = evaluate.combine(part("iframe"), part("mv"), part("residual"), part("dmc"), wi=1.0)
    assert acc_eq != exp["hmdb51/gen_flow/split1"]
    # dump / load round trip in the reference's layout (sorted by name)
    names = ["b.avi", "a.avi", "c.avi"]
    output = [(np.full((1, 51), i, np.float32), i) for i in range(3)]
    p = str(tmp_path / "s.npz")
    evaluate.save_scores(p, output, names)
    s, l, n = evaluate.load_scores(p)
    assert n.tolist() == ["a.avi", "b.avi", "c.avi"] and l.tolist() == [1, 0, 2]
    assert s.shape == (3, 51) and s[0, 0] == 1.0
    with pytest.raises(ValueError):
        evaluate.combine((s, l), (s, l[::-1].copy()), (s, l))


def test_miopen_find_db_is_shipped_and_selected(monkeypatch, tmp_path):
    """enable_find() turns on cudnn.benchmark (the reference's train.py:118) and points MIOpen at
    the shipped find-db unless the user already chose a path."""
    import os
    from dmcnet_amd import miopen
    names = os.listdir(miopen._DB_DIR)
    assert any(n.endswith(".ufdb.txt") for n in names) and any(n.endswith(".udb.txt") for n in names)
    prev = torch.backends.cudnn.benchmark
    try:
        monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
        path = miopen.enable_find()
        assert torch.backends.cudnn.benchmark is True
        assert os.environ["MIOPEN_USER_DB_PATH"] == path and os.path.isdir(path)
        monkeypatch.setenv("MIOPEN_USER_DB_PATH", str(tmp_path))
        assert miopen.enable_find() == str(tmp_path)           # the user's choice wins
    finally:
        torch.backends.cudnn.benchmark = prev


def test_grouped_adam_multi_tensor_group_matches_torch_adam():
    """A param group holding several tensors (the bucket key grew per parameter and the
    unpack raised): GroupedAdam must equal torch.optim.Adam step for step."""
    torch.manual_seed(3)
    a = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    b = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 3))
    b.load_state_dict(a.state_dict())
    oa = train.GroupedAdam(a.parameters(), lr=0.01, weight_decay=1e-4, eps=1e-3)     # ONE group, four tensors
    ob = torch.optim.Adam(b.parameters(), lr=0.01, weight_decay=1e-4, eps=1e-3)
    x = torch.randn(6, 5)
    for _ in range(3):
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            net(x).square().mean().backward()
            opt.step()
    for p, q in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(p, q, rtol=1e-6, atol=1e-7)


def test_flow_loss_variants_and_attention_weighting():
    """--loss_mse choices (code/dmcnet/train.py:166-172) and the att-weighted form (:335); the MSE
    kernel itself has no CPU path, so the plain MSELoss case raises loudly here."""
    g = torch.randn(2, 2, 8, 8, requires_grad=True)
    f, att = torch.randn(2, 2, 8, 8), torch.rand(2, 2, 8, 8, requires_grad=True)
    assert torch.equal(train.flow_loss("L1", g, f), F.l1_loss(g, f))
    assert torch.equal(train.flow_loss("SmoothL1Loss", g, f), F.smooth_l1_loss(g, f))
    l = train.flow_loss("MSELoss", g, f, att)
    assert torch.equal(l, F.mse_loss(att * g, att * f))
    l.backward()
    assert att.grad is not None and g.grad is not None       # the gradient reaches att through both arguments
    with pytest.raises(ValueError):
        train.flow_loss("Huber", g, f)
    with pytest.raises(Exception):
        train.flow_loss("MSELoss", g, f)                      # HIP only
    m = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, arch_estimator="DenseNetTiny")
    with pytest.raises(ValueError):
        train.DmcnetTrainStep(m, 3, 1.0, 10.0, 0.01, 1e-4, 0.01, 1.0, att=1)      # model built with att=0
    with pytest.raises(ValueError):
        train.DmcnetTrainStep(m, 3, 1.0, 10.0, 0.01, 1e-4, 0.01, 1.0, loss_mse="Huber")


def test_resnet_build_warns_without_weights_and_uses_torchvision_init(tmp_path):
    from dmcnet_amd import resnet
    with pytest.warns(UserWarning, match="pretrained=True"):
        net = resnet.build("resnet18", pretrained=True)
    bn = net.layer1[0].bn1
    assert float(bn.weight.min()) == 1.0 and float(bn.bias.abs().max()) == 0.0
    w = net.layer2[0].conv1.weight                       # kaiming_normal_(fan_out): std = sqrt(2 / (128*9))
    assert abs(float(w.std()) - (2.0 / (128 * 9)) ** 0.5) < 0.1 * (2.0 / (128 * 9)) ** 0.5
    sd = {k: torch.full_like(v, 0.5) if v.is_floating_point() else v for k, v in net.state_dict().items()}
    path = str(tmp_path / "rn18.pt")
    torch.save(sd, path)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                   # with weights there is nothing to warn about
        net2 = resnet.build("resnet18", pretrained=True, weights=path)
    assert float(net2.conv1.weight.mean()) == 0.5


# ------------------------------------------------------------- a19: tensor contract vs the reference
def _coviar_dataset(tmp_path, is_train, minmax, with_flip, monkeypatch):
    import sys
    from tests.golden import coviar_fixture as CF
    monkeypatch.setitem(sys.modules, "coviar", CF.coviar_module())
    data_root, flow_root, lst = CF.write_dataset(str(tmp_path))
    ts = [transforms.GroupCenterCrop(CF.CROP)] + ([transforms.GroupRandomHorizontalFlip()] if with_flip else [])
    return dataset.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 0, False, transforms.Compose(ts),
                                 3, is_train, True, 12, mv_minmaxnorm=minmax)


def test_dataset_item_bit_exact_vs_reference_getitem(golden, tmp_path, monkeypatch):
    """SURVEY 8 a19, pinned: the reference's own CoviarDataSet.__getitem__ (code/dmcnet/dataset.py:151-281)
    was run on the seeded stand-in of tests/golden/coviar_fixture.py (golden G9,
    tests/golden/make_golden_dataset.py); this dataset must return the same 4-tuple bit for bit --
    sampling, RNG call order, the MV scaling, clipping, crop, flip with x negation, /255 and
    normalisation.  (flow_ds_factor = 16: the next test.)"""
    from tests.golden import coviar_fixture as CF
    g = golden("g9_dataset_item")
    flips = 0
    for tag, is_train, minmax, seed, index, with_flip in CF.CASES:
        ds = _coviar_dataset(tmp_path, is_train, minmax, with_flip, monkeypatch)
        assert len(ds) == len(CF.VIDEOS)
        random.seed(seed)
        flow, mv, res, label = ds[index]
        assert label == int(g[tag + "_label"])
        assert torch.equal(flow, torch.from_numpy(g[tag + "_flow"])), tag
        assert torch.equal(mv, torch.from_numpy(g[tag + "_mv"])), tag
        assert torch.equal(res, torch.from_numpy(g[tag + "_res"])), tag
        # the GPU-side plan draws the same random numbers and describes the same geometry
        random.seed(seed)
        frames, box, out, flip, label2 = ds.raw_item(index)
        assert label2 == label and out == (CF.CROP, CF.CROP) and frames.shape == (3, CF.H0, CF.W0, 7)
        assert box.tolist() == [(CF.H0 - CF.CROP) // 2, (CF.W0 - CF.CROP) // 2, CF.CROP, CF.CROP,
                                CF.CROP, CF.CROP, 0, 0]
        flips += int(flip)
        crop = [transforms.apply_plan(f, box, out, flip) for f in frames]
        back = dataset.to_tensors(np.transpose(np.array(crop), (0, 3, 1, 2)), 0)
        assert torch.equal(back[1], mv) and torch.equal(back[0], flow) and torch.equal(back[2], res)
    assert 0 < flips < sum(1 for c in CF.CASES if c[5])      # both flip outcomes are covered


def test_dataset_item_flow_ds_factor_16_vs_reference_getitem(golden, tmp_path, monkeypatch):
    """The flow_ds_factor = 16 leg of the contract (BASELINE config 2's setting; code/dmcnet/dataset.py:229-246) against the
    reference's own __getitem__ (golden g9_dataset_item_ds16: the reference's lines around the blockify call ran as they are,
    block_reduce itself -- skimage is absent -- as a numpy stand-in written from its documentation,
    tests/golden/make_golden_dataset.py): whole 16 x 16 blocks (crop 48) and ragged, zero-padded ones (crop 40), bit for bit;
    the GPU-side plan (raw item -> apply_plan -> to_tensors) likewise."""
    import sys
    from tests.golden import coviar_fixture as CF
    g = golden("g9_dataset_item_ds16")
    monkeypatch.setitem(sys.modules, "coviar", CF.coviar_module())
    data_root, flow_root, lst = CF.write_dataset(str(tmp_path))
    for tag, is_train, minmax, seed, index, with_flip in CF.CASES:
        for crop in CF.DS16_CROPS:
            ts = [transforms.GroupCenterCrop(crop)] + ([transforms.GroupRandomHorizontalFlip()] if with_flip else [])
            ds = dataset.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 16, False, transforms.Compose(ts),
                                       3, is_train, True, 12, mv_minmaxnorm=minmax)
            key = "%s_c%d" % (tag, crop)
            random.seed(seed)
            flow, mv, res, label = ds[index]
            assert label == int(g[key + "_label"])
            assert torch.equal(flow, torch.from_numpy(g[key + "_flow"])), key
            assert torch.equal(mv, torch.from_numpy(g[key + "_mv"])), key
            assert torch.equal(res, torch.from_numpy(g[key + "_res"])), key
            random.seed(seed)
            frames, box, out, flip, _ = ds.raw_item(index)
            cropd = [transforms.apply_plan(f, box, out, flip) for f in frames]
            back = dataset.to_tensors(np.transpose(np.array(cropd), (0, 3, 1, 2)), 16)
            assert torch.equal(back[0], flow) and torch.equal(back[1], mv) and torch.equal(back[2], res), key


def test_ten_crop_and_upsample_interp_vs_reference_getitem(golden, tmp_path, monkeypatch):
    """Round 6: the 10-crop test transform (GroupOverSample, code/dmcnet/transforms.py:77-114 -- five offsets x (crop, mirrored
    crop with negated x components), offset-major) and ``upsample_interp=True`` (code/dmcnet/dataset.py:236-246: block means
    interpolated linearly with scipy's interp1d, one axis after the other) against the reference's own __getitem__ with its
    own transform / its own interp1d lines (golden g9_dataset_item_extra), bit for bit."""
    import sys
    from tests.golden import coviar_fixture as CF
    g = golden("g9_dataset_item_extra")
    monkeypatch.setitem(sys.modules, "coviar", CF.coviar_module())
    data_root, flow_root, lst = CF.write_dataset(str(tmp_path))
    for tag, is_train, minmax, seed, index, _ in (CF.CASES[0], CF.CASES[2]):
        ds = dataset.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 0, False,
                                   transforms.Compose([transforms.GroupOverSample(CF.CROP, None)]), 3, is_train, True, 12,
                                   mv_minmaxnorm=minmax)
        random.seed(seed)
        flow, mv, res, label = ds[index]
        key = tag + "_over"
        assert tuple(flow.shape) == (30, 2, CF.CROP, CF.CROP) and label == int(g[key + "_label"])
        assert torch.equal(flow, torch.from_numpy(g[key + "_flow"])) and torch.equal(mv, torch.from_numpy(g[key + "_mv"]))
        assert torch.equal(res, torch.from_numpy(g[key + "_res"]))
        for crop in CF.DS16_CROPS:
            ds = dataset.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 16, True,
                                       transforms.Compose([transforms.GroupCenterCrop(crop)]), 3, is_train, True, 12,
                                       mv_minmaxnorm=minmax)
            random.seed(seed)
            flow, _, _, label = ds[index]
            key = "%s_interp_c%d" % (tag, crop)
            assert label == int(g[key + "_label"]) and torch.equal(flow, torch.from_numpy(g[key + "_flow"])), key
    # with the optional scale step: 5 offsets x 2 x frames, every crop of the requested size, flips are mirror images
    frames = [np.random.RandomState(i).randint(0, 256, (64, 80, 7)).astype(np.uint8) for i in range(2)]
    out = transforms.GroupOverSample(48, 56)(frames)
    assert len(out) == 20 and all(o.shape == (48, 48, 7) for o in out)
    assert np.array_equal(out[1][:, ::-1, 1], out[0][:, :, 1]) and np.array_equal(256 - out[1][:, ::-1, 0], out[0][:, :, 0])


def test_geometry_plan_matches_applied_transforms():
    """plan + (crop, resize, flip) == applying the Compose, for the reference's train and val pipelines."""
    rs = np.random.RandomState(5)
    frames = [rs.randint(0, 256, (256, 340, 7)).astype(np.uint8) for _ in range(2)]
    pipes = [transforms.Compose([transforms.GroupMultiScaleCrop(224, [1, .875, .75]),
                                 transforms.GroupRandomHorizontalFlip()]),
             transforms.Compose([transforms.GroupCenterCrop(224)]),
             transforms.Compose([transforms.GroupScale(256), transforms.GroupCenterCrop(224)])]
    for pipe in pipes:
        for seed in range(6):
            random.seed(seed)
            want = pipe(frames)
            random.seed(seed)
            plan, out, flip = transforms.geometry_plan(pipe, frames[0].shape)
            state = random.getstate()
            got = [transforms.apply_plan(f, plan, out, flip) for f in frames]
            for a, b in zip(got, want):
                assert np.array_equal(np.asarray(a), np.asarray(b))
            random.seed(seed)
            pipe(frames)
            assert random.getstate() == state                  # same RNG consumption
    with pytest.raises(ValueError):
        transforms.geometry_plan(transforms.Compose([transforms.GroupScale(256), transforms.GroupScale(224)]),
                                 (240, 320, 7))


# ------------------------------------------------------------------ I3D trainer policy (BASELINE config 5)
def _i3d_cfg(golden):
    g = golden("g10_i3d_trainer")
    return g, eval(str(g["cfg"]))


def _make_i3d_trainer(c, net, group=None):
    from dmcnet_amd import i3d_train as IT
    opts = IT.make_optimizers(net, c["lr_base"], c["lr_base2"], optim="adam", adv=c["adv"])
    mk = lambda base: IT.MultiFactorScheduler(list(c["sched_steps"]), base_lr=base, factor=c["lr_factor"])
    tr = IT.I3DTrainer(net, opts, mk(c["lr_base"]), mk(c["lr_base2"]), mk(c["lr_d"]), adv=c["adv"],
                       iter_size=c["iter_size"], epoch_thre=c["epoch_thre"], detach=c["detach"], group=group,
                       losses_fn=O.i3d_losses)      # the package's own assembly is HIP-only; same reference lines, restated on CPU ops
    return tr, opts


def test_i3d_scheduler_matches_reference_table(golden):
    """MultiFactorScheduler against the table the reference's class produced (warm-up halving for the
    first 99 updates, factor steps)."""
    from dmcnet_amd import i3d_train as IT
    g = golden("g10_i3d_trainer")
    s = IT.MultiFactorScheduler([2, 14, 18], base_lr=0.1, factor=0.1, step_counter=2)
    np.testing.assert_array_equal(np.array([s.update() for _ in range(130)]), g["sched_table"])


def test_i3d_trainer_policy_vs_reference_fit(golden):
    """The reference's own ``model.fit`` (3 epochs x 8 micro-batches, iter_size 2, two stages, D / G
    alternation) against I3DTrainer on the same tiny network: optimizer group layout, every learning
    rate and EVERY parameter after every micro-batch, bit for bit: same torch CPU ops in the same order, on ONE
    thread like the golden run (the summation order of the CPU convolution backward depends on the thread
    count, and Adam(eps 1e-8) amplifies a 1-ulp difference in a near-zero gradient to a full +-lr step)."""
    from tests.golden import tiny_i3d
    g, c = _i3d_cfg(golden)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    net = tiny_i3d.build(c["seed_net"])
    np.testing.assert_array_equal(tiny_i3d.snapshot(net), g["init"])
    tr, opts = _make_i3d_trainer(c, net)
    layout = [(k, len(opts[k].param_groups), [len(gr["params"]) for gr in opts[k].param_groups],
               [gr.get("lr_mult", 1.0) for gr in opts[k].param_groups]) for k in sorted(opts)]
    assert repr(layout) == str(g["group_layout"])
    assert opts["optimizer_3"].defaults["eps"] == 0.001 and opts["optimizer_mse"].defaults["eps"] == 1e-08
    assert opts["optimizer_mse_2"].defaults["eps"] == 0.001 and opts["optimizer_2"].defaults["lr"] == c["lr_base2"]
    data = tiny_i3d.batches(c["seed_data"], c["epochs"], c["per_epoch"])
    k, phases = 0, []
    for ep in range(c["epochs"]):
        for ib, (x, t) in enumerate(data[ep]):
            _, _, phase, stepped = tr.step(x, t, ep, ib)
            phases.append((phase, stepped))
            np.testing.assert_array_equal(tiny_i3d.snapshot(net), g["params"][k], err_msg="epoch %d batch %d" % (ep, ib))
            lrs = [gr["lr"] for name in sorted(opts) for gr in opts[name].param_groups]
            np.testing.assert_array_equal(np.array(lrs), g["lrs"][k], err_msg="lr, epoch %d batch %d" % (ep, ib))
            k += 1
    torch.set_num_threads(nthreads)
    assert phases[:8] == [("D", False), ("D", True), ("G", False), ("G", True)] * 2
    assert tr.optimizer is opts["optimizer_2"] and tr.optimizer_mse is opts["optimizer_mse_2"]   # stage 2 took over


def _i3d_ddp_worker(rank, world, port, out_dir, cfg):
    from tests.golden import tiny_i3d
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = tiny_i3d.build(300 + rank)                   # different init per rank: the trainer broadcasts
    tr, _ = _make_i3d_trainer(cfg, net)
    data = tiny_i3d.batches(77, 2, 4, b=2)
    log = []
    for ep in range(2):
        for ib, (x, t) in enumerate(data[ep]):
            sh = slice(rank, rank + 1)                 # one clip of each two-clip micro-batch per rank
            tr.step(x[sh], t[sh], ep, ib)
            log.append(list(tr.exchanged))
    torch.save({"params": tiny_i3d.snapshot(net), "log": log}, os.path.join(out_dir, "i%d.pt" % rank))
    dist.destroy_process_group()


def test_i3d_trainer_two_ranks_gloo_equals_full_micro_batches(golden, tmp_path):
    """Two ranks, one clip each, against one process on the two-clip micro-batches (the losses are
    means over clips, so averaged gradients are the full-batch gradients); the exchange happens only on
    stepping micro-batches and only for the optimizers that step."""
    from tests.golden import tiny_i3d
    _, c = _i3d_cfg(golden)
    port = _free_port()
    mp.spawn(_i3d_ddp_worker, args=(2, port, str(tmp_path), c), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "i%d.pt" % r), weights_only=False) for r in (0, 1))
    np.testing.assert_array_equal(r0["params"], r1["params"])
    net = tiny_i3d.build(300)
    tr, _ = _make_i3d_trainer(c, net)
    data = tiny_i3d.batches(77, 2, 4, b=2)
    for ep in range(2):
        for ib, (x, t) in enumerate(data[ep]):
            tr.step(x, t, ep, ib)
    np.testing.assert_allclose(r0["params"], tiny_i3d.snapshot(net), rtol=2e-4, atol=2e-6)
    names = [[n for n, _ in e] for e in r0["log"]]
    assert names[0] == [] and names[1] == ["optimizer", "optimizer_3"] and names[2] == [] and names[3] == ["optimizer_mse"]


def test_geometry_plans_are_validated_on_the_host():
    """ops.check_geometry_plans (used by prepare_inputs / DevicePrep for host-side plans): a box that leaves the frame, a
    non-positive size or an output window outside the resized box raises instead of reaching the kernel."""
    from dmcnet_amd import ops
    good = torch.tensor([[16, 58, 224, 224, 224, 224, 0, 0], [0, 0, 256, 340, 256, 340, 16, 58]], dtype=torch.int32)
    ops.check_geometry_plans(good, 256, 340, 224, 224)
    for bad in ([40, 58, 224, 224, 224, 224, 0, 0],          # y0 + h > H0
                [0, 200, 224, 224, 224, 224, 0, 0],          # x0 + w > W0
                [0, 0, 224, 224, 200, 224, 0, 0],            # output taller than the resized box
                [0, 0, 224, 224, 224, 224, 0, 8],            # window leaves the resized box
                [-1, 0, 224, 224, 224, 224, 0, 0], [0, 0, 0, 224, 224, 224, 0, 0]):
        with pytest.raises(ValueError):
            ops.check_geometry_plans(torch.tensor([good[0].tolist(), bad], dtype=torch.int32), 256, 340, 224, 224)


def test_settle_host_and_side_stream_scope_on_cpu():
    """Host-side helpers of the training step that must be harmless without a GPU: train.settle_host() parks the
    long-lived objects once per process (a second call is a no-op), ops.wgrad_side_stream() only counts scopes and its
    join does nothing while no launch left the main stream."""
    import gc
    from dmcnet_amd import ops, train
    train._SETTLED[0] = False
    train.settle_host()
    frozen = gc.get_freeze_count()
    assert train._SETTLED[0] and frozen > 0
    train.settle_host()
    assert gc.get_freeze_count() == frozen
    gc.unfreeze()
    assert ops._WGRAD_SCOPE[0] == 0
    with ops.wgrad_side_stream():
        assert ops._WGRAD_SCOPE[0] == 1
        with ops.wgrad_side_stream():
            assert ops._WGRAD_SCOPE[0] == 2
    assert ops._WGRAD_SCOPE[0] == 0 and not ops._WGRAD_PENDING[0]
    ops.join_wgrad_stream()                                  # nothing pending: must not touch CUDA
