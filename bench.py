#!/usr/bin/env python3
"""Headline benchmark: clips/sec of one full DMC-Net training step (DMC generator + ResNet-18 +
flow-MSE + consensus CE, backward, two Adam steps), 3 segments x 224x224 per clip, 40 clips per
GPU (weak scaling), synthetic MV/residual/flow already resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  The `roofline` object is for the
dominant hand-written kernel, the generator forward, timed with HIP events on its stream inside
the timed region; the `cpu_baseline` object is the CPU oracle (a port of the reference's CPU
path, validated against it by tests/golden) timed on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector == fp32 matrix peak
HBM_PEAK_GBS = 8000.0
GEN_FLOP_PER_PX = 9108         # 4,554 MAC, SURVEY.md 8(d)
GEN_BYTES_PER_PX = 28          # read 5 ch + write 2 ch fp32 (fused, inference-style)

HP = dict(lr=0.01, weight_decay=1e-4, lr_cls_mult=0.01, lr_mse_mult=1.0)
TRAFFIC_JSON = os.path.join("profiles", "r6_gen_traffic.json")


def measured_traffic(px):
    """HBM bytes per generator-forward call from the PMC counters: collected with rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE (separate passes, calibrated; tools/pmc_traffic.py) on the SAME kernel
    source -- the JSON records the sha256 of gen_tiny.hip + gen_x3.hip + gen_fused.hip it was measured on and the figure is
    reported only while that still matches the sources in the tree (else null: stale)."""
    import hashlib
    path = os.path.join(ROOT, TRAFFIC_JSON)
    if not os.path.exists(path):
        return None, "no PMC measurement in the tree (%s)" % TRAFFIC_JSON
    rec = json.load(open(path))
    csrc = os.path.join(ROOT, "dmc-net_amd", "csrc")
    both = b"".join(open(os.path.join(csrc, f), "rb").read() for f in ("gen_tiny.hip", "gen_x3.hip", "gen_fused.hip"))
    sha = hashlib.sha256(both).hexdigest()[:16]
    if rec.get("kernel_source_sha16") != sha:
        return None, "%s was measured on another version of gen_tiny.hip / gen_x3.hip / gen_fused.hip (stale, not reported)" % TRAFFIC_JSON
    return int(rec["gen_fwd_bytes_per_px"] * px), "%s: %s; kernel source sha16 %s" % (TRAFFIC_JSON, rec["method"], sha)


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))     # beyond ~64 threads torch's CPU convs stop scaling


CPU_TIMED_STEPS = 5          # 2 warm-ups + >= 5 timed steps, median
CPU_WARMUP_STEPS = 2


def cpu_baseline(batch, num_segments, num_class, budget_s=100.0):
    """The oracle's dmcnet train step on the host cores (kind 'port')."""
    from oracle import dmc_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    kw = dict(base_model="resnet18", use_databn=0, gen_flow_or_delta=1, arch_estimator="DenseNetTiny")
    m = O.OracleModel(num_class, num_segments, "mv", **kw).train()
    oc, og = O.make_optimizers(m, **HP)
    # bounded sample: probe with 2 clips, then size the timed batch for ~budget_s of CPU work
    probe = O.synthetic_batch(1234, 2, num_segments, num_class, flow_ds_factor=16)
    O.dmcnet_train_step(m, oc, og, probe, num_segments, 1.0, 10.0)      # warm-up (allocator, oneDNN)
    t0 = time.time()
    O.dmcnet_train_step(m, oc, og, probe, num_segments, 1.0, 10.0)
    per_clip = (time.time() - t0) / 2
    # the same B=40 batch the GPU steps on.  The budget (5 timed steps + 2 warm-ups of ~6-8 s each on the
    # GPU box's 16 cores) covers it; only a much slower host cuts the sample, and the line says so.
    b = int(max(2, min(batch, budget_s / (CPU_TIMED_STEPS + CPU_WARMUP_STEPS) / max(1.4 * per_clip, 1e-6))))   # 1.4: large batches run ~40 % slower per clip
    data = O.synthetic_batch(1234, b, num_segments, num_class, flow_ds_factor=16)
    for _ in range(CPU_WARMUP_STEPS):
        O.dmcnet_train_step(m, oc, og, data, num_segments, 1.0, 10.0)  # warm-ups at the timed size (the first two steps are still warming)
    times = []
    for _ in range(CPU_TIMED_STEPS):
        t0 = time.time()
        O.dmcnet_train_step(m, oc, og, data, num_segments, 1.0, 10.0)
        times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(b / dt, 3), "unit": "clips/sec", "cores": cores, "kind": "port",
            "sample": "oracle dmcnet train step (torch CPU fp32, %d threads), %d clips x %d segments x "
                      "224x224 per step (%s), median of %d timed "
                      "steps after %d warm-ups at that size (SURVEY 8d)" % (cores, b, num_segments,
                                                               "the full B=%d batch" % batch if b == batch else
                                                               "the B=%d workload cut to fit ~%ds of CPU work" % (batch, int(budget_s)),
                                                               CPU_TIMED_STEPS, CPU_WARMUP_STEPS),
            "step_s": [round(t, 3) for t in times]}


def bench_prepare(dev, n_frames, flow_ds_factor, iters=20):
    """Sub-line for the HBM-bound input-preparation kernel: uint8 frames as decoded
    (256x340x7) -> crop + flip + blockify + normalised fp32 planes at 224x224.  Algorithmic bytes per
    output pixel: 7 read + 28 written = 35.  Two workloads: the validation-style copy crop (no
    resampling) and the training-style random-scale crop (bilinear resize of the box)."""
    import random
    from dmcnet_amd import ops, transforms
    g = torch.Generator(device=dev).manual_seed(7)
    frames = ops.u8_frames_buffer((n_frames, 256, 340, 7), dev)
    frames.copy_(torch.randint(0, 256, frames.shape, generator=g, device=dev, dtype=torch.uint8))
    out = {}
    pipes = {"copy_crop": transforms.Compose([transforms.GroupCenterCrop(224), transforms.GroupRandomHorizontalFlip()]),
             "multiscale_crop_resize": transforms.Compose([transforms.GroupMultiScaleCrop(224, [1, .875, .75]),
                                                           transforms.GroupRandomHorizontalFlip()])}
    random.seed(11)
    for name, pipe in pipes.items():
        plans, flips = [], []
        for _ in range(n_frames):
            plan, size, flip = transforms.geometry_plan(pipe, (256, 340))
            plans.append(plan); flips.append(int(flip))
        boxes = torch.tensor(plans, dtype=torch.int32, device=dev)
        fl = torch.tensor(flips, dtype=torch.uint8, device=dev)
        for _ in range(3):
            ops.prepare_inputs(frames, fl, flow_ds_factor, boxes=boxes, out_size=(224, 224))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            ops.prepare_inputs(frames, fl, flow_ds_factor, boxes=boxes, out_size=(224, 224))
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
        px = n_frames * 224 * 224
        src = sum(p[2] * p[3] for p in plans) * 7            # bytes of the boxes actually sampled
        gbs = (src + px * 28) / (ms * 1e-3) / 1e9
        out[name] = {"ms": round(ms, 4), "GB/s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    out["note"] = ("dmc_prepare_inputs_crop, %d frames 256x340x7 uint8 -> 224x224 fp32 planes, flow_ds_factor %d; "
                   "bytes = 7 B per sampled source pixel + 28 B per output pixel; bound: hbm" % (n_frames, flow_ds_factor))
    return out


def classifier_conv_roofline(conv_spans, n_frames, conv_arith, hw=224):
    """The ResNet-18 3x3 / 1x1 convolutions (everything but conv1) as one kernel family: algorithmic FLOPs of the
    forward, data-gradient and weight-gradient GEMMs of a step (2 x MACs each) over the HIP-event time of their C-ABI
    calls, against the matrix peak of the arithmetic they run in: bf16x3 = six bf16 MFMAs per fp32 product block
    (2,500 / 6 TFLOP/s fp32-equivalent, MI355X_MICROARCH.md dense bf16 peak), fp32 MFMA = 157.3."""
    if not conv_spans:
        return None
    macs, cin, res = 0, 64, hw // 4
    for i, cout in enumerate((64, 128, 256, 512)):
        if i:
            res //= 2
            macs += res * res * cout * cin * (9 + 1)              # stride-2 3x3 + the 1x1 shortcut
            macs += 3 * res * res * cout * cout * 9
        else:
            macs += 4 * res * res * cout * cin * 9
        cin = cout
    flop = 2.0 * macs * n_frames
    out = {"unit": "TFLOP/s (fp32-equivalent)", "bound": "mfma",
           "peak": round(2500.0 / 6, 1) if conv_arith else FP32_PEAK_TFLOPS,
           "arithmetic": "bf16x3" if conv_arith else "fp32 MFMA", "gflop_per_pass": round(flop / 1e9, 1)}
    tot_ms = 0.0
    for name, key in (("forward", "conv_nhwc_fwd"), ("data_gradient", "conv_nhwc_dgrad"), ("weight_gradient", "conv_nhwc_wgrad")):
        if key in conv_spans:
            ms = conv_spans[key]                                    # median over 7 probed steps of the step's summed spans
            out[name] = {"ms_per_step": round(ms, 3), "achieved": round(flop / (ms * 1e-3) / 1e12, 1)}
            tot_ms += ms
    if tot_ms > 0:
        out["ms_per_step"] = round(tot_ms, 3)
        out["achieved"] = round(3 * flop / (tot_ms * 1e-3) / 1e12, 1)
        out["frac"] = round(out["achieved"] / out["peak"], 4)
    out["note"] = ("HIP events around the C-ABI calls (weight split / pack launches included) in 7 steps after the timed "
                   "region, median over the steps; the span names are ops.conv_nhwc_*")
    return out


def add_rank_spread(comm, world, dev, ms_per_step, exposed_ms, host_clean_ms):
    """The per-rank part of the ``comm`` object (every --config prints the same keys): step time, exposed wait and clean host
    cost of every rank, the host cores they share, the device."""
    mine = torch.tensor([ms_per_step, exposed_ms or 0.0, host_clean_ms], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    comm["ms_per_step_by_rank"] = [round(float(t[0]), 3) for t in every]
    comm["ms_per_step_rank_min"] = round(min(float(t[0]) for t in every), 3)
    comm["ms_per_step_rank_max"] = round(max(float(t[0]) for t in every), 3)
    comm["exposed_wait_ms_per_step_rank_max"] = round(max(float(t[1]) for t in every), 4)
    comm["host_clean_ms_per_step_by_rank"] = [round(float(t[2]), 3) for t in every]
    comm["host_cores_usable"] = usable_cores()
    comm["device_of_rank0"] = torch.cuda.get_device_name(dev)
    return comm


def bench_i3d(args, rank, world, dev):
    """BASELINE config 5: I3D over the per-frame DMC generator; micro-batch of 3 clips x T frames,
    trunk under bf16 autocast, generator fp32; D and G phases alternate (iter_size 1)."""
    from dmcnet_amd import i3d, i3d_train, ops
    torch.manual_seed(0)
    b = 3 if args.batch == 40 else args.batch
    net = i3d.I3D(args.num_class, modality="flow+mp4", dropout_prob=0.85, arch_estimator="DenseNetTiny",
                  arch_d="Discriminator").to(dev).train()
    net.trunk_dtype = torch.bfloat16
    # the shipped recipe's optimizers / schedulers / two-stage policy; iter_size 1: every micro-step steps
    # (and, with --gpus N, exchanges the stepping optimizers' gradients) -- the recipe's 32 would amortise both
    trainer = i3d_train.recipe_trainer(net, batch_size=b, world_size=world, iter_size=1)
    # the measurement hooks of tools/host_contention.sh (see main()): independent replicas / a stubbed collective
    stubbed = False
    if world > 1 and os.environ.get("DMC_BENCH_REPLICAS") == "1":
        trainer.world = 1                             # no gradient exchange at all
    elif world > 1 and os.environ.get("DMC_BENCH_STUB_ALLREDUCE") == "1":
        if os.environ.get("DMC_BENCH_TEST_HOOKS") != "1":
            raise SystemExit("DMC_BENCH_STUB_ALLREDUCE=1 disables the gradient exchange: refused unless DMC_BENCH_TEST_HOOKS=1 is set as well")
        stubbed = True
        i3d_train.dist.all_reduce = lambda *a, **k: None          # the trainer's exchange is blocking: nothing to wait for
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    data = torch.randn((b, 7, args.clip_length, 224, 224), generator=g, device=dev)
    target = torch.randint(0, args.num_class, (b,), generator=g, device=dev)
    counter = [0]

    def one():
        out = trainer.step(data, target, 0, counter[0])
        counter[0] += 1
        return out[0], out[1], out[2]

    graph_note = "eager (one launch per kernel)"
    if args.graph:
        # forward + losses + backward of a micro-batch replayed from a hipGraph (one per phase kind: D, G), captured during the
        # warm-up; learning-rate policy, gradient exchange and Adam stay eager.  The input buffers are the graphs' own.
        trainer.enable_graphs(warmup=1)
        for _ in range(4):
            one()
        if trainer.static_batch() is not None:
            sdata, starget = trainer.static_batch()
            sdata.copy_(data); starget.copy_(target)
            data, target = sdata, starget
            graph_note = ("hipGraph replay of forward + losses + backward per phase kind (%d graphs), policy / exchange / Adam eager; "
                          "kernels_ms from HIP events in 6 eager micro-steps run right after the timed region" % len(trainer._graphs))
    for _ in range(args.warmup):
        one()
    from dmcnet_amd import train as _train
    _train.settle_host()                             # host GC pauses out of the timed region (as the training driver does)
    probe = ops.EventProbe(None if args.all_spans else ("gen_tiny_fwd", "gen_tiny_bwd"))
    if not args.graph:
        ops.PROBE = probe
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ops.profile_mark()                               # brackets the timed region in a rocprofv3 trace (tools/rocprof_region.py)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, losses, _ = one()
    ops.profile_mark()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host time to ENQUEUE a step (~ ms_per_step: launch-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = elapsed_rank = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    if args.graph:                                   # events cannot be recorded inside a replay: eager micro-steps for the spans
        ops.PROBE = probe                            # (with a probe installed the trainer runs a micro-step eagerly)
        for _ in range(6):
            one()
        torch.cuda.synchronize()
    ops.PROBE = None
    spans = probe.summary()
    clean = []                                       # clean host cost: each micro-step enqueued on an empty queue (see main())
    for _ in range(7):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        one()
        clean.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    host_clean_ms = sorted(clean)[len(clean) // 2]
    comm = {"backend": None, "world_size": 1, "note": "single process: no gradient exchange"}
    if world > 1:
        comm = trainer.comm_summary()
        if stubbed:
            comm["allreduce"] = "stubbed"
        if trainer.world == 1:
            comm["note"] = "DMC_BENCH_REPLICAS=1: independent replicas, no gradient exchange (host-contention measurement)"
        comm = add_rank_spread(comm, world, dev, elapsed_rank / args.steps * 1e3, None, host_clean_ms)
    if rank != 0:
        dist.destroy_process_group()
        return
    px = b * args.clip_length * 224 * 224
    fwd_ms = spans["gen_tiny_fwd"][0]
    tf = px * GEN_FLOP_PER_PX / (fwd_ms * 1e-3) / 1e12
    print(json.dumps({
        "metric": "clips/sec (%d-frame 224x224 clips) dmcnet_I3D train micro-step" % args.clip_length,
        "value": round(world * b * args.steps / elapsed, 3), "unit": "clips/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "host_enqueue_ms_per_step": round(host_ms, 3), "host_clean_ms_per_step": round(host_clean_ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 trunk / f32 generator",
        "data": "synthetic",
        "config": {"workload": "dmcnet_I3D HMDB-51, DenseNetTiny generator per frame + I3D trunk + Discriminator, "
                               "micro-batch %d clips x %d frames per GPU, alternating D/G phases, the reference's 5-optimizer two-stage "
                               "policy (stage 1), iter_size 1" % (b, args.clip_length),
                   "global_batch": world * b, "parallelism": "dp%d" % world,
                   "final_losses": [round(float(l), 5) for l in losses]},
        "roofline": {"kernel": "dmc_gen_tiny_fwd (%d frames)" % (b * args.clip_length), "bound": "mfma",
                     "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / FP32_PEAK_TFLOPS, 4), "traffic": None, "launch_ms": round(fwd_ms, 4)},
        "kernels_ms": {k: round(v[0], 4) for k, v in spans.items()}, "comm": comm, "launch": graph_note}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=40, help="clips per GPU")
    ap.add_argument("--num-class", type=int, default=51)
    ap.add_argument("--config", default="dmcnet", choices=["dmcnet", "gan", "i3d"])
    ap.add_argument("--clip-length", type=int, default=64, help="frames per clip (i3d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--miopen-find", type=int, default=1,
                    help="1 (default) = cudnn.benchmark as the reference's train.py:118 sets it: MIOpen "
                         "picks solvers by search, answered from the find-db shipped in "
                         "dmc-net_amd/miopen_db; 0 = MIOpen's heuristic picks")
    ap.add_argument("--graph", type=int, default=-1,
                    help="-1 (default) = 1 for --config i3d (its ~840 launches per micro-step keep the host busy for 13-15 ms: "
                         "forward + losses + backward are replayed from a hipGraph per phase kind, I3DTrainer.enable_graphs; the "
                         "policy and Adam stay eager), 0 for the others (measured: no gain, the host needs 4.4 of 11.4 ms).  1 = capture the whole training step (forward, losses, backward, Adam) in a hipGraph after the "
                         "warm-up and time graph replays (single GPU): removes the launch gaps between the ~290 kernels "
                         "of a step.  The generator kernels' durations are then taken from eager steps run right after "
                         "the timed region (events cannot be recorded inside a replay)")
    ap.add_argument("--all-spans", action="store_true",
                    help="HIP-event spans around every C-ABI call (kernels_ms lists them all); default: the "
                         "generator forward / backward only, which the roofline object needs")
    ap.add_argument("--own-conv", type=int, default=1,
                    help="1 (default) = the classifier's 3x3 / 1x1 convolutions on this package's matrix-core NHWC "
                         "kernels (fused conv -> bn op); 0 = PyTorch-ROCm (MIOpen) convolutions")
    ap.add_argument("--conv-arith", type=int, default=1,
                    help="arithmetic of this package's NHWC convolutions (option conv_arith): 0 = fp32 MFMA, 1 = bf16x3 "
                         "(fp32 operands as three bf16 slices, six bf16 MFMAs per product block, fp32 accumulate: "
                         "fp32-level error at a multiple of the fp32 instruction's rate)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="library kernel-selection option for A/B runs (dmc_set_option), e.g. --option gen_x3=2; recorded in the "
                         "JSON line's config")
    args = ap.parse_args()
    if args.graph < 0:
        args.graph = 1 if args.config == "i3d" else 0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    # test hooks: DMC_FORCE_DEVICE / DMC_DIST_BACKEND=gloo let two ranks share one GPU so the N>1
    # code path can be exercised on a 1-GPU box (never set by the driver)
    if "DMC_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["DMC_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("DMC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import dmcnet_amd
    from dmcnet_amd import dataset, ddp, miopen, ops, resnet, train
    resnet.OWN_CONV = bool(args.own_conv)
    dmcnet_amd._lib.check(dmcnet_amd._lib.load().dmc_set_option(b"conv_arith", int(args.conv_arith)), "dmc_set_option")
    for kv in args.option:
        name, value = kv.split("=")
        dmcnet_amd._lib.check(dmcnet_amd._lib.load().dmc_set_option(name.encode(), int(value)), "dmc_set_option %s" % kv)
    if args.miopen_find:
        miopen.enable_find()          # before the first convolution of the process
    if args.config == "i3d":
        return bench_i3d(args, rank, world, dev)
    S = 3
    torch.manual_seed(0)
    gan = args.config == "gan"
    model = dmcnet_amd.Model(args.num_class, S, "mv", base_model="resnet18", use_databn=0,
                             gen_flow_or_delta=1, arch_estimator="DenseNetTiny",
                             arch_d="Discriminator3" if gan else None).to(dev).train()
    torch.backends.cudnn.benchmark = bool(args.miopen_find)   # (enable_find above also set the db path)
    if os.environ.get("DMC_CHANNELS_LAST") == "0":
        model.base_model.to(memory_format=torch.contiguous_format)
    # DMC_BENCH_REPLICAS=1 (test hook, tools/host_contention.sh): the N ranks run INDEPENDENT replicas -- no gradient exchange --
    # so that what N processes cost each other on the HOST can be measured on a box whose transport (gloo through host memory
    # when N ranks share one GPU) would otherwise dominate
    replicas = os.environ.get("DMC_BENCH_REPLICAS") == "1"
    reducer = ddp.for_model(model) if (world > 1 and not replicas) else None
    # DMC_BENCH_STUB_ALLREDUCE=1 (test hook, same tool): the reducer runs in full -- post-accumulate hooks, bucket copies,
    # side-stream joins, waits -- but the collective itself returns at once (a completed Work): what the N > 1 path costs the
    # HOST without the host-memory transport of gloo, which an RCCL run does not have.  Gradients are NOT exchanged.
    stubbed = False
    if reducer is not None and os.environ.get("DMC_BENCH_STUB_ALLREDUCE") == "1":
        if os.environ.get("DMC_BENCH_TEST_HOOKS") != "1":
            raise SystemExit("DMC_BENCH_STUB_ALLREDUCE=1 disables the gradient exchange: a measurement hook of tools/host_contention.sh, "
                             "refused unless DMC_BENCH_TEST_HOOKS=1 is set as well")
        stubbed = True

        class _DoneWork(object):
            def wait(self, *a, **k):
                return True

            def is_completed(self):
                return True
        ddp.dist.all_reduce = lambda *a, **k: _DoneWork()
    if gan:
        stepper = train.GanTrainStep(model, S, 1.0, 1.0, 0.01, 10.0, lr_d_mult=1.0, reducer=reducer, **HP)
    else:
        stepper = train.DmcnetTrainStep(model, S, 1.0, 10.0, reducer=reducer, **HP)
    batch = dataset.synthetic_batch_on_device(1234 + rank, args.batch, S, args.num_class, dev,
                                              flow_ds_factor=0 if gan else 16)

    def one(i):
        return stepper.step(batch, i) if gan else stepper.step(batch)

    for i in range(args.warmup):
        one(i)
    train.settle_host()                              # as driver.train_epoch does after its first steps (host GC pauses)
    if reducer is not None:
        reducer.time_waits = True                    # exposed communication of the timed steps (comm object below)
    probe = ops.EventProbe(None if args.all_spans else ("gen_tiny_fwd", "gen_tiny_bwd"))
    graphs = None
    if args.graph and world == 1:
        # whole-step capture: one graph per distinct step kind (GAN: the D step and the G step).  Inputs are
        # the resident synthetic batch, every workspace comes from the graph's private pool, the fused Adam
        # kernels keep their step counters on the device, learning rates are baked in (constant here).
        try:
            torch.cuda.synchronize()
            graphs, gouts = [], []
            for kind in range(2 if gan else 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    gouts.append(one(kind))
                graphs.append(g)
            torch.cuda.synchronize()
        except Exception as e:                       # capture is an optimisation, never a requirement
            sys.stderr.write("bench: hipGraph capture failed (%s: %s); timing eager steps\n" % (type(e).__name__, e))
            graphs = None
    if graphs is None:
        ops.PROBE = probe
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.profile_mark()
    # one event per step boundary (recorded on the launch stream, read after the region): per-step
    # device times for the median; the headline stays the wall time of the whole region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        if graphs is None:
            out = one(i)
        else:
            graphs[i % len(graphs)].replay()
            out = gouts[i % len(graphs)]
        marks[i + 1].record()
    ops.profile_mark()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host time to ENQUEUE a step (~ ms_per_step: launch-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if graphs is not None:                           # kernel durations: eager steps right after the timed replays
        ops.PROBE = probe
        for i in range(6):
            one(i)
        torch.cuda.synchronize()
    ops.PROBE = None
    conv_spans = None
    if args.own_conv and not gan and world == 1:
        # classifier convolutions (the step's dominant kernel family): HIP-event spans around their C-ABI calls in 7
        # extra steps right AFTER the timed region (57 spans per step would cost the timed steps ~2 %)
        per_step = []
        for i in range(7):
            cprobe = ops.EventProbe(("conv_nhwc_fwd", "conv_nhwc_dgrad", "conv_nhwc_wgrad"))
            ops.PROBE = cprobe
            one(i)
            per_step.append({k: v[0] * v[1] for k, v in cprobe.summary().items()})    # ms of this step's calls
        ops.PROBE = None
        # median over the steps (a span also contains whatever the host did between its two event records: one stalled
        # step must not decide the figure)
        conv_spans = {k: sorted(st[k] for st in per_step if k in st)[len(per_step) // 2] for k in per_step[0]}
    # CLEAN host cost of a step: each step enqueued on an EMPTY launch queue (synchronize() in front, outside the clock), so
    # the figure holds no queue back-pressure -- host_enqueue_ms_per_step above is read while the queue is full and mostly
    # measures the GPU.  With N ranks every rank does this at the same time (a barrier in front): the ranks contend for the
    # host's cores exactly as in the timed region.
    clean = []
    for i in range(9):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        one(i)
        clean.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize()
    host_clean_ms = sorted(clean)[len(clean) // 2]
    comm = {"backend": None, "world_size": 1, "note": "single process: no gradient exchange"}
    if world > 1:
        # self-diagnosing multi-GPU line: which communicator, what travelled, how much of it was exposed, rank spread
        comm = reducer.comm_summary() if reducer is not None else {
            "backend": dist.get_backend(), "world_size": world, "exposed_wait_ms_per_step": None,
            "note": "DMC_BENCH_REPLICAS=1: independent replicas, no gradient exchange (host-contention measurement)"}
        if stubbed:
            comm["allreduce"] = "stubbed"          # DMC_BENCH_STUB_ALLREDUCE: NO gradients were exchanged in this run
        add_rank_spread(comm, world, dev, elapsed / args.steps * 1e3, comm["exposed_wait_ms_per_step"], host_clean_ms)
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(out["loss"])
    spans = probe.summary()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if os.environ.get("DMC_BENCH_DUMP_STEPS"):      # diagnosis: device time of every timed step, in order
        sys.stderr.write("bench: step ms " + " ".join("%.2f" % marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)) + "\n")
    if gan and args.steps > 1:      # D and G steps alternate: a "step" is their pair average
        pair = [marks[i].elapsed_time(marks[i + 2]) / 2 for i in range(0, args.steps - 1, 2)]
        step_ms = sorted(pair)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])

    if rank == 0:
        n_frames = args.batch * S
        px = n_frames * 224 * 224
        fwd_ms, _ = spans["gen_tiny_fwd"]
        tf = px * GEN_FLOP_PER_PX / (fwd_ms * 1e-3) / 1e12
        gbs = px * GEN_BYTES_PER_PX / (fwd_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(px)
        line = {
            "metric": "clips/sec (3-seg 224x224) DMC-gen+ResNet-18 train step",
            "value": round(world * args.batch * args.steps / elapsed, 3), "unit": "clips/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_median": round(median_ms, 3), "host_enqueue_ms_per_step": round(host_ms, 3),
            "host_clean_ms_per_step": round(host_clean_ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("dmcnet_GAN HMDB-51 split1 (Discriminator3, alternating D/G)" if gan else
                                    "HMDB-51 split1 dmcnet (no GAN), 3 segments, ResNet-18, DenseNetTiny "
                                    "generator, delta mode, MSE x10") +
                                   ", batch %d clips/GPU, random-init weights" % args.batch,
                       "global_batch": world * args.batch, "num_class": args.num_class,
                       "parallelism": "dp%d" % world + (" (all-reduce STUBBED: gradients not exchanged)" if stubbed else ""),
                       "final_loss": round(loss, 6),
                       **({"options": args.option} if args.option else {}),
                       "generator_kernels": (("libdmcnet_hip gen_fused: the forward as ONE launch (fp32 4x4x1 MFMA, features line-buffered in LDS, "
                                              "layers pipelined across waves); " if dmcnet_amd._lib.load().dmc_get_option(b"gen_fused") else
                                              "libdmcnet_hip gen_tiny / gen_x3 layer-by-layer forward (gen_x3 mask %d); " % dmcnet_amd._lib.load().dmc_get_option(b"gen_x3")) +
                                             "backward: gen_tiny ring kernels, Winograd F(2x2,3x3) mask %#x (bit 8 + K: data-gradient group K; fp32)"
                                             % dmcnet_amd._lib.load().dmc_get_option(b"gen_wino")),
                       "classifier_convs": (("libdmcnet_hip conv_x3s (3x3 stride 1) + conv_x3q (the stride-2 blocks: 3x3 stride 2 fused with "
                                             "the 1x1 shortcut on space-to-depth slice tensors): every operand pre-split into "
                                             "bf16x3 slice tensors by its producer, " if ops.X3Q else
                                             "libdmcnet_hip conv_x3s (3x3 stride 1: operands pre-split into bf16x3 slice tensors by "
                                             "their producers; stride-2 data gradient likewise) + conv_nhwc (stride-2 / 1x1), ")
                                            if (args.conv_arith and ops.X3S) else "libdmcnet_hip conv_nhwc, ") +
                                           ("bf16x3 arithmetic (fp32 values; every fp32 "
                                            "product formed from three bf16 slices by six bf16 MFMAs, fp32 accumulate: "
                                            "error vs fp64 <= the fp32 MFMA's, tools/conv_x3_check.py)"
                                            if args.conv_arith else "fp32 MFMA") if args.own_conv
                                           else "PyTorch-ROCm (MIOpen fp32, NHWC, solver search)"},
            "roofline": {
                "kernel": "dmc_gen_tiny_fwd (EstimatorDenseNetTiny forward, %d frames)" % n_frames,
                # the fused fp32 generator is FMA-bound (325 FLOP/B >> ridge 20 FLOP/B): the binding
                # roof is the fp32 vector/matrix peak, reported in the "mfma" slot of the schema
                "bound": "mfma", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                "launch_ms": round(fwd_ms, 4),
                "hbm": {"achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 5),
                        "note": "algorithmic 28 B/px; 'traffic' = measured HBM bytes of the forward's launches",
                        "traffic_source": traffic_src,
                        "traffic_gbs": None if traffic is None else round(traffic / (fwd_ms * 1e-3) / 1e9, 1)}},
            "kernels_ms": {k: round(v[0], 4) for k, v in spans.items()},
            "roofline_classifier_convs": classifier_conv_roofline(conv_spans, n_frames, args.conv_arith),
            "comm": comm,
            "launch": ("hipGraph replay of the captured step; roofline.launch_ms from HIP events around the same "
                       "C-ABI call in 6 eager steps run right after the timed region") if graphs is not None else
                      "eager (one launch per kernel); roofline.launch_ms from HIP events inside the timed region",
        }
        if world == 1:
            line["prepare_inputs"] = bench_prepare(dev, n_frames, 0 if gan else 16)
        if world == 1 and not args.no_cpu_baseline and not gan:
            line["cpu_baseline"] = cpu_baseline(args.batch, S, args.num_class)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
