#!/usr/bin/env python3
"""Throughput of the post-decode MV / residual extraction (csrc/coviar_post.hip) on a BASELINE-size batch:
120 chains (40 clips x 3 segments) of 340 x 256 MPEG-4 frames, GOP positions 1 .. 11, vectors already resident in HBM.
HIP events on the launch stream around dmc_mv_gop_batch (owner pass + back-trace); algorithmic bytes per call:
  owner maps   4 B/px/frame cleared + 4 B/px/frame read by the trace (the atomicMax traffic stays in L2)
  MV out       8 B/px/chain          residual: + 3 (cur) + 3 (ref, gathered) + 12 (out) B/px/chain
next to the C oracle (oracle/coviar_post_ref.c, one core) on a sample of the same chains.
    python tools/coviar_post_bench.py [--chains 120] [--iters 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dmcnet_amd                                   # noqa: E402
from dmcnet_amd import _lib as L                    # noqa: E402
from dmcnet_amd.ops import _stream                  # noqa: E402
from tests import coviar_post_ref as R              # noqa: E402  (synthetic vector lists + the CPU oracle: the checker / baseline)

ap = argparse.ArgumentParser()
ap.add_argument("--chains", type=int, default=120)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cpu-chains", type=int, default=12)
args = ap.parse_args()
H, W = 256, 340
rs = np.random.RandomState(5)
samples = [R.synthetic_gop(rs, H, W, 1 + (i % 11), extra=10) for i in range(args.chains)]
lib = L.load()
dev = "cuda:0"
recs, frame_off, chain_off = [], [0], [0]
for frames in samples:
    for sd, _ in frames:
        if sd is not None:
            recs.append(np.ascontiguousarray(sd).view(np.uint8).reshape(-1))
            frame_off.append(frame_off[-1] + sd.shape[0])
    chain_off.append(len(frame_off) - 1)
n_frames, n_mv = len(frame_off) - 1, frame_off[-1]
mvs = torch.from_numpy(np.concatenate(recs)).to(dev)
d_frame = torch.tensor(frame_off, dtype=torch.int32, device=dev)
d_chain = torch.tensor(chain_off, dtype=torch.int32, device=dev)
owner = torch.empty(lib.dmc_mv_owner_bytes(n_frames, H, W) // 4, dtype=torch.int32, device=dev)
ref = torch.from_numpy(np.stack([f[0][1] for f in samples])).to(dev)
cur = torch.from_numpy(np.stack([f[-1][1] for f in samples])).to(dev)
mv_out = torch.zeros((args.chains, H, W, 2), dtype=torch.int32, device=dev)
res_out = torch.zeros((args.chains, H, W, 3), dtype=torch.int32, device=dev)


def call(residual):
    L.check(lib.dmc_mv_gop_batch(L.ptr(mvs), 40, n_mv, L.ptr(d_frame), n_frames, L.ptr(d_chain), args.chains, L._P(0), L.ptr(owner),
                                 L.ptr(ref), L.ptr(cur), L._P(0), L.ptr(mv_out), L.ptr(res_out) if residual else L._P(0), L._P(0),
                                 H, W, _stream()), "dmc_mv_gop_batch")


out = {"workload": "%d chains of 340x256 frames, %d P-frames, %d motion vectors" % (args.chains, n_frames, n_mv)}
px = H * W
for tag, residual in (("mv", False), ("mv+residual", True)):
    for _ in range(3):
        call(residual)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        call(residual)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.iters
    nbytes = px * (8 * n_frames + args.chains * (8 + (18 if residual else 0))) + n_mv * 40
    out[tag] = {"ms_per_call": round(ms, 4), "chains_per_s": round(args.chains / ms * 1e3, 1),
                "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000.0, 4)}
# CPU oracle (one core) on a sample, frame by frame as decode_video does
t0 = time.perf_counter()
for frames in samples[:args.cpu_chains]:
    R.decode_video_policy(frames, R.MV, 1, H, W)
    R.decode_video_policy(frames, R.RESIDUAL, 1, H, W)
dt = time.perf_counter() - t0
out["cpu_oracle"] = {"chains_per_s": round(args.cpu_chains / dt, 2), "cores": 1, "kind": "port", "sample": "%d of the chains, MV and residual each" % args.cpu_chains}
# the device result equals the oracle on the sampled chains
call(True)
torch.cuda.synchronize()
for i in (0, args.chains // 2, args.chains - 1):
    assert np.array_equal(mv_out[i].cpu().numpy(), R.decode_video_policy(samples[i], R.MV, 1, H, W))
    assert np.array_equal(res_out[i].cpu().numpy(), R.decode_video_policy(samples[i], R.RESIDUAL, 1, H, W))
out["checked"] = "bit-exact against the oracle on 3 chains"
print(json.dumps(out))
