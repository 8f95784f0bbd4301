"""Post-decode half of the reference's ``coviar.load``: motion-vector rasterisation, accumulation back to the I-frame
and the motion-compensated residual, on the device.

Reference: code/dmcnet/data_loader/coviar_data_loader.c -- ``create_and_load_mv_residual`` :71-177 and the per-frame
policy of ``decode_video`` :273-375 (which frames are kept, when the function is called).  The bitstream decode is
FFmpeg's and is not here: the caller hands over, per decoded frame of ONE GOP up to the target position, the frame's
``AV_FRAME_DATA_MOTION_VECTORS`` side data (``None`` when FFmpeg exports none, e.g. the I-frame) and its BGR picture.

``load_post_decode`` mirrors ``coviar.load(video, gop, pos, representation, accumulate)``'s return value for one sample;
``extract_batch`` does a batch of samples with two kernel launches (``dmc_mv_gop_batch``).  No CPU path.
"""
import numpy as np
import torch

from . import _lib
from .ops import _stream

MV, RESIDUAL = 1, 2       # the loader's representation codes (coviar_data_loader.c:21-23)

#: AVMotionVector, libavutil/motion_vector.h (public ABI).  sizeof 40 since libavutil 55.63, 24 before.
AVMV_DTYPE = np.dtype({"names": ["source", "w", "h", "src_x", "src_y", "dst_x", "dst_y", "flags", "motion_x", "motion_y", "motion_scale"],
                       "formats": ["<i4", "u1", "u1", "<i2", "<i2", "<i2", "<i2", "<u8", "<i4", "<i4", "<u2"],
                       "offsets": [0, 4, 5, 6, 8, 10, 12, 16, 24, 28, 32], "itemsize": 40})
AVMV_DTYPE_OLD = np.dtype({"names": ["source", "w", "h", "src_x", "src_y", "dst_x", "dst_y", "flags"],
                           "formats": ["<i4", "u1", "u1", "<i2", "<i2", "<i2", "<i2", "<u8"],
                           "offsets": [0, 4, 5, 6, 8, 10, 12, 16], "itemsize": 24})


def _chain_of(frames, accumulate):
    """Which calls of create_and_load_mv_residual decode_video makes for this sample (:363-364: side data present and
    (accumulate or cur_pos == pos_target)), and whether the output steps behind `cur_pos > 0` (:128) run."""
    pos_target = len(frames) - 1
    calls = [sd for cur_pos, (sd, _bgr) in enumerate(frames)
             if sd is not None and (accumulate or cur_pos == pos_target)]
    target_called = frames[pos_target][0] is not None
    return calls, target_called, pos_target


def extract_batch(samples, representation, accumulate, device="cuda", size=None):
    """samples: list of per-sample frame lists ``[(side_data or None, bgr uint8 [H,W,3] or None), ...]`` (cur_pos 0 ..
    pos_target, same H x W for the whole batch; ``size=(H, W)`` when no picture is passed, as MV extraction needs none).  Returns an int32 CUDA tensor [len(samples), H, W, 2] (MV) or
    [len(samples), H, W, 3] (RESIDUAL) -- per sample what ``coviar.load`` returns."""
    if representation not in (MV, RESIDUAL):
        raise ValueError("representation must be MV (1) or RESIDUAL (2)")
    lib = _lib.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.DmcHipError("coviar_post runs on the HIP extension only (no CPU fallback)")
    n = len(samples)
    hw = None if size is None else (int(size[0]), int(size[1]))
    recs, frame_off, chain_off, emit, stride = [], [0], [0], [], None
    for frames in samples:
        calls, target_called, pos_target = _chain_of(frames, accumulate)
        for sd in calls:
            sd = np.ascontiguousarray(sd)
            if stride is None:
                stride = sd.dtype.itemsize
            elif sd.dtype.itemsize != stride:
                raise ValueError("mixed AVMotionVector sizes in one batch")
            recs.append(sd.view(np.uint8).reshape(-1))
            frame_off.append(frame_off[-1] + sd.shape[0])
        chain_off.append(len(frame_off) - 1)
        gate = target_called and pos_target > 0
        if representation == MV and not accumulate:
            gate = target_called                       # :111-113 sits in front of `if (cur_pos > 0)`
        emit.append(1 if gate else 0)
        for _sd, bgr in frames:
            if bgr is not None:
                hw = bgr.shape[:2] if hw is None else hw
                if tuple(bgr.shape[:2]) != tuple(hw):
                    raise ValueError("frames of different sizes in one batch")
    if hw is None:
        raise ValueError("no BGR frame given: the frame size is unknown")
    H, W = int(hw[0]), int(hw[1])
    stride = 40 if stride is None else stride
    n_mv, n_frames = frame_off[-1], len(frame_off) - 1
    mvs = torch.from_numpy(np.concatenate(recs) if recs else np.zeros(0, np.uint8)).to(dev)
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
    d_frame, d_chain, d_emit = i32(frame_off), i32(chain_off), i32(emit)
    owner = torch.empty(max(1, lib.dmc_mv_owner_bytes(n_frames, H, W) // 4), dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    ch = 2 if representation == MV else 3
    out = torch.zeros((n, H, W, ch), dtype=torch.int32, device=dev)     # PyArray_ZEROS, :292-309
    ref = cur = None
    if representation == RESIDUAL:
        ref_np = np.zeros((n, H, W, 3), np.uint8)
        cur_np = np.zeros((n, H, W, 3), np.uint8)
        for i, frames in enumerate(samples):
            pos_target = len(frames) - 1
            ref_pos = 0 if accumulate else pos_target - 1              # :346-351
            if pos_target > 0 and frames[ref_pos][1] is not None:
                ref_np[i] = frames[ref_pos][1]
            if frames[pos_target][1] is not None:
                cur_np[i] = frames[pos_target][1]
        ref, cur = torch.from_numpy(ref_np).to(dev), torch.from_numpy(cur_np).to(dev)
    _lib.check(lib.dmc_mv_gop_batch(_lib.ptr(mvs), stride, n_mv, _lib.ptr(d_frame), n_frames, _lib.ptr(d_chain), n,
                                    _lib.ptr(d_emit), _lib.ptr(owner), _lib.ptr(ref), _lib.ptr(cur), _lib._P(0),
                                    _lib.ptr(out) if representation == MV else _lib._P(0),
                                    _lib.ptr(out) if representation == RESIDUAL else _lib._P(0),
                                    _lib.ptr(bad), H, W, _stream()), "dmc_mv_gop_batch")
    if int(bad.item()) != 0:
        raise AssertionError("mv->source == -1 violated for %d vectors (coviar_data_loader.c:86)" % int(bad.item()))
    return out


def load_post_decode(frames, representation, accumulate, device="cuda", size=None):
    """One sample; see ``extract_batch``.  Returns int32 [H,W,2] (MV) or [H,W,3] (RESIDUAL) on the device."""
    return extract_batch([frames], representation, accumulate, device, size)[0]
