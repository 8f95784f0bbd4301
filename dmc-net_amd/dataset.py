"""Data side of the hot path: the integer segment-index sampling of the reference's
``CoviarDataSet`` (bit-exact), its tensor contract, and synthetic clips with that contract.

Reference: code/dmcnet/dataset.py (paths relative to the reference root) --
``get_seg_range`` :46-60, ``get_gop_pos`` :63-73, ``_get_train_frame_index`` :130-137,
``_get_test_frame_index`` :139-149, flow frame number :178, blockified flow :229-246,
normalisation :251-263, returned tuple :278.  Decoding (``coviar`` = FFmpeg MPEG-4) and the
TV-L1 flow JPEGs are outside the path: ``CoviarDataSet`` keeps the constructor and the sampling
and needs ``coviar`` only when a real video is read.
"""
import os
import random

import numpy as np
import torch
import torch.utils.data as data

GOP_SIZE = 12
_MOTION = ("residual", "mv", "flow")
_STD = (0.229, 0.224, 0.225)
STD_MEAN = float(torch.tensor(_STD, dtype=torch.float32).mean())   # what torch.mean(_input_std) gives


def get_seg_range(n, num_segments, seg, representation):
    """[begin, end) of segment ``seg``; P-frame representations skip frame 0 (an I-frame)."""
    shift = 1 if representation in _MOTION else 0
    n -= shift
    seg_size = float(n - 1) / num_segments
    begin = int(np.round(seg_size * seg))            # np.round: half to even, on float64
    end = int(np.round(seg_size * (seg + 1)))
    if end == begin:
        end = begin + 1
    return begin + shift, end + shift


def get_gop_pos(frame_idx, representation, gop_size=None):
    gop = GOP_SIZE if gop_size is None else gop_size
    index, pos = divmod(frame_idx, gop)
    if representation in _MOTION:
        if pos == 0:                                  # borrow the previous GOP's last P-frame
            index, pos = index - 1, gop - 1
    else:
        pos = 0
    return index, pos


def train_frame_index(num_frames, seg, num_segments, representation, rng=random):
    begin, end = get_seg_range(num_frames, num_segments, seg, representation)
    return get_gop_pos(rng.randint(begin, end - 1), representation)


def test_frame_index(num_frames, seg, num_segments, representation):
    shift = 1 if representation in _MOTION else 0
    num_frames -= shift
    v = int(np.round(float(num_frames - 1) / num_segments * (seg + 0.5))) + shift
    return get_gop_pos(v, representation)


test_frame_index.__test__ = False


def flow_frame_number(gop_index, gop_pos, gop_size=None):
    return gop_index * (GOP_SIZE if gop_size is None else gop_size) + gop_pos + 1


def blockify(flow, factor, upsample_interp=False):
    """Mean over factor x factor blocks (ragged edges zero-padded, as skimage's block_reduce), brought back to the
    input size by repetition or, ``upsample_interp``, by the reference's linear interpolation along one axis after the
    other (code/dmcnet/dataset.py:236-246: ``scipy.interpolate.interp1d`` between ``linspace(0, 1, n_blocks)`` and
    ``linspace(0, 1, n_blocks * factor)`` -- the block means sit at the END points of the axis, not at block centres; kept
    as written).  ``flow`` [..., H, W] numpy."""
    h, w = flow.shape[-2:]
    ph, pw = (-h) % factor, (-w) % factor
    pad = [(0, 0)] * (flow.ndim - 2) + [(0, ph), (0, pw)]
    x = np.pad(flow.astype(np.float64), pad)
    lead = x.shape[:-2]
    x = x.reshape(lead + ((h + ph) // factor, factor, (w + pw) // factor, factor)).mean(axis=(-3, -1))
    if not upsample_interp:
        return x.repeat(factor, axis=-2).repeat(factor, axis=-1)[..., :h, :w]
    from scipy import interpolate
    for axis in (-2, -1):
        n = x.shape[axis]
        x = interpolate.interp1d(np.linspace(0, 1, n), x, kind="linear", axis=axis)(np.linspace(0, 1, n * factor))
    return x[..., :h, :w]


def to_tensors(frames, flow_ds_factor=0, upsample_interp=False):
    """``frames`` [S,7,H,W] (uint8 or int) = [flow2, mv2, res3] -> the reference's
    (input_flow, input_mv, input_residual) fp32 tensors for representation 'mv'."""
    flow, mv, res = frames[:, 0:2], frames[:, 2:4], frames[:, 4:]
    if flow_ds_factor != 0:
        flow = blockify(flow, flow_ds_factor, upsample_interp)
    std = torch.tensor(_STD, dtype=torch.float32).reshape(1, 3, 1, 1)
    f = torch.from_numpy(np.ascontiguousarray(flow)).float() / 255.0
    m = torch.from_numpy(np.ascontiguousarray(mv)).float() / 255.0
    r = torch.from_numpy(np.ascontiguousarray(res)).float() / 255.0
    return (f - 0.5) / torch.mean(std), (m - 0.5) / torch.mean(std), (r - 0.5) / std


def synthetic_clip_u8(rs, num_segments, size=224):
    """One clip of uint8 frames [S,7,H,W]: MV constant per 16x16 macroblock (sigma 6 px on the
    +-20 -> +-127.5 scale), residual sigma 12, flow sigma 10 around 128."""
    mb = (size + 15) // 16
    mv = np.clip(128 + np.round(rs.normal(0, 6, (num_segments, 2, mb, mb)) * 127.5 / 20), 0, 255)
    mv = mv.repeat(16, axis=2).repeat(16, axis=3)[..., :size, :size]
    res = np.clip(128 + np.round(rs.normal(0, 12, (num_segments, 3, size, size))), 0, 255)
    flow = np.clip(128 + np.round(rs.normal(0, 10, (num_segments, 2, size, size))), 0, 255)
    return np.concatenate((flow, mv, res), axis=1).astype(np.uint8)


class SyntheticCoviarDataSet(data.Dataset):
    """Same item contract as ``CoviarDataSet`` (input_flow, input_mv, input_residual, label) on
    seeded synthetic frames; no files."""

    def __init__(self, length, num_class, num_segments=3, flow_ds_factor=0, size=224, seed=1234):
        self.length, self.num_class, self.num_segments = length, num_class, num_segments
        self.flow_ds_factor, self.size, self.seed = flow_ds_factor, size, seed

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        rs = np.random.RandomState(self.seed + index)
        frames = synthetic_clip_u8(rs, self.num_segments, self.size)
        flow, mv, res = to_tensors(frames, self.flow_ds_factor)
        return flow, mv, res, int(rs.randint(0, self.num_class))

    def raw_item(self, index):
        """uint8 frames for the GPU-side preparation (same contract as ``CoviarDataSet.raw_item``)."""
        rs = np.random.RandomState(self.seed + index)
        frames = synthetic_clip_u8(rs, self.num_segments, self.size)           # [S,7,H,W]
        label = int(rs.randint(0, self.num_class))
        hwc = np.ascontiguousarray(frames.transpose(0, 2, 3, 1))
        sz = self.size
        return hwc, np.asarray([0, 0, sz, sz, sz, sz, 0, 0], np.int32), (sz, sz), False, label


def synthetic_batch_on_device(seed, batch, num_segments, num_class, device, size=224,
                              flow_ds_factor=0):
    """A batch with the dataset's value distribution, generated directly in HBM (bench input)."""
    g = torch.Generator(device=device).manual_seed(seed)
    mb = (size + 15) // 16

    def u8(sigma, shape, scale=1.0):
        x = torch.randn(shape, generator=g, device=device) * sigma * scale
        return torch.clamp(128 + torch.round(x), 0, 255)

    mv = u8(6, (batch, num_segments, 2, mb, mb), 127.5 / 20)
    mv = mv.repeat_interleave(16, -2).repeat_interleave(16, -1)[..., :size, :size]
    res = u8(12, (batch, num_segments, 3, size, size))
    flow = u8(10, (batch, num_segments, 2, size, size))
    if flow_ds_factor:
        f = flow_ds_factor
        flow = torch.nn.functional.avg_pool2d(flow.flatten(0, 1), f).repeat_interleave(f, -2) \
            .repeat_interleave(f, -1).reshape(batch, num_segments, 2, size, size)
    std = torch.tensor(_STD, device=device).reshape(1, 1, 3, 1, 1)
    flow = ((flow / 255.0 - 0.5) / STD_MEAN).contiguous()
    mv = ((mv / 255.0 - 0.5) / STD_MEAN).contiguous()
    res = ((res / 255.0 - 0.5) / std).contiguous()
    target = torch.randint(0, num_class, (batch,), generator=g, device=device)
    return flow, mv, res, target


class CoviarDataSet(data.Dataset):
    """Constructor and sampling of the reference's dataset (code/dmcnet/dataset.py:76-281).
    Reading real videos needs the ``coviar`` extension and PIL, imported on first use."""

    def __init__(self, data_root, flow_root, data_name, video_list, representation, new_length,
                 flow_ds_factor, upsample_interp, transform, num_segments, is_train, accumulate,
                 gop, mv_minmaxnorm=0, viz=False, flow_folder="tvl1"):
        global GOP_SIZE
        GOP_SIZE = gop
        self._data_root, self._flow_root, self._data_name = data_root, flow_root, data_name
        self._representation, self._new_length = representation, new_length
        self._flow_ds_factor, self._upsample_interp = flow_ds_factor, upsample_interp
        self._transform, self._num_segments, self._is_train = transform, num_segments, is_train
        self._accumulate, self._mv_minmaxnorm, self._viz = accumulate, mv_minmaxnorm, viz
        self._flow_folder = flow_folder
        self._video_list = []
        self._load_list(video_list)

    @staticmethod
    def _flow_dir(flow_root, video_path):
        parts = video_path.split("/")
        return os.path.join(flow_root, parts[-2], parts[-1][:-4])

    def _load_list(self, video_list):
        from coviar import get_num_frames
        with open(video_list, "r") as f:
            for line in f:
                video, _, label = line.strip().split()
                path = os.path.join(self._data_root, video[:-4] + ".mp4")
                n_flow = len(os.listdir(self._flow_dir(self._flow_root, path))) / 3
                self._video_list.append((path, int(label), min(get_num_frames(path), n_flow)))

    def _get_train_frame_index(self, num_frames, seg):
        return train_frame_index(num_frames, seg, self._num_segments, self._representation)

    def _get_test_frame_index(self, num_frames, seg):
        return test_frame_index(num_frames, seg, self._num_segments, self._representation)

    def __len__(self):
        return len(self._video_list)

    def _load_frames(self, index):
        """The decoded side of ``__getitem__`` (code/dmcnet/dataset.py:151-213): S uint8 HWC frames
        [flow_x flow_y mv_x mv_y r g b] of the sampled positions and the label."""
        from coviar import load
        from PIL import Image
        if self._representation != "mv":
            raise NotImplementedError("the DMC-Net recipes use representation 'mv'")
        path, label, num_frames = random.choice(self._video_list) if self._is_train \
            else self._video_list[index]
        flow_dir = self._flow_dir(self._flow_root, path)
        frames = []
        for seg in range(self._num_segments):
            pick = self._get_train_frame_index if self._is_train else self._get_test_frame_index
            gop_index, gop_pos = pick(num_frames, seg)
            idx = flow_frame_number(gop_index, gop_pos)
            flow = np.stack([np.array(Image.open(os.path.join(
                flow_dir, "flow_%s_%05d.jpg" % (axis, idx))).convert("L")) for axis in "xy"], -1)
            mv = load(path, gop_index, gop_pos, 1, self._accumulate)
            if mv is None:
                mv = np.zeros((256, 256, 2))
            else:
                mv = mv.astype(np.float64)
                if self._mv_minmaxnorm == 1:
                    mv *= 127.5 / 20
                mv = np.clip(mv + 128, 0, 255).astype(np.uint8)
            res = np.clip(load(path, gop_index, gop_pos, 2, self._accumulate) + 128, 0, 255)
            frames.append(np.concatenate((flow, mv, res.astype(np.uint8)), axis=2))
        return frames, label

    def __getitem__(self, index):
        frames, label = self._load_frames(index)
        frames = np.transpose(np.array(self._transform(frames)), (0, 3, 1, 2))
        flow, mv, res = to_tensors(frames, self._flow_ds_factor, bool(self._upsample_interp))
        return flow, mv, res, label

    def raw_item(self, index):
        """The same sample for the GPU-side preparation (``ops.prepare_inputs`` / ``DevicePrep``):
        ``(frames uint8 [S,H0,W0,7] as decoded, plan int32 [8], out_size (h, w), flip bool, label)``
        (plan: transforms.geometry_plan).
        Consumes the RNG exactly as ``__getitem__`` does; 7 B/px cross PCIe instead of 28."""
        from . import transforms as T
        frames, label = self._load_frames(index)
        box, out, flip = T.geometry_plan(self._transform, frames[0].shape)
        return np.ascontiguousarray(np.stack(frames)).astype(np.uint8), np.asarray(box, np.int32), out, flip, label


class RawView(data.Dataset):
    """``ds.raw_item`` as a Dataset (for a DataLoader whose batches go to :class:`DevicePrep`)."""

    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, index):
        return self.ds.raw_item(index)


def collate_raw(items):
    """Batch of raw items -> (frames uint8 [B,S,H0,W0,7], plans int32 [B,8], out_size, flips uint8 [B],
    labels int64 [B]); frames of one batch must share (H0, W0) and out_size."""
    frames = torch.from_numpy(np.stack([it[0] for it in items]))
    boxes = torch.from_numpy(np.stack([it[1] for it in items]))
    outs = {tuple(it[2]) for it in items}
    if len(outs) != 1:
        raise ValueError("collate_raw: mixed output sizes %s" % sorted(outs))
    flips = torch.tensor([1 if it[3] else 0 for it in items], dtype=torch.uint8)
    labels = torch.tensor([int(it[4]) for it in items], dtype=torch.int64)
    return frames, boxes, outs.pop(), flips, labels


class DevicePrep(object):
    """Raw batch (``collate_raw``) -> the (input_flow, input_mv, input_residual, target) batch of the
    reference's loader, computed on the GPU by ``dmc_prepare_inputs_crop``: crop, bilinear resize,
    flip with x negation, 16x16 flow blockify, /255 and normalisation
    (code/dmcnet/transforms.py:36-139, code/dmcnet/dataset.py:215-263)."""

    def __init__(self, device, flow_ds_factor=0):
        self.device, self.flow_ds_factor = torch.device(device), flow_ds_factor

    def __call__(self, raw):
        from . import ops
        frames, boxes, out, flips, labels = raw
        b, s = frames.shape[:2]
        fr = ops.u8_frames_buffer((b * s,) + tuple(frames.shape[2:]), self.device)
        fr.copy_(frames.flatten(0, 1), non_blocking=True)
        if not boxes.is_cuda:        # host plans: refuse boxes that leave the frame (the kernel cannot truncate like numpy slicing)
            ops.check_geometry_plans(boxes, int(frames.shape[2]), int(frames.shape[3]), int(out[0]), int(out[1]))
        bx = boxes.to(self.device, non_blocking=True).repeat_interleave(s, 0)
        fl = flips.to(self.device, non_blocking=True).repeat_interleave(s, 0)
        flow, mv, res = ops.prepare_inputs(fr, fl, self.flow_ds_factor, boxes=bx, out_size=out)
        shp = lambda t: t.reshape((b, s) + tuple(t.shape[1:]))
        return shp(flow), shp(mv), shp(res), labels.to(self.device, non_blocking=True)
