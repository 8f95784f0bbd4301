"""CPU-side group augmentations (upstream of the hot path; kept so that
``Model.get_augmentation()`` and the train/test drivers have the callables the reference
provides in code/dmcnet/transforms.py).  Frames are HWC arrays with channels
[flow_x, flow_y, mv_x, mv_y, res_r, res_g, res_b]; numpy only -- OpenCV is not in this image,
so bilinear resizing (cv2.INTER_LINEAR's half-pixel-centre convention) is done here.
RNG call order per sample matches the reference: choice(pairs), randint, randint, random().
"""
import random

import numpy as np


class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def resize_bilinear(img, out_h, out_w):
    """HWC -> out_h x out_w x C, sample positions (i + 0.5) * scale - 0.5, edge clamped."""
    h, w = img.shape[:2]
    if (h, w) == (out_h, out_w):
        return img
    src = img.astype(np.float32)

    def axis(n_out, n_in):
        pos = (np.arange(n_out, dtype=np.float32) + 0.5) * (n_in / float(n_out)) - 0.5
        lo = np.floor(pos).astype(np.int64)
        frac = pos - lo
        return np.clip(lo, 0, n_in - 1), np.clip(lo + 1, 0, n_in - 1), frac

    y0, y1, fy = axis(out_h, h)
    x0, x1, fx = axis(out_w, w)
    top = src[y0][:, x0] * (1 - fx)[None, :, None] + src[y0][:, x1] * fx[None, :, None]
    bot = src[y1][:, x0] * (1 - fx)[None, :, None] + src[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    if np.issubdtype(img.dtype, np.integer):
        out = np.clip(np.rint(out), np.iinfo(img.dtype).min, np.iinfo(img.dtype).max)
    return out.astype(img.dtype)


def flip_with_x_negation(img):
    """Mirror horizontally; the x components of flow (ch 0) and MV (ch 2), stored around 128,
    change sign (code/dmcnet/transforms.py:47-58)."""
    out = img[:, ::-1, :].astype(np.int32)
    for ch in (0, 2):
        out[..., ch] = 256 - out[..., ch]      # 128 - (v - 128)
    return out


class GroupCenterCrop(object):
    def __init__(self, size):
        self.size = size

    def __call__(self, group):
        h, w = group[0].shape[:2]
        top, left = (h - self.size) // 2, (w - self.size) // 2
        return [im[top:top + self.size, left:left + self.size] for im in group]


class GroupScale(object):
    def __init__(self, size):
        self.size = size

    def __call__(self, group):
        return [resize_bilinear(im, self.size, self.size) for im in group]


class GroupOverSample(object):
    """The 10-crop test-time transform (code/dmcnet/transforms.py:77-114, used by code/dmcnet/test.py:96-99 for
    ``--test-crops 10``): [optional GroupScale,] then for each of the five fixed offsets (four corners + centre,
    ``fill_fix_offset(False, ...)``) and each frame: the crop, then its mirror image with the x components of flow and MV
    negated around 128 -- 10 x len(group) frames, offset-major, crop before flip."""

    def __init__(self, crop_size, scale_size=None):
        self.crop_size = (crop_size, crop_size) if isinstance(crop_size, int) else crop_size
        self.scale_worker = GroupScale(scale_size) if scale_size is not None else None

    def __call__(self, group):
        if self.scale_worker is not None:
            group = self.scale_worker(group)
        a, b = group[0].shape[:2]
        ca, cb = self.crop_size
        out = []
        for oa, ob in GroupMultiScaleCrop.fill_fix_offset(False, a, b, ca, cb):
            for im in group:
                crop = im[oa:oa + ca, ob:ob + cb]
                out.append(crop)
                out.append(flip_with_x_negation(crop))
        return out


class GroupRandomHorizontalFlip(object):
    def __call__(self, group, is_mv_or_flow=False):
        if random.random() < 0.5:
            return [flip_with_x_negation(im) for im in group]
        return group


class GroupMultiScaleCrop(object):
    def __init__(self, input_size, scales=None, max_distort=1, fix_crop=False, more_fix_crop=True):
        self.scales = scales if scales is not None else [1, .875, .75, .66]
        self.max_distort = max_distort
        self.fix_crop = fix_crop
        self.more_fix_crop = more_fix_crop
        self.input_size = [input_size, input_size] if isinstance(input_size, int) else input_size

    def __call__(self, group):
        cw, ch, ow, oh = self._sample_crop_size(group[0].shape)
        return [resize_bilinear(im[ow:ow + cw, oh:oh + ch], self.input_size[0], self.input_size[1])
                for im in group]

    def _sample_crop_size(self, im_size):
        a, b = im_size[0], im_size[1]
        sizes = [int(min(a, b) * s) for s in self.scales]
        snap = lambda v, t: t if abs(v - t) < 3 else v
        hs = [snap(v, self.input_size[1]) for v in sizes]
        ws = [snap(v, self.input_size[0]) for v in sizes]
        pairs = [(w, h) for i, h in enumerate(hs) for j, w in enumerate(ws)
                 if abs(i - j) <= self.max_distort]
        w, h = random.choice(pairs)
        if self.fix_crop:
            ow, oh = random.choice(self.fill_fix_offset(self.more_fix_crop, a, b, w, h))
        else:
            ow = random.randint(0, a - w)
            oh = random.randint(0, b - h)
        return w, h, ow, oh

    @staticmethod
    def fill_fix_offset(more_fix_crop, image_w, image_h, crop_w, crop_h):
        sw, sh = (image_w - crop_w) // 4, (image_h - crop_h) // 4
        grid = [(0, 0), (4, 0), (0, 4), (4, 4), (2, 2)]
        if more_fix_crop:
            grid += [(0, 2), (4, 2), (2, 4), (2, 0), (1, 1), (3, 1), (1, 3), (3, 3)]
        return [(i * sw, j * sh) for i, j in grid]


def geometry_plan(transform, im_shape):
    """What ``transform`` (a Compose of the classes above) WOULD do to frames of ``im_shape`` (H, W[, C]),
    without touching pixels: ``(plan, (out_h, out_w), flip)`` with
    ``plan = (y0, x0, h, w, rh, rw, cy, cx)`` -- the box (y0, x0, h, w) of the frame is resized to
    rh x rw and the out_h x out_w window at (cy, cx) of that image is the result -- and the
    horizontal-flip decision.  Covers the reference's pipelines: crop -> resize -> flip (training)
    and scale -> centre crop (validation / test).  The random draws are made in the order applying the
    transforms makes them (choice(pairs), randint, randint, random()), so a loader that ships the
    plan to the GPU (``ops.prepare_inputs``) consumes the RNG exactly like one that applies the
    transforms on the CPU."""
    h, w = int(im_shape[0]), int(im_shape[1])
    box, rs, win, flip = [0, 0, h, w], None, None, False      # rs: resized size; win: (cy, cx, oh, ow) after it
    ts = transform.transforms if isinstance(transform, Compose) else [transform]

    def crop(top, left, ch, cw):
        nonlocal box, win
        if rs is None:                                           # still in source pixels
            box = [box[0] + top, box[1] + left, ch, cw]
        else:                                                    # a window of the resized image
            cy, cx = (win[0], win[1]) if win else (0, 0)
            win = (cy + top, cx + left, ch, cw)

    for t in ts:
        if flip:
            raise ValueError("geometry_plan: a geometric transform after the flip is not supported")
        cur_h, cur_w = (win[2], win[3]) if win else (rs if rs else (box[2], box[3]))
        if isinstance(t, GroupCenterCrop):
            crop((cur_h - t.size) // 2, (cur_w - t.size) // 2, t.size, t.size)
        elif isinstance(t, (GroupScale, GroupMultiScaleCrop)):
            if rs is not None:
                raise ValueError("geometry_plan: two resampling passes; apply the transforms on the CPU")
            if isinstance(t, GroupMultiScaleCrop):
                cw, ch, ow, oh = t._sample_crop_size((cur_h, cur_w))
                crop(ow, oh, cw, ch)
                rs = (t.input_size[0], t.input_size[1])
            else:
                rs = (t.size, t.size)
        elif isinstance(t, GroupRandomHorizontalFlip):
            flip = random.random() < 0.5
        else:
            raise ValueError("geometry_plan: unsupported transform %r" % (t,))
    if rs is None:
        rs = (box[2], box[3])
    if win is None:
        win = (0, 0, rs[0], rs[1])
    return (box[0], box[1], box[2], box[3], rs[0], rs[1], win[0], win[1]), (win[2], win[3]), flip


def apply_plan(img, plan, out, flip):
    """The CPU evaluation of a plan (what the GPU kernel computes in uint8)."""
    y0, x0, h, w, rh, rw, cy, cx = plan
    r = resize_bilinear(img[y0:y0 + h, x0:x0 + w], rh, rw)[cy:cy + out[0], cx:cx + out[1]]
    return flip_with_x_negation(r) if flip else r
