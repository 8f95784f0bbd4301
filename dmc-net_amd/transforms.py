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
