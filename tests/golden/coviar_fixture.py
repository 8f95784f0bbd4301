"""Seeded stand-in for the data side of ``CoviarDataSet`` (test infrastructure, shared by the golden
generator -- which drives the REFERENCE's ``__getitem__`` with it -- and by the tests, which drive
this repository's dataset with the same files and the same stub).

* ``coviar.load(path, gop_index, gop_pos, representation_idx, accumulate)`` -> int32 arrays
  ``[H0, W0, 2]`` (motion vectors, sigma 6) or ``[H0, W0, 3]`` (residual, sigma 12), a function of
  the arguments only; ``coviar.get_num_frames(path)`` -> NUM_FRAMES.
* ``write_dataset(root)`` writes the TV-L1 flow frames the dataset pairs with them,
  ``<root>/flow/<class>/<video>/flow_{x,y}_%05d.jpg`` -- PNG-encoded (lossless; PIL identifies the
  format from the content), so the pixels do not depend on a JPEG codec -- and a video list.
"""
import os
import types
import zlib

import numpy as np

H0, W0 = 64, 80
NUM_FRAMES = 30
VIDEOS = [("brush_hair/clip_a.avi", 0), ("cartwheel/clip_b.avi", 7)]


def _rs(*key):
    return np.random.RandomState(zlib.crc32(repr(key).encode()) % (2 ** 31))


def load(path, gop_index, gop_pos, representation_idx, accumulate):
    rs = _rs("coviar", os.path.basename(path), int(gop_index), int(gop_pos), int(representation_idx), bool(accumulate))
    if representation_idx == 1:
        return np.round(rs.normal(0, 6, (H0, W0, 2))).astype(np.int32)
    return np.round(rs.normal(0, 12, (H0, W0, 3))).astype(np.int32)


def get_num_frames(path):
    return NUM_FRAMES


def coviar_module():
    m = types.ModuleType("coviar")
    m.load, m.get_num_frames = load, get_num_frames
    return m


def flow_image(video, axis, idx):
    rs = _rs("flow", video, axis, int(idx))
    return np.clip(128 + np.round(rs.normal(0, 10, (H0, W0))), 0, 255).astype(np.uint8)


def write_dataset(root):
    """Returns (data_root, flow_root, video_list_path)."""
    from PIL import Image
    data_root, flow_root = os.path.join(root, "mpeg4"), os.path.join(root, "flow")
    os.makedirs(data_root, exist_ok=True)
    lines = []
    for video, label in VIDEOS:
        cls, name = video.split("/")
        d = os.path.join(flow_root, cls, name[:-4])
        os.makedirs(d, exist_ok=True)
        for idx in range(1, NUM_FRAMES + 1):
            for axis in "xy":
                with open(os.path.join(d, "flow_%s_%05d.jpg" % (axis, idx)), "wb") as f:
                    Image.fromarray(flow_image(video, axis, idx), mode="L").save(f, format="PNG")
            # the reference counts len(os.listdir(flow_path)) / 3: a third file per frame, as a TV-L1 dump has
            open(os.path.join(d, "img_%05d.jpg" % idx), "wb").close()
        lines.append("%s %d %d" % (video, NUM_FRAMES, label))
    lst = os.path.join(root, "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(lines) + "\n")
    return data_root, flow_root, lst


#: (tag, is_train, mv_minmaxnorm, python random seed, dataset index, with flip transform)
CASES = [("test0", False, 0, 0, 0, False), ("test1", False, 1, 0, 1, False),
         ("train_s3", True, 1, 3, 0, True), ("train_s4", True, 0, 4, 0, True),
         ("train_s5", True, 1, 5, 0, True), ("train_s8", True, 1, 8, 0, True)]
CROP = 48
DS16_CROPS = (48, 40)      # flow_ds_factor = 16 cases: whole blocks / ragged blocks (zero-padded by block_reduce)
