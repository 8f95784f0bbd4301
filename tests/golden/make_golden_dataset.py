#!/usr/bin/env python3
"""G9: the tensor contract of the dataset, produced by the REFERENCE's own
``CoviarDataSet.__getitem__`` (code/dmcnet/dataset.py:151-281) run in this container:

    python tests/golden/make_golden_dataset.py

The reference's dataset.py / transforms.py are imported as they are (stub ``cv2`` / ``skimage`` /
``torchvision`` modules for what this image lacks; ``np.float = float`` because dataset.py:41 uses
the removed alias), ``coviar.load`` is the seeded stand-in of tests/golden/coviar_fixture.py and the
flow frames are the files that module writes.  Transforms: the reference's own ``GroupCenterCrop``
and ``GroupRandomHorizontalFlip`` (pure numpy; ``GroupScale`` / ``GroupMultiScaleCrop`` call
cv2.resize, which is absent).  Stored: the reference's 4-tuple per case, ``flow_ds_factor = 0``
(g9_dataset_item.npz) and ``flow_ds_factor = 16`` (g9_dataset_item_ds16.npz: BASELINE config 2's setting).
For the second file the reference's own lines around the blockify call (dataset.py:229-246: which tensor is reduced, the
``repeat`` back, the crop to the input size, its place BEFORE /255 and the normalisation) run as they are; only
``skimage.measure.block_reduce`` itself -- absent from this image -- is a numpy stand-in written from its documentation
(pad with ``cval = 0`` to whole blocks, apply ``func`` over every block).  Crops of 48 (whole 16 x 16 blocks) and 40
(the ragged, zero-padded path).
"""
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden import coviar_fixture as CF            # noqa: E402
from tests.golden.make_golden import import_ref, install_stubs   # noqa: E402


def block_reduce_standin(image, block_size, func=np.sum, cval=0):
    """skimage.measure.block_reduce as documented: the image is padded with ``cval`` where it is not a whole number of blocks,
    then ``func`` is applied over each block's axes."""
    pad = [(0, (-s) % b) for s, b in zip(image.shape, block_size)]
    image = np.pad(image, pad, mode="constant", constant_values=cval)
    shape = []
    for s, b in zip(image.shape, block_size):
        shape += [s // b, b]
    return func(image.reshape(shape), axis=tuple(range(1, 2 * image.ndim, 2)))


def main():
    install_stubs()
    sys.modules["coviar"] = CF.coviar_module()
    if not hasattr(np, "float"):
        np.float = float                                   # dataset.py:41 (removed numpy alias)
    ref_ds = import_ref("dmcnet", "dataset")
    ref_tf = import_ref("dmcnet", "transforms")
    Compose = sys.modules["torchvision"].transforms.Compose
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        data_root, flow_root, lst = CF.write_dataset(tmp)
        for tag, is_train, minmax, seed, index, with_flip in CF.CASES:
            ts = [ref_tf.GroupCenterCrop(CF.CROP)] + ([ref_tf.GroupRandomHorizontalFlip()] if with_flip else [])
            ds = ref_ds.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 0, False, Compose(ts), 3,
                                      is_train, True, 12, mv_minmaxnorm=minmax)
            assert len(ds) == len(CF.VIDEOS)
            random.seed(seed)
            flow, mv, res, label = ds[index]
            out[tag + "_flow"], out[tag + "_mv"], out[tag + "_res"] = flow.numpy(), mv.numpy(), res.numpy()
            out[tag + "_label"] = np.int64(label)
            print(tag, tuple(flow.shape), tuple(mv.shape), tuple(res.shape), label, float(mv.mean()))
    np.savez_compressed(os.path.join(HERE, "g9_dataset_item.npz"), **out)
    # flow_ds_factor = 16 through the reference's own __getitem__ (block_reduce: the stand-in above)
    ref_ds.block_reduce = block_reduce_standin
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        data_root, flow_root, lst = CF.write_dataset(tmp)
        for tag, is_train, minmax, seed, index, with_flip in CF.CASES:
            for crop in CF.DS16_CROPS:
                ts = [ref_tf.GroupCenterCrop(crop)] + ([ref_tf.GroupRandomHorizontalFlip()] if with_flip else [])
                ds = ref_ds.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 16, False, Compose(ts), 3,
                                          is_train, True, 12, mv_minmaxnorm=minmax)
                random.seed(seed)
                flow, mv, res, label = ds[index]
                key = "%s_c%d" % (tag, crop)
                out[key + "_flow"], out[key + "_mv"], out[key + "_res"] = flow.numpy(), mv.numpy(), res.numpy()
                out[key + "_label"] = np.int64(label)
                print(key, tuple(flow.shape), flow.dtype, label, float(flow.mean()))
    np.savez_compressed(os.path.join(HERE, "g9_dataset_item_ds16.npz"), **out)
    # round 6: the 10-crop test transform (the reference's own GroupOverSample, transforms.py:77-114, without its optional
    # cv2 GroupScale) and upsample_interp=True (dataset.py:236-246: the reference's own scipy interp1d lines; block_reduce:
    # the stand-in above), both through the reference's __getitem__
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        data_root, flow_root, lst = CF.write_dataset(tmp)
        for tag, is_train, minmax, seed, index, _with_flip in (CF.CASES[0], CF.CASES[2]):
            ds = ref_ds.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 0, False,
                                      Compose([ref_tf.GroupOverSample(CF.CROP, None)]), 3, is_train, True, 12, mv_minmaxnorm=minmax)
            random.seed(seed)
            flow, mv, res, label = ds[index]
            assert flow.shape[0] == 30
            key = tag + "_over"
            out[key + "_flow"], out[key + "_mv"], out[key + "_res"] = flow.numpy(), mv.numpy(), res.numpy()
            out[key + "_label"] = np.int64(label)
            for crop in CF.DS16_CROPS:
                ds = ref_ds.CoviarDataSet(data_root, flow_root, "hmdb51", lst, "mv", 1, 16, True,
                                          Compose([ref_tf.GroupCenterCrop(crop)]), 3, is_train, True, 12, mv_minmaxnorm=minmax)
                random.seed(seed)
                flow, mv, res, label = ds[index]
                key = "%s_interp_c%d" % (tag, crop)
                out[key + "_flow"], out[key + "_label"] = flow.numpy(), np.int64(label)
                print(key, tuple(flow.shape), flow.dtype, float(flow.mean()), float(flow.std()))
    np.savez_compressed(os.path.join(HERE, "g9_dataset_item_extra.npz"), **out)


if __name__ == "__main__":
    main()
