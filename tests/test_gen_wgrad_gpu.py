"""EstimatorDenseNetTiny weight gradient, row-sliding form (csrc/gen_wgrad.hip; option gen_wgrad_path = 5, the default): every operand
split into its bf16 slices once, into LDS rings a workgroup walks down 32-column strips two rows at a time; only left neighbours
are needed (the strip's left column comes from the previous segment's cache, or from memory where a workgroup's range begins).
Against the CPU oracle, against the tile kernel (gen_wgrad_path = 4: the same bf16x3 products in another summation order) and
against an fp64 evaluation, on the shapes its geometry cares about: odd heights (a last row pair with one row), widths that are not
whole strips, images of two rows, more segments than workgroups and fewer, several bands per strip (heights above 28), ranges that
begin at a strip > 0 (the cold fill of the left-column cache); bitwise determinism.
Reference semantics: autograd of code/dmcnet/model.py:187-194."""
import copy

import pytest
import torch

import dmcnet_amd
from tests.test_hip_parity import DEV, rel_err, rnd, tiny_pair

pytestmark = pytest.mark.gpu


def _set(name, value):
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(name)
    dmcnet_amd._lib.check(lib.dmc_set_option(name, value), "dmc_set_option")
    return before


def _grads(m, x, r, path, request=None):
    before = _set(b"gen_wgrad_path", path)
    try:
        m.zero_grad()
        y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=True)
        (y * r.to(DEV)).sum().backward()
        return {k: p.grad.clone() for k, p in m.named_parameters()}
    finally:
        _set(b"gen_wgrad_path", before)


SHAPES = [(1, 5, 2, 4), (1, 5, 3, 8), (2, 5, 2, 32), (1, 5, 5, 36), (2, 5, 8, 64), (3, 5, 17, 220), (1, 5, 33, 224), (2, 5, 16, 60),
          (1, 5, 7, 12), (2, 5, 23, 100), (1, 5, 2, 224), (1, 5, 40, 228), (5, 5, 9, 28), (2, 5, 64, 260), (300, 5, 4, 32)]


@pytest.mark.parametrize("shape", SHAPES)
def test_row_sliding_wgrad_vs_oracle_and_tile_kernel(shape):
    o, m = tiny_pair(12)
    x, r = rnd(7, shape), rnd(8, (shape[0], 2) + shape[2:])
    ((o(x) + x[:, :2]) * r).sum().backward()
    g5 = _grads(m, x, r, 5)
    g4 = _grads(m, x, r, 4)
    for k, po in o.named_parameters():
        assert rel_err(g5[k], po.grad) < 1e-4, k
        assert rel_err(g5[k], g4[k]) < 1e-5, k               # the same products, another order of the fp32 sums


def test_row_sliding_wgrad_full_frames_fp64_and_determinism():
    """Eight full 224 x 224 frames (448 segments on 256 workgroups: one or two each, most ranges begin at a strip > 0): as close to
    fp64 as the tile kernel, twice the same bits."""
    o, m = tiny_pair(14)
    x, r = rnd(21, (8, 5, 224, 224)), rnd(22, (8, 2, 224, 224))
    o64 = copy.deepcopy(o).double()
    ((o64(x.double()) + x[:, :2].double()) * r.double()).sum().backward()
    g5, g5b, g4 = _grads(m, x, r, 5), _grads(m, x, r, 5), _grads(m, x, r, 4)
    for k, p64 in o64.named_parameters():
        assert torch.equal(g5[k], g5b[k]), k
        scale = float(p64.grad.abs().max())
        e5 = float((g5[k].double().cpu() - p64.grad).abs().max()) / scale
        e4 = float((g4[k].double().cpu() - p64.grad).abs().max()) / scale
        assert e5 <= max(2 * e4, 2e-6), (k, e5, e4)
