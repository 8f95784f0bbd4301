"""EstimatorDenseNetTiny forward as ONE launch (csrc/gen_fused.hip; option gen_fused bit 0, the default) and its data gradient as
ONE launch (gen_fused_bwd.hip; bit 1, opt-in: measured slower than the five layer launches): strips of <= 118 columns
walked row by row, six layers pipelined across the waves of a workgroup.  The oracle / golden cases of tests/test_hip_parity.py
run on it by default; here: the shapes its geometry cares about (one strip with one or two pixel halves, two strips with even
and odd widths, images shorter than the pipeline, more strips than workgroups), the saved features the backward reads, the
layer-by-layer path as a second witness, an fp64 anchor and bitwise determinism.
Reference semantics: code/dmcnet/model.py:172-194, :111-119, :341-346."""
import copy

import pytest
import torch

import dmcnet_amd
from dmcnet_amd import ops
from tests.test_hip_parity import DEV, rel_err, rnd, tiny_pair

pytestmark = pytest.mark.gpu


def _set(name, value):
    lib = dmcnet_amd._lib.load()
    before = lib.dmc_get_option(name)
    dmcnet_amd._lib.check(lib.dmc_set_option(name, value), "dmc_set_option")
    return before


def _all_fused():
    """3 = the one-launch forward AND the one-launch data gradient: the latter lost its measurement and is compiled into the
    -DDMC_MEASURE build only (DMC_HIP_LIB=.../libdmcnet_hip_measure.so); the product library serves bit 0 = the forward."""
    return 3 if dmcnet_amd._lib.load().dmc_get_option(b"measure_build") == 1 else 1


def test_fused_forward_is_the_default():
    assert dmcnet_amd._lib.load().dmc_get_option(b"gen_fused") == 1      # bit 0: forward (default), bit 1: data gradient (opt-in)


# widths: <= 62 one pixel half; 63 .. 118 two halves of one strip; 119 .. 224 two strips (odd widths: unequal strips);
# heights below the 11-step pipeline depth; 225 takes the layer-by-layer kernels
@pytest.mark.parametrize("shape", [(1, 5, 1, 1), (2, 5, 2, 3), (1, 5, 5, 62), (1, 5, 7, 63), (2, 5, 9, 64), (1, 5, 12, 117),
                                   (1, 5, 13, 118), (1, 5, 11, 119), (2, 5, 10, 120), (1, 5, 8, 121), (1, 5, 17, 223),
                                   (2, 5, 30, 224), (1, 5, 3, 224), (1, 5, 1, 224), (1, 5, 26, 180), (1, 5, 6, 225)])
@pytest.mark.parametrize("delta,fused", [(False, 1), (True, 1), (True, 3)])
def test_fused_forward_edge_shapes(shape, delta, fused, request):
    if fused == 3 and _all_fused() != 3:
        pytest.skip("the one-launch data gradient exists in the -DDMC_MEASURE build only")
    before = _set(b"gen_fused", fused)                 # 3: the one-launch data gradient too
    request.addfinalizer(lambda: _set(b"gen_fused", before))
    o, m = tiny_pair(12)
    x = rnd(7, shape)
    yo = o(x) + (x[:, :2] if delta else 0)
    y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=delta)
    assert rel_err(y, yo) < 1e-5
    r = rnd(8, tuple(yo.shape))
    (yo * r).sum().backward()
    (y * r.to(DEV)).sum().backward()          # the backward kernels read the features the fused forward saved
    for (k, po), (_, pm) in zip(o.named_parameters(), m.named_parameters()):
        assert rel_err(pm.grad, po.grad) < 1e-4, k


def test_fused_forward_more_strips_than_workgroups():
    """600 strips of 10 x 20 pixels on <= 256 persistent workgroups: every workgroup walks several strips (the rings are re-used
    without clearing: rows outside an image are skipped, never stored)."""
    o, m = tiny_pair(5)
    x = rnd(9, (600, 5, 10, 20))
    yo = o(x) + x[:, :2]
    y = m.forward_mv_res(x[:, :2].contiguous().to(DEV), x[:, 2:].contiguous().to(DEV), add_mv=True)
    assert rel_err(y, yo) < 1e-5


def test_fused_forward_saved_features_and_second_witness():
    """The 28 saved feature planes (what the data- and weight-gradient kernels read) against an fp64 evaluation of the oracle's
    layers, and output + features against the layer-by-layer kernels (exact-fp32 too: equal to rounding, not bit for bit)."""
    o, m = tiny_pair(13)
    o64 = copy.deepcopy(o).double()
    mv, res = rnd(1, (3, 2, 72, 224)), rnd(2, (3, 3, 72, 224))
    x64 = torch.cat([mv, res], 1).double()
    feats64 = []
    xin = x64
    for i in range(5):
        f = getattr(o64, "conv_%d" % i)(xin)
        feats64.append(f)
        xin = torch.cat((f, xin), 1)
    y64 = o64.predict_flow(xin) + mv.double()
    ws, bs = m._params()
    got = {}
    F = _all_fused()
    for fused in (F, 0):
        before = _set(b"gen_fused", fused)
        try:
            with torch.enable_grad():
                y = ops.gen_tiny(mv.to(DEV), res.to(DEV), ws, bs, add_mv=True)
            saved = y.grad_fn.saved_tensors[2].view(3, 28, 72, 224)
            got[fused] = (y.detach().clone(), saved.clone())
        finally:
            _set(b"gen_fused", before)
    f64 = torch.cat(feats64, 1)                                  # physical order: y0 | y1 | y2 | y3 | y4
    e_f, e_l = rel_err(got[F][1], f64), rel_err(got[0][1], f64)
    assert e_f <= max(2 * e_l, 1e-6), (e_f, e_l)
    e_f, e_l = rel_err(got[F][0], y64), rel_err(got[0][0], y64)
    assert e_f <= max(2 * e_l, 1e-6), (e_f, e_l)
    assert rel_err(got[F][0], got[0][0]) < 2e-6 and rel_err(got[F][1], got[0][1]) < 2e-6


def test_fused_forward_full_frames_vs_fp64_and_determinism():
    """Output at 224 x 224 against fp64, no further than the layer-by-layer path; the loss of the fused MSE epilogue; every
    parameter gradient (through the saved features) against the fp64 backward forced to the run's own LeakyReLU branches
    (tests/gen_conditioned.py); two runs bit-identical."""
    from tests.gen_conditioned import conditioned_report
    o, m = tiny_pair(13)
    o64 = copy.deepcopy(o).double()
    mv, res, flow = rnd(1, (4, 2, 224, 224)), rnd(2, (4, 3, 224, 224)), rnd(4, (4, 2, 224, 224))
    r = rnd(3, (4, 2, 224, 224))
    x = torch.cat([mv, res], 1)
    with torch.no_grad():
        y64 = o64(x.double()) + mv.double()
    loss64 = float(((y64 - flow.double()) ** 2).mean())
    runs = {}
    F = _all_fused()
    for fused in (F, F, 0):
        before = _set(b"gen_fused", fused)
        try:
            m.zero_grad()
            y = m.forward_mv_res(mv.to(DEV), res.to(DEV), add_mv=True)
            saved = y.grad_fn.saved_tensors[2].view(4, 28, 224, 224).clone()
            (y * r.to(DEV)).sum().backward()
            with torch.no_grad():
                _, loss = m.forward_mv_res_mse(mv.to(DEV), res.to(DEV), flow.to(DEV), add_mv=True)
            runs.setdefault(fused, []).append((y.detach().clone(), [p.grad.clone() for p in m.parameters()], float(loss), saved))
        finally:
            _set(b"gen_fused", before)
    a, b = runs[F]
    assert torch.equal(a[0], b[0]) and a[2] == b[2] and torch.equal(a[3], b[3])
    for ga, gb in zip(a[1], b[1]):
        assert torch.equal(ga, gb)
    lay = runs[0][0]
    assert rel_err(a[0], y64) <= max(2 * rel_err(lay[0], y64), 1e-6)
    assert abs(a[2] - loss64) <= 1e-5 * abs(loss64)
    rep = conditioned_report(o, o64, x, r, a[0], a[1], a[3])
    assert rep["flips_hip"] <= 4 * rep["flips_ref"] + 16, rep
    for k, (e_hip, e_ref) in rep["params"].items():
        assert e_hip <= max(2 * e_ref, 2e-6), (k, e_hip, e_ref)


def test_fused_forward_inference_keeps_no_features():
    """Under no_grad the op passes saved = NULL: the one-launch forward writes the output only (8 instead of 120 B/px) -- the
    result is bit for bit the training-mode one; the layer-by-layer kernels, which pass features from launch to launch through
    that buffer, refuse a NULL one at the C ABI."""
    import ctypes
    _, m = tiny_pair(21)
    mv, res = rnd(31, (3, 2, 50, 224)).to(DEV), rnd(32, (3, 3, 50, 224)).to(DEV)
    y_train = m.forward_mv_res(mv, res, add_mv=True)
    assert y_train.grad_fn is not None
    with torch.no_grad():
        y_eval = m.forward_mv_res(mv, res, add_mv=True)
    assert y_eval.grad_fn is None and torch.equal(y_eval, y_train.detach())
    L, lib = dmcnet_amd._lib, dmcnet_amd._lib.load()
    ws, bs = m._params()
    ws, bs = [w.detach().contiguous() for w in ws], [b.detach().contiguous() for b in bs]
    out = torch.empty_like(y_eval)
    work = torch.empty(lib.dmc_gen_tiny_workspace_bytes() // 4, device=DEV)
    # the success path of saved == NULL at the C ABI itself (a.feat == nullptr in gen_fused_kernel), bit for bit
    out.fill_(float("nan"))
    L.check(lib.dmc_gen_tiny_fwd(L.ptr(mv), L.ptr(res), L.ptr_array(ws), L.ptr_array(bs), L.ptr(out), ctypes.c_void_p(0), L.ptr(work),
                                 3, 50, 224, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "dmc_gen_tiny_fwd")
    assert torch.equal(out, y_train.detach())
    # ... and through ops: with TRAINABLE parameters, no_grad must not allocate the feature buffer (needs_input_grad alone reads
    # True under no_grad; ops.gen_tiny passes the grad mode), grad mode must
    calls = []
    orig = dmcnet_amd.ops._floats
    dmcnet_amd.ops._floats = lambda nbytes, dev: (calls.append(nbytes), orig(nbytes, dev))[1]
    try:
        assert all(p.requires_grad for p in m.parameters())
        with torch.no_grad():
            m.forward_mv_res(mv, res, add_mv=True)
        eval_allocs = list(calls)
        del calls[:]
        m.forward_mv_res(mv, res, add_mv=True)
        train_allocs = list(calls)
    finally:
        dmcnet_amd.ops._floats = orig
    feat = lib.dmc_gen_tiny_saved_bytes(3, 50, 224)
    assert feat not in eval_allocs and feat in train_allocs, (feat, eval_allocs, train_allocs)
    before = _set(b"gen_fused", 0)
    try:
        rc = lib.dmc_gen_tiny_fwd(L.ptr(mv), L.ptr(res), L.ptr_array(ws), L.ptr_array(bs), L.ptr(out), ctypes.c_void_p(0), L.ptr(work),
                                  3, 50, 224, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc != 0 and b"saved == NULL" in lib.dmc_last_error()
    finally:
        _set(b"gen_fused", before)


def test_grid_reserve_cus_changes_no_result():
    """Option grid_reserve_cus (CUs the persistent grids leave to other streams, e.g. RCCL's channel kernels): the generator's
    one-launch forward, its backward (ring / Winograd / row-sliding weight-gradient kernels, all persistent) and the
    patch-resident 3-D convolution give the same values on 256 and on 192 CUs -- every output element is produced by one
    workgroup whichever way the work list is dealt; sums over workgroups (weight gradients, the fused MSE) may group their
    partials differently and are compared to fp32 rounding."""
    from dmcnet_amd import ops
    o, m = tiny_pair(17)
    mv, res, flow = rnd(41, (6, 2, 224, 224)).to(DEV), rnd(42, (6, 3, 224, 224)).to(DEV), rnd(43, (6, 2, 224, 224)).to(DEV)
    x3 = rnd(44, (2, 48, 4, 28, 28)).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last_3d)
    w3 = (rnd(45, (96, 48, 3, 3, 3)) * 0.05).to(DEV)
    runs = []
    for reserve in (0, 64):
        before = _set(b"grid_reserve_cus", reserve)
        try:
            m.zero_grad()
            y, loss = m.forward_mv_res_mse(mv, res, flow, add_mv=True)
            (y.sum() * 1e-3 + loss).backward()
            with torch.no_grad():
                y3 = ops.conv3d_bf16(x3, w3)
            runs.append((y.detach().clone(), float(loss), [p.grad.clone() for p in m.parameters()], y3.clone()))
        finally:
            _set(b"grid_reserve_cus", before)
    a, b = runs
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
    assert abs(a[1] - b[1]) <= 1e-6 * abs(a[1])
    for ga, gb in zip(a[2], b[2]):
        assert rel_err(gb, ga) < 1e-5

