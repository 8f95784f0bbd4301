"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/dmcnet_hip.h declares (no compute calls without a GPU); host-side Model surface."""
import ctypes
import os
import re

import pytest
import torch

import dmcnet_amd
from dmcnet_amd import _lib
from oracle import dmc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dmcnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libdmcnet_hip.so does not export " + n
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES out of sync with the header"


def test_size_queries_and_error_reporting_without_gpu():
    lib = _lib.load()
    assert lib.dmc_version() >= 100
    assert lib.dmc_gen_tiny_workspace_bytes() == (7788 + 256) * 4 + 36 * 1024       # + the bf16x3 fragments of gen_x3.hip
    assert lib.dmc_gen_tiny_saved_bytes(120, 224, 224) == 120 * 28 * 224 * 224 * 4
    assert lib.dmc_gen_tiny_partials_bytes(1, 8, 32) == (1 + 16) * 31 * 256 * 4
    # invalid arguments are rejected before any launch
    rc = lib.dmc_flow_mse_fwd(None, None, None, None, 0, None)
    assert rc == -1 and b"dmc_flow_mse_fwd" in lib.dmc_last_error()


def test_hot_path_fails_loudly_on_cpu_tensors():
    m = dmcnet_amd.Model(51, 3, "mv", base_model="resnet18", use_databn=0, gen_flow_or_delta=1,
                         arch_estimator="DenseNetTiny")
    mv, res = torch.zeros(1, 3, 2, 32, 32), torch.zeros(1, 3, 3, 32, 32)
    with pytest.raises(_lib.DmcHipError):
        m(mv, res)
    with pytest.raises(_lib.DmcHipError):
        dmcnet_amd.ops.flow_mse(torch.zeros(4), torch.zeros(4))
    # the classifier-side HIP ops refuse CPU tensors too (their callers fall back to stock modules
    # only through the *_supported() predicates, which are False on the CPU)
    bn = torch.nn.BatchNorm2d(64)
    xcl = torch.zeros(2, 64, 8, 8).contiguous(memory_format=torch.channels_last)
    assert not dmcnet_amd.ops.bn_act_supported(xcl) and not dmcnet_amd.ops.bn_relu_pool_supported(xcl)
    assert not dmcnet_amd.ops.stem_conv_supported(torch.zeros(1, 2, 8, 8), torch.zeros(64, 2, 7, 7))
    for call in (lambda: dmcnet_amd.ops.bn_act(xcl, bn), lambda: dmcnet_amd.ops.bn_relu_pool(xcl, bn),
                 lambda: dmcnet_amd.ops.stem_conv(torch.zeros(1, 2, 8, 8), torch.zeros(64, 2, 7, 7))):
        with pytest.raises(_lib.DmcHipError):
            call()


@pytest.mark.parametrize("arch_d", [None, "Discriminator", "Discriminator3", "Discriminator4"])
@pytest.mark.parametrize("est", ["DenseNetTiny", "ContextNetwork", "DenseNetSmall",
                                 "DenseNetTinyEarlyFusionStack"])
def test_state_dict_keys_match_reference_layout(arch_d, est):
    kw = dict(base_model="resnet18", use_databn=1, gen_flow_or_delta=1, arch_estimator=est)
    m = dmcnet_amd.Model(51, 3, "mv", arch_d=arch_d, **kw)
    o = O.OracleModel(51, 3, "mv", arch_d=arch_d, **kw)
    a, b = m.state_dict(), o.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)
    if arch_d is not None:
        bn = m.discriminator.discriminator_block_2[3]
        assert bn.eps == 0.8 and bn.momentum == 0.1      # nn.BatchNorm2d(C, 0.8) binds eps
    assert m.crop_size == 224 and m.scale_size == 256
    assert callable(m.get_augmentation())


def test_other_backbones_construct():
    m = dmcnet_amd.Model(101, 3, "mv", base_model="resnet50", arch_estimator="DenseNetTiny")
    o = O.OracleModel(101, 3, "mv", base_model="resnet50", arch_estimator="DenseNetTiny")
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())
    with pytest.raises(ValueError):
        dmcnet_amd.Model(51, 3, "mv", base_model="vgg16")


def test_i3d_3d_ops_host_side_without_gpu():
    """Host-only parts of the I3D 3-D entry points: MaxPool3dTFPadding output extents equal the stock module's
    (ConstantPad3d + MaxPool3d(ceil_mode=True), code/dmcnet_I3D/network/i3d.py:406-418) on the trunk's pools,
    workspace sizes, argument validation; on the CPU the predicates are False and the ops raise (no fallback);
    the I3D modules themselves stay constructible and runnable with the stock ops."""
    import ctypes
    from dmcnet_amd import i3d, ops
    lib = _lib.load()
    od, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    for kernel, stride, (d, h, w) in [((1, 3, 3), (1, 2, 2), (32, 112, 112)), ((3, 3, 3), (2, 2, 2), (32, 28, 28)),
                                      ((2, 2, 2), (2, 2, 2), (16, 14, 14)), ((2, 2, 2), (2, 2, 2), (5, 7, 7)),
                                      ((3, 3, 3), (1, 1, 1), (8, 7, 7)), ((3, 3, 3), (2, 2, 2), (9, 13, 6))]:
        assert lib.dmc_maxpool3d_tf_out_shape(d, h, w, 64, *kernel, *stride, ctypes.byref(od), ctypes.byref(oh), ctypes.byref(ow)) == 1
        ref = i3d.MaxPool3dTFPadding(kernel, stride)(torch.zeros(1, 8, d, h, w))
        assert (od.value, oh.value, ow.value) == tuple(ref.shape[2:]), (kernel, stride, (d, h, w))
    assert lib.dmc_maxpool3d_tf_out_shape(8, 8, 8, 12, 3, 3, 3, 1, 1, 1, None, None, None) == 0      # C % 8 != 0
    assert lib.dmc_conv3d_bf16_supported(3, 32, 56, 56, 64, 192, 3, 3, 3) == 1
    assert lib.dmc_conv3d_bf16_supported(3, 32, 56, 56, 2, 64, 7, 7, 7) == 0                         # the stem stays stock
    assert lib.dmc_conv3d_bf16_supported(3, 32, 56, 56, 20, 64, 3, 3, 3) == 0                        # Cin % 8 != 0
    assert lib.dmc_conv3d_bf16_wpack_bytes(64, 192, 3, 3, 3) >= 2 * 256 * 27 * 64
    assert lib.dmc_conv3d_bf16_wgrad_bytes(3, 32, 56, 56, 64, 192, 3, 3, 3) >= 192 * 27 * 64 * 4
    rc = lib.dmc_conv3d_bf16_fwd(None, None, 1, 1, 1, None, None, None, 1, 1, 1, 1, 8, 8, 1, 1, 1, None)
    assert rc == -1 and b"dmc_conv3d_bf16_fwd" in lib.dmc_last_error()
    x = torch.zeros(1, 16, 2, 4, 4, dtype=torch.bfloat16)
    assert not ops.conv3d_bf16_supported(x, torch.zeros(8, 16, 3, 3, 3)) and not ops.maxpool3d_tf_supported(x, (3, 3, 3), (1, 1, 1))
    with pytest.raises(_lib.DmcHipError):
        ops.conv3d_bf16(x, torch.zeros(8, 16, 3, 3, 3))
    unit = i3d.Unit3Dpy(16, 8, (3, 3, 3))
    assert unit(torch.zeros(1, 16, 2, 4, 4)).shape == (1, 8, 2, 4, 4)                                # stock path on the CPU


def test_classifier_conv_entry_points_host_side_without_gpu():
    """Host-only parts of the round-2 classifier convolution entry points: which channel pairs take the bf16x3
    kernels (and therefore the one-launch weight split, dmc_conv_nhwc_split), the option that switches the arithmetic,
    and argument validation of the split and of the data gradient with the residual addend -- all of which return
    before any launch; the identity-shortcut blocks of resnet.py (torchvision BasicBlock / Bottleneck behind
    code/dmcnet/model.py:305) run on the stock ops on the CPU with the residual-gradient link switched on or off."""
    from dmcnet_amd import resnet
    lib = _lib.load()
    assert lib.dmc_get_option(b"conv_arith") == 1 and lib.dmc_get_option(b"gen_wgrad_path") == 5
    # generator kernel selection: layer 1 on gen_x3.hip, data-gradient groups 0 and 1 on the Winograd ring kernel;
    # the measurement-only options are off
    assert lib.dmc_get_option(b"gen_x3") == 2 and lib.dmc_get_option(b"gen_wino") == 0x300
    assert lib.dmc_get_option(b"gen_fused") == 1                 # bit 0: the one-launch forward (csrc/gen_fused.hip); bit 1 (opt-in): the one-launch data gradient (gen_fused_bwd.hip)
    # the options that switch parts of a kernel off (results wrong) exist only in the -DDMC_MEASURE build: the product library
    # refuses to set them and reads them as 0 -- no option value can make it compute something else than the reference
    for name in (b"gen_ablate", b"gen_stagger", b"conv_ablate"):
        assert lib.dmc_get_option(name) == 0
        assert lib.dmc_set_option(name, 1) != 0 and b"DMC_MEASURE" in lib.dmc_last_error()
        assert lib.dmc_get_option(name) == 0
    # kernel variants that lost their A/B measurement are compiled into the -DDMC_MEASURE build only (round 6): the product
    # library refuses the option VALUES that select them and keeps the value it had
    assert lib.dmc_get_option(b"measure_build") == 0
    for name, bad, good in ((b"gen_fused", (2, 3), 1), (b"gen_layer_path", (3, 4, 5), 1), (b"gen_wgrad_path", (0, 1, 2, 3), 5)):
        for v in bad:
            assert lib.dmc_set_option(name, v) != 0 and b"DMC_MEASURE" in lib.dmc_last_error(), (name, v)
            assert lib.dmc_get_option(name) == good
        assert lib.dmc_set_option(name, good) == 0
    assert lib.dmc_set_option(b"gen_wgrad_path", 4) == 0 and lib.dmc_set_option(b"gen_wgrad_path", 5) == 0
    assert lib.dmc_set_option(b"gen_fused", 0) == 0 and lib.dmc_set_option(b"gen_fused", 1) == 0
    # grid_reserve_cus (round 6): CUs every persistent grid leaves idle (room for RCCL's channel kernels during the backward
    # pass).  Host side: the value is validated, buffer-size queries do NOT shrink with it (a buffer sized before the option
    # changed stays large enough), the default is 0.
    assert lib.dmc_get_option(b"grid_reserve_cus") == 0
    sizes = (lib.dmc_gen_tiny_mse_partials_bytes(), lib.dmc_gen_tiny_partials_bytes(120, 224, 224), lib.dmc_gen_tiny_workspace_bytes())
    for bad in (-1, 129, 1000):
        assert lib.dmc_set_option(b"grid_reserve_cus", bad) != 0 and lib.dmc_get_option(b"grid_reserve_cus") == 0
    assert lib.dmc_set_option(b"grid_reserve_cus", 32) == 0 and lib.dmc_get_option(b"grid_reserve_cus") == 32
    assert sizes == (lib.dmc_gen_tiny_mse_partials_bytes(), lib.dmc_gen_tiny_partials_bytes(120, 224, 224), lib.dmc_gen_tiny_workspace_bytes())
    assert lib.dmc_set_option(b"grid_reserve_cus", 0) == 0
    assert lib.dmc_get_option(b"no_such_option") == -1 and lib.dmc_set_option(b"no_such_option", 1) != 0
    for cin, cout, want in [(64, 64, 1), (64, 128, 1), (512, 512, 1), (256, 64, 1), (3, 64, 0), (16, 32, 0), (96, 64, 0)]:
        assert lib.dmc_conv_nhwc_presplit_supported(cin, cout) == want, (cin, cout)
    _lib.check(lib.dmc_set_option(b"conv_arith", 0), "dmc_set_option")
    try:
        assert lib.dmc_conv_nhwc_presplit_supported(64, 64) == 0                     # fp32-MFMA arithmetic: no slices
    finally:
        _lib.check(lib.dmc_set_option(b"conv_arith", 1), "dmc_set_option")
    assert lib.dmc_conv_nhwc_wt_bytes(64, 128, 3, 3) == 3 * 2 * 64 * 128 * 9         # three bf16 slices
    assert lib.dmc_conv_nhwc_split(None, None, None, 64, 64, 3, 3, None) == -1
    assert b"dmc_conv_nhwc_split" in lib.dmc_last_error()
    dummy = torch.zeros(16)
    p = _lib.ptr(dummy)
    assert lib.dmc_conv_nhwc_split(p, p, p, 16, 32, 3, 3, None) == -1                # not a bf16x3 shape
    assert lib.dmc_conv_nhwc_dgrad_add(p, p, p, None, p, 1, 8, 8, 64, 64, 3, 3, 1, 1, None) == -1     # no addend
    assert lib.dmc_conv_nhwc_dgrad_add(p, p, p, p, p, 1, 8, 8, 64, 64, 3, 3, 2, 1, None) == -1        # stride 2
    assert b"stride 1" in lib.dmc_last_error()
    assert lib.dmc_conv_nhwc_dgrad_add(p, p, p, p, p, 1, 8, 8, 16, 32, 3, 3, 1, 1, None) == -1        # not a bf16x3 shape
    torch.manual_seed(0)
    unit = resnet.ResidualUnit("basic", 16, 16, 1).train()
    x0 = torch.randn(2, 16, 6, 6)
    grads = []
    for linked in (True, False):
        old = resnet.RESIDUAL_GRAD_LINK
        resnet.RESIDUAL_GRAD_LINK = linked
        try:
            unit.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            unit(x).square().sum().backward()
            grads.append([x.grad.clone()] + [q.grad.clone() for q in unit.parameters()])
        finally:
            resnet.RESIDUAL_GRAD_LINK = old
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_stride2_pair_host_side_without_gpu():
    """Host-only parts of the round-4 stride-2 block pair (conv_x3q.hip; torchvision BasicBlock conv1 + downsample behind
    code/dmcnet/model.py:305): which blocks qualify (ops.s2_pair_usable -- asked by the consumer AND by the producer of the
    block input, which then writes space-to-depth slices and no fp32 tensor), size / capability queries, argument validation
    before any launch, and the stock-op fallback of such a block on the CPU."""
    from dmcnet_amd import ops, resnet
    lib = _lib.load()
    # capability / sizes: the classifier's three pairs at 120 frames, and shapes the kernels do not cover
    for n, oh, cin, cout in ((120, 28, 64, 128), (120, 14, 128, 256), (120, 7, 256, 512)):
        assert lib.dmc_x3q_supported(n, oh, oh, cin, cout) == 1 and lib.dmc_x3q_conv_wgrad_supported(n, oh, oh, cin, cout) == 1
        assert lib.dmc_x3q_wpack_bytes(cin, cout) == cin * cout * 10 * 6              # nine taps + the shortcut, three bf16 slices
        assert lib.dmc_x3q_conv_wgrad_bytes(n, oh, oh, cin, cout) > 0
        m = n * oh * oh
        assert lib.dmc_x3q_stat_blocks(n, oh, oh, cout) in ((m + 255) // 256, (m + 127) // 128)
    assert lib.dmc_x3q_supported(2, 14, 14, 48, 64) == 0                               # Cin % 64
    assert lib.dmc_x3q_supported(2, 14, 300, 64, 64) == 0                              # a 256-pixel tile's patch does not fit the LDS
    assert lib.dmc_x3q_conv_wgrad_supported(2, 10, 10, 64, 64) == 0                    # no weight-gradient configuration for this grid
    dummy = torch.zeros(16)
    p = _lib.ptr(dummy)
    assert lib.dmc_x3q_conv_fwd(p, p, p, p, None, None, 0, 2, 14, 14, 48, 64, None) != 0 and b"unsupported shape" in lib.dmc_last_error()
    assert lib.dmc_x3q_conv_fwd(p, p, p, p, p, p, 1, 120, 28, 28, 64, 128, None) != 0 and b"rows" in lib.dmc_last_error()
    assert lib.dmc_x3q_split(p, p, 1, 7, 8, 16, None) != 0                             # odd height
    assert lib.dmc_x3q_pack_weights(p, None, p, p, 64, 64, None) != 0
    # eligibility of a block
    torch.manual_seed(0)
    unit = resnet.ResidualUnit("basic", 64, 128, 2).train()
    ident = resnet.ResidualUnit("basic", 64, 64, 1).train()
    assert ops.s2_pair_usable((4, 64, 28, 28), unit)
    assert not ops.s2_pair_usable((4, 64, 28, 28), ident)                              # no downsample branch
    assert not ops.s2_pair_usable((4, 64, 27, 28), unit)                               # odd height
    assert not ops.s2_pair_usable((4, 32, 28, 28), unit)                               # channel mismatch
    with torch.no_grad():
        assert not ops.s2_pair_usable((4, 64, 28, 28), unit)                           # a training op: autograd must be on
    unit.downsample[1].eval()
    assert not ops.s2_pair_usable((4, 64, 28, 28), unit)                               # BatchNorms in mixed modes
    unit.train()
    old = ops.X3Q
    ops.X3Q = False
    try:
        assert not ops.s2_pair_usable((4, 64, 28, 28), unit)
    finally:
        ops.X3Q = old
    _lib.check(lib.dmc_set_option(b"conv_arith", 0), "dmc_set_option")
    try:
        assert not ops.s2_pair_usable((4, 64, 28, 28), unit)                           # fp32-MFMA arithmetic: no slice tensors
    finally:
        _lib.check(lib.dmc_set_option(b"conv_arith", 1), "dmc_set_option")
    # on the CPU the block runs on the stock modules (no s2d slices are attached, nothing raises)
    x = torch.randn(2, 64, 8, 8, requires_grad=True)
    out = unit(x)
    out.square().mean().backward()
    assert out.shape == (2, 128, 4, 4) and x.grad is not None and ops.x3q_of(out) is None


def test_small_channel_conv_dispatch_host_side_without_gpu():
    """conv_small.hip (the discriminator's 16- / 32-channel stride-1 3x3 blocks, code/dmcnet_GAN/model.py:254-279) is
    dispatched INSIDE dmc_conv_nhwc_fwd / _dgrad: the size queries a caller makes first must describe that kernel -- the
    statistics partials have one row per persistent workgroup, the weight workspace holds the k-padded fragments."""
    lib = _lib.load()
    # 16 channels: five k-blocks of 32 (two taps x 16 channels, the last half empty) x 3 slices x 16 rows x 32 x 2 bytes
    assert lib.dmc_conv_nhwc_wt_bytes(16, 16, 3, 3) == 3 * 5 * 16 * 32 * 2
    assert lib.dmc_conv_nhwc_wt_bytes(32, 32, 3, 3) == 3 * 9 * 2 * 16 * 32 * 2 == 32 * 32 * 9 * 6
    # 16 -> 32 (the stride-2 block): five k-blocks x two row tiles x 3 slices of 1 KB fragments (forward); the data gradient's 27 fit
    assert lib.dmc_conv_nhwc_wt_bytes(16, 32, 3, 3) == 3 * 5 * 2 * 1024 >= 16 * 32 * 9 * 6
    assert lib.dmc_conv_nhwc_wt_bytes(32, 64, 3, 3) == 32 * 64 * 9 * 6                 # (not a small-channel forward shape)
    tiles16 = 240 * 112 * 7                                                             # 16-pixel row segments
    assert lib.dmc_conv_nhwc_stat_blocks(240, 112, 112, 16, 16, 3, 1, 1) == min(2048, (tiles16 + 3) // 4)
    assert lib.dmc_conv_nhwc_stat_blocks(2, 5, 3, 32, 32, 3, 1, 1) == (2 * 5 * 1 + 7) // 8
    # the 16 -> 32 stride-2 block on even maps: one row per persistent workgroup of four waves over the 16-pixel output segments
    assert lib.dmc_conv_nhwc_stat_blocks(240, 112, 112, 16, 32, 3, 2, 1) == min(2048, (240 * 56 * 4 + 3) // 4)
    # odd sizes / other channel pairs keep the implicit-GEMM kernels' tile rows
    assert lib.dmc_conv_nhwc_stat_blocks(2, 13, 11, 16, 32, 3, 2, 1) == (2 * 7 * 6 + 127) // 128
    assert lib.dmc_conv_nhwc_stat_blocks(240, 56, 56, 32, 64, 3, 2, 1) == (240 * 28 * 28 + 255) // 256
    _lib.check(lib.dmc_set_option(b"conv_cfg", 301), "dmc_set_option")                 # 301: the small-channel kernel off
    try:
        assert lib.dmc_conv_nhwc_stat_blocks(240, 112, 112, 16, 16, 3, 1, 1) == (240 * 112 * 112 + 255) // 256
    finally:
        _lib.check(lib.dmc_set_option(b"conv_cfg", 0), "dmc_set_option")


def test_build_manifest_describes_the_library_on_disk():
    """build() decides by CONTENT: the manifest next to the library holds the sha256 of every source it was built from and of the
    library itself; an unchanged tree is recognised as up to date (no compiler run), and the record says which it was."""
    import importlib.util
    import __graft_entry__ as G
    G.build()
    spec = importlib.util.spec_from_file_location("_dmc_build_t", os.path.join(G.ROOT, "dmc-net_amd", "build.py"))
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    m = B.read_manifest()
    assert m is not None and m["sources"] == B.source_hashes() and m["library_sha256"] == B._sha(B.LIB)
    assert any(k.endswith("coviar_post.hip") for k in m["sources"]) and "include/dmcnet_hip.h" in m["sources"]
    assert not B._stale()
    B.build_library()
    assert B.LAST_BUILD["action"].startswith("up to date")
