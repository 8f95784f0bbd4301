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
    assert lib.dmc_gen_tiny_workspace_bytes() == (7788 + 256) * 4
    assert lib.dmc_gen_tiny_saved_bytes(120, 224, 224) == 120 * 28 * 224 * 224 * 4
    assert lib.dmc_gen_tiny_partials_bytes(1, 8, 32) == (1 + 16) * 28 * 256 * 4
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
