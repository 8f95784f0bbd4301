"""Time dmc_stem3d_bf16_wgrad on the I3D micro-step's shape (3 clips x 64 frames x 224^2): the plane form (default) against the first
form (conv_cfg 11), and check both against each other.  python tools/stem3d_wgrad_bench.py [N T H W]"""
import json
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401  (registers the package name)
import dmcnet_amd
from dmcnet_amd import _lib

shape = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [3, 64, 224, 224]
n, t, h, w = shape
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(3)
x = torch.randn(n, 2, t, h, w, device=dev)
od, oh, ow = (t - 2) // 2 + 1, (h - 2) // 2 + 1, (w - 2) // 2 + 1
dy = torch.randn(n, od, oh, ow, 64, device=dev).bfloat16()
ws = torch.empty(lib.dmc_stem3d_bf16_wgrad_workspace_bytes(n, t, h, w), dtype=torch.uint8, device=dev)
out = {}
res = {}
for name, cfg in (("planes", 0), ("scatter", 11)):
    _lib.check(lib.dmc_set_option(b"conv_cfg", cfg), "set")
    dw = torch.empty(64, 2, 7, 7, 7, device=dev)
    def run():
        _lib.check(lib.dmc_stem3d_bf16_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(ws), n, t, h, w, None), "wgrad")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    out[name] = {"ms_per_call": round(e0.elapsed_time(e1) / 20, 4)}
    res[name] = dw.clone()
_lib.check(lib.dmc_set_option(b"conv_cfg", 0), "set")
d = (res["planes"].double() - res["scatter"].double()).abs().max() / res["scatter"].double().abs().max()
out["max_rel_diff_between_forms"] = float(d)
flops = 2.0 * n * od * oh * ow * 64 * 686
out["useful_TFLOPs_planes"] = round(flops / out["planes"]["ms_per_call"] / 1e9, 1)
print(json.dumps(out))
