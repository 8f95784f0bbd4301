"""Time dmc_stem3d_bf16_dgrad on the I3D micro-step's shape (3 clips x 64 frames x 224^2): the block form (default) against the row
kernels (conv_cfg 12), and check both against each other.  python tools/stem3d_dgrad_bench.py [N T H W]"""
import json
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
import dmcnet_amd
from dmcnet_amd import _lib

shape = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else [3, 64, 224, 224]
n, t, h, w = shape
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(3)
wt = torch.randn(64, 2, 7, 7, 7, device=dev) * 0.05
od, oh, ow = (t - 2) // 2 + 1, (h - 2) // 2 + 1, (w - 2) // 2 + 1
dy = torch.randn(n, od, oh, ow, 64, device=dev).bfloat16()
ws = torch.empty(lib.dmc_stem3d_bf16_dgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
out, res = {}, {}
for name, cfg in (("block", 0), ("rows", 12)):
    _lib.check(lib.dmc_set_option(b"conv_cfg", cfg), "set")
    dx = torch.empty(n, 2, t, h, w, device=dev)
    def run():
        _lib.check(lib.dmc_stem3d_bf16_dgrad(_lib.ptr(dy), _lib.ptr(wt), _lib.ptr(dx), _lib.ptr(ws), n, t, h, w, None), "dgrad")
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
    res[name] = dx.clone()
_lib.check(lib.dmc_set_option(b"conv_cfg", 0), "set")
out["max_rel_diff_between_forms"] = float((res["block"].double() - res["rows"].double()).abs().max() / res["rows"].double().abs().max())
out["useful_TFLOPs_block"] = round(2.0 * n * od * oh * ow * 64 * 686 / out["block"]["ms_per_call"] / 1e9, 1)
print(json.dumps(out))
