"""Clock counts per phase of stem3d_dgrad_blk_kernel (a library built with -DD3_TIMING writes them over the head of dx; results are
wrong in that build).  DMC_HIP_LIB=<that library> python tools/stem3d_dgrad_phases.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
import dmcnet_amd
from dmcnet_amd import _lib
n, t, h, w = 3, 64, 224, 224
lib = _lib.load()
dev = torch.device("cuda:0")
wt = torch.randn(64, 2, 7, 7, 7, device=dev) * 0.05
dy = torch.randn(n, 32, 112, 112, 64, device=dev).bfloat16()
ws = torch.empty(lib.dmc_stem3d_bf16_dgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
dx = torch.empty(n, 2, t, h, w, device=dev)
for _ in range(3):
    _lib.check(lib.dmc_stem3d_bf16_dgrad(_lib.ptr(dy), _lib.ptr(wt), _lib.ptr(dx), _lib.ptr(ws), n, t, h, w, None), "dgrad")
torch.cuda.synchronize()
part = dx.view(-1)[:256 * 4 * 8].view(256, 4, 8)[:, :, :6].double()
names = ["block prologue", "staging (store row s+1, request row s+3)", "fragments + matrix instructions", "barrier", "fold", "block-end barrier"]
m = part.mean(0)
for u in range(4):
    print("wave", u, {names[k]: int(m[u, k]) for k in range(6)}, "total", int(m[u].sum()))
# build that library on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DD3_TIMING -I include -I dmc-net_amd/csrc \
#   -c dmc-net_amd/csrc/stem3d_bf16.hip -o /tmp/s3t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_d3t.so /tmp/s3t.o <the other objects>
