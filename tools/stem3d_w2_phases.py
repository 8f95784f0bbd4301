"""Clock counts per phase of stem3d_w2_kernel (a library built with -DW2_TIMING overwrites the head of every wave's partial with them;
results are wrong in that build).  DMC_HIP_LIB=<that library> python tools/stem3d_w2_phases.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
import dmcnet_amd
from dmcnet_amd import _lib
n, t, h, w = 3, 64, 224, 224
lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.randn(n, 2, t, h, w, device=dev)
dy = torch.randn(n, 32, 112, 112, 64, device=dev).bfloat16()
ws = torch.empty(lib.dmc_stem3d_bf16_wgrad_workspace_bytes(n, t, h, w), dtype=torch.uint8, device=dev)
dw = torch.empty(64, 2, 7, 7, 7, device=dev)
for _ in range(3):
    _lib.check(lib.dmc_stem3d_bf16_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(ws), n, t, h, w, None), "wgrad")
torch.cuda.synchronize()
off = n * (t + 5) * (h + 5) * 1024
part = ws[off:off + 256 * 4 * 224 * 64 * 4].view(torch.float32).view(256, 4, 224 * 64)[:, :, :8].double()
names = ["prologue", "first fragments landed", "k loop", "dy store + wait", "barrier", "transfers issued", "dy loads issued", "addresses + first fragment reads issued"]
m = part.mean(0)
for u in range(4):
    print("wave", u, {names[k]: int(m[u, k]) for k in range(8)}, "total", int(m[u].sum()))
print("(s_memtime ticks at 100 MHz: x ~21-24 for shader clocks; 42 rows per workgroup)")
# build that library on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DW2_TIMING -I include -I dmc-net_amd/csrc \
#   -c dmc-net_amd/csrc/stem3d_bf16.hip -o /tmp/s3t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_w2t.so /tmp/s3t.o <the other objects>
