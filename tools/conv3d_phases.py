"""Clock counts per phase of conv3d_bf16_kernel's consumer wave 0 (a library built with -DC3D_TIMING writes them over the statistics
partials; results wrong).  DMC_HIP_LIB=<that library> python tools/conv3d_phases.py [N Cin D HW Cout]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
import dmcnet_amd
from dmcnet_amd import _lib
n, cin, d, hw, cout = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [3, 480, 16, 14, 192]
lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.randn(n, d, hw, hw, cin, device=dev).bfloat16()
w = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.05
y = torch.empty(n, d, hw, hw, cout, device=dev, dtype=torch.bfloat16)
nblk = lib.dmc_conv3d_bf16_stat_blocks_k(n, d, hw, hw, cin, cout, 1, 1, 1)
part = torch.zeros(nblk, cout, 2, device=dev)
wp = torch.empty(lib.dmc_conv3d_bf16_wpack_bytes(cin, cout, 1, 1, 1), dtype=torch.uint8, device=dev)
def run():
    _lib.check(lib.dmc_conv3d_bf16_fwd(_lib.ptr(x), _lib.ptr(w), cin, 1, 1, _lib.ptr(wp), _lib.ptr(y), _lib.ptr(part), n, d, hw, hw, cin, cout, 1, 1, 1, None), "fwd")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("ms per call (pack + conv)", e0.elapsed_time(e1) / 20, "partial rows", nblk)
p = part.view(nblk, -1)
ny = (cout + 63) // 64
names = ["setup + first step landed", "fragments + matrix instructions", "barrier (waiting for the loaders)", "epilogue"]
vals = torch.stack([p[:, 8 * j: 8 * j + 4] for j in range(min(ny, 3))]).double().mean((0, 1))
print({names[k]: int(vals[k]) for k in range(4)}, "total", int(vals.sum()))
# build that library on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DC3D_TIMING -I include -I dmc-net_amd/csrc \
#   -c dmc-net_amd/csrc/conv3d_bf16.hip -o /tmp/c3t.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_c3t.so /tmp/c3t.o <the other objects>
