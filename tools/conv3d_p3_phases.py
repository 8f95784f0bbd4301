import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import dmcnet_amd
from dmcnet_amd import _lib as L
from dmcnet_amd.ops import _stream
lib = L.load()
assert lib.dmc_get_option(b"measure_build") == 1
lib.dmc_set_option(b"conv_ablate", 128)
for name, n, cin, d, hw, cout in (("2c", 3, 64, 32, 56, 192), ("3b.b1", 3, 96, 32, 28, 128), ("3c.b1", 3, 128, 32, 28, 192), ("4f.b1", 3, 160, 16, 14, 320)):
    x = torch.randn(n, d, hw, hw, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    y = torch.empty(n, d, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
    wp = torch.empty(lib.dmc_conv3d_bf16_wpack_bytes(cin, cout, 3, 3, 3), dtype=torch.uint8, device="cuda")
    nb = lib.dmc_conv3d_bf16_stat_blocks_k(n, d, hw, hw, cin, cout, 3, 3, 3)
    st = torch.zeros(nb * cout * 2, device="cuda")
    T = 27
    for it in range(3):
        L.check(lib.dmc_conv3d_bf16_fwd(L.ptr(x), L.ptr(w), cin * T, T, 1, L.ptr(wp), L.ptr(y), L.ptr(st), n, d, hw, hw, cin, cout, 3, 3, 3, _stream()), "fwd")
    torch.cuda.synchronize()
    t = st[:256 * 4 * 4].view(256, 4, 4).cpu()
    tot = t[:, :, 3].mean().item()
    print("%-6s per consumer wave (clocks, mean over 256 workgroups x 4 waves): compute %.0f  wait %.0f  epilogue %.0f  total %.0f   (%.0f%% / %.0f%% / %.0f%%)" % (
        name, t[:, :, 0].mean(), t[:, :, 1].mean(), t[:, :, 2].mean(), tot, 100 * t[:, :, 0].mean() / tot, 100 * t[:, :, 1].mean() / tot, 100 * t[:, :, 2].mean() / tot))
