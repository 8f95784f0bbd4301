// Unit3Dpy as ONE C-ABI call each way (host code only: it sequences the entry points of conv3d_bf16.hip and bn3d_bf16.hip).
//
// The I3D trunk (/root/reference/code/dmcnet_I3D/network/i3d.py:328-403: Conv3d -> BatchNorm3d -> ReLU, 57 units) issues
// ~940 launches per micro-step and is bound by the host that enqueues them: per unit the Python side made three foreign
// calls and six allocations each way.  dmc_unit3d_bf16_fwd / _bwd carve every intermediate (statistics partials, both packed
// weight layouts, the statistics, the BatchNorm-backward scratch, the weight gradient's partials) out of ONE workspace per
// direction and run the same kernels in the same order: identical results, a third of the host work.
#include <hip/hip_runtime.h>

#include "dmc_common.h"
#include "dmcnet_hip.h"

using namespace dmc;

namespace {

size_t up256(size_t b) { return (b + 255) / 256 * 256; }

struct FwdLayout { size_t part, wpack_f, wpack_b, stats, total; int nblk; };
FwdLayout fwd_layout(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW) {
    FwdLayout l;
    l.nblk = dmc_conv3d_bf16_stat_blocks_k(N, D, H, W, Cin, Cout, KD, KH, KW);
    const size_t wp = up256(dmc_conv3d_bf16_wpack_bytes(Cin, Cout, KD, KH, KW));
    l.part = 0;
    l.wpack_f = up256((size_t)l.nblk * Cout * 2 * sizeof(float));
    l.wpack_b = l.wpack_f + wp;
    l.stats = l.wpack_b + wp;
    l.total = l.stats + up256((size_t)2 * Cout * sizeof(float));
    return l;
}

}  // namespace

extern "C" {

size_t dmc_unit3d_bf16_fwd_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW) {
    return fwd_layout(N, D, H, W, Cin, Cout, KD, KH, KW).total;
}

size_t dmc_unit3d_bf16_bwd_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW) {
    return up256(dmc_bn3d_bf16_scratch_bytes(Cout)) + up256(dmc_conv3d_bf16_wgrad_bytes(N, D, H, W, Cin, Cout, KD, KH, KW));
}

// out_ld: elements between consecutive pixels of `out` (Cout: a dense tensor; more: a channel slice of a wider NDHWC tensor)
int dmc_unit3d_bf16_fwd_into(const void* x, const float* w, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, void* workspace, void* y, void* out, long out_ld, int N, int D, int H, int W, int Cin,
                             int Cout, int KD, int KH, int KW, int relu, float eps, float momentum, dmc_stream_t stream) {
    if (!x || !w || !gamma || !beta || !workspace || !y || !out) return fail(DMC_E_INVALID, "dmc_unit3d_bf16_fwd: null pointer");
    const FwdLayout l = fwd_layout(N, D, H, W, Cin, Cout, KD, KH, KW);
    char* ws = static_cast<char*>(workspace);
    const int T = KD * KH * KW;
    int rc = dmc_conv3d_bf16_pack(w, (long)Cin * T, T, 1, ws + l.wpack_f, ws + l.wpack_b, Cin, Cout, KD, KH, KW, stream);
    if (rc) return rc;
    rc = dmc_conv3d_bf16_fwd(x, nullptr, (long)Cin * T, T, 1, ws + l.wpack_f, y, reinterpret_cast<float*>(ws + l.part), N, D, H, W,
                             Cin, Cout, KD, KH, KW, stream);
    if (rc) return rc;
    return dmc_bn3d_bf16_fwd_ld(y, reinterpret_cast<const float*>(ws + l.part), l.nblk, gamma, beta, reinterpret_cast<float*>(ws + l.stats),
                                running_mean, running_var, out, out_ld, (long)N * D * H * W, Cout, relu, eps, momentum, stream);
}

int dmc_unit3d_bf16_fwd(const void* x, const float* w, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, void* workspace, void* y, void* out, int N, int D, int H, int W, int Cin, int Cout,
                        int KD, int KH, int KW, int relu, float eps, float momentum, dmc_stream_t stream) {
    return dmc_unit3d_bf16_fwd_into(x, w, gamma, beta, running_mean, running_var, workspace, y, out, Cout, N, D, H, W, Cin, Cout, KD, KH,
                                    KW, relu, eps, momentum, stream);
}

int dmc_unit3d_bf16_bwd(const void* dout, long dout_ld, const void* x, const void* y, const void* fwd_workspace, const float* gamma,
                        const float* beta, void* bwd_workspace, void* dy, void* dx, float* dw, float* dgamma, float* dbeta, int N,
                        int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW, int relu, dmc_stream_t stream) {
    if (!dout || !x || !y || !fwd_workspace || !gamma || !beta || !bwd_workspace || !dy || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_unit3d_bf16_bwd: null pointer");
    const FwdLayout l = fwd_layout(N, D, H, W, Cin, Cout, KD, KH, KW);
    const char* fw = static_cast<const char*>(fwd_workspace);
    char* bw = static_cast<char*>(bwd_workspace);
    const int T = KD * KH * KW;
    int rc = dmc_bn3d_bf16_bwd(dout, dout_ld, y, reinterpret_cast<const float*>(fw + l.stats), gamma, beta, reinterpret_cast<float*>(bw),
                               dy, dgamma, dbeta, (long)N * D * H * W, Cout, relu, stream);
    if (rc) return rc;
    if (dx) {       // weights packed by the forward's launch
        rc = dmc_conv3d_bf16_dgrad(dy, nullptr, (long)Cin * T, T, 1, const_cast<char*>(fw + l.wpack_b), dx, N, D, H, W, Cin, Cout, KD, KH,
                                   KW, stream);
        if (rc) return rc;
    }
    if (dw)
        rc = dmc_conv3d_bf16_wgrad(x, dy, dw, reinterpret_cast<float*>(bw + up256(dmc_bn3d_bf16_scratch_bytes(Cout))), N, D, H, W, Cin,
                                   Cout, KD, KH, KW, stream);
    return rc;
}

}  // extern "C"
