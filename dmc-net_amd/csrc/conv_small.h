// Internal interface of conv_small.hip (16- / 32-channel 3x3 stride-1 convolutions in bf16x3 arithmetic); called from the
// dmc_conv_nhwc_* entry points of conv_nhwc.hip, which dispatch the shapes it covers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace dmc {
bool csm_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int csm_stat_blocks(int N, int H, int W, int C);
size_t csm_wpack_bytes(int C);
int csm_fwd(const float* x, const float* w, void* wpack, const float* bias, const float* keep, float* y, double* stat_part,
            int stat_blocks, int N, int H, int W, int C, int act, hipStream_t s);
// weight gradient: workspace >= csm_wgrad_groups() x C x 9 x C floats (dmc_conv_nhwc_wgrad_bytes); dw [C][3][3][C] (OHWI)
int csm_wgrad_groups(int N, int H, int W, int C);
int csm_wgrad(const float* x, const float* g, float* dw, float* workspace, int N, int H, int W, int C, hipStream_t s);
// forward of Conv2d(16, 32, 3, 2, 1) on even-sized maps (bias / LeakyReLU(0.2) / keep mask / fp64 statistics partials as csm_fwd)
bool csm_fwd_s2_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int csm_fwd_s2_stat_blocks(int N, int H, int W);
size_t csm_fwd_s2_wpack_bytes();
int csm_fwd_s2(const float* x, const float* w, void* wpack, const float* bias, const float* keep, float* y, double* stat_part,
               int stat_blocks, int N, int H, int W, int act, hipStream_t s);
// data gradient of the same block: dx [N][H][W][16] from dy [N][H/2][W/2][32] in ONE launch (all four parity classes)
bool csm_dgrad_s2_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
size_t csm_dgrad_s2_wpack_bytes();
int csm_dgrad_s2(const float* dy, const float* w, void* wpack, float* dx, int N, int H, int W, hipStream_t s);
// weight gradient of the 3 x 3 / stride-2 / padding-1 blocks with Cout = 2 Cin (Cin 16 or 32, even H and W): workspace >= 512 x
// Cout x 9 x Cin floats (dmc_conv_nhwc_wgrad_bytes); dw [Cout][3][3][Cin] (OHWI)
bool csm_wgrad_s2_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int csm_wgrad_s2(const float* x, const float* g, float* dw, float* workspace, int N, int H, int W, int Cin, hipStream_t s);
int csm_dgrad(const float* dy, const float* w, void* wpack, float* dx, int N, int H, int W, int C, hipStream_t s);
}  // namespace dmc
