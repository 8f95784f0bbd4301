// EstimatorDenseNetTiny hidden layers in bf16x3 arithmetic on v_mfma_f32_16x16x32_bf16 (gen_x3.hip): interface to gen_tiny.hip.
#pragma once
#include "dmc_common.h"

namespace dmc {

// bytes of the packed weight fragments (all layers this file serves), kept behind the fp32 parameter block of the workspace
size_t gen_x3_frag_bytes();
// layers this path serves (forward hidden layers): K in [0, 3)
bool gen_x3_supported(int K, int H, int W);
// pk: the packed fp32 parameters (WF | BF | ..., dmc_common.h); frags: gen_x3_frag_bytes() bytes, 16-byte aligned
int gen_x3_pack(const float* pk, void* frags, hipStream_t s);
// y_K = LeakyReLU(0.1)(conv3x3(cat(mv, res, y_0 .. y_{K-1})) + b_K) for frames [0, N): feat is [N][28][H][W]
int gen_x3_layer(int K, const float* mv, const float* res, float* feat, const float* pk, const void* frags, int N, int H, int W,
                 hipStream_t s);

}  // namespace dmc
