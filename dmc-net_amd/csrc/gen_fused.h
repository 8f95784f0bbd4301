// EstimatorDenseNetTiny forward as ONE launch (gen_fused.hip): interface to gen_tiny.hip.
#pragma once
#include "dmc_common.h"

namespace dmc {

// shapes the fused forward serves: any H, W <= 224 (one strip up to 118 columns, two strips above)
bool gen_fused_supported(int H, int W);
// upper bound of the per-wave loss partials one launch writes (doubles)
int gen_fused_max_partials();
// prm: the six weights [Cout][Cin][3][3] and six biases as PyTorch holds them (no repack).  out = predict_flow(...) [+ mv]; feat ([N][28][H][W], may be null: inference, nothing saved) receives y0 .. y4;
// flow != null: sum((out - flow)^2) per wave into mse_part[0 .. *nparts)
int gen_fused_fwd(const float* mv, const float* res, float* feat, float* out, const ParamPtrs& prm, const float* flow,
                  double* mse_part, int* nparts, int N, int H, int W, int add_mv, hipStream_t s);

// g_0 .. g_4 (gbuf, [N][28][H][W]) = the data gradient of every feature group from dL/dout and the saved features, one launch
int gen_fused_bwd_data(const float* gout, const float* feat, float* gbuf, const float* pk, int N, int H, int W, hipStream_t s);

}  // namespace dmc
