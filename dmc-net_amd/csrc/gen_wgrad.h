// Row-sliding weight gradient of EstimatorDenseNetTiny (gen_wgrad.hip): interface to gen_tiny.hip.
#pragma once
#include "dmc_common.h"

namespace dmc {

// partial layout of one workgroup: 31 accumulator tiles of 16 x 16 (tile A: 9 column tiles + its bias tile, tile B: 21 column tiles),
// element (row r, column j) of tile `slot` at slot * 256 + r * 16 + j.  Column tile 3 gt + dx holds the columns g = 16 gt + j with
// g = 33 dy + p (p = physical input channel), g = 99 = the ones column (bias, read at dx = 1); tile A owns gt = 0, 2, 4.
constexpr int WR_NA = 10, WR_NB = 21, WR_WPART = (WR_NA + WR_NB) * 256;

// shapes the kernel serves (rows of 16-byte chunks)
bool gen_wgrad_rs_supported(int H, int W);
// the number of workgroups (= partials) a launch writes for this shape, <= max_groups
int gen_wgrad_rs_groups(int N, int H, int W, int max_groups);
// partials[g][WR_WPART], g < groups: sum over this workgroup's pixels of (gradient plane) x (input plane shifted by the tap);
// zero: >= 16 bytes of zeros in device memory (what a chunk outside the image reads)
int gen_wgrad_rs(const float* mv, const float* res, const float* feat, const float* gout, const float* gbuf, const float* zero,
                 float* partials, int N, int H, int W, int groups, hipStream_t s);

}  // namespace dmc
