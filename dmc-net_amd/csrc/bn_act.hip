// Fused BatchNorm2d [+ residual add] [+ ReLU] for NHWC activations (gfx950), forward and backward.
//
// Replaces, inside the ResNet `base_model` the reference takes from torchvision
// (code/dmcnet/model.py:305,352): `bn(x)`, `relu(bn(x))` and `relu(bn(x) + identity)` of the stem
// and of every BasicBlock/Bottleneck, i.e. nn.BatchNorm2d (train: batch statistics, biased
// variance for normalisation, unbiased for running_var, momentum 0.1) followed by the in-place
// ReLU and the residual addition.  These ops are pure HBM streaming; fusing them removes the
// intermediate tensors (forward: 3-4 passes over the activation instead of 6; backward: 5-7
// instead of 8-9).  The ReLU mask is recomputed from x in the backward pass, so the output does
// not have to be re-read when there is no residual.
//
// Layout: x, residual, y, dy, dx are [M][C] row-major with M = N*H*W (a channels_last tensor's
// memory).  C % 4 == 0, C/4 <= 256 and 256 % (C/4) == 0 (64, 128, 256, 512, 1024).
#include "dmc_common.h"

using namespace dmc;

namespace {

constexpr int MAX_SPLIT = BN_MAX_SPLIT;   // row-splits of the per-channel reductions (scratch is sized for this)

struct BnArgs {
    const float* x;
    const float* res;      // nullable
    const float* gamma;
    const float* beta;
    const float* stats;    // mean[C], invstd[C]
    const float* dy;
    float* y;
    float* dx;
    float* dres;           // nullable
    int M, C, relu;
    unsigned char* mask;   // nullable: 4 ReLU sign bits per float4 (written by the forward, read by the backward)
    // discriminator blocks (activation BEFORE the BatchNorm: x = keep * LeakyReLU(conv)): the backward
    // continues through the mask and the LeakyReLU, dx *= keep[n][c] * (x > 0 ? 1 : slope)
    const float* keep;     // nullable [N][C]
    int hw;                // pixels per frame (row -> n)
    float slope;           // 0: no activation derivative
};

// MODE 0: per-channel (sum x, sum x^2).   MODE 1: (sum dz, sum dz * xhat), dz = dy * relu'(.)
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(BnArgs a, double* __restrict__ scratch) {
    __shared__ double sm[2][256][4];
    const int cq = a.C >> 2;                    // float4 columns
    const int tx = threadIdx.x % cq, ty = threadIdx.x / cq, rows = 256 / cq;
    const int per = (a.M + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r0 = blockIdx.x * per, r1 = (r0 + per < a.M) ? r0 + per : a.M;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    float4 mean = s0, istd = s0, g = s0, b = s0;
    if (MODE == 1) {
        mean = reinterpret_cast<const float4*>(a.stats)[tx];
        istd = reinterpret_cast<const float4*>(a.stats + a.C)[tx];
        g = reinterpret_cast<const float4*>(a.gamma)[tx];
        b = reinterpret_cast<const float4*>(a.beta)[tx];
    }
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += rows) {
        const size_t o = (size_t)r * cq + tx;
        const float4 x = reinterpret_cast<const float4*>(a.x)[o];
        if (MODE == 0) {
            s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
            s1.x = fmaf(x.x, x.x, s1.x); s1.y = fmaf(x.y, x.y, s1.y);
            s1.z = fmaf(x.z, x.z, s1.z); s1.w = fmaf(x.w, x.w, s1.w);
        } else {
            float4 d = reinterpret_cast<const float4*>(a.dy)[o];
            const float4 xh = make_float4((x.x - mean.x) * istd.x, (x.y - mean.y) * istd.y,
                                          (x.z - mean.z) * istd.z, (x.w - mean.w) * istd.w);
            if (a.relu && a.mask) {                 // sign bits saved by the forward: no residual re-read
                const unsigned mb = a.mask[o];
                d.x = (mb & 1) ? d.x : 0.f; d.y = (mb & 2) ? d.y : 0.f;
                d.z = (mb & 4) ? d.z : 0.f; d.w = (mb & 8) ? d.w : 0.f;
            } else if (a.relu) {
                float4 v = make_float4(fmaf(xh.x, g.x, b.x), fmaf(xh.y, g.y, b.y),
                                       fmaf(xh.z, g.z, b.z), fmaf(xh.w, g.w, b.w));
                if (a.res) {
                    const float4 rr = reinterpret_cast<const float4*>(a.res)[o];
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f;
                d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
            }
            s0.x += d.x; s0.y += d.y; s0.z += d.z; s0.w += d.w;
            s1.x = fmaf(d.x, xh.x, s1.x); s1.y = fmaf(d.y, xh.y, s1.y);
            s1.z = fmaf(d.z, xh.z, s1.z); s1.w = fmaf(d.w, xh.w, s1.w);
        }
    }
    sm[0][threadIdx.x][0] = s0.x; sm[0][threadIdx.x][1] = s0.y;
    sm[0][threadIdx.x][2] = s0.z; sm[0][threadIdx.x][3] = s0.w;
    sm[1][threadIdx.x][0] = s1.x; sm[1][threadIdx.x][1] = s1.y;
    sm[1][threadIdx.x][2] = s1.z; sm[1][threadIdx.x][3] = s1.w;
    __syncthreads();
    // fixed-order column sums over the `rows` thread rows
    for (int i = threadIdx.x; i < cq * 8; i += 256) {
        const int col = i >> 3, which = (i >> 2) & 1, e = i & 3;
        double acc = 0.0;
        for (int t = 0; t < rows; ++t) acc += sm[which][t * cq + col][e];
        scratch[((size_t)(col * 4 + e) * MAX_SPLIT + blockIdx.x) * 2 + which] = acc;   // [channel][split][2]
    }
}

// Sum of the `split` partials of channel c, pair `which`: 32 lanes per channel, fixed order.
__device__ __forceinline__ void reduce_partials(const double* __restrict__ scratch, int C, int c,
                                                int split, int sub, double& s, double& ss) {
    s = 0.0; ss = 0.0;
    // scratch is [channel][split][2]: the 32 lanes of a channel read consecutive 16-byte pairs
    if (c < C) {
        const double2* row = reinterpret_cast<const double2*>(scratch) + (size_t)c * MAX_SPLIT;
#pragma unroll 4
        for (int sp = sub; sp < split; sp += 32) {
            const double2 v = row[sp];
            s += v.x;
            ss += v.y;
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 32);
        ss += __shfl_down(ss, o, 32);
    }
}

// block = 256 threads = 8 channels x 32 partial-lanes
__global__ __launch_bounds__(256) void bn_stats_final_kernel(
    const double* __restrict__ scratch, float* __restrict__ stats, float* __restrict__ running_mean,
    float* __restrict__ running_var, int C, long count, int training, float eps, float momentum,
    int split) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), sub = threadIdx.x & 31;
    if (!training) {
        if (c < C && sub == 0) {
            stats[c] = running_mean[c];
            stats[C + c] = rsqrtf(running_var[c] + eps);
        }
        return;
    }
    double s, ss;
    reduce_partials(scratch, C, c, split, sub, s, ss);
    if (c >= C || sub != 0) return;
    const double mean = s / (double)count;
    double var = ss / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

// One workgroup per channel: every thread fetches its <= 4 partials at once (one memory round trip), fixed-order tree.
// (32 lanes per channel walked the partials in 8 dependent batches: 5 us alone, but this launch sits between the two
// BatchNorm-backward passes on the step's critical path and its loads queue behind a concurrent weight gradient's --
// 31 us per launch, 20 launches per step, under ops.WGRAD_STREAM.)
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* __restrict__ scratch,
                                                           float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int C,
                                                           int split) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double2* row = reinterpret_cast<const double2*>(scratch) + (size_t)c * MAX_SPLIT;
    double2 v[MAX_SPLIT / 256];
#pragma unroll
    for (int k = 0; k < MAX_SPLIT / 256; ++k) {
        const int sp = (int)threadIdx.x + 256 * k;
        v[k] = sp < split ? row[sp] : make_double2(0.0, 0.0);
    }
    double s = 0.0, ss = 0.0;
#pragma unroll
    for (int k = 0; k < MAX_SPLIT / 256; ++k) { s += v[k].x; ss += v[k].y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); ss += __shfl_down(ss, o, 64); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    dbeta[c] = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    dgamma[c] = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
}

__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(BnArgs a) {
    const int cq = a.C >> 2;
    const size_t total = (size_t)a.M * cq;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int tx = (int)(o & (size_t)(cq - 1));     // cq is a power of two
        const float4 mean = reinterpret_cast<const float4*>(a.stats)[tx];
        const float4 istd = reinterpret_cast<const float4*>(a.stats + a.C)[tx];
        const float4 g = reinterpret_cast<const float4*>(a.gamma)[tx];
        const float4 b = reinterpret_cast<const float4*>(a.beta)[tx];
        const float4 x = reinterpret_cast<const float4*>(a.x)[o];
        float4 v = make_float4(fmaf((x.x - mean.x) * istd.x, g.x, b.x), fmaf((x.y - mean.y) * istd.y, g.y, b.y),
                               fmaf((x.z - mean.z) * istd.z, g.z, b.z), fmaf((x.w - mean.w) * istd.w, g.w, b.w));
        if (a.res) {
            const float4 rr = reinterpret_cast<const float4*>(a.res)[o];
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (a.relu) {
            if (a.mask)
                a.mask[o] = (unsigned char)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        reinterpret_cast<float4*>(a.y)[o] = v;
    }
}

__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(BnArgs a, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta,
                                                           float inv_count) {
    const int cq = a.C >> 2;
    const size_t total = (size_t)a.M * cq;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int tx = (int)(o & (size_t)(cq - 1));     // cq is a power of two
        const float4 mean = reinterpret_cast<const float4*>(a.stats)[tx];
        const float4 istd = reinterpret_cast<const float4*>(a.stats + a.C)[tx];
        const float4 g = reinterpret_cast<const float4*>(a.gamma)[tx];
        const float4 b = reinterpret_cast<const float4*>(a.beta)[tx];
        const float4 dg = reinterpret_cast<const float4*>(dgamma)[tx];
        const float4 db = reinterpret_cast<const float4*>(dbeta)[tx];
        const float4 x = reinterpret_cast<const float4*>(a.x)[o];
        float4 d = reinterpret_cast<const float4*>(a.dy)[o];
        const float4 xh = make_float4((x.x - mean.x) * istd.x, (x.y - mean.y) * istd.y,
                                      (x.z - mean.z) * istd.z, (x.w - mean.w) * istd.w);
        if (a.relu && a.mask) {
            const unsigned mb = a.mask[o];
            d.x = (mb & 1) ? d.x : 0.f; d.y = (mb & 2) ? d.y : 0.f;
            d.z = (mb & 4) ? d.z : 0.f; d.w = (mb & 8) ? d.w : 0.f;
        } else if (a.relu) {
            float4 v = make_float4(fmaf(xh.x, g.x, b.x), fmaf(xh.y, g.y, b.y), fmaf(xh.z, g.z, b.z),
                                   fmaf(xh.w, g.w, b.w));
            if (a.res) {
                const float4 rr = reinterpret_cast<const float4*>(a.res)[o];
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f;
            d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
        }
        if (a.dres) reinterpret_cast<float4*>(a.dres)[o] = d;
        float4 r;
        r.x = g.x * istd.x * (d.x - db.x * inv_count - xh.x * dg.x * inv_count);
        r.y = g.y * istd.y * (d.y - db.y * inv_count - xh.y * dg.y * inv_count);
        r.z = g.z * istd.z * (d.z - db.z * inv_count - xh.z * dg.z * inv_count);
        r.w = g.w * istd.w * (d.w - db.w * inv_count - xh.w * dg.w * inv_count);
        if (a.slope != 0.f) {
            float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.keep) k = reinterpret_cast<const float4*>(a.keep)[(size_t)((o / cq) / a.hw) * cq + tx];
            r.x *= k.x * (x.x > 0.f ? 1.f : a.slope); r.y *= k.y * (x.y > 0.f ? 1.f : a.slope);
            r.z *= k.z * (x.z > 0.f ? 1.f : a.slope); r.w *= k.w * (x.w > 0.f ? 1.f : a.slope);
        }
        reinterpret_cast<float4*>(a.dx)[o] = r;
    }
}

// ------------------------------------------------------------------------------------------
// Stem: BatchNorm + ReLU + MaxPool2d(3, stride 2, padding 1) in one pass each way.
//
// Replaces bn1 -> relu -> maxpool of the ResNet the reference builds (code/dmcnet/model.py:305;
// torchvision's `self.maxpool(self.relu(self.bn1(x)))`).  The normalised, rectified 112x112x64
// tensor (385 MB at 120 frames) is never written: the forward emits the pooled map only; the
// backward recomputes relu(bn(x)) per 3x3 window to find each window's arg-max -- PyTorch's rule:
// scan the window row-major, take a value only if it is strictly greater, so ties go to the first
// element -- routes the pooled gradient to it, and feeds the BatchNorm backward sums / dx
// directly.  Tiles of 8x8 windows: step A computes one arg-max code per (window, channel) into
// LDS (9x9 windows: the halo row/column of windows that own pixels of this tile), step B gathers
// per input pixel from the <= 4 windows that contain it, in fixed (py, px) order.
// ------------------------------------------------------------------------------------------
struct PoolArgs {
    const float* x;        // [N,H,W,C]
    const float* gamma;
    const float* beta;
    const float* stats;    // mean[C], invstd[C]
    const float* dpool;    // [N,PH,PW,C]
    float* ypool;          // [N,PH,PW,C]
    float* dx;             // [N,H,W,C]
    int N, H, W, C, PH, PW;
};

__device__ __forceinline__ float4 bn_relu4(const float4 x, const float4 mean, const float4 istd,
                                           const float4 g, const float4 b) {
    return make_float4(fmaxf(fmaf((x.x - mean.x) * istd.x, g.x, b.x), 0.f), fmaxf(fmaf((x.y - mean.y) * istd.y, g.y, b.y), 0.f),
                       fmaxf(fmaf((x.z - mean.z) * istd.z, g.z, b.z), 0.f), fmaxf(fmaf((x.w - mean.w) * istd.w, g.w, b.w), 0.f));
}

__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs a) {
    const int cq = a.C >> 2;
    const size_t total = (size_t)a.N * a.PH * a.PW * cq;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int tx = (int)(o & (size_t)(cq - 1));
        size_t r = o / cq;
        const int px = (int)(r % a.PW); r /= a.PW;
        const int py = (int)(r % a.PH);
        const int n = (int)(r / a.PH);
        const float4 mean = reinterpret_cast<const float4*>(a.stats)[tx];
        const float4 istd = reinterpret_cast<const float4*>(a.stats + a.C)[tx];
        const float4 g = reinterpret_cast<const float4*>(a.gamma)[tx];
        const float4 b = reinterpret_cast<const float4*>(a.beta)[tx];
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);           // relu(.) >= 0 and the window centre is always inside
        // clamped coordinates: a clamped position repeats a value of the same window, which cannot
        // change the maximum -- nine independent loads, no branches
        float4 xv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            int iy = 2 * py + k / 3 - 1, ix = 2 * px + k % 3 - 1;
            iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
            ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            xv[k] = reinterpret_cast<const float4*>(a.x)[((size_t)(n * a.H + iy) * a.W + ix) * cq + tx];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float4 v = bn_relu4(xv[k], mean, istd, g, b);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        reinterpret_cast<float4*>(a.ypool)[o] = m;
    }
}

constexpr int PT = 8;                          // pooled windows per tile side
// PASS 1: per-channel (sum dz, sum dz * xhat) into scratch[block];  PASS 2: dx
template <int PASS>
__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolArgs a, double* __restrict__ scratch,
                                                       const float* __restrict__ dgamma,
                                                       const float* __restrict__ dbeta, float inv_count,
                                                       unsigned* __restrict__ gcodes) {
    __shared__ unsigned codes[(PT + 1) * (PT + 1) * 64];       // arg-max position (0..8) per channel, 4 per word
    __shared__ double sm[PASS == 1 ? 2 : 1][PASS == 1 ? 256 : 1][4];
    const int cq = a.C >> 2;
    const int tx = threadIdx.x % cq, ty = threadIdx.x / cq, rows = 256 / cq;
    const float4 mean = reinterpret_cast<const float4*>(a.stats)[tx];
    const float4 istd = reinterpret_cast<const float4*>(a.stats + a.C)[tx];
    const float4 g = reinterpret_cast<const float4*>(a.gamma)[tx];
    const float4 b = reinterpret_cast<const float4*>(a.beta)[tx];
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    if (PASS == 2) {
        dg = reinterpret_cast<const float4*>(dgamma)[tx];
        db = reinterpret_cast<const float4*>(dbeta)[tx];
    }
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    const int tiles_x = (a.PW + PT - 1) / PT, tiles_y = (a.PH + PT - 1) / PT;
    const int ntiles = a.N * tiles_y * tiles_x;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / (tiles_y * tiles_x), tr = tile - n * tiles_y * tiles_x;
        const int wy0 = (tr / tiles_x) * PT, wx0 = (tr % tiles_x) * PT;
        __syncthreads();                                        // previous tile's codes are no longer read
        // ---- step A: arg-max code of windows (wy0 .. wy0+PT, wx0 .. wx0+PT) ----
        for (int wi = ty; wi < (PT + 1) * (PT + 1); wi += rows) {
            const int py = wy0 + wi / (PT + 1), px = wx0 + wi % (PT + 1);
            unsigned code = 0xffffffffu;
            if (PASS == 2 && gcodes) {
                // the arg-max codes were computed (and stored by each window's owner tile) in pass 1
                if (py < a.PH && px < a.PW) code = gcodes[((size_t)(n * a.PH + py) * a.PW + px) * cq + tx];
            } else if (py < a.PH && px < a.PW) {
                float4 mv = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                unsigned cx = 0, cy = 0, cz = 0, cw = 0;
                float4 xv[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {                    // nine independent loads (clamped address)
                    int iy = 2 * py + k / 3 - 1, ix = 2 * px + k % 3 - 1;
                    iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
                    ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
                    xv[k] = reinterpret_cast<const float4*>(a.x)[((size_t)(n * a.H + iy) * a.W + ix) * cq + tx];
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int iy = 2 * py + k / 3 - 1, ix = 2 * px + k % 3 - 1;
                    const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    const float4 v = bn_relu4(xv[k], mean, istd, g, b);
                    if (in && v.x > mv.x) { mv.x = v.x; cx = k; }
                    if (in && v.y > mv.y) { mv.y = v.y; cy = k; }
                    if (in && v.z > mv.z) { mv.z = v.z; cz = k; }
                    if (in && v.w > mv.w) { mv.w = v.w; cw = k; }
                }
                code = cx | (cy << 8) | (cz << 16) | (cw << 24);
                if (PASS == 1 && gcodes && wi / (PT + 1) < PT && wi % (PT + 1) < PT)      // owned window
                    gcodes[((size_t)(n * a.PH + py) * a.PW + px) * cq + tx] = code;
            }
            codes[wi * cq + tx] = code;
        }
        __syncthreads();
        // ---- step B: input pixels (2 wy0 .. 2 wy0 + 15, 2 wx0 .. 2 wx0 + 15) ----
        for (int pi = ty; pi < 4 * PT * PT; pi += rows) {
            const int iy = 2 * wy0 + pi / (2 * PT), ix = 2 * wx0 + pi % (2 * PT);
            if (iy >= a.H || ix >= a.W) continue;
            const size_t o = ((size_t)(n * a.H + iy) * a.W + ix) * cq + tx;
            const float4 x = reinterpret_cast<const float4*>(a.x)[o];
            const float4 xh = make_float4((x.x - mean.x) * istd.x, (x.y - mean.y) * istd.y,
                                          (x.z - mean.z) * istd.z, (x.w - mean.w) * istd.w);
            const float4 v = make_float4(fmaf(xh.x, g.x, b.x), fmaf(xh.y, g.y, b.y), fmaf(xh.z, g.z, b.z), fmaf(xh.w, g.w, b.w));
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            // windows that contain the pixel: py = iy/2 (and iy/2 + 1 when iy is odd), same in x
            const int py0 = iy >> 1, px0 = ix >> 1, npy = 1 + (iy & 1), npx = 1 + (ix & 1);
            float4 gp[4];
            unsigned cd[4];
            bool okc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                        // four independent loads (clamped), then select
                const int jy = q >> 1, jx = q & 1;
                const int py = py0 + jy, px = px0 + jx;
                okc[q] = jy < npy && jx < npx && py < a.PH && px < a.PW;
                const int pyc = okc[q] ? py : py0, pxc = okc[q] ? px : px0;
                cd[q] = codes[((pyc - wy0) * (PT + 1) + (pxc - wx0)) * cq + tx];
                gp[q] = reinterpret_cast<const float4*>(a.dpool)[((size_t)(n * a.PH + pyc) * a.PW + pxc) * cq + tx];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int py = py0 + (q >> 1), px = px0 + (q & 1);
                const unsigned k = (unsigned)((iy - 2 * py + 1) * 3 + (ix - 2 * px + 1));
                if (okc[q] && (cd[q] & 0xff) == k) d.x += gp[q].x;
                if (okc[q] && ((cd[q] >> 8) & 0xff) == k) d.y += gp[q].y;
                if (okc[q] && ((cd[q] >> 16) & 0xff) == k) d.z += gp[q].z;
                if (okc[q] && (cd[q] >> 24) == k) d.w += gp[q].w;
            }
            d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f;
            d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
            if (PASS == 1) {
                s0.x += d.x; s0.y += d.y; s0.z += d.z; s0.w += d.w;
                s1.x = fmaf(d.x, xh.x, s1.x); s1.y = fmaf(d.y, xh.y, s1.y);
                s1.z = fmaf(d.z, xh.z, s1.z); s1.w = fmaf(d.w, xh.w, s1.w);
            } else {
                float4 r;
                r.x = g.x * istd.x * (d.x - db.x * inv_count - xh.x * dg.x * inv_count);
                r.y = g.y * istd.y * (d.y - db.y * inv_count - xh.y * dg.y * inv_count);
                r.z = g.z * istd.z * (d.z - db.z * inv_count - xh.z * dg.z * inv_count);
                r.w = g.w * istd.w * (d.w - db.w * inv_count - xh.w * dg.w * inv_count);
                reinterpret_cast<float4*>(a.dx)[o] = r;
            }
        }
    }
    if (PASS == 1) {
        sm[0][threadIdx.x][0] = s0.x; sm[0][threadIdx.x][1] = s0.y;
        sm[0][threadIdx.x][2] = s0.z; sm[0][threadIdx.x][3] = s0.w;
        sm[1][threadIdx.x][0] = s1.x; sm[1][threadIdx.x][1] = s1.y;
        sm[1][threadIdx.x][2] = s1.z; sm[1][threadIdx.x][3] = s1.w;
        __syncthreads();
        for (int i = threadIdx.x; i < cq * 8; i += 256) {
            const int col = i >> 3, which = (i >> 2) & 1, e = i & 3;
            double acc = 0.0;
            for (int t = 0; t < rows; ++t) acc += sm[which][t * cq + col][e];
            scratch[((size_t)(col * 4 + e) * MAX_SPLIT + blockIdx.x) * 2 + which] = acc;   // [channel][split][2]
        }
    }
}

// dx = dz * keep[n][c] * (z > 0 ? 1 : slope): the mask + LeakyReLU derivative alone (first
// discriminator block, which has no BatchNorm)
__global__ __launch_bounds__(256) void act_mask_bwd_kernel(const float* __restrict__ z, const float* __restrict__ keep,
                                                           const float* __restrict__ dz, float* __restrict__ dx,
                                                           int M, int C, int hw, float slope) {
    const int cq = C >> 2;
    const size_t total = (size_t)M * cq;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int tx = (int)(o % cq);
        const float4 x = reinterpret_cast<const float4*>(z)[o];
        float4 d = reinterpret_cast<const float4*>(dz)[o];
        float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
        if (keep) k = reinterpret_cast<const float4*>(keep)[(size_t)((o / cq) / hw) * cq + tx];
        d.x *= k.x * (x.x > 0.f ? 1.f : slope); d.y *= k.y * (x.y > 0.f ? 1.f : slope);
        d.z *= k.z * (x.z > 0.f ? 1.f : slope); d.w *= k.w * (x.w > 0.f ? 1.f : slope);
        reinterpret_cast<float4*>(dx)[o] = d;
    }
}

// per-channel sums from bn_partial_kernel<0>'s partials (the bias gradient of a convolution)
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const double* __restrict__ scratch,
                                                                float* __restrict__ out, int C, int split) {
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), sub = threadIdx.x & 31;
    double s, ss;
    reduce_partials(scratch, C, c, split, sub, s, ss);
    if (c < C && sub == 0) out[c] = (float)s;
}

bool shape_ok(int M, int C) {
    const int cq = C / 4;
    return M > 0 && C > 0 && C % 4 == 0 && cq <= 256 && 256 % cq == 0;
}

// row-splits of a per-channel reduction: a workgroup covers 1024 / C rows per iteration; at most ~8 iterations per thread
// (two batches of four independent loads) as long as MAX_SPLIT allows -- the small late-stage maps (5,880 rows x 512
// channels) were latency-bound at 64 rows per workgroup: 91 workgroups, 32 dependent iterations each, 15 us for 24 MB
int split_of(int M, int C) {
    long sp = (long)M * C / 8192;
    return sp < 1 ? 1 : (sp > MAX_SPLIT ? MAX_SPLIT : (int)sp);
}

int stream_blocks(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}


// ---- variants that write bf16x3 slice tensors (dmc_common.h; consumed by conv_x3s.hip) -------------------------
// Eight channels per thread (two float4 in, one 16-byte store per slice).  The forward writes the normalised
// activation as fp32 (y, nullable) and / or as slices (ys, nullable): the next convolution reads the slices, only a
// residual add / pooling / the classifier head need fp32.  The backward writes the convolution's output gradient the
// same way (dx / dxs): its only readers are that convolution's data- and weight-gradient kernels.
__device__ __forceinline__ void load8(const float* p, size_t i8, float (&v)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[2 * i8], b = reinterpret_cast<const float4*>(p)[2 * i8 + 1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float* p, size_t i8, const float (&v)[8]) {
    reinterpret_cast<float4*>(p)[2 * i8] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[2 * i8 + 1] = make_float4(v[4], v[5], v[6], v[7]);
}

// s2d_h / s2d_w > 0: the slices are written SPACE-TO-DEPTH for a stride-2 consumer (conv_x3q.hip): pixel (n, y, x) of the
// H x W map goes to parity class 2 (y & 1) + (x & 1), position (n, y / 2, x / 2) of [3][4][C/16][M/4][16].
__global__ __launch_bounds__(256) void bn_apply_fwd_x3s_kernel(BnArgs a, unsigned short* __restrict__ ys, int s2d_h, int s2d_w) {
    const int c8 = a.C >> 3;
    const size_t total = (size_t)a.M * c8;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int tx = (int)(o & (size_t)(c8 - 1));     // c8 is a power of two
        const size_t m = o / c8;
        float mean[8], istd[8], g[8], b[8], x[8], v[8];
        load8(a.stats, tx, mean); load8(a.stats + a.C, tx, istd); load8(a.gamma, tx, g); load8(a.beta, tx, b);
        load8(a.x, o, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf((x[e] - mean[e]) * istd[e], g[e], b[e]);
        if (a.res) {
            float rr[8];
            load8(a.res, o, rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e];
        }
        if (a.relu) {
            if (a.mask) {
                unsigned mb = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) mb |= (v[e] > 0.f ? 1u : 0u) << e;
                reinterpret_cast<unsigned short*>(a.mask)[o] = (unsigned short)((mb & 15u) | ((mb >> 4) << 8));   // one byte per float4
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (a.y) store8(a.y, o, v);
        if (ys) {
            if (s2d_w) {
                const int xx = (int)(m % (size_t)s2d_w), yy = (int)((m / (size_t)s2d_w) % (size_t)s2d_h);
                const size_t n = m / ((size_t)s2d_w * s2d_h);
                const int cls = (yy & 1) * 2 + (xx & 1), nchunk = a.C >> 4;
                x3s_store8(ys, v, (size_t)a.M >> 2, 4 * nchunk, (n * (s2d_h >> 1) + (yy >> 1)) * (s2d_w >> 1) + (xx >> 1), tx + cls * 2 * nchunk);
            } else {
                x3s_store8(ys, v, (size_t)a.M, a.C >> 4, m, tx);
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn_apply_bwd_x3s_kernel(BnArgs a, const float* __restrict__ dgamma,
                                                               const float* __restrict__ dbeta, float inv_count,
                                                               unsigned short* __restrict__ dxs) {
    const int c8 = a.C >> 3;
    const size_t total = (size_t)a.M * c8;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int tx = (int)(o & (size_t)(c8 - 1));
        const size_t m = o / c8;
        float mean[8], istd[8], g[8], b[8], dg[8], db[8], x[8], d[8], xh[8], r[8];
        load8(a.stats, tx, mean); load8(a.stats + a.C, tx, istd); load8(a.gamma, tx, g); load8(a.beta, tx, b);
        load8(dgamma, tx, dg); load8(dbeta, tx, db);
        load8(a.x, o, x); load8(a.dy, o, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[e] = (x[e] - mean[e]) * istd[e];
        if (a.relu && a.mask) {
            const unsigned mw = reinterpret_cast<const unsigned short*>(a.mask)[o];
            const unsigned mb = (mw & 15u) | ((mw >> 8) << 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = ((mb >> e) & 1u) ? d[e] : 0.f;
        } else if (a.relu) {
            float rr[8];
            if (a.res) load8(a.res, o, rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = fmaf(xh[e], g[e], b[e]);
                if (a.res) v += rr[e];
                d[e] = v > 0.f ? d[e] : 0.f;
            }
        }
        if (a.dres) store8(a.dres, o, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = g[e] * istd[e] * (d[e] - db[e] * inv_count - xh[e] * dg[e] * inv_count);
        if (a.dx) store8(a.dx, o, r);
        if (dxs) x3s_store8(dxs, r, (size_t)a.M, a.C >> 4, m, tx);
    }
}

// pool_fwd_kernel with eight channels per thread, pooled map as fp32 (nullable) and / or slices
// codes / xmax (both or neither, nullable): the arg-max record for the streaming backward (dmc_bn_relu_pool_bwd_arg) --
// codes = the window position 0..8 of each channel's maximum under PyTorch's rule (row-major scan of the in-bounds
// positions, strictly greater wins), one byte per channel, the layout pool_bwd_kernel<2> reads; xmax = the RAW input at
// that position [N][PH][PW][C], from which the backward's first pass forms xhat and the ReLU derivative without
// touching the 4x larger input again.
__global__ __launch_bounds__(256) void pool_fwd_x3s_kernel(PoolArgs a, unsigned short* __restrict__ ys, unsigned* __restrict__ codes,
                                                           float* __restrict__ xmax) {
    const int c8 = a.C >> 3;
    const size_t total = (size_t)a.N * a.PH * a.PW * c8;
    for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
        const int tx = (int)(o & (size_t)(c8 - 1));
        size_t r = o / c8;
        const size_t mp = r;
        const int px = (int)(r % a.PW); r /= a.PW;
        const int py = (int)(r % a.PH);
        const int n = (int)(r / a.PH);
        float mean[8], istd[8], g[8], b[8], mx[8];
        load8(a.stats, tx, mean); load8(a.stats + a.C, tx, istd); load8(a.gamma, tx, g); load8(a.beta, tx, b);
        float xv[9][8];
#pragma unroll
        for (int k = 0; k < 9; ++k) {                          // clamped coordinates repeat a value of the same window
            int iy = 2 * py + k / 3 - 1, ix = 2 * px + k % 3 - 1;
            iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
            ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            load8(a.x, ((size_t)(n * a.H + iy) * a.W + ix) * c8 + tx, xv[k]);
        }
        if (codes) {
            float bx[8];
            unsigned ck[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; bx[e] = 0.f; ck[e] = 0; }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int iy = 2 * py + k / 3 - 1, ix = 2 * px + k % 3 - 1;
                const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = fmaxf(fmaf((xv[k][e] - mean[e]) * istd[e], g[e], b[e]), 0.f);
                    const bool take = in && v > mx[e];
                    mx[e] = take ? v : mx[e];
                    bx[e] = take ? xv[k][e] : bx[e];
                    ck[e] = take ? (unsigned)k : ck[e];
                }
            }
            reinterpret_cast<uint2*>(codes)[o] = make_uint2(ck[0] | (ck[1] << 8) | (ck[2] << 16) | (ck[3] << 24),
                                                            ck[4] | (ck[5] << 8) | (ck[6] << 16) | (ck[7] << 24));
            store8(xmax, o, bx);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[e] = 0.f;          // relu(.) >= 0 and the window centre is always inside
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], fmaxf(fmaf((xv[k][e] - mean[e]) * istd[e], g[e], b[e]), 0.f));
        }
        if (a.ypool) store8(a.ypool, o, mx);
        if (ys) x3s_store8(ys, mx, (size_t)a.N * a.PH * a.PW, a.C >> 4, mp, tx);
    }
}

bool shape_ok8(int M, int C) { return shape_ok(M, C) && C % 16 == 0 && C >= 16; }

}  // namespace

extern "C" {

int dmc_bn_act_supported(int M, int C) { return shape_ok(M, C) ? 1 : 0; }

size_t dmc_bn_act_stats_bytes(int C) { return (((size_t)2 * C * sizeof(float)) + 15) / 16 * 16; }
size_t dmc_bn_act_scratch_bytes(int C) { return (size_t)MAX_SPLIT * C * 2 * sizeof(double); }

int dmc_bn_act_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float* y, float* stats, void* scratch_,
                   unsigned char* relu_mask, int M, int C, int relu, int training, float eps,
                   float momentum, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !running_mean || !running_var || !y || !stats || (training && !scratch_))
        return fail(DMC_E_INVALID, "dmc_bn_act_fwd: null pointer");
    if (!shape_ok(M, C)) return fail(DMC_E_INVALID, "dmc_bn_act_fwd: unsupported shape M=%d C=%d", M, C);
    hipStream_t s = (hipStream_t)stream;
    BnArgs a = {x, residual, gamma, beta, stats, nullptr, y, nullptr, nullptr, M, C, relu, relu_mask};
    double* scratch = static_cast<double*>(scratch_);
    int rc;
    const int split = split_of(M, C);
    if (training) {
        bn_partial_kernel<0><<<split, 256, 0, s>>>(a, scratch);
        if ((rc = check_launch("bn_partial"))) return rc;
    }
    bn_stats_final_kernel<<<(C + 7) / 8, 256, 0, s>>>(scratch, stats, running_mean, running_var, C,
                                                       (long)M, training, eps, momentum, split);
    if ((rc = check_launch("bn_stats_final"))) return rc;
    bn_apply_fwd_kernel<<<stream_blocks((size_t)M * (C / 4)), 256, 0, s>>>(a);
    return check_launch("bn_apply_fwd");
}

int dmc_bn_act_bwd(const float* x, const float* residual, const float* gamma, const float* beta,
                   const float* stats, void* scratch_, const float* dy, float* dx, float* dresidual,
                   float* dgamma, float* dbeta, const unsigned char* relu_mask, int M, int C, int relu,
                   dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || !scratch_ || !dy || !dx || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_bn_act_bwd: null pointer");
    if (!shape_ok(M, C)) return fail(DMC_E_INVALID, "dmc_bn_act_bwd: unsupported shape M=%d C=%d", M, C);
    hipStream_t s = (hipStream_t)stream;
    BnArgs a = {x, residual, gamma, beta, stats, dy, nullptr, dx, dresidual, M, C, relu,
                const_cast<unsigned char*>(relu_mask)};
    double* scratch = static_cast<double*>(scratch_);
    int rc;
    const int split = split_of(M, C);
    bn_partial_kernel<1><<<split, 256, 0, s>>>(a, scratch);
    if ((rc = check_launch("bn_bwd_partial"))) return rc;
    bn_bwd_final_kernel<<<C, 256, 0, s>>>(scratch, dgamma, dbeta, C, split);
    if ((rc = check_launch("bn_bwd_final"))) return rc;
    bn_apply_bwd_kernel<<<stream_blocks((size_t)M * (C / 4)), 256, 0, s>>>(a, dgamma, dbeta, 1.f / (float)M);
    return check_launch("bn_apply_bwd");
}

int dmc_bn_relu_pool_supported(int N, int H, int W, int C) {
    return N > 0 && H > 0 && W > 0 && shape_ok(1, C) && C / 4 <= 64 ? 1 : 0;
}

int dmc_bn_relu_pool_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* y_pool, float* stats, void* scratch_, int N, int H,
                         int W, int C, int training, float eps, float momentum, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !running_mean || !running_var || !y_pool || !stats || (training && !scratch_))
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_fwd: null pointer");
    if (!dmc_bn_relu_pool_supported(N, H, W, C))
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_fwd: unsupported shape N=%d H=%d W=%d C=%d", N, H, W, C);
    hipStream_t s = (hipStream_t)stream;
    const int M = N * H * W;
    BnArgs a = {x, nullptr, gamma, beta, stats, nullptr, nullptr, nullptr, nullptr, M, C, 1, nullptr};
    double* scratch = static_cast<double*>(scratch_);
    int rc;
    const int split = split_of(M, C);
    if (training) {
        bn_partial_kernel<0><<<split, 256, 0, s>>>(a, scratch);
        if ((rc = check_launch("bn_partial"))) return rc;
    }
    bn_stats_final_kernel<<<(C + 7) / 8, 256, 0, s>>>(scratch, stats, running_mean, running_var, C,
                                                       (long)M, training, eps, momentum, split);
    if ((rc = check_launch("bn_stats_final"))) return rc;
    PoolArgs p = {x, gamma, beta, stats, nullptr, y_pool, nullptr, N, H, W, C, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    pool_fwd_kernel<<<stream_blocks((size_t)N * p.PH * p.PW * (C / 4)), 256, 0, s>>>(p);
    return check_launch("pool_fwd");
}

size_t dmc_bn_relu_pool_codes_bytes(int N, int H, int W, int C) {
    return (size_t)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (C / 4) * sizeof(unsigned);
}

int dmc_bn_relu_pool_bwd(const float* x, const float* gamma, const float* beta, const float* stats,
                         void* scratch_, const float* d_pool, float* dx, float* dgamma, float* dbeta,
                         void* codes, int N, int H, int W, int C, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || !scratch_ || !d_pool || !dx || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_bwd: null pointer");
    if (!dmc_bn_relu_pool_supported(N, H, W, C))
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_bwd: unsupported shape N=%d H=%d W=%d C=%d", N, H, W, C);
    hipStream_t s = (hipStream_t)stream;
    PoolArgs p = {x, gamma, beta, stats, d_pool, nullptr, dx, N, H, W, C, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    double* scratch = static_cast<double*>(scratch_);
    const long tiles = (long)N * ((p.PH + PT - 1) / PT) * ((p.PW + PT - 1) / PT);
    const int blocks = (int)(tiles < MAX_SPLIT ? tiles : MAX_SPLIT);
    int rc;
    pool_bwd_kernel<1><<<blocks, 256, 0, s>>>(p, scratch, nullptr, nullptr, 0.f, (unsigned*)codes);
    if ((rc = check_launch("pool_bwd_partial"))) return rc;
    bn_bwd_final_kernel<<<C, 256, 0, s>>>(scratch, dgamma, dbeta, C, blocks);
    if ((rc = check_launch("bn_bwd_final"))) return rc;
    pool_bwd_kernel<2><<<blocks * 4 > 8192 ? 8192 : blocks * 4, 256, 0, s>>>(p, nullptr, dgamma, dbeta,
                                                                             1.f / (float)((long)N * H * W), (unsigned*)codes);
    return check_launch("pool_bwd_apply");
}

// y = relu?((x - mean) * invstd * gamma + beta [+ residual]) with GIVEN statistics (stats = mean[C],
// invstd[C]): the apply pass alone, for callers whose producer already reduced the statistics
// (dmc_conv_nhwc_fwd + dmc_conv_nhwc_stats_final).  relu_mask as in dmc_bn_act_fwd (nullable).
int dmc_bn_apply_act_nhwc(const float* x, const float* residual, const float* gamma, const float* beta,
                          const float* stats, float* y, unsigned char* relu_mask, int M, int C, int relu,
                          dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || !y) return fail(DMC_E_INVALID, "dmc_bn_apply_act_nhwc: null pointer");
    if (!shape_ok(M, C)) return fail(DMC_E_INVALID, "dmc_bn_apply_act_nhwc: unsupported shape M=%d C=%d", M, C);
    BnArgs a = {x, residual, gamma, beta, stats, nullptr, y, nullptr, nullptr, M, C, relu, relu_mask};
    bn_apply_fwd_kernel<<<stream_blocks((size_t)M * (C / 4)), 256, 0, (hipStream_t)stream>>>(a);
    return check_launch("bn_apply_act_nhwc");
}

int dmc_bn_apply_nhwc(const float* x, const float* gamma, const float* beta, const float* stats, float* y,
                      int M, int C, dmc_stream_t stream) {
    return dmc_bn_apply_act_nhwc(x, nullptr, gamma, beta, stats, y, nullptr, M, C, 0, stream);
}

// Backward of  y = BN(z),  z = keep[n][c] * LeakyReLU_slope(pre)  down to d(pre): the BatchNorm
// backward (batch statistics; dgamma, dbeta as usual) followed by the mask and the LeakyReLU
// derivative (sign(z) == sign(pre) wherever keep != 0).  gamma == NULL: no BatchNorm (dz = dy).
// hw = pixels per frame (maps a row of the [M][C] matrix to its frame for `keep`).
int dmc_bn_bwd_act_nhwc(const float* z, const float* gamma, const float* beta, const float* stats, void* scratch_,
                        const float* dy, float* dpre, float* dgamma, float* dbeta, const float* keep, int hw,
                        float slope, int M, int C, dmc_stream_t stream) {
    if (!z || !dy || !dpre || hw <= 0) return fail(DMC_E_INVALID, "dmc_bn_bwd_act_nhwc: bad argument");
    if (!shape_ok(M, C)) return fail(DMC_E_INVALID, "dmc_bn_bwd_act_nhwc: unsupported shape M=%d C=%d", M, C);
    hipStream_t s = (hipStream_t)stream;
    if (!gamma) {
        act_mask_bwd_kernel<<<stream_blocks((size_t)M * (C / 4)), 256, 0, s>>>(z, keep, dy, dpre, M, C, hw, slope);
        return check_launch("act_mask_bwd");
    }
    if (!beta || !stats || !scratch_ || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_bn_bwd_act_nhwc: null BatchNorm pointer");
    BnArgs a = {z, nullptr, gamma, beta, stats, dy, nullptr, dpre, nullptr, M, C, 0, nullptr, keep, hw, slope};
    double* scratch = static_cast<double*>(scratch_);
    int rc;
    const int split = split_of(M, C);
    bn_partial_kernel<1><<<split, 256, 0, s>>>(a, scratch);
    if ((rc = check_launch("bn_bwd_partial"))) return rc;
    bn_bwd_final_kernel<<<C, 256, 0, s>>>(scratch, dgamma, dbeta, C, split);
    if ((rc = check_launch("bn_bwd_final"))) return rc;
    bn_apply_bwd_kernel<<<stream_blocks((size_t)M * (C / 4)), 256, 0, s>>>(a, dgamma, dbeta, 1.f / (float)M);
    return check_launch("bn_apply_bwd_act");
}

// out[c] = sum over the M rows of g[.][c] (the bias gradient of a convolution), deterministic.
int dmc_channel_sum_nhwc(const float* g, void* scratch_, float* out, int M, int C, dmc_stream_t stream) {
    if (!g || !scratch_ || !out) return fail(DMC_E_INVALID, "dmc_channel_sum_nhwc: null pointer");
    if (!shape_ok(M, C)) return fail(DMC_E_INVALID, "dmc_channel_sum_nhwc: unsupported shape M=%d C=%d", M, C);
    hipStream_t s = (hipStream_t)stream;
    BnArgs a = {g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, C, 0, nullptr};
    const int split = split_of(M, C);
    bn_partial_kernel<0><<<split, 256, 0, s>>>(a, static_cast<double*>(scratch_));
    int rc = check_launch("channel_sum_partial");
    if (rc) return rc;
    channel_sum_final_kernel<<<(C + 7) / 8, 256, 0, s>>>(static_cast<const double*>(scratch_), out, C, split);
    return check_launch("channel_sum_final");
}

/* dmc_bn_apply_act_nhwc with slice output: y (fp32, nullable) and / or ys (bf16x3 slice tensor of the same values,
 * nullable; dmc_x3s_slices_bytes(M, C)); C % 16 == 0. */
int dmc_bn_apply_act_x3s(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                         float* y, void* ys, unsigned char* relu_mask, int M, int C, int relu, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || (!y && !ys)) return fail(DMC_E_INVALID, "dmc_bn_apply_act_x3s: null pointer");
    if (!shape_ok8(M, C)) return fail(DMC_E_INVALID, "dmc_bn_apply_act_x3s: unsupported shape M=%d C=%d", M, C);
    BnArgs a = {x, residual, gamma, beta, stats, nullptr, y, nullptr, nullptr, M, C, relu, relu_mask};
    bn_apply_fwd_x3s_kernel<<<stream_blocks((size_t)M * (C / 8)), 256, 0, (hipStream_t)stream>>>(a, static_cast<unsigned short*>(ys), 0, 0);
    return check_launch("bn_apply_act_x3s");
}

/* dmc_bn_apply_act_x3s whose slice output is SPACE-TO-DEPTH over the [N][H][W] pixel grid (H, W even): the layout the
 * stride-2 block pair reads (dmc_x3q_*, include/dmcnet_hip.h); y (fp32, nullable) keeps the ordinary NHWC layout. */
int dmc_bn_apply_act_x3q(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                         float* y, void* yq, unsigned char* relu_mask, int N, int H, int W, int C, int relu, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || !yq) return fail(DMC_E_INVALID, "dmc_bn_apply_act_x3q: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (long)N * H * W >= (1L << 31)) return fail(DMC_E_INVALID, "dmc_bn_apply_act_x3q: bad geometry");
    const int M = N * H * W;
    if (!shape_ok8(M, C)) return fail(DMC_E_INVALID, "dmc_bn_apply_act_x3q: unsupported shape M=%d C=%d", M, C);
    BnArgs a = {x, residual, gamma, beta, stats, nullptr, y, nullptr, nullptr, M, C, relu, relu_mask};
    bn_apply_fwd_x3s_kernel<<<stream_blocks((size_t)M * (C / 8)), 256, 0, (hipStream_t)stream>>>(a, static_cast<unsigned short*>(yq), H, W);
    return check_launch("bn_apply_act_x3q");
}

/* dmc_bn_act_bwd with the BatchNorm input gradient as fp32 (dx, nullable) and / or as a slice tensor (dxs, nullable). */
int dmc_bn_act_bwd_x3s(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                       void* scratch_, const float* dy, float* dx, void* dxs, float* dresidual, float* dgamma, float* dbeta,
                       const unsigned char* relu_mask, int M, int C, int relu, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || !scratch_ || !dy || (!dx && !dxs) || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_bn_act_bwd_x3s: null pointer");
    if (!shape_ok8(M, C)) return fail(DMC_E_INVALID, "dmc_bn_act_bwd_x3s: unsupported shape M=%d C=%d", M, C);
    hipStream_t s = (hipStream_t)stream;
    BnArgs a = {x, residual, gamma, beta, stats, dy, nullptr, dx, dresidual, M, C, relu, const_cast<unsigned char*>(relu_mask)};
    double* scratch = static_cast<double*>(scratch_);
    int rc;
    const int split = split_of(M, C);
    bn_partial_kernel<1><<<split, 256, 0, s>>>(a, scratch);
    if ((rc = check_launch("bn_bwd_partial"))) return rc;
    bn_bwd_final_kernel<<<C, 256, 0, s>>>(scratch, dgamma, dbeta, C, split);
    if ((rc = check_launch("bn_bwd_final"))) return rc;
    bn_apply_bwd_x3s_kernel<<<stream_blocks((size_t)M * (C / 8)), 256, 0, s>>>(a, dgamma, dbeta, 1.f / (float)M, static_cast<unsigned short*>(dxs));
    return check_launch("bn_apply_bwd_x3s");
}

/* dmc_bn_relu_pool_fwd with the pooled map as fp32 (y_pool, nullable) and / or as a slice tensor (ys, nullable). */
int dmc_bn_relu_pool_fwd_arg(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float* y_pool, void* ys, void* codes, float* xmax, float* stats, void* scratch_, int stat_split,
                             int N, int H, int W, int C, int training, float eps, float momentum, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !running_mean || !running_var || (!y_pool && !ys) || !stats || (training && !scratch_))
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_fwd_x3s: null pointer");
    if (stat_split < 0 || stat_split > MAX_SPLIT) return fail(DMC_E_INVALID, "dmc_bn_relu_pool_fwd_arg: bad stat_split %d", stat_split);
    if ((codes == nullptr) != (xmax == nullptr)) return fail(DMC_E_INVALID, "dmc_bn_relu_pool_fwd_arg: codes and xmax go together");
    if (!dmc_bn_relu_pool_supported(N, H, W, C) || C % 16 != 0)
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_fwd_x3s: unsupported shape N=%d H=%d W=%d C=%d", N, H, W, C);
    hipStream_t s = (hipStream_t)stream;
    const int M = N * H * W;
    BnArgs a = {x, nullptr, gamma, beta, stats, nullptr, nullptr, nullptr, nullptr, M, C, 1, nullptr};
    double* scratch = static_cast<double*>(scratch_);
    int rc;
    // stat_split > 0: the producer of x (dmc_stem_fwd_x3_stats) left that many partial sums per channel in scratch
    const int split = stat_split > 0 ? stat_split : split_of(M, C);
    if (training && stat_split == 0) {
        bn_partial_kernel<0><<<split, 256, 0, s>>>(a, scratch);
        if ((rc = check_launch("bn_partial"))) return rc;
    }
    bn_stats_final_kernel<<<(C + 7) / 8, 256, 0, s>>>(scratch, stats, running_mean, running_var, C, (long)M, training, eps, momentum, split);
    if ((rc = check_launch("bn_stats_final"))) return rc;
    PoolArgs p = {x, gamma, beta, stats, nullptr, y_pool, nullptr, N, H, W, C, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    pool_fwd_x3s_kernel<<<stream_blocks((size_t)N * p.PH * p.PW * (C / 8)), 256, 0, s>>>(p, static_cast<unsigned short*>(ys),
                                                                                         static_cast<unsigned*>(codes), xmax);
    return check_launch("pool_fwd_x3s");
}

int dmc_bn_relu_pool_fwd_x3s(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float* y_pool, void* ys, float* stats, void* scratch_, int N, int H, int W, int C, int training,
                             float eps, float momentum, dmc_stream_t stream) {
    return dmc_bn_relu_pool_fwd_arg(x, gamma, beta, running_mean, running_var, y_pool, ys, nullptr, nullptr, stats, scratch_, 0, N, H, W, C,
                                    training, eps, momentum, stream);
}

/* Backward of dmc_bn_relu_pool_fwd_arg from its arg-max record: the BatchNorm-backward sums come from ONE streaming pass
 * over the pooled-size tensors (d_pool, xmax) -- sum(dz) and sum(dz * xhat) over input pixels = the same sums over the
 * windows, each window's gradient counted at its arg-max -- instead of a pass over the 4x larger input with window
 * gathers (0.225 -> 0.05 ms at 120 x 112 x 112 x 64); the second pass routes d_pool through `codes` and writes dx. */
int dmc_bn_relu_pool_bwd_arg(const float* x, const float* gamma, const float* beta, const float* stats, void* scratch_,
                             const float* d_pool, const void* codes, const float* xmax, float* dx, float* dgamma, float* dbeta,
                             int N, int H, int W, int C, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || !scratch_ || !d_pool || !codes || !xmax || !dx || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_bwd_arg: null pointer");
    if (!dmc_bn_relu_pool_supported(N, H, W, C) || C % 16 != 0)
        return fail(DMC_E_INVALID, "dmc_bn_relu_pool_bwd_arg: unsupported shape N=%d H=%d W=%d C=%d", N, H, W, C);
    hipStream_t s = (hipStream_t)stream;
    PoolArgs p = {x, gamma, beta, stats, d_pool, nullptr, dx, N, H, W, C, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    double* scratch = static_cast<double*>(scratch_);
    const int MP = N * p.PH * p.PW;
    BnArgs a = {xmax, nullptr, gamma, beta, stats, d_pool, nullptr, nullptr, nullptr, MP, C, 1, nullptr};
    const int split = split_of(MP, C);
    int rc;
    bn_partial_kernel<1><<<split, 256, 0, s>>>(a, scratch);
    if ((rc = check_launch("pool_bwd_sums"))) return rc;
    bn_bwd_final_kernel<<<C, 256, 0, s>>>(scratch, dgamma, dbeta, C, split);
    if ((rc = check_launch("bn_bwd_final"))) return rc;
    const long tiles = (long)N * ((p.PH + PT - 1) / PT) * ((p.PW + PT - 1) / PT);
    const long blocks = tiles * 4 > 8192 ? 8192 : tiles * 4;
    pool_bwd_kernel<2><<<(int)blocks, 256, 0, s>>>(p, nullptr, dgamma, dbeta, 1.f / (float)((long)N * H * W),
                                                   const_cast<unsigned*>(static_cast<const unsigned*>(codes)));
    return check_launch("pool_bwd_apply");
}

/* Second half of dmc_bn_act_bwd_x3s alone: dgamma / dbeta are INPUTS (reduced by the launch that produced dy,
 * dmc_x3s_conv_dgrad_bnb); one streaming pass writes dx (fp32, nullable) / dxs (slices, nullable) / dresidual. */
int dmc_bn_act_bwd_x3s_apply(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                             const float* dy, float* dx, void* dxs, float* dresidual, const float* dgamma, const float* dbeta,
                             const unsigned char* relu_mask, int M, int C, int relu, dmc_stream_t stream) {
    if (!x || !gamma || !beta || !stats || !dy || (!dx && !dxs) || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_bn_act_bwd_x3s_apply: null pointer");
    if (!shape_ok8(M, C)) return fail(DMC_E_INVALID, "dmc_bn_act_bwd_x3s_apply: unsupported shape M=%d C=%d", M, C);
    BnArgs a = {x, residual, gamma, beta, stats, dy, nullptr, dx, dresidual, M, C, relu, const_cast<unsigned char*>(relu_mask)};
    bn_apply_bwd_x3s_kernel<<<stream_blocks((size_t)M * (C / 8)), 256, 0, (hipStream_t)stream>>>(a, dgamma, dbeta, 1.f / (float)M, static_cast<unsigned short*>(dxs));
    return check_launch("bn_apply_bwd_x3s");
}

}  // extern "C"
