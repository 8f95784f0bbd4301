// NHWC fp32 convolutions on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
//
// One implicit-GEMM family serves the PatchGAN discriminator blocks
// (code/dmcnet_GAN/model.py:254-279,332-366: Conv2d 3x3, stride 2 or 1, padding 1, bias) and the
// ResNet-18 classifier's convolutions (torchvision BasicBlock: 3x3 stride 1/2 and the 1x1 stride-2
// shortcuts, built at code/dmcnet/model.py:305, run at :352):
//
//   forward        y[p][co]  = sum_{tap,ci} x[p*s + tap - pad][ci] * w[co][tap][ci]
//   data gradient  the same kernel on dy with the weights re-packed as wt[ci][8 - tap][co]
//                  (stride 2: one launch per output-parity class, each with its own tap subset, so no
//                  multiply is wasted on structural zeros)
//   weight grad.   dw[co][tap][ci] = sum_p dy[p][co] * x[p*s + tap - pad][ci]      (GEMM over pixels,
//                  split over workgroups, fixed-order two-stage reduction: deterministic, no atomics,
//                  no zero-fill)
//
// Activations NHWC ([pixel][channel], the memory of a channels_last tensor), weights OHWI
// ([Cout][KH][KW][Cin], the memory of a channels_last weight).  GEMM view of the forward:
// rows = output channels (A operand = weights), columns = output pixels (B operand = the input at the
// tap-shifted pixel), K = (tap, ci).  A 256-thread workgroup owns a BM-pixel x BN-channel tile; per
// K-step (one tap, BK input channels) every thread moves its share of both operand tiles
// global -> registers -> LDS (16-byte loads along ci, register double buffer: the loads of step t+1
// are in flight during the MFMAs of step t; one barrier per step), the waves read 16-byte fragments
// along k and issue 16x16x4 MFMAs.  The f32 MFMA is 16x slower than the bf16 one, so operand
// bandwidth is never the limit here; the design goal is simply an MFMA stream without gaps.
// C/D layout: a lane holds 4 consecutive output channels of one pixel -> one 16-byte store per tile.
//
// Fused epilogues (forward): + bias, LeakyReLU(0.2), Dropout2d keep-mask, and per-channel
// (sum, sum of squares) partials of the result for the BatchNorm that follows (fp64 across lanes,
// waves and workgroups, fixed order).
#include "dmc_common.h"

using namespace dmc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x;        // [N][H][W][Cin]
    const float* w;        // [Cout][KH*KW][Cin]
    float* y;              // [N][OH][OW][Cout]
    const float* bias;     // [Cout] or null
    const float* keep;     // [N][Cout] or null   (Dropout2d keep mask, already divided by 1 - p)
    double* stat_part;     // [gridDim.x][Cout][2] or null
    int N, H, W, Cin, OH, OW, Cout;
    int KK;                // taps in the weight array (KH * KW): weight rows are [co][KK][Cin]
    int ntaps;             // taps this launch enumerates
    int stride;            // input step per sub-grid pixel
    int oy0, ox0, ostep;   // written pixel = (oy0 + yy * ostep, ox0 + xx * ostep); forward: 0, 0, 1
    int OHs, OWs;          // extent of the (yy, xx) sub-grid
    int act;               // 0: none, 1: LeakyReLU(0.2)
    // per tap: input offset (dy, dx) and index in the weight array, 4 bits each (offsets biased by 8),
    // packed in 64-bit words so that the (wave-uniform) lookup is two scalar shifts -- a byte table in
    // the kernel argument is read with vector loads, and waiting for one drains every prefetch in flight
    unsigned long long tap_dy, tap_dx, tap_w;
    int M;                 // N * OHs * OWs

    __host__ void set_tap(int t, int dy, int dx, int wi) {
        const unsigned long long m = ~(15ull << (4 * t));
        tap_dy = (tap_dy & m) | ((unsigned long long)(dy + 8) << (4 * t));
        tap_dx = (tap_dx & m) | ((unsigned long long)(dx + 8) << (4 * t));
        tap_w = (tap_w & m) | ((unsigned long long)wi << (4 * t));
    }
};

// One index map serves every use.  The launch enumerates a sub-grid (yy, xx) of output pixels and a
// list of taps; tap t reads input pixel (yy * stride + tap_dy[t], xx * stride + tap_dx[t]) (zero when
// outside the H x W image) against weight slice tap_w[t]:
//   forward (stride s, padding p):      tap_dy = ky - p, tap_w = ky * KW + kx, the whole output grid;
//   data gradient, stride 1:            tap_dy = p - ky (the same weight slice), on dy with packed weights;
//   data gradient, stride 2, class (py, px) of dx pixels (iy = 2 yy + py): only the taps with
//       (py + p - ky) even contribute, tap_dy = (py + p - ky) / 2, stride 1, written pixel = class grid.
template <int BM, int BN, int BK>
struct Tile {
    static constexpr int QK = BK / 4;                     // 16-byte quads along k per row
    static constexpr int PITCH = BK + 4;                  // floats per LDS row (16-byte aligned, spreads banks)
    static constexpr int NA = BM * QK / 256;              // quads of the pixel tile per thread
    static constexpr int NB = (BN * QK + 255) / 256;      // quads of the weight tile per thread
    static constexpr int LDS_FLOATS = 2 * (BM + BN) * PITCH;
};

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(256) void conv_nhwc_kernel(ConvArgs a) {
    using T = Tile<BM, BN, BK>;
    constexpr int QK = T::QK, PITCH = T::PITCH, NA = T::NA, NB = T::NB;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;   // 16x16 sub-tiles per wave: pixels x channels
    static_assert(WM * WN == 4 && BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "wave layout");
    static_assert(BM * QK % 256 == 0, "pixel tile / thread mapping");
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    constexpr int BUF = (BM + BN) * PITCH;                 // one buffer: pixel tile, then weight tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = blockIdx.x * BM, co0 = blockIdx.y * BN;
    const int cinq = a.Cin;                                // floats per input pixel

    // ---- per-thread share of the operand tiles ----
    // pixel tile: row p = tid / QK + r * (256 / QK), quad q = tid % QK
    const int q = tid % QK;
    long a_base[NA];           // element offset of (n, iy0, ix0, 0), iy0/ix0 = input coords of tap offset (0, 0)
    int a_iy[NA], a_ix[NA];    // those coords (validity tests); a_iy = INT_MIN/2 for rows beyond M
#pragma unroll
    for (int r = 0; r < NA; ++r) {
        const int p = tid / QK + r * (256 / QK);
        const int m = m0 + p;
        if (m < a.M) {
            const int n = m / (a.OHs * a.OWs), rem = m - n * (a.OHs * a.OWs);
            const int yy = rem / a.OWs, xx = rem - yy * a.OWs;
            a_iy[r] = yy * a.stride;
            a_ix[r] = xx * a.stride;
            a_base[r] = (((long)n * a.H + a_iy[r]) * a.W + a_ix[r]) * cinq + 4 * q;
        } else {
            a_iy[r] = -(1 << 28); a_ix[r] = 0; a_base[r] = 0;
        }
    }
    // weight tile: row c = tid / QK + r * (256 / QK) (< BN), quad q
    const int nci = a.Cin / BK;                            // K-steps per tap
    const int T_steps = a.ntaps * nci;

    float4 ra[NA], rb[NB];
    auto load_step = [&](int t) {
        const int tap = t / nci, ci0 = (t - tap * nci) * BK;
        const int dy = (int)((a.tap_dy >> (4 * tap)) & 15) - 8, dx = (int)((a.tap_dx >> (4 * tap)) & 15) - 8;
        const long toff = ((long)dy * a.W + dx) * cinq + ci0;
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            const int iy = a_iy[r] + dy, ix = a_ix[r] + dx;
            const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            ra[r] = ok ? *reinterpret_cast<const float4*>(a.x + a_base[r] + toff) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int tw = (int)((a.tap_w >> (4 * tap)) & 15);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int c = tid / QK + r * (256 / QK);
            const bool ok = c < BN && co0 + c < a.Cout;
            rb[r] = ok ? *reinterpret_cast<const float4*>(a.w + ((long)(co0 + c) * a.KK + tw) * cinq + ci0 + 4 * q)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int r = 0; r < NA; ++r)
            *reinterpret_cast<float4*>(lds + buf * BUF + (tid / QK + r * (256 / QK)) * PITCH + 4 * q) = ra[r];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int c = tid / QK + r * (256 / QK);
            if (c < BN) *reinterpret_cast<float4*>(lds + buf * BUF + (BM + c) * PITCH + 4 * q) = rb[r];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int i16 = lane & 15, kq = lane >> 4;
    const int prow0 = wm * (BM / WM) + i16, crow0 = wn * (BN / WN) + i16;

    if (T_steps > 0) {
        load_step(0);
        store_step(0);
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < T_steps; ++t) {
        const int buf = t & 1;
        if (t + 1 < T_steps) load_step(t + 1);              // in flight during the MFMAs below
        const float* A = lds + buf * BUF;
        const float* B = A + BM * PITCH;
#pragma unroll
        for (int h = 0; h < BK / 16; ++h) {
            float4 xa[TM], wb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                xa[i] = *reinterpret_cast<const float4*>(A + (prow0 + 16 * i) * PITCH + 16 * h + 4 * kq);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wb[j] = *reinterpret_cast<const float4*>(B + (crow0 + 16 * j) * PITCH + 16 * h + 4 * kq);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].x, xa[i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].y, xa[i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].z, xa[i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j].w, xa[i].w, acc[i][j], 0, 0, 0);
                }
        }
        if (t + 1 < T_steps) {
            store_step(buf ^ 1);                             // nobody reads that buffer during this step
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds pixel column i16 of tile i, channels 4*kq .. 4*kq+3 of tile j ----
    float s1[TN][4], s2[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[j][e] = 0.f; s2[j][e] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + 16 * i + i16;
        const bool mok = m < a.M;
        int n = 0;
        long opix = 0;
        if (mok) {
            n = m / (a.OHs * a.OWs);
            const int rem = m - n * (a.OHs * a.OWs);
            const int yy = rem / a.OWs, xx = rem - yy * a.OWs;
            opix = ((long)n * a.OH + (a.oy0 + yy * a.ostep)) * a.OW + (a.ox0 + xx * a.ostep);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = co0 + wn * (BN / WN) + 16 * j + 4 * kq;
            if (co >= a.Cout) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (a.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + co);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (a.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
            }
            if (a.keep && mok) {
                const float4 kv = *reinterpret_cast<const float4*>(a.keep + (long)n * a.Cout + co);
                v[0] *= kv.x; v[1] *= kv.y; v[2] *= kv.z; v[3] *= kv.w;
            }
            if (mok) {
                *reinterpret_cast<float4*>(a.y + opix * a.Cout + co) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[j][e] += v[e]; s2[j][e] += v[e] * v[e]; }
            }
        }
    }
    if (a.stat_part) {
        // per-channel sums of this workgroup's tile: lanes (16 pixel columns) -> waves (WM) -> one store
        __syncthreads();                                      // the operand tiles are dead: reuse the LDS
        double* red = reinterpret_cast<double*>(lds);         // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double d1 = (double)s1[j][e], d2 = (double)s2[j][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    d1 += __shfl_xor(d1, o, 64);
                    d2 += __shfl_xor(d2, o, 64);
                }
                if (i16 == 0) {
                    const int c = wn * (BN / WN) + 16 * j + 4 * kq + e;
                    red[(wm * BN + c) * 2 + 0] = d1;
                    red[(wm * BN + c) * 2 + 1] = d2;
                }
            }
        __syncthreads();
        if (tid < BN && co0 + tid < a.Cout) {
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { d1 += red[(w * BN + tid) * 2 + 0]; d2 += red[(w * BN + tid) * 2 + 1]; }
            double* dst = a.stat_part + ((size_t)blockIdx.x * a.Cout + co0 + tid) * 2;
            dst[0] = d1; dst[1] = d2;
        }
    }
}

// ---- data-gradient weights: wt[ci][tap][co] = w[co][tap][ci] ------------------------------------
__global__ __launch_bounds__(256) void conv_pack_wt_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                           int Cout, int KK, int Cin) {
    const long total = (long)Cout * KK * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int co = (int)(i % Cout), tap = (int)((i / Cout) % KK), ci = (int)(i / ((long)Cout * KK));
        wt[i] = w[((long)co * KK + tap) * Cin + ci];
    }
}

// ---- BatchNorm statistics from the forward epilogue's partials -----------------------------------
// partials [nblk][C][2] (sum, sum of squares over each workgroup's pixels) -> stats (mean, invstd) and
// the running-statistics update of nn.BatchNorm2d (biased variance normalises, unbiased one is tracked).
__global__ __launch_bounds__(256) void conv_stats_final_kernel(const double* __restrict__ part, int nblk, int C,
                                                               long count, float* __restrict__ stats,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float eps,
                                                               float momentum) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s += part[((size_t)b * C + c) * 2 + 0];
        ss += part[((size_t)b * C + c) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); ss += __shfl_down(ss, o, 64); }
    if (lane != 0) return;
    const double mean = s / (double)count;
    double var = ss / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}


// ------------------------------------------------------------------------------------------
// Weight gradient: dw[co][tap][ci] = sum_p dy[p][co] * x[p*s + tap - pad][ci].
//
// GEMM with rows = co (A operand = dy, read "transposed": lane co, k = pixel), columns = ci
// (B operand = the tap-shifted input), K = output pixels.  A workgroup owns a 64 x 64 tile of ONE
// tap's [Cout][Cin] matrix and one slice of the pixel range (split-K: grid.y); it stages 32
// pixels per step -- dy rows [32][64 co] and x rows [32][64 ci], both contiguous 256-byte rows --
// and each wave accumulates a 32 x 32 block (2 x 2 MFMA tiles, K = 4 pixels per instruction).
// Partials [slice][Cout][KK][Cin] are summed in slice order by wgrad_reduce_kernel:
// deterministic, no atomics, nothing to zero first.
// ------------------------------------------------------------------------------------------
struct WgradConvArgs {
    const float* x;        // [N][H][W][Cin]
    const float* dy;       // [N][OH][OW][Cout]
    float* part;           // [nslice][Cout][KK][Cin]  (nslice == 1: dw itself)
    int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
    int M;                 // N * OH * OW
    int per_slice;         // pixels per slice (multiple of 32)
    int tiles_ci, tiles_co;
};

constexpr int WG_KP = 32;                       // pixels per step
constexpr int WG_T = 64;                        // tile edge (co and ci)
constexpr int WG_PITCH = WG_T + 16;             // 80 floats: the 4 pixel rows of a k-group hit distinct bank quarters

__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradConvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * WG_KP * WG_PITCH];     // [buf][dy | x][32][80]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KK = a.KH * a.KW;
    int tile = blockIdx.x;
    const int tci = tile % a.tiles_ci; tile /= a.tiles_ci;
    const int tap = tile % KK; tile /= KK;
    const int tco = tile;
    const int co0 = tco * WG_T, ci0 = tci * WG_T;
    const int ky = tap / a.KW - a.pad, kx = tap % a.KW - a.pad;
    const int p_begin = blockIdx.y * a.per_slice;
    const int p_end = p_begin + a.per_slice < a.M ? p_begin + a.per_slice : a.M;

    // staging: thread -> (pixel row tid / 16 + 16 r, quad tid % 16), r = 0, 1, for both tiles
    const int q = tid & 15, prow = tid >> 4;
    const bool co_ok = co0 + 4 * q < a.Cout, ci_ok = ci0 + 4 * q < a.Cin;
    float4 rd[2], rx[2];
    auto load_step = [&](int p0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = p0 + prow + 16 * r;
            rd[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            rx[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < p_end) {
                if (co_ok) rd[r] = *reinterpret_cast<const float4*>(a.dy + (long)m * a.Cout + co0 + 4 * q);
                const int n = m / (a.OH * a.OW), rem = m - n * (a.OH * a.OW);
                const int oy = rem / a.OW, ox = rem - oy * a.OW;
                const int iy = oy * a.stride + ky, ix = ox * a.stride + kx;
                if (ci_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                    rx[r] = *reinterpret_cast<const float4*>(a.x + (((long)n * a.H + iy) * a.W + ix) * a.Cin + ci0 + 4 * q);
            }
        }
    };
    auto store_step = [&](int buf) {
        float* d = lds + buf * (2 * WG_KP * WG_PITCH);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            *reinterpret_cast<float4*>(d + (prow + 16 * r) * WG_PITCH + 4 * q) = rd[r];
            *reinterpret_cast<float4*>(d + WG_KP * WG_PITCH + (prow + 16 * r) * WG_PITCH + 4 * q) = rx[r];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, kq = lane >> 4;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;      // this wave's 32 x 32 block: rows co, columns ci

    int it = 0;
    if (p_begin < p_end) { load_step(p_begin); store_step(0); }
    __syncthreads();
#pragma unroll 1
    for (int p0 = p_begin; p0 < p_end; p0 += WG_KP, ++it) {
        const int buf = it & 1;
        const bool more = p0 + WG_KP < p_end;
        if (more) load_step(p0 + WG_KP);
        const float* D = lds + buf * (2 * WG_KP * WG_PITCH);
        const float* X = D + WG_KP * WG_PITCH;
#pragma unroll
        for (int g = 0; g < WG_KP / 4; ++g) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = D[(4 * g + kq) * WG_PITCH + wr + 16 * i + i16];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = X[(4 * g + kq) * WG_PITCH + wc + 16 * j + i16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            store_step(buf ^ 1);
            __syncthreads();
        }
    }
    // D layout: lane holds column ci = i16 of tile j, rows co = 4 kq + e of tile i
    float* out = a.part + (size_t)blockIdx.y * a.Cout * KK * a.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wc + 16 * j + i16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + wr + 16 * i + 4 * kq + e;
                if (co < a.Cout && ci < a.Cin) out[((size_t)co * KK + tap) * a.Cin + ci] = acc[i][j][e];
            }
        }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                int nslice, long numel) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < numel; i += (long)gridDim.x * 1024) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nslice; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * numel + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(dw + i) = s;
    }
}


// ------------------------------------------------------------------------------------------
// Weight gradient for few channels (Cin <= 32): all nine taps in one pass.
//
// With 16 or 32 input channels a [Cout x Cin] tile per tap leaves the 64 x 64 kernel above 75-94 %
// idle and re-reads the activations nine times.  Here a workgroup walks 8 x 8 blocks of output
// pixels; per block it stages the dy tile [64 px][CO_T] and the input patch ((8 s + 2)^2 pixels, all
// Cin channels) once, and each of its 4 waves takes 4 of the 16 pixel groups (K = 4 pixels per MFMA)
// against ALL columns (tap, ci): CO_T/16 x 9 Cin/16 accumulator tiles per wave, kept in registers over
// the workgroup's whole run of blocks, then summed across waves (LDS, fixed order) and written as one
// partial [Cout][9][Cin] per workgroup for conv_wgrad_reduce_kernel.
// ------------------------------------------------------------------------------------------
template <int CIN, int CO_T, int STRIDE>
struct SmallWg {
    static constexpr int PW = 8 * STRIDE + 2;              // patch edge (pixels)
    // pitches (floats): consecutive k-lanes (pixels) of a ds_read_b32 must land 16 banks apart
    static constexpr int XP = STRIDE == 1 ? 48 : 40;       // patch pixel pitch (>= CIN; pixel step = STRIDE * XP)
    static constexpr int DP = 48;                          // dy row pitch (>= CO_T)
    static constexpr int X_FLOATS = PW * PW * XP;
    static constexpr int D_FLOATS = 64 * DP;
    static constexpr int NT = 9 * CIN / 16, MT = CO_T / 16;
    static constexpr int OUT = CO_T * 9 * CIN;             // floats of one partial tile
};

template <int CIN, int CO_T, int STRIDE>
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(WgradConvArgs a, int blocks_y, int blocks_x,
                                                               int nblocks) {
    using G = SmallWg<CIN, CO_T, STRIDE>;
    constexpr int PW = G::PW, XP = G::XP, DP = G::DP, NT = G::NT, MT = G::MT;
    __shared__ __attribute__((aligned(16))) float lds[G::X_FLOATS + G::D_FLOATS > G::OUT ? G::X_FLOATS + G::D_FLOATS : G::OUT];
    float* xs = lds;
    float* ds = lds + G::X_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int co0 = blockIdx.y * CO_T;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this wave's pixel groups g = 4 wave .. 4 wave + 3: group g covers row g / 2, columns 4 (g % 2) .. +3
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int n = b / (blocks_y * blocks_x), r = b - n * (blocks_y * blocks_x);
        const int oy0 = (r / blocks_x) * 8, ox0 = (r % blocks_x) * 8;
        const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
        __syncthreads();                                    // previous block's tiles are no longer read
        // ---- stage the input patch: PW*PW pixels x CIN/4 quads ----
        constexpr int XQ = CIN / 4;
        for (int e = tid; e < PW * PW * XQ; e += 256) {
            const int q = e % XQ, pp = e / XQ, py = pp / PW, px = pp - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v = *reinterpret_cast<const float4*>(a.x + (((long)n * a.H + iy) * a.W + ix) * CIN + 4 * q);
            *reinterpret_cast<float4*>(xs + pp * XP + 4 * q) = v;
        }
        // ---- stage dy: 64 pixels x CO_T/4 quads (zero outside the image) ----
        constexpr int DQ = CO_T / 4;
        for (int e = tid; e < 64 * DQ; e += 256) {
            const int q = e % DQ, p = e / DQ, oy = oy0 + (p >> 3), ox = ox0 + (p & 7);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy < a.OH && ox < a.OW && co0 + 4 * q < a.Cout)
                v = *reinterpret_cast<const float4*>(a.dy + (((long)n * a.OH + oy) * a.OW + ox) * a.Cout + co0 + 4 * q);
            *reinterpret_cast<float4*>(ds + p * DP + 4 * q) = v;
        }
        __syncthreads();
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int g = wave * 4 + gg;
            const int prow = g >> 1, pcol = (g & 1) * 4 + kq;                 // this lane's pixel of the group
            float av[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = ds[(prow * 8 + pcol) * DP + 16 * i + i16];
            const float* xrow = xs + ((prow * STRIDE) * PW + pcol * STRIDE) * XP + i16;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float* xt = xrow + ((t / 3) * PW + (t % 3)) * XP;
#pragma unroll
                for (int c = 0; c < CIN / 16; ++c) {
                    const float bv = xt[16 * c];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[i][t * (CIN / 16) + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[i][t * (CIN / 16) + c], 0, 0, 0);
                }
            }
        }
    }
    // ---- cross-wave sum (fixed order) and store: lane holds column ci = i16 of tile (t, c), rows co = 4 kq + e ----
    __syncthreads();
    float* out = a.part + ((size_t)blockIdx.x * a.Cout + co0) * 9 * CIN;     // partial [wg][Cout][9][CIN]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int t = j / (CIN / 16), c = j % (CIN / 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int idx = ((16 * i + 4 * kq + e) * 9 + t) * CIN + 16 * c + i16;
                        lds[idx] = (w == 0 ? 0.f : lds[idx]) + acc[i][j][e];
                    }
                }
        }
        __syncthreads();
    }
    for (int e = tid; e < G::OUT; e += 256)
        if (co0 + e / (9 * CIN) < a.Cout) out[e] = lds[e];
}

template <int CIN, int CO_T, int STRIDE>
int launch_wgrad_small(WgradConvArgs a, float* dw, float* workspace, hipStream_t s) {
    const int by = (a.OH + 7) / 8, bx = (a.OW + 7) / 8, nblocks = a.N * by * bx;
    const int co_tiles = (a.Cout + CO_T - 1) / CO_T;
    int wgs = 512 / co_tiles;                                   // ~2 workgroups per CU in total
    if (wgs > nblocks) wgs = nblocks;
    a.part = wgs > 1 ? workspace : dw;
    conv_wgrad_small_kernel<CIN, CO_T, STRIDE><<<dim3(wgs, co_tiles), 256, 0, s>>>(a, by, bx, nblocks);
    int rc = check_launch("conv_wgrad_small");
    if (rc || wgs == 1) return rc;
    const long numel = (long)a.Cout * 9 * CIN;
    const long blocks = (numel / 4 + 255) / 256;
    conv_wgrad_reduce_kernel<<<(int)(blocks > 1024 ? 1024 : blocks), 256, 0, s>>>(workspace, dw, wgs, numel);
    return check_launch("conv_wgrad_reduce");
}

bool wgrad_small_ok(int Cin, int KH, int KW, int pad) { return (Cin == 16 || Cin == 32) && KH == 3 && KW == 3 && pad == 1; }

int wgrad_slices(long M, int tiles) {
    // enough workgroups to fill 256 CUs about three times over, slices of at least 1024 pixels
    long want = (768 + tiles - 1) / tiles;
    const long max_by_len = (M + 1023) / 1024;
    if (want > max_by_len) want = max_by_len;
    return (int)(want < 1 ? 1 : want);
}

struct ConvShape {
    int N, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW;
};

bool shape_supported(const ConvShape& s) {
    if (s.N <= 0 || s.H <= 0 || s.W <= 0) return false;
    if (s.Cin % 16 != 0 || s.Cout % 16 != 0) return false;
    if (!((s.KH == 3 && s.KW == 3 && s.pad == 1) || (s.KH == 1 && s.KW == 1 && s.pad == 0))) return false;
    if (s.stride != 1 && s.stride != 2) return false;
    return (long)s.N * s.H * s.W * (s.Cin > s.Cout ? s.Cin : s.Cout) < (1L << 31);
}

int block_pixels(int cout, long M) {
    if (cout >= 64) return M >= 32768 ? 128 : 64;
    return cout == 32 ? 128 : 256;
}

template <int BM, int BN, int BK, int WM, int WN>
int launch_cfg(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN);
    conv_nhwc_kernel<BM, BN, BK, WM, WN><<<grid, 256, 0, s>>>(a);
    return check_launch("conv_nhwc");
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
    if (a.M <= 0) return DMC_OK;
    const bool k32 = a.Cin % 32 == 0;
    const int bm = block_pixels(a.Cout, a.M);
    if (a.Cout >= 64) {
        if (bm == 128) return k32 ? launch_cfg<128, 64, 32, 2, 2>(a, s) : launch_cfg<128, 64, 16, 2, 2>(a, s);
        return k32 ? launch_cfg<64, 64, 32, 2, 2>(a, s) : launch_cfg<64, 64, 16, 2, 2>(a, s);
    }
    if (a.Cout >= 32) return k32 ? launch_cfg<128, 32, 32, 4, 1>(a, s) : launch_cfg<128, 32, 16, 4, 1>(a, s);
    return k32 ? launch_cfg<256, 16, 32, 4, 1>(a, s) : launch_cfg<256, 16, 16, 4, 1>(a, s);
}

}  // namespace

extern "C" {

int dmc_conv_nhwc_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    ConvShape s = {N, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0};
    return shape_supported(s) ? 1 : 0;
}

// number of [Cout][2] double partial rows the forward writes when asked for statistics
int dmc_conv_nhwc_stat_blocks(int N, int H, int W, int Cout, int KH, int stride, int pad) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long M = (long)N * OH * OW;
    const int bm = block_pixels(Cout, M);
    return (int)((M + bm - 1) / bm);
}

int dmc_conv_nhwc_fwd(const float* x, const float* w, const float* bias, const float* keep, float* y,
                      double* stat_partials, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                      int pad, int act, dmc_stream_t stream) {
    if (!x || !w || !y) return fail(DMC_E_INVALID, "dmc_conv_nhwc_fwd: null pointer");
    ConvShape sh = {N, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0};
    if (!shape_supported(sh))
        return fail(DMC_E_INVALID, "dmc_conv_nhwc_fwd: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d s=%d p=%d",
                    N, H, W, Cin, Cout, KH, KW, stride, pad);
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.keep = keep; a.stat_part = stat_partials;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.OH = (H + 2 * pad - KH) / stride + 1; a.OW = (W + 2 * pad - KW) / stride + 1;
    a.KK = KH * KW; a.ntaps = KH * KW; a.stride = stride;
    a.oy0 = 0; a.ox0 = 0; a.ostep = 1; a.OHs = a.OH; a.OWs = a.OW; a.act = act;
    a.tap_dy = a.tap_dx = a.tap_w = 0;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
            const int t = ky * KW + kx;
            a.set_tap(t, ky - pad, kx - pad, t);
        }
    a.M = N * a.OH * a.OW;
    return launch_conv(a, (hipStream_t)stream);
}

size_t dmc_conv_nhwc_wt_bytes(int Cin, int Cout, int KH, int KW) { return (size_t)Cin * Cout * KH * KW * sizeof(float); }

// dx [N,H,W,Cin] from dy [N,OH,OW,Cout]; wt: workspace of dmc_conv_nhwc_wt_bytes() (the packed weights)
int dmc_conv_nhwc_dgrad(const float* dy, const float* w, float* wt, float* dx, int N, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream) {
    if (!dy || !w || !wt || !dx) return fail(DMC_E_INVALID, "dmc_conv_nhwc_dgrad: null pointer");
    ConvShape sh = {N, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0};
    if (!shape_supported(sh)) return fail(DMC_E_INVALID, "dmc_conv_nhwc_dgrad: unsupported shape");
    hipStream_t s = (hipStream_t)stream;
    const int KK = KH * KW;
    const long total = (long)Cin * Cout * KK;
    conv_pack_wt_kernel<<<(int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256), 256, 0, s>>>(w, wt, Cout, KK, Cin);
    int rc = check_launch("conv_pack_wt");
    if (rc) return rc;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    ConvArgs a;
    a.x = dy; a.w = wt; a.y = dx; a.bias = nullptr; a.keep = nullptr; a.stat_part = nullptr;
    a.N = N; a.H = OH; a.W = OW; a.Cin = Cout;           // the "input" of this GEMM is dy
    a.OH = H; a.OW = W; a.Cout = Cin;                    // its "output" is dx
    a.KK = KK; a.stride = 1; a.act = 0;
    a.tap_dy = a.tap_dx = a.tap_w = 0;
    for (int py = 0; py < stride; ++py)
        for (int px = 0; px < stride; ++px) {
            a.oy0 = py; a.ox0 = px; a.ostep = stride;
            a.OHs = (H - py + stride - 1) / stride; a.OWs = (W - px + stride - 1) / stride;
            int nt = 0;
            for (int ky = 0; ky < KH; ++ky)
                for (int kx = 0; kx < KW; ++kx) {
                    if ((py + pad - ky) % stride != 0 || (px + pad - kx) % stride != 0) continue;
                    a.set_tap(nt, (py + pad - ky) / stride, (px + pad - kx) / stride, ky * KW + kx);
                    ++nt;
                }
            a.ntaps = nt;
            a.M = N * a.OHs * a.OWs;
            if ((rc = launch_conv(a, s))) return rc;
        }
    return DMC_OK;
}

// BatchNorm statistics from the partials the forward wrote (nblk = dmc_conv_nhwc_stat_blocks()):
// stats [2*C] = (mean, invstd); running_mean / running_var updated as nn.BatchNorm2d does.
int dmc_conv_nhwc_stats_final(const double* partials, int nblk, int C, long count, float* stats,
                              float* running_mean, float* running_var, float eps, float momentum,
                              dmc_stream_t stream) {
    if (!partials || !stats || !running_mean || !running_var || nblk <= 0 || C <= 0 || count <= 0)
        return fail(DMC_E_INVALID, "dmc_conv_nhwc_stats_final: bad argument");
    conv_stats_final_kernel<<<(C + 3) / 4, 256, 0, (hipStream_t)stream>>>(partials, nblk, C, count, stats,
                                                                          running_mean, running_var, eps, momentum);
    return check_launch("conv_stats_final");
}

// dw [Cout][KH][KW][Cin] (the memory of a channels_last weight gradient).  workspace:
// dmc_conv_nhwc_wgrad_bytes() bytes (the split-K partials; unused when one slice suffices).
size_t dmc_conv_nhwc_wgrad_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (wgrad_small_ok(Cin, KH, KW, pad)) return (size_t)512 * Cout * 9 * Cin * sizeof(float) + 16;
    const int tiles = ((Cout + WG_T - 1) / WG_T) * ((Cin + WG_T - 1) / WG_T) * KH * KW;
    const int ns = wgrad_slices((long)N * OH * OW, tiles);
    return (size_t)(ns > 1 ? ns : 0) * Cout * KH * KW * Cin * sizeof(float) + 16;
}

int dmc_conv_nhwc_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream) {
    if (!x || !dy || !dw || !workspace) return fail(DMC_E_INVALID, "dmc_conv_nhwc_wgrad: null pointer");
    ConvShape sh = {N, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0};
    if (!shape_supported(sh)) return fail(DMC_E_INVALID, "dmc_conv_nhwc_wgrad: unsupported shape");
    hipStream_t s = (hipStream_t)stream;
    WgradConvArgs a;
    a.x = x; a.dy = dy; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW;
    a.stride = stride; a.pad = pad;
    a.OH = (H + 2 * pad - KH) / stride + 1; a.OW = (W + 2 * pad - KW) / stride + 1;
    a.M = N * a.OH * a.OW;
    if (wgrad_small_ok(Cin, KH, KW, pad)) {
        a.part = nullptr; a.per_slice = 0; a.tiles_ci = a.tiles_co = 0;
        if (Cin == 16 && stride == 1) return Cout % 32 == 0 ? launch_wgrad_small<16, 32, 1>(a, dw, workspace, s) : launch_wgrad_small<16, 16, 1>(a, dw, workspace, s);
        if (Cin == 16) return Cout % 32 == 0 ? launch_wgrad_small<16, 32, 2>(a, dw, workspace, s) : launch_wgrad_small<16, 16, 2>(a, dw, workspace, s);
        if (stride == 1) return Cout % 32 == 0 ? launch_wgrad_small<32, 32, 1>(a, dw, workspace, s) : launch_wgrad_small<32, 16, 1>(a, dw, workspace, s);
        return Cout % 32 == 0 ? launch_wgrad_small<32, 32, 2>(a, dw, workspace, s) : launch_wgrad_small<32, 16, 2>(a, dw, workspace, s);
    }
    a.tiles_ci = (Cin + WG_T - 1) / WG_T; a.tiles_co = (Cout + WG_T - 1) / WG_T;
    const int tiles = a.tiles_ci * a.tiles_co * KH * KW;
    const int ns = wgrad_slices(a.M, tiles);
    a.per_slice = (int)((((long)a.M + ns - 1) / ns + WG_KP - 1) / WG_KP * WG_KP);
    a.part = ns > 1 ? workspace : dw;
    conv_wgrad_kernel<<<dim3(tiles, ns), 256, 0, s>>>(a);
    int rc = check_launch("conv_wgrad");
    if (rc || ns == 1) return rc;
    const long numel = (long)Cout * KH * KW * Cin;
    const long blocks = (numel / 4 + 255) / 256;
    conv_wgrad_reduce_kernel<<<(int)(blocks > 1024 ? 1024 : blocks), 256, 0, s>>>(workspace, dw, ns, numel);
    return check_launch("conv_wgrad_reduce");
}

}  // extern "C"
