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
#include "conv_small.h"
#include <type_traits>

using namespace dmc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x;        // [N][H][W][Cin]
    const float* w;        // [Cout][KH*KW][Cin]
    float* y;              // [N][OH][OW][Cout]
    const float* bias;     // [Cout] or null
    const float* keep;     // [N][Cout] or null   (Dropout2d keep mask, already divided by 1 - p)
    const float* addend = nullptr;   // [N][OH][OW][Cout] or null: added to the result before the store (conv_tile_epilogue only)
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
    int ablate;            // measurement only (option conv_ablate): 1 = no transfers, 2 = no barriers
    int stat_blocks = -1;  // host only: rows of stat_part the caller allocated (checked against the launch grid; -1 = no check)

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
                const double d1 = row16_sum((double)s1[j][e]), d2 = row16_sum((double)s2[j][e]);
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

// ------------------------------------------------------------------------------------------
// Second generation of the same implicit GEMM for Cin % 32 == 0 (the ResNet's 3x3 / 1x1 convolutions, the
// discriminator blocks from 32 channels on, and every data gradient whose dy has >= 32 channels):
//   * v_mfma_f32_32x32x2_f32: 64 cycles per instruction and per dependent accumulator, so one wave per SIMD
//     already keeps the matrix pipe full, and half the LDS operand traffic of the 16x16x4 form;
//     rows = output channels (A = weights), columns = pixels (B = input), 32 x 32 tiles;
//   * operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = eight
//     128-byte rows of 32 k-values per instruction), no staging registers, double-buffered: the
//     transfers of K-step t+1 are in flight during the MFMAs of step t, one barrier per step; each wave
//     issues its share in four pieces behind the MFMAs of the step's four k-groups;
//   * LDS-DMA writes lane-linear, so rows cannot be padded; the 16-byte quad q of tile row r is stored
//     in slot q ^ ((r >> 1) & 7) instead -- applied to the SOURCE address of each lane and again by the
//     fragment reads -- which makes every ds_read_b128 of 16 consecutive rows conflict-free;
//   * out-of-image taps and rows beyond M / Cout read 16 zero bytes (g_zeros).
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __attribute__((aligned(16))) float g_zeros[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16_v(const void* src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"((unsigned long long)src), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

// Epilogue shared by the implicit-GEMM kernels on 32 x 32 tiles: + bias, LeakyReLU(0.2), Dropout2d keep mask, the
// store, and the per-channel (sum, sum of squares) partials of the stored values for the BatchNorm that follows.
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_tile_epilogue(const ConvArgs& a, f32x16 (&acc)[TM][TN], void* lds, int m0, int co0,
                                                   int wm, int wn, int l31, int khalf, int tid) {
    const int prow0 = wm * (BM / WM);
    // ---- lane holds pixel column l31 of tile i; channels 8 gq + 4 khalf + e of tile j in acc[4 gq + e] ----
    // one channel tile j at a time, so that only 2 x 16 statistics accumulators are live next to acc
    long opix[TM];
    int nimg[TM];
    bool mok[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + prow0 + 32 * i + l31;
        mok[i] = m < a.M;
        nimg[i] = 0; opix[i] = 0;
        if (mok[i]) {
            const int n = m / (a.OHs * a.OWs);
            const int rem = m - n * (a.OHs * a.OWs);
            const int yy = rem / a.OWs, xx = rem - yy * a.OWs;
            nimg[i] = n;
            opix[i] = ((long)n * a.OH + (a.oy0 + yy * a.ostep)) * a.OW + (a.ox0 + xx * a.ostep);
        }
    }
    double* red = reinterpret_cast<double*>(lds);             // [WM][BN][2]
    if (a.stat_part) __syncthreads();                         // every wave is done with the operand tiles
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float s1[16], s2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = co0 + wn * (BN / WN) + 32 * j + 8 * gq + 4 * khalf;
                if (co >= a.Cout) continue;
                float v[4] = {acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
                if (a.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.bias + co);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (a.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (a.keep && mok[i]) {
                    const float4 kv = *reinterpret_cast<const float4*>(a.keep + (long)nimg[i] * a.Cout + co);
                    v[0] *= kv.x; v[1] *= kv.y; v[2] *= kv.z; v[3] *= kv.w;
                }
                if (a.addend && mok[i]) {
                    const float4 av = *reinterpret_cast<const float4*>(a.addend + opix[i] * a.Cout + co);
                    v[0] += av.x; v[1] += av.y; v[2] += av.z; v[3] += av.w;
                }
                if (mok[i]) {
                    *reinterpret_cast<float4*>(a.y + opix[i] * a.Cout + co) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[4 * gq + e] += v[e]; s2[4 * gq + e] += v[e] * v[e]; }
                }
            }
        if (a.stat_part) {
            // per-channel sums of this workgroup's tile: 32 pixel lanes -> waves (WM) -> one store per channel
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const double d1 = half32_sum_hi((double)s1[e]), d2 = half32_sum_hi((double)s2[e]);
                if (l31 == 31) {
                    const int c = wn * (BN / WN) + 32 * j + 8 * (e >> 2) + 4 * khalf + (e & 3);
                    red[(wm * BN + c) * 2 + 0] = d1;
                    red[(wm * BN + c) * 2 + 1] = d2;
                }
            }
        }
    }
    if (a.stat_part) {
        __syncthreads();
        for (int c = tid; c < BN; c += WM * WN * 64)
            if (co0 + c < a.Cout) {
                double d1 = 0.0, d2 = 0.0;
#pragma unroll
                for (int w = 0; w < WM; ++w) { d1 += red[(w * BN + c) * 2 + 0]; d2 += red[(w * BN + c) * 2 + 1]; }
                double* dst = a.stat_part + ((size_t)blockIdx.x * a.Cout + co0 + c) * 2;
                dst[0] = d1; dst[1] = d2;
            }
    }
}

// ---- fp32 products from bf16 matrix-core instructions ("bf16x3") -----------------------------------
// An fp32 value is the exact sum of three bf16 values: s0 = its upper 16 bits (sign, exponent, 7 mantissa bits:
// 8 significant bits, truncated), s1 = the upper 16 bits of the (exact) remainder, s2 = the second remainder,
// which has at most 8 significant bits left and is therefore a bf16 value itself.  The product of two such
// sums is formed from six of the nine slice products, (0,0) (0,1) (1,0) (0,2) (2,0) (1,1) -- each one exact in
// the fp32 accumulator's input precision -- and the three omitted ones are below 2^-23 of |a||b|, the size of
// one fp32 rounding of the product.  v_mfma_f32_32x32x16_bf16 retires 16 k-values per 32 cycles where
// v_mfma_f32_32x32x2_f32 retires 2 per 64: six of them per 16 k-values are 2.67x the fp32 instruction's rate.
// (Inf inputs become NaN, values below 2^-110 lose their low slices: neither occurs in a finite training run.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split3 { u32x4 s[3]; };
__device__ __forceinline__ Split3 split_bf16x3(const f32x4& lo, const f32x4& hi) {
    const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        u0[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u0[e] & 0xffff0000u);
        u1[e] = __float_as_uint(r1);
        u2[e] = __float_as_uint(r1 - __uint_as_float(u1[e] & 0xffff0000u));
    }
    Split3 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {                         // (upper half of value 2e+1, upper half of value 2e)
        r.s[0][e] = __builtin_amdgcn_perm(u0[2 * e + 1], u0[2 * e], 0x07060302u);
        r.s[1][e] = __builtin_amdgcn_perm(u1[2 * e + 1], u1[2 * e], 0x07060302u);
        r.s[2][e] = __builtin_amdgcn_perm(u2[2 * e + 1], u2[2 * e], 0x07060302u);
    }
    return r;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv2_kernel(ConvArgs a) {
    constexpr int NW = WM * WN;                            // waves: 4, or 8 (smaller wave tiles, twice the waves per SIMD)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;   // 32 x 32 tiles per wave: pixels x channels
    static_assert((NW == 4 || NW == 8) && BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "wave layout");
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "transfer split");
    constexpr int NDP = BM / 8 / NW, NDW = BN / 8 / NW;    // DMA instructions per wave and step: pixel rows, weight rows
    constexpr int ND = NDP + NDW;
    constexpr int BUF = (BM + BN) * 128;                   // bytes per buffer: tile rows of 32 floats
    __shared__ __attribute__((aligned(1024))) float lds[2 * (BM + BN) * 32];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = blockIdx.x * BM, co0 = blockIdx.y * BN;
    const unsigned lds0 = lds_addr_of(lds);
    const unsigned long long zeros = (unsigned long long)g_zeros;

    // ---- this lane's share of the transfers: instruction d = wave + NW j moves tile rows 8 d .. 8 d + 7,
    // lane -> (row 8 d + lane / 8, slot lane % 8), source quad = slot ^ ((row >> 1) & 7).  Per pixel row the
    // lane keeps the address of its quad at tap offset (0, 0) and a bit mask of the taps that fall inside
    // the image, so a step costs one add and one select per transfer ----
    const int rr = lane >> 3, sl = lane & 7;
    unsigned long long p_addr[NDP];
    unsigned p_mask[NDP];
#pragma unroll
    for (int j = 0; j < NDP; ++j) {
        const int row = 8 * (wave + NW * j) + rr;
        const int q = sl ^ ((row >> 1) & 7);
        const int m = m0 + row;
        p_mask[j] = 0; p_addr[j] = zeros;
        if (m < a.M) {
            const int n = m / (a.OHs * a.OWs), rem = m - n * (a.OHs * a.OWs);
            const int yy = rem / a.OWs, xx = rem - yy * a.OWs;
            const int iy0 = yy * a.stride, ix0 = xx * a.stride;
            p_addr[j] = (unsigned long long)a.x + ((((long)n * a.H + iy0) * a.W + ix0) * a.Cin + 4 * q) * 4;
            for (int t = 0; t < a.ntaps; ++t) {
                const int iy = iy0 + (int)((a.tap_dy >> (4 * t)) & 15) - 8, ix = ix0 + (int)((a.tap_dx >> (4 * t)) & 15) - 8;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) p_mask[j] |= 1u << t;
            }
        }
    }
    unsigned long long w_addr[NDW];   // (co, tap 0, ci 0) + the lane's quad; Cout % BN == 0: every row exists
#pragma unroll
    for (int j = 0; j < NDW; ++j) {
        const int row = 8 * (wave + NW * j) + rr;           // row within the weight tile
        const int q = sl ^ ((row >> 1) & 7);
        w_addr[j] = (unsigned long long)a.w + ((long)(co0 + row) * a.KK * a.Cin + 4 * q) * 4;
    }

    // step state of the transfers being issued (the NEXT K-step): all scalar
    int tap_n = 0, ci_n = 0;
    long toff = 0, woff = 0;
    auto set_step = [&]() {
        const int dy = (int)((a.tap_dy >> (4 * tap_n)) & 15) - 8, dx = (int)((a.tap_dx >> (4 * tap_n)) & 15) - 8;
        const int tw = (int)((a.tap_w >> (4 * tap_n)) & 15);
        toff = (((long)dy * a.W + dx) * a.Cin + ci_n) * 4;
        woff = ((long)tw * a.Cin + ci_n) * 4;
    };
    auto issue_one = [&](int j, int buf) {                          // j is a compile-time constant at every call
        const unsigned base = lds0 + buf * BUF;
        if (j < NDP) {
            const bool ok = (p_mask[j < NDP ? j : 0] >> tap_n) & 1;
            const unsigned long long src = ok ? p_addr[j < NDP ? j : 0] + (unsigned long long)toff : zeros;
            dma16_v(reinterpret_cast<const void*>(src), base + (wave + NW * j) * 1024);
        } else {
            const int jw = j - NDP;
            dma16_v(reinterpret_cast<const void*>(w_addr[jw >= 0 && jw < NDW ? jw : 0] + (unsigned long long)woff),
                    base + BM * 128 + (wave + NW * jw) * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment reads: lane -> tile row (lane & 31), k-quad 2 g + (lane >> 5), stored in slot quad ^ ((row >> 1) & 7);
    // tile bases are multiples of 32 rows, so the swizzle term depends on the lane only
    const int l31 = lane & 31, khalf = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    int foff[4];                                                  // byte offset within a 32-row tile, per k-group g
#pragma unroll
    for (int g = 0; g < 4; ++g) foff[g] = l31 * 128 + (((2 * g + khalf) ^ swz) << 4);
    const int prow0 = wm * (BM / WM), crow0 = BM + wn * (BN / WN);

    const int T_steps = a.ntaps * (a.Cin >> 5);
    if (T_steps > 0) {
        set_step();
#pragma unroll
        for (int j = 0; j < ND; ++j) issue_one(j, 0);
        ci_n = 32;
        if (ci_n == a.Cin) { ci_n = 0; tap_n = 1; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int t = 0; t < T_steps; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < T_steps;
        if (more) set_step();
        const char* base = reinterpret_cast<const char*>(lds) + buf * BUF;
        // fragment registers are double-buffered over the four k-groups: the LDS reads of group g + 1 are issued
        // before the MFMAs of group g (with one register set every group started by waiting ~100 cycles for
        // its operands with a single MFMA left in the pipe)
        float4 xa[2][TM], wa[2][TN];
        auto load_frags = [&](int g, int sel) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                xa[sel][i] = *reinterpret_cast<const float4*>(base + (prow0 + 32 * i) * 128 + foff[g]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wa[sel][j] = *reinterpret_cast<const float4*>(base + (crow0 + 32 * j) * 128 + foff[g]);
        };
        load_frags(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1;
            if (g + 1 < 4) load_frags(g + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][j].x, xa[cur][i].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][j].y, xa[cur][i].y, acc[i][j], 0, 0, 0);
            // the next step's transfers, a quarter of them behind each k-group's MFMAs (they land during the
            // remaining groups; nothing is issued, or waited for, in front of the step's first MFMA).  A
            // dedicated producer wave was measured too: its ~40 transfers per step serialise on one wave's
            // issue (60-100 cycles each) and the consumers wait at the barrier (0.29 -> 0.46 ms for layer1)
            if (more && !(DMC_ABL(a.ablate) & 1)) {
#pragma unroll
                for (int j = g * ND / 4; j < (g + 1) * ND / 4; ++j) issue_one(j, buf ^ 1);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][j].z, xa[cur][i].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[cur][j].w, xa[cur][i].w, acc[i][j], 0, 0, 0);
        }
        if (more) {
            ci_n += 32;
            if (ci_n == a.Cin) { ci_n = 0; ++tap_n; }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // step t+1 has landed (this wave's share), own reads of buf done
        __builtin_amdgcn_s_barrier();                              // ... everyone's; and nobody reads buf any more
    }

    conv_tile_epilogue<BM, BN, WM, WN, TM, TN>(a, acc, lds, m0, co0, wm, wn, l31, khalf, tid);
}

// ------------------------------------------------------------------------------------------
// Third generation: the same implicit GEMM in bf16x3 arithmetic (above) for Cin % 32 == 0, Cout % 64 == 0.
//   * weights arrive already split (conv_split_w_kernel: three bf16 slices [3][Cout][KK][Cin], a few microseconds per
//     call), so only the activation fragments are split in registers (11 VALU instructions per pair of values,
//     overlapped with the matrix pipe); activations stay fp32 in HBM and LDS: nothing upstream changes;
//   * K-step = one tap x 32 channels = two k-blocks of 16; per buffer BM pixel rows of 128 bytes (slot swizzle as in
//     conv2) + 3 x BN weight rows of 64 bytes (four 16-byte slots, slot = quad ^ ((row >> 2) & 3): rows r, r+4, r+8,
//     r+12 share their banks and take different slots, so a ds_read_b128 of 16 rows is conflict-free);
//   * the loop is software-pipelined across k-blocks AND steps: while the six-MFMA groups of one k-block run, the
//     next block's fragments are read and split; the step's only barrier stands in front of its LAST k-block, where
//     this wave's transfers of step t+1 were issued a whole step earlier (nobody waits for memory there), and
//     the first fragments of step t+1 are read from the other buffer during that last block; the transfers of
//     step t+2 go into the buffer the barrier just freed.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int ABL = 0>     // ABL: measurement only (option conv_ablate bits 4, 8)
__global__ __launch_bounds__(WM * WN * 64) void conv3_kernel(ConvArgs a) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BM % (8 * NW) == 0 && BN % 16 == 0, "tile layout");
    constexpr int NDP = BM / 8 / NW;                       // pixel transfers per wave and step (8 rows of 128 B)
    constexpr int WD = 3 * BN / 16;                        // weight transfers per step (16 rows of 64 B), whole workgroup
    constexpr int NDW = (WD + NW - 1) / NW;
    constexpr int PIXB = BM * 128, BUF = PIXB + 3 * BN * 64;
    extern __shared__ __attribute__((aligned(1024))) float lds3[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = blockIdx.x * BM, co0 = blockIdx.y * BN;
    const unsigned lds0 = lds_addr_of(lds3);
    const unsigned long long zeros = (unsigned long long)g_zeros;

    // ---- transfers ----
    const int rr = lane >> 3, sl = lane & 7;
    unsigned long long p_addr[NDP];
    unsigned p_mask[NDP];
#pragma unroll
    for (int j = 0; j < NDP; ++j) {
        const int row = 8 * (wave + NW * j) + rr;
        const int q = sl ^ ((row >> 1) & 7);
        const int m = m0 + row;
        p_mask[j] = 0; p_addr[j] = zeros;
        if (m < a.M) {
            const int n = m / (a.OHs * a.OWs), rem = m - n * (a.OHs * a.OWs);
            const int yy = rem / a.OWs, xx = rem - yy * a.OWs;
            const int iy0 = yy * a.stride, ix0 = xx * a.stride;
            p_addr[j] = (unsigned long long)a.x + ((((long)n * a.H + iy0) * a.W + ix0) * a.Cin + 4 * q) * 4;
            for (int t = 0; t < a.ntaps; ++t) {
                const int iy = iy0 + (int)((a.tap_dy >> (4 * t)) & 15) - 8, ix = ix0 + (int)((a.tap_dx >> (4 * t)) & 15) - 8;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) p_mask[j] |= 1u << t;
            }
        }
    }
    // weight transfer e moves rows 16 e .. 16 e + 15 of the stacked [3][BN] tile: lane -> (row 16 e + lane / 4, slot lane % 4)
    unsigned long long w_addr[NDW];
#pragma unroll
    for (int j = 0; j < NDW; ++j) {
        const int R = 16 * (wave + NW * j) + (lane >> 2);
        const int slice = R / BN, c = R - slice * BN;
        const int q = (lane & 3) ^ ((c >> 2) & 3);
        w_addr[j] = (unsigned long long)a.w + (((long)(slice < 3 ? slice : 0) * a.Cout + co0 + c) * a.KK * a.Cin + 8 * q) * 2;
    }

    int tap_n = 0, ci_n = 0;
    long toff = 0, woff = 0;
    auto set_step = [&]() {
        const int dy = (int)((a.tap_dy >> (4 * tap_n)) & 15) - 8, dx = (int)((a.tap_dx >> (4 * tap_n)) & 15) - 8;
        const int tw = (int)((a.tap_w >> (4 * tap_n)) & 15);
        toff = (((long)dy * a.W + dx) * a.Cin + ci_n) * 4;
        woff = ((long)tw * a.Cin + ci_n) * 2;
    };
    auto advance = [&]() {
        ci_n += 32;
        if (ci_n == a.Cin) { ci_n = 0; ++tap_n; }
    };
    auto issue_all = [&](int buf) {
        const unsigned base = lds0 + buf * BUF;
#pragma unroll
        for (int j = 0; j < NDP; ++j) {
            const bool ok = (p_mask[j] >> tap_n) & 1;
            const unsigned long long src = ok ? p_addr[j] + (unsigned long long)toff : zeros;
            dma16_v(reinterpret_cast<const void*>(src), base + (wave + NW * j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < NDW; ++j)
            if (WD % NW == 0 || wave + NW * j < WD)
                dma16_v(reinterpret_cast<const void*>(w_addr[j] + (unsigned long long)woff), base + PIXB + (wave + NW * j) * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addressing (byte offsets inside a buffer) ----
    const int l31 = lane & 31, khalf = lane >> 5;
    const int prow0 = wm * (BM / WM), crow0 = wn * (BN / WN);
    int xoff[2][2], wfo[2];                                // [k-block][quad of the pair] / [k-block]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        xoff[kb][0] = (prow0 + l31) * 128 + (((4 * kb + 2 * khalf) ^ ((l31 >> 1) & 7)) << 4);
        xoff[kb][1] = (prow0 + l31) * 128 + (((4 * kb + 2 * khalf + 1) ^ ((l31 >> 1) & 7)) << 4);
        wfo[kb] = PIXB + (crow0 + l31) * 64 + (((2 * kb + khalf) ^ ((l31 >> 2) & 3)) << 4);
    }
    // fragment reads: per-lane byte offsets (swizzled slots) + compile-time buffer / tile offsets, so that every
    // ds_read_b128 is "lane address + immediate"
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    lds_cptr const L = (lds_cptr)lds3;
    struct Raw { f32x4 lo[TM], hi[TM]; };
    bool first_reads = true;
    auto load_raw = [&](auto bufc, auto kbc, Raw& r) {
        constexpr int buf = decltype(bufc)::value, kb = decltype(kbc)::value;
        if ((ABL & 8) && !first_reads) return;             // ablation: fragments read once (results wrong)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            r.lo[i] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(L + xoff[kb][0] + (buf * BUF + i * 32 * 128));
            r.hi[i] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(L + xoff[kb][1] + (buf * BUF + i * 32 * 128));
        }
    };
    u32x4 wf[TN][3];                                       // weight fragments: ONE set; a slice is re-read for the next
                                                           // k-block right after its last MFMA of this one
    auto load_w = [&](auto bufc, auto kbc, int sidx) {
        constexpr int buf = decltype(bufc)::value, kb = decltype(kbc)::value;
        if ((ABL & 8) && !first_reads) return;
#pragma unroll
        for (int j = 0; j < TN; ++j)
            wf[j][sidx] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(L + wfo[kb] + (buf * BUF + (sidx * BN + 32 * j) * 64));
    };
    auto split_all = [&](const Raw& r, Split3 (&xs)[TM]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (ABL & 4) {                       // ablation: no split (results wrong)
                xs[i].s[0] = __builtin_bit_cast(u32x4, r.lo[i]); xs[i].s[1] = __builtin_bit_cast(u32x4, r.hi[i]);
                xs[i].s[2] = __builtin_bit_cast(u32x4, r.lo[i]);
            } else {
                xs[i] = split_bf16x3(r.lo[i], r.hi[i]);
            }
        }
    };
    auto mfma_terms = [&](const Split3 (&xs)[TM], int wsl, int xlo, int xhi) {   // weight slice wsl x input slices xhi .. xlo
#pragma unroll
        for (int xsl = xhi; xsl >= xlo; --xsl)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16(wf[j][wsl], xs[i].s[xsl], acc[i][j]);
    };
    // the six slice products of a k-block, grouped by weight slice: (0,2) (0,1) (0,0) | (1,1) (1,0) | (2,0); each weight
    // slice of the next block is read right behind its last product of this one.  The split of the NEXT block's
    // activations (11 VALU instructions per value pair) is spread evenly behind this block's MFMAs by
    // sched_group_barriers -- one MFMA (32 cycles of the matrix pipe), then a few VALU instructions (4 cycles of
    // issue each): bunched up, as the compiler schedules them on its own, a wave issues VALU work for longer than
    // its MFMAs last and the pipe drains; `skip` leading MFMAs carry no VALU work (they cover the LDS latency of
    // fragments read just in front of the block)
    constexpr int NMF = 6 * TM * TN, NVALU = 44 * TM;
    auto block = [&](const Split3 (&xs)[TM], bool have_next, auto nbuf, auto nkb, const Raw& r, Split3 (&xn)[TM], auto skipc) {
        constexpr int skip = decltype(skipc)::value;
        constexpr int per = (NVALU + (NMF - skip) - 1) / (NMF - skip);
        mfma_terms(xs, 0, 0, 2);
        if (have_next) load_w(nbuf, nkb, 0);
        if (have_next) split_all(r, xn);
        mfma_terms(xs, 1, 0, 1);
        if (have_next) load_w(nbuf, nkb, 1);
        mfma_terms(xs, 2, 0, 0);
        if (have_next) load_w(nbuf, nkb, 2);
        if (have_next) {                                   // materialise the slices HERE (they are used beyond the barrier
#pragma unroll                                             // and would be sunk to their use otherwise)
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(xn[i].s[0]), "+v"(xn[i].s[1]), "+v"(xn[i].s[2]));
#pragma unroll
            for (int k = 0; k < NMF; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (k >= skip) __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using ISKIP = std::integral_constant<int, (NMF >= 12 ? 4 : 2)>;

    const int T_steps = a.ntaps * (a.Cin >> 5);
    const bool dma_on = !(DMC_ABL(a.ablate) & 1);
    if (T_steps == 0) {                                    // a parity class of a strided data gradient without taps: zeros
        conv_tile_epilogue<BM, BN, WM, WN, TM, TN>(a, acc, lds3, m0, co0, wm, wn, l31, khalf, tid);
        return;
    }
    Raw raw0, raw1;                                        // fp32 fragments of the next step's two k-blocks
    Split3 x0[TM], x1[TM];
    set_step(); issue_all(0); advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_raw(I0{}, I0{}, raw0);
    load_raw(I0{}, I1{}, raw1);
    load_w(I0{}, I0{}, 0); load_w(I0{}, I0{}, 1); load_w(I0{}, I0{}, 2);
    if (T_steps > 1) { set_step(); issue_all(1); advance(); }
    first_reads = false;
    split_all(raw0, x0);
    // one step on buffer `buf` (compile-time).  k-block 0 runs while block 1's fragments (read a block ago) are split;
    // then the step's only barrier -- this wave's transfers of step t+1 were issued a step ago, its reads of `buf` are
    // complete --; k-block 1 runs while BOTH fragments of step t+1 are read from the other buffer, its first block
    // is split, and the transfers of step t+2 are issued into the buffer just released
    auto step = [&](auto bufc, auto lastc, int t) {
        constexpr int buf = decltype(bufc)::value;
        constexpr bool more = !decltype(lastc)::value;         // compile-time: the loop body has no conditional reads
        using NB = std::integral_constant<int, buf ^ 1>;
        block(x0, true, bufc, I1{}, raw1, x1, I0{});
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) { load_raw(NB{}, I0{}, raw0); load_raw(NB{}, I1{}, raw1); }
        if (t + 2 < T_steps && dma_on) { set_step(); issue_all(buf); advance(); }
        block(x1, more, NB{}, I0{}, raw0, x0, ISKIP{});
    };
    // T_steps is even (Cin % 64 == 0): pairs of steps on buffers 0 and 1; the last pair is peeled
    int t = 0;
#pragma unroll 1
    for (; t + 2 < T_steps; t += 2) {
        step(I0{}, I0{}, t);
        step(I1{}, I0{}, t + 1);
    }
    step(I0{}, I0{}, t);
    step(I1{}, I1{}, t + 1);
    conv_tile_epilogue<BM, BN, WM, WN, TM, TN>(a, acc, lds3, m0, co0, wm, wn, l31, khalf, tid);
}

// weights as three bf16 slices: ws[s][row][tap][col] (16-bit words); rows / cols = (co, ci) of w[co][tap][ci], or,
// transposed, (ci, co) for the data gradient (the conv_pack_wt_kernel order)
__global__ __launch_bounds__(256) void conv_split_w_kernel(const float* __restrict__ w, unsigned short* __restrict__ ws,
                                                           int Cout, int KK, int Cin, int transposed) {
    const long total = (long)Cout * KK * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v;
        if (transposed) {
            const int co = (int)(i % Cout), tap = (int)((i / Cout) % KK), ci = (int)(i / ((long)Cout * KK));
            v = w[((long)co * KK + tap) * Cin + ci];
        } else {
            v = w[i];
        }
        const unsigned u0 = __float_as_uint(v);
        const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = __float_as_uint(r1);
        const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
        ws[i] = (unsigned short)(u0 >> 16);
        ws[total + i] = (unsigned short)(u1 >> 16);
        ws[2 * total + i] = (unsigned short)(u2 >> 16);
    }
}

// both layouts in one launch (blockIdx.y = 0: the forward's [3][Cout][KK][Cin]; 1: the data gradient's transposed [3][Cin][KK][Cout])
__global__ __launch_bounds__(256) void conv_split_w2_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                                            unsigned short* __restrict__ wt, int Cout, int KK, int Cin) {
    const bool transposed = blockIdx.y == 1;
    unsigned short* ws = transposed ? wt : wf;
    const long total = (long)Cout * KK * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v;
        if (transposed) {
            const int co = (int)(i % Cout), tap = (int)((i / Cout) % KK), ci = (int)(i / ((long)Cout * KK));
            v = w[((long)co * KK + tap) * Cin + ci];
        } else {
            v = w[i];
        }
        const unsigned u0 = __float_as_uint(v);
        const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = __float_as_uint(r1);
        const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
        ws[i] = (unsigned short)(u0 >> 16);
        ws[total + i] = (unsigned short)(u1 >> 16);
        ws[2 * total + i] = (unsigned short)(u2 >> 16);
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
    // one workgroup per channel: 256 lanes stride over the (up to ~12,000) workgroup partials, then a fixed-order
    // tree (a single wave per channel walked them in 180 dependent-latency steps: 20 us per call)
    __shared__ double red[2][4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const double2 v = *reinterpret_cast<const double2*>(part + ((size_t)b * C + c) * 2);
        s += v.x;
        ss += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); ss += __shfl_down(ss, o, 64); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
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

// sum of the nslice partials: 16 float4 outputs x 16 slice lanes per workgroup (each lane sums every 16th slice in
// order, then the 16 lanes are combined in order through LDS): fixed summation order, numel / 64 workgroups
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                int nslice, long numel) {
    __shared__ float4 red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (long i0 = (long)blockIdx.x * 64; i0 < numel; i0 += (long)gridDim.x * 64) {
        const long i = i0 + 4 * o;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < numel)
            for (int k = sl; k < nslice; k += 16) {
                const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * numel + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[sl][o] = s;
        __syncthreads();
        if (sl == 0 && i < numel) {
            float4 t = red[0][o];
#pragma unroll
            for (int k = 1; k < 16; ++k) { const float4 v = red[k][o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(dw + i) = t;
        }
        __syncthreads();
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
    const long blocks = (numel + 63) / 64;
    conv_wgrad_reduce_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(workspace, dw, wgs, numel);
    return check_launch("conv_wgrad_reduce");
}

bool wgrad_small_ok(int Cin, int KH, int KW, int pad) { return (Cin == 16 || Cin == 32) && KH == 3 && KW == 3 && pad == 1; }

// ------------------------------------------------------------------------------------------
// Weight gradient, second generation (Cin % 64 == 0, Cout % 64 == 0; 3x3 pad 1 or 1x1 pad 0; stride 1 / 2).
//
// A workgroup owns a 64 (co) x 64 (ci) block of dw for ALL taps and a run of "steps"; a step is R full
// output rows of one image.  Per step it stages, by LDS-DMA and double-buffered, the dy tile
// [R*OW pixels][64 co] and the input patch [(R-1)s+KS rows][(OW-1)s+KS columns][64 ci] (zeros outside the
// image) ONCE, and each of the 4 waves accumulates its 32 x 32 quarter of the block for every tap from
// that patch: per pair of pixels one A read (dy), NT B reads (x at the tap offsets), NT
// v_mfma_f32_32x32x2_f32 on NT independent accumulators (9 x 16 registers, resident over the whole run).
// The activations are read from L2 about once (rows shared by consecutive steps) instead of once per
// tap.  Unpadded 256-byte pixel rows: the operand reads are 32 consecutive floats per half-wave
// (conflict-free), so nothing needs a swizzle.  Partials [group][Cout][KK][Cin] are summed in group order
// by conv_wgrad_reduce_kernel: deterministic, no atomics, nothing to zero.
// ------------------------------------------------------------------------------------------
struct Wgrad2Args {
    const float* x;        // [N][H][W][Cin]
    const float* dy;       // [N][OH][OW][Cout]
    float* part;           // [groups][Cout][KK][Cin]
    int N, H, W, Cin, OH, OW, Cout;
    int R, P, Ppad;        // output rows per step, pixels per step (R * OW), rounded up to a multiple of 4
    int XR, XC;            // patch rows / columns
    int groups_per_img;    // ceil(OH / R)
    int steps, per_group;  // total steps, steps per workgroup
    int tiles_ci;
    int dma_per_it;        // transfers a wave issues per pixel-pair iteration
    unsigned inv_xc, inv_ow;   // ceil(2^32 / XC), ceil(2^32 / OW): exact quotients for the small indices used here
};

// 3x3: 8 waves = 4 quarters of the 64 x 64 block x 2 tap groups (taps 0-4 / 5-8), two waves per SIMD: while one
// waits for its LDS operands or issues transfers, the other's MFMAs keep the pipe busy (with 4 waves of 9
// accumulators each, one per SIMD, the pipe idled 40 % of the time).  1x1: 4 waves.
template <int KS>
struct Wg2 {
    static constexpr int NT = KS * KS;
    static constexpr int WAVES = KS == 3 ? 8 : 4;
    static constexpr int NTL = KS == 3 ? 5 : 1;             // accumulators per wave
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(Wg2<KS>::WAVES * 64) void conv_wgrad2_kernel(Wgrad2Args a) {
    constexpr int NT = Wg2<KS>::NT, PAD = KS / 2, WAVES = Wg2<KS>::WAVES, NTL = Wg2<KS>::NTL;
    extern __shared__ __attribute__((aligned(1024))) float lds2[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int quarter = wave & 3, tg = wave >> 2;                  // tap group: taps tap0 .. tap0 + ntl - 1
    const int wi = quarter & 1, wj = quarter >> 1;                 // this wave's co half / ci half of the block
    const int tap0 = tg * NTL, ntl = NT - tap0 < NTL ? NT - tap0 : NTL;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int dy_rows = a.Ppad;                                    // pixel rows of the dy tile
    const int x_rows = a.XR * a.XC;                                // pixel rows of the patch
    const int dy_dma = (dy_rows + 3) >> 2, x_dma = (x_rows + 3) >> 2;   // 1 KB transfers (4 pixel rows each)
    const int buf_floats = (dy_dma + x_dma) * 256;                  // every float of a buffer is written by each step's transfers
    const unsigned lds0 = lds_addr_of(lds2);
    const char* zeros = reinterpret_cast<const char*>(g_zeros);

    const int s_begin = blockIdx.y * a.per_group;
    const int s_end = s_begin + a.per_group < a.steps ? s_begin + a.per_group : a.steps;
    const int sub = lane >> 4, q16 = lane & 15;                    // DMA: lane -> (pixel row 4 d + sub, 16-byte quad q16)

    // transfers of one step, as slots d = 0 .. dy_dma + x_dma - 1 (1 KB each); wave w issues slots w, w + WAVES, ...
    // The step's scalars are set by set_step(); issue_slot(d) is then one transfer.  The next step's slots are
    // issued a few per pixel-pair iteration of the current step, behind its MFMAs.
    const int n_slots = dy_dma + x_dma;
    const char* dyb = nullptr;
    const char* xb = nullptr;
    int prem = 0, iy0 = 0;
    unsigned ibase = 0;
    auto set_step = [&](int step, int buf) {
        const int n = step / a.groups_per_img, oy0 = (step - n * a.groups_per_img) * a.R;
        ibase = lds0 + (unsigned)buf * buf_floats * 4;
        dyb = reinterpret_cast<const char*>(a.dy + (((long)n * a.OH + oy0) * a.OW) * a.Cout + co0 + 4 * q16);
        prem = (a.OH - oy0) * a.OW;                                // pixels of this image at or below row oy0
        iy0 = oy0 * STRIDE - PAD;
        xb = reinterpret_cast<const char*>(a.x + ((long)n * a.H * a.W) * a.Cin + ci0 + 4 * q16);
    };
    auto issue_slot = [&](int d) {
        if (d < dy_dma) {                                           // wave-uniform
            const int p = 4 * d + sub;
            const bool ok = p < a.P && p < prem;
            dma16_v(ok ? dyb + (long)p * a.Cout * 4 : zeros, ibase + d * 1024);
        } else {
            const int pp = 4 * (d - dy_dma) + sub;
            const int pr = (int)(((unsigned long long)pp * a.inv_xc) >> 32), pc = pp - pr * a.XC;
            const int iy = iy0 + pr, ix = pc - PAD;
            const bool ok = pp < x_rows && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            dma16_v(ok ? xb + ((long)iy * a.W + ix) * a.Cin * 4 : zeros, ibase + d * 1024);
        }
    };

    f32x16 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    int tapoff[NTL];                                                // float offset of this wave's taps in the patch
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
        const int tap = tap0 + (t < ntl ? t : ntl - 1);             // (an unused slot repeats the last tap: valid address)
        tapoff[t] = ((tap / KS) * a.XC + (tap % KS)) * 64;
    }

    if (s_begin < s_end) {
        set_step(s_begin, 0);
#pragma unroll 1
        for (int d = wave; d < n_slots; d += WAVES) issue_slot(d);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int wrap = (STRIDE * a.XC - a.OW * STRIDE) * 64;
    const int n_it = a.Ppad >> 2;                                   // iterations of two pixel pairs
#pragma unroll 1
    for (int step = s_begin; step < s_end; ++step) {
        const int buf = (step - s_begin) & 1;
        const bool more = step + 1 < s_end;
        if (more) set_step(step + 1, buf ^ 1);
        int d_n = more ? wave : n_slots;
        const float* D = lds2 + buf * buf_floats + 32 * wi + l31;
        const float* X = lds2 + buf * buf_floats + dy_dma * 256 + 32 * wj + l31;
        // this lane's pixel of the pair: q = 2 s + khalf -> (r, c); patch offset ((r s) XC + c s) * 64 floats
        int c = khalf, q = khalf;                                   // q = khalf < OW (OW >= 2)
        int xoff = c * STRIDE * 64;
        // operands of two pixel pairs; padding pixels (q >= P: dy is zero there) read the patch origin
        // instead of running past its end
        auto load_pairs = [&](float (&av)[2], float (&bv)[2][NTL]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                av[u] = D[q * 64];
                const int xo = q < a.P ? xoff : 0;
#pragma unroll
                for (int t = 0; t < NTL; ++t) bv[u][t] = X[xo + tapoff[t]];
                q += 2; c += 2; xoff += 2 * STRIDE * 64;
                if (c >= a.OW) { c -= a.OW; xoff += wrap; }
            }
        };
        auto mfma_pairs = [&](const float (&av)[2], const float (&bv)[2][NTL]) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < NTL; ++t)
                    if (t < NTL - 1 || ntl == NTL)                  // the last slot is idle in the 4-tap group (wave-uniform)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][t], acc[t], 0, 0, 0);
        };
        auto issue_some = [&]() {
            if (d_n < n_slots) { issue_slot(d_n); d_n += WAVES; }
            if (a.dma_per_it > 1 && d_n < n_slots) { issue_slot(d_n); d_n += WAVES; }
        };
        // register double buffer: the LDS reads of iteration i + 1 are in flight during the MFMAs of iteration i.
        // Every load is unconditional (iterations past the end re-read valid LDS and are never multiplied), so
        // the two register sets stay distinct and no copy or early wait separates a read from its use.
        float av0[2], bv0[2][NTL], av1[2], bv1[2][NTL];
        load_pairs(av0, bv0);
#pragma unroll 1
        for (int it = 0; it < n_it; it += 2) {
            load_pairs(av1, bv1);
            mfma_pairs(av0, bv0);
            issue_some();
            load_pairs(av0, bv0);
            if (it + 1 < n_it) mfma_pairs(av1, bv1);
            issue_some();
        }
#pragma unroll 1
        for (; d_n < n_slots; d_n += WAVES) issue_slot(d_n);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // D layout: lane holds column ci = l31, rows co = 8 gq + 4 khalf + e in acc[t][4 gq + e]
    float* out = a.part + (size_t)blockIdx.y * a.Cout * NT * a.Cin;
    const int ci = ci0 + 32 * wj + l31;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
        if (t < ntl) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + 32 * wi + 8 * (e >> 2) + 4 * khalf + (e & 3);
                out[((size_t)co * NT + tap0 + t) * a.Cin + ci] = acc[t][e];
            }
        }
}

bool wgrad2_ok(int Cin, int Cout, int KH, int KW, int pad, int stride, int OW) {
    if (option(OPT_CONV_PATH) != 1) return false;
    if (Cin % 64 != 0 || Cout % 64 != 0 || OW < 2) return false;
    return (KH == 3 && KW == 3 && pad == 1) || (KH == 1 && KW == 1 && pad == 0);
}

// rows per step: the largest R (<= OH) whose double-buffered tiles fit the LDS budget
struct Wgrad2Plan { int R, P, Ppad, XR, XC, gpi, steps, tiles, groups, per_group; size_t lds_bytes; };
Wgrad2Plan wgrad2_plan(int N, int OH, int OW, int Cin, int Cout, int KS, int stride) {
    Wgrad2Plan p;
    p.XC = (OW - 1) * stride + KS;
    auto bytes = [&](int R) {
        const int P = R * OW, Ppad = (P + 3) & ~3, XR = (R - 1) * stride + KS;
        const size_t buf = (size_t)(((Ppad + 3) / 4 + (XR * p.XC + 3) / 4) * 256);
        return 2 * buf * sizeof(float);
    };
    int R = 1;
    // prefer an R that divides OH (no half-empty last group); pixels per step capped at 128
    for (int cand = 1; cand <= OH; ++cand)
        if (OH % cand == 0 && cand * OW <= 128 && bytes(cand) <= 150 * 1024) R = cand;
    p.R = R; p.P = R * OW; p.Ppad = (p.P + 3) & ~3; p.XR = (R - 1) * stride + KS;
    p.lds_bytes = bytes(R);
    p.gpi = (OH + R - 1) / R;
    p.steps = N * p.gpi;
    p.tiles = (Cout / 64) * (Cin / 64);
    int groups = 256 / p.tiles;
    if (groups < 1) groups = 1;
    if (groups > p.steps) groups = p.steps;
    p.per_group = (p.steps + groups - 1) / groups;
    p.groups = (p.steps + p.per_group - 1) / p.per_group;
    return p;
}

template <int KS, int STRIDE>
int launch_wgrad2(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout,
                  int OH, int OW, hipStream_t s) {
    const Wgrad2Plan p = wgrad2_plan(N, OH, OW, Cin, Cout, KS, STRIDE);
    if (p.lds_bytes > 160 * 1024) return fail(DMC_E_INVALID, "conv_wgrad2: row of %d pixels does not fit the LDS", OW);
    Wgrad2Args a;
    a.x = x; a.dy = dy; a.part = p.groups > 1 ? workspace : dw;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.R = p.R; a.P = p.P; a.Ppad = p.Ppad; a.XR = p.XR; a.XC = p.XC;
    a.groups_per_img = p.gpi; a.steps = p.steps; a.per_group = p.per_group; a.tiles_ci = Cin / 64;
    a.inv_xc = (unsigned)((0x100000000ull + p.XC - 1) / p.XC);
    a.inv_ow = (unsigned)((0x100000000ull + OW - 1) / OW);
    {
        const int slots = ((p.Ppad + 3) / 4 + (p.XR * p.XC + 3) / 4 + Wg2<KS>::WAVES - 1) / Wg2<KS>::WAVES, its = p.Ppad / 4;
        a.dma_per_it = (slots + its - 1) / its;
    }
    // more than 64 KB of dynamic LDS needs the attribute; set once per instantiation (thread-safe static)
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&conv_wgrad2_kernel<KS, STRIDE>), 160 * 1024);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "conv_wgrad2: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    conv_wgrad2_kernel<KS, STRIDE><<<dim3(p.tiles, p.groups), Wg2<KS>::WAVES * 64, p.lds_bytes, s>>>(a);
    int rc = check_launch("conv_wgrad2");
    if (rc || p.groups == 1) return rc;
    const long numel = (long)Cout * KS * KS * Cin;
    const long blocks = (numel + 63) / 64;
    conv_wgrad_reduce_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(workspace, dw, p.groups, numel);
    return check_launch("conv_wgrad_reduce");
}

// ------------------------------------------------------------------------------------------
// Weight gradient in bf16x3 arithmetic (option conv_arith = 1; 3x3, Cin % 64 == 0, Cout % 64 == 0).
// The GEMM contracts over PIXELS, and v_mfma_f32_32x32x16_bf16 wants eight consecutive k-values of one row in
// one lane -- eight pixels of one channel, which in NHWC tiles are 256 bytes apart.  So a lane gathers its
// eight pixels with eight ds_read_b32 (conflict-free: the 32 lanes of a half-wave read 32 consecutive
// channels), splits them into bf16 slices and packs pairs with v_perm.  What keeps the VALU work below the
// matrix pipe's time is the choice of the eight pixels: they are consecutive in ONE output row, so the three
// horizontal taps of a tap row read the shifted windows [c, c+8), [c+1, c+9), [c+2, c+10) of one patch row:
// ten values are read and split once (stride 2: 17) and packed three times -- 120 VALU instructions next to
// 18 MFMAs of 32 cycles per 16 pixels.  Rows are padded to a multiple of 8 pixels in LDS (the padding's dy
// rows are transferred from the zero block).
//   workgroup = 64 co x 64 ci block of dw, all nine taps, a run of steps (R output rows of one image each);
//   12 waves = 4 quarters (32 co x 32 ci) x 3 tap rows, three per SIMD; 3 x 16 accumulators per wave.
// Staging (dy tile + input patch per step, LDS-DMA, double-buffered), partials and their fixed-order
// reduction are those of conv_wgrad2_kernel.
// ------------------------------------------------------------------------------------------
struct Wgrad3Args {
    const float* x;        // [N][H][W][Cin]
    const float* dy;       // [N][OH][OW][Cout]
    float* part;           // [groups][Cout][9][Cin]
    int N, H, W, Cin, OH, OW, Cout;
    int R, OWp;            // output rows per step; row length padded to a multiple of 8
    int G, gpr;            // 8-pixel groups per step (R * OWp / 8), per row
    int dyrows;            // pixel rows of the dy tile: 16 * ceil(G / 2)
    int XR, XC;            // patch rows / columns: (R - 1) s + 3, (OWp - 1) s + 3
    int groups_per_img, steps, per_group, tiles_ci;
    int dma_per_it;
    unsigned inv_xc, inv_owp;
};

template <int STRIDE>
__global__ __launch_bounds__(768) void conv_wgrad3_kernel(Wgrad3Args a) {
    constexpr int WAVES = 12, NRAW = 7 * STRIDE + 3;               // patch values per lane and k-block
    extern __shared__ __attribute__((aligned(1024))) float lds4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5;
    const int quarter = wave & 3, trow = wave >> 2;                // tap row: taps 3 trow .. 3 trow + 2
    const int wi = quarter & 1, wj = quarter >> 1;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int x_rows = a.XR * a.XC;
    const int dy_dma = a.dyrows >> 2, x_dma = (x_rows + 3) >> 2;    // 1 KB transfers (4 pixel rows each)
    const int buf_floats = (dy_dma + x_dma) * 256;
    const unsigned lds0 = lds_addr_of(lds4);
    const char* zeros = reinterpret_cast<const char*>(g_zeros);

    const int s_begin = blockIdx.y * a.per_group;
    const int s_end = s_begin + a.per_group < a.steps ? s_begin + a.per_group : a.steps;
    const int sub = lane >> 4, q16 = lane & 15;

    const int n_slots = dy_dma + x_dma;
    const char* dyb = nullptr;
    const char* xb = nullptr;
    int rows_left = 0, iy0 = 0;
    unsigned ibase = 0;
    auto set_step = [&](int step, int buf) {
        const int n = step / a.groups_per_img, oy0 = (step - n * a.groups_per_img) * a.R;
        ibase = lds0 + (unsigned)buf * buf_floats * 4;
        dyb = reinterpret_cast<const char*>(a.dy + (((long)n * a.OH + oy0) * a.OW) * a.Cout + co0 + 4 * q16);
        rows_left = a.OH - oy0 < a.R ? a.OH - oy0 : a.R;
        iy0 = oy0 * STRIDE - 1;
        xb = reinterpret_cast<const char*>(a.x + ((long)n * a.H * a.W) * a.Cin + ci0 + 4 * q16);
    };
    auto issue_slot = [&](int d) {
        if (d < dy_dma) {                                           // wave-uniform
            const int p = 4 * d + sub;
            const int r = (int)(((unsigned long long)p * a.inv_owp) >> 32), c = p - r * a.OWp;
            const bool ok = r < rows_left && c < a.OW;
            dma16_v(ok ? dyb + ((long)r * a.OW + c) * a.Cout * 4 : zeros, ibase + d * 1024);
        } else {
            const int pp = 4 * (d - dy_dma) + sub;
            const int pr = (int)(((unsigned long long)pp * a.inv_xc) >> 32), pc = pp - pr * a.XC;
            const int iy = iy0 + pr, ix = pc - 1;
            const bool ok = pp < x_rows && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            dma16_v(ok ? xb + ((long)iy * a.W + ix) * a.Cin * 4 : zeros, ibase + d * 1024);
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (s_begin < s_end) {
        set_step(s_begin, 0);
#pragma unroll 1
        for (int d = wave; d < n_slots; d += WAVES) issue_slot(d);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int n_kb = (a.G + 1) >> 1;                                // k-blocks (pairs of 8-pixel groups) per step
#pragma unroll 1
    for (int step = s_begin; step < s_end; ++step) {
        const int buf = (step - s_begin) & 1;
        const bool more = step + 1 < s_end;
        if (more) set_step(step + 1, buf ^ 1);
        int d_n = more ? wave : n_slots;
        const float* D = lds4 + buf * buf_floats + 32 * wi + l31;
        const float* X = lds4 + buf * buf_floats + dy_dma * 256 + trow * a.XC * 64 + 32 * wj + l31;
        // this lane's group of the pair: g = 2 kb + khalf -> (row r, first column 8 cg)
        int g = khalf, r = 0, cg = khalf;
        if (cg >= a.gpr) { cg -= a.gpr; ++r; }
        struct Raw { float d[8], x[NRAW]; };
        auto load_raw = [&](Raw& w) {
            const float* dp = D + (g < 2 * n_kb ? g : 0) * 512;     // g < dyrows / 8
            const float* xp = X + (g < a.G ? (r * STRIDE * a.XC + cg * 8 * STRIDE) * 64 : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) w.d[j] = dp[j * 64];
#pragma unroll
            for (int j = 0; j < NRAW; ++j) w.x[j] = xp[j * 64];
            g += 2; cg += 2;
            if (cg >= a.gpr) { cg -= a.gpr; ++r; }
            if (cg >= a.gpr) { cg -= a.gpr; ++r; }
        };
        auto process = [&](const Raw& w) {
            unsigned a0[8], a1[8], a2[8], b0[NRAW], b1[NRAW], b2[NRAW];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a0[j] = __float_as_uint(w.d[j]);
                const float r1 = w.d[j] - __uint_as_float(a0[j] & 0xffff0000u);
                a1[j] = __float_as_uint(r1);
                a2[j] = __float_as_uint(r1 - __uint_as_float(a1[j] & 0xffff0000u));
            }
#pragma unroll
            for (int j = 0; j < NRAW; ++j) {
                b0[j] = __float_as_uint(w.x[j]);
                const float r1 = w.x[j] - __uint_as_float(b0[j] & 0xffff0000u);
                b1[j] = __float_as_uint(r1);
                b2[j] = __float_as_uint(r1 - __uint_as_float(b1[j] & 0xffff0000u));
            }
            u32x4 A0, A1, A2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                A0[e] = __builtin_amdgcn_perm(a0[2 * e + 1], a0[2 * e], 0x07060302u);
                A1[e] = __builtin_amdgcn_perm(a1[2 * e + 1], a1[2 * e], 0x07060302u);
                A2[e] = __builtin_amdgcn_perm(a2[2 * e + 1], a2[2 * e], 0x07060302u);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {                           // horizontal tap t: pixel j reads patch value j s + t
                u32x4 B0, B1, B2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    B0[e] = __builtin_amdgcn_perm(b0[(2 * e + 1) * STRIDE + t], b0[2 * e * STRIDE + t], 0x07060302u);
                    B1[e] = __builtin_amdgcn_perm(b1[(2 * e + 1) * STRIDE + t], b1[2 * e * STRIDE + t], 0x07060302u);
                    B2[e] = __builtin_amdgcn_perm(b2[(2 * e + 1) * STRIDE + t], b2[2 * e * STRIDE + t], 0x07060302u);
                }
                acc[t] = mfma_bf16(A0, B2, acc[t]);
                acc[t] = mfma_bf16(A2, B0, acc[t]);
                acc[t] = mfma_bf16(A1, B1, acc[t]);
                acc[t] = mfma_bf16(A0, B1, acc[t]);
                acc[t] = mfma_bf16(A1, B0, acc[t]);
                acc[t] = mfma_bf16(A0, B0, acc[t]);
            }
        };
        auto issue_some = [&]() {
#pragma unroll 1
            for (int i = 0; i < a.dma_per_it && d_n < n_slots; ++i, d_n += WAVES) issue_slot(d_n);
        };
        Raw w0, w1;
        load_raw(w0);
#pragma unroll 1
        for (int kb = 0; kb < n_kb; kb += 2) {
            load_raw(w1);                                           // past the end: valid LDS, never multiplied
            process(w0);
            issue_some();
            load_raw(w0);
            if (kb + 1 < n_kb) process(w1);
            issue_some();
        }
#pragma unroll 1
        for (; d_n < n_slots; d_n += WAVES) issue_slot(d_n);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float* out = a.part + (size_t)blockIdx.y * a.Cout * 9 * a.Cin;
    const int ci = ci0 + 32 * wj + l31;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + 32 * wi + 8 * (e >> 2) + 4 * khalf + (e & 3);
            out[((size_t)co * 9 + 3 * trow + t) * a.Cin + ci] = acc[t][e];
        }
}

bool wgrad3_ok(int Cin, int Cout, int KH, int KW, int pad, int stride) {
    if (option(OPT_CONV_PATH) != 1 || option(OPT_CONV_ARITH) != 1) return false;
    return Cin % 64 == 0 && Cout % 64 == 0 && KH == 3 && KW == 3 && pad == 1 && (stride == 1 || stride == 2);
}

struct Wgrad3Plan { int R, OWp, G, dyrows, XR, XC, gpi, steps, tiles, groups, per_group; size_t lds_bytes; };
Wgrad3Plan wgrad3_plan(int N, int OH, int OW, int Cin, int Cout, int stride) {
    Wgrad3Plan p;
    p.OWp = (OW + 7) & ~7;
    p.XC = (p.OWp - 1) * stride + 3;
    auto bytes = [&](int R) {
        const int G = R * p.OWp / 8, dyrows = 16 * ((G + 1) / 2), XR = (R - 1) * stride + 3;
        return (size_t)2 * (dyrows / 4 + (XR * p.XC + 3) / 4) * 1024;
    };
    int R = 1;
    for (int cand = 1; cand <= OH; ++cand)
        if (OH % cand == 0 && cand * p.OWp <= 128 && bytes(cand) <= 150 * 1024) R = cand;
    p.R = R; p.G = R * p.OWp / 8; p.dyrows = 16 * ((p.G + 1) / 2); p.XR = (R - 1) * stride + 3;
    p.lds_bytes = bytes(R);
    p.gpi = (OH + R - 1) / R;
    p.steps = N * p.gpi;
    p.tiles = (Cout / 64) * (Cin / 64);
    int groups = 256 / p.tiles;
    if (groups < 1) groups = 1;
    if (groups > p.steps) groups = p.steps;
    p.per_group = (p.steps + groups - 1) / groups;
    p.groups = (p.steps + p.per_group - 1) / p.per_group;
    return p;
}

template <int STRIDE>
int launch_wgrad3(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout,
                  int OH, int OW, hipStream_t s) {
    const Wgrad3Plan p = wgrad3_plan(N, OH, OW, Cin, Cout, STRIDE);
    if (p.lds_bytes > 160 * 1024) return fail(DMC_E_INVALID, "conv_wgrad3: row of %d pixels does not fit the LDS", OW);
    Wgrad3Args a;
    a.x = x; a.dy = dy; a.part = p.groups > 1 ? workspace : dw;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.R = p.R; a.OWp = p.OWp; a.G = p.G; a.gpr = p.OWp / 8; a.dyrows = p.dyrows; a.XR = p.XR; a.XC = p.XC;
    a.groups_per_img = p.gpi; a.steps = p.steps; a.per_group = p.per_group; a.tiles_ci = Cin / 64;
    a.inv_xc = (unsigned)((0x100000000ull + p.XC - 1) / p.XC);
    a.inv_owp = (unsigned)((0x100000000ull + p.OWp - 1) / p.OWp);
    {
        const int slots = (p.dyrows / 4 + (p.XR * p.XC + 3) / 4 + 11) / 12, its = (p.G + 1) / 2;
        a.dma_per_it = (slots + its - 1) / its;
    }
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&conv_wgrad3_kernel<STRIDE>), 160 * 1024);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "conv_wgrad3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    conv_wgrad3_kernel<STRIDE><<<dim3(p.tiles, p.groups), 768, p.lds_bytes, s>>>(a);
    int rc = check_launch("conv_wgrad3");
    if (rc || p.groups == 1) return rc;
    const long numel = (long)Cout * 9 * Cin;
    const long blocks = (numel + 63) / 64;
    conv_wgrad_reduce_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(workspace, dw, p.groups, numel);
    return check_launch("conv_wgrad_reduce");
}

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

// ---- tile selection ---------------------------------------------------------------------------
// Second-generation kernel (Cin % 32 == 0, Cout % 32 == 0): 128 x 64 tiles while they still give every CU
// several workgroups, 64 x 64 below that (the 14 x 14 and 7 x 7 layers: 736 workgroups of 64 x 64 balance
// over 256 CUs where 184 of 128 x 128 cannot).
bool use_v2(int cin, int cout) { return option(OPT_CONV_PATH) == 1 && cin % 32 == 0 && cout % 32 == 0; }

// configuration of the second-generation kernel: 0 = 128 x 32, 1 = 128 x 64, 2 = 64 x 64, 3 = 256 x 64,
// 4 = 128 x 128, 5 = 64 x 128 (pixels x channels), 4 waves each; 6 = 128 x 128, 7 = 256 x 64, 8 = 128 x 64 with 8 waves
int v2_choice(int cout, long M) {
    if (cout % 64 != 0) return 0;
    const int cfg = option(OPT_CONV_CFG);                  // measurement switch: 0 = automatic
    if ((cfg >= 1 && cfg <= 3) || cfg == 7 || cfg == 8) return cfg;
    if ((cfg == 4 || cfg == 5 || cfg == 6) && cout % 128 == 0) return cfg;
    // the largest tile that still gives the 256 CUs at least ~2.5 workgroups each (measured, N = 120: layer1
    // 256 x 64 with 4 waves; layer2 and the stride-2 convolutions 128 x 128 with 8 waves; layer3 128 x 64 with 8
    // waves; layer4 64 x 64)
    const long need = 640;
    if (cout % 128 == 0 && ((M + 127) / 128) * (cout / 128) >= need) return 6;
    if (((M + 255) / 256) * (cout / 64) >= need) return 3;
    if (((M + 127) / 128) * (cout / 64) >= need) return 8;
    return 2;
}

int block_pixels_v2(int cout, long M) {
    const int c = v2_choice(cout, M);
    return (c == 3 || c == 7) ? 256 : (c == 2 || c == 5) ? 64 : 128;
}

int block_pixels_v1(int cout, long M) {
    if (cout >= 64) return M >= 32768 ? 128 : 64;
    return cout == 32 ? 128 : 256;
}

// Third-generation kernel (bf16x3 arithmetic, option conv_arith = 1): Cin % 64 == 0 (an even number of K-steps), Cout % 64 == 0.
bool use_v3(int cin, int cout) { return option(OPT_CONV_PATH) == 1 && option(OPT_CONV_ARITH) == 1 && cin % 64 == 0 && cout % 64 == 0; }

// wave tiles are wide in channels (TN = 2 .. 4 tiles): the weights arrive split, the activation fragments are split by the
// wave that reads them, so the VALU work per MFMA falls with the number of channels a wave covers
//   0 = 256 x 128 (8 x 1 waves), 1 = 256 x 64 (8 x 1), 2 = 128 x 128 (4 x 2), 3 = 128 x 256 (4 x 2), 4 = 64 x 128 (2 x 2),
//   5 = 128 x 64 (4 x 1), 6 = 128 x 128 (4 x 1: 32 x 128 per wave), 7 = 64 x 128 (2 x 1),
//   8 = 256 x 128 (4 x 1: 64 x 128 per wave), 9 = 128 x 128 (2 x 1),
//   10 = 384 x 128 (4 x 2: 96 x 64 per wave), 11 = 192 x 128 (2 x 2), 12 = 96 x 128 (3 x 1): pixel extents that divide
//   the layer2 / layer3 / layer4 problems of a 120-frame batch into ~one workgroup per CU
int v3_choice(int cout, long M) {
    const int cfg = option(OPT_CONV_CFG);                  // measurement switch: 0 = automatic, 1 + configuration otherwise
    if (cfg >= 1 && cfg <= 13) {
        const int c = cfg - 1;
        if (c != 10 && ((c == 1 || c == 5) || (c == 3 ? cout % 256 == 0 : cout % 128 == 0))) return c;   // 10 (384 x 128) spilled: removed
    }
    const long need = 512;
    if (cout % 128 == 0 && ((M + 255) / 256) * (cout / 128) >= need) return 0;
    if (cout % 128 == 0 && ((M + 127) / 128) * (cout / 128) >= need) return 2;
    if (cout % 128 != 0) return ((M + 255) / 256) * (cout / 64) >= need ? 1 : 5;
    if (cout % 128 == 0 && ((M + 191) / 192) * (cout / 128) >= 200) return 11;   // layer3: 246 workgroups of 192 x 128 (176 TFLOP/s; 167 with 32 x 128 wave tiles)
    return 6;                                              // layer3 / layer4: 32 x 128 per wave (147 -> 167, 156 -> 162 TFLOP/s vs 64 x 128 on 2 x 2 waves)
}

int block_pixels_v3(int cout, long M) {
    const int c = v3_choice(cout, M);
    return (c == 0 || c == 1 || c == 8) ? 256 : (c == 4 || c == 7) ? 64 : c == 10 ? 384 : c == 11 ? 192 : c == 12 ? 96 : 128;
}

int block_pixels(int cin, int cout, long M) {
    if (use_v3(cin, cout)) return block_pixels_v3(cout, M);
    return use_v2(cin, cout) ? block_pixels_v2(cout, M) : block_pixels_v1(cout, M);
}

// the statistics partials are [gridDim.x][Cout][2]: the caller sized them with dmc_conv_nhwc_stat_blocks(); the tile choice is
// taken again at launch time (it reads the conv_cfg / conv_arith options), so the launch refuses a buffer of another height
// instead of writing past it
int stat_rows_ok(const ConvArgs& a, unsigned grid_x) {
    if (!a.stat_part || a.stat_blocks < 0 || (unsigned)a.stat_blocks == grid_x) return DMC_OK;
    return fail(DMC_E_INVALID, "conv_nhwc: statistics partials have %d rows but this launch writes %u (dmc_conv_nhwc_stat_blocks was "
                               "called under other options?)", a.stat_blocks, grid_x);
}

template <int BM, int BN, int WM, int WN>
int launch_cfg3(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds_bytes = 2 * (BM * 128 + 3 * BN * 64);
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&conv3_kernel<BM, BN, WM, WN>), (int)lds_bytes);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "conv3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN);
    if (int rc = stat_rows_ok(a, grid.x)) return rc;
#ifdef DMC_MEASURE
    if constexpr (BM == 128 && BN == 128 && WM == 4 && WN == 2) {       // ablation variants of ONE configuration (measurement only)
        const int abl = DMC_ABL(a.ablate) & 12;
        if (abl) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_kernel<BM, BN, WM, WN, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_kernel<BM, BN, WM, WN, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_kernel<BM, BN, WM, WN, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (abl == 4) conv3_kernel<BM, BN, WM, WN, 4><<<grid, WM * WN * 64, lds_bytes, s>>>(a);
            else if (abl == 8) conv3_kernel<BM, BN, WM, WN, 8><<<grid, WM * WN * 64, lds_bytes, s>>>(a);
            else conv3_kernel<BM, BN, WM, WN, 12><<<grid, WM * WN * 64, lds_bytes, s>>>(a);
            return check_launch("conv3 (ablation)");
        }
    }
#endif
    conv3_kernel<BM, BN, WM, WN><<<grid, WM * WN * 64, lds_bytes, s>>>(a);
    return check_launch("conv3");
}

// a.w = the split weights (conv_split_w_kernel)
int launch_conv3(ConvArgs a, hipStream_t s) {
    if (a.M <= 0) return DMC_OK;
    a.ablate = option(OPT_CONV_ABLATE);
    switch (v3_choice(a.Cout, a.M)) {
        case 0: return launch_cfg3<256, 128, 8, 1>(a, s);
        case 1: return launch_cfg3<256, 64, 8, 1>(a, s);
        case 2: return launch_cfg3<128, 128, 4, 2>(a, s);
        case 3: return launch_cfg3<128, 256, 4, 2>(a, s);
        case 4: return launch_cfg3<64, 128, 2, 2>(a, s);
        case 6: return launch_cfg3<128, 128, 4, 1>(a, s);
        case 7: return launch_cfg3<64, 128, 2, 1>(a, s);
        case 8: return launch_cfg3<256, 128, 4, 1>(a, s);
        case 9: return launch_cfg3<128, 128, 2, 1>(a, s);
        case 11: return launch_cfg3<192, 128, 2, 2>(a, s);
        case 12: return launch_cfg3<96, 128, 3, 1>(a, s);
        default: return launch_cfg3<128, 64, 4, 1>(a, s);
    }
}

int split_weights(const float* w, void* wpack, int Cout, int KK, int Cin, int transposed, hipStream_t s) {
    const long total = (long)Cout * KK * Cin;
    const long blocks = (total + 255) / 256;
    conv_split_w_kernel<<<(int)(blocks > 2048 ? 2048 : blocks), 256, 0, s>>>(w, reinterpret_cast<unsigned short*>(wpack), Cout, KK, Cin, transposed);
    return check_launch("conv_split_w");
}

template <int BM, int BN, int BK, int WM, int WN>
int launch_cfg(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN);
    if (int rc = stat_rows_ok(a, grid.x)) return rc;
    conv_nhwc_kernel<BM, BN, BK, WM, WN><<<grid, 256, 0, s>>>(a);
    return check_launch("conv_nhwc");
}

template <int BM, int BN, int WM, int WN>
int launch_cfg2(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN);
    if (int rc = stat_rows_ok(a, grid.x)) return rc;
    conv2_kernel<BM, BN, WM, WN><<<grid, WM * WN * 64, 0, s>>>(a);
    return check_launch("conv2");
}

int launch_conv(ConvArgs a, hipStream_t s) {
    if (a.M <= 0) return DMC_OK;
    a.ablate = option(OPT_CONV_ABLATE);
    if (use_v2(a.Cin, a.Cout)) {
        switch (v2_choice(a.Cout, a.M)) {
            case 0: return launch_cfg2<128, 32, 4, 1>(a, s);
            case 1: return launch_cfg2<128, 64, 2, 2>(a, s);
            case 3: return launch_cfg2<256, 64, 4, 1>(a, s);
            case 4: return launch_cfg2<128, 128, 2, 2>(a, s);
            case 5: return launch_cfg2<64, 128, 1, 4>(a, s);
            case 6: return launch_cfg2<128, 128, 2, 4>(a, s);
            case 7: return launch_cfg2<256, 64, 4, 2>(a, s);
            case 8: return launch_cfg2<128, 64, 4, 2>(a, s);
            default: return launch_cfg2<64, 64, 2, 2>(a, s);
        }
    }
    const bool k32 = a.Cin % 32 == 0;
    const int bm = block_pixels_v1(a.Cout, a.M);
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
int dmc_conv_nhwc_stat_blocks(int N, int H, int W, int Cin, int Cout, int KH, int stride, int pad) {
    if (csm_supported(N, H, W, Cin, Cout, KH, KH, stride, pad)) return csm_stat_blocks(N, H, W, Cin);
    if (csm_fwd_s2_supported(N, H, W, Cin, Cout, KH, KH, stride, pad)) return csm_fwd_s2_stat_blocks(N, H, W);
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KH) / stride + 1;
    const long M = (long)N * OH * OW;
    const int bm = block_pixels(Cin, Cout, M);
    return (int)((M + bm - 1) / bm);
}

int dmc_conv_nhwc_fwd(const float* x, const float* w, void* wpack, const float* bias, const float* keep, float* y,
                      double* stat_partials, int stat_blocks, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                      int pad, int act, dmc_stream_t stream) {
    if (!x || !y || (!w && !wpack)) return fail(DMC_E_INVALID, "dmc_conv_nhwc_fwd: null pointer");
    if (stat_partials && stat_blocks <= 0) return fail(DMC_E_INVALID, "dmc_conv_nhwc_fwd: stat_blocks must be the row count of stat_partials");
    ConvShape sh = {N, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0};
    if (!shape_supported(sh))
        return fail(DMC_E_INVALID, "dmc_conv_nhwc_fwd: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d s=%d p=%d",
                    N, H, W, Cin, Cout, KH, KW, stride, pad);
    // 16- / 32-channel stride-1 3x3 layers (the discriminator's high-resolution blocks): the small-channel bf16x3 kernel
    if (w && wpack && csm_supported(N, H, W, Cin, Cout, KH, KW, stride, pad))
        return csm_fwd(x, w, wpack, bias, keep, y, stat_partials, stat_blocks, N, H, W, Cin, act, (hipStream_t)stream);
    if (w && wpack && csm_fwd_s2_supported(N, H, W, Cin, Cout, KH, KW, stride, pad))
        return csm_fwd_s2(x, w, wpack, bias, keep, y, stat_partials, stat_blocks, N, H, W, act, (hipStream_t)stream);
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.keep = keep; a.stat_part = stat_partials;
    a.stat_blocks = stat_partials ? stat_blocks : -1;
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
    if (use_v3(Cin, Cout)) {
        if (!wpack) return fail(DMC_E_INVALID, "dmc_conv_nhwc_fwd: the bf16x3 arithmetic needs the weight workspace");
        if (w) {                                            // w == NULL: wpack already holds the slices (dmc_conv_nhwc_split)
            int rc = split_weights(w, wpack, Cout, a.KK, Cin, 0, (hipStream_t)stream);
            if (rc) return rc;
        }
        a.w = reinterpret_cast<const float*>(wpack);
        return launch_conv3(a, (hipStream_t)stream);
    }
    if (!w) return fail(DMC_E_INVALID, "dmc_conv_nhwc_fwd: pre-split weights are for the bf16x3 kernels only");
    return launch_conv(a, (hipStream_t)stream);
}

// 1 if the forward / data gradient of this channel pair run on the bf16x3 kernels (then dmc_conv_nhwc_split applies)
int dmc_conv_nhwc_presplit_supported(int Cin, int Cout) { return use_v3(Cin, Cout) && use_v3(Cout, Cin) ? 1 : 0; }

// split the weights for the forward (wpack_f) and the data gradient (wpack_t, transposed) in ONE launch; each workspace
// has dmc_conv_nhwc_wt_bytes() bytes; dmc_conv_nhwc_fwd / _dgrad called with w == NULL then use theirs as it is
int dmc_conv_nhwc_split(const float* w, void* wpack_f, void* wpack_t, int Cin, int Cout, int KH, int KW, dmc_stream_t stream) {
    if (!w || !wpack_f || !wpack_t) return fail(DMC_E_INVALID, "dmc_conv_nhwc_split: null pointer");
    if (!dmc_conv_nhwc_presplit_supported(Cin, Cout)) return fail(DMC_E_INVALID, "dmc_conv_nhwc_split: not a bf16x3 shape");
    const long total = (long)Cout * KH * KW * Cin;
    const long blocks = (total + 255) / 256;
    conv_split_w2_kernel<<<dim3((unsigned)(blocks > 1024 ? 1024 : blocks), 2), 256, 0, (hipStream_t)stream>>>(
        w, reinterpret_cast<unsigned short*>(wpack_f), reinterpret_cast<unsigned short*>(wpack_t), Cout, KH * KW, Cin);
    return check_launch("conv_split_w2");
}

// packed-weight workspace of the forward and the data gradient: fp32 transposed weights or three bf16 slices
size_t dmc_conv_nhwc_wt_bytes(int Cin, int Cout, int KH, int KW) {
    const size_t plain = (size_t)Cin * Cout * KH * KW * 6;
    size_t small = (Cin == Cout && (Cin == 16 || Cin == 32) && KH == 3 && KW == 3) ? csm_wpack_bytes(Cin) : 0;   // k padded to 32
    if (Cin == 16 && Cout == 32 && KH == 3 && KW == 3) small = csm_fwd_s2_wpack_bytes();
    return plain > small ? plain : small;
}

// dx [N,H,W,Cin] from dy [N,OH,OW,Cout]; wt: workspace of dmc_conv_nhwc_wt_bytes() (the packed weights)
static int conv_dgrad(const float* dy, const float* w, float* wt, const float* addend, float* dx, int N, int H, int W, int Cin,
                      int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream) {
    if (!dy || !wt || !dx) return fail(DMC_E_INVALID, "dmc_conv_nhwc_dgrad: null pointer");
    ConvShape sh = {N, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0};
    if (!shape_supported(sh)) return fail(DMC_E_INVALID, "dmc_conv_nhwc_dgrad: unsupported shape");
    hipStream_t s = (hipStream_t)stream;
    if (w && !addend && csm_supported(N, H, W, Cin, Cout, KH, KW, stride, pad))
        return csm_dgrad(dy, w, wt, dx, N, H, W, Cin, s);
    if (w && !addend && csm_dgrad_s2_supported(N, H, W, Cin, Cout, KH, KW, stride, pad))
        return csm_dgrad_s2(dy, w, wt, dx, N, H, W, s);
    const int KK = KH * KW;
    const long total = (long)Cin * Cout * KK;
    const bool v3 = use_v3(Cout, Cin);                    // the GEMM contracts over Cout and produces Cin channels
    int rc;
    if (!w && !v3) return fail(DMC_E_INVALID, "dmc_conv_nhwc_dgrad: pre-split weights are for the bf16x3 kernels only");
    if (v3) {
        rc = w ? split_weights(w, wt, Cout, KK, Cin, 1, s) : DMC_OK;   // w == NULL: wt already holds the transposed slices
    } else {
        conv_pack_wt_kernel<<<(int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256), 256, 0, s>>>(w, wt, Cout, KK, Cin);
        rc = check_launch("conv_pack_wt");
    }
    if (rc) return rc;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    ConvArgs a;
    a.x = dy; a.w = wt; a.y = dx; a.bias = nullptr; a.keep = nullptr; a.stat_part = nullptr; a.addend = addend;
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
            if ((rc = v3 ? launch_conv3(a, s) : launch_conv(a, s))) return rc;
        }
    return DMC_OK;
}

int dmc_conv_nhwc_dgrad(const float* dy, const float* w, float* wt, float* dx, int N, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream) {
    return conv_dgrad(dy, w, wt, nullptr, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, stream);
}

// dx = data gradient + addend (same shape as dx): the residual branch's gradient of a ResNet block joins the main branch's in
// the epilogue of the block's first convolution instead of a separate elementwise pass.  Stride 1, bf16x3 shapes
// (dmc_conv_nhwc_presplit_supported) only.
int dmc_conv_nhwc_dgrad_add(const float* dy, const float* w, float* wt, const float* addend, float* dx, int N, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream) {
    if (!addend) return fail(DMC_E_INVALID, "dmc_conv_nhwc_dgrad_add: null pointer");
    if (stride != 1 || !use_v3(Cout, Cin)) return fail(DMC_E_INVALID, "dmc_conv_nhwc_dgrad_add: stride 1 and bf16x3 shapes only");
    return conv_dgrad(dy, w, wt, addend, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, stream);
}

// BatchNorm statistics from the partials the forward wrote (nblk = dmc_conv_nhwc_stat_blocks()):
// stats [2*C] = (mean, invstd); running_mean / running_var updated as nn.BatchNorm2d does.
int dmc_conv_nhwc_stats_final(const double* partials, int nblk, int C, long count, float* stats,
                              float* running_mean, float* running_var, float eps, float momentum,
                              dmc_stream_t stream) {
    if (!partials || !stats || !running_mean || !running_var || nblk <= 0 || C <= 0 || count <= 0)
        return fail(DMC_E_INVALID, "dmc_conv_nhwc_stats_final: bad argument");
    conv_stats_final_kernel<<<C, 256, 0, (hipStream_t)stream>>>(partials, nblk, C, count, stats,
                                                                          running_mean, running_var, eps, momentum);
    return check_launch("conv_stats_final");
}

// dw [Cout][KH][KW][Cin] (the memory of a channels_last weight gradient).  workspace:
// dmc_conv_nhwc_wgrad_bytes() bytes (the split-K partials; unused when one slice suffices).
size_t dmc_conv_nhwc_wgrad_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (wgrad_small_ok(Cin, KH, KW, pad)) {
        // sized for whichever kernel the launch may pick: the choice depends on process-wide options (conv_arith, conv_cfg) that can
        // differ between this call and the launch, the group count of the small-channel kernel only on the shape -- take the larger
        int groups = 512;
        if (csm_wgrad_groups(N, H, W, Cin) > groups) groups = csm_wgrad_groups(N, H, W, Cin);
        return (size_t)groups * Cout * 9 * Cin * sizeof(float) + 16;
    }
    if (wgrad3_ok(Cin, Cout, KH, KW, pad, stride)) {
        const Wgrad3Plan p = wgrad3_plan(N, OH, OW, Cin, Cout, stride);
        return (size_t)(p.groups > 1 ? p.groups : 0) * Cout * KH * KW * Cin * sizeof(float) + 16;
    }
    if (wgrad2_ok(Cin, Cout, KH, KW, pad, stride, OW)) {
        const Wgrad2Plan p = wgrad2_plan(N, OH, OW, Cin, Cout, KH, stride);
        return (size_t)(p.groups > 1 ? p.groups : 0) * Cout * KH * KW * Cin * sizeof(float) + 16;
    }
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
    if (csm_supported(N, H, W, Cin, Cout, KH, KW, stride, pad) && option(OPT_CONV_CFG) != 302)   // 302: the fp32-MFMA form (A/B runs)
        return csm_wgrad(x, dy, dw, workspace, N, H, W, Cin, s);
    if (csm_wgrad_s2_supported(N, H, W, Cin, Cout, KH, KW, stride, pad))
        return csm_wgrad_s2(x, dy, dw, workspace, N, H, W, Cin, s);
    if (wgrad_small_ok(Cin, KH, KW, pad)) {
        a.part = nullptr; a.per_slice = 0; a.tiles_ci = a.tiles_co = 0;
        if (Cin == 16 && stride == 1) return Cout % 32 == 0 ? launch_wgrad_small<16, 32, 1>(a, dw, workspace, s) : launch_wgrad_small<16, 16, 1>(a, dw, workspace, s);
        if (Cin == 16) return Cout % 32 == 0 ? launch_wgrad_small<16, 32, 2>(a, dw, workspace, s) : launch_wgrad_small<16, 16, 2>(a, dw, workspace, s);
        if (stride == 1) return Cout % 32 == 0 ? launch_wgrad_small<32, 32, 1>(a, dw, workspace, s) : launch_wgrad_small<32, 16, 1>(a, dw, workspace, s);
        return Cout % 32 == 0 ? launch_wgrad_small<32, 32, 2>(a, dw, workspace, s) : launch_wgrad_small<32, 16, 2>(a, dw, workspace, s);
    }
    if (wgrad3_ok(Cin, Cout, KH, KW, pad, stride))
        return stride == 1 ? launch_wgrad3<1>(x, dy, dw, workspace, N, H, W, Cin, Cout, a.OH, a.OW, s)
                           : launch_wgrad3<2>(x, dy, dw, workspace, N, H, W, Cin, Cout, a.OH, a.OW, s);
    if (wgrad2_ok(Cin, Cout, KH, KW, pad, stride, a.OW)) {
        if (KH == 3) return stride == 1 ? launch_wgrad2<3, 1>(x, dy, dw, workspace, N, H, W, Cin, Cout, a.OH, a.OW, s)
                                        : launch_wgrad2<3, 2>(x, dy, dw, workspace, N, H, W, Cin, Cout, a.OH, a.OW, s);
        return stride == 1 ? launch_wgrad2<1, 1>(x, dy, dw, workspace, N, H, W, Cin, Cout, a.OH, a.OW, s)
                           : launch_wgrad2<1, 2>(x, dy, dw, workspace, N, H, W, Cin, Cout, a.OH, a.OW, s);
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
    const long blocks = (numel + 63) / 64;
    conv_wgrad_reduce_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(workspace, dw, ns, numel);
    return check_launch("conv_wgrad_reduce");
}

}  // extern "C"
