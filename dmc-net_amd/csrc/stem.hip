// Weight gradient of the classifier's first convolution (2 -> 64 channels, 7x7, stride 2, pad 3)
// for gfx950.
//
// Reference behaviour: the conv1 the reference installs for the 2-channel flow input,
// code/dmcnet/model.py:285-294 (nn.Conv2d(2, 64, 7, stride 2, padding 3, bias=False)), whose
// weight gradient PyTorch's autograd computes inside loss.backward() (code/dmcnet/train.py:258).
// With 2 input channels the library's implicit-GEMM weight-gradient kernel ran at ~24 TFLOP/s
// (0.80 ms per step at 120 frames); this is the generator's weight-gradient scheme applied to it:
//
//   dW[co][(ci,ky,kx)] = sum_{n,oy,ox} dy[n][oy][ox][co] * x[n][ci][2 oy + ky - 3][2 ox + kx - 3]
//
// one fp32 MFMA GEMM (v_mfma_f32_16x16x4_f32) with M = 64 output channels (4 tiles; row r of tile
// mt is channel 4 r + mt, so ONE ds_read_b128 of the NHWC gradient yields all four A fragments),
// N = 98 (ci, ky, kx) columns in 7 tiles, K = output pixels, 4 per MFMA.  Workgroup = 7 consumer
// waves (one output row of a 7 x 16 tile each) + 1 producer wave that stages, with LDS-DMA, the
// next tile's gradient rows (7 x 16 px x 64 ch, 1 KB per instruction) and input patch (2 planes x
// 19 rows x 40 columns).  Accumulators live in registers for the whole launch and are reduced in
// a fixed order (waves, then workgroups): deterministic, unlike the library's atomic split-K.
#include "dmc_common.h"

using namespace dmc;

namespace {

constexpr int S_CO = 64, S_CI = 2, S_K = 7, S_NCOL = S_CI * S_K * S_K;     // 98
constexpr int S_NT = (S_NCOL + 15) / 16;                                 // 7 column tiles
constexpr int S_MT = S_CO / 16;                                          // 4 row tiles
constexpr int S_TH = 7, S_TW = 16;                                       // output tile (rows = consumer waves)
constexpr int S_XROWS = 2 * S_TH + 5, S_XPITCH = 40;                     // input patch 19 x 40 (cols 2 ox0 - 4 ..)
constexpr int S_XPLANE = S_XROWS * S_XPITCH;                             // 760
constexpr int S_X = S_CI * S_XPLANE;                                     // 1520 floats
constexpr int S_G = S_TH * S_TW * S_CO;                                  // 7168 floats of gradient
constexpr int S_BUF = S_X + S_G;                                         // 8688 floats = 34,752 B; x2 buffers
constexpr int S_PART = S_MT * S_NT * 256;                                // 7168 floats per workgroup partial
constexpr int S_MAX_GROUPS = 256, S_RED = 16;
static_assert(S_PART <= 2 * S_BUF, "cross-wave reduction reuses the tile buffers");

struct StemArgs {
    const float* x;        // [N, 2, H, W]
    const float* dy;       // [N, OH, OW, 64]
    float* partials;
    int N, H, W, OH, OW, tiles_x, tiles_y;
};

// producer: stage one tile (output rows oy0.., columns ox0..)
__device__ __forceinline__ void stem_stage(const StemArgs& a, float* buf, int tile, int per_frame, int lane) {
    const int n = tile / per_frame, r0 = tile - n * per_frame;
    const int oy0 = (r0 / a.tiles_x) * S_TH, ox0 = (r0 % a.tiles_x) * S_TW;
    const unsigned buf_byte = lds_addr_of(buf);
    // ---- gradient rows: 16 px x 64 ch = 4 KB contiguous per output row ----
    const bool g_interior = oy0 + S_TH <= a.OH && ox0 + S_TW <= a.OW;
#pragma unroll
    for (int row = 0; row < S_TH; ++row) {
        const unsigned long long rbase =
            (unsigned long long)(a.dy + (((size_t)n * a.OH + (oy0 + row)) * a.OW + ox0) * S_CO);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned dst = buf_byte + (unsigned)(S_X + (row * S_TW + q * 4) * S_CO) * 4;
            if (g_interior) {
                lds_dma16(rbase + q * 1024, (unsigned)lane * 16, dst);
            } else {
                const bool ok = oy0 + row < a.OH && ox0 + q * 4 + (lane >> 4) < a.OW;
                if (ok) lds_dma16(rbase + q * 1024, (unsigned)lane * 16, dst);
                else reinterpret_cast<float4*>(buf + S_X + (row * S_TW + q * 4) * S_CO)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // ---- input patch: rows 2 oy0 - 3 .. 2 oy0 + 15, columns 2 ox0 - 4 .. 2 ox0 + 35 ----
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 4;
    const bool x_interior = iy0 >= 0 && iy0 + S_XROWS <= a.H && ix0 >= 0 && ix0 + S_XPITCH <= a.W;
    const size_t HW = (size_t)a.H * a.W;
#pragma unroll
    for (int ci = 0; ci < S_CI; ++ci) {
        const long org = ((long)iy0 * a.W + ix0) * 4;
        const unsigned long long pbase = (unsigned long long)(a.x + ((size_t)n * S_CI + ci) * HW) + (unsigned long long)org;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int L = 64 * h + lane;                              // chunk index: (row, 16-byte chunk)
            const int row = L / 10, ch = L - row * 10;
            const unsigned voff = (unsigned)(row * a.W + ch * 4) * 4;
            const unsigned dst = buf_byte + (unsigned)(ci * S_XPLANE + 256 * h) * 4;
            if (L < S_XROWS * 10) {
                const int iy = iy0 + row, ix = ix0 + ch * 4;
                const bool ok = x_interior || (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W);
                if (ok) lds_dma16(pbase, voff, dst);
                else reinterpret_cast<float4*>(buf + ci * S_XPLANE)[L] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void stem_wgrad_kernel(StemArgs a) {
    __shared__ __attribute__((aligned(16))) float lds2[2 * S_BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15, kq = lane >> 4;
    const int per_frame = a.tiles_x * a.tiles_y;
    const int ntiles = a.N * per_frame;

    mfma_f32x4 acc[S_MT][S_NT];
#pragma unroll
    for (int m = 0; m < S_MT; ++m)
#pragma unroll
        for (int t = 0; t < S_NT; ++t) acc[m][t] = (mfma_f32x4){0.f, 0.f, 0.f, 0.f};

    if (wave == S_TH) {
        // ------------------------------ producer wave ------------------------------
        int it = 0;
        if ((int)blockIdx.x < ntiles) stem_stage(a, lds2, blockIdx.x, per_frame, lane);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
            const int next = tile + (int)gridDim.x;
            if (next < ntiles) stem_stage(a, lds2 + ((it + 1) & 1) * S_BUF, next, per_frame, lane);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        // ------------------------------ consumer waves ------------------------------
        // B fragment of column tile t: x[ci][2 oy + ky - 3][2 ox + kx - 3] for (ci, ky, kx) = column 16 t + j
        int offB[S_NT];
#pragma unroll
        for (int t = 0; t < S_NT; ++t) {
            int nn = 16 * t + j;
            nn = nn < S_NCOL ? nn : S_NCOL - 1;
            const int ci = nn / 49, rem = nn - ci * 49, ky = rem / 7, kx = rem - ky * 7;
            offB[t] = ci * S_XPLANE + (2 * wave + ky) * S_XPITCH + kx + 1 + 2 * kq;
        }
        // A fragments: the four channels 4 j .. 4 j + 3 of pixel kq of the group, one 16-byte read
        const int offA = S_X + (wave * S_TW + kq) * S_CO + 4 * j;
        int it = 0;
        asm volatile("s_barrier" ::: "memory");
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
            const float* lds = lds2 + (it & 1) * S_BUF;
#pragma unroll
            for (int g = 0; g < S_TW / 4; ++g) {
                const float4 av = *reinterpret_cast<const float4*>(lds + offA + g * 4 * S_CO);
                const float am[S_MT] = {av.x, av.y, av.z, av.w};
                float b[S_NT];
#pragma unroll
                for (int t = 0; t < S_NT; ++t) b[t] = lds[offB[t] + g * 8];
#pragma unroll
                for (int m = 0; m < S_MT; ++m)
#pragma unroll
                    for (int t = 0; t < S_NT; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[m], b[t], acc[m][t], 0, 0, 0);
            }
            asm volatile("s_barrier" ::: "memory");        // next tile staged, this buffer may be refilled
        }
    }
    // cross-wave reduction in LDS, fixed order (wave 0 stores, waves 1..6 add in turn)
    float* lds = lds2;
    for (int w = 0; w < S_TH; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < S_MT; ++m)
#pragma unroll
                for (int t = 0; t < S_NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = (m * S_NT + t) * 256 + (kq * 4 + q) * 16 + j;   // C row = kq*4+q, col = j
                        lds[idx] = (w == 0 ? 0.f : lds[idx]) + acc[m][t][q];
                    }
        }
        __syncthreads();
    }
    float* part = a.partials + (size_t)blockIdx.x * S_PART;
    for (int i = threadIdx.x; i < S_PART; i += 512) part[i] = lds[i];
}

// partials [groups][S_PART] -> [S_RED][S_PART] (fixed order), written behind the raw partials
__global__ __launch_bounds__(256) void stem_reduce1_kernel(float* __restrict__ partials, int groups) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int per = (groups + S_RED - 1) / S_RED;
    const int g0 = blockIdx.y * per, g1 = (g0 + per < groups) ? g0 + per : groups;
    float s = 0.f;
    for (int g = g0; g < g1; ++g) s += partials[(size_t)g * S_PART + i];
    partials[(size_t)(groups + blockIdx.y) * S_PART + i] = s;
}

// [S_RED][4 x 7 tiles][16][16] -> dW [64][2][7][7]
__global__ void stem_reduce2_kernel(const float* __restrict__ partials, int groups, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S_CO * S_NCOL) return;
    const int co = i / S_NCOL, nn = i - co * S_NCOL;
    const int m = co & 3, row = co >> 2, t = nn >> 4, col = nn & 15;
    const size_t off = (size_t)(m * S_NT + t) * 256 + row * 16 + col;
    float s = 0.f;
    for (int c = 0; c < S_RED; ++c) s += partials[(size_t)(groups + c) * S_PART + off];
    dw[i] = s;
}


// ------------------------------------------------------------------------------------------
// Forward of the same convolution: y[n][oy][ox][co] = sum_k w[co][k] x[n][ci][2 oy + ky - 3][2 ox + kx - 3],
// k = (ci, ky, kx), 98 values.  GEMM on v_mfma_f32_32x32x2_f32 (exact fp32): rows = the 64 output channels (two
// tiles), columns = 32 consecutive output pixels of a row, 49 k-pairs.  A wave keeps ALL weights as A operands in
// registers for the whole launch (2 x 49 VGPRs: lane (row, half) holds w[32 ct + row][ci = half][ky][kx]) and walks
// 32-pixel tiles; the B operand of k-pair kp is ONE input value per lane, x at this lane's pixel and tap
// (ky, kx) of channel `half`, read straight from global memory (lanes read every second float of an input row: the
// whole 2-channel input is 48 MB and lives in L2 / MALL), the next tile's 49 values are in flight during the 98
// MFMAs of the current one.  No LDS, no barriers.  Output NHWC (a lane holds 4 consecutive channels of its pixel:
// 16-byte stores).  Replaces MIOpen's implicit-GEMM forward (0.25 ms at 120 frames), the last library convolution of
// config 2.
// ------------------------------------------------------------------------------------------
typedef float stem_f32x16 __attribute__((ext_vector_type(16)));

struct StemFwdArgs {
    const float* x;        // [N, 2, H, W]
    const float* w;        // [64, 2, 7, 7] by element strides
    float* y;              // [N, OH, OW, 64]
    int N, H, W, OH, OW, tiles_x;
    long ws_co, ws_ci, ws_ky, ws_kx;
};

__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(StemFwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l31 = lane & 31, half = lane >> 5;
    const int HW = a.H * a.W;

    // k-pair kp = (ky, kx), the two k-values of the pair are the two input channels: lane half = ci, so both halves
    // of a wave read the same tap of their own channel plane (one per-lane base, a wave-uniform tap offset)
    float wr[2][49];
#pragma unroll
    for (int kp = 0; kp < 49; ++kp) {
        const long o = half * a.ws_ci + (kp / 7) * a.ws_ky + (kp % 7) * a.ws_kx;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wr[ct][kp] = a.w[(32 * ct + l31) * a.ws_co + o];
    }

    const long ntiles = (long)a.N * a.OH * a.tiles_x;
    const long stride = (long)gridDim.x * 4;
    auto load_tile = [&](long tile, float (&xv)[49]) {
        const int xt = (int)(tile % a.tiles_x);
        const long r = tile / a.tiles_x;
        const int oy = (int)(r % a.OH);
        const int n = (int)(r / a.OH);
        const int ox = 32 * xt + l31;
        const int iy0 = 2 * oy - 3, ix0 = 2 * ox - 3;
        const float* plane = a.x + ((long)n * 2 + half) * HW;                  // this lane's channel plane
        const bool cols_inside = 64 * xt - 3 >= 0 && 64 * xt + 62 + 3 < a.W;    // wave-uniform: no column of the tile is clipped
        unsigned cm = 0;                                                        // per lane: taps kx whose column exists
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) cm |= (ix0 + kx >= 0 && ix0 + kx < a.W) ? 1u << kx : 0u;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = iy0 + ky;                                            // wave-uniform
            if (iy < 0 || iy >= a.H) {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) xv[ky * 7 + kx] = 0.f;
                continue;
            }
            const int rowoff = iy * a.W + ix0;
            if (cols_inside) {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) xv[ky * 7 + kx] = plane[rowoff + kx];
            } else {
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const bool ok = (cm >> kx) & 1;
                    const float v = plane[ok ? rowoff + kx : iy * a.W];
                    xv[ky * 7 + kx] = ok ? v : 0.f;
                }
            }
        }
    };

    long tile = (long)blockIdx.x * 4 + wave;
    float xc[49], xn[49];
    if (tile < ntiles) load_tile(tile, xc);
    for (; tile < ntiles; tile += stride) {
        const long next = tile + stride;
        if (next < ntiles) load_tile(next, xn);
        stem_f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
#pragma unroll
        for (int kp = 0; kp < 49; ++kp)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[ct][kp], xc[kp], acc[ct], 0, 0, 0);
        const int xt = (int)(tile % a.tiles_x);
        const long r = tile / a.tiles_x;
        const int ox = 32 * xt + l31;
        if (ox < a.OW) {
            float* dst = a.y + (r * a.OW + ox) * S_CO;            // r = n * OH + oy
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(dst + 32 * ct + 8 * g + 4 * half) =
                        make_float4(acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]);
        }
#pragma unroll
        for (int kp = 0; kp < 49; ++kp) xc[kp] = xn[kp];
    }
}

// ------------------------------------------------------------------------------------------
// The same forward in bf16x3 arithmetic (conv_nhwc.hip: fp32 = the exact sum of three bf16 slices; six slice products
// per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulate, error <= one fp32 rounding of the product), on the
// scheme of the I3D stem (stem3d_bf16.hip): the input is first written as three zero-padded bf16 slice volumes
// [3][N][H+6][Wp] of (channel 0, channel 1) dwords, in which the 7 taps x 2 channels (+ 2 zero-weight slots) of a kernel
// row for output pixel ox are 32 contiguous bytes starting at pixel 2 ox; 7 k-blocks x 6 slice products x 2 channel tiles
// = 84 MFMAs of 32 cycles per 32-pixel tile where the fp32 form issues 98 of 64; the split weights [3][7][64][16] (43 KB)
// sit in LDS; persistent 8-wave workgroups.
// ------------------------------------------------------------------------------------------
typedef __bf16 stem_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned stem_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned stem_u32x2 __attribute__((ext_vector_type(2)));
constexpr int SX_WLDS = 3 * 7 * S_CO * 32;                // 43,008 bytes

__device__ __forceinline__ void stem_split3(float v, unsigned& s0, unsigned& s1, unsigned& s2) {
    const unsigned u0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
    s0 = u0 >> 16; s1 = u1 >> 16; s2 = u2 >> 16;
}

// x fp32 [N][2][H][W] -> xs [3][N][Hp][Wp] dwords (slice of channel 0 | slice of channel 1 << 16), zero borders (3 in front)
__global__ __launch_bounds__(256) void stem_prep_x3_kernel(const float* __restrict__ x, unsigned* __restrict__ xs, int N, int H, int W,
                                                           int Hp, int Wp) {
    const long vol = (long)N * Hp * Wp, plane = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < vol; i += (long)gridDim.x * 256) {
        const int xx = (int)(i % Wp);
        const long r = i / Wp;
        const int yy = (int)(r % Hp);
        const int n = (int)(r / Hp);
        const int h = yy - 3, w = xx - 3;
        unsigned a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
        if (h >= 0 && h < H && w >= 0 && w < W) {
            const long o = (long)n * 2 * plane + (long)h * W + w;
            stem_split3(x[o], a0, a1, a2);
            stem_split3(x[o + plane], b0, b1, b2);
        }
        xs[i] = a0 | (b0 << 16); xs[vol + i] = a1 | (b1 << 16); xs[2 * vol + i] = a2 | (b2 << 16);
    }
}

// w [64][2][7][7] by element strides -> wp3 [3][7 ky][64 co][16] bf16: slot 2 kx + ci, slots 14, 15 zero
__global__ __launch_bounds__(256) void stem_pack_w3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, long s_co, long s_ci,
                                                           long s_ky, long s_kx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 7 * S_CO * 16) return;
    const int j = i & 15, co = (i >> 4) & 63, ky = i >> 10;
    unsigned s0 = 0, s1 = 0, s2 = 0;
    if (j < 14) stem_split3(w[co * s_co + (j & 1) * s_ci + ky * s_ky + (j >> 1) * s_kx], s0, s1, s2);
    wp[i] = (unsigned short)s0; wp[7 * S_CO * 16 + i] = (unsigned short)s1; wp[2 * 7 * S_CO * 16 + i] = (unsigned short)s2;
}

struct StemX3Args {
    const unsigned* xs;    // [3][N][Hp][Wp]
    const unsigned short* wp;   // [3][7][64][16]
    float* y;              // [N][OH][OW][64]
    int N, OH, OW, Hp, Wp, tiles_x;
    long vol;              // N * Hp * Wp
    double* stat_part;     // nullable: per-workgroup (sum y, sum y^2) per channel, [64][BN_MAX_SPLIT][2]
};

// Epilogue through LDS (per-wave [32 pixels][32 channels] tile, pitch 36 floats, once per 32-channel half): the accumulator
// layout gives every lane 32 channels of ONE pixel -- stored directly that is eight 32-byte pieces per 256-byte pixel row
// and instruction; re-read as [8 pixels][8 lanes x float4] each instruction writes whole 128-byte lines, and a lane sees
// EIGHT fixed channels of every pixel it stores, so bn1's batch statistics (a.stat_part) cost 16 accumulators instead of
// 64.  16 waves per workgroup (four per SIMD: the loads of a tile are issued up front, other waves' MFMAs cover them).
constexpr int SX_THREADS = 1024, SX_WAVES = SX_THREADS / 64;
constexpr int SX_EP_PITCH = 36;
constexpr int SX_EP_BYTES = 32 * SX_EP_PITCH * 4;         // 4,608 per wave
constexpr int SX_LDS = SX_WLDS + SX_WAVES * SX_EP_BYTES;  // 116,736

__global__ __launch_bounds__(SX_THREADS) void stem_fwd_x3_kernel(StemX3Args a) {
    extern __shared__ __attribute__((aligned(16))) char wlds3[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < SX_WLDS / 16; i += SX_THREADS) reinterpret_cast<stem_u32x4*>(wlds3)[i] = reinterpret_cast<const stem_u32x4*>(a.wp)[i];
    __syncthreads();
    const long ntiles = (long)a.N * a.OH * a.tiles_x;
    const long stride = (long)gridDim.x * SX_WAVES;
    const int woff = l31 * 32 + half * 16;
    float* ep = reinterpret_cast<float*>(wlds3 + SX_WLDS + wave * SX_EP_BYTES);
    const int q8 = lane >> 3, c4 = (lane & 7) * 4;
    // Statistics: per lane, sums of (v - K) and (v - K)^2 in fp32 around a PIVOT K = the lane's first value of each channel
    // (|v - K| is of the order of the standard deviation, so E[x^2] - E[x]^2 does not cancel when |mean| >> std as plain
    // fp32 sums of v and v^2 would); the pivot is removed in fp64 before the lanes / waves are combined.
    float ssum[2][4], ssq[2][4], piv[2][4];                 // channels 32 ct + c4 .. + 3, pixels = q8 (mod 8)
    int cnt = 0;                                            // values per channel this lane has added
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) ssum[ct][j] = ssq[ct][j] = piv[ct][j] = 0.f;
    bool first_tile = true;
    for (long tile = (long)blockIdx.x * SX_WAVES + wave; tile < ntiles; tile += stride) {
        const int xt = (int)(tile % a.tiles_x);
        const long r = tile / a.tiles_x;                    // n * OH + oy
        const int oy = (int)(r % a.OH);
        const int n = (int)(r / a.OH);
        const int ox = 32 * xt + l31;
        const int oxc = ox < a.OW ? ox : a.OW - 1;
        const unsigned* base = a.xs + ((long)n * a.Hp + 2 * oy) * a.Wp + 2 * oxc + 4 * half;
        stem_f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
        stem_u32x4 xb[7][3];
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                const unsigned* p = base + sl * a.vol + (long)ky * a.Wp;
                const stem_u32x2 lo = *reinterpret_cast<const stem_u32x2*>(p), hi = *reinterpret_cast<const stem_u32x2*>(p + 2);
                xb[ky][sl] = stem_u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                stem_u32x4 wf[3];
#pragma unroll
                for (int sl = 0; sl < 3; ++sl)
                    wf[sl] = *reinterpret_cast<const stem_u32x4*>(wlds3 + ((sl * 7 + ky) * S_CO + 32 * ct) * 32 + woff);
                auto mm = [&](int i, int j) {
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(stem_bf16x8, wf[i]), __builtin_bit_cast(stem_bf16x8, xb[ky][j]),
                                                                      acc[ct], 0, 0, 0);
                };
                mm(0, 2); mm(2, 0); mm(1, 1); mm(0, 1); mm(1, 0); mm(0, 0);      // small terms first
            }
        float* dst = a.y + (r * a.OW + 32 * xt) * S_CO + c4;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            // accumulators of this channel half -> the wave's LDS tile [pixel l31][channel 8 g + 4 half ..]
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(ep + l31 * SX_EP_PITCH + 8 * g + 4 * half) =
                    make_float4(acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]);
            __builtin_amdgcn_wave_barrier();                 // same wave: LDS instructions complete in order
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = 8 * i + q8;
                const float4 v = *reinterpret_cast<const float4*>(ep + px * SX_EP_PITCH + c4);
                if (first_tile && i == 0) {                  // (wave-uniform) any finite pivot is exact; this one is close to the mean
                    piv[ct][0] = v.x; piv[ct][1] = v.y; piv[ct][2] = v.z; piv[ct][3] = v.w;
                }
                if (32 * xt + px < a.OW) {
                    *reinterpret_cast<float4*>(dst + (long)px * S_CO + 32 * ct) = v;
                    if (a.stat_part) {
                        const float d0 = v.x - piv[ct][0], d1 = v.y - piv[ct][1], d2 = v.z - piv[ct][2], d3 = v.w - piv[ct][3];
                        ssum[ct][0] += d0; ssum[ct][1] += d1; ssum[ct][2] += d2; ssum[ct][3] += d3;
                        ssq[ct][0] = fmaf(d0, d0, ssq[ct][0]); ssq[ct][1] = fmaf(d1, d1, ssq[ct][1]);
                        ssq[ct][2] = fmaf(d2, d2, ssq[ct][2]); ssq[ct][3] = fmaf(d3, d3, ssq[ct][3]);
                        if (ct == 0) ++cnt;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        first_tile = false;
    }
    if (!a.stat_part) return;
    // pivot removed in fp64 (sum v = S1 + n K, sum v^2 = S2 + 2 K S1 + n K^2), then over the eight pixel groups of the wave
    // and over the waves, all in fp64 and in a fixed order
    __syncthreads();                                          // every wave is done with the weights in LDS
    double* red = reinterpret_cast<double*>(wlds3);           // [wave][which][channel]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double k = (double)piv[ct][j], s1 = (double)ssum[ct][j], nn = (double)cnt;
            double t1 = s1 + nn * k, t2 = (double)ssq[ct][j] + 2.0 * k * s1 + nn * k * k;
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) {
                t1 += __shfl_xor(t1, m, 64);
                t2 += __shfl_xor(t2, m, 64);
            }
            if (lane < 8) {
                red[(wave * 2 + 0) * S_CO + 32 * ct + c4 + j] = t1;
                red[(wave * 2 + 1) * S_CO + 32 * ct + c4 + j] = t2;
            }
        }
    __syncthreads();
    if (tid < 2 * S_CO) {
        const int c = tid & (S_CO - 1), which = tid >> 6;
        double t = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < SX_WAVES; ++w8) t += red[(w8 * 2 + which) * S_CO + c];
        a.stat_part[((size_t)c * BN_MAX_SPLIT + blockIdx.x) * 2 + which] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Data gradient of conv1 (the gradient of the 2-channel cue; needed when the classifier's loss reaches the generator: the
// GAN variant, /root/reference/code/dmcnet_GAN/model.py:557-561), bf16x3 arithmetic.  Per INPUT row (n, h):
//     Q[ow][(kx, c)] = sum over the ky whose stride-2 window reaches row h (3 or 4 of them), and over co, of
//                      dy[oh][ow][co] * w[co][c][ky][kx]                       (oh = (h + 3 - ky) / 2)
//     dx[c][h][w]    = sum over kx = w + 1 (mod 2) of Q[(w + 3 - kx) / 2][(kx, c)]
// The first line is a GEMM with M = the OW output pixels of the row, N = 14 (+ 2 zero) columns, K = 64 channels x 3..4
// window rows on v_mfma_f32_16x16x32_bf16 (A fragments: eight consecutive channels of a dy pixel, split into three bf16
// slices in registers; B fragments: 16 contiguous bytes of the pre-split weights [slice][ky][(kx, c)][co]); the second
// line is a 1-D fold of the row through LDS.  One wave per input row: nothing is scattered, the [pixels][98] column
// matrix of the GEMM + col2im formulation (590 MB at 120 frames) never exists, deterministic.  (Scheme of
// stem3d_dgrad_kernel, stem3d_bf16.hip.)
// ------------------------------------------------------------------------------------------
typedef float stem_f32x4 __attribute__((ext_vector_type(4)));
constexpr int SD_WAVES = 4, SD_MT = 8;                      // up to 128 output pixels per row

// w [64][2][7][7] by element strides -> wq [3 slices][7 ky][16 j = 2 kx + c][64 co] bf16; j = 14, 15 zero
__global__ __launch_bounds__(256) void stem_pack_wq3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wq, long s_co, long s_ci,
                                                            long s_ky, long s_kx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 7 * 16 * S_CO) return;
    const int co = i & 63, j = (i >> 6) & 15, ky = i >> 10;
    unsigned s0 = 0, s1 = 0, s2 = 0;
    if (j < 14) stem_split3(w[co * s_co + (j & 1) * s_ci + ky * s_ky + (j >> 1) * s_kx], s0, s1, s2);
    wq[i] = (unsigned short)s0; wq[7 * 16 * S_CO + i] = (unsigned short)s1; wq[2 * 7 * 16 * S_CO + i] = (unsigned short)s2;
}

struct StemDgArgs {
    const float* dy;            // [N][OH][OW][64]
    const unsigned short* wq;   // [3][7][16][64]
    float* dx;                  // [N][2][H][W]
    int N, H, W, OH, OW;
};

__device__ __forceinline__ void stem_split8(const float4 lo, const float4 hi, stem_u32x4 (&sl)[3]) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned u[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        u[0][e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u[0][e] & 0xffff0000u);
        u[1][e] = __float_as_uint(r1);
        u[2][e] = __float_as_uint(r1 - __uint_as_float(u[1][e] & 0xffff0000u));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) sl[k][e] = __builtin_amdgcn_perm(u[k][2 * e + 1], u[k][2 * e], 0x07060302u);
}

template <int MTC>                                          // 16-pixel row tiles when known at compile time; 0 = runtime
__global__ __launch_bounds__(SD_WAVES * 64) void stem_dgrad_x3_kernel(StemDgArgs a) {
    __shared__ float qlds[SD_WAVES][SD_MT * 16 * 16];       // Q tile of each wave: [ow][16]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, kg = lane >> 4;
    float* Q = qlds[wave];
    const long nrows = (long)a.N * a.H;
    const int mtiles = MTC ? MTC : (a.OW + 15) / 16;
    const long plane = (long)a.H * a.W;
    for (long row = (long)blockIdx.x * SD_WAVES + wave; row < nrows; row += (long)gridDim.x * SD_WAVES) {
        const int h = (int)(row % a.H);
        const int n = (int)(row / a.H);
        stem_f32x4 acc[SD_MT];
#pragma unroll
        for (int m = 0; m < SD_MT; ++m) acc[m] = stem_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ky = (h + 1) & 1; ky < S_K; ky += 2) {
            const int oh2 = h + 3 - ky;
            if (oh2 < 0 || (oh2 >> 1) >= a.OH) continue;
            const float* drow = a.dy + ((long)n * a.OH + (oh2 >> 1)) * a.OW * S_CO;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                stem_u32x4 bf[3];
#pragma unroll
                for (int sl = 0; sl < 3; ++sl)
                    bf[sl] = *reinterpret_cast<const stem_u32x4*>(a.wq + ((sl * 7 + ky) * 16 + l15) * S_CO + 32 * kb + 8 * kg);
                float4 lo[SD_MT], hi[SD_MT];
#pragma unroll
                for (int m = 0; m < SD_MT; ++m) {
                    if (MTC ? m >= MTC : m >= mtiles) continue;
                    int ow = 16 * m + l15;
                    ow = ow < a.OW ? ow : a.OW - 1;          // clipped rows: valid memory, their Q rows are never read
                    const float4* p = reinterpret_cast<const float4*>(drow + (long)ow * S_CO + 32 * kb + 8 * kg);
                    lo[m] = p[0]; hi[m] = p[1];
                }
#pragma unroll
                for (int m = 0; m < SD_MT; ++m) {
                    if (MTC ? m >= MTC : m >= mtiles) continue;
                    stem_u32x4 af[3];
                    stem_split8(lo[m], hi[m], af);
                    auto mm = [&](int i, int j) {
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(stem_bf16x8, af[i]), __builtin_bit_cast(stem_bf16x8, bf[j]),
                                                                         acc[m], 0, 0, 0);
                    };
                    mm(0, 2); mm(2, 0); mm(1, 1); mm(0, 1); mm(1, 0); mm(0, 0);      // small terms first
                }
            }
        }
        // C layout of 16x16: lane (column l15 = j, rows 4 kg + q = ow within the tile)
#pragma unroll
        for (int m = 0; m < SD_MT; ++m) {
            if (m >= mtiles) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) Q[(16 * m + 4 * kg + q) * 16 + l15] = acc[m][q];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // fold: dx[c][w] = sum over kx = w + 1 (mod 2) of Q[(w + 3 - kx) / 2][2 kx + c]
        float* dst = a.dx + (long)n * 2 * plane + (long)h * a.W;
        for (int i = lane; i < 2 * a.W; i += 64) {
            const int c = i / a.W, w = i - c * a.W;
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kx = ((w + 1) & 1) + 2 * u;
                const int ow2 = w + 3 - kx;
                if (kx < S_K && ow2 >= 0 && (ow2 >> 1) < a.OW) sum += Q[(ow2 >> 1) * 16 + 2 * kx + c];
            }
            dst[c * plane + w] = sum;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

int stem_groups(int N, int H, int W) {
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const long tiles = (long)N * ((OH + S_TH - 1) / S_TH) * ((OW + S_TW - 1) / S_TW);
    return (int)(tiles < S_MAX_GROUPS ? tiles : S_MAX_GROUPS);
}

}  // namespace

extern "C" {

int dmc_stem_wgrad_supported(int H, int W) { return H > 0 && W > 0 && W % 4 == 0 && ((long)H * W) % 4 == 0; }

size_t dmc_stem_wgrad_partials_bytes(int N, int H, int W) {
    return (size_t)(stem_groups(N, H, W) + S_RED) * S_PART * sizeof(float);
}

int dmc_stem_wgrad(const float* x, const float* dy, float* dw, float* partials, int N, int H, int W,
                   dmc_stream_t stream) {
    if (!x || !dy || !dw || !partials) return fail(DMC_E_INVALID, "dmc_stem_wgrad: null pointer");
    if (N <= 0 || !dmc_stem_wgrad_supported(H, W))
        return fail(DMC_E_INVALID, "dmc_stem_wgrad: unsupported shape N=%d H=%d W=%d (W %% 4 == 0 required)", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    StemArgs a;
    a.x = x; a.dy = dy; a.partials = partials;
    a.N = N; a.H = H; a.W = W; a.OH = (H + 1) / 2; a.OW = (W + 1) / 2;
    a.tiles_x = (a.OW + S_TW - 1) / S_TW;
    a.tiles_y = (a.OH + S_TH - 1) / S_TH;
    const int groups = stem_groups(N, H, W);
    stem_wgrad_kernel<<<groups, 512, 0, s>>>(a);
    int rc = check_launch("stem_wgrad");
    if (rc) return rc;
    stem_reduce1_kernel<<<dim3(S_PART / 256, S_RED), 256, 0, s>>>(partials, groups);
    if ((rc = check_launch("stem_reduce1"))) return rc;
    stem_reduce2_kernel<<<(S_CO * S_NCOL + 255) / 256, 256, 0, s>>>(partials, groups, dw);
    return check_launch("stem_reduce2");
}

// y [N, OH, OW, 64] (NHWC: the memory of a channels_last [N,64,OH,OW] tensor) = conv2d(x [N,2,H,W], w, stride 2,
// padding 3); w [64,2,7,7] addressed by its element strides (contiguous or channels_last)
int dmc_stem_fwd(const float* x, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, float* y, int N, int H, int W,
                 dmc_stream_t stream) {
    if (!x || !w || !y) return fail(DMC_E_INVALID, "dmc_stem_fwd: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail(DMC_E_INVALID, "dmc_stem_fwd: bad shape");
    StemFwdArgs a;
    a.x = x; a.w = w; a.y = y; a.N = N; a.H = H; a.W = W; a.OH = (H + 1) / 2; a.OW = (W + 1) / 2;
    a.tiles_x = (a.OW + 31) / 32;
    a.ws_co = ws_co; a.ws_ci = ws_ci; a.ws_ky = ws_ky; a.ws_kx = ws_kx;
    const long tiles = (long)N * a.OH * a.tiles_x;
    long blocks = (tiles + 3) / 4;
    if (blocks > 512) blocks = 512;                       // two workgroups per CU, every wave walks many tiles
    stem_fwd_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(a);
    return check_launch("stem_fwd");
}

// bf16x3 form of dmc_stem_fwd (fp32-accurate; see stem_fwd_x3_kernel): workspace of dmc_stem_fwd_x3_workspace_bytes()
size_t dmc_stem_fwd_x3_workspace_bytes(int N, int H, int W) {
    const long Wp = (W + 8 + 3) / 4 * 4;
    return (size_t)3 * N * (H + 6) * Wp * 4 + (size_t)3 * 7 * S_CO * 16 * 2 + 64;
}
int dmc_stem_fwd_x3_stat_blocks(int N, int H, int W) {
    const long tiles = (long)N * ((H + 1) / 2) * (((W + 1) / 2 + 31) / 32);
    const long blocks = (tiles + SX_WAVES - 1) / SX_WAVES;
    return (int)(blocks > 256 ? 256 : blocks);             // one 16-wave workgroup per CU, every wave walks many tiles
}
int dmc_stem_fwd_x3(const float* x, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, void* workspace, float* y, int N,
                    int H, int W, dmc_stream_t stream) {
    return dmc_stem_fwd_x3_stats(x, w, ws_co, ws_ci, ws_ky, ws_kx, workspace, y, nullptr, N, H, W, stream);
}
int dmc_stem_fwd_x3_stats(const float* x, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, void* workspace, float* y,
                          void* stat_scratch, int N, int H, int W, dmc_stream_t stream) {
    if (!x || !w || !workspace || !y) return fail(DMC_E_INVALID, "dmc_stem_fwd_x3: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail(DMC_E_INVALID, "dmc_stem_fwd_x3: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int Hp = H + 6, Wp = (W + 8 + 3) / 4 * 4;
    const long vol = (long)N * Hp * Wp;
    unsigned* xs = (unsigned*)workspace;
    unsigned short* wp = (unsigned short*)((char*)workspace + (size_t)3 * vol * 4);
    stem_prep_x3_kernel<<<(int)((vol + 255) / 256 > 8192 ? 8192 : (vol + 255) / 256), 256, 0, s>>>(x, xs, N, H, W, Hp, Wp);
    int rc = check_launch("stem_prep_x3");
    if (rc) return rc;
    stem_pack_w3_kernel<<<(7 * S_CO * 16 + 255) / 256, 256, 0, s>>>(w, wp, ws_co, ws_ci, ws_ky, ws_kx);
    if ((rc = check_launch("stem_pack_w3"))) return rc;
    StemX3Args a;
    a.xs = xs; a.wp = wp; a.y = y; a.N = N; a.OH = (H + 1) / 2; a.OW = (W + 1) / 2; a.Hp = Hp; a.Wp = Wp;
    a.tiles_x = (a.OW + 31) / 32; a.vol = vol;
    a.stat_part = static_cast<double*>(stat_scratch);
    const long blocks = dmc_stem_fwd_x3_stat_blocks(N, H, W);
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&stem_fwd_x3_kernel), SX_LDS);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "dmc_stem_fwd_x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    stem_fwd_x3_kernel<<<(int)blocks, SX_THREADS, SX_LDS, s>>>(a);
    return check_launch("stem_fwd_x3");
}

// Data gradient of conv1 (see stem_dgrad_x3_kernel): dx [N][2][H][W] fp32 contiguous from dy [N][OH][OW][64] fp32 (the memory
// of a channels_last [N,64,OH,OW] tensor) and w [64,2,7,7] by its element strides; stride 2, padding 3; bf16x3 arithmetic
// (fp32-level error), deterministic.  W <= 256.  workspace: dmc_stem_dgrad_workspace_bytes().
size_t dmc_stem_dgrad_workspace_bytes(void) { return (size_t)3 * 7 * 16 * S_CO * 2 + 64; }
int dmc_stem_dgrad_supported(int H, int W) { return H > 0 && W > 0 && (W + 1) / 2 <= 16 * SD_MT ? 1 : 0; }
int dmc_stem_dgrad(const float* dy, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, void* workspace, float* dx, int N,
                   int H, int W, dmc_stream_t stream) {
    if (!dy || !w || !workspace || !dx) return fail(DMC_E_INVALID, "dmc_stem_dgrad: null pointer");
    if (N <= 0 || !dmc_stem_dgrad_supported(H, W)) return fail(DMC_E_INVALID, "dmc_stem_dgrad: unsupported shape N=%d H=%d W=%d (W <= 256)", N, H, W);
    hipStream_t s = (hipStream_t)stream;
    unsigned short* wq = static_cast<unsigned short*>(workspace);
    stem_pack_wq3_kernel<<<(7 * 16 * S_CO + 255) / 256, 256, 0, s>>>(w, wq, ws_co, ws_ci, ws_ky, ws_kx);
    int rc = check_launch("stem_pack_wq3");
    if (rc) return rc;
    StemDgArgs a;
    a.dy = dy; a.wq = wq; a.dx = dx; a.N = N; a.H = H; a.W = W; a.OH = (H + 1) / 2; a.OW = (W + 1) / 2;
    const long rows = (long)N * H;
    long blocks = (rows + SD_WAVES - 1) / SD_WAVES;
    if (blocks > 4096) blocks = 4096;
    if (a.OW == 112) stem_dgrad_x3_kernel<7><<<(int)blocks, SD_WAVES * 64, 0, s>>>(a);      // 224-wide frames
    else stem_dgrad_x3_kernel<0><<<(int)blocks, SD_WAVES * 64, 0, s>>>(a);
    return check_launch("stem_dgrad_x3");
}

}  // extern "C"
