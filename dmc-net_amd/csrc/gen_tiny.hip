// EstimatorDenseNetTiny forward / backward for gfx950.
//
// Reference behaviour: code/dmcnet/model.py:172-194 (EstimatorDenseNetTiny), :111-119 (conv,
// predict_flow), :341-346 (cat(mv,res), +input_mv).  Nothing here is derived from reference
// source text; the reference has no kernels at all.
//
// This file holds the "layerwise" path: one launch per layer, features kept in a
// [N][28][H][W] buffer in physical (append) channel order.  It handles any H, W and is the
// path the backward pass reads its saved activations from.
#include <stdlib.h>

#include "dmc_common.h"

using namespace dmc;

namespace {

// ------------------------------------------------------------------------------------------
// parameter repack: PyTorch [Cout][Cin_logical][3][3] -> WF | BF | WB (see dmc_common.h)
// ------------------------------------------------------------------------------------------
__global__ void pack_params_kernel(ParamPtrs P, float* __restrict__ pk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PACKED_TOTAL) return;
    if (i < WF_TOTAL) {
        int k = 0;
        while (k < NL - 1 && i >= wf_off(k + 1)) ++k;
        const int cin = cin_of(k), cout = cout_of(k);
        const int r = i - wf_off(k);
        const int co = r % cout, tap = (r / cout) % 9, p = r / (cout * 9);
        pk[i] = P.w[k][(co * cin + logical_of(k, p)) * 9 + tap];
    } else if (i < NPARAM) {
        int k = 0;
        while (k < NL - 1 && i >= bf_off(k + 1)) ++k;
        pk[i] = P.b[k][i - bf_off(k)];
    } else {
        int k = 1;
        while (k < NL - 1 && i >= wb_off(k + 1)) ++k;
        const int cin = cin_of(k), D = cin - NIN;
        const int r = i - wb_off(k);
        const int cd = r % D, tap = (r / D) % 9, cg = r / (D * 9);
        // d/dx of a correlation is a correlation with the flipped kernel: tap -> 8 - tap
        pk[i] = P.w[k][(cg * cin + logical_of(k, cd + NIN)) * 9 + (8 - tap)];
    }
}

__device__ __forceinline__ const float* in_plane(const float* mv, const float* res,
                                                 const float* feat, int n, int p, size_t HW) {
    return p < 2 ? mv + ((size_t)n * 2 + p) * HW
                 : p < NIN ? res + ((size_t)n * 3 + (p - 2)) * HW
                           : feat + ((size_t)n * NFEAT + (p - NIN)) * HW;
}

// ------------------------------------------------------------------------------------------
// One 3x3 layer over a 32x32-pixel tile per workgroup (256 threads).
//
//   lane = (row r of the tile, 4-pixel strip s); it produces 4 pixels x ALL output channels, so
//   every input value fetched from LDS feeds 3*COUT FMAs and the weights are wave-uniform:
//   they arrive through scalar loads and enter the FMAs as SGPR operands.
//   Input planes are staged in chunks of 8 channels, branch-free (clamped addresses, values
//   zeroed by a select), as [34 rows][32] interior + two compact halo-column arrays, so the
//   lane's reads are one conflict-free ds_read_b128 and two conflict-free ds_read_b32 per row.
//   52 KB of LDS per workgroup -> 3 workgroups (12 waves) per CU; another workgroup's FMAs
//   cover this one's staging.
//
// MODE 0: forward hidden layer K   (inputs mv/res/feat[0..), output LeakyReLU(0.1) -> feat)
// MODE 1: forward last layer       (output -> out, optionally + mv)
// MODE 2: data-gradient of layer K (inputs g_K, output accumulated into gbuf[0..D), the
//         channels of y_{K-1} finalised with LeakyReLU'(y_{K-1}))
// ------------------------------------------------------------------------------------------
constexpr int LT = 32;                       // tile edge (pixels)
constexpr int LCH = 8;                       // channels per staged chunk
constexpr int LROWS = LT + 2;
constexpr int L_MAIN = LROWS * LT;           // 1088 floats
constexpr int L_HALO = LROWS * 8;            // per side
constexpr int L_PLANE = L_MAIN + 2 * L_HALO; // 1632 floats per channel
constexpr int L_LDS = LCH * L_PLANE;         // 13056 floats = 52,224 B

struct LayerArgs {
    const float* mv;      // [N,2,H,W]
    const float* res;     // [N,3,H,W]
    const float* feat;    // [N,28,H,W] features y0..y4 (read)
    float* feat_out;      // same buffer (written by MODE 0)
    const float* gout;    // [N,2,H,W]  dL/d(out)            (MODE 2)
    float* gbuf;          // [N,28,H,W] feature gradients     (MODE 2)
    const float* pk;      // packed parameters
    float* out;           // [N,2,H,W]                        (MODE 1)
    int H, W, add_mv;
};

template <int MODE, int K>
__device__ __forceinline__ const float* layer_in_plane(const LayerArgs& a, int n, int c, size_t HW) {
    if (MODE == 2)
        return K == 5 ? a.gout + ((size_t)n * 2 + c) * HW
                      : a.gbuf + ((size_t)n * NFEAT + (yoff(K) - NIN) + c) * HW;
    return in_plane(a.mv, a.res, a.feat, n, c, HW);
}

template <int MODE, int K, bool VEC4>
__global__ __launch_bounds__(256) void gen_layer_kernel(LayerArgs a) {
    constexpr int CIN = MODE == 2 ? cout_of(K) : cin_of(K);
    constexpr int COUT = MODE == 2 ? cin_of(K) - NIN : cout_of(K);
    __shared__ __attribute__((aligned(16))) float lds[L_LDS];
    const int n = blockIdx.z, ty0 = blockIdx.y * LT, tx0 = blockIdx.x * LT;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int tid = threadIdx.x, r = tid >> 3, s = tid & 7;

    float acc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = MODE == 2 ? 0.f : a.pk[bf_off(K) + co];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][j] = bv;
    }
    const float* wbase = a.pk + (MODE == 2 ? wb_off(K) : wf_off(K));

#pragma unroll 1
    for (int c0 = 0; c0 < CIN; c0 += LCH) {
        const int nch = (CIN - c0) < LCH ? (CIN - c0) : LCH;
        if (c0 > 0) __syncthreads();
        // ---- stage nch planes: rows ty0-1..ty0+32, interior cols tx0..tx0+31, halo cols ----
        {
            const int q = tid & 7, xx = tx0 + 4 * q;
            for (int rr = tid >> 3; rr < nch * LROWS; rr += 32) {
                const int c = rr / LROWS, row = rr - c * LROWS;
                const int yy = ty0 - 1 + row;
                const bool rowok = (yy >= 0) && (yy < H);
                const float* src = layer_in_plane<MODE, K>(a, n, c0 + c, HW) + (size_t)(rowok ? yy : 0) * W;
                float4 v;
                if (VEC4) {
                    const bool ok = rowok && xx < W;
                    v = *reinterpret_cast<const float4*>(src + (ok ? xx : 0));
                    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    v.x = (rowok && xx + 0 < W) ? src[xx + 0 < W ? xx + 0 : 0] : 0.f;
                    v.y = (rowok && xx + 1 < W) ? src[xx + 1 < W ? xx + 1 : 0] : 0.f;
                    v.z = (rowok && xx + 2 < W) ? src[xx + 2 < W ? xx + 2 : 0] : 0.f;
                    v.w = (rowok && xx + 3 < W) ? src[xx + 3 < W ? xx + 3 : 0] : 0.f;
                }
                *reinterpret_cast<float4*>(lds + c * L_PLANE + row * LT + 4 * q) = v;
                // halo columns of strip q: left = col 4q-1, right = col 4q+4 (tile-relative)
                const int xl = xx - 1, xr = xx + 4;
                const bool okl = rowok && xl >= 0 && xl < W, okr = rowok && xr < W;
                const float hl = src[okl ? xl : 0], hr = src[okr ? xr : 0];
                lds[c * L_PLANE + L_MAIN + row * 8 + q] = okl ? hl : 0.f;
                lds[c * L_PLANE + L_MAIN + L_HALO + row * 8 + q] = okr ? hr : 0.f;
            }
        }
        __syncthreads();
        // ---- accumulate ----
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const float* pl = lds + c * L_PLANE;
            float xv[3][6];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int row = r + ky;
                const float4 m = *reinterpret_cast<const float4*>(pl + row * LT + 4 * s);
                xv[ky][0] = pl[L_MAIN + row * 8 + s];
                xv[ky][1] = m.x; xv[ky][2] = m.y; xv[ky][3] = m.z; xv[ky][4] = m.w;
                xv[ky][5] = pl[L_MAIN + L_HALO + row * 8 + s];
            }
            const float* wp = wbase + (c0 + c) * 9 * COUT;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const float wv = wp[(ky * 3 + kx) * COUT + co];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[co][j] = fmaf(xv[ky][j + kx], wv, acc[co][j]);
                    }
        }
    }

    // ---- epilogue ----
    const int y = ty0 + r, x0 = tx0 + 4 * s;
    if (y >= H || x0 >= W) return;
    const size_t pix = (size_t)y * W + x0;
    if (MODE == 0) {
        float* dst = a.feat_out + ((size_t)n * NFEAT + (yoff(K) - NIN)) * HW + pix;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[co][j] > 0.f ? acc[co][j] : 0.1f * acc[co][j];
            if (VEC4) {
                *reinterpret_cast<float4*>(dst + co * HW) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x0 + j < W) dst[co * HW + j] = v[j];
            }
        }
    } else if (MODE == 1) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float* dst = a.out + ((size_t)n * 2 + co) * HW + pix;
            const float* m = a.mv + ((size_t)n * 2 + co) * HW + pix;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (VEC4 || x0 + j < W) dst[j] = acc[co][j] + (a.add_mv ? m[j] : 0.f);
        }
    } else {
        constexpr int TOP0 = COUT - cout_of(K - 1 < 0 ? 0 : K - 1);
#pragma unroll
        for (int cd = 0; cd < COUT; ++cd) {
            float* dst = a.gbuf + ((size_t)n * NFEAT + cd) * HW + pix;
            const float* f = a.feat + ((size_t)n * NFEAT + cd) * HW + pix;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (VEC4 || x0 + j < W) {
                    float v = acc[cd][j];
                    if (K != 5) v += dst[j];
                    if (cd >= TOP0) v *= (f[j] > 0.f ? 1.f : 0.1f);
                    dst[j] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, weight path: one fp32-MFMA GEMM over pixels.
//
//   dW[(k,co)][(ci,tap)] = sum_px g_k[co][px] * x[ci][px + tap]
//
// Rows M = the 30 (layer, cout) pairs of all six layers, split in two 16-row tiles
//   tile A = g0(8) g1(8)            tile B = g2(6) g3(4) g4(2) g5(2) + 2 zero rows
// columns N = (physical input channel, tap) = 33*9 = 297 (+ column 297 == 1 for the bias
// gradient), in 19 tiles of 16; K = pixels, 4 per v_mfma_f32_16x16x4_f32.  Tile A only needs
// input channels < 13 (N tiles 0..7) plus the bias tile: 28 MFMAs per 4 pixels, 66 % of them
// useful -- exact fp32 (an fmaf chain per accumulator) at the matrix-core rate, the VALU stays
// free for addressing.  Persistent workgroups stage an 8x32-pixel tile of the 33 input/feature
// planes (halo 1) and of the 30 gradient planes in LDS; the 28 accumulator tiles (112
// registers) live in registers for the whole launch and are reduced across waves (fixed
// order) and across workgroups (second kernel, fixed order): deterministic.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WT_H = 8, WT_W = 32;
constexpr int WX_PITCH = 36, WX_PLANE = 364;           // plane % 32 == 12: conflict-light gathers
constexpr int WG_PLANE = WT_H * WT_W + 2;              // 258: conflict-free A-fragment reads
constexpr int WX_FLOATS = 33 * WX_PLANE, WG_FLOATS = 32 * WG_PLANE;
constexpr int WT_LDS = WX_FLOATS + WG_FLOATS;          // 20268 floats = 81,072 B -> 2 WG / CU
constexpr int NT_A = 9, NT_B = 19, NT_ALL = NT_A + NT_B;   // accumulator tiles per wave
constexpr int WPART = NT_ALL * 256;                    // floats per workgroup partial
constexpr int WGRAD_MAX_GROUPS = 512;
constexpr int BIAS_COL = 297;

struct WgradArgs {
    const float* mv;
    const float* res;
    const float* feat;
    const float* gout;
    const float* gbuf;
    float* partials;
    int N, H, W, tiles_x, tiles_y;
};

// Generic (any W) staging: one bounds-checked element per thread-iteration.
__device__ __forceinline__ void wgrad_stage_tile_generic(const WgradArgs& a, float* lds, int n,
                                                         int ty0, int tx0, size_t HW) {
    constexpr int COLS = WT_W + 2, ROWS = WT_H + 2;
    for (int i = threadIdx.x; i < 33 * ROWS * COLS; i += 256) {
        const int c = i / (ROWS * COLS);
        const int rem = i - c * (ROWS * COLS);
        const int row = rem / COLS, col = rem - row * COLS;
        const int yy = ty0 - 1 + row, xx = tx0 - 1 + col;
        float v = 0.f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
            v = in_plane(a.mv, a.res, a.feat, n, c, HW)[(size_t)yy * a.W + xx];
        lds[c * WX_PLANE + row * WX_PITCH + col] = v;
    }
    float* gl = lds + WX_FLOATS;
    for (int i = threadIdx.x; i < 30 * WT_H * WT_W; i += 256) {
        const int c = i / (WT_H * WT_W);
        const int rem = i - c * (WT_H * WT_W);
        const int row = rem / WT_W, col = rem - row * WT_W;
        const int yy = ty0 + row, xx = tx0 + col;
        float v = 0.f;
        if (yy < a.H && xx < a.W) {
            const float* src = c < NFEAT ? a.gbuf + ((size_t)n * NFEAT + c) * HW
                                         : a.gout + ((size_t)n * 2 + (c - NFEAT)) * HW;
            v = src[(size_t)yy * a.W + xx];
        }
        gl[c * WG_PLANE + rem] = v;
    }
}

// W % 4 == 0: 16-byte global loads, 8 lanes per 32-pixel row, all loads of a thread independent.
//   x planes: rows ty0-1 .. ty0+8, cols tx0-1 .. tx0+32 (LDS col 0 / 33 = halo), zero outside
//   gradient planes 0..27 = g0..g4 (gbuf), 28..29 = g5 (grad_out)
__device__ __forceinline__ void wgrad_stage_tile_vec(const WgradArgs& a, float* lds, int n, int ty0,
                                                     int tx0, size_t HW) {
    const int q = threadIdx.x & 7, xx = tx0 + 4 * q;
    const bool colok = xx < a.W;                       // W % 4 == 0 -> the whole quad is in or out
#pragma unroll 2
    for (int rr = threadIdx.x >> 3; rr < 33 * (WT_H + 2); rr += 32) {
        const int c = rr / (WT_H + 2), row = rr - c * (WT_H + 2);
        const int yy = ty0 - 1 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (colok && yy >= 0 && yy < a.H)
            v = *reinterpret_cast<const float4*>(in_plane(a.mv, a.res, a.feat, n, c, HW) + (size_t)yy * a.W + xx);
        float* d = lds + c * WX_PLANE + row * WX_PITCH + 1 + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = threadIdx.x; i < 33 * (WT_H + 2) * 2; i += 256) {
        const int rr = i >> 1, side = i & 1;
        const int c = rr / (WT_H + 2), row = rr - c * (WT_H + 2);
        const int yy = ty0 - 1 + row, xh = side ? tx0 + WT_W : tx0 - 1;
        float v = 0.f;
        if (yy >= 0 && yy < a.H && xh >= 0 && xh < a.W)
            v = in_plane(a.mv, a.res, a.feat, n, c, HW)[(size_t)yy * a.W + xh];
        lds[c * WX_PLANE + row * WX_PITCH + (side ? WT_W + 1 : 0)] = v;
    }
    float* gl = lds + WX_FLOATS;
#pragma unroll 2
    for (int rr = threadIdx.x >> 3; rr < 30 * WT_H; rr += 32) {
        const int c = rr / WT_H, row = rr - c * WT_H;
        const int yy = ty0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (colok && yy < a.H) {
            const float* src = c < NFEAT ? a.gbuf + ((size_t)n * NFEAT + c) * HW
                                         : a.gout + ((size_t)n * 2 + (c - NFEAT)) * HW;
            v = *reinterpret_cast<const float4*>(src + (size_t)yy * a.W + xx);
        }
        float* d = gl + c * WG_PLANE + row * WT_W + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

template <bool VEC4>
__global__ __launch_bounds__(256, 2) void gen_bwd_weight_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[WT_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;          // column / pixel-in-group of this lane
    float* gl = lds + WX_FLOATS;
    for (int i = threadIdx.x; i < 2 * WG_PLANE; i += 256) gl[30 * WG_PLANE + i] = 0.f;

    // per-lane gather offsets of the B fragments (x[(ci,tap)][pixel kq of the group])
    int offB[NT_B];
#pragma unroll
    for (int t = 0; t < NT_B; ++t) {
        int nn = 16 * t + j;
        nn = nn < 297 ? nn : 296;
        const int ci = nn / 9, tap = nn - ci * 9;
        offB[t] = ci * WX_PLANE + (tap / 3) * WX_PITCH + (tap % 3) + kq;
    }
    const bool ones = (16 * 18 + j) == BIAS_COL;      // bias column lives in N tile 18
    // A fragments: row j of the M tile, pixel kq
    const int offA0 = j * WG_PLANE + kq;              // tile A: planes 0..15
    const int offA1 = (16 + j) * WG_PLANE + kq;       // tile B: planes 16..29 (+2 zero planes)

    f32x4 accA[NT_A], accB[NT_B];
#pragma unroll
    for (int t = 0; t < NT_A; ++t) accA[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT_B; ++t) accB[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const size_t HW = (size_t)a.H * a.W;
    const int per_frame = a.tiles_x * a.tiles_y;
    const int ntiles = a.N * per_frame;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / per_frame, r0 = tile - n * per_frame;
        const int ty0 = (r0 / a.tiles_x) * WT_H, tx0 = (r0 % a.tiles_x) * WT_W;
        __syncthreads();
        if (VEC4) wgrad_stage_tile_vec(a, lds, n, ty0, tx0, HW);
        else wgrad_stage_tile_generic(a, lds, n, ty0, tx0, HW);
        __syncthreads();
        // 64 groups of 4 pixels per tile, 16 per wave: wave w takes rows 2w, 2w+1
#pragma unroll 1
        for (int g = 0; g < 16; ++g) {
            const int r = wave * 2 + (g >> 3), c0 = (g & 7) * 4;
            const int xb = r * WX_PITCH + c0, gb = r * WT_W + c0;
            const float a0 = gl[offA0 + gb], a1 = gl[offA1 + gb];
            float b[NT_B];
#pragma unroll
            for (int t = 0; t < NT_B; ++t) b[t] = lds[offB[t] + xb];
            if (ones) b[18] = 1.f;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                accA[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t], accA[t], 0, 0, 0);
            accA[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[18], accA[8], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT_B; ++t)
                accB[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t], accB[t], 0, 0, 0);
        }
    }
    // cross-wave reduction in LDS, fixed order (wave 0 stores, waves 1..3 add in turn)
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NT_ALL; ++t) {
                const f32x4 v = t < NT_A ? accA[t < NT_A ? t : 0] : accB[t >= NT_A ? t - NT_A : 0];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = t * 256 + (kq * 4 + q) * 16 + j;   // C row = kq*4+q, col = j
                    lds[idx] = (w == 0 ? 0.f : lds[idx]) + v[q];
                }
            }
        }
        __syncthreads();
    }
    float* part = a.partials + (size_t)blockIdx.x * WPART;
    for (int i = threadIdx.x; i < WPART; i += 256) part[i] = lds[i];
}

// partials [groups][28 tiles][16][16] -> the 12 gradient tensors in PyTorch layout
__global__ void gen_bwd_weight_reduce_kernel(const float* __restrict__ partials, int groups,
                                             GradPtrs G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NPARAM) return;
    int k = 0, co, col, p = 0, tap = 0;
    if (i < WF_TOTAL) {
        while (k < NL - 1 && i >= wf_off(k + 1)) ++k;
        const int cout = cout_of(k);
        const int r = i - wf_off(k);
        co = r % cout; tap = (r / cout) % 9; p = r / (cout * 9);
        col = p * 9 + tap;
    } else {
        while (k < NL - 1 && i >= bf_off(k + 1)) ++k;
        co = i - bf_off(k);
        col = BIAS_COL;
    }
    const int nt = col >> 4, jj = col & 15;
    int slot, row;
    if (k < 2) {                      // tile A: g0 rows 0..7, g1 rows 8..15
        row = k * 8 + co;
        slot = nt == 18 ? 8 : nt;
    } else {                          // tile B: g2 0..5, g3 6..9, g4 10..11, g5 12..13
        row = (k == 2 ? 0 : k == 3 ? 6 : k == 4 ? 10 : 12) + co;
        slot = NT_A + nt;
    }
    const size_t off = (size_t)slot * 256 + row * 16 + jj;
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += partials[(size_t)g * WPART + off];
    if (i < WF_TOTAL)
        G.w[k][(co * cin_of(k) + logical_of(k, p)) * 9 + tap] = s;
    else
        G.b[k][co] = s;
}

int wgrad_groups(int N, int H, int W) {
    const long tiles = (long)N * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    return (int)(tiles < WGRAD_MAX_GROUPS ? tiles : WGRAD_MAX_GROUPS);
}

// Frames are processed in chunks small enough for a chunk's planes (inputs + 28 features, or
// 28 features + 28 gradients) to stay resident in the 256 MiB Infinity Cache between layers.
int frames_per_pass(int N, int H, int W) {
    static const long budget = [] {
        const char* e = getenv("DMC_GEN_CACHE_MB");
        return (e ? atol(e) : (1L << 20)) << 20;   // default: one pass (chunking measured slower)
    }();
    const long per_frame = (long)(NIN + 2 * NFEAT) * H * W * sizeof(float);
    long f = budget / (per_frame > 0 ? per_frame : 1);
    return (int)(f < 1 ? 1 : (f > N ? N : f));
}

template <int MODE, int K>
int launch_layer(LayerArgs a, int n0, int N, hipStream_t s) {
    const size_t HW = (size_t)a.H * a.W;
    a.mv = a.mv ? a.mv + (size_t)n0 * 2 * HW : nullptr;
    a.res = a.res ? a.res + (size_t)n0 * 3 * HW : nullptr;
    a.feat = a.feat ? a.feat + (size_t)n0 * NFEAT * HW : nullptr;
    a.feat_out = a.feat_out ? a.feat_out + (size_t)n0 * NFEAT * HW : nullptr;
    a.gout = a.gout ? a.gout + (size_t)n0 * 2 * HW : nullptr;
    a.gbuf = a.gbuf ? a.gbuf + (size_t)n0 * NFEAT * HW : nullptr;
    a.out = a.out ? a.out + (size_t)n0 * 2 * HW : nullptr;
    const dim3 grid((a.W + LT - 1) / LT, (a.H + LT - 1) / LT, N);
    if (a.W % 4 == 0) gen_layer_kernel<MODE, K, true><<<grid, 256, 0, s>>>(a);
    else gen_layer_kernel<MODE, K, false><<<grid, 256, 0, s>>>(a);
    return check_launch("gen_layer");
}

int pack(const float* const* w, const float* const* b, float* pk, hipStream_t s) {
    ParamPtrs P;
    for (int k = 0; k < NL; ++k) {
        if (!w[k] || (b && !b[k])) return fail(DMC_E_INVALID, "null weight/bias pointer %d", k);
        P.w[k] = w[k];
        P.b[k] = b ? b[k] : w[k];   // bias slots are unused by the backward pass
    }
    pack_params_kernel<<<(PACKED_TOTAL + 255) / 256, 256, 0, s>>>(P, pk);
    return check_launch("pack_params");
}

}  // namespace

extern "C" {

size_t dmc_gen_tiny_workspace_bytes(void) { return (size_t)PACKED_TOTAL * sizeof(float); }

size_t dmc_gen_tiny_saved_bytes(int N, int H, int W) {
    return (size_t)N * NFEAT * H * W * sizeof(float);
}
size_t dmc_gen_tiny_gbuf_bytes(int N, int H, int W) { return dmc_gen_tiny_saved_bytes(N, H, W); }
size_t dmc_gen_tiny_partials_bytes(int N, int H, int W) {
    return (size_t)wgrad_groups(N, H, W) * WPART * sizeof(float);
}

int dmc_gen_tiny_fwd(const float* mv, const float* res, const float* const* w,
                     const float* const* b, float* out, float* saved, float* workspace, int N,
                     int H, int W, int add_mv_delta, dmc_stream_t stream) {
    if (!mv || !res || !w || !b || !out || !saved || !workspace)
        return fail(DMC_E_INVALID, "dmc_gen_tiny_fwd: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail(DMC_E_INVALID, "dmc_gen_tiny_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    int rc = pack(w, b, workspace, s);
    if (rc) return rc;
    LayerArgs a;
    a.mv = mv; a.res = res; a.feat = saved; a.feat_out = saved; a.gout = nullptr; a.gbuf = nullptr;
    a.pk = workspace; a.out = out; a.H = H; a.W = W; a.add_mv = add_mv_delta;
    const int step = frames_per_pass(N, H, W);
    for (int n0 = 0; n0 < N; n0 += step) {
        const int nn = (N - n0) < step ? (N - n0) : step;
        if ((rc = launch_layer<0, 0>(a, n0, nn, s))) return rc;
        if ((rc = launch_layer<0, 1>(a, n0, nn, s))) return rc;
        if ((rc = launch_layer<0, 2>(a, n0, nn, s))) return rc;
        if ((rc = launch_layer<0, 3>(a, n0, nn, s))) return rc;
        if ((rc = launch_layer<0, 4>(a, n0, nn, s))) return rc;
        if ((rc = launch_layer<1, 5>(a, n0, nn, s))) return rc;
    }
    return DMC_OK;
}

int dmc_gen_tiny_bwd(const float* mv, const float* res, const float* const* w, const float* saved,
                     const float* grad_out, float* const* dw, float* const* db, float* gbuf,
                     float* partials, float* workspace, int N, int H, int W, dmc_stream_t stream) {
    if (!mv || !res || !w || !saved || !grad_out || !dw || !db || !gbuf || !partials || !workspace)
        return fail(DMC_E_INVALID, "dmc_gen_tiny_bwd: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail(DMC_E_INVALID, "dmc_gen_tiny_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    GradPtrs G;
    for (int k = 0; k < NL; ++k) {
        if (!dw[k] || !db[k]) return fail(DMC_E_INVALID, "dmc_gen_tiny_bwd: null grad pointer %d", k);
        G.w[k] = dw[k];
        G.b[k] = db[k];
    }
    int rc = pack(w, nullptr, workspace, s);
    if (rc) return rc;
    LayerArgs la;
    la.mv = mv; la.res = res; la.feat = saved; la.feat_out = nullptr; la.gout = grad_out; la.gbuf = gbuf;
    la.pk = workspace; la.out = nullptr; la.H = H; la.W = W; la.add_mv = 0;
    const int step = frames_per_pass(N, H, W);
    for (int n0 = 0; n0 < N; n0 += step) {
        const int nn = (N - n0) < step ? (N - n0) : step;
        if ((rc = launch_layer<2, 5>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 4>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 3>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 2>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 1>(la, n0, nn, s))) return rc;
    }

    WgradArgs a;
    a.mv = mv; a.res = res; a.feat = saved; a.gout = grad_out; a.gbuf = gbuf; a.partials = partials;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + WT_W - 1) / WT_W;
    a.tiles_y = (H + WT_H - 1) / WT_H;
    const int groups = wgrad_groups(N, H, W);
    if (W % 4 == 0) gen_bwd_weight_kernel<true><<<groups, 256, 0, s>>>(a);
    else gen_bwd_weight_kernel<false><<<groups, 256, 0, s>>>(a);
    if ((rc = check_launch("gen_bwd_weight"))) return rc;
    gen_bwd_weight_reduce_kernel<<<(NPARAM + 127) / 128, 128, 0, s>>>(partials, groups, G);
    return check_launch("gen_bwd_weight_reduce");
}

}  // extern "C"
