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
#include "gen_x3.h"
#include "gen_fused.h"
#include "gen_wgrad.h"

using namespace dmc;

namespace {

// ------------------------------------------------------------------------------------------
// parameter repack: PyTorch [Cout][Cin_logical][3][3] -> WF | BF | WB (see dmc_common.h)
// ------------------------------------------------------------------------------------------
__global__ void pack_params_kernel(ParamPtrs P, float* __restrict__ pk, unsigned short* __restrict__ x3_frags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PACKED_TOTAL + ZERO_PAD) {          // the bf16x3 weight fragments of gen_x3.hip (forward only: x3_frags != null)
        if (x3_frags) gen_x3_pack_thread(P, x3_frags, i - (PACKED_TOTAL + ZERO_PAD));
        return;
    }
    if (i >= PACKED_TOTAL) {
        pk[i] = 0.f;                         // zero words: source of out-of-image LDS-DMA lanes
    } else if (i < WF_TOTAL) {
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
        int j = 0;
        while (j < 4 && i >= wb_off(j + 1)) ++j;
        const int D = cout_of(j);
        const int r = i - wb_off(j);
        const int cd = r % D, tap = (r / D) % 9, c = r / (D * 9);
        // input channel c of group j -> (layer k > j, its output channel cg): the gradient
        // channels are ordered g_{j+1}, ..., g_5
        int k = j + 1, cg = c;
        while (cg >= cout_of(k)) { cg -= cout_of(k); ++k; }
        // d/dx of a correlation is a correlation with the flipped kernel: tap -> 8 - tap
        pk[i] = P.w[k][(cg * cin_of(k) + logical_of(k, yoff(j) + cd)) * 9 + (8 - tap)];
    }
}

__device__ __forceinline__ const float* in_plane(const float* mv, const float* res,
                                                 const float* feat, int n, int p, size_t HW) {
    return p < 2 ? mv + ((size_t)n * 2 + p) * HW
                 : p < NIN ? res + ((size_t)n * 3 + (p - 2)) * HW
                           : feat + ((size_t)n * NFEAT + (p - NIN)) * HW;
}

// ------------------------------------------------------------------------------------------
// One 3x3 layer over a full-width row band per workgroup: 256 x 8 pixels, 512 threads.
//
//   lane = (row r of the tile, 4-pixel strip s); it produces 4 pixels x ALL output channels, so
//   every input value fetched from LDS feeds 3*COUT FMAs and the weights are wave-uniform:
//   they arrive through scalar loads and enter the FMAs as SGPR operands.
//   The tile spans the whole image width (224 <= 256), so there is NO horizontal halo to fetch:
//   with 32-pixel-wide tiles every 128-byte row segment dragged in two more cache lines for its
//   halo columns and the layer kernels ran at the HBM roof (PMC: 1,648 B/px forward, 5.7 TB/s).
//   Input planes are staged 4 channels at a time, branch-free (clamped addresses, values
//   zeroed by a select), as [10 rows][256] interior + two compact halo-column arrays, so the
//   lane's reads are one conflict-free ds_read_b128 and two conflict-free ds_read_b32 per row.
//   61 KB of LDS per workgroup -> 2 workgroups (16 waves) per CU; one workgroup's FMAs cover the
//   other's staging.
//
// MODE 0: forward hidden layer K   (inputs mv/res/feat[0..), output LeakyReLU(0.1) -> feat)
// MODE 1: forward last layer       (output -> out, optionally + mv)
// MODE 2: data gradient of feature group y_K (K = 4..0): gathers from every later layer's
//         gradient g_{K+1..5} in one pass and writes g_K = dL/dy_K * LeakyReLU'(y_K) once (no
//         read-modify-write of the gradient buffer)
// ------------------------------------------------------------------------------------------
constexpr int LTW = 256, LTH = 8;            // tile width / height (pixels)
constexpr int LNS = LTW / 4;                 // 64 strips per row = one wave per tile row
constexpr int LTHREADS = LNS * LTH;          // 512
constexpr int LCH = 4;                       // channels per staged chunk
constexpr int LROWS = LTH + 2;
constexpr int L_MAIN = LROWS * LTW;          // 2560 floats
constexpr int L_HALO = LROWS * LNS;          // 640 per side
constexpr int L_PLANE = L_MAIN + 2 * L_HALO; // 3840 floats per channel
constexpr int L_LDS = LCH * L_PLANE;         // 15360 floats = 61,440 B

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
    const float* mse_flow;   // [N,2,H,W] flow target: the fused layer-4+5 kernel also reduces sum((out - flow)^2) ...
    double* mse_part;        // ... into one double per workgroup (null: no loss)
};

template <int MODE, int K>
__device__ __forceinline__ const float* layer_in_plane(const LayerArgs& a, int n, int c, size_t HW) {
    if (MODE == 2) {
        constexpr int NG = gin_of(K) - 2;           // channels that live in gbuf; then grad_out
        return c < NG ? a.gbuf + ((size_t)n * NFEAT + (yoff(K + 1) - NIN) + c) * HW
                      : a.gout + ((size_t)n * 2 + (c - NG)) * HW;
    }
    return in_plane(a.mv, a.res, a.feat, n, c, HW);
}

template <int MODE, int K, bool VEC4>
__global__ __launch_bounds__(LTHREADS, 4) void gen_layer_kernel(LayerArgs a) {   // 2 WG/CU -> <= 128 VGPRs
    constexpr int CIN = MODE == 2 ? gin_of(K) : cin_of(K);
    constexpr int COUT = cout_of(K);
    __shared__ __attribute__((aligned(16))) float lds[L_LDS];
    const int n = blockIdx.z, ty0 = blockIdx.y * LTH, tx0 = blockIdx.x * LTW;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const int tid = threadIdx.x, s = tid % LNS;
    const int r = __builtin_amdgcn_readfirstlane(tid / LNS);       // tile row == wave index

    float acc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = MODE == 2 ? 0.f : a.pk[bf_off(K) + co];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][j] = bv;
    }
    const float* wbase = a.pk + (MODE == 2 ? wb_off(K) : wf_off(K));

    // Staging is software-pipelined through registers: the global loads of chunk c+1 are issued
    // before the FMAs of chunk c and written to LDS after them, so HBM/L2 latency overlaps
    // compute (PMC before this: waves parked 68 % of their cycles at s_waitcnt / s_barrier).
    // Wave w owns tile row w (waves 0,1 also rows 8,9): row, validity and plane base are
    // wave-uniform (scalar); only the column offset is per lane.
    const int xx = tx0 + 4 * s;
    const bool colok = xx < W;
    const int xl = xx - 1, xr = xx + 4;
    const bool okl = xl >= 0 && xl < W, okr = xr < W;
    const int ol = okl ? xl : 0, orr = okr ? xr : 0, oc = colok ? xx : 0;
    float4 sv[LCH][2];
    float shl[LCH][2], shr[LCH][2];

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            if (c0 + c < CIN) {
                const float* plane = layer_in_plane<MODE, K>(a, n, c0 + c, HW);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = r + pass * LTH;
                    if (row < LROWS) {                       // wave-uniform
                        int yy = ty0 - 1 + row;
                        yy = yy < 0 ? 0 : (yy < H ? yy : H - 1);
                        const float* src = plane + (size_t)yy * W;
                        if (VEC4) {
                            sv[c][pass] = *reinterpret_cast<const float4*>(src + oc);
                        } else {
                            sv[c][pass].x = src[xx + 0 < W ? xx + 0 : 0];
                            sv[c][pass].y = src[xx + 1 < W ? xx + 1 : 0];
                            sv[c][pass].z = src[xx + 2 < W ? xx + 2 : 0];
                            sv[c][pass].w = src[xx + 3 < W ? xx + 3 : 0];
                        }
                        shl[c][pass] = src[ol];
                        shr[c][pass] = src[orr];
                    }
                }
            }
        }
    };
    auto store_chunk = [&](int c0) {
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            if (c0 + c < CIN) {
                float* dl = lds + c * L_PLANE;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = r + pass * LTH;
                    if (row < LROWS) {
                        const int yy = ty0 - 1 + row;
                        const bool rowok = (yy >= 0) && (yy < H);
                        float4 v = sv[c][pass];
                        if (VEC4) {
                            if (!(rowok && colok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        } else {
                            v.x = (rowok && xx + 0 < W) ? v.x : 0.f;
                            v.y = (rowok && xx + 1 < W) ? v.y : 0.f;
                            v.z = (rowok && xx + 2 < W) ? v.z : 0.f;
                            v.w = (rowok && xx + 3 < W) ? v.w : 0.f;
                        }
                        *reinterpret_cast<float4*>(dl + row * LTW + 4 * s) = v;
                        dl[L_MAIN + row * LNS + s] = (rowok && okl) ? shl[c][pass] : 0.f;
                        dl[L_MAIN + L_HALO + row * LNS + s] = (rowok && okr) ? shr[c][pass] : 0.f;
                    }
                }
            }
        }
    };

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
#pragma unroll 1
    for (int c0 = 0; c0 < CIN; c0 += LCH) {
        const int nch = (CIN - c0) < LCH ? (CIN - c0) : LCH;
        const bool more = c0 + LCH < CIN;
        if (more) load_chunk(c0 + LCH);                      // in flight during the FMAs below
        // ---- accumulate ----
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const float* pl = lds + c * L_PLANE;
            float xv[3][6];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int row = r + ky;
                const float4 m = *reinterpret_cast<const float4*>(pl + row * LTW + 4 * s);
                xv[ky][0] = pl[L_MAIN + row * LNS + s];
                xv[ky][1] = m.x; xv[ky][2] = m.y; xv[ky][3] = m.z; xv[ky][4] = m.w;
                xv[ky][5] = pl[L_MAIN + L_HALO + row * LNS + s];
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
        if (more) {
            __syncthreads();                                 // everyone is done reading this chunk
            store_chunk(c0 + LCH);
            __syncthreads();
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
        float* dst = a.gbuf + ((size_t)n * NFEAT + (yoff(K) - NIN)) * HW + pix;
        const float* f = a.feat + ((size_t)n * NFEAT + (yoff(K) - NIN)) * HW + pix;
#pragma unroll
        for (int cd = 0; cd < COUT; ++cd) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (VEC4 || x0 + j < W)
                    dst[cd * HW + j] = acc[cd][j] * (f[cd * HW + j] > 0.f ? 1.f : 0.1f);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fast path of the same layer for W % 4 == 0 and W <= 256 (one tile spans the image width):
//   * input planes arrive by LDS-DMA: one 16-byte-per-lane instruction moves one whole row
//     (wave-uniform plane / row / validity, a lane only adds its column; lanes right of the image
//     read a zero word), 4 channels per chunk, double-buffered -- chunk c+1 is in flight while the
//     FMAs of chunk c run; no VGPRs hold data in flight, one barrier per chunk;
//   * no halo columns are fetched or stored at all: a lane's left / right neighbour pixels are
//     its neighbour LANES' quads, obtained with DPP wave_shr:1 / wave_shl:1 (zero shifted in at
//     the image border, which is exactly the convolution's zero padding).
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int D_PLANE = LROWS * LTW;               // 2560 floats per channel
constexpr int D_BUF = LCH * D_PLANE;               // 10240 floats = 40,960 B; x2 = 81,920 B

template <int MODE, int K>
__device__ __forceinline__ void layer_dma_chunk(const LayerArgs& a, float* buf, int n, int c0, int nch,
                                                int ty0, size_t HW, int wave, int lane,
                                                const float* zero) {
    const bool colok = 4 * lane < a.W;
#pragma unroll 1
    for (int rr = wave; rr < nch * LROWS; rr += LTH) {          // wave-uniform
        const int c = rr / LROWS, row = rr - c * LROWS;
        const int yy = ty0 - 1 + row;
        const float* plane = layer_in_plane<MODE, K>(a, n, c0 + c, HW);
        const float* src = (colok && yy >= 0 && yy < a.H) ? plane + (size_t)yy * a.W + 4 * lane : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + rr * LTW), 16, 0, 0);
    }
}

__device__ __forceinline__ float dpp_from_left(float v) {     // lane i <- lane i-1, lane 0 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {    // lane i <- lane i+1, lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

template <int MODE, int K>
__global__ __launch_bounds__(LTHREADS, 4) void gen_layer_dma_kernel(LayerArgs a) {
    constexpr int CIN = MODE == 2 ? gin_of(K) : cin_of(K);
    constexpr int COUT = cout_of(K);
    constexpr int NCHUNK = (CIN + LCH - 1) / LCH;
    __shared__ __attribute__((aligned(16))) float lds[2 * D_BUF];
    const int n = blockIdx.z, ty0 = blockIdx.y * LTH;
    const size_t HW = (size_t)a.H * a.W;
    const int tid = threadIdx.x, s = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);       // tile row == wave index
    const float* zero = a.pk + PACKED_TOTAL;

    layer_dma_chunk<MODE, K>(a, lds, n, 0, CIN < LCH ? CIN : LCH, ty0, HW, r, s, zero);

    float acc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = MODE == 2 ? 0.f : a.pk[bf_off(K) + co];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][j] = bv;
    }
    const float* wbase = a.pk + (MODE == 2 ? wb_off(K) : wf_off(K));
    __syncthreads();

#pragma unroll 1
    for (int ch = 0; ch < NCHUNK; ++ch) {
        const int c0 = ch * LCH;
        const int nch = (CIN - c0) < LCH ? (CIN - c0) : LCH;
        const float* buf = lds + (ch & 1) * D_BUF;
        if (ch + 1 < NCHUNK) {
            const int c1 = c0 + LCH;
            layer_dma_chunk<MODE, K>(a, lds + ((ch + 1) & 1) * D_BUF, n, c1,
                                     (CIN - c1) < LCH ? (CIN - c1) : LCH, ty0, HW, r, s, zero);
        }
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const float* pl = buf + c * D_PLANE;
            float xv[3][6];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float4 m = *reinterpret_cast<const float4*>(pl + (r + ky) * LTW + 4 * s);
                xv[ky][0] = dpp_from_left(m.w);
                xv[ky][1] = m.x; xv[ky][2] = m.y; xv[ky][3] = m.z; xv[ky][4] = m.w;
                xv[ky][5] = dpp_from_right(m.x);
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
        __syncthreads();     // next chunk's DMA has landed; this chunk's buffer may be refilled
    }

    const int y = ty0 + r, x0 = 4 * s;
    if (y >= a.H || x0 >= a.W) return;
    const size_t pix = (size_t)y * a.W + x0;
    if (MODE == 0) {
        float* dst = a.feat_out + ((size_t)n * NFEAT + (yoff(K) - NIN)) * HW + pix;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[co][j] > 0.f ? acc[co][j] : 0.1f * acc[co][j];
            *reinterpret_cast<float4*>(dst + co * HW) = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else if (MODE == 1) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float4 v = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
            if (a.add_mv) {
                const float4 m = *reinterpret_cast<const float4*>(a.mv + ((size_t)n * 2 + co) * HW + pix);
                v.x += m.x; v.y += m.y; v.z += m.z; v.w += m.w;
            }
            *reinterpret_cast<float4*>(a.out + ((size_t)n * 2 + co) * HW + pix) = v;
        }
    } else {
        float* dst = a.gbuf + ((size_t)n * NFEAT + (yoff(K) - NIN)) * HW + pix;
        const float* f = a.feat + ((size_t)n * NFEAT + (yoff(K) - NIN)) * HW + pix;
#pragma unroll
        for (int cd = 0; cd < COUT; ++cd) {
            const float4 fv = *reinterpret_cast<const float4*>(f + cd * HW);
            *reinterpret_cast<float4*>(dst + cd * HW) =
                make_float4(acc[cd][0] * (fv.x > 0.f ? 1.f : 0.1f), acc[cd][1] * (fv.y > 0.f ? 1.f : 0.1f),
                            acc[cd][2] * (fv.z > 0.f ? 1.f : 0.1f), acc[cd][3] * (fv.w > 0.f ? 1.f : 0.1f));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Matrix-core path of the same layer (W % 4 == 0, W <= 256): v_mfma_f32_4x4x1_16b_f32.
//
// The layers have 2..8 output channels, far too few to fill a 16- or 32-row MFMA tile.  The
// 4x4x1 instruction is 16 independent 4x4 outer products; seen across the wave it is
//     D[i][lane] += A[4*(lane/4) + i] * B[lane]          (i = 0..3, one K step)
// i.e. "4 output rows x 64 pixels, one pixel per lane", at the same 256 MAC / 8 cycles as the big
// shapes.  Mapping (one wave = one 256-pixel image row, 4 segments of 64 pixels):
//   * B = one staged input value per lane:  in[ci][y + dy - 1][x]      (a conflict-free ds_read_b32)
//   * K = (ci, dy): the vertical taps are GATHERED by reading the row above / below;
//   * rows rho = dx * COUT + co, 3*COUT of them in NT = ceil(3*COUT/4) row tiles: the horizontal
//     taps are PUSHED -- row (dx, co) accumulates  P_dx[co][x] = sum_{ci,dy} w[co][ci][dy][dx] *
//     in[ci][y+dy-1][x]  at the INPUT column, and the epilogue forms
//         out[co][x] = P_0[x-1] + P_1[x] + P_2[x+1]
//     with two DPP lane shifts per channel (wave_shr / wave_shl; the lane at a segment edge takes
//     the neighbour segment's edge lane through v_readlane).  Columns >= W hold zeros in LDS, so
//     image borders need no special case.
//   MAC efficiency of the tiles: COUT 8 -> 24/24 rows, 6 -> 18/20, 4 -> 12/12, 2 -> 6/8.
//
// Workgroup = 7 consumer waves (one tile row each) + 1 producer wave, persistent over a contiguous
// run of (tile, 4-channel chunk) items.  The producer stages chunks into a 3-deep LDS ring with
// LDS-DMA and is the only wave that ever blocks on the memory pipe; the consumers only see
// barriers.  (With every wave issuing its share of the DMA, all of them stalled at issue whenever
// the CU's miss queue was full, and staging and MFMA time simply added up.)
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_shr_fill(float cur, float lane0_value) {   // lane i <- cur[i-1]; lane 0 <- lane0_value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane0_value),
                                                                 __builtin_bit_cast(int, cur), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl_fill(float cur, float lane63_value) {  // lane i <- cur[i+1]; lane 63 <- lane63_value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lane63_value),
                                                                 __builtin_bit_cast(int, cur), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

constexpr int M_SEGS = LTW / 64;                    // 4 segments of 64 pixels per tile row
constexpr int PT_H = 8;                             // tile rows
constexpr int P_CONS = 7;                           // consumer waves; wave 7 is the producer
constexpr int P_MAXW = P_CONS * M_SEGS * 64 / PT_H; // 224: widest image the 7 x 256 pixel slots cover
constexpr int P_ROWS = PT_H + 2;                    // staged rows per channel
// (a staged channel of the default geometry is P_ROWS * LTW = 2560 floats, a chunk buffer LCH of them = 40,960 B: RingGeo below)
constexpr int RING = 3;                             // chunk buffers: one computing, two in flight
constexpr int RING_DMA = LCH * P_ROWS;              // 40 row transfers per chunk, all by the producer
static_assert(RING_DMA <= 63, "vmcnt is a 6-bit counter");

// Tile geometry of the ring kernels.  HALF = false: 8-row tiles, four pixels per lane, one 8-wave workgroup per CU (the
// default).  HALF = true (measurement variant, option gen_layer_path = 3, forward layers 2 and 3): 4-row tiles, TWO
// pixels per lane -- 7 consumer waves x 128 pixels = 4 x 224 -- with the ring sized by the layer's chunk, so that two
// workgroups fit a CU (<= 80 KB each, <= 128 VGPRs) and their push epilogues fall out of phase (round-2 verdict, variant ii).
// VAR = 2 (measurement variant iii, option gen_layer_path = 4, forward layer 2): the default tile, but a TWO-stage ring sized by
// the layer's chunk (61 KB + weights), so that two workgroups fit a CU with full-width accesses and no extra halo rows.
template <int VAR>
struct RingGeo {
    static constexpr bool HALF = VAR == 1;
    static constexpr int ROWS = HALF ? 4 : PT_H;            // tile rows
    static constexpr int SEGS = HALF ? 2 : M_SEGS;          // consecutive pixels per lane
    static constexpr int STAGES = VAR == 2 ? 2 : RING;      // chunk buffers of the ring
    static constexpr int SROWS = ROWS + 2;                  // staged rows per channel
    static constexpr int PLANE = SROWS * LTW;               // floats per staged channel
};

// Scheduling fences around each stage's MFMA group: needed while the kernels were register-starved
// (without them the compiler hoisted every LDS load of a chunk and spilled); since the producer /
// consumer split its own schedule is as good or better (layers 4+5: 286 -> 265 us), except for the
// 6-tile (Cout 8) push kernels, which keep them.
constexpr int FENCE_MIN_NT = 6;
template <int MODE, int K, int VAR = 0>
struct MfmaGeom {
    static constexpr int CIN = MODE == 2 ? gin_of(K) : cin_of(K);
    static constexpr int COUT = cout_of(K);
    static constexpr int NROW = 3 * COUT;
    static constexpr int NT = (NROW + 3) / 4;
    static constexpr int NTP = NT <= 2 ? 2 : NT <= 4 ? 4 : 8;     // padded tile count (vector loads)
    // channels per staged chunk: 4, except 3 for the compute-bound layer 2 (21 input channels pad
    // to nothing instead of 24: -12 % MFMAs; measured 242 -> 229 us).  Layer 3 (27 = 9 x 3) is at
    // the HBM roof and got slower with the smaller chunks (more barriers, less data in flight)
    // (VAR = 2, two workgroups per CU: layer 3's 27 channels in 3-channel chunks too, so that its two-stage ring fits 80 KB)
    static constexpr int CH = (MODE != 2 && (CIN == 21 || (VAR == 2 && CIN == 27))) ? 3 : LCH;
    static constexpr int NCHUNK = (CIN + CH - 1) / CH;
    static constexpr int DMA = CH * P_ROWS;                        // row transfers per chunk
    static constexpr bool FENCE = FENCE_MIN_NT <= NT;             // scheduling fences around the MFMA groups
    static constexpr int WL = NCHUNK * CH * 3 * 4 * NTP;           // LDS weight copy, zero rows for channels >= CIN
};

// LDS-DMA issued through inline assembly: the compiler does not see an LDS write, so it inserts
// no s_waitcnt vmcnt(0) in front of the consumers' ds_reads (with the builtin it does, which
// serialises every prefetch with the compute it was meant to overlap).  Completion is tracked
// by hand with s_waitcnt vmcnt(<row transfers per chunk>): "everything but the newest chunk has landed".
__device__ __forceinline__ void dma_row16(unsigned long long src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

__device__ __forceinline__ void dma_row4(unsigned long long src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                 :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

// scalar row base + per-lane byte offset: the producer's address arithmetic is all SALU, it never
// competes with the consumers' MFMAs for the vector issue port
__device__ __forceinline__ void dma_row16_s(unsigned long long sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

struct RingArgs {
    LayerArgs a;
    int tiles_y, ntiles;                              // row bands per frame, tiles in this launch
    int ablate;                                       // measurement only (option gen_ablate): 1 = no transfers, 2 = no epilogue stores
    int stagger = 0;                                  // gen_wino_kernel: start delay per phase step, in 10 ns ticks (option gen_stagger)
};

// producer: stage chunk c of tile (n, ty0) -- 4 channels x 10 rows, one 1 KB row per instruction.
// Only lanes left of the image's right edge are active (EXEC is restricted by the caller; the
// consumers never read staged columns >= W).  Channels >= CIN and rows outside the image come
// from the 1 KB of zero words behind the packed parameters.  All address arithmetic is scalar.
template <int MODE, int K, int HALF = 0>
__device__ __forceinline__ void ring_stage(const LayerArgs& a, unsigned slot_byte, int n, int ty0, int c,
                                           size_t HW, unsigned voff, const float* zero) {
    constexpr int CIN = MfmaGeom<MODE, K, HALF>::CIN;
    constexpr int PT_H = RingGeo<HALF>::ROWS, P_ROWS = RingGeo<HALF>::SROWS;       // (shadow the default geometry)
    const unsigned long long zaddr = (unsigned long long)zero;
    const bool interior = ty0 >= 1 && ty0 + PT_H < a.H;                            // rows ty0-1 .. ty0+PT_H all inside
    constexpr int CH = MfmaGeom<MODE, K, HALF>::CH;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
        const int ch = c * CH + cc;
        unsigned long long base;
        long pidx;
        if (MODE == 2) {
            constexpr int NG = gin_of(K) - 2;
            const long s1 = ch >= NG;                                  // 0: gbuf plane, 1: grad_out plane
            base = (unsigned long long)a.gbuf + s1 * ((unsigned long long)a.gout - (unsigned long long)a.gbuf);
            pidx = (long)n * (NFEAT - s1 * (NFEAT - 2)) + ch + (1 - s1) * (yoff(K + 1) - NIN) - s1 * NG;
        } else {
            const long s1 = ch >= 2, s2 = ch >= NIN;                   // mv | res | feat
            base = (unsigned long long)a.mv + s1 * ((unsigned long long)a.res - (unsigned long long)a.mv) +
                   s2 * ((unsigned long long)a.feat - (unsigned long long)a.res);
            pidx = (long)n * (2 + s1 + s2 * (NFEAT - 3)) + ch - s1 * 2 - s2 * (NIN - 2);
        }
        const bool chok = ch < CIN;
        // scalar pointer to row ty0-1 of the plane (never dereferenced while that row is outside the image)
        const unsigned long long row0 = base + (unsigned long long)((pidx * (long)HW + (long)(ty0 - 1) * a.W) * 4);
        unsigned long long sptr = chok ? row0 : zaddr;
        const unsigned long long sstride = chok ? (unsigned long long)a.W * 4 : 0;
        const unsigned dst = slot_byte + (unsigned)(cc * P_ROWS) * (LTW * 4);
        if (interior) {
#pragma unroll
            for (int row = 0; row < P_ROWS; ++row) {
                dma_row16_s(sptr, voff, dst + row * (LTW * 4));
                sptr += sstride;
            }
        } else {
#pragma unroll
            for (int row = 0; row < P_ROWS; ++row) {
                const int yy = ty0 - 1 + row;
                dma_row16_s((yy >= 0 && yy < a.H) ? sptr : zaddr, voff, dst + row * (LTW * 4));
                sptr += sstride;
            }
        }
    }
}

// consumer: one staged chunk = CH stages of 3 K-steps (dy) x 4 segments x NT row tiles.  The LDS
// operands of stage cc+1 are requested before the MFMAs of stage cc (register double buffer);
// sched_barriers keep the compiler from hoisting every load of the chunk to the top.
template <int MODE, int K, int NT_, int HALF = 0>
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[RingGeo<HALF>::SEGS][NT_], const float* buf, const float* wl,
                                           int c0, int boffq, int lane) {
    using G = MfmaGeom<MODE, K, HALF>;
    constexpr int M_SEGS = RingGeo<HALF>::SEGS, P_PLANE = RingGeo<HALF>::PLANE;    // (shadow the default geometry)
    float w[2][3][G::NTP], b[2][3][M_SEGS];
    auto load_stage = [&](int cc, int sel) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float* p = wl + (((c0 + cc) * 3 + dy) * 4 + (lane & 3)) * G::NTP;
            if (G::NTP == 2) {
                const float2 v = *reinterpret_cast<const float2*>(p);
                w[sel][dy][0] = v.x; w[sel][dy][1] = v.y;
            } else {
#pragma unroll
                for (int q = 0; q < G::NTP / 4; ++q) {
                    if (q * 4 < G::NT) {
                        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
                        w[sel][dy][4 * q + 0] = v.x; w[sel][dy][4 * q + 1] = v.y;
                        w[sel][dy][4 * q + 2] = v.z; w[sel][dy][4 * q + 3] = v.w;
                    }
                }
            }
            // the lane's consecutive pixels: one 16-byte (8-byte) read feeds the four (two) "segments"
            if constexpr (M_SEGS == 4) {
                const float4 bq = *reinterpret_cast<const float4*>(buf + cc * P_PLANE + dy * LTW + boffq);
                b[sel][dy][0] = bq.x; b[sel][dy][1] = bq.y; b[sel][dy][2] = bq.z; b[sel][dy][3] = bq.w;
            } else {
                const float2 bq = *reinterpret_cast<const float2*>(buf + cc * P_PLANE + dy * LTW + boffq);
                b[sel][dy][0] = bq.x; b[sel][dy][1] = bq.y;
            }
        }
    };
    load_stage(0, 0);
#pragma unroll
    for (int cc = 0; cc < G::CH; ++cc) {
        if (cc + 1 < G::CH) load_stage(cc + 1, (cc + 1) & 1);
        if (G::FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
                for (int t = 0; t < G::NT; ++t)
                    acc[s][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[cc & 1][dy][t], b[cc & 1][dy][s], acc[s][t], 0, 0, 0);
        if (G::FENCE) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MODE, int K, int HALF = 0>                  // HALF: the RingGeo variant (0 default, 1 half tiles, 2 two-stage ring)
__global__ __launch_bounds__(LTHREADS, HALF ? 4 : 2) void gen_layer_mfma_kernel(RingArgs ra) {
    using G = MfmaGeom<MODE, K, HALF>;
    constexpr int CIN = G::CIN, COUT = G::COUT, NT = G::NT, NTP = G::NTP, NCHUNK = G::NCHUNK;
    // (shadow the default geometry; the half-tile ring is sized by the layer's chunk so that two workgroups fit a CU)
    constexpr int PT_H = RingGeo<HALF>::ROWS, M_SEGS = RingGeo<HALF>::SEGS;
    constexpr int P_BUF = (HALF ? G::CH : LCH) * RingGeo<HALF>::PLANE;
    constexpr int RING = RingGeo<HALF>::STAGES, AHEAD = RING - 1;      // (shadow) chunks the producer runs ahead of the consumers
    __shared__ __attribute__((aligned(16))) float lds[RING * P_BUF + G::WL + P_CONS * 2 * 8];
    const LayerArgs& a = ra.a;
    float* wl = lds + RING * P_BUF;
    float* xchg = wl + G::WL;                                      // [wave][first P_2 | last P_0][co]
    const size_t HW = (size_t)a.H * a.W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave: 0..6 = consumers, 7 = producer
    const float* zero = a.pk + PACKED_TOTAL;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;          // LDS byte address of the ring

    // Tile schedule: workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8), each with
    // its own L2.  In round k the nwg/8 workgroups of one XCD take nwg/8 CONSECUTIVE tiles (vertically
    // adjacent bands of the same frames), so the halo rows two neighbouring bands share are fetched
    // through the same L2 at about the same time and go to HBM once.  (Contiguous per-workgroup runs
    // re-fetched every halo row from HBM: PMC 767 B/px forward instead of the 618 B/px of the old
    // one-tile-per-workgroup grid.)
    const int nwg = gridDim.x;
    const int t_begin = nwg % 8 == 0 ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int t_step = nwg;
    const int nitems = t_begin < ra.ntiles ? (ra.ntiles - t_begin + nwg - 1) / nwg * NCHUNK : 0;
    if (nitems == 0) return;

    // weights -> LDS as [ci][dy][i][NTP]: element (ci, dy, rho = 4*t + i), rho = dx*COUT + co
    const float* wbase = a.pk + (MODE == 2 ? wb_off(K) : wf_off(K));
    for (int idx = tid; idx < G::WL; idx += LTHREADS) {
        const int t = idx % NTP, i = (idx / NTP) % 4, dy = (idx / (4 * NTP)) % 3, ci = idx / (12 * NTP);
        const int rho = 4 * t + i;
        wl[idx] = (t < NT && rho < G::NROW && ci < CIN) ? wbase[(ci * 9 + dy * 3 + rho / COUT) * COUT + rho % COUT] : 0.f;
    }
    __syncthreads();

    if (r == P_CONS) {
        // ------------------------------ producer wave ------------------------------
        if (4 * lane >= a.W) return;                                   // EXEC = lanes that hold image columns (a staged row = 64 lanes x 16 bytes either way)
        const unsigned voff = (unsigned)lane * 16;
#pragma unroll 1
        for (int pre = 0; pre < AHEAD && pre < nitems; ++pre) {
            const int tile = t_begin + (pre / NCHUNK) * t_step, n = tile / ra.tiles_y;
            ring_stage<MODE, K, HALF>(a, lds0 + pre * (P_BUF * 4), n, (tile - n * ra.tiles_y) * PT_H, pre % NCHUNK, HW, voff, zero);
        }
        int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
        for (int q = 0; q < nitems; ++q) {
            // chunk q has landed (only chunk q+1 may still be in flight); consumers are done with q-1
            if (AHEAD == 2 && q + 1 < nitems && !(DMC_ABL(ra.ablate) & 1)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(G::CH * RingGeo<HALF>::SROWS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (q + AHEAD < nitems && !(DMC_ABL(ra.ablate) & 1)) {
                const int c2 = c + AHEAD, tile2 = tile + (c2 / NCHUNK) * t_step, n2 = tile2 / ra.tiles_y;
                int slot2 = slot + AHEAD; slot2 = slot2 >= RING ? slot2 - RING : slot2;
                ring_stage<MODE, K, HALF>(a, lds0 + (unsigned)slot2 * (P_BUF * 4), n2, (tile2 - n2 * ra.tiles_y) * PT_H,
                                    c2 % NCHUNK, HW, voff, zero);
            }
            slot = slot + 1 == RING ? 0 : slot + 1;
            if (++c == NCHUNK) {
                c = 0; tile += t_step;
                asm volatile("s_barrier" ::: "memory");           // the consumers' edge exchange of this tile
            }
        }
        return;
    }

    // ------------------------------ consumer waves ------------------------------
    // The tile's PT_H x W pixels are numbered row-major, p = y*W + x; consumer wave r owns pixels
    // [256 r, 256 r + 256) and lane l of it the FOUR CONSECUTIVE pixels 256 r + 4 l + e, e = 0..3 (W % 4 ==
    // 0: a quad never straddles rows; a 224-wide row is 56 lanes, so waves straddle rows and nothing is
    // wasted on columns >= W).  MFMA "segment" e = element e of every lane's quad: its B operands come
    // from one 16-byte LDS read, the horizontal neighbours of pixels e = 1, 2 are the lane's own
    // accumulators (only e = 0 / e = 3 need a lane shift), and every global access is 16 bytes per lane.
    const int p0 = r * (M_SEGS * 64) + M_SEGS * lane;
    const int yl0 = p0 / a.W, x0 = p0 - yl0 * a.W;
    const int boffq = p0 < PT_H * a.W ? yl0 * LTW + x0 : 0;   // LDS offset of the quad in staged row 0 (the row above it)
    const bool at_left = x0 == 0, at_right = x0 + M_SEGS == a.W;
    f32x4 acc[M_SEGS][NT];
#pragma unroll
    for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
    for (int q = 0; q < nitems; ++q) {
        asm volatile("s_barrier" ::: "memory");
        if (!(DMC_ABL(ra.ablate) & 4) || r < 4)      // ablate 4: one computing wave per SIMD (waves 4 .. 6 only keep the barriers)
            mfma_chunk<MODE, K, NT, HALF>(acc, lds + slot * P_BUF, wl, c * G::CH, boffq, lane);
        slot = slot + 1 == RING ? 0 : slot + 1;
        if (++c < NCHUNK) continue;
        c = 0;

        // ---- epilogue: horizontal taps (own registers / lane shifts), then bias / activation / store ----
        const int n = tile / ra.tiles_y, ty0 = (tile - n * ra.tiles_y) * PT_H;
        const int pmax = (a.H - ty0 < PT_H ? a.H - ty0 : PT_H) * a.W;      // pixels of this tile inside the image
        const size_t tile_pix = (size_t)ty0 * a.W;
        const bool inside = p0 < pmax;
        const size_t pix = tile_pix + (inside ? p0 : 0);
        float4 extra[COUT];              // MODE 1: mv (delta add), MODE 2: y_K (LeakyReLU'): one batch of loads
        if (MODE != 0) {
            static_assert(MODE == 0 || HALF != 1, "the half-tile variant is built for the forward's inner layers only");
#pragma unroll
            for (int co = 0; co < COUT; ++co)
                extra[co] = MODE == 1 ? (a.add_mv ? *reinterpret_cast<const float4*>(a.mv + ((size_t)n * 2 + co) * HW + pix)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f))
                                      : *reinterpret_cast<const float4*>(a.feat + ((size_t)n * NFEAT + (yoff(K) - NIN) + co) * HW + pix);
        }
        // the pixel left of this wave's first one / right of its last one belongs to the neighbour
        // wave: exchange those two partial sums per channel through LDS
        float nb_l[COUT], nb_r[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const int r0 = co, r2 = 2 * COUT + co;
            if (lane == 0) xchg[(r * 2 + 0) * 8 + co] = acc[0][r2 / 4][r2 % 4];
            if (lane == 63) xchg[(r * 2 + 1) * 8 + co] = acc[M_SEGS - 1][r0 / 4][r0 % 4];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            nb_l[co] = r > 0 ? xchg[((r - 1) * 2 + 1) * 8 + co] : 0.f;
            nb_r[co] = r + 1 < P_CONS ? xchg[((r + 1) * 2 + 0) * 8 + co] : 0.f;
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const int r0 = co, r1 = COUT + co, r2 = 2 * COUT + co;
            // (the shifts run with all lanes active -- a DPP read from an inactive lane is invalid -- and
            // only then are the image-border lanes masked: the zero padding of the convolution)
            const float sh_l = dpp_shr_fill(acc[M_SEGS - 1][r0 / 4][r0 % 4], nb_l[co]);
            const float sh_r = dpp_shl_fill(acc[0][r2 / 4][r2 % 4], nb_r[co]);
            float v[M_SEGS];
#pragma unroll
            for (int e = 0; e < M_SEGS; ++e) {
                const float from_l = e > 0 ? acc[e > 0 ? e - 1 : 0][r0 / 4][r0 % 4] : (at_left ? 0.f : sh_l);
                const float from_r = e + 1 < M_SEGS ? acc[e + 1 < M_SEGS ? e + 1 : e][r2 / 4][r2 % 4] : (at_right ? 0.f : sh_r);
                v[e] = acc[e][r1 / 4][r1 % 4] + from_l + from_r;
            }
            const float ex[4] = {MODE != 0 ? extra[co].x : 0.f, MODE != 0 ? extra[co].y : 0.f,
                                 MODE != 0 ? extra[co].z : 0.f, MODE != 0 ? extra[co].w : 0.f};
#pragma unroll
            for (int e = 0; e < M_SEGS; ++e) {
                if (MODE == 0) {
                    v[e] += a.pk[bf_off(K) + co];
                    v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
                } else if (MODE == 1) {
                    v[e] += a.pk[bf_off(K) + co];
                    v[e] += ex[e];
                } else {
                    v[e] *= ex[e] > 0.f ? 1.f : 0.1f;
                }
            }
            if constexpr (HALF == 1) {
                if (inside && !(DMC_ABL(ra.ablate) & 2))
                    *reinterpret_cast<float2*>(a.feat_out + ((size_t)n * NFEAT + (yoff(K) - NIN) + co) * HW + pix) = make_float2(v[0], v[1]);
            } else if (inside && !(DMC_ABL(ra.ablate) & 2)) {
                const float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (MODE == 0) *reinterpret_cast<float4*>(a.feat_out + ((size_t)n * NFEAT + (yoff(K) - NIN) + co) * HW + pix) = o;
                else if (MODE == 1) *reinterpret_cast<float4*>(a.out + ((size_t)n * 2 + co) * HW + pix) = o;
                else *reinterpret_cast<float4*>(a.gbuf + ((size_t)n * NFEAT + (yoff(K) - NIN) + co) * HW + pix) = o;
            }
        }
#pragma unroll
        for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile += t_step;
    }
}

// ------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) form of the ring kernel (option gen_wino; hidden layers of the forward and the data-gradient groups).
//
// The direct kernels above are bound by the matrix pipe (ablations: `gen_ablate`; one computing wave per SIMD runs the
// 4x4x1 MFMAs back to back at 8.6 cycles each): 9 multiplications per output value and (cin, cout) pair.  The minimal
// filtering algorithm F(2x2, 3x3) needs 16 per 2 x 2 OUTPUT BLOCK and pair -- 4 per output value, 2.25x fewer:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d = the block's 4 x 4 input window, g = the 3 x 3 filter
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// In the transformed domain a layer is 16 independent [Cout x Cin] x [Cin x blocks] products -- again v_mfma_f32_4x4x1
// with "one pixel per lane", the pixel now being a 2 x 2 output block and the K-steps (cin, position): rows = output
// channels (no push rows, no horizontal-tap epilogue, no exchange between waves), 16 Cin NT MFMAs per 64 blocks = 256
// pixels where the push form issues 4 x 3 Cin NT' (layer 2: 672 against 1,260 per wave and tile).
// Same tile (8 rows x W <= 224 pixels = 4 x 112 blocks = 7 consumer waves x 64 lanes), same producer wave, same
// three-stage ring of LDS-DMA'd channel chunks as gen_layer_mfma_kernel (ring_stage is shared).  Per staged channel a
// lane reads its block's 4 x 4 window (columns 2 tc - 1 .. 2 tc + 2: the columns left of 0 and right of W - 1 are the
// 32 pad floats of a 256-float LDS row, zeroed once -- the transfers are EXEC-masked to the image width and never
// touch them), transforms it with 32 additions (B^T d B has no multiplications) and issues the 16 NT MFMAs; the
// transformed filters U = G g G^T are computed once per workgroup (in double, rounded once) into LDS as
// [ci][row tile][4 positions][row-in-tile][4].  Epilogue: A^T M A per output channel (24 additions), bias +
// LeakyReLU (forward) or LeakyReLU' of the saved feature (data gradient), 8-byte stores (two pixels of a row per lane:
// 512 contiguous bytes per wave and row).
// Arithmetic: fp32 throughout; the result differs from the direct form's by the usual Winograd rounding (the transforms
// add / subtract values of like magnitude: measured <= 3x the direct kernels' distance from an fp64 evaluation,
// tests/test_gen_wino_gpu.py), not bit for bit.
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
// One staged channel's window: three 8-byte reads per row at constant offsets from one address.  VOLATILE loads: the compiler
// neither pairs them into ds_read2_b64 (half the LDS rate) nor into dword pairs that need an address register each, and it
// still counts them in its s_waitcnt bookkeeping.
typedef const volatile __attribute__((address_space(3))) f32x2* wino_lds_t;
__device__ __forceinline__ void wino_window(f32x2 (&d)[4][3], const float* p) {
    const wino_lds_t q = (wino_lds_t)(__attribute__((address_space(3))) void*)(void*)const_cast<float*>(p);
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int k = 0; k < 3; ++k) d[y][k] = q[y * (LTW / 2) + k];
}
constexpr int WN_PAD = 4;                          // floats in front of the ring: column -1 of its first row
template <int MODE, int K>
struct WinoGeom {
    using G = MfmaGeom<MODE, K, 0>;
    static constexpr int CIN = G::CIN, COUT = G::COUT, CH = G::CH, NCHUNK = G::NCHUNK;
    static constexpr int NT = (COUT + 3) / 4;                     // row tiles = output channels / 4
    static constexpr int WL = NCHUNK * CH * NT * 64;              // [ci][t][position quad][row-in-tile][4], zero for ci >= CIN
};
__host__ __device__ inline bool wino_shape_ok(int H, int W) { return W % 4 == 0 && W <= P_MAXW && W >= 64 && H >= 1; }

template <int MODE, int K, int DBG = 0>     // DBG (measurement only, option gen_ablate >> 8): 1 no transform, 2 no MFMAs, 3 no window reads, 4 no filter reads
__global__ __launch_bounds__(LTHREADS, 2) void gen_wino_kernel(RingArgs ra) {
    static_assert(MODE == 0 || MODE == 2, "hidden layers of the forward / data-gradient groups");
    using WG = WinoGeom<MODE, K>;
    using G = MfmaGeom<MODE, K, 0>;
    constexpr int CIN = WG::CIN, COUT = WG::COUT, NT = WG::NT, NCHUNK = WG::NCHUNK, CH = WG::CH;
    constexpr int P_BUF = LCH * RingGeo<0>::PLANE;
    constexpr int AHEAD = RING - 1;
    __shared__ __attribute__((aligned(16))) float lds_all[WN_PAD + RING * P_BUF + WG::WL];
    float* lds = lds_all + WN_PAD;
    float* wl = lds + RING * P_BUF;
    const LayerArgs& a = ra.a;
    const size_t HW = (size_t)a.H * a.W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave: 0..6 = consumers, 7 = producer
    const float* zero = a.pk + PACKED_TOTAL;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;

    const int nwg = gridDim.x;                                    // XCD-aware tile schedule, as above
    const int t_begin = nwg % 8 == 0 ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int t_step = nwg;
    const int nitems = t_begin < ra.ntiles ? (ra.ntiles - t_begin + nwg - 1) / nwg * NCHUNK : 0;
    if (nitems == 0) return;

    // transformed filters U[pa][pb] = sum_{dy,dx} G[pa][dy] G[pb][dx] g[dy][dx]
    const float* wbase = a.pk + (MODE == 2 ? wb_off(K) : wf_off(K));
    for (int idx = tid; idx < WG::WL; idx += LTHREADS) {
        const int e = idx & 3, i = (idx >> 2) & 3, pq = (idx >> 4) & 3, t = (idx >> 6) % NT, ci = idx / (64 * NT);
        const int co = 4 * t + i;
        double u = 0.0;
        if (co < COUT && ci < CIN) {
            const double Gm[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    u += Gm[pq][dy] * Gm[e][dx] * (double)wbase[(ci * 9 + dy * 3 + dx) * COUT + co];
        }
        wl[idx] = (float)u;
    }
    // the pad columns [W, 256) of every ring row (and the four floats in front of the ring) are the zero padding in x
    {
        const int padw = LTW - a.W;
        for (int idx = tid; idx < RING * LCH * P_ROWS * padw; idx += LTHREADS) {
            const int row = idx / padw, col = a.W + (idx - row * padw);
            lds[row * LTW + col] = 0.f;
        }
        if (tid < WN_PAD) lds_all[tid] = 0.f;
    }
    // Staggered start: a tile ends with a burst of stores (COUT planes x 8 rows), and workgroups that run in phase all write
    // at the same time -- the chip alternates between a read / compute phase and a write phase that drains at the HBM rate with
    // nothing else in flight (38 us of 116 for layer 0).  Workgroups started (b / 8 mod 8) steps apart keep the mix constant.
    if (DMC_ABL(ra.stagger) > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 7) * (unsigned)ra.stagger;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();

    if (r == P_CONS) {
        // ------------------------------ producer wave (as in gen_layer_mfma_kernel, no epilogue barrier) ------------------------------
        if (4 * lane >= a.W) return;
        const unsigned voff = (unsigned)lane * 16;
#pragma unroll 1
        for (int pre = 0; pre < AHEAD && pre < nitems; ++pre) {
            const int tile = t_begin + (pre / NCHUNK) * t_step, n = tile / ra.tiles_y;
            ring_stage<MODE, K, 0>(a, lds0 + pre * (P_BUF * 4), n, (tile - n * ra.tiles_y) * PT_H, pre % NCHUNK, HW, voff, zero);
        }
        int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
        for (int q = 0; q < nitems; ++q) {
            if (q + 1 < nitems && !(DMC_ABL(ra.ablate) & 1)) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(G::CH * P_ROWS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (q + AHEAD < nitems && !(DMC_ABL(ra.ablate) & 1)) {
                const int c2 = c + AHEAD, tile2 = tile + (c2 / NCHUNK) * t_step, n2 = tile2 / ra.tiles_y;
                int slot2 = slot + AHEAD; slot2 = slot2 >= RING ? slot2 - RING : slot2;
                ring_stage<MODE, K, 0>(a, lds0 + (unsigned)slot2 * (P_BUF * 4), n2, (tile2 - n2 * ra.tiles_y) * PT_H,
                                       c2 % NCHUNK, HW, voff, zero);
            }
            slot = slot + 1 == RING ? 0 : slot + 1;
            if (++c == NCHUNK) { c = 0; tile += t_step; }
        }
        return;
    }

    // ------------------------------ consumer waves ------------------------------
    // lane = one 2 x 2 output block of the tile: block row tr (0..3), block column tc (0 .. W/2 - 1)
    const int bx = a.W >> 1;
    const int t = r * 64 + lane;
    const bool valid = t < 4 * bx;
    const int tv = valid ? t : 0;
    const int tr = tv / bx, tc = tv - tr * bx;
    // window origin: staged row 2 tr (image row ty0 + 2 tr - 1), column 2 tc - 2: three 8-byte reads per row fetch columns
    // 2 tc - 2 .. 2 tc + 3 (the window is the middle four): ds_read_b64 moves 256 B per clock without bank conflicts where
    // dword reads at this two-dword lane stride are 2-way conflicted at 128 B per clock
    const int wofs = (2 * tr) * LTW + 2 * tc - 2;
    const float* wl_lane = wl + (lane & 3) * 4;
    f32x4 acc[16][NT];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[p][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transformed filters of one (channel, row tile) GROUP = 16 positions = four 16-byte reads; group g + 1 is requested
    // before the 16 MFMAs of group g (register double buffer), across chunk and tile boundaries too (the table is static)
    constexpr int NG = CH * NT, PLANE = RingGeo<0>::PLANE;
    static_assert(NG % 2 == 0, "the buffer parity of a chunk's first group is fixed");
    float4 wq[2][4];
    auto load_group = [&](float4 (&w)[4], int g) {
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) w[pq] = *reinterpret_cast<const float4*>(wl_lane + (g * 4 + pq) * 16);
    };
    load_group(wq[0], 0);

    int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
    for (int q = 0; q < nitems; ++q) {
        asm volatile("s_barrier" ::: "memory");
        {
            const float* buf = lds + slot * P_BUF + wofs;
            const int cn = c + 1 == NCHUNK ? 0 : c + 1;
            f32x2 d[4][3];
            if (DBG == 3) {
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int k = 0; k < 3; ++k) d[y][k] = (f32x2){(float)(lane + y), (float)(k + q)};
            } else wino_window(d, buf);
#pragma unroll
            for (int cc = 0; cc < CH; ++cc) {
                // B^T d: rows combined, vectorised over the column pairs (v_pk_add_f32; the unused outer halves fall away)
                f32x2 u[4][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u[0][k] = d[0][k] - d[2][k]; u[1][k] = d[1][k] + d[2][k];
                    u[2][k] = d[2][k] - d[1][k]; u[3][k] = d[1][k] - d[3][k];
                }
                if (cc + 1 < CH && DBG != 3) wino_window(d, buf + (cc + 1) * PLANE);    // in flight behind this channel's MFMAs
                // (B^T d) B: columns -1, 0, 1, 2 of a row are u[.][0].y, u[.][1].x, u[.][1].y, u[.][2].x
                float v[4][4];
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    v[y][0] = u[y][0].y - u[y][1].y;
                    v[y][1] = u[y][1].x + u[y][1].y;
                    v[y][2] = u[y][1].y - u[y][1].x;
                    v[y][3] = u[y][1].x - u[y][2].x;
                    if (DBG == 1) { v[y][0] = d[y][0].y; v[y][1] = d[y][1].x; v[y][2] = d[y][1].y; v[y][3] = d[y][2].x; }
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int g = cc * NT + tt;
                    if (DBG == 4) { }
                    else if (g + 1 < NG) load_group(wq[(g + 1) & 1], (c * CH) * NT + g + 1);
                    else load_group(wq[0], (cn * CH) * NT);
                    __builtin_amdgcn_sched_barrier(0);           // (the compiler would sink the requests down to their first use)
                    const float4 (&w)[4] = wq[g & 1];
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) {
                        if (DBG == 2) {          // the operands stay alive, the matrix pipe idles
                            asm volatile("" :: "v"(w[pq].x), "v"(w[pq].y), "v"(w[pq].z), "v"(w[pq].w), "v"(v[pq][0]), "v"(v[pq][1]), "v"(v[pq][2]), "v"(v[pq][3]));
                            continue;
                        }
                        acc[4 * pq + 0][tt] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[pq].x, v[pq][0], acc[4 * pq + 0][tt], 0, 0, 0);
                        acc[4 * pq + 1][tt] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[pq].y, v[pq][1], acc[4 * pq + 1][tt], 0, 0, 0);
                        acc[4 * pq + 2][tt] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[pq].z, v[pq][2], acc[4 * pq + 2][tt], 0, 0, 0);
                        acc[4 * pq + 3][tt] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[pq].w, v[pq][3], acc[4 * pq + 3][tt], 0, 0, 0);
                    }
                }
            }
        }
        slot = slot + 1 == RING ? 0 : slot + 1;
        if (++c < NCHUNK) continue;
        c = 0;

        // ---- epilogue: A^T M A per output channel, bias / activation, store ----
        const int n = tile / ra.tiles_y, ty0 = (tile - n * ra.tiles_y) * PT_H;
        const int y0 = ty0 + 2 * tr;
        const bool ok0 = valid && y0 < a.H, ok1 = valid && y0 + 1 < a.H;
        const size_t pix0 = (size_t)(ok0 ? y0 : 0) * a.W + 2 * tc, pix1 = (size_t)(ok1 ? y0 + 1 : 0) * a.W + 2 * tc;
        float2 ex0[COUT], ex1[COUT];                 // MODE 2: the saved feature y_K (LeakyReLU'), one batch of loads
        if (MODE == 2) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float* fp = a.feat + ((size_t)n * NFEAT + (yoff(K) - NIN) + co) * HW;
                ex0[co] = *reinterpret_cast<const float2*>(fp + pix0);
                ex1[co] = *reinterpret_cast<const float2*>(fp + pix1);
            }
        }
        // two output channels at a time: components (2 h, 2 h + 1) of an accumulator are a register pair -> v_pk_add_f32
        // (the same additions in the same order as one channel at a time: bit-identical)
#pragma unroll
        for (int cp = 0; cp < (COUT + 1) / 2; ++cp) {
            const int tt = cp >> 1, hh = cp & 1;
            auto M = [&](int p) { return (f32x2){acc[p][tt][2 * hh], acc[p][tt][2 * hh + 1]}; };
            f32x2 s0[4], s1[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const f32x2 m0 = M(x), m1 = M(4 + x), m2 = M(8 + x), m3 = M(12 + x);
                s0[x] = (m0 + m1) + m2;
                s1[x] = (m1 - m2) - m3;
            }
            const f32x2 t00 = (s0[0] + s0[1]) + s0[2], t01 = (s0[1] - s0[2]) - s0[3];
            const f32x2 t10 = (s1[0] + s1[1]) + s1[2], t11 = (s1[1] - s1[2]) - s1[3];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int co = 2 * cp + e;
                if (co >= COUT) continue;
                float o[2][2] = {{t00[e], t01[e]}, {t10[e], t11[e]}};
                const float exv[2][2] = {{MODE == 2 ? ex0[co].x : 0.f, MODE == 2 ? ex0[co].y : 0.f},
                                         {MODE == 2 ? ex1[co].x : 0.f, MODE == 2 ? ex1[co].y : 0.f}};
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        if (MODE == 0) {
                            o[y][x] += a.pk[bf_off(K) + co];
                            o[y][x] = o[y][x] > 0.f ? o[y][x] : 0.1f * o[y][x];
                        } else {
                            o[y][x] *= exv[y][x] > 0.f ? 1.f : 0.1f;
                        }
                    }
                // (ablate 8: the arithmetic without the stores.  Measured and not kept: the results held in registers and stored a
                // few channels per chunk of the NEXT tile -- no gain without spills, a loss with them: the stores' cost is their HBM
                // traffic, not the moment of their issue)
                float* op = (MODE == 0 ? a.feat_out : a.gbuf) + ((size_t)n * NFEAT + (yoff(K) - NIN) + co) * HW;
                const bool st = !(DMC_ABL(ra.ablate) & 2) && (!(DMC_ABL(ra.ablate) & 8) || o[0][0] == 1.2345678e33f);
                if (st) {
                    if (ok0) *reinterpret_cast<float2*>(op + pix0) = make_float2(o[0][0], o[0][1]);
                    if (ok1) *reinterpret_cast<float2*>(op + pix1) = make_float2(o[1][0], o[1][1]);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) acc[p][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile += t_step;
    }
}

// ------------------------------------------------------------------------------------------
// Layers 4 and 5 fused (forward).  Both read the same 31 planes (mv, res, y0..y3) and sit at the
// HBM roof, so one pass over those planes serves both: while the chunks stream through the ring,
// layer 4 accumulates for tile rows ty0-1 .. ty0+8 (10 rows: the rows layer 5 needs of y4) and
// layer 5 accumulates its 31-channel part for rows ty0 .. ty0+7; then y4 = LeakyReLU(.) goes to
// HBM (the 8 owned rows) and into an LDS tile (all 10 rows, zero outside the image), layer 5 adds
// its two y4 channels from that tile and finishes (+bias, +mv).  12 staged rows per 8 instead of
// 2 x 10, and the second read of the 31 planes disappears: -40 % of these two layers' traffic.
// Same push formulation, flattened pixel slots and producer / consumer ring as above; consumer
// wave r owns 256 pixels for both layers (shared B operands) plus 64 halo pixels of layer 4.
// ------------------------------------------------------------------------------------------
constexpr int F_LCH = 3;                             // channels per chunk
constexpr int F_ROWS = PT_H + 4;                     // staged rows ty0-2 .. ty0+9
constexpr int F_PLANE = F_ROWS * LTW;                // 3072 floats
constexpr int F_BUF = F_LCH * F_PLANE;               // 9216 floats = 36,864 B per chunk
constexpr int F_DMA = F_LCH * F_ROWS;                // 36 row transfers per chunk
constexpr int F_CIN = cin_of(4);                     // 31 streamed channels
constexpr int F_NCHUNK = (F_CIN + F_LCH - 1) / F_LCH;   // 11
constexpr int F_CH = F_NCHUNK * F_LCH;               // 33 (channels 31, 32: zero rows)
constexpr int F_S5 = 4;                              // shared segments per wave (the tile's 8 rows); + 1 halo segment for layer 4
constexpr int F_Y4ROWS = PT_H + 2;
constexpr int F_Y4 = 2 * F_Y4ROWS * LTW;             // y4 tile: 2 channels x 10 rows x 256
constexpr int F_WL = F_CH * 3 * 4 * 4;               // combined weight table [ci][dy][i][4]: row tiles 0..2 (rows 0..5 layer 4, 6..11 layer 5)
static_assert(F_DMA <= 63, "vmcnt is a 6-bit counter");
static_assert(P_CONS * 64 == 2 * P_MAXW, "7 x 64 halo pixel slots cover the two halo rows of layer 4");

__device__ __forceinline__ void fuse_stage(const LayerArgs& a, unsigned slot_byte, int n, int ty0, int c,
                                           size_t HW, unsigned voff, const float* zero) {
    const unsigned long long zaddr = (unsigned long long)zero;
    const bool interior = ty0 >= 2 && ty0 + PT_H + 2 <= a.H;          // rows ty0-2 .. ty0+9 all inside
#pragma unroll
    for (int cc = 0; cc < F_LCH; ++cc) {
        const int ch = c * F_LCH + cc;
        const long s1 = ch >= 2, s2 = ch >= NIN;                   // mv | res | feat
        const unsigned long long base = (unsigned long long)a.mv + s1 * ((unsigned long long)a.res - (unsigned long long)a.mv) +
                                        s2 * ((unsigned long long)a.feat - (unsigned long long)a.res);
        const long pidx = (long)n * (2 + s1 + s2 * (NFEAT - 3)) + ch - s1 * 2 - s2 * (NIN - 2);
        const bool chok = ch < F_CIN;
        const unsigned long long row0 = base + (unsigned long long)((pidx * (long)HW + (long)(ty0 - 2) * a.W) * 4);
        unsigned long long sptr = chok ? row0 : zaddr;
        const unsigned long long sstride = chok ? (unsigned long long)a.W * 4 : 0;
        const unsigned dst = slot_byte + (unsigned)(cc * F_ROWS) * (LTW * 4);
        if (interior) {
#pragma unroll
            for (int row = 0; row < F_ROWS; ++row) {
                dma_row16_s(sptr, voff, dst + row * (LTW * 4));
                sptr += sstride;
            }
        } else {
#pragma unroll
            for (int row = 0; row < F_ROWS; ++row) {
                const int yy = ty0 - 2 + row;
                dma_row16_s((yy >= 0 && yy < a.H) ? sptr : zaddr, voff, dst + row * (LTW * 4));
                sptr += sstride;
            }
        }
    }
}

// horizontal-tap combine of one layer's accumulators (rows BASE + dx*2 + co of NTL row tiles) over a run
// of NS consecutive 64-pixel segments: out[s][co] = P1[p] + P0[p-1] + P2[p+1] with neighbour lanes /
// segments / waves, zero at the image borders.  push_edges publishes the run's two edge partial
// sums for the neighbour waves; a barrier must separate it from push_combine2.
template <int NS, int NTL, int BASE>
__device__ __forceinline__ void push_edges(const f32x4 (&acc)[NS][NTL], float* xchg, int r, int lane) {
#pragma unroll
    for (int co = 0; co < 2; ++co) {
        const int r0 = BASE + co, r2 = BASE + 4 + co;
        if (lane == 0) xchg[(r * 2 + 0) * 2 + co] = acc[0][r2 / 4][r2 % 4];
        if (lane == 63) xchg[(r * 2 + 1) * 2 + co] = acc[NS - 1][r0 / 4][r0 % 4];
    }
}
template <int NS, int NTL, int BASE>
__device__ __forceinline__ void push_combine2(const f32x4 (&acc)[NS][NTL], float (&out)[NS][2], const float* xchg, int r,
                                              const bool (&at_left)[NS], const bool (&at_right)[NS]) {
    float nb_l[2], nb_r[2];
#pragma unroll
    for (int co = 0; co < 2; ++co) {
        nb_l[co] = r > 0 ? xchg[((r - 1) * 2 + 1) * 2 + co] : 0.f;
        nb_r[co] = r + 1 < P_CONS ? xchg[((r + 1) * 2 + 0) * 2 + co] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int co = 0; co < 2; ++co) {
            const int r0 = BASE + co, r1 = BASE + 2 + co, r2 = BASE + 4 + co;
            const float p0 = acc[s][r0 / 4][r0 % 4], p1 = acc[s][r1 / 4][r1 % 4], p2 = acc[s][r2 / 4][r2 % 4];
            const float edge_l = s > 0 ? lane_bcast(acc[s > 0 ? s - 1 : 0][r0 / 4][r0 % 4], 63) : nb_l[co];
            const float edge_r = s + 1 < NS ? lane_bcast(acc[s + 1 < NS ? s + 1 : s][r2 / 4][r2 % 4], 0) : nb_r[co];
            const float sh_l = dpp_shr_fill(p0, edge_l), sh_r = dpp_shl_fill(p2, edge_r);
            out[s][co] = p1 + (at_left[s] ? 0.f : sh_l) + (at_right[s] ? 0.f : sh_r);
        }
}

__global__ __launch_bounds__(LTHREADS, 2) void gen_l45_kernel(RingArgs ra) {
    __shared__ __attribute__((aligned(16))) float lds[RING * F_BUF + F_Y4 + F_WL + 3 * P_CONS * 4];
    const LayerArgs& a = ra.a;
    float* y4t = lds + RING * F_BUF;
    float* wl = y4t + F_Y4;
    float* xchg4 = wl + F_WL;
    float* xchg5 = xchg4 + P_CONS * 4;
    const size_t HW = (size_t)a.H * a.W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave: 0..6 = consumers, 7 = producer
    const float* zero = a.pk + PACKED_TOTAL;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;

    const int nwg = gridDim.x;
    const int t_begin = nwg % 8 == 0 ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int t_step = nwg;
    const int nitems = t_begin < ra.ntiles ? (ra.ntiles - t_begin + nwg - 1) / nwg * F_NCHUNK : 0;
    if (nitems == 0) return;

    // weights -> LDS as [ci][dy][i][4]: row rho = 4 t + i; rho 0..5 = layer 4 (dx*2 + co), rho 6..11 =
    // layer 5: twelve rows fill three row tiles exactly (separately each layer pads 6 rows to 8)
    for (int e = tid; e < F_WL; e += LTHREADS) {
        const int t = e % 4, i = (e / 4) % 4, dy = (e / 16) % 3, ci = e / 48;
        const int rho = 4 * t + i;
        float v = 0.f;
        if (rho < 6) {
            if (ci < cin_of(4)) v = a.pk[wf_off(4) + (ci * 9 + dy * 3 + rho / 2) * 2 + rho % 2];
        } else if (rho < 12) {
            if (ci < cin_of(5)) v = a.pk[wf_off(5) + (ci * 9 + dy * 3 + (rho - 6) / 2) * 2 + (rho - 6) % 2];
        }
        wl[e] = v;
    }
    __syncthreads();

    if (r == P_CONS) {
        // ------------------------------ producer wave ------------------------------
        if (4 * lane >= a.W) return;
        const unsigned voff = (unsigned)lane * 16;
#pragma unroll 1
        for (int pre = 0; pre < 2 && pre < nitems; ++pre) {
            const int tile = t_begin + (pre / F_NCHUNK) * t_step, n = tile / ra.tiles_y;
            fuse_stage(a, lds0 + pre * (F_BUF * 4), n, (tile - n * ra.tiles_y) * PT_H, pre % F_NCHUNK, HW, voff, zero);
        }
        int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
        for (int q = 0; q < nitems; ++q) {
            if (q + 1 < nitems) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(F_DMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (q + 2 < nitems) {
                const int c2 = c + 2, tile2 = tile + (c2 / F_NCHUNK) * t_step, n2 = tile2 / ra.tiles_y;
                int slot2 = slot + 2; slot2 = slot2 >= RING ? slot2 - RING : slot2;
                fuse_stage(a, lds0 + (unsigned)slot2 * (F_BUF * 4), n2, (tile2 - n2 * ra.tiles_y) * PT_H,
                           c2 % F_NCHUNK, HW, voff, zero);
            }
            slot = slot + 1 == RING ? 0 : slot + 1;
            if (++c == F_NCHUNK) {
                c = 0; tile += t_step;
                asm volatile("s_barrier\n\ts_barrier\n\ts_barrier" ::: "memory");   // edge exchange 4, y4 tile ready, edge exchange 5
            }
        }
        return;
    }

    // ------------------------------ consumer waves ------------------------------
    // Wave r owns pixels [256 r, 256 r + 256) of the tile's 8 rows for BOTH layers (4 shared segments:
    // same pixel, same taps -> the B operands are loaded once and feed both accumulator sets) plus,
    // for layer 4 only, 64 of the 2 x W halo pixels (row ty0-1 then row ty0+8): 7 x 64 = 2 x 224.
    int boff[F_S5];                                   // shared segments: staged row above the pixel = yl + 1
    bool lft[F_S5], rgt[F_S5];
#pragma unroll
    for (int s = 0; s < F_S5; ++s) {
        const int p = r * (F_S5 * 64) + s * 64 + lane;
        const int yl = p / a.W, x = p - yl * a.W;
        boff[s] = p < PT_H * a.W ? (yl + 1) * LTW + x : 0;
        lft[s] = x == 0; rgt[s] = x == a.W - 1;
    }
    const int h = r * 64 + lane;                      // halo pixel of layer 4
    const bool h_ok = h < 2 * a.W, h_top = h < a.W;
    const int hx = h_top ? h : h - a.W;
    const int boffe = h_ok ? (h_top ? 0 : (PT_H + 1) * LTW) + hx : 0;     // staged row 0 / 9 is the row above the halo pixel
    const bool lfte[1] = {hx == 0}, rgte[1] = {hx == a.W - 1};
    float* xchg4e = xchg5 + P_CONS * 4;

    f32x4 acc[F_S5][3], acc4e[1][2];                 // shared segments: rows of both layers; halo segment: layer 4
#pragma unroll
    for (int s = 0; s < F_S5; ++s)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc4e[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc4e[0][1] = acc4e[0][0];
    const float bias4[2] = {a.pk[bf_off(4) + 0], a.pk[bf_off(4) + 1]};
    const float bias5[2] = {a.pk[bf_off(5) + 0], a.pk[bf_off(5) + 1]};

    float mse_acc = 0.f;                              // this lane's sum of (out - flow)^2 over the workgroup's tiles
    int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
    for (int q = 0; q < nitems; ++q) {
        asm volatile("s_barrier" ::: "memory");
        {
            const float* buf = lds + slot * F_BUF;
            float w[2][3][3], b[2][3][F_S5 + 1];
            auto load_stage = [&](int cc, int sel) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float4 u = *reinterpret_cast<const float4*>(wl + (((c * F_LCH + cc) * 3 + dy) * 4 + (lane & 3)) * 4);
                    w[sel][dy][0] = u.x; w[sel][dy][1] = u.y; w[sel][dy][2] = u.z;
#pragma unroll
                    for (int s = 0; s < F_S5; ++s) b[sel][dy][s] = buf[cc * F_PLANE + dy * LTW + boff[s]];
                    b[sel][dy][F_S5] = buf[cc * F_PLANE + dy * LTW + boffe];
                }
            };
            load_stage(0, 0);
#pragma unroll
            for (int cc = 0; cc < F_LCH; ++cc) {
                if (cc + 1 < F_LCH) load_stage(cc + 1, (cc + 1) & 1);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                    for (int s = 0; s < F_S5; ++s)
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            acc[s][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[cc & 1][dy][t], b[cc & 1][dy][s], acc[s][t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 2; ++t)      // halo pixels: rows 0..5 (layer 4); rows 6, 7 of tile 1 are ignored
                        acc4e[0][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[cc & 1][dy][t], b[cc & 1][dy][F_S5], acc4e[0][t], 0, 0, 0);
                }
            }
        }
        slot = slot + 1 == RING ? 0 : slot + 1;
        if (++c < F_NCHUNK) continue;
        c = 0;

        const int n = tile / ra.tiles_y, ty0 = (tile - n * ra.tiles_y) * PT_H;
        const int pmax = (a.H - ty0 < PT_H ? a.H - ty0 : PT_H) * a.W;
        const size_t tile_pix = (size_t)ty0 * a.W;
        // the delta-mode mv values of the layer-5 epilogue: one batch of loads, issued now so that
        // their latency hides behind the layer-4 epilogue and the y4 K-steps
        float mvv[F_S5][2], mse_f[F_S5][2];
#pragma unroll
        for (int s = 0; s < F_S5; ++s) {
            const int p = r * (F_S5 * 64) + s * 64 + lane;
            const size_t pp = p < pmax ? tile_pix + p : 0;
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                mvv[s][co] = a.add_mv ? a.mv[((size_t)n * 2 + co) * HW + pp] : 0.f;
                mse_f[s][co] = a.mse_flow ? a.mse_flow[((size_t)n * 2 + co) * HW + pp] : 0.f;
            }
        }
        // ---- layer 4: combine, bias, LeakyReLU; y4 -> HBM (owned rows) and LDS tile (10 rows, 0 outside) ----
        push_edges<F_S5, 3, 0>(acc, xchg4, r, lane);
        push_edges<1, 2, 0>(acc4e, xchg4e, r, lane);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float o4[F_S5][2], o4e[1][2];
        push_combine2<F_S5, 3, 0>(acc, o4, xchg4, r, lft, rgt);
        push_combine2<1, 2, 0>(acc4e, o4e, xchg4e, r, lfte, rgte);
#pragma unroll
        for (int s = 0; s < F_S5; ++s) {
            const int p = r * (F_S5 * 64) + s * 64 + lane;
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                float v = o4[s][co] + bias4[co];
                v = v > 0.f ? v : 0.1f * v;
                v = p < pmax ? v : 0.f;                                    // rows below the image: zero padding for layer 5
                if (p < PT_H * a.W) y4t[co * (F_Y4ROWS * LTW) + boff[s]] = v;    // tile row yl + 1
                if (p < pmax) a.feat_out[((size_t)n * NFEAT + (yoff(4) - NIN) + co) * HW + tile_pix + p] = v;
            }
        }
        {
            const bool inimg = h_top ? ty0 >= 1 : ty0 + PT_H < a.H;
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                float v = o4e[0][co] + bias4[co];
                v = v > 0.f ? v : 0.1f * v;
                if (h_ok) y4t[co * (F_Y4ROWS * LTW) + boffe] = inimg ? v : 0.f;  // tile row 0 / 9
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // y4 tile complete
        // ---- layer 5: the two y4 channels (K = 2 x 3 dy) from the LDS tile ----
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                // rows 6..11 only (tiles 1, 2): layer 4 does not read y4, its rows of this table are zero
                const float4 v = *reinterpret_cast<const float4*>(wl + (((F_CIN + cy) * 3 + dy) * 4 + (lane & 3)) * 4);
#pragma unroll
                for (int s = 0; s < F_S5; ++s) {
                    // pixel row ty0+yl <-> y4-tile row yl+1 (= boff's row); tap dy reads tile row yl + dy
                    const float bb = y4t[cy * (F_Y4ROWS * LTW) + boff[s] - LTW + dy * LTW];
                    acc[s][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(v.y, bb, acc[s][1], 0, 0, 0);
                    acc[s][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(v.z, bb, acc[s][2], 0, 0, 0);
                }
            }
        push_edges<F_S5, 3, 6>(acc, xchg5, r, lane);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float o5[F_S5][2];
        push_combine2<F_S5, 3, 6>(acc, o5, xchg5, r, lft, rgt);
#pragma unroll
        for (int s = 0; s < F_S5; ++s) {
            const int p = r * (F_S5 * 64) + s * 64 + lane;
            if (p < pmax) {
#pragma unroll
                for (int co = 0; co < 2; ++co) {
                    const float v = o5[s][co] + bias5[co] + mvv[s][co];
                    a.out[((size_t)n * 2 + co) * HW + tile_pix + p] = v;
                    if (a.mse_flow) {                                        // wave-uniform
                        const float d = v - mse_f[s][co];
                        mse_acc = fmaf(d, d, mse_acc);
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < F_S5; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc4e[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc4e[0][1] = acc4e[0][0];
        tile += t_step;
    }
    if (a.mse_part) {
        // one partial per consumer wave (no barrier: the producer wave may already have left): fp32 per lane over
        // the ~100 values it produced, fp64 across the lanes, fixed order -> deterministic
        double d = (double)mse_acc;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
        if (lane == 0) a.mse_part[(size_t)blockIdx.x * P_CONS + r] = d;
    }
}

// ------------------------------------------------------------------------------------------
// Gather variant of the matrix-core layer kernel for layers with 8 output channels.
//
// Same 4x4x1 MFMA, same producer / consumer ring, but ALL nine taps are gathered through K:
//   rows = output channel (NT = COUT/4 row tiles), K = (cin, dy, dx), B = in[ci][y+dy-1][x+dx-1].
// For COUT 8 that is exactly the MFMA count of the push form (2 tiles x 9 taps = 6 tiles x 3), but
//   * 32 accumulator registers instead of 96  -> 16 waves per workgroup fit (14 consumers + 2
//     producers), so a tile is 16 rows: 18 staged rows per 16 instead of 10 per 8 (-10 % traffic),
//     and chunks are 2 channels, so 13 / 21 / 22 input channels pad to 14 / 22 / 22, not 16 / 24 / 24;
//   * no push epilogue: no DPP shifts, no cross-wave exchange, no border masks -- the per-tile
//     timeline of the push kernel showed ~8,000 of 30,000 cycles there for COUT 8.
// Staged rows sit at float offset 4 of their 256-float LDS row; columns 3 and 4+W.. stay zero (the
// ring is cleared once, transfers are EXEC-masked to the image width), which is the convolution's
// zero padding in x.  B operands cost three ds_reads per staged value (one per dx): ~2/3 of the LDS
// bandwidth with 8 output channels, too much with 2 or 4 -- those layers keep the push kernel.
// ------------------------------------------------------------------------------------------
constexpr int GT_H = 16;                            // tile rows
constexpr int G_CONS = 14, G_PROD = 2, G_WAVES = G_CONS + G_PROD, G_THREADS = G_WAVES * 64;
constexpr int G_LCH = 2;                            // channels per chunk, one per producer wave
constexpr int G_ROWS = GT_H + 2;
constexpr int G_PLANE = G_ROWS * LTW;               // 4608 floats
constexpr int G_BUF = G_LCH * G_PLANE;              // 9216 floats = 36,864 B per chunk
constexpr int G_DMA = G_ROWS;                       // 18 row transfers per chunk per producer
constexpr int G_XOFF = 4;                           // staged column 0 lives at float 4 of the LDS row
static_assert(G_CONS * M_SEGS * 64 == GT_H * P_MAXW, "14 x 256 pixel slots cover a 16 x 224 tile");

template <int MODE, int K>
struct GatherGeom {
    static constexpr int CIN = MODE == 2 ? gin_of(K) : cin_of(K);
    static constexpr int COUT = cout_of(K);
    static constexpr int NT = (COUT + 3) / 4;
    static constexpr int NCHUNK = (CIN + G_LCH - 1) / G_LCH;
    static constexpr int WL = NCHUNK * G_LCH * 9 * 4 * NT;           // [ci][tap][i][t], zero rows for channels >= CIN
};

// producer pj stages channel pj of chunk c: 18 rows, one 1 KB row per instruction (scalar addressing)
template <int MODE, int K>
__device__ __forceinline__ void gather_stage(const LayerArgs& a, unsigned slot_byte, int n, int ty0, int c,
                                             size_t HW, unsigned voff, const float* zero, int pj) {
    constexpr int CIN = GatherGeom<MODE, K>::CIN;
    const unsigned long long zaddr = (unsigned long long)zero;
    const bool interior = ty0 >= 1 && ty0 + GT_H < a.H;
    const int ch = c * G_LCH + pj;
    unsigned long long base;
    long pidx;
    if (MODE == 2) {
        constexpr int NG = gin_of(K) - 2;
        const long s1 = ch >= NG;
        base = (unsigned long long)a.gbuf + s1 * ((unsigned long long)a.gout - (unsigned long long)a.gbuf);
        pidx = (long)n * (NFEAT - s1 * (NFEAT - 2)) + ch + (1 - s1) * (yoff(K + 1) - NIN) - s1 * NG;
    } else {
        const long s1 = ch >= 2, s2 = ch >= NIN;
        base = (unsigned long long)a.mv + s1 * ((unsigned long long)a.res - (unsigned long long)a.mv) +
               s2 * ((unsigned long long)a.feat - (unsigned long long)a.res);
        pidx = (long)n * (2 + s1 + s2 * (NFEAT - 3)) + ch - s1 * 2 - s2 * (NIN - 2);
    }
    const bool chok = ch < CIN;
    const unsigned long long row0 = base + (unsigned long long)((pidx * (long)HW + (long)(ty0 - 1) * a.W) * 4);
    unsigned long long sptr = chok ? row0 : zaddr;
    const unsigned long long sstride = chok ? (unsigned long long)a.W * 4 : 0;
    const unsigned dst = slot_byte + (unsigned)(pj * G_ROWS) * (LTW * 4) + G_XOFF * 4;
    if (interior) {
#pragma unroll
        for (int row = 0; row < G_ROWS; ++row) {
            dma_row16_s(sptr, voff, dst + row * (LTW * 4));
            sptr += sstride;
        }
    } else {
#pragma unroll
        for (int row = 0; row < G_ROWS; ++row) {
            const int yy = ty0 - 1 + row;
            dma_row16_s((yy >= 0 && yy < a.H) ? sptr : zaddr, voff, dst + row * (LTW * 4));
            sptr += sstride;
        }
    }
}

template <int MODE, int K>
__global__ __launch_bounds__(G_THREADS) void gen_layer_gather_kernel(RingArgs ra) {
    using G = GatherGeom<MODE, K>;
    constexpr int CIN = G::CIN, COUT = G::COUT, NT = G::NT, NCHUNK = G::NCHUNK;
    __shared__ __attribute__((aligned(16))) float lds[RING * G_BUF + G::WL];
    const LayerArgs& a = ra.a;
    float* wl = lds + RING * G_BUF;
    const size_t HW = (size_t)a.H * a.W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave: 0..13 consumers, 14..15 producers
    const float* zero = a.pk + PACKED_TOTAL;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;

    const int nwg = gridDim.x;
    const int t_begin = nwg % 8 == 0 ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int t_step = nwg;
    const int nitems = t_begin < ra.ntiles ? (ra.ntiles - t_begin + nwg - 1) / nwg * NCHUNK : 0;
    if (nitems == 0) return;

    // clear the ring once (guard columns and everything right of the image stay zero for good)
    for (int i = tid; i < RING * G_BUF / 4; i += G_THREADS)
        reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // weights -> LDS as [ci][tap][i][t]: output channel co = 4 t + i
    const float* wbase = a.pk + (MODE == 2 ? wb_off(K) : wf_off(K));
    for (int idx = tid; idx < G::WL; idx += G_THREADS) {
        const int t = idx % NT, i = (idx / NT) % 4, tap = (idx / (4 * NT)) % 9, ci = idx / (36 * NT);
        const int co = 4 * t + i;
        wl[idx] = (co < COUT && ci < CIN) ? wbase[(ci * 9 + tap) * COUT + co] : 0.f;
    }
    __syncthreads();

    if (r >= G_CONS) {
        // ------------------------------ producer waves ------------------------------
        const int pj = r - G_CONS;
        if (4 * lane >= a.W) return;                                   // EXEC = lanes that hold image columns
        const unsigned voff = (unsigned)lane * 16;
#pragma unroll 1
        for (int pre = 0; pre < 2 && pre < nitems; ++pre) {
            const int tile = t_begin + (pre / NCHUNK) * t_step, n = tile / ra.tiles_y;
            gather_stage<MODE, K>(a, lds0 + pre * (G_BUF * 4), n, (tile - n * ra.tiles_y) * GT_H, pre % NCHUNK, HW, voff, zero, pj);
        }
        int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
        for (int q = 0; q < nitems; ++q) {
            if (q + 1 < nitems) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(G_DMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (q + 2 < nitems) {
                const int c2 = c + 2, tile2 = tile + (c2 / NCHUNK) * t_step, n2 = tile2 / ra.tiles_y;
                int slot2 = slot + 2; slot2 = slot2 >= RING ? slot2 - RING : slot2;
                gather_stage<MODE, K>(a, lds0 + (unsigned)slot2 * (G_BUF * 4), n2, (tile2 - n2 * ra.tiles_y) * GT_H,
                                      c2 % NCHUNK, HW, voff, zero, pj);
            }
            slot = slot + 1 == RING ? 0 : slot + 1;
            if (++c == NCHUNK) { c = 0; tile += t_step; }
        }
        return;
    }

    // ------------------------------ consumer waves ------------------------------
    int boff[M_SEGS];                 // LDS offset of the tap (dy = 0, dx = 1) of the lane's pixel
#pragma unroll
    for (int s = 0; s < M_SEGS; ++s) {
        const int p = r * (M_SEGS * 64) + s * 64 + lane;
        const int yl = p / a.W, x = p - yl * a.W;
        boff[s] = p < GT_H * a.W ? yl * LTW + x + G_XOFF : G_XOFF;
    }
    f32x4 acc[M_SEGS][NT];
#pragma unroll
    for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Results of the previous tile wait in registers and are stored a few per chunk while the next
    // tile computes: 14 waves storing 32 values each at once is a 115 KB burst that the memory
    // system absorbs at its ~10 B/clk per-CU share (10,000 of a tile's 56,000 cycles, measured).
    constexpr int NOUT = M_SEGS * COUT, PER = (NOUT + NCHUNK - 1) / NCHUNK;
    float pend[M_SEGS][COUT];
#pragma unroll
    for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
        for (int co = 0; co < COUT; ++co) pend[s][co] = 0.f;
    long pend_base = -1;                                   // element offset of (frame, channel yoff(K), tile pixel 0); -1: nothing pending
    int pend_max = 0;
    float* const outp = MODE == 0 ? a.feat_out : a.gbuf;
    auto flush_group = [&](int grp) {                      // stores values [grp*PER, grp*PER+PER) of the pending tile
#pragma unroll
        for (int gsel = 0; gsel < NCHUNK; ++gsel) {
            if (grp == gsel) {
#pragma unroll
                for (int k = gsel * PER; k < gsel * PER + PER && k < NOUT; ++k) {
                    const int s = k / COUT, co = k % COUT;
                    const int p = r * (M_SEGS * 64) + s * 64 + lane;
                    if (p < pend_max) outp[pend_base + (long)co * (long)HW + p] = pend[s][co];
                }
            }
        }
    };

    int tile = t_begin, c = 0, slot = 0;
#pragma unroll 1
    for (int q = 0; q < nitems; ++q) {
        asm volatile("s_barrier" ::: "memory");
        if (pend_base >= 0) flush_group(c);
        {
            // one chunk = 2 channels x 3 stages (dy) of 3 K-steps (dx) x 4 segments x NT tiles; the
            // operands of the next stage are requested before the MFMAs of the current one
            const float* buf = lds + slot * G_BUF;
            const float* wp = wl + (c * G_LCH) * 36 * NT + (lane & 3) * NT;
            float w[2][3][NT], b[2][3][M_SEGS];
            auto load_stage = [&](int st, int sel) {
                const int cc = st / 3, dy = st % 3;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* p = wp + ((cc * 9 + dy * 3 + dx) * 4) * NT;
                    if (NT == 2) {
                        const float2 v = *reinterpret_cast<const float2*>(p);
                        w[sel][dx][0] = v.x; w[sel][dx][NT - 1] = v.y;
                    } else {
                        w[sel][dx][0] = p[0];
                    }
#pragma unroll
                    for (int s = 0; s < M_SEGS; ++s) b[sel][dx][s] = buf[cc * G_PLANE + dy * LTW + boff[s] + dx - 1];
                }
            };
            load_stage(0, 0);
#pragma unroll
            for (int st = 0; st < G_LCH * 3; ++st) {
                if (st + 1 < G_LCH * 3) load_stage(st + 1, (st + 1) & 1);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[s][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[st & 1][dx][t], b[st & 1][dx][s], acc[s][t], 0, 0, 0);
            }
        }
        slot = slot + 1 == RING ? 0 : slot + 1;
        if (++c < NCHUNK) continue;
        c = 0;

        // ---- epilogue: bias / activation into the pending registers (stored during the next tile) ----
        const int n = tile / ra.tiles_y, ty0 = (tile - n * ra.tiles_y) * GT_H;
        const int pmax = (a.H - ty0 < GT_H ? a.H - ty0 : GT_H) * a.W;
        const long base = ((long)n * NFEAT + (yoff(K) - NIN)) * (long)HW + (long)ty0 * a.W;
        if (MODE == 2) {
            float fv[M_SEGS][COUT];                       // y_K for LeakyReLU': one batch of loads
#pragma unroll
            for (int s = 0; s < M_SEGS; ++s) {
                const int p = r * (M_SEGS * 64) + s * 64 + lane;
                const long pp = p < pmax ? p : 0;
#pragma unroll
                for (int co = 0; co < COUT; ++co) fv[s][co] = a.feat[base + (long)co * (long)HW + pp];
            }
#pragma unroll
            for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
                for (int co = 0; co < COUT; ++co) pend[s][co] = acc[s][co / 4][co % 4] * (fv[s][co] > 0.f ? 1.f : 0.1f);
        } else {
#pragma unroll
            for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float v = acc[s][co / 4][co % 4] + a.pk[bf_off(K) + co];
                    pend[s][co] = v > 0.f ? v : 0.1f * v;
                }
        }
        pend_base = base;
        pend_max = pmax;
#pragma unroll
        for (int s = 0; s < M_SEGS; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[s][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile += t_step;
    }
    if (pend_base >= 0)
        for (int grp = 0; grp < NCHUNK; ++grp) flush_group(grp);
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
// tile image: [33 planes][10 rows][40 floats]; a row holds image columns tx0-4 .. tx0+35 (the tile,
// its halo column on each side and 3+3 unused), i.e. ten 16-byte chunks that are each entirely
// inside or outside the image when W % 4 == 0 -> one 16-byte LDS-DMA lane per chunk
constexpr int WX_PITCH = 40, WX_COL0 = 3;              // LDS column of image column tx0-1
constexpr int WX_PLANE = (WT_H + 2) * WX_PITCH + 4;    // 404: plane % 32 == 20, conflict-light gathers
constexpr int WX_FLOATS = 33 * WX_PLANE;               // 13332
constexpr int WT_LDS = (WX_FLOATS + 63) / 64 * 64;     // 13376 floats; x2 buffers = 107,008 B
constexpr int NT_A = 9, NT_B = 19, NT_ALL = NT_A + NT_B;   // accumulator tiles per wave
constexpr int WPART = NT_ALL * 256;                    // floats per workgroup partial
constexpr int NT3_A = 10, NT3_B = 21, WPART3 = (NT3_A + NT3_B) * 256;   // layout of the shared-window bf16x3 kernel (below)
constexpr int WPART_MAX = WPART3;
static_assert(WR_WPART == WPART3 && WR_NA == NT3_A, "gen_wgrad.hip writes the shared-window partial size");
constexpr int WGRAD_MAX_GROUPS = 256;
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

// Producer side: one wave-instruction of LDS-DMA (global_load_lds_dword) fills 64 consecutive
// floats of the tile image; every lane supplies its own source address (or the address of a zero
// word for pixels outside the image and for padding), so the padded / halo'd layout costs nothing
// and no VGPRs are tied up by data in flight.
// LDS-DMA of one tile (W % 4 == 0): one dwordx4 instruction moves half a plane (5 rows x 10
// chunks = 50 lanes x 16 B), so the plane base is wave-uniform and a lane only contributes its
// fixed (row, chunk): 66 instructions per tile, ~8 per wave, ~10 vector integer ops each.
// (Issue cost matters: ~100 cycles per LDS-DMA instruction, and decoding a flat 64-float chunk per
// lane cost ~25 vector ops each and held the MFMA rate at 70 % -- tools/ubench/mfma_f32_issue.hip.)
__device__ __forceinline__ void wgrad_dma_tile(const WgradArgs& a, float* buf, int tile,
                                               int per_frame, size_t HW, int pw, int lane,
                                               const float* zero) {
    const int n = tile / per_frame, r0 = tile - n * per_frame;
    const int ty0 = (r0 / a.tiles_x) * WT_H, tx0 = (r0 % a.tiles_x) * WT_W;
    const float* mvp = a.mv + (size_t)n * 2 * HW;
    const float* rsp = a.res + (size_t)n * 3 * HW;
    const float* ftp = a.feat + (size_t)n * NFEAT * HW;
    const int lrow = lane / 10, chunk = lane - lrow * 10;      // lanes 0..49
    const int xx = tx0 - 4 + 4 * chunk;
    const bool colok = xx >= 0 && xx < a.W;
    if (lane < 50) {
#pragma unroll 1
        for (int hp = pw; hp < 66; hp += 8) {                   // half-planes, wave-uniform
            const int plane = hp >> 1, row = (hp & 1) * 5 + lrow;
            const int yy = ty0 - 1 + row;
            const float* base = plane < 2 ? mvp + (size_t)plane * HW
                              : plane < NIN ? rsp + (size_t)(plane - 2) * HW
                                            : ftp + (size_t)(plane - NIN) * HW;
            const float* src = (colok && yy >= 0 && yy < a.H) ? base + (size_t)yy * a.W + xx : zero;
            dma_row16((unsigned long long)src,
                      (unsigned)(size_t)(lptr_t)(buf + plane * WX_PLANE + (hp & 1) * 5 * WX_PITCH));
        }
    }
}

// Any W: dword granularity, one instruction per (plane, row), 34 lanes.
__device__ __forceinline__ void wgrad_dma_tile_generic(const WgradArgs& a, float* buf, int tile,
                                                       int per_frame, size_t HW, int pw, int lane,
                                                       const float* zero) {
    const int n = tile / per_frame, r0 = tile - n * per_frame;
    const int ty0 = (r0 / a.tiles_x) * WT_H, tx0 = (r0 % a.tiles_x) * WT_W;
    const int xx = tx0 - 1 + lane;
    const bool colok = lane < WT_W + 2 && xx >= 0 && xx < a.W;
    if (lane < WT_W + 2) {
#pragma unroll 1
        for (int rr = pw; rr < 33 * (WT_H + 2); rr += 8) {
            const int plane = rr / (WT_H + 2), row = rr - plane * (WT_H + 2);
            const int yy = ty0 - 1 + row;
            const float* base = in_plane(a.mv, a.res, a.feat, n, plane, HW);
            const float* src = (colok && yy >= 0 && yy < a.H) ? base + (size_t)yy * a.W + xx : zero;
            dma_row4((unsigned long long)src,
                     (unsigned)(size_t)(lptr_t)(buf + plane * WX_PLANE + row * WX_PITCH + WX_COL0));
        }
    }
}

// 512 threads per workgroup, one workgroup per CU, two LDS buffers.  Per tile every wave first
// issues its share of the LDS-DMA for the NEXT tile (asynchronous, no VGPRs), then runs the MFMAs
// of one row (8 groups of 4 pixels) of the CURRENT tile; two MFMA waves per SIMD cover each
// other's LDS gathers (one wave per SIMD measured 50 % matrix-pipe utilisation).
template <bool VEC4>
__global__ __launch_bounds__(512, 2) void gen_bwd_weight_kernel(WgradArgs a, const float* zero) {
    __shared__ __attribute__((aligned(16))) float lds2[2 * WT_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15, kq = lane >> 4;          // column / pixel-in-group of this lane
    const bool ones = (16 * 18 + j) == BIAS_COL;      // bias column lives in N tile 18
    const size_t HW = (size_t)a.H * a.W;
    // A fragments (gradient rows) come straight from global memory: lane (j, kq) needs row j of
    // each M tile at pixel kq of the group -- 2 loads per 28 MFMAs, requested one tile ahead.
    //   tile A rows 0..15 = gbuf channels 0..15 (g0, g1)
    //   tile B rows 0..11 = gbuf channels 16..27 (g2, g3, g4), rows 12,13 = grad_out, 14,15 = 0
    const bool rowB = j < 14;
    const size_t chanA = (size_t)j * HW;
    const size_t chanB = (j < 12 ? (size_t)(16 + j) : j < 14 ? (size_t)(j - 12) : (size_t)0) * HW;

    f32x4 accA[NT_A], accB[NT_B];
#pragma unroll
    for (int t = 0; t < NT_A; ++t) accA[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT_B; ++t) accB[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int per_frame = a.tiles_x * a.tiles_y;
    const int ntiles = a.N * per_frame;
    // per-lane gather offsets of the B fragments (x[(ci,tap)][pixel kq of the group])
    int offB[NT_B];
#pragma unroll
    for (int t = 0; t < NT_B; ++t) {
        int nn = 16 * t + j;
        nn = nn < 297 ? nn : 296;
        const int ci = nn / 9, tap = nn - ci * 9;
        offB[t] = ci * WX_PLANE + (tap / 3) * WX_PITCH + (tap % 3) + kq + WX_COL0;
    }
    float cur0[8], cur1[8], nxt0[8], nxt1[8];
    auto request_row = [&](int tile, float (&v0)[8], float (&v1)[8]) {
        const int n = tile / per_frame, r0 = tile - n * per_frame;
        const int ty0 = (r0 / a.tiles_x) * WT_H, tx0 = (r0 % a.tiles_x) * WT_W;
        const float* gA = a.gbuf + (size_t)n * NFEAT * HW + chanA;
        const float* gB = (j < 12 ? a.gbuf + (size_t)n * NFEAT * HW : a.gout + (size_t)n * 2 * HW) + chanB;
        int yy = ty0 + wave;
        yy = yy < a.H ? yy : a.H - 1;                 // clamped: masked to zero when used
        const size_t rowoff = (size_t)yy * a.W;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            int xx = tx0 + g * 4 + kq;
            xx = xx < a.W ? xx : a.W - 1;
            v0[g] = gA[rowoff + xx];
            v1[g] = gB[rowoff + xx];
        }
    };
    int it = 0;
    if ((int)blockIdx.x < ntiles) {
        if (VEC4) wgrad_dma_tile(a, lds2, blockIdx.x, per_frame, HW, wave, lane, zero);
        else wgrad_dma_tile_generic(a, lds2, blockIdx.x, per_frame, HW, wave, lane, zero);
        request_row(blockIdx.x, cur0, cur1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the inline-asm LDS-DMA is invisible to the compiler
    __syncthreads();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const float* lds = lds2 + (it & 1) * WT_LDS;
        const bool more = tile + (int)gridDim.x < ntiles;
        if (more) {
            if (VEC4) wgrad_dma_tile(a, lds2 + ((it + 1) & 1) * WT_LDS, tile + gridDim.x, per_frame, HW, wave, lane, zero);
            else wgrad_dma_tile_generic(a, lds2 + ((it + 1) & 1) * WT_LDS, tile + gridDim.x, per_frame, HW, wave, lane, zero);
            request_row(tile + gridDim.x, nxt0, nxt1);
        }
        const int r0 = tile % per_frame;
        const int ty0 = (r0 / a.tiles_x) * WT_H, tx0 = (r0 % a.tiles_x) * WT_W;
        const bool rowok = ty0 + wave < a.H;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int c0 = g * 4;
            const bool ok = rowok && (tx0 + c0 + kq < a.W);
            const float a0 = ok ? cur0[g] : 0.f;
            const float a1 = (ok && rowB) ? cur1[g] : 0.f;
            const int xb = wave * WX_PITCH + c0;
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // DMA of the next tile has landed
#pragma unroll
        for (int g = 0; g < 8; ++g) { cur0[g] = nxt0[g]; cur1[g] = nxt1[g]; }
    }
    // cross-wave reduction in LDS, fixed order (wave 0 stores, waves 1..7 add in turn)
    float* lds = lds2;
    for (int w = 0; w < 8; ++w) {
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
    for (int i = threadIdx.x; i < WPART; i += 512) part[i] = lds[i];
}

// ------------------------------------------------------------------------------------------
// Producer / consumer version of the weight-gradient kernel (W % 4 == 0).
//
// Timeline of the kernel above, per tile and wave (s_memtime): ~15,000 cycles blocked issuing its
// share of the LDS-DMA and the 16 global gathers of its gradient rows (194 scattered vector-memory
// instructions per tile and CU), THEN 8,200 cycles of MFMA -- staging and matrix time added up.
// Here wave 7 is the only wave that touches global memory: it stages, with LDS-DMA and scalar
// address arithmetic, the 33 input planes (9 rows x 40 columns) AND the 30 gradient planes
// (7 rows x 32) of the next tile while waves 0..6 run the MFMAs of one tile row each, with both
// operands coming from LDS.  96 row-contiguous transfers per tile instead of 194 scattered ones.
// ------------------------------------------------------------------------------------------
constexpr int PW_H = 7;                                  // tile rows = consumer waves
constexpr int PW_XROWS = PW_H + 2;
constexpr int PW_XPLANE = 372;                           // 9 x 40 = 360 floats + pad (372 % 32 == 20: conflict-light gathers)
constexpr int PW_GPLANE = 228;                           // 7 x 32 = 224 floats + pad (A-fragment reads 2-way)
constexpr int PW_X = 33 * PW_XPLANE;                     // 12,276
constexpr int PW_G = 30 * PW_GPLANE;                     // 6,840
constexpr int PW_BUF = (PW_X + PW_G + 3) / 4 * 4;        // 19,116 floats = 76,464 B; x2 = 152,928 B
static_assert(2 * PW_BUF * 4 <= 160 * 1024, "LDS");
static_assert(WPART <= 2 * PW_BUF, "cross-wave reduction reuses the tile buffers");

__device__ __forceinline__ void dma16_s(unsigned long long sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}
// producer: stage one tile.  Lane L of instruction h (L = 64 h + lane) moves 16-byte chunk
// (row L / 10, chunk L % 10) of an input plane; lane (row, chunk) = (lane / 8, lane % 8) of a gradient plane.
// do_x: stage the 33 input planes; gradient planes [gp0, gp1) are staged by the caller too (the bf16x3 consumers take the
// 30 gradient planes off the producer wave, 4-5 each: with 2.67x fewer matrix cycles the single producer had become the
// kernel's critical path)
__device__ __forceinline__ void wgrad_stage(const WgradArgs& a, float* buf, unsigned buf_byte, int tile, int per_frame,
                                            size_t HW, int lane, bool do_x = true, int gp0 = 0, int gp1 = 30) {
    const int n = tile / per_frame, r0 = tile - n * per_frame;
    const int ty0 = (r0 / a.tiles_x) * PW_H, tx0 = (r0 % a.tiles_x) * WT_W;
    const int W = a.W;
    // ---- input planes: rows ty0-1 .. ty0+7, columns tx0-4 .. tx0+35 ----
    const bool x_interior = ty0 >= 1 && ty0 + PW_H < a.H && tx0 >= 4 && tx0 + WT_W + 4 <= W;
    const int L0 = lane, L1 = 64 + lane;
    const int row0 = L0 / 10, ch0 = L0 - row0 * 10, row1 = L1 / 10, ch1 = L1 - row1 * 10;
    const unsigned voff0 = (unsigned)(row0 * W + ch0 * 4) * 4, voff1 = (unsigned)(row1 * W + ch1 * 4) * 4;
    const long xorg = ((long)(ty0 - 1) * W + (tx0 - 4)) * 4;          // byte offset of (row ty0-1, col tx0-4) in a plane
    bool ok0 = false, ok1 = false;
    if (!x_interior) {
        const int y0 = ty0 - 1 + row0, x0 = tx0 - 4 + ch0 * 4, y1 = ty0 - 1 + row1, x1 = tx0 - 4 + ch1 * 4;
        ok0 = y0 >= 0 && y0 < a.H && x0 >= 0 && x0 < W;
        ok1 = y1 >= 0 && y1 < a.H && x1 >= 0 && x1 < W;
    }
    if (do_x)
#pragma unroll
    for (int plane = 0; plane < 33; ++plane) {
        const float* pb = plane < 2 ? a.mv + ((size_t)n * 2 + plane) * HW
                        : plane < NIN ? a.res + ((size_t)n * 3 + (plane - 2)) * HW
                                      : a.feat + ((size_t)n * NFEAT + (plane - NIN)) * HW;
        const unsigned long long sbase = (unsigned long long)pb + (unsigned long long)xorg;
        const unsigned dst = buf_byte + (unsigned)(plane * PW_XPLANE) * 4;
        if (x_interior) {
            dma16_s(sbase, voff0, dst);
            if (lane < PW_XROWS * 10 - 64) dma16_s(sbase, voff1, dst + 1024);
        } else {
            // edge tile: lanes whose chunk lies outside the image do not transfer; they zero their chunk
            float4* l0 = reinterpret_cast<float4*>(buf + plane * PW_XPLANE) + lane;
            if (ok0) dma16_s(sbase, voff0, dst); else *l0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < PW_XROWS * 10 - 64) {
                if (ok1) dma16_s(sbase, voff1, dst + 1024); else l0[64] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    // ---- gradient planes: rows ty0 .. ty0+6, columns tx0 .. tx0+31 (zero outside the image) ----
    const bool g_interior = ty0 + PW_H <= a.H && tx0 + WT_W <= W;
    const int grow = lane >> 3, gch = lane & 7;
    const unsigned gvoff = (unsigned)(grow * W + gch * 4) * 4;
    const bool gok = ty0 + grow < a.H && tx0 + gch * 4 < W;
    const long gorg = ((long)ty0 * W + tx0) * 4;
    if (lane < PW_H * 8) {
#pragma unroll 1
        for (int gp = gp0; gp < gp1; ++gp) {
            const float* pb = gp < NFEAT ? a.gbuf + ((size_t)n * NFEAT + gp) * HW
                                         : a.gout + ((size_t)n * 2 + (gp - NFEAT)) * HW;
            const unsigned long long sbase = (unsigned long long)pb + (unsigned long long)gorg;
            const unsigned dst = buf_byte + (unsigned)(PW_X + gp * PW_GPLANE) * 4;
            if (g_interior) dma16_s(sbase, gvoff, dst);
            else if (gok) dma16_s(sbase, gvoff, dst);
            else reinterpret_cast<float4*>(buf + PW_X + gp * PW_GPLANE)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// bf16x3 helpers (see conv_nhwc.hip: an fp32 value = the exact sum of three bf16 slices; six slice products per
// fp32 product on the bf16 matrix cores, fp32 accumulate, error <= one fp32 rounding of the product)
typedef __bf16 gen_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gen_u32x4 __attribute__((ext_vector_type(4)));
struct GenSplit3 { gen_u32x4 s[3]; };
__device__ __forceinline__ GenSplit3 gen_split_bf16x3(const float (&v)[8]) {
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        u0[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u0[e] & 0xffff0000u);
        u1[e] = __float_as_uint(r1);
        u2[e] = __float_as_uint(r1 - __uint_as_float(u1[e] & 0xffff0000u));
    }
    GenSplit3 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r.s[0][e] = __builtin_amdgcn_perm(u0[2 * e + 1], u0[2 * e], 0x07060302u);
        r.s[1][e] = __builtin_amdgcn_perm(u1[2 * e + 1], u1[2 * e], 0x07060302u);
        r.s[2][e] = __builtin_amdgcn_perm(u2[2 * e + 1], u2[2 * e], 0x07060302u);
    }
    return r;
}
__device__ __forceinline__ f32x4 gen_mfma_x3(const GenSplit3& a, const GenSplit3& b, f32x4 c) {
    // small terms first: (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
    auto mm = [&](int i, int j, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gen_bf16x8, a.s[i]), __builtin_bit_cast(gen_bf16x8, b.s[j]), acc, 0, 0, 0);
    };
    c = mm(0, 2, c); c = mm(2, 0, c); c = mm(1, 1, c); c = mm(0, 1, c); c = mm(1, 0, c); c = mm(0, 0, c);
    return c;
}

// X3 = true: the same GEMM in bf16x3 arithmetic -- one v_mfma_f32_16x16x32_bf16 k-block = the 32 pixels of a tile row
// (lane (j, kq) holds pixels 8 kq .. 8 kq + 7 of its gradient plane / its (ci, tap) column: eight consecutive floats of an
// LDS row), six MFMAs of 16 cycles per accumulator tile and row where the fp32 form issues eight of 32: 2.67x fewer
// matrix cycles; the split of a fragment (44 VALU instructions) is shared by the two row tiles.  Same C layout, same
// reductions.
// X3 = 2: the bf16x3 form with the three horizontal taps of a (ci, dy) pair in ONE lane: column tile tt = 3 gt + dx holds
// columns g = 16 gt + j = 3 ci + dy; the lane reads the 10 floats x[8 kq - 1 .. 8 kq + 8] of its row once, splits them once
// (40 VALU instructions) and forms the three 8-pixel fragments from shared packed pairs (dx = 0: P0..P3, dx = 2: P1..P4,
// dx = 1: Q0..Q3) -- 67 VALU instructions per three column tiles where X3 = 1 spends 132.  7 x 3 = 21 column tiles for
// the rows of layers 2-5 (+ the ones column for their bias at g = 99, dx = 0), 3 x 3 for those of layers 0 / 1.
template <int X3>
__global__ __launch_bounds__(512, 2) void gen_bwd_weight_pc_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds2[2 * PW_BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 15, kq = lane >> 4;          // column / pixel-in-group of this lane
    const bool ones = (16 * 18 + j) == BIAS_COL;      // bias column lives in N tile 18
    const size_t HW = (size_t)a.H * a.W;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds2;
    const int per_frame = a.tiles_x * a.tiles_y;
    const int ntiles = a.N * per_frame;
    const bool producer = wave == PW_H;
    // XCD-aware tile schedule (workgroup b runs on XCD b % 8): in round i the 32 workgroups of an XCD take 32 CONSECUTIVE
    // tiles, so the halo columns / rows that neighbouring tiles share (a 9 x 40 patch per 7 x 32 tile, 160-byte row
    // segments straddling 128-byte lines) hit in that XCD's L2 instead of being fetched from HBM by eight different L2s
    const bool xcd_sched = gridDim.x % 8 == 0;
    const int xs_per = (int)gridDim.x / 8;
    auto tile_of = [&](int i) -> int {
        return xcd_sched ? (i * 8 + (int)(blockIdx.x & 7)) * xs_per + (int)(blockIdx.x >> 3) : (int)blockIdx.x + i * (int)gridDim.x;
    };

    constexpr int NA = X3 >= 2 ? NT3_A : NT_A, NB = X3 >= 2 ? NT3_B : NT_B, WP = (NA + NB) * 256;
    f32x4 accA[NA], accB[NB];
#pragma unroll
    for (int t = 0; t < NA; ++t) accA[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NB; ++t) accB[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (producer) {
        int it = 0;
        if (tile_of(0) < ntiles) wgrad_stage(a, lds2, lds0, tile_of(0), per_frame, HW, lane);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int tile = tile_of(0); tile < ntiles; tile = tile_of(++it)) {
            const int next = tile_of(it + 1);
            if (next < ntiles)
                wgrad_stage(a, lds2 + ((it + 1) & 1) * PW_BUF, lds0 + (unsigned)(((it + 1) & 1) * PW_BUF) * 4, next, per_frame, HW, lane,
                            true, 0, X3 != 0 ? 0 : 30);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        // per-lane gather offsets of the B fragments (x[(ci,tap)][pixel kq of the group])
        int offB[NT_B];
#pragma unroll
        for (int t = 0; t < NT_B; ++t) {
            int nn = 16 * t + j;
            nn = nn < 297 ? nn : 296;
            const int ci = nn / 9, tap = nn - ci * 9;
            offB[t] = ci * PW_XPLANE + (tap / 3) * WX_PITCH + (tap % 3) + kq + WX_COL0;
        }
        int offW[7];                                       // X3 == 2: start of this lane's 10-float window per group tile
#pragma unroll
        for (int gt = 0; gt < 7; ++gt) {
            int g = 16 * gt + j;
            g = g < 99 ? g : 98;
            offW[gt] = (g / 3) * PW_XPLANE + (g % 3) * WX_PITCH + WX_COL0 + 8 * kq;
        }
        const bool ones3 = j == 3;                         // X3 == 2: g = 99 (group tile 6, column 3) is the ones column
        // A fragments: row j of M tile A = gradient plane j; of M tile B = plane 16 + j (j < 14), else zero
        const bool rowB = j < 14;
        const int offA0 = PW_X + j * PW_GPLANE + kq;
        const int offA1 = PW_X + (rowB ? 16 + j : 0) * PW_GPLANE + kq;
        // The bias gradient of layers 0 and 1 (M tile A) is a plain row sum of the A operand: it is
        // accumulated on the vector ALU (one add per group) instead of spending a 9th MFMA per group
        // on a column of ones; tile B's bias column shares N tile 18 with real columns and stays.
        float bias_a = 0.f;
        int it = 0;
        asm volatile("s_barrier" ::: "memory");
        for (int tile = tile_of(0); tile < ntiles; tile = tile_of(++it)) {
            const float* lds = lds2 + (it & 1) * PW_BUF;
            if constexpr (X3 != 0) {
                const int next = tile_of(it + 1);
                if (next < ntiles)                        // this wave's share of the next tile's gradient planes
                    wgrad_stage(a, lds2 + ((it + 1) & 1) * PW_BUF, lds0 + (unsigned)(((it + 1) & 1) * PW_BUF) * 4, next, per_frame, HW, lane,
                                false, (30 * wave) / PW_H, (30 * (wave + 1)) / PW_H);
                // fragments: 8 consecutive pixels 8 kq .. 8 kq + 7 of this wave's tile row (offA / offB carry + kq: + 7 kq more)
                const int sA = wave * WT_W + 7 * kq, sB = wave * WX_PITCH + 7 * kq;
                float av[8];
                const float4 p0 = *reinterpret_cast<const float4*>(lds + offA0 + sA), p1 = *reinterpret_cast<const float4*>(lds + offA0 + sA + 4);
                av[0] = p0.x; av[1] = p0.y; av[2] = p0.z; av[3] = p0.w; av[4] = p1.x; av[5] = p1.y; av[6] = p1.z; av[7] = p1.w;
#pragma unroll
                for (int e = 0; e < 8; ++e) bias_a += av[e];
                const GenSplit3 a0s = gen_split_bf16x3(av);
                const float4 q0 = *reinterpret_cast<const float4*>(lds + offA1 + sA), q1 = *reinterpret_cast<const float4*>(lds + offA1 + sA + 4);
                av[0] = q0.x; av[1] = q0.y; av[2] = q0.z; av[3] = q0.w; av[4] = q1.x; av[5] = q1.y; av[6] = q1.z; av[7] = q1.w;
#pragma unroll
                for (int e = 0; e < 8; ++e) av[e] = rowB ? av[e] : 0.f;
                const GenSplit3 a1s = gen_split_bf16x3(av);
                if constexpr (X3 >= 2) {
#pragma unroll
                    for (int gt = 0; gt < 7; ++gt) {
                        float wv[10];
                        if constexpr (X3 == 3) {
                            // the window starts at float 3 (mod 4) of its row: one 8-byte, two 16-byte, one 8-byte read.  Every
                            // lane's address is the same mod 4 floats (rows and planes are 16-byte multiples for the LDS-DMA),
                            // so dword reads find 8 of the 32 banks (4-way conflicts in each half-wave); the wide reads are
                            // serviced in 16-byte slots of the 256-byte bank row: modelled 180 LDS cycles per wave and tile
                            // where the dword reads take 640
                            const float* wp = lds + offW[gt] + wave * WX_PITCH;
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 e0 = *static_cast<const f32x2*>(__builtin_assume_aligned(wp - 1, 8));
                            const f32x4 m0 = *static_cast<const f32x4*>(__builtin_assume_aligned(wp + 1, 16));
                            const f32x4 m1 = *static_cast<const f32x4*>(__builtin_assume_aligned(wp + 5, 16));
                            const f32x2 e9 = *static_cast<const f32x2*>(__builtin_assume_aligned(wp + 9, 8));
                            wv[0] = e0.y; wv[1] = m0.x; wv[2] = m0.y; wv[3] = m0.z; wv[4] = m0.w;
                            wv[5] = m1.x; wv[6] = m1.y; wv[7] = m1.z; wv[8] = m1.w; wv[9] = e9.x;
                        } else {
#pragma unroll
                            for (int e = 0; e < 10; ++e) wv[e] = lds[offW[gt] + wave * WX_PITCH + e];
                        }
                        if (gt == 6 && ones3) {
#pragma unroll
                            for (int e = 0; e < 10; ++e) wv[e] = 1.f;
                        }
                        unsigned u[3][10];
#pragma unroll
                        for (int e = 0; e < 10; ++e) {
                            u[0][e] = __float_as_uint(wv[e]);
                            const float r1 = wv[e] - __uint_as_float(u[0][e] & 0xffff0000u);
                            u[1][e] = __float_as_uint(r1);
                            u[2][e] = __float_as_uint(r1 - __uint_as_float(u[1][e] & 0xffff0000u));
                        }
                        unsigned P[3][5], Q[3][4];
#pragma unroll
                        for (int sl = 0; sl < 3; ++sl) {
#pragma unroll
                            for (int k = 0; k < 5; ++k) P[sl][k] = __builtin_amdgcn_perm(u[sl][2 * k + 1], u[sl][2 * k], 0x07060302u);
#pragma unroll
                            for (int k = 0; k < 4; ++k) Q[sl][k] = __builtin_amdgcn_perm(u[sl][2 * k + 2], u[sl][2 * k + 1], 0x07060302u);
                        }
#pragma unroll
                        for (int dxi = 0; dxi < 3; ++dxi) {
                            GenSplit3 bs;
#pragma unroll
                            for (int sl = 0; sl < 3; ++sl)
#pragma unroll
                                for (int k = 0; k < 4; ++k) bs.s[sl][k] = dxi == 0 ? P[sl][k] : dxi == 2 ? P[sl][k + 1] : Q[sl][k];
                            const int tt = 3 * gt + dxi;
                            if (gt < 3) accA[tt < NA ? tt : 0] = gen_mfma_x3(a0s, bs, accA[tt < NA ? tt : 0]);
                            accB[tt] = gen_mfma_x3(a1s, bs, accB[tt]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NT_B; ++t) {
                        float bv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) bv[e] = lds[offB[t] + sB + e];
                        if (t == 18 && ones) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) bv[e] = 1.f;
                        }
                        const GenSplit3 bs = gen_split_bf16x3(bv);
                        if (t < 8) accA[t] = gen_mfma_x3(a0s, bs, accA[t]);
                        accB[t < NB ? t : 0] = gen_mfma_x3(a1s, bs, accB[t < NB ? t : 0]);
                    }
                }
            } else
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int sA = wave * WT_W + g * 4, sB = wave * WX_PITCH + g * 4;
                const float a0 = lds[offA0 + sA];
                const float a1r = lds[offA1 + sA];
                const float a1 = rowB ? a1r : 0.f;
                float b[NT_B];
#pragma unroll
                for (int t = 0; t < NT_B; ++t) b[t] = lds[offB[t] + sB];
                if (ones) b[18] = 1.f;
                bias_a += a0;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    accA[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t], accA[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT_B; ++t)
                    accB[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t], accB[t], 0, 0, 0);
            }
            if constexpr (X3 != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's transfers have landed
            asm volatile("s_barrier" ::: "memory");        // next tile staged, this buffer may be refilled
        }
        // fold the row sums into accumulator slot 8 (N tile 18), column BIAS_COL - 288 = 9, in the MFMA
        // C layout: lane (j, kq) holds rows kq*4 + q of column j
        float tot = bias_a;
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);                                    // every lane of row j: the row's total
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float rv = __shfl(tot, kq * 4 + q);                  // lanes 0..15 hold rows 0..15
            if constexpr (X3 >= 2) accA[9][q] = ones3 ? rv : 0.f;      // slot 9, column 3
            else accA[8][q] = ones ? rv : 0.f;
        }
    }
    // cross-wave reduction in LDS, fixed order (wave 0 stores, waves 1..6 add in turn)
    float* lds = lds2;
    for (int w = 0; w < PW_H; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NA + NB; ++t) {
                const f32x4 v = t < NA ? accA[t < NA ? t : 0] : accB[t >= NA ? t - NA : 0];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = t * 256 + (kq * 4 + q) * 16 + j;   // C row = kq*4+q, col = j
                    lds[idx] = (w == 0 ? 0.f : lds[idx]) + v[q];
                }
            }
        }
        __syncthreads();
    }
    float* part = a.partials + (size_t)blockIdx.x * WP;
    for (int i = threadIdx.x; i < WP; i += 512) part[i] = lds[i];
}

// Stage 1 of the cross-workgroup reduction: partials [groups][WPART] -> [RED_CHUNKS][WPART],
// coalesced over the WPART axis, fixed summation order.
constexpr int RED_CHUNKS = 16;
__global__ __launch_bounds__(256) void gen_bwd_weight_reduce1_kernel(float* __restrict__ partials,
                                                                     int groups, int wpart) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // < wpart (7168 = 28 * 256, or 31 * 256 for the shared-window layout)
    const int per = (groups + RED_CHUNKS - 1) / RED_CHUNKS;
    const int g0 = blockIdx.y * per, g1 = (g0 + per < groups) ? g0 + per : groups;
    float s = 0.f;
    for (int g = g0; g < g1; ++g) s += partials[(size_t)g * wpart + i];
    // results are written behind the raw partials (the buffer is sized for groups + RED_CHUNKS)
    partials[(size_t)(groups + blockIdx.y) * wpart + i] = s;
}

// Stage 2: [RED_CHUNKS][28 tiles][16][16] -> the 12 gradient tensors in PyTorch layout
__global__ void gen_bwd_weight_reduce_kernel(const float* __restrict__ partials, int groups,
                                             GradPtrs G, int layout3) {
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
    int nt = col >> 4, jj = col & 15;
    int slot, row;
    const int na = layout3 ? NT3_A : NT_A, wpart = layout3 ? WPART3 : WPART;
    if (layout3 == 1) {               // shared-window layout: column tile 3 gt + dx, column g & 15 with g = 3 p + dy; bias: g = 99, dx = 0
        const int g = col == BIAS_COL ? 99 : 3 * p + tap / 3, dxi = col == BIAS_COL ? 0 : tap % 3;
        nt = 3 * (g >> 4) + dxi; jj = g & 15;
    } else if (layout3 == 2) {        // gen_wgrad.hip: g = 33 dy + p; bias: g = 99, dx = 1; tile A holds gt = 0, 2, 4 in slots 3 (gt / 2) + dx
        const int g = col == BIAS_COL ? 99 : 33 * (tap / 3) + p, dxi = col == BIAS_COL ? 1 : tap % 3;
        nt = 3 * (g >> 4) + dxi; jj = g & 15;
        if (k < 2 && col != BIAS_COL) nt = 3 * (g >> 5) + dxi;
    }
    if (k < 2) {                      // tile A: g0 rows 0..7, g1 rows 8..15
        row = k * 8 + co;
        slot = layout3 ? (col == BIAS_COL ? 9 : nt) : (nt == 18 ? 8 : nt);
    } else {                          // tile B: g2 0..5, g3 6..9, g4 10..11, g5 12..13
        row = (k == 2 ? 0 : k == 3 ? 6 : k == 4 ? 10 : 12) + co;
        slot = na + nt;
    }
    const size_t off = (size_t)slot * 256 + row * 16 + jj;
    float s = 0.f;
    for (int c = 0; c < RED_CHUNKS; ++c) s += partials[(size_t)(groups + c) * wpart + off];
    if (i < WF_TOTAL)
        G.w[k][(co * cin_of(k) + logical_of(k, p)) * 9 + tap] = s;
    else
        G.b[k][co] = s;
}

// loss = sum of the fused kernel's per-wave partials / numel (fixed order)
__global__ __launch_bounds__(256) void gen_mse_final_kernel(const double* __restrict__ part, int n,
                                                            float* __restrict__ loss_out, double numel) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss_out = (float)((sm[0] + sm[1] + sm[2] + sm[3]) / numel);
}

int wgrad_groups(int N, int H, int W) {
    const long tiles = (long)N * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    return (int)(tiles < WGRAD_MAX_GROUPS ? tiles : WGRAD_MAX_GROUPS);
}

// All frames in one pass per layer (processing the frames in Infinity-Cache-sized chunks was
// measured slower: under-filled grids).
int frames_per_pass(int N, int, int) { const int f = option(OPT_GEN_FRAMES); return f > 0 && f < N ? f : N; }

int num_cus_hw() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}
int num_cus() { return persistent_cus(num_cus_hw()); }       // what a persistent grid fills (option grid_reserve_cus)

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
    if constexpr (MODE != 1) {
        // option gen_wino: bit K = forward hidden layer K, bit 8 + K = data-gradient group K on the Winograd kernel
        if (((option(OPT_GEN_WINO) >> (MODE == 0 ? K : 8 + K)) & 1) && wino_shape_ok(a.H, a.W)) {
            RingArgs ra;
            ra.a = a;
            ra.tiles_y = (a.H + PT_H - 1) / PT_H;
            ra.ntiles = ra.tiles_y * N;
            ra.ablate = option(OPT_GEN_ABLATE);
            ra.stagger = option(OPT_GEN_STAGGER);
            const int wgs = ra.ntiles < num_cus() ? ra.ntiles : num_cus();
            const int dbg = ra.ablate >> 8;
            ra.ablate &= 255;
#ifdef DMC_MEASURE
            if constexpr ((MODE == 0 && K == 2) || (MODE == 0 && K == 3)) {
                if (dbg == 1) gen_wino_kernel<MODE, K, 1><<<wgs, LTHREADS, 0, s>>>(ra);
                else if (dbg == 2) gen_wino_kernel<MODE, K, 2><<<wgs, LTHREADS, 0, s>>>(ra);
                else if (dbg == 3) gen_wino_kernel<MODE, K, 3><<<wgs, LTHREADS, 0, s>>>(ra);
                else if (dbg == 4) gen_wino_kernel<MODE, K, 4><<<wgs, LTHREADS, 0, s>>>(ra);
                else gen_wino_kernel<MODE, K><<<wgs, LTHREADS, 0, s>>>(ra);
            } else
#endif
            gen_wino_kernel<MODE, K><<<wgs, LTHREADS, 0, s>>>(ra);
            (void)dbg;
            return check_launch("gen_wino");
        }
    }
    const dim3 grid((a.W + LTW - 1) / LTW, (a.H + LTH - 1) / LTH, N);
    // measured per layer (N=120, 224x224): the LDS-DMA + DPP kernel wins where little arithmetic
    // rides on each staged channel (Cout 2: layers 4, 5; gradient groups 2, 3, 4), the
    // register-pipelined kernel wins for the Cout 8 / 6 / 4 layers
    constexpr bool DMA_WINS = (MODE != 2 && K >= 4) || (MODE == 2 && K >= 2);
    const int path = option(OPT_GEN_LAYER_PATH);
    // measured per layer (N=120, 224x224): the gather form wins where the push kernel pads the
    // input channels most (5 -> 8, 13 -> 16, 14 -> 16: layers 0, 1 and gradient group 1); the
    // Cout-6 layers lose (18 instead of 15 MFMAs per channel and pixel), the rest are at the HBM roof
    constexpr bool GATHER_FORM = (MODE == 0 && K <= 1) || (MODE == 2 && K == 1);
    const int gpath = option(OPT_GEN_GATHER);
    if (GATHER_FORM && gpath == 1 && (path == 1 || path == 3 || path == 4 || path == 5) && a.W % 4 == 0 && a.W <= P_MAXW) {
        RingArgs ra;
        ra.ablate = 0;
        ra.a = a;
        ra.tiles_y = (a.H + GT_H - 1) / GT_H;
        ra.ntiles = ra.tiles_y * N;
        const int wgs = ra.ntiles < num_cus() ? ra.ntiles : num_cus();
        if constexpr (GATHER_FORM) gen_layer_gather_kernel<MODE, K><<<wgs, G_THREADS, 0, s>>>(ra);
    } else if (MODE == 0 && ((path == 1 && K == 2) || (MEASURE_BUILD && path == 4 && (K == 2 || K == 3))) && a.W % 4 == 0 && a.W <= P_MAXW) {
        // variant (iii): the default tile with a two-stage ring, two 8-wave workgroups per CU.  Forward layer 2 (the most
        // matrix-bound layer) gains 7.5 % from the second resident workgroup (214.7 -> 198.5 us, matrix pipe busy 0.534 ->
        // 0.573) and takes it by default; layer 3 (HBM-bound, needs 3-channel chunks to fit) loses 1.7 %: option value 4 only.
        // Option value 5 = every layer on the three-stage single-workgroup kernel (the default before)
        if constexpr (MODE == 0 && (K == 2 || (MEASURE_BUILD && K == 3))) {
            RingArgs ra;
            ra.a = a;
            ra.tiles_y = (a.H + PT_H - 1) / PT_H;
            ra.ntiles = ra.tiles_y * N;
            ra.ablate = option(OPT_GEN_ABLATE);
            const int wgs = ra.ntiles < 2 * num_cus() ? ra.ntiles : 2 * num_cus();
            gen_layer_mfma_kernel<MODE, K, 2><<<wgs, LTHREADS, 0, s>>>(ra);
        }
    } else if (MEASURE_BUILD && path == 3 && MODE == 0 && (K == 2 || K == 3) && a.W % 4 == 0 && a.W <= P_MAXW) {
        // measurement variant (round-2 verdict, variant ii): 4-row tiles, two 8-wave workgroups per CU -- -DDMC_MEASURE build only
        if constexpr (MEASURE_BUILD && MODE == 0 && (K == 2 || K == 3)) {
            RingArgs ra;
            ra.a = a;
            ra.tiles_y = (a.H + RingGeo<1>::ROWS - 1) / RingGeo<1>::ROWS;
            ra.ntiles = ra.tiles_y * N;
            ra.ablate = option(OPT_GEN_ABLATE);
            const int wgs = ra.ntiles < 2 * num_cus() ? ra.ntiles : 2 * num_cus();
            gen_layer_mfma_kernel<MODE, K, 1><<<wgs, LTHREADS, 0, s>>>(ra);
        }
    } else if ((path == 1 || path == 3 || path == 4 || path == 5) && a.W % 4 == 0 && a.W <= P_MAXW) {
        RingArgs ra;
        ra.ablate = 0;
        ra.a = a;
        ra.tiles_y = (a.H + PT_H - 1) / PT_H;
        ra.ntiles = ra.tiles_y * N;
        ra.ablate = option(OPT_GEN_ABLATE);
        const int wgs = ra.ntiles < num_cus() ? ra.ntiles : num_cus();
        gen_layer_mfma_kernel<MODE, K><<<wgs, LTHREADS, 0, s>>>(ra);
    }
    else if (DMA_WINS && a.W % 4 == 0 && a.W <= LTW) gen_layer_dma_kernel<MODE, K><<<grid, LTHREADS, 0, s>>>(a);
    else if (a.W % 4 == 0) gen_layer_kernel<MODE, K, true><<<grid, LTHREADS, 0, s>>>(a);
    else gen_layer_kernel<MODE, K, false><<<grid, LTHREADS, 0, s>>>(a);
    return check_launch("gen_layer");
}

int pack(const float* const* w, const float* const* b, float* pk, hipStream_t s, unsigned short* x3_frags = nullptr) {
    ParamPtrs P;
    for (int k = 0; k < NL; ++k) {
        if (!w[k] || (b && !b[k])) return fail(DMC_E_INVALID, "null weight/bias pointer %d", k);
        P.w[k] = w[k];
        P.b[k] = b ? b[k] : w[k];   // bias slots are unused by the backward pass
    }
    const int threads = PACKED_TOTAL + ZERO_PAD + (x3_frags ? GX_PACK_THREADS : 0);
    pack_params_kernel<<<(threads + 255) / 256, 256, 0, s>>>(P, pk, x3_frags);
    return check_launch("pack_params");
}

}  // namespace

extern "C" {

// the packed fp32 parameters + zero words, then the bf16x3 weight fragments of gen_x3.hip
size_t dmc_gen_tiny_workspace_bytes(void) { return (size_t)(PACKED_TOTAL + ZERO_PAD) * sizeof(float) + gen_x3_frag_bytes(); }

size_t dmc_gen_tiny_saved_bytes(int N, int H, int W) {
    return (size_t)N * NFEAT * H * W * sizeof(float);
}
size_t dmc_gen_tiny_gbuf_bytes(int N, int H, int W) { return dmc_gen_tiny_saved_bytes(N, H, W); }
size_t dmc_gen_tiny_partials_bytes(int N, int H, int W) {
    return (size_t)(wgrad_groups(N, H, W) + RED_CHUNKS) * WPART_MAX * sizeof(float);
}

// flow != null: also reduce sum((out - flow)^2) into mse_part (the fused kernel's epilogue); *fused_wgs receives
// the number of partials written, 0 if the shape took a path without the fused epilogue
static int gen_tiny_fwd_impl(const float* mv, const float* res, const float* const* w,
                             const float* const* b, float* out, float* saved, float* workspace, int N,
                             int H, int W, int add_mv_delta, const float* flow, double* mse_part, int* fused_wgs,
                             dmc_stream_t stream) {
    if (!mv || !res || !w || !b || !out || !workspace)
        return fail(DMC_E_INVALID, "dmc_gen_tiny_fwd: null pointer");
    // saved == NULL (inference: nothing kept for a backward pass) is served by the one-launch forward only -- the layer-by-layer
    // kernels pass the features from launch to launch THROUGH that buffer
    if (!saved && !((option(OPT_GEN_FUSED) & 1) && N > 0 && H > 0 && W > 0 && gen_fused_supported(H, W)))
        return fail(DMC_E_INVALID, "dmc_gen_tiny_fwd: saved == NULL needs the fused forward (option gen_fused bit 0, W <= 224)");
    if (N <= 0 || H <= 0 || W <= 0) return fail(DMC_E_INVALID, "dmc_gen_tiny_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (fused_wgs) *fused_wgs = 0;
    // option gen_fused (default): the whole forward as ONE launch (gen_fused.hip: line-buffered in LDS, layers pipelined
    // across waves); the layer-by-layer kernels below serve wider images and the A/B options
    if ((option(OPT_GEN_FUSED) & 1) && gen_fused_supported(H, W)) {
        ParamPtrs P;                                 // (read in place: the one-launch forward needs no repacked parameter block)
        for (int k = 0; k < NL; ++k) {
            if (!w[k] || !b[k]) return fail(DMC_E_INVALID, "null weight/bias pointer %d", k);
            P.w[k] = w[k];
            P.b[k] = b[k];
        }
        return gen_fused_fwd(mv, res, saved, out, P, flow, mse_part, fused_wgs, N, H, W, add_mv_delta, s);
    }
    int x3mask = option(OPT_GEN_X3);
    for (int K = 0; K < GX_LAYERS; ++K)
        if (!gen_x3_supported(K, H, W)) x3mask &= ~(1 << K);
    x3mask &= (1 << GX_LAYERS) - 1;
    if (wino_shape_ok(H, W)) x3mask &= ~option(OPT_GEN_WINO);        // a layer on the Winograd kernel (launch_layer) does not take gen_x3.hip
    // one launch packs the fp32 parameter block and (when a layer takes that path) the bf16x3 fragments behind it
    int rc = pack(w, b, workspace, s, x3mask ? reinterpret_cast<unsigned short*>(workspace + PACKED_TOTAL + ZERO_PAD) : nullptr);
    if (rc) return rc;
    LayerArgs a;
    a.mv = mv; a.res = res; a.feat = saved; a.feat_out = saved; a.gout = nullptr; a.gbuf = nullptr;
    a.pk = workspace; a.out = out; a.H = H; a.W = W; a.add_mv = add_mv_delta;
    a.mse_flow = nullptr; a.mse_part = nullptr;
    if (fused_wgs) *fused_wgs = 0;
    const int step = frames_per_pass(N, H, W);
    for (int n0 = 0; n0 < N; n0 += step) {
        const int nn = (N - n0) < step ? (N - n0) : step;
        // option gen_x3: bit K = hidden layer K on the bf16x3 16x16x32 kernel of gen_x3.hip
        const size_t HWn = (size_t)H * W;
        auto x3 = [&](int K) {
            return gen_x3_layer(K, mv + (size_t)n0 * 2 * HWn, res + (size_t)n0 * 3 * HWn, saved + (size_t)n0 * NFEAT * HWn, workspace,
                                workspace + PACKED_TOTAL + ZERO_PAD, nn, H, W, s);
        };
        if ((rc = ((x3mask >> 0) & 1) ? x3(0) : launch_layer<0, 0>(a, n0, nn, s))) return rc;
        if ((rc = ((x3mask >> 1) & 1) ? x3(1) : launch_layer<0, 1>(a, n0, nn, s))) return rc;
        if ((rc = ((x3mask >> 2) & 1) ? x3(2) : launch_layer<0, 2>(a, n0, nn, s))) return rc;
        if ((rc = launch_layer<0, 3>(a, n0, nn, s))) return rc;
        const int fuse45 = option(OPT_GEN_FUSE45), lpath = option(OPT_GEN_LAYER_PATH);
        if (fuse45 && (lpath == 1 || lpath == 3 || lpath == 4 || lpath == 5) && W % 4 == 0 && W <= P_MAXW) {
            RingArgs ra;
            ra.ablate = 0;
            ra.a = a;
            const size_t HWf = (size_t)H * W;
            ra.a.mv = mv + (size_t)n0 * 2 * HWf; ra.a.res = res + (size_t)n0 * 3 * HWf;
            ra.a.feat = saved + (size_t)n0 * NFEAT * HWf; ra.a.feat_out = saved + (size_t)n0 * NFEAT * HWf;
            ra.a.out = out + (size_t)n0 * 2 * HWf;
            ra.tiles_y = (H + PT_H - 1) / PT_H;
            ra.ntiles = ra.tiles_y * nn;
            const int wgs = ra.ntiles < num_cus() ? ra.ntiles : num_cus();
            if (flow && mse_part && step >= N) {              // (one pass over all frames: one partial set)
                ra.a.mse_flow = flow; ra.a.mse_part = mse_part;
                if (fused_wgs) *fused_wgs = wgs * P_CONS;
            }
            gen_l45_kernel<<<wgs, LTHREADS, 0, s>>>(ra);
            if ((rc = check_launch("gen_l45"))) return rc;
        } else {
            if ((rc = launch_layer<0, 4>(a, n0, nn, s))) return rc;
            if ((rc = launch_layer<1, 5>(a, n0, nn, s))) return rc;
        }
    }
    return DMC_OK;
}

int dmc_gen_tiny_fwd(const float* mv, const float* res, const float* const* w,
                     const float* const* b, float* out, float* saved, float* workspace, int N,
                     int H, int W, int add_mv_delta, dmc_stream_t stream) {
    return gen_tiny_fwd_impl(mv, res, w, b, out, saved, workspace, N, H, W, add_mv_delta, nullptr, nullptr, nullptr,
                             stream);
}

size_t dmc_gen_tiny_mse_partials_bytes(void) {
    const int fparts = num_cus_hw() * P_CONS > gen_fused_max_partials() ? num_cus_hw() * P_CONS : gen_fused_max_partials();
    const size_t fused = (size_t)fparts * sizeof(double), plain = dmc_flow_mse_partials_bytes();
    return fused > plain ? fused : plain;
}

int dmc_gen_tiny_fwd_mse(const float* mv, const float* res, const float* const* w, const float* const* b,
                         const float* flow, float* out, float* saved, float* workspace, float* loss_out,
                         void* mse_partials, int N, int H, int W, int add_mv_delta, dmc_stream_t stream) {
    if (!flow || !loss_out || !mse_partials) return fail(DMC_E_INVALID, "dmc_gen_tiny_fwd_mse: null pointer");
    int wgs = 0;
    int rc = gen_tiny_fwd_impl(mv, res, w, b, out, saved, workspace, N, H, W, add_mv_delta, flow,
                               static_cast<double*>(mse_partials), &wgs, stream);
    if (rc) return rc;
    const size_t numel = (size_t)N * 2 * H * W;
    if (wgs == 0)       // this shape took a path without the fused epilogue: the streaming reduction instead
        return dmc_flow_mse_fwd(out, flow, loss_out, static_cast<float*>(mse_partials), numel, stream);
    gen_mse_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(static_cast<const double*>(mse_partials), wgs, loss_out, (double)numel);
    return check_launch("gen_mse_final");
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
    la.mse_flow = nullptr; la.mse_part = nullptr;
    const int step = frames_per_pass(N, H, W);
    // option gen_fused bit 1: the five data-gradient groups as ONE launch (gen_fused_bwd.hip) -- measured slower than the five
    // launches (0.75 vs 0.60 ms): the kernel and the option value exist in the -DDMC_MEASURE build only
#ifdef DMC_MEASURE
    const bool fused_data = (option(OPT_GEN_FUSED) & 2) && gen_fused_supported(H, W);
    if (fused_data && (rc = gen_fused_bwd_data(grad_out, saved, gbuf, workspace, N, H, W, s))) return rc;
#else
    const bool fused_data = false;
#endif
    for (int n0 = 0; n0 < N && !fused_data; n0 += step) {
        const int nn = (N - n0) < step ? (N - n0) : step;
        if ((rc = launch_layer<2, 4>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 3>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 2>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 1>(la, n0, nn, s))) return rc;
        if ((rc = launch_layer<2, 0>(la, n0, nn, s))) return rc;
    }

    WgradArgs a;
    a.mv = mv; a.res = res; a.feat = saved; a.gout = grad_out; a.gbuf = gbuf; a.partials = partials;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + WT_W - 1) / WT_W;
    int groups = wgrad_groups(N, H, W);
    const int wpath = option(OPT_GEN_WGRAD_PATH);
    const bool rs = wpath == 5 && gen_wgrad_rs_supported(H, W);            // row-sliding kernel (gen_wgrad.hip)
    const int layout3 = rs ? 2 : (W % 4 == 0 && wpath >= 3) ? 1 : 0;
    if (rs) {
        groups = gen_wgrad_rs_groups(N, H, W, groups);
        if ((rc = gen_wgrad_rs(mv, res, saved, grad_out, gbuf, workspace + PACKED_TOTAL, partials, N, H, W, groups, s))) return rc;
    } else if (W % 4 == 0 && wpath >= 1) {
        a.tiles_y = (H + PW_H - 1) / PW_H;
        // paths 1 .. 3 (fp32 producer / consumer, earlier bf16x3 forms): slower predecessors of path 4, -DDMC_MEASURE build only
        if (!MEASURE_BUILD || wpath >= 4) gen_bwd_weight_pc_kernel<3><<<groups, 512, 0, s>>>(a);
        else if constexpr (MEASURE_BUILD) {
            if (wpath == 3) gen_bwd_weight_pc_kernel<2><<<groups, 512, 0, s>>>(a);
            else if (wpath == 2) gen_bwd_weight_pc_kernel<1><<<groups, 512, 0, s>>>(a);
            else gen_bwd_weight_pc_kernel<0><<<groups, 512, 0, s>>>(a);
        }
    } else {
        a.tiles_y = (H + WT_H - 1) / WT_H;
        // (W % 4 == 0 reaches this branch only with gen_wgrad_path 0, a -DDMC_MEASURE value)
        bool done = false;
        if constexpr (MEASURE_BUILD) {
            if (W % 4 == 0) {
                gen_bwd_weight_kernel<true><<<groups, 512, 0, s>>>(a, workspace + PACKED_TOTAL);
                done = true;
            }
        }
        if (!done) gen_bwd_weight_kernel<false><<<groups, 512, 0, s>>>(a, workspace + PACKED_TOTAL);
    }
    if ((rc = check_launch("gen_bwd_weight"))) return rc;
    const int wpart = layout3 ? WPART3 : WPART;
    gen_bwd_weight_reduce1_kernel<<<dim3(wpart / 256, RED_CHUNKS), 256, 0, s>>>(partials, groups, wpart);
    if ((rc = check_launch("gen_bwd_weight_reduce1"))) return rc;
    gen_bwd_weight_reduce_kernel<<<(NPARAM + 127) / 128, 128, 0, s>>>(partials, groups, G, layout3);
    return check_launch("gen_bwd_weight_reduce");
}

}  // extern "C"
