// EstimatorDenseNetTiny forward / backward for gfx950.
//
// Reference behaviour: code/dmcnet/model.py:172-194 (EstimatorDenseNetTiny), :111-119 (conv,
// predict_flow), :341-346 (cat(mv,res), +input_mv).  Nothing here is derived from reference
// source text; the reference has no kernels at all.
//
// This file holds the "layerwise" path: one launch per layer, features kept in a
// [N][28][H][W] buffer in physical (append) channel order.  It handles any H, W and is the
// path the backward pass reads its saved activations from.
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

// Loads the 3x6 neighbourhood (rows y-1..y+1, cols x0-1..x0+4) of one plane, zero outside.
template <bool VEC4>
__device__ __forceinline__ void load_patch(const float* __restrict__ plane, int y, int x0, int H,
                                           int W, float (&xv)[3][6]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        const bool rowok = (yy >= 0) && (yy < H);
        const float* row = plane + (size_t)(rowok ? yy : 0) * W;
        if (VEC4 && rowok && x0 + 4 < W && x0 > 0) {
            const float4 c = *reinterpret_cast<const float4*>(row + x0);
            xv[ky][0] = row[x0 - 1];
            xv[ky][1] = c.x; xv[ky][2] = c.y; xv[ky][3] = c.z; xv[ky][4] = c.w;
            xv[ky][5] = row[x0 + 4];
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int xx = x0 - 1 + j;
                xv[ky][j] = (rowok && xx >= 0 && xx < W) ? row[xx] : 0.f;
            }
        }
    }
}

__device__ __forceinline__ const float* in_plane(const float* mv, const float* res,
                                                 const float* feat, int n, int p, size_t HW) {
    return p < 2 ? mv + ((size_t)n * 2 + p) * HW
                 : p < NIN ? res + ((size_t)n * 3 + (p - 2)) * HW
                           : feat + ((size_t)n * NFEAT + (p - NIN)) * HW;
}

// ------------------------------------------------------------------------------------------
// forward, one layer: thread = 4 horizontally adjacent pixels x all COUT channels.
// Weights are wave-uniform -> scalar loads, FMAs take them as SGPR operands.
// block (64, 4): x = strip, y = row
// ------------------------------------------------------------------------------------------
template <int K, bool VEC4>
__global__ __launch_bounds__(256) void gen_layer_fwd_kernel(
    const float* __restrict__ mv, const float* __restrict__ res, float* feat,
    const float* __restrict__ pk, float* __restrict__ out, int H, int W, int add_mv) {
    constexpr int CIN = cin_of(K), COUT = cout_of(K);
    const int n = blockIdx.z;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (y >= H || x0 >= W) return;
    const size_t HW = (size_t)H * W;

    float acc[COUT][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float bv = pk[bf_off(K) + co];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][j] = bv;
    }
    const float* wk = pk + wf_off(K);
#pragma unroll 1
    for (int p = 0; p < CIN; ++p) {
        float xv[3][6];
        load_patch<VEC4>(in_plane(mv, res, feat, n, p, HW), y, x0, H, W, xv);
        const float* wp = wk + p * 9 * COUT;
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
    const size_t pix = (size_t)y * W + x0;
    if (K < 5) {
        float* dst = feat + ((size_t)n * NFEAT + (yoff(K) - NIN)) * HW + pix;
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
    } else {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float* dst = out + ((size_t)n * 2 + co) * HW + pix;
            const float* m = mv + ((size_t)n * 2 + co) * HW + pix;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (VEC4 || x0 + j < W) dst[j] = acc[co][j] + (add_mv ? m[j] : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, data path of layer K (K = 5..1): correlation of g_K (COUT[K] channels) with the
// flipped weights into the gradient of every feature channel below it, accumulated in gbuf.
// After layer K's contribution the channels of y_{K-1} are complete (all their consumers
// K..5 are done), so they are turned into g_{K-1} = dL/dy_{K-1} * LeakyReLU'(.) right here.
// ------------------------------------------------------------------------------------------
template <int K, bool VEC4>
__global__ __launch_bounds__(256) void gen_layer_bwd_data_kernel(
    const float* __restrict__ gout, const float* __restrict__ feat, float* gbuf,
    const float* __restrict__ pk, int H, int W) {
    constexpr int G = cout_of(K), D = cin_of(K) - NIN, TOP0 = D - cout_of(K - 1);
    const int n = blockIdx.z;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (y >= H || x0 >= W) return;
    const size_t HW = (size_t)H * W;

    float acc[D][4];
#pragma unroll
    for (int cd = 0; cd < D; ++cd)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[cd][j] = 0.f;
#pragma unroll 1
    for (int cg = 0; cg < G; ++cg) {
        const float* plane = (K == 5) ? gout + ((size_t)n * 2 + cg) * HW
                                      : gbuf + ((size_t)n * NFEAT + (yoff(K) - NIN) + cg) * HW;
        float xv[3][6];
        load_patch<VEC4>(plane, y, x0, H, W, xv);
        const float* wp = pk + wb_off(K) + cg * 9 * D;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int cd = 0; cd < D; ++cd) {
                    const float wv = wp[(ky * 3 + kx) * D + cd];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[cd][j] = fmaf(xv[ky][j + kx], wv, acc[cd][j]);
                }
    }
    const size_t pix = (size_t)y * W + x0;
#pragma unroll
    for (int cd = 0; cd < D; ++cd) {
        float* dst = gbuf + ((size_t)n * NFEAT + cd) * HW + pix;
        const float* a = feat + ((size_t)n * NFEAT + cd) * HW + pix;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (VEC4 || x0 + j < W) {
                float v = acc[cd][j];
                if (K != 5) v += dst[j];
                if (cd >= TOP0) v *= (a[j] > 0.f ? 1.f : 0.1f);
                dst[j] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, weight path.  Persistent workgroups; each stages an 8x32-pixel tile of all 33
// input/feature planes (halo 1) in LDS.  A LANE owns one (layer, input channel, ky) triple and
// keeps its 3 x COUT partial sums in registers for the whole launch; the pixel loop is
// wave-uniform, so dL/dy values come in through scalar loads and feed the FMAs as SGPR
// operands.  One extra lane per layer has x == 1 and thereby accumulates the bias gradient.
// Per-workgroup partials are reduced by a second kernel in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------
constexpr int WT_H = 8, WT_W = 32, WT_ROWS = WT_H + 2, WT_PITCH = 40, WT_COL0 = 3;
constexpr int WT_LDS = 33 * WT_ROWS * WT_PITCH;
constexpr int WGRAD_MAX_GROUPS = 768;

template <int K, int CHUNK>
struct WTask {
    static constexpr int CIN = cin_of(K), COUT = cout_of(K), NIT = CIN * 3;
    float acc[3][COUT];
    int ci, ky;
    bool active, isbias;

    __device__ __forceinline__ void init(int lane) {
        const int item = CHUNK * 64 + lane;
        isbias = (item == NIT);
        active = (item <= NIT);
        const int it = item < NIT ? item : NIT - 1;
        ci = it / 3;
        ky = it % 3;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[kx][co] = 0.f;
    }

    // gk: plane of g_K channel 0 of this frame; (ty0, tx0): tile origin; th x tw valid pixels
    __device__ __forceinline__ void tile(const float* lds, const float* __restrict__ gk, size_t HW,
                                         int W, int ty0, int tx0, int th, int tw) {
        const float* xbase = lds + (ci * WT_ROWS + ky) * WT_PITCH + WT_COL0;
        for (int r = 0; r < th; ++r) {
            const float* xr = xbase + r * WT_PITCH;
            const float* grow = gk + (size_t)(ty0 + r) * W + tx0;
            for (int s = 0; s * 4 < tw; ++s) {
                float x[6];
                const float4 c = *reinterpret_cast<const float4*>(xr + 4 * s + 1);
                x[0] = xr[4 * s];
                x[1] = c.x; x[2] = c.y; x[3] = c.z; x[4] = c.w;
                x[5] = xr[4 * s + 5];
                if (isbias) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) x[j] = 1.f;
                }
                const bool full = (4 * s + 3 < tw);
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    float g[4];
                    const float* gp = grow + co * HW + 4 * s;
                    if (full) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) g[j] = gp[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) g[j] = (4 * s + j < tw) ? gp[j] : 0.f;
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[kx][co] = fmaf(x[j + kx], g[j], acc[kx][co]);
                }
            }
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ part) const {
        if (!active) return;
        if (isbias) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) part[bf_off(K) + co] = acc[1][co];
        } else {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int co = 0; co < COUT; ++co)
                    part[wf_off(K) + ((ci * 3 + ky) * 3 + kx) * COUT + co] = acc[kx][co];
        }
    }
};

struct WgradArgs {
    const float* mv;
    const float* res;
    const float* feat;
    const float* gout;
    const float* gbuf;
    float* partials;
    int N, H, W, tiles_x, tiles_y;
};

__device__ __forceinline__ const float* g_plane(const WgradArgs& a, int k, int n, size_t HW) {
    return k == 5 ? a.gout + (size_t)n * 2 * HW
                  : a.gbuf + ((size_t)n * NFEAT + (yoff(k) - NIN)) * HW;
}

__device__ __forceinline__ void wgrad_stage_tile(const WgradArgs& a, float* lds, int n, int ty0,
                                                 int tx0, size_t HW) {
    constexpr int COLS = WT_W + 2;
    for (int i = threadIdx.x; i < 33 * WT_ROWS * COLS; i += 256) {
        const int c = i / (WT_ROWS * COLS);
        const int rem = i - c * (WT_ROWS * COLS);
        const int row = rem / COLS, col = rem - row * COLS;
        const int yy = ty0 - 1 + row, xx = tx0 - 1 + col;
        float v = 0.f;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
            v = in_plane(a.mv, a.res, a.feat, n, c, HW)[(size_t)yy * a.W + xx];
        lds[(c * WT_ROWS + row) * WT_PITCH + WT_COL0 + col] = v;
    }
}

// The tile loop, run by every wave with its own task set (T1 [, T2 [, T3]]).
template <class T1, int K1, class T2, int K2, class T3, int K3>
__device__ __forceinline__ void wgrad_wave_loop(const WgradArgs& a, float* lds, int lane) {
    T1 t1; T2 t2; T3 t3;
    t1.init(lane); t2.init(lane); t3.init(lane);
    const size_t HW = (size_t)a.H * a.W;
    const int per_frame = a.tiles_x * a.tiles_y;
    const int ntiles = a.N * per_frame;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = t / per_frame, r = t - n * per_frame;
        const int ty0 = (r / a.tiles_x) * WT_H, tx0 = (r % a.tiles_x) * WT_W;
        const int th = min(WT_H, a.H - ty0), tw = min(WT_W, a.W - tx0);
        __syncthreads();   // previous tile fully consumed
        wgrad_stage_tile(a, lds, n, ty0, tx0, HW);
        __syncthreads();
        t1.tile(lds, g_plane(a, K1, n, HW), HW, a.W, ty0, tx0, th, tw);
        if (K2 >= 0) t2.tile(lds, g_plane(a, K2 < 0 ? 0 : K2, n, HW), HW, a.W, ty0, tx0, th, tw);
        if (K3 >= 0) t3.tile(lds, g_plane(a, K3 < 0 ? 0 : K3, n, HW), HW, a.W, ty0, tx0, th, tw);
    }
    float* part = a.partials + (size_t)blockIdx.x * NPARAM;
    t1.store(part);
    if (K2 >= 0) t2.store(part);
    if (K3 >= 0) t3.store(part);
}

__global__ __launch_bounds__(256) void gen_bwd_weight_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[WT_LDS];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // cost per pixel of a task = 3 * COUT; the four waves carry 30 / 30 / 30 / 24
    if (wave == 0) {
        wgrad_wave_loop<WTask<1, 0>, 1, WTask<5, 0>, 5, WTask<5, 0>, -1>(a, lds, lane);
    } else if (wave == 1) {
        wgrad_wave_loop<WTask<0, 0>, 0, WTask<5, 1>, 5, WTask<5, 1>, -1>(a, lds, lane);
    } else if (wave == 2) {
        wgrad_wave_loop<WTask<2, 0>, 2, WTask<3, 0>, 3, WTask<3, 0>, -1>(a, lds, lane);
    } else {
        wgrad_wave_loop<WTask<3, 1>, 3, WTask<4, 0>, 4, WTask<4, 1>, 4>(a, lds, lane);
    }
}

// partials [groups][NPARAM] (packed order) -> the 12 gradient tensors in PyTorch layout
__global__ void gen_bwd_weight_reduce_kernel(const float* __restrict__ partials, int groups,
                                             GradPtrs G) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NPARAM) return;
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += partials[(size_t)g * NPARAM + i];
    if (i < WF_TOTAL) {
        int k = 0;
        while (k < NL - 1 && i >= wf_off(k + 1)) ++k;
        const int cin = cin_of(k), cout = cout_of(k);
        const int r = i - wf_off(k);
        const int co = r % cout, tap = (r / cout) % 9, p = r / (cout * 9);
        G.w[k][(co * cin + logical_of(k, p)) * 9 + tap] = s;
    } else {
        int k = 0;
        while (k < NL - 1 && i >= bf_off(k + 1)) ++k;
        G.b[k][i - bf_off(k)] = s;
    }
}

int wgrad_groups(int N, int H, int W) {
    const long tiles = (long)N * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    return (int)(tiles < WGRAD_MAX_GROUPS ? tiles : WGRAD_MAX_GROUPS);
}

template <int K>
int launch_fwd_layer(const float* mv, const float* res, float* feat, const float* pk, float* out,
                     int N, int H, int W, int add_mv, hipStream_t s) {
    const dim3 block(64, 4), grid((W + 255) / 256, (H + 3) / 4, N);
    if (W % 4 == 0)
        gen_layer_fwd_kernel<K, true><<<grid, block, 0, s>>>(mv, res, feat, pk, out, H, W, add_mv);
    else
        gen_layer_fwd_kernel<K, false><<<grid, block, 0, s>>>(mv, res, feat, pk, out, H, W, add_mv);
    return check_launch("gen_layer_fwd");
}

template <int K>
int launch_bwd_data_layer(const float* gout, const float* feat, float* gbuf, const float* pk, int N,
                          int H, int W, hipStream_t s) {
    const dim3 block(64, 4), grid((W + 255) / 256, (H + 3) / 4, N);
    if (W % 4 == 0)
        gen_layer_bwd_data_kernel<K, true><<<grid, block, 0, s>>>(gout, feat, gbuf, pk, H, W);
    else
        gen_layer_bwd_data_kernel<K, false><<<grid, block, 0, s>>>(gout, feat, gbuf, pk, H, W);
    return check_launch("gen_layer_bwd_data");
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
    return (size_t)wgrad_groups(N, H, W) * NPARAM * sizeof(float);
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
    if ((rc = launch_fwd_layer<0>(mv, res, saved, workspace, out, N, H, W, add_mv_delta, s))) return rc;
    if ((rc = launch_fwd_layer<1>(mv, res, saved, workspace, out, N, H, W, add_mv_delta, s))) return rc;
    if ((rc = launch_fwd_layer<2>(mv, res, saved, workspace, out, N, H, W, add_mv_delta, s))) return rc;
    if ((rc = launch_fwd_layer<3>(mv, res, saved, workspace, out, N, H, W, add_mv_delta, s))) return rc;
    if ((rc = launch_fwd_layer<4>(mv, res, saved, workspace, out, N, H, W, add_mv_delta, s))) return rc;
    return launch_fwd_layer<5>(mv, res, saved, workspace, out, N, H, W, add_mv_delta, s);
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
    if ((rc = launch_bwd_data_layer<5>(grad_out, saved, gbuf, workspace, N, H, W, s))) return rc;
    if ((rc = launch_bwd_data_layer<4>(grad_out, saved, gbuf, workspace, N, H, W, s))) return rc;
    if ((rc = launch_bwd_data_layer<3>(grad_out, saved, gbuf, workspace, N, H, W, s))) return rc;
    if ((rc = launch_bwd_data_layer<2>(grad_out, saved, gbuf, workspace, N, H, W, s))) return rc;
    if ((rc = launch_bwd_data_layer<1>(grad_out, saved, gbuf, workspace, N, H, W, s))) return rc;

    WgradArgs a;
    a.mv = mv; a.res = res; a.feat = saved; a.gout = grad_out; a.gbuf = gbuf; a.partials = partials;
    a.N = N; a.H = H; a.W = W;
    a.tiles_x = (W + WT_W - 1) / WT_W;
    a.tiles_y = (H + WT_H - 1) / WT_H;
    const int groups = wgrad_groups(N, H, W);
    gen_bwd_weight_kernel<<<groups, 256, 0, s>>>(a);
    if ((rc = check_launch("gen_bwd_weight"))) return rc;
    gen_bwd_weight_reduce_kernel<<<(NPARAM + 127) / 128, 128, 0, s>>>(partials, groups, G);
    return check_launch("gen_bwd_weight_reduce");
}

}  // extern "C"
