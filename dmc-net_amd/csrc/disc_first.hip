// First discriminator block's convolution: Conv2d(2, COUT, 3, stride 2, padding 1) on the NCHW
// 2-channel cue (generated / TV-L1 flow), code/dmcnet_GAN/model.py:254-265 as instantiated at :334
// (`discriminator_block(ch_in, 16, bn=False)`), with the block's LeakyReLU(0.2) and Dropout2d keep
// mask fused, writing NHWC for the matrix-core blocks that follow (conv_nhwc.hip).
//
// 18 multiply-adds per output value: far too thin for the matrix cores and HBM-bound (401 KB read,
// 803 KB written per 224x224 frame), so these are direct VALU kernels: one thread per output pixel,
// all COUT channels in registers, weights through the scalar cache.
//   forward   x [M,2,H,W] -> z [M,OH,OW,COUT] = keep * lrelu(conv + bias)
//   dgrad     g [M,OH,OW,COUT] -> dx [M,2,H,W]     (gathers the <= 4 taps whose parity matches; an LDS-staged form with one thread per
//             2 x 2 input pixels and static taps was built and measured SLOWER, 295 vs 254 us at 384 frames: 288 scalar-cache
//             weights per thread)
//   wgrad     dw [COUT,2,3,3], db [COUT]: per-workgroup partials over a pixel range, then a
//             fixed-order reduction (deterministic)
#include "dmc_common.h"

using namespace dmc;

namespace {

template <int COUT>
__global__ __launch_bounds__(256) void disc_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ keep, float* __restrict__ z,
                                                             int M, int H, int W, int OH, int OW, int act) {
    const long total = (long)M * OH * OW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int n = (int)(p / ((long)OH * OW));
        const int rem = (int)(p - (long)n * OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        float v[2][9];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    v[ci][ky * 3 + kx] = ok ? x[(((long)n * 2 + ci) * H + iy) * W + ix] : 0.f;
                }
        float out[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float acc = bias ? bias[co] : 0.f;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(v[ci][t], w[(co * 2 + ci) * 9 + t], acc);
            if (act) acc = acc > 0.f ? acc : 0.2f * acc;
            if (keep) acc *= keep[(long)n * COUT + co];
            out[co] = acc;
        }
        float4* dst = reinterpret_cast<float4*>(z + p * COUT);
#pragma unroll
        for (int c4 = 0; c4 < COUT / 4; ++c4)
            dst[c4] = make_float4(out[4 * c4], out[4 * c4 + 1], out[4 * c4 + 2], out[4 * c4 + 3]);
    }
}

// Forward through LDS (COUT = 16, W % 4 == 0, W <= 512): a workgroup owns FOUR output rows of one frame.  It stages the nine input
// rows 2 oy0 - 1 .. 2 oy0 + 7 of both planes with coalesced 16-byte loads (zeros outside the image, a zero pad left and right of every
// row), then a thread = (output pixel, channel quad): its 18 taps are LDS reads shared by the pixel's four lanes, its 72 weights sit in
// registers, and the 64 lanes of a wave store 1 KB of consecutive NHWC memory.  (One thread per pixel with all 16 channels made
// every vector access 64-byte strided: 192 us for the I3D recipe's 384 frames against a stream floor of 85.)
constexpr int DF_ROWS = 4, DF_PAD = 4;
__global__ __launch_bounds__(256) void disc_first_fwd_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, const float* __restrict__ keep,
                                                                 float* __restrict__ z, int H, int W, int OH, int OW, int act,
                                                                 int strips) {
    extern __shared__ __attribute__((aligned(16))) float df_lds[];          // [2 planes][9 rows][W + 8]
    const int tid = threadIdx.x;
    const int n = (int)blockIdx.x / strips, oy0 = ((int)blockIdx.x % strips) * DF_ROWS;
    const int pitch = W + 2 * DF_PAD;
    // ---- stage ----
    const int q_per_row = pitch / 4;                                         // float4 slots per LDS row (pads included)
    const int nslots = 2 * 9 * q_per_row;
    for (int i = tid; i < nslots; i += 256) {
        const int q = i % q_per_row, r = (i / q_per_row) % 9, c = i / (9 * q_per_row);
        const int iy = 2 * oy0 - 1 + r, ix = 4 * q - DF_PAD;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4*>(x + (((size_t)n * 2 + c) * H + iy) * W + ix);
        *reinterpret_cast<float4*>(df_lds + (c * 9 + r) * pitch + 4 * q) = v;
    }
    // ---- this thread's channel quad: weights, bias, keep ----
    const int c4 = tid & 3;
    float wr[4][18], br[4], kr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int co = 4 * c4 + e;
#pragma unroll
        for (int t = 0; t < 18; ++t) wr[e][t] = w[co * 18 + t];              // [co][ci][ky][kx]
        br[e] = bias ? bias[co] : 0.f;
        kr[e] = keep ? keep[(size_t)n * 16 + co] : 1.f;
    }
    __syncthreads();
    const int items = DF_ROWS * OW * 4;
    for (int it = tid; it < items; it += 256) {
        const int ox = (it >> 2) % OW, r = (it >> 2) / OW;
        const int oy = oy0 + r;
        if (oy >= OH) break;
        float acc[4] = {br[0], br[1], br[2], br[3]};
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* row = df_lds + (ci * 9 + 2 * r + ky) * pitch + DF_PAD + 2 * ox - 1;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = row[kx];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, wr[e][ci * 9 + ky * 3 + kx], acc[e]);
                }
            }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = acc[e];
            if (act) t = t > 0.f ? t : 0.2f * t;
            o[e] = t * kr[e];
        }
        *reinterpret_cast<float4*>(z + (((size_t)n * OH + oy) * OW + ox) * 16 + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void disc_first_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                               float* __restrict__ dx, int M, int H, int W, int OH,
                                                               int OW) {
    const long total = (long)M * H * W;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int n = (int)(p / ((long)H * W));
        const int rem = (int)(p - (long)n * H * W);
        const int iy = rem / W, ix = rem - iy * W;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= OH) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= OW) continue;
                const float4* gp = reinterpret_cast<const float4*>(g + (((long)n * OH + oy) * OW + ox) * COUT);
#pragma unroll
                for (int c4 = 0; c4 < COUT / 4; ++c4) {
                    const float4 gv = gp[c4];
                    const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = 4 * c4 + e;
                        a0 = fmaf(gs[e], w[(co * 2 + 0) * 9 + ky * 3 + kx], a0);
                        a1 = fmaf(gs[e], w[(co * 2 + 1) * 9 + ky * 3 + kx], a1);
                    }
                }
            }
        }
        dx[((long)n * 2 + 0) * H * W + rem] = a0;
        dx[((long)n * 2 + 1) * H * W + rem] = a1;
    }
}

// Weight / bias gradient on the matrix cores: one GEMM over output pixels,
//   rows = co (16; A operand = g, lane (co = l & 15, k = l >> 4) reads g[pixel 4 q + k][co]: the wave's
//          64 lanes read 256 contiguous bytes),
//   cols = j: 18 (ci, tap) columns of dw + a ones column for db, in two 16-column tiles (B operand =
//          the tap-shifted input, gathered from the NCHW planes: L1 / L2 hits, each value serves 2.25 taps),
//   K    = pixels, 4 per v_mfma_f32_16x16x4_f32.
// A wave owns a contiguous run of pixel quads and keeps its two accumulator tiles in registers; pixel
// coordinates advance incrementally (no divisions in the loop).  Partials [wave][2][16 x 16] are summed
// in fixed order in fp64 by the final kernel: deterministic.
constexpr int FW_WAVES = 2048;                 // 256 CUs x 8 waves
typedef float fw_f32x4 __attribute__((ext_vector_type(4)));

template <int COUT>
__global__ __launch_bounds__(256) void disc_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                               float* __restrict__ part, int M, int H, int W, int OH,
                                                               int OW, long quads_per_wave) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const long total = (long)M * OH * OW;
    const long q0 = (long)gw * quads_per_wave;
    long q1 = q0 + quads_per_wave;
    const long nquads = (total + 3) / 4;
    if (q1 > nquads) q1 = nquads;
    // column roles of this lane: tile 0 column j0 = i16 (ci = j0 / 9, tap = j0 % 9); tile 1 column
    // 16 + i16: j = 16, 17 -> (ci 1, taps 7, 8), j = 18 -> ones (db), the rest zero
    const int ci0 = i16 / 9, t0 = i16 % 9, ky0 = t0 / 3 - 1, kx0 = t0 % 3 - 1;
    const int ky1 = 1, kx1 = (i16 == 0) ? 0 : 1;            // taps 7 = (2, 1), 8 = (2, 2) -> offsets (+1, 0), (+1, +1)
    const bool t1_tap = i16 < 2, t1_one = i16 == 2;
    const bool co_ok = i16 < COUT;
    fw_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // this lane's pixel of the current quad
    long p = 4 * q0 + kq;
    int n = 0, oy = 0, ox = 0;
    if (q0 < q1) {
        n = (int)(p / ((long)OH * OW));
        const int rem = (int)(p - (long)n * OH * OW);
        oy = rem / OW; ox = rem - oy * OW;
    }
    const long plane = (long)H * W;
    constexpr int U = 8;                                     // quads in flight per lane
    for (long q = q0; q < q1; q += U) {
        float av[U], b0[U], b1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool pok = p < total && q + u < q1;
            av[u] = 0.f; b0[u] = 0.f; b1[u] = (t1_one && pok) ? 1.f : 0.f;
            if (pok) {
                if (co_ok) av[u] = g[p * COUT + i16];
                const int iy = 2 * oy + ky0, ix = 2 * ox + kx0;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) b0[u] = x[((long)n * 2 + ci0) * plane + (long)iy * W + ix];
                if (t1_tap) {
                    const int jy = 2 * oy + ky1, jx = 2 * ox + kx1;
                    if (jy < H && jx < W) b1[u] = x[((long)n * 2 + 1) * plane + (long)jy * W + jx];
                }
            }
            p += 4; ox += 4;
            if (ox >= OW) {                                  // next row / frame: once per OW / 4 quads
                n = (int)(p / ((long)OH * OW));
                const int rem = (int)(p - (long)n * OH * OW);
                oy = rem / OW; ox = rem - oy * OW;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b0[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b1[u], acc1, 0, 0, 0);
        }
    }
    // D layout: lane holds column j = i16 (+16), rows co = 4 kq + e
    float* out = part + (size_t)gw * 512;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        out[(4 * kq + e) * 16 + i16] = acc0[e];
        out[256 + (4 * kq + e) * 16 + i16] = acc1[e];
    }
}

template <int COUT>
__global__ __launch_bounds__(64) void disc_first_wgrad_final_kernel(const float* __restrict__ part, int nwaves,
                                                                    float* __restrict__ dw, float* __restrict__ db) {
    // one wave per output (co, j): 64 lanes stride over the partials, fp64, fixed order
    const int t = blockIdx.x;                       // < COUT * 19
    const int co = t % COUT, j = t / COUT, lane = threadIdx.x;
    const int idx = (j >> 4) * 256 + co * 16 + (j & 15);
    double s = 0.0;
    for (int b = lane; b < nwaves; b += 64) s += (double)part[(size_t)b * 512 + idx];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane != 0) return;
    if (j < 18) dw[(co * 2 + j / 9) * 9 + j % 9] = (float)s;
    else if (db) db[co] = (float)s;
}

int blocks_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" {

int dmc_disc_first_supported(int Cout) { return (Cout == 16 || Cout == 8) ? 1 : 0; }

int dmc_disc_first_fwd(const float* x, const float* w, const float* bias, const float* keep, float* z, int M,
                       int H, int W, int Cout, int act, dmc_stream_t stream) {
    if (!x || !w || !z) return fail(DMC_E_INVALID, "dmc_disc_first_fwd: null pointer");
    if (M <= 0 || H <= 0 || W <= 0 || !dmc_disc_first_supported(Cout))
        return fail(DMC_E_INVALID, "dmc_disc_first_fwd: unsupported shape M=%d H=%d W=%d Cout=%d", M, H, W, Cout);
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const int nb = blocks_for((long)M * OH * OW);
    if (Cout == 16 && W % 4 == 0 && W <= 512 && option(OPT_CONV_CFG) != 305) {          // 305: the one-thread-per-pixel form (A/B)
        const int strips = (OH + DF_ROWS - 1) / DF_ROWS;
        const size_t lds = (size_t)2 * 9 * (W + 2 * DF_PAD) * sizeof(float);
        disc_first_fwd_lds_kernel<<<M * strips, 256, lds, s>>>(x, w, bias, keep, z, H, W, OH, OW, act, strips);
        return check_launch("disc_first_fwd_lds");
    }
    if (Cout == 16) disc_first_fwd_kernel<16><<<nb, 256, 0, s>>>(x, w, bias, keep, z, M, H, W, OH, OW, act);
    else disc_first_fwd_kernel<8><<<nb, 256, 0, s>>>(x, w, bias, keep, z, M, H, W, OH, OW, act);
    return check_launch("disc_first_fwd");
}

int dmc_disc_first_dgrad(const float* g, const float* w, float* dx, int M, int H, int W, int Cout,
                         dmc_stream_t stream) {
    if (!g || !w || !dx) return fail(DMC_E_INVALID, "dmc_disc_first_dgrad: null pointer");
    if (M <= 0 || H <= 0 || W <= 0 || !dmc_disc_first_supported(Cout))
        return fail(DMC_E_INVALID, "dmc_disc_first_dgrad: unsupported shape");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const int nb = blocks_for((long)M * H * W);
    if (Cout == 16) disc_first_dgrad_kernel<16><<<nb, 256, 0, s>>>(g, w, dx, M, H, W, OH, OW);
    else disc_first_dgrad_kernel<8><<<nb, 256, 0, s>>>(g, w, dx, M, H, W, OH, OW);
    return check_launch("disc_first_dgrad");
}

size_t dmc_disc_first_wgrad_bytes(int Cout) { (void)Cout; return (size_t)FW_WAVES * 512 * sizeof(float); }

int dmc_disc_first_wgrad(const float* x, const float* g, float* dw, float* db, void* workspace, int M, int H,
                         int W, int Cout, dmc_stream_t stream) {
    if (!x || !g || !dw || !workspace) return fail(DMC_E_INVALID, "dmc_disc_first_wgrad: null pointer");
    if (M <= 0 || H <= 0 || W <= 0 || !dmc_disc_first_supported(Cout))
        return fail(DMC_E_INVALID, "dmc_disc_first_wgrad: unsupported shape");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const long nquads = ((long)M * OH * OW + 3) / 4;
    long per = (nquads + FW_WAVES - 1) / FW_WAVES;
    if (per < 16) per = 16;
    const int nwaves = (int)((nquads + per - 1) / per);
    const int nb = (nwaves + 3) / 4;
    float* part = static_cast<float*>(workspace);
    if (Cout == 16) disc_first_wgrad_kernel<16><<<nb, 256, 0, s>>>(x, g, part, M, H, W, OH, OW, per);
    else disc_first_wgrad_kernel<8><<<nb, 256, 0, s>>>(x, g, part, M, H, W, OH, OW, per);
    int rc = check_launch("disc_first_wgrad");
    if (rc) return rc;
    if (Cout == 16) disc_first_wgrad_final_kernel<16><<<16 * 19, 64, 0, s>>>(part, nb * 4, dw, db);
    else disc_first_wgrad_final_kernel<8><<<8 * 19, 64, 0, s>>>(part, nb * 4, dw, db);
    return check_launch("disc_first_wgrad_final");
}

}  // extern "C"
