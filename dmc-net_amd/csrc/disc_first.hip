// First discriminator block's convolution: Conv2d(2, COUT, 3, stride 2, padding 1) on the NCHW
// 2-channel cue (generated / TV-L1 flow), code/dmcnet_GAN/model.py:254-265 as instantiated at :334
// (`discriminator_block(ch_in, 16, bn=False)`), with the block's LeakyReLU(0.2) and Dropout2d keep
// mask fused, writing NHWC for the matrix-core blocks that follow (conv_nhwc.hip).
//
// 18 multiply-adds per output value: far too thin for the matrix cores and HBM-bound (401 KB read,
// 803 KB written per 224x224 frame), so these are direct VALU kernels: one thread per output pixel,
// all COUT channels in registers, weights through the scalar cache.
//   forward   x [M,2,H,W] -> z [M,OH,OW,COUT] = keep * lrelu(conv + bias)
//   dgrad     g [M,OH,OW,COUT] -> dx [M,2,H,W]     (gathers the <= 4 taps whose parity matches)
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

// thread t < COUT * 19 owns output (co, j): j < 18 = (ci, tap) of dw, j == 18 = db; it walks the
// workgroup's pixel range (g values and input taps come from L1 / the scalar-friendly broadcast reads).
constexpr int FW_MAX_BLOCKS = 1024;
template <int COUT>
__global__ __launch_bounds__(320) void disc_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                               double* __restrict__ part, int M, int H, int W, int OH,
                                                               int OW) {
    const int t = threadIdx.x;
    if (t >= COUT * 19) return;
    const int co = t % COUT, j = t / COUT;
    const int ci = j / 9, tap = j % 9, ky = tap / 3, kx = tap % 3;
    const long total = (long)M * OH * OW;
    const long per = (total + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = p0 + per < total ? p0 + per : total;
    double acc = 0.0;
    for (long q0 = p0; q0 < p1; q0 += 256) {                 // short fp32 runs, fp64 across them
        const long q1 = q0 + 256 < p1 ? q0 + 256 : p1;
        float run = 0.f;
#pragma unroll 8
        for (long p = q0; p < q1; ++p) {
            const float gv = g[p * COUT + co];
            float xv = 1.f;
            if (j < 18) {
                const int n = (int)(p / ((long)OH * OW));
                const int rem = (int)(p - (long)n * OH * OW);
                const int oy = rem / OW, ox = rem - oy * OW;
                const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
                xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((long)n * 2 + ci) * H + iy) * W + ix] : 0.f;
            }
            run = fmaf(gv, xv, run);
        }
        acc += (double)run;
    }
    part[(size_t)blockIdx.x * (COUT * 19) + t] = acc;
}

template <int COUT>
__global__ void disc_first_wgrad_final_kernel(const double* __restrict__ part, int nblk, float* __restrict__ dw,
                                              float* __restrict__ db) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= COUT * 19) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * (COUT * 19) + t];
    const int co = t % COUT, j = t / COUT;
    if (j < 18) dw[(co * 2 + j / 9) * 9 + j % 9] = (float)s;
    else if (db) db[co] = (float)s;
}

int blocks_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

int wgrad_blocks(long pixels) {
    long b = pixels / 2048;
    return (int)(b < 1 ? 1 : (b > FW_MAX_BLOCKS ? FW_MAX_BLOCKS : b));
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

size_t dmc_disc_first_wgrad_bytes(int Cout) { return (size_t)FW_MAX_BLOCKS * Cout * 19 * sizeof(double); }

int dmc_disc_first_wgrad(const float* x, const float* g, float* dw, float* db, void* workspace, int M, int H,
                         int W, int Cout, dmc_stream_t stream) {
    if (!x || !g || !dw || !workspace) return fail(DMC_E_INVALID, "dmc_disc_first_wgrad: null pointer");
    if (M <= 0 || H <= 0 || W <= 0 || !dmc_disc_first_supported(Cout))
        return fail(DMC_E_INVALID, "dmc_disc_first_wgrad: unsupported shape");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const int nb = wgrad_blocks((long)M * OH * OW);
    double* part = static_cast<double*>(workspace);
    if (Cout == 16) disc_first_wgrad_kernel<16><<<nb, 320, 0, s>>>(x, g, part, M, H, W, OH, OW);
    else disc_first_wgrad_kernel<8><<<nb, 320, 0, s>>>(x, g, part, M, H, W, OH, OW);
    int rc = check_launch("disc_first_wgrad");
    if (rc) return rc;
    if (Cout == 16) disc_first_wgrad_final_kernel<16><<<(16 * 19 + 63) / 64, 64, 0, s>>>(part, nb, dw, db);
    else disc_first_wgrad_final_kernel<8><<<(8 * 19 + 63) / 64, 64, 0, s>>>(part, nb, dw, db);
    return check_launch("disc_first_wgrad_final");
}

}  // extern "C"
