// BatchNorm3d (+ ReLU) of the I3D trunk's Unit3Dpy on bf16 NDHWC tensors, training mode (BASELINE config 5).
//
// Replaces `self.batch3d(out)` + `F.relu` in the reference's Unit3Dpy.forward
// (code/dmcnet_I3D/network/i3d.py:394-398; torch.nn.BatchNorm3d with its defaults eps 1e-5, momentum 0.1) and
// their autograd, for the units whose convolution runs on conv3d_bf16.hip: the convolution's epilogue already
// reduced the per-channel (sum, sum of squares) of its rounded output per workgroup, so the forward is
//     bn3d_stats_final  (partials -> mean, invstd, running statistics; one workgroup per channel, fp64)
//     bn3d_apply        (one streaming pass: y bf16 -> relu(gamma (y - mean) invstd + beta) bf16)
// and the backward
//     bn3d_bwd_partial  (one pass over dout, y: g = dout [* (out > 0)], per-channel sum g, sum g xhat)
//     bn3d_bwd_final    (-> dgamma, dbeta and the two per-channel coefficients of the input gradient)
//     bn3d_bwd_apply    (one pass: dy = gamma invstd (g - mean(g) - xhat mean(g xhat)) bf16)
// -- five launches where the stock path (MIOpen's NDHWC BatchNorm + clamp + threshold_backward) issues ten, and
// no fp32 intermediate.  Arithmetic in fp32 per element, fp64 for the per-channel sums, fixed summation order
// (deterministic).  A thread owns 8 channels (16 bytes) of a pixel; a 256-thread workgroup covers
// P = 256 / (C / 8) pixels per iteration.
#include "dmc_common.h"

using namespace dmc;

namespace {

typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned f2bf(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = bf2f(v[e] & 0xffffu); f[2 * e + 1] = bf2f(v[e] >> 16); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = f2bf(f[2 * e]) | (f2bf(f[2 * e + 1]) << 16);
    return v;
}

constexpr int BN3_MAXBLK = 1024;

// partials [nblk][C][2] floats -> stats [2][C] = (mean, invstd); running statistics as nn.BatchNorm3d updates them
__global__ __launch_bounds__(256) void bn3d_stats_final_kernel(const float* __restrict__ part, int nblk, int C, long count,
                                                               float* __restrict__ stats, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float eps, float momentum) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)b * C + c) * 2);
        s += (double)v.x;
        ss += (double)v.y;
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
    if (running_mean) {
        const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

__global__ __launch_bounds__(256) void bn3d_apply_kernel(const bf16_t* __restrict__ y, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16_t* __restrict__ out, long M, int C, int relu, long out_ld) {
    const int C8 = C >> 3, P = 256 / C8;
    const int chunk = threadIdx.x % C8, pl = threadIdx.x / C8;
    if (pl >= P) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 8 * chunk + e;
        sc[e] = gamma[c] * stats[C + c];
        sh[e] = beta[c] - stats[c] * sc[e];
    }
    for (long m = (long)blockIdx.x * P + pl; m < M; m += (long)gridDim.x * P) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(y + m * C + 8 * chunk), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f[e] = f[e] * sc[e] + sh[e];
            if (relu) f[e] = f[e] > 0.f ? f[e] : 0.f;
        }
        *reinterpret_cast<u32x4*>(out + m * out_ld + 8 * chunk) = pack8(f);
    }
}

// per workgroup: sums over its pixels of g and g * xhat per channel -> part [gridDim.x][C][2] floats
__global__ __launch_bounds__(256) void bn3d_bwd_partial_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                               const float* __restrict__ stats, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ part, long M,
                                                               int C, int relu, long dout_ld) {
    __shared__ float red[256 * 16];
    const int C8 = C >> 3, P = 256 / C8;
    const int chunk = threadIdx.x % C8, pl = threadIdx.x / C8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (pl < P) {
        float mean[8], inv[8], sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * chunk + e;
            mean[e] = stats[c]; inv[e] = stats[C + c];
            sc[e] = gamma[c] * inv[e]; sh[e] = beta[c] - mean[e] * sc[e];
        }
        for (long m = (long)blockIdx.x * P + pl; m < M; m += (long)gridDim.x * P) {
            float g[8], v[8];
            unpack8(*reinterpret_cast<const u32x4*>(dout + m * dout_ld + 8 * chunk), g);
            unpack8(*reinterpret_cast<const u32x4*>(y + m * C + 8 * chunk), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (relu && !(v[e] * sc[e] + sh[e] > 0.f)) g[e] = 0.f;
                s1[e] += g[e];
                s2[e] += g[e] * ((v[e] - mean[e]) * inv[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = s1[e]; red[threadIdx.x * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    // fixed order over the pixel lanes: thread (chunk, e16) sums plane 0 .. P-1
    for (int i = threadIdx.x; i < C8 * 16; i += 256) {
        const int ch = i / 16, e16 = i % 16;
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += red[(p * C8 + ch) * 16 + e16];
        const int c = 8 * ch + (e16 & 7);
        part[((size_t)blockIdx.x * C + c) * 2 + (e16 >> 3)] = s;
    }
}

// -> dgamma, dbeta, coef [2][C] = (mean of g, mean of g xhat)
__global__ __launch_bounds__(256) void bn3d_bwd_final_kernel(const float* __restrict__ part, int nblk, int C, long count,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ coef) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)b * C + c) * 2);
        s += (double)v.x;
        ss += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); ss += __shfl_down(ss, o, 64); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
    coef[c] = (float)(s / (double)count);
    coef[C + c] = (float)(ss / (double)count);
}

__global__ __launch_bounds__(256) void bn3d_bwd_apply_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                             const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ coef,
                                                             bf16_t* __restrict__ dy, long M, int C, int relu, long dout_ld) {
    const int C8 = C >> 3, P = 256 / C8;
    const int chunk = threadIdx.x % C8, pl = threadIdx.x / C8;
    if (pl >= P) return;
    float mean[8], inv[8], sc[8], sh[8], mg[8], mgx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 8 * chunk + e;
        mean[e] = stats[c]; inv[e] = stats[C + c];
        sc[e] = gamma[c] * inv[e]; sh[e] = beta[c] - mean[e] * sc[e];
        mg[e] = coef[c]; mgx[e] = coef[C + c];
    }
    for (long m = (long)blockIdx.x * P + pl; m < M; m += (long)gridDim.x * P) {
        float g[8], v[8];
        unpack8(*reinterpret_cast<const u32x4*>(dout + m * dout_ld + 8 * chunk), g);
        unpack8(*reinterpret_cast<const u32x4*>(y + m * C + 8 * chunk), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (relu && !(v[e] * sc[e] + sh[e] > 0.f)) g[e] = 0.f;
            const float xh = (v[e] - mean[e]) * inv[e];
            g[e] = sc[e] * (g[e] - mg[e] - xh * mgx[e]);
        }
        *reinterpret_cast<u32x4*>(dy + m * C + 8 * chunk) = pack8(g);
    }
}

// out = ((a + b) + c) + d, every sum rounded to bf16 -- what three torch additions of bf16 tensors produce, in one pass (the gradient
// of an Inception block's input is the sum of its four branches' data gradients: 27 additions per micro-step, 9 launches here)
__global__ __launch_bounds__(256) void add4_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const bf16_t* __restrict__ c,
                                                        const bf16_t* __restrict__ d, bf16_t* __restrict__ out, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float fa[8], fb[8], fc[8], fd[8];
        unpack8(reinterpret_cast<const u32x4*>(a)[i], fa);
        unpack8(reinterpret_cast<const u32x4*>(b)[i], fb);
        unpack8(reinterpret_cast<const u32x4*>(c)[i], fc);
        unpack8(reinterpret_cast<const u32x4*>(d)[i], fd);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] = bf2f(f2bf(bf2f(f2bf(fa[e] + fb[e])) + fc[e])) + fd[e];
        reinterpret_cast<u32x4*>(out)[i] = pack8(fa);
    }
}

bool bn3_ok(long M, int C) { return M > 0 && C > 0 && C % 8 == 0 && C <= 2048; }
int bn3_blocks(long M, int C) {
    const int P = 256 / (C / 8);
    long b = (M + P - 1) / P;
    b = (b + 1) / 2;                                       // two pixel groups per thread while BN3_MAXBLK allows: the Inception maps are
                                                           // small (9,408 pixels x 128 channels = 2.4 MB) and eight dependent iterations on 74
                                                           // workgroups left these kernels latency-bound (10-11 us each, 171 launches per micro-step)
    return (int)(b < 1 ? 1 : (b > BN3_MAXBLK ? BN3_MAXBLK : b));
}

}  // namespace

extern "C" {

int dmc_bn3d_bf16_supported(long M, int C) { return bn3_ok(M, C) ? 1 : 0; }
// floats of the backward's partial-sum workspace
size_t dmc_bn3d_bf16_scratch_bytes(int C) { return ((size_t)BN3_MAXBLK * C * 2 + 2 * (size_t)C) * sizeof(float); }

// stats [2*C] = (mean, invstd) from the convolution's partials [nblk][C][2] (dmc_conv3d_bf16_fwd with stat_partials);
// running_mean / running_var updated as nn.BatchNorm3d does (NULL to skip); then out = relu?(bn(y)) in bf16
// out_ld: elements between consecutive pixels of `out` (C for a dense tensor; the width of the concatenated tensor when `out` is a
// channel slice of an Inception block's output, which is then written in place -- no torch.cat afterwards)
int dmc_bn3d_bf16_fwd_ld(const void* y, const float* partials, int nblk, const float* gamma, const float* beta, float* stats,
                         float* running_mean, float* running_var, void* out, long out_ld, long M, int C, int relu, float eps,
                         float momentum, dmc_stream_t stream) {
    if (!y || !partials || !gamma || !beta || !stats || !out || nblk <= 0) return fail(DMC_E_INVALID, "dmc_bn3d_bf16_fwd: bad argument");
    if (!bn3_ok(M, C)) return fail(DMC_E_INVALID, "dmc_bn3d_bf16_fwd: unsupported shape M=%ld C=%d", M, C);
    if (out_ld < C || out_ld % 8 != 0 || ((size_t)out & 15)) return fail(DMC_E_INVALID, "dmc_bn3d_bf16_fwd: out_ld=%ld / out must keep 16-byte pixel rows (C=%d)", out_ld, C);
    hipStream_t s = (hipStream_t)stream;
    if (!DMC_ABL(option(OPT_CONV_ABLATE) & 1024))   // measurement build: the per-channel finalisation launches
    bn3d_stats_final_kernel<<<C, 256, 0, s>>>(partials, nblk, C, M, stats, running_mean, running_var, eps, momentum);
    int rc = check_launch("bn3d_stats_final");
    if (rc) return rc;
    bn3d_apply_kernel<<<bn3_blocks(M, C), 256, 0, s>>>((const bf16_t*)y, stats, gamma, beta, (bf16_t*)out, M, C, relu, out_ld);
    return check_launch("bn3d_apply");
}
int dmc_bn3d_bf16_fwd(const void* y, const float* partials, int nblk, const float* gamma, const float* beta, float* stats,
                      float* running_mean, float* running_var, void* out, long M, int C, int relu, float eps, float momentum,
                      dmc_stream_t stream) {
    return dmc_bn3d_bf16_fwd_ld(y, partials, nblk, gamma, beta, stats, running_mean, running_var, out, C, M, C, relu, eps, momentum, stream);
}

// dy (gradient of the convolution output), dgamma, dbeta from dout; scratch: dmc_bn3d_bf16_scratch_bytes(C).
// dout_ld: elements between consecutive pixels of dout (C for a dense tensor; the width of the concatenated tensor
// when dout is a channel slice of an Inception block's gradient, which is then read in place)
int dmc_bn3d_bf16_bwd(const void* dout, long dout_ld, const void* y, const float* stats, const float* gamma, const float* beta,
                      float* scratch, void* dy, float* dgamma, float* dbeta, long M, int C, int relu, dmc_stream_t stream) {
    if (!dout || !y || !stats || !gamma || !beta || !scratch || !dy || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_bn3d_bf16_bwd: null pointer");
    if (!bn3_ok(M, C) || dout_ld < C || dout_ld % 8 != 0) return fail(DMC_E_INVALID, "dmc_bn3d_bf16_bwd: unsupported shape M=%ld C=%d ld=%ld", M, C, dout_ld);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = bn3_blocks(M, C);
    float* coef = scratch + (size_t)BN3_MAXBLK * C * 2;
    bn3d_bwd_partial_kernel<<<nblk, 256, 0, s>>>((const bf16_t*)dout, (const bf16_t*)y, stats, gamma, beta, scratch, M, C, relu, dout_ld);
    int rc = check_launch("bn3d_bwd_partial");
    if (rc) return rc;
    if (!DMC_ABL(option(OPT_CONV_ABLATE) & 1024))
    bn3d_bwd_final_kernel<<<C, 256, 0, s>>>(scratch, nblk, C, M, dgamma, dbeta, coef);
    if ((rc = check_launch("bn3d_bwd_final"))) return rc;
    bn3d_bwd_apply_kernel<<<nblk, 256, 0, s>>>((const bf16_t*)dout, (const bf16_t*)y, stats, gamma, beta, coef, (bf16_t*)dy, M, C, relu, dout_ld);
    return check_launch("bn3d_bwd_apply");
}

// out = ((a + b) + c) + d on bf16 tensors of n elements (n % 8 == 0, 16-byte aligned), each sum rounded to bf16 as torch's additions are
int dmc_add4_bf16(const void* a, const void* b, const void* c, const void* d, void* out, long n, dmc_stream_t stream) {
    if (!a || !b || !c || !d || !out || n <= 0 || n % 8 != 0 || (((size_t)a | (size_t)b | (size_t)c | (size_t)d | (size_t)out) & 15))
        return fail(DMC_E_INVALID, "dmc_add4_bf16: null / unaligned pointer or n %% 8 != 0");
    const long n8 = n / 8;
    const long blocks = (n8 + 255) / 256;
    add4_bf16_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c,
                                                                                           (const bf16_t*)d, (bf16_t*)out, n8);
    return check_launch("add4_bf16");
}

}  // extern "C"
