// Discriminator block tail for gfx950: LeakyReLU(0.2) -> Dropout2d(0.25) -> BatchNorm2d(eps 0.8).
//
// Reference behaviour: discriminator_block / discriminator_block2,
// code/dmcnet_GAN/model.py:254-279 (`nn.BatchNorm2d(out_filters, 0.8)` => eps = 0.8,
// momentum 0.1).  These are HBM-bound streaming kernels: the forward reads x twice (statistics
// pass + normalise pass; the second read is an L2/Infinity-Cache hit for the block sizes of
// Discriminator3) and writes y once.
#include "dmc_common.h"

using namespace dmc;

namespace {

constexpr int SPLIT = 64;   // partial reductions per channel

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : 0.2f * v; }

// block (256) reduces two quantities over its share of (n, hw) for channel blockIdx.x
// MODE 0: (sum z, sum z^2) with z = keep * lrelu(x)
// MODE 1: (sum dy, sum dy * zhat)
template <int MODE>
__global__ __launch_bounds__(256) void tail_partial_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ keep,
                                                           const float* __restrict__ dy,
                                                           const float* __restrict__ stats,
                                                           double* __restrict__ scratch, int N,
                                                           int C, int HW) {
    __shared__ double sm[2][4];
    const int c = blockIdx.x, sp = blockIdx.y;
    const long total = (long)N * HW;
    const long per = (total + SPLIT - 1) / SPLIT;
    const long lo = sp * per, hi = (lo + per < total) ? lo + per : total;
    float mean = 0.f, invstd = 0.f;
    if (MODE == 1) { mean = stats[c]; invstd = stats[C + c]; }
    double a = 0.0, b = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
        const int n = (int)(i / HW);
        const int hw = (int)(i - (long)n * HW);
        const size_t idx = ((size_t)n * C + c) * HW + hw;
        const float k = keep ? keep[n * C + c] : 1.f;
        const float z = k * lrelu02(x[idx]);
        if (MODE == 0) {
            a += (double)z;
            b += (double)z * (double)z;
        } else {
            const float g = dy[idx];
            a += (double)g;
            b += (double)g * (double)((z - mean) * invstd);
        }
    }
    a = wsum(a);
    b = wsum(b);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = a; sm[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        scratch[((size_t)sp * C + c) * 2 + 0] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        scratch[((size_t)sp * C + c) * 2 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    }
}

__global__ void tail_stats_final_kernel(const double* __restrict__ scratch, float* __restrict__ stats,
                                        float* __restrict__ running_mean,
                                        float* __restrict__ running_var, int C, long count,
                                        int training, float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (!training) {
        stats[c] = running_mean[c];
        stats[C + c] = rsqrtf(running_var[c] + eps);
        return;
    }
    double s = 0.0, ss = 0.0;
    for (int sp = 0; sp < SPLIT; ++sp) {
        s += scratch[((size_t)sp * C + c) * 2 + 0];
        ss += scratch[((size_t)sp * C + c) * 2 + 1];
    }
    const double mean = s / (double)count;
    double var = ss / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
}

__global__ __launch_bounds__(256) void tail_apply_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ keep,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const float* __restrict__ stats,
                                                         float* __restrict__ y, int C, int HW,
                                                         int use_bn) {
    const int nc = blockIdx.x;            // n * C + c (grid.x: no 65,535 limit on frames x channels)
    const int c = nc % C;
    const float k = keep ? keep[nc] : 1.f;
    float scale = 1.f, shift = 0.f;
    if (use_bn) {
        scale = stats[C + c] * gamma[c];
        shift = beta[c] - stats[c] * scale;
    }
    const float* xp = x + (size_t)nc * HW;
    float* yp = y + (size_t)nc * HW;
    const int i0 = (blockIdx.y * 256 + threadIdx.x) * 4;
    if ((HW & 3) == 0) {
        if (i0 < HW) {
            const float4 v = *reinterpret_cast<const float4*>(xp + i0);
            *reinterpret_cast<float4*>(yp + i0) =
                make_float4(fmaf(k * lrelu02(v.x), scale, shift), fmaf(k * lrelu02(v.y), scale, shift),
                            fmaf(k * lrelu02(v.z), scale, shift), fmaf(k * lrelu02(v.w), scale, shift));
        }
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j < HW) yp[i0 + j] = fmaf(k * lrelu02(xp[i0 + j]), scale, shift);
    }
}

__global__ void tail_bwd_final_kernel(const double* __restrict__ scratch, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int sp = 0; sp < SPLIT; ++sp) {
        s += scratch[((size_t)sp * C + c) * 2 + 0];
        ss += scratch[((size_t)sp * C + c) * 2 + 1];
    }
    dbeta[c] = (float)s;
    dgamma[c] = (float)ss;
}

__global__ __launch_bounds__(256) void tail_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ keep, const float* __restrict__ gamma,
    const float* __restrict__ stats, const float* __restrict__ dgamma,
    const float* __restrict__ dbeta, const float* __restrict__ dy, float* __restrict__ dx, int C,
    int HW, float inv_count, int use_bn) {
    const int nc = blockIdx.x;
    const int c = nc % C;
    const float k = keep ? keep[nc] : 1.f;
    float mean = 0.f, invstd = 1.f, gs = 1.f, mb = 0.f, mg = 0.f;
    if (use_bn) {
        mean = stats[c];
        invstd = stats[C + c];
        gs = gamma[c] * invstd;
        mb = dbeta[c] * inv_count;
        mg = dgamma[c] * inv_count;
    }
    const float* xp = x + (size_t)nc * HW;
    const float* gp = dy + (size_t)nc * HW;
    float* dp = dx + (size_t)nc * HW;
    const int i0 = (blockIdx.y * 256 + threadIdx.x) * 4;
    for (int j = 0; j < 4; ++j) {
        if (i0 + j < HW) {
            const float xv = xp[i0 + j];
            float dz = gp[i0 + j];
            if (use_bn) {
                const float zhat = (k * lrelu02(xv) - mean) * invstd;
                dz = gs * (dz - mb - zhat * mg);
            }
            dp[i0 + j] = dz * k * (xv > 0.f ? 1.f : 0.2f);
        }
    }
}

}  // namespace

extern "C" {

size_t dmc_disc_tail_stats_bytes(int C) {
    // [2*C floats (mean, invstd)] padded to 16 B, then SPLIT * C * 2 doubles of scratch
    const size_t head = (((size_t)2 * C * sizeof(float)) + 15) / 16 * 16;
    return head + (size_t)SPLIT * C * 2 * sizeof(double);
}

static double* scratch_of(float* stats, int C) {
    const size_t head = (((size_t)2 * C * sizeof(float)) + 15) / 16 * 16;
    return reinterpret_cast<double*>(reinterpret_cast<char*>(stats) + head);
}

int dmc_disc_tail_fwd(const float* x, const float* keep, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float* y, float* stats, int N,
                      int C, int H, int W, int use_bn, int training, float eps, float momentum,
                      dmc_stream_t stream) {
    if (!x || !y) return fail(DMC_E_INVALID, "dmc_disc_tail_fwd: null pointer");
    if (use_bn && (!gamma || !beta || !running_mean || !running_var || !stats))
        return fail(DMC_E_INVALID, "dmc_disc_tail_fwd: null BatchNorm pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(DMC_E_INVALID, "dmc_disc_tail_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    int rc;
    if (use_bn) {
        double* scratch = scratch_of(stats, C);
        if (training) {
            tail_partial_kernel<0><<<dim3(C, SPLIT), 256, 0, s>>>(x, keep, nullptr, nullptr, scratch, N, C, HW);
            if ((rc = check_launch("disc_tail_partial"))) return rc;
        }
        tail_stats_final_kernel<<<(C + 63) / 64, 64, 0, s>>>(scratch, stats, running_mean, running_var, C,
                                                             (long)N * HW, training, eps, momentum);
        if ((rc = check_launch("disc_tail_stats_final"))) return rc;
    }
    tail_apply_kernel<<<dim3(N * C, (HW + 1023) / 1024), 256, 0, s>>>(x, keep, gamma, beta, stats, y, C, HW, use_bn);
    return check_launch("disc_tail_apply");
}

int dmc_disc_tail_bwd(const float* x, const float* keep, const float* gamma, float* stats,
                      const float* dy, float* dx, float* dgamma, float* dbeta, int N, int C, int H,
                      int W, int use_bn, dmc_stream_t stream) {
    if (!x || !dy || !dx) return fail(DMC_E_INVALID, "dmc_disc_tail_bwd: null pointer");
    if (use_bn && (!gamma || !stats || !dgamma || !dbeta))
        return fail(DMC_E_INVALID, "dmc_disc_tail_bwd: null BatchNorm pointer");
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(DMC_E_INVALID, "dmc_disc_tail_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    int rc;
    if (use_bn) {
        double* scratch = scratch_of(stats, C);
        tail_partial_kernel<1><<<dim3(C, SPLIT), 256, 0, s>>>(x, keep, dy, stats, scratch, N, C, HW);
        if ((rc = check_launch("disc_tail_bwd_partial"))) return rc;
        tail_bwd_final_kernel<<<(C + 63) / 64, 64, 0, s>>>(scratch, dgamma, dbeta, C);
        if ((rc = check_launch("disc_tail_bwd_final"))) return rc;
    }
    tail_bwd_apply_kernel<<<dim3(N * C, (HW + 1023) / 1024), 256, 0, s>>>(
        x, keep, gamma, stats, dgamma, dbeta, dy, dx, C, HW, 1.f / ((float)N * (float)HW), use_bn);
    return check_launch("disc_tail_bwd_apply");
}

}  // extern "C"
