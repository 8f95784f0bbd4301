// Flow-reconstruction MSE and TSN consensus + cross-entropy for gfx950.
//
// Reference behaviour: nn.MSELoss()(gen_flow, input_flow)  code/dmcnet/train.py:167,245;
// output.view(-1,S,C).mean(1) + CrossEntropyLoss           code/dmcnet/train.py:239-241.
#include <atomic>
#include <string.h>

#include "dmc_common.h"

using namespace dmc;

namespace {

constexpr int MSE_BLOCKS = 2048;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

// HBM-bound streaming reduction: 16 B per lane per load, grid-stride, fp32 per-lane partials
// promoted to double for the cross-lane / cross-block sums (fixed order -> deterministic).
__global__ __launch_bounds__(256) void flow_mse_partial_kernel(const float* __restrict__ a,
                                                               const float* __restrict__ b,
                                                               double* __restrict__ partials,
                                                               size_t numel) {
    __shared__ double sm[4];
    const size_t n4 = numel / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a4[i], y = b4[i];
        const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        s0 = fmaf(d0, d0, s0); s1 = fmaf(d1, d1, s1); s2 = fmaf(d2, d2, s2); s3 = fmaf(d3, d3, s3);
    }
    double s = (double)s0 + (double)s1 + (double)s2 + (double)s3;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < numel; ++i) { const double d = (double)a[i] - (double)b[i]; s += d * d; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void flow_mse_final_kernel(const double* __restrict__ partials, int nblocks,
                                      float* __restrict__ loss_out, size_t numel) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss_out = (float)((sm[0] + sm[1] + sm[2] + sm[3]) / (double)numel);
}

__global__ __launch_bounds__(256) void flow_mse_bwd_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ grad_loss,
                                                           float* __restrict__ ga, size_t numel) {
    const float scale = 2.f * (*grad_loss) / (float)numel;
    const size_t n4 = numel / 4;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* g4 = reinterpret_cast<float4*>(ga);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = a4[i], y = b4[i];
        g4[i] = make_float4((x.x - y.x) * scale, (x.y - y.y) * scale, (x.z - y.z) * scale,
                            (x.w - y.w) * scale);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < numel; ++i) ga[i] = (a[i] - b[i]) * scale;
}

// One workgroup of 16 waves; wave w handles clips w, w+16, ...; lanes cover the classes.
__global__ __launch_bounds__(1024) void consensus_ce_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ target,
                                                            float* __restrict__ consensus,
                                                            float* __restrict__ loss_out,
                                                            float* __restrict__ grad, int B, int S,
                                                            int C) {
    __shared__ float wave_loss[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_s = 1.f / (float)S, inv_n = 1.f / ((float)B * (float)S);
    float loss = 0.f;
    for (int b = wave; b < B; b += 16) {
        const float* lb = logits + (size_t)b * S * C;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) {
            float a = 0.f;
            for (int s = 0; s < S; ++s) a += lb[(size_t)s * C + c];
            a *= inv_s;
            consensus[(size_t)b * C + c] = a;
            mx = fmaxf(mx, a);
        }
        mx = __shfl(wave_max(mx), 0, 64);
        float se = 0.f;
        for (int c = lane; c < C; c += 64) {
            float a = 0.f;
            for (int s = 0; s < S; ++s) a += lb[(size_t)s * C + c];
            se += __expf(a * inv_s - mx);
        }
        se = __shfl(wave_sum(se), 0, 64);
        const float lse = __logf(se) + mx;
        // a label outside [0, C) never indexes memory: the clip's loss and gradient become NaN (torch's
        // CrossEntropyLoss device-asserts instead; an asynchronous C ABI cannot raise)
        const int64_t t64 = target[b];
        const bool t_ok = t64 >= 0 && t64 < (int64_t)C;
        const int t = t_ok ? (int)t64 : 0;
        if (lane == 0) {
            float a = 0.f;
            for (int s = 0; s < S; ++s) a += lb[(size_t)s * C + t];
            loss += t_ok ? lse - a * inv_s : NAN;
        }
        if (grad != nullptr) {
            for (int c = lane; c < C; c += 64) {
                float a = 0.f;
                for (int s = 0; s < S; ++s) a += lb[(size_t)s * C + c];
                const float p = __expf(a * inv_s - lse);
                const float g = t_ok ? (p - (c == t ? 1.f : 0.f)) * inv_n : NAN;
                for (int s = 0; s < S; ++s) grad[((size_t)b * S + s) * C + c] = g;
            }
        }
    }
    if (lane == 0) wave_loss[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += wave_loss[w];
        *loss_out = s / (float)B;
    }
}

const char* const OPTION_NAMES[OPT_COUNT] = {"gen_layer_path", "gen_gather", "gen_fuse45", "gen_wgrad_path",
                                             "gen_fuse_fwd", "gen_fuse_bwd", "gen_frames", "conv_path", "conv_cfg", "conv_ablate", "gen_ablate", "conv_arith", "conv3d_wgrad", "gen_x3", "gen_wino", "gen_stagger", "gen_fused", "grid_reserve_cus"};
std::atomic<int> g_options[OPT_COUNT] = {{1}, {1}, {1}, {5}, {1}, {1}, {0}, {1}, {0}, {0}, {0}, {1}, {2}, {2}, {768}, {0}, {1}, {0}};
int option_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, OPTION_NAMES[i]) == 0) return i;
    return -1;
}

}  // namespace

namespace dmc {
int option(Option o) { return !MEASURE_BUILD && measure_only(o) ? 0 : g_options[o].load(std::memory_order_relaxed); }
}  // namespace dmc

// Empty kernel with a recognisable name: bench.py brackets its timed region with it so that a
// rocprofv3 kernel trace can be cut to that region (tools/rocprof_region.py).
__global__ void dmc_profile_mark_kernel() {}

extern "C" {

size_t dmc_flow_mse_partials_bytes(void) { return (size_t)MSE_BLOCKS * sizeof(double); }

int dmc_flow_mse_fwd(const float* gen_flow, const float* flow, float* loss_out, float* partials,
                     size_t numel, dmc_stream_t stream) {
    if (!gen_flow || !flow || !loss_out || !partials || numel == 0)
        return fail(DMC_E_INVALID, "dmc_flow_mse_fwd: null pointer or empty tensor");
    hipStream_t s = (hipStream_t)stream;
    size_t want = (numel / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > MSE_BLOCKS ? MSE_BLOCKS : want));
    flow_mse_partial_kernel<<<blocks, 256, 0, s>>>(gen_flow, flow, (double*)partials, numel);
    int rc = check_launch("flow_mse_partial");
    if (rc) return rc;
    flow_mse_final_kernel<<<1, 256, 0, s>>>((const double*)partials, blocks, loss_out, numel);
    return check_launch("flow_mse_final");
}

int dmc_flow_mse_bwd(const float* gen_flow, const float* flow, const float* grad_loss,
                     float* grad_gen, size_t numel, dmc_stream_t stream) {
    if (!gen_flow || !flow || !grad_loss || !grad_gen || numel == 0)
        return fail(DMC_E_INVALID, "dmc_flow_mse_bwd: null pointer or empty tensor");
    size_t want = (numel / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    flow_mse_bwd_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(gen_flow, flow, grad_loss, grad_gen, numel);
    return check_launch("flow_mse_bwd");
}

int dmc_consensus_ce_fwd_bwd(const float* logits, const int64_t* target, float* consensus,
                             float* loss_out, float* grad_logits, int B, int S, int C,
                             dmc_stream_t stream) {
    if (!logits || !target || !consensus || !loss_out)
        return fail(DMC_E_INVALID, "dmc_consensus_ce_fwd_bwd: null pointer");
    if (B <= 0 || S <= 0 || C <= 0) return fail(DMC_E_INVALID, "dmc_consensus_ce_fwd_bwd: bad shape");
    consensus_ce_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(logits, target, consensus, loss_out,
                                                              grad_logits, B, S, C);
    return check_launch("consensus_ce");
}

int dmc_profile_mark(dmc_stream_t stream) {
    dmc_profile_mark_kernel<<<1, 64, 0, (hipStream_t)stream>>>();
    return check_launch("dmc_profile_mark");
}

int dmc_version(void) { return 200; }   // 0.2.0

int dmc_set_option(const char* name, int value) {
    const int i = option_index(name);
    if (i < 0) return fail(DMC_E_INVALID, "dmc_set_option: unknown option '%s'", name ? name : "(null)");
    if (!MEASURE_BUILD && measure_only(i))
        return fail(DMC_E_INVALID, "dmc_set_option: '%s' switches parts of a kernel off (results wrong) and exists only in the -DDMC_MEASURE build", name);
    if (i == OPT_GRID_RESERVE_CUS && (value < 0 || value > 128))
        return fail(DMC_E_INVALID, "dmc_set_option: grid_reserve_cus = %d (0 .. 128 CUs may be left to other streams)", value);
    if (!MEASURE_BUILD && !product_value(i, value))
        return fail(DMC_E_INVALID, "dmc_set_option: %s = %d selects a kernel variant that lost its A/B measurement; it is compiled into the "
                    "-DDMC_MEASURE build only", name, value);
    g_options[i].store(value, std::memory_order_relaxed);
    return DMC_OK;
}
int dmc_get_option(const char* name) {
    if (name && !strcmp(name, "measure_build")) return MEASURE_BUILD ? 1 : 0;      // read-only: which build this library is
    const int i = option_index(name);
    return i < 0 ? -1 : (!MEASURE_BUILD && measure_only(i)) ? 0 : g_options[i].load(std::memory_order_relaxed);
}
const char* dmc_last_error(void) { return err_buf(); }

}  // extern "C"
