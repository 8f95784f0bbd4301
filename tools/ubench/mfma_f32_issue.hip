// Micro-benchmark: what limits v_mfma_f32_16x16x4_f32 issue in the weight-gradient loop?
//   variant 0: MFMAs only (28 independent accumulators, operands in registers)
//   variant 1: + 19 ds_read_b32 gathers per group feeding the B operands (as the wgrad kernel)
//   variant 2: variant 1 with the gathers software-pipelined one group ahead
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_issue.hip -o mfma_ubench && ./mfma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 19, NA = 9;

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    __shared__ float lds[12032];
    for (int i = threadIdx.x; i < 12032; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
    int offB[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        int nn = 16 * t + j; nn = nn < 297 ? nn : 296;
        const int ci = nn / 9, tap = nn - ci * 9;
        offB[t] = ci * 364 + (tap / 3) * 36 + (tap % 3) + kq;
    }
    f32x4 accA[NA], accB[NB];
    for (int t = 0; t < NA; ++t) accA[t] = (f32x4){0, 0, 0, 0};
    for (int t = 0; t < NB; ++t) accB[t] = (f32x4){0, 0, 0, 0};
    float a0 = 1.0f + lane * 1e-3f, a1 = 0.5f;
    float b[NB], bn[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) { b[t] = 0.25f * t; bn[t] = b[t]; }
    if (VARIANT == 2) {
#pragma unroll
        for (int t = 0; t < NB; ++t) bn[t] = lds[offB[t] + wave * 36];
    }
    for (int it = 0; it < iters; ++it) {
        if (VARIANT >= 3) __syncthreads();
        if (VARIANT == 5) {           // ~25 dependent-ish integer VALU ops x 24 "DMA address" computations
            unsigned v = lane + it;
#pragma unroll 1
            for (int m = 0; m < 24; ++m) {
                unsigned L = 64u * (m * 8 + wave) + lane;
                unsigned plane = L / 364u, rem = L - plane * 364u;
                unsigned row = rem / 36u, col = rem - row * 36u;
                bool ok = plane < 33 && row < 10 && col < 34 && (row + it) < 100000u;
                unsigned long long base = plane < 2 ? 1000ull + plane * 50176ull : plane < 5 ? 9000ull + (plane - 2) * 50176ull : 777ull + (plane - 5) * 50176ull;
                unsigned long long src = ok ? base + (unsigned long long)(row * 224u + col) : 0ull;
                v += (unsigned)(src >> 3);
            }
            a0 += (float)(v & 1) * 1e-20f;
        }
        if (VARIANT == 4) {           // + the per-tile global A-operand requests of the real kernel
            a0 += out[(it * 37 + lane) & 1023] * 1e-9f;
            a1 += out[(it * 53 + lane + 2048) & 4095] * 1e-9f;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int xb = wave * 36 + g * 4;
            if (VARIANT == 1 || VARIANT >= 3) {
#pragma unroll
                for (int t = 0; t < NB; ++t) b[t] = lds[offB[t] + xb];
            }
            if (VARIANT == 2) {
#pragma unroll
                for (int t = 0; t < NB; ++t) b[t] = bn[t];
                const int xn = wave * 36 + ((g + 1) & 7) * 4;
#pragma unroll
                for (int t = 0; t < NB; ++t) bn[t] = lds[offB[t] + xn];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) accA[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t], accA[t], 0, 0, 0);
            accA[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[18], accA[8], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NB; ++t) accB[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t], accB[t], 0, 0, 0);
        }
    }
    float s = 0;
    for (int t = 0; t < NA; ++t) s += accA[t][0] + accA[t][3];
    for (int t = 0; t < NB; ++t) s += accB[t][1] + accB[t][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(const char* name, int threads, int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V><<<blocks, threads>>>(out, 10);
    hipEventRecord(e0);
    k<V><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * (threads / 64) * iters * 8 * 28;
    const double flops = mfma * 2 * 16 * 16 * 4;
    printf("%-44s %4d thr x %4d blk: %.3f ms  %.1f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.1GHz)\n", name, threads, blocks, ms,
           flops / ms / 1e9, ms * 1e-3 * 2.1e9 / (mfma / 1024.0));
    hipFree(out);
}

int main() {
    run<0>("MFMA only", 256, 256);
    run<0>("MFMA only", 512, 256);
    run<0>("MFMA only", 512, 512);
    run<1>("MFMA + 19 LDS gathers/group", 256, 256);
    run<1>("MFMA + 19 LDS gathers/group", 512, 256);
    run<2>("MFMA + gathers pipelined 1 group ahead", 256, 256);
    run<2>("MFMA + gathers pipelined 1 group ahead", 512, 256);
    run<3>("MFMA + gathers + barrier per 8 groups", 512, 256);
    run<3>("MFMA + gathers + barrier per 8 groups", 256, 512);
    run<4>("  ... + dependent global loads per tile", 512, 256);
    run<5>("MFMA + gathers + barrier + DMA address math", 512, 256);
    return 0;
}
