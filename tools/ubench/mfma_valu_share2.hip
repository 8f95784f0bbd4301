// One SIMD's issue: a wave of back-to-back MFMAs next to 0 .. 3 waves of INDEPENDENT vector instructions (four chains each).
// 1,024 threads on one CU: wave w runs on SIMD w % 4.  Waves 0-3: N v_mfma_f32_16x16x32_bf16 (four accumulators); of the
// other twelve, the first 4 V (V = 0 .. 3 per SIMD) issue 4 N v_add_u32.  Clocks per MFMA and per v_add for every V.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_share2.hip -o mfma_valu_share2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(int mfma, int nv, int n, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {};
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, threadIdx.x, 8};
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 && mfma) {
        for (int i = 0; i < n; i += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
        }
    }
    if (wave >= 4 && wave < 4 + 4 * nv) {
        for (int i = 0; i < n / 2; ++i) {
            asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                         "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(threadIdx.x));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 128); hipMalloc(&sink, 4096);
    const int n = 4096;
    for (int mfma = 0; mfma < 2; ++mfma)
        for (int nv = 0; nv <= 3; ++nv) {
            if (!mfma && !nv) continue;
            unsigned long long h[16];
            for (int r = 0; r < 2; ++r) { k<<<1, 1024>>>(mfma, nv, n, out, sink); hipDeviceSynchronize(); }
            hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
            printf("MFMA wave %s, %d vector waves per SIMD: %.1f clocks per MFMA, %.2f clocks per v_add of a vector wave (%.2f per v_add of the SIMD)\n",
                   mfma ? "on " : "off", nv, mfma ? (double)h[0] / n : 0.0, nv ? (double)h[4] / (4.0 * n) : 0.0, nv ? (double)h[4] / (4.0 * n) / nv : 0.0);
        }
    return 0;
}
