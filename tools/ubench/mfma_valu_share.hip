// Do a wave of MFMAs and a wave of plain vector instructions on the SAME SIMD overlap?  One workgroup of 8 waves on one CU: waves w and
// w + 4 share a SIMD.  mode 0: waves 0-3 issue N independent v_mfma_f32_16x16x32_bf16 each (four accumulators, round robin);
// mode 1: waves 4-7 issue 4 N v_add_u32 each (four independent chains); mode 2: both.  Prints the clocks of each mode
// (s_memtime): both ~ max -> the vector ALU runs under the matrix pipe; both ~ sum -> they share the issue port.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_share.hip -o mfma_valu_share
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int mode, int n, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4] = {};
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, threadIdx.x, 8};
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 && (mode == 0 || mode == 2)) {
        for (int i = 0; i < n; i += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
        }
    }
    if (wave >= 4 && (mode == 1 || mode == 2)) {
        for (int i = 0; i < n; ++i) {
            asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(threadIdx.x));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)(v0 + v1 + v2 + v3);
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    const int n = 4096;
    for (int mode = 0; mode < 3; ++mode) {
        unsigned long long h[8];
        for (int r = 0; r < 2; ++r) { k<<<1, 512>>>(mode, n, out, sink); hipDeviceSynchronize(); }
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("mode %d (%s): MFMA wave %llu clocks (%.1f per MFMA), vector wave %llu clocks (%.2f per v_add)\n", mode,
               mode == 0 ? "MFMA only" : mode == 1 ? "VALU only" : "both", h[0], (double)h[0] / n, h[4], (double)h[4] / (4.0 * n));
    }
    return 0;
}
