// Rate of v_mfma_f32_32x32x16_bf16 under sustained load (the bf16x3 convolutions' instruction): accumulators
// per wave, waves per SIMD, with / without VALU work of the split (11 instructions per value pair) between the MFMAs.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_loop.hip -o mfma_bf16 && ./mfma_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int VALU>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int steps, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    u32x4 x = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    u32x4 y = x;
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed + (float)(threadIdx.x + e);
    for (int t = 0; t < steps; ++t) {
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[k % NACC], 0, 0, 0);
            if (VALU > 0 && k % (24 / VALU) == 0) {
                // the split of one pair of values: 11 VALU instructions
                unsigned u0[2], u1[2], u2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float a0 = v[(k + e) & 7];
                    u0[e] = __float_as_uint(a0);
                    const float r1 = a0 - __uint_as_float(u0[e] & 0xffff0000u);
                    u1[e] = __float_as_uint(r1);
                    u2[e] = __float_as_uint(r1 - __uint_as_float(u1[e] & 0xffff0000u));
                }
                y[k & 3] = __builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u) ^ __builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u) ^ __builtin_amdgcn_perm(u2[1], u2[0], 0x07060302u);
                v[k & 7] += 1.0f;
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int VALU>
void run(const char* name, float* out, int blocks) {
    const int steps = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    loop_kernel<NACC, VALU><<<blocks, 256>>>(out, steps, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    loop_kernel<NACC, VALU><<<blocks, 256>>>(out, steps, 1.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double n_mfma = (double)blocks * 4 * steps * 24;
    const double flop = n_mfma * 2.0 * 32 * 32 * 16;
    // cycles per MFMA per SIMD at 2.4 GHz: (ms * 2.4e6 cycles) / (MFMAs per SIMD)
    const double per_simd = n_mfma / 1024.0;
    printf("%-52s blocks %4d  %8.3f ms  %7.1f TFLOP/s  %5.1f cycles@2.4GHz per MFMA\n", name, blocks, ms, flop / ms / 1e9, ms * 2.4e6 / per_simd);
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    run<4, 0>("4 acc, MFMA only, 1 wave/SIMD", out, 256);
    run<2, 0>("2 acc, MFMA only, 1 wave/SIMD", out, 256);
    run<1, 0>("1 acc, MFMA only, 1 wave/SIMD", out, 256);
    run<2, 0>("2 acc, MFMA only, 2 waves/SIMD", out, 512);
    run<4, 0>("4 acc, MFMA only, 2 waves/SIMD", out, 512);
    run<1, 0>("1 acc, MFMA only, 2 waves/SIMD", out, 512);
    run<2, 4>("2 acc, 4 pair-splits (44 VALU) per 24 MFMA, 2 w/SIMD", out, 512);
    run<2, 8>("2 acc, 8 pair-splits (88 VALU) per 24 MFMA, 2 w/SIMD", out, 512);
    run<4, 8>("4 acc, 8 pair-splits (88 VALU) per 24 MFMA, 2 w/SIMD", out, 512);
    run<4, 12>("4 acc, 12 pair-splits (132 VALU) per 24 MFMA, 2 w/SIMD", out, 512);
    run<4, 8>("4 acc, 8 pair-splits per 24 MFMA, 1 w/SIMD", out, 256);
    run<4, 8>("4 acc, 8 pair-splits per 24 MFMA, 4 w/SIMD", out, 1024);
    return 0;
}
