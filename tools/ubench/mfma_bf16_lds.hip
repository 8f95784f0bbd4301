// v_mfma_f32_32x32x16_bf16 fed like the bf16x3 convolution's inner loop: per "block" a wave reads RD 16-byte fragments
// from LDS (random data), optionally splits 8 of the values (44 VALU) and issues 12 MFMAs on 2 accumulators.
// Variants: random vs constant data (power), LDS reads on/off, split on/off, barrier per 2 blocks on/off.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_lds.hip -o mfma_bf16_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mf(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int LDSRD, int SPLIT, int BAR>
__global__ __launch_bounds__(512) void loop_kernel(const unsigned* __restrict__ src, float* out, int steps) {
    extern __shared__ __attribute__((aligned(1024))) unsigned lds[];
    for (int i = threadIdx.x; i < 20480; i += 512) lds[i] = src[(i * 7 + blockIdx.x) & 0xffff];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    u32x4 w[2][3], x[3];
    for (int j = 0; j < 2; ++j) for (int s = 0; s < 3; ++s) w[j][s] = *reinterpret_cast<const u32x4*>(src + ((lane * 4 + j * 256 + s * 512 + wave * 2048) & 0xfffc));
    for (int s = 0; s < 3; ++s) x[s] = *reinterpret_cast<const u32x4*>(src + ((lane * 4 + s * 768 + wave * 1024 + 4096) & 0xfffc));
    const unsigned* L = lds + lane * 4 + (wave & 3) * 512;
    for (int t = 0; t < steps; ++t) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x4 r0, r1;
            if (LDSRD) {
                r0 = *reinterpret_cast<const f32x4*>(L + kb * 4096 + 256);
                r1 = *reinterpret_cast<const f32x4*>(L + kb * 4096 + 2304);
            }
            // weight slice 0 products
            for (int xs = 2; xs >= 0; --xs) for (int j = 0; j < 2; ++j) acc[j] = mf(w[j][0], x[xs], acc[j]);
            if (LDSRD) for (int j = 0; j < 2; ++j) w[j][0] = *reinterpret_cast<const u32x4*>(L + kb * 4096 + 8192 + j * 1024);
            for (int xs = 1; xs >= 0; --xs) for (int j = 0; j < 2; ++j) acc[j] = mf(w[j][1], x[xs], acc[j]);
            if (LDSRD) for (int j = 0; j < 2; ++j) w[j][1] = *reinterpret_cast<const u32x4*>(L + kb * 4096 + 10240 + j * 1024);
            for (int j = 0; j < 2; ++j) acc[j] = mf(w[j][2], x[0], acc[j]);
            if (LDSRD) for (int j = 0; j < 2; ++j) w[j][2] = *reinterpret_cast<const u32x4*>(L + kb * 4096 + 12288 + j * 1024);
            if (SPLIT && LDSRD) {
                const float v[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                unsigned u0[8], u1[8], u2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    u0[e] = __float_as_uint(v[e]);
                    const float q1 = v[e] - __uint_as_float(u0[e] & 0xffff0000u);
                    u1[e] = __float_as_uint(q1);
                    u2[e] = __float_as_uint(q1 - __uint_as_float(u1[e] & 0xffff0000u));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[0][e] = __builtin_amdgcn_perm(u0[2 * e + 1], u0[2 * e], 0x07060302u);
                    x[1][e] = __builtin_amdgcn_perm(u1[2 * e + 1], u1[2 * e], 0x07060302u);
                    x[2][e] = __builtin_amdgcn_perm(u2[2 * e + 1], u2[2 * e], 0x07060302u);
                }
            } else if (LDSRD) {
                x[0] = __builtin_bit_cast(u32x4, r0); x[1] = __builtin_bit_cast(u32x4, r1);
            }
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int LDSRD, int SPLIT, int BAR>
void run(const char* name, const unsigned* src, float* out, int blocks) {
    const int steps = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&loop_kernel<LDSRD, SPLIT, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    loop_kernel<LDSRD, SPLIT, BAR><<<blocks, 512, 81920>>>(src, out, steps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    loop_kernel<LDSRD, SPLIT, BAR><<<blocks, 512, 81920>>>(src, out, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double n_mfma = (double)blocks * 8 * steps * 24;
    printf("%-58s blocks %4d  %8.3f ms  %7.1f TFLOP/s (bf16)  %5.1f cycles@2.4GHz per MFMA per SIMD\n", name, blocks, ms,
           n_mfma * 2.0 * 32 * 32 * 16 / ms / 1e9, ms * 2.4e6 / (n_mfma / 1024.0));
}

int main(int argc, char** argv) {
    const int constant = argc > 1 ? atoi(argv[1]) : 0;
    unsigned* h = (unsigned*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) {
        // random bf16 pairs / fp32 values of moderate magnitude
        const unsigned m = constant ? 0x3f803f80u : (((unsigned)rand() & 0x807f) | 0x3f00) << 16 | (((unsigned)rand() & 0x807f) | 0x3f00) | ((unsigned)rand() & 0x7f0000);
        h[i] = m;
    }
    unsigned* src; float* out;
    hipMalloc(&src, 65536 * 4); hipMalloc(&out, 2048 * 512 * sizeof(float));
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    printf("data: %s\n", constant ? "constant" : "random");
    run<0, 0, 0>("registers only (no LDS reads)", src, out, 256);
    run<1, 0, 0>("LDS fragment reads (8 per 12 MFMA), no split", src, out, 256);
    run<1, 1, 0>("LDS reads + split (44 VALU per 12 MFMA)", src, out, 256);
    run<1, 1, 1>("LDS reads + split + barrier per 24 MFMA", src, out, 256);
    run<1, 1, 1>("same, 735 workgroups (2.87 per CU)", src, out, 735);
    return 0;
}
