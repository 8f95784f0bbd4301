// Issue-rate skeleton of the NHWC convolution's consumer loop (conv_nhwc.hip conv2_kernel): per "step" a wave
// reads READS 16-byte fragments from LDS and issues 64 v_mfma_f32_32x32x2_f32 on TM x TN accumulators; no
// global traffic, no barriers.  Variants: accumulators per wave, LDS reads per step, waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma32_loop.hip -o mfma32 && ./mfma32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int READS, int LDS_KB>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int steps) {
    __shared__ float lds[LDS_KB * 256];
    for (int i = threadIdx.x; i < LDS_KB * 256; i += 256) lds[i] = (float)(i & 15) * 0.0625f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float4 f[READS > 0 ? READS : 1];
    for (int r = 0; r < (READS > 0 ? READS : 1); ++r) f[r] = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int t = 0; t < steps; ++t) {
        const float* base = lds + (t & 1) * 2048;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int r = 0; r < READS / 4; ++r)
                f[(g & 1) * (READS / 8 > 0 ? READS / 8 : 1) + r % (READS / 8 > 0 ? READS / 8 : 1)] =
                    *reinterpret_cast<const float4*>(base + ((lane & 31) * 32 + ((2 * g + (lane >> 5) + r) & 7) * 4));
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 u = f[k % (READS > 0 ? READS : 1)];
                const float av = k & 1 ? u.x : u.y, bv = k & 2 ? u.z : u.w;
                acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[k % NACC], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int READS, int LDS_KB>
void run(const char* name, float* out, int blocks) {
    const int steps = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    loop_kernel<NACC, READS, LDS_KB><<<blocks, 256>>>(out, steps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    loop_kernel<NACC, READS, LDS_KB><<<blocks, 256>>>(out, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 4 * steps * 64 * 2.0 * 32 * 32 * 2;
    printf("%-44s blocks %4d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    run<4, 0, 16>("4 acc, no LDS reads, 1 wave/SIMD", out, 256);
    run<4, 0, 16>("4 acc, no LDS reads, 2 waves/SIMD", out, 512);
    run<4, 16, 64>("4 acc, 16 reads/step, 2 waves/SIMD", out, 512);
    run<4, 16, 64>("4 acc, 16 reads/step, 1 wave/SIMD (256 blocks)", out, 256);
    run<2, 12, 48>("2 acc, 12 reads/step, 3 waves/SIMD", out, 768);
    run<1, 8, 32>("1 acc, 8 reads/step, 4 waves/SIMD", out, 1024);
    run<4, 16, 64>("4 acc, 16 reads/step, 735 blocks (2.87/CU)", out, 735);
    return 0;
}
