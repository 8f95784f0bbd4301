// Micro-benchmark / probe for v_mfma_f32_4x4x1_16b_f32 as the engine of the generator's 3x3 layers.
//   part 1: operand layout probe (which lane's A / B element lands in D[i] of lane l)
//   part 2: issue-rate skeleton of the layer loop: per 4-channel chunk, 4 segments x 12 K-steps x
//           NT row tiles, B operand by ds_read_b32 (one pixel per lane), A (weights) in registers
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_4x4_conv.hip -o mfma44 && ./mfma44
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out) {
    const int l = threadIdx.x;
    f32x4 z = {0, 0, 0, 0};
    f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);
    f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);
    for (int i = 0; i < 4; ++i) { out[i * 64 + l] = da[i]; out[256 + i * 64 + l] = db[i]; }
}

template <int NT, int SEGS>
__global__ __launch_bounds__(512, 2) void skel(float* out, int chunks) {
    __shared__ float lds[2 * 4 * 10 * 256];
    for (int i = threadIdx.x; i < 2 * 4 * 10 * 256; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    f32x4 acc[SEGS][NT];
    for (int s = 0; s < SEGS; ++s) for (int t = 0; t < NT; ++t) acc[s][t] = (f32x4){0, 0, 0, 0};
    const float* wl = lds + 1000;
    for (int ch = 0; ch < chunks; ++ch) {
        const float* buf = lds + (ch & 1) * 10240;
        float wr[12][NT];
#pragma unroll
        for (int k = 0; k < 12; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t) wr[k][t] = wl[(ch & 3) * 300 + (k * NT + t) * 4 + (lane & 3)];
#pragma unroll
        for (int s = 0; s < SEGS; ++s) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int cc = k / 3, dy = k % 3;
                const float b = buf[cc * 2560 + (r + dy) * 256 + s * 64 + lane];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[k][t], b, acc[s][t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float sum = 0;
    for (int s = 0; s < SEGS; ++s) for (int t = 0; t < NT; ++t) sum += acc[s][t][0] + acc[s][t][1] + acc[s][t][2] + acc[s][t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int NT, int SEGS>
void run(int blocks) {
    float* out; hipMalloc(&out, (size_t)blocks * 512 * 4);
    const int chunks = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    skel<NT, SEGS><<<blocks, 512>>>(out, 10);
    hipEventRecord(e0);
    skel<NT, SEGS><<<blocks, 512>>>(out, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 8 * chunks * SEGS * 12 * NT;
    printf("NT=%d SEGS=%d blocks=%d: %.3f ms  %.1f TFLOP/s  (%.2f cyc/MFMA/SIMD @2.4GHz)\n", NT, SEGS, blocks, ms,
           mfma * 2 * 256 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
    hipFree(out);
}

int main() {
    float* d; hipMalloc(&d, 512 * 4);
    probe<<<1, 64>>>(d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("A source lane (+1) for D[i][lane], lanes 0..19:\n");
    for (int i = 0; i < 4; ++i) { printf(" i=%d:", i); for (int l = 0; l < 20; ++l) printf(" %2.0f", h[i * 64 + l]); printf(" ... l=63: %2.0f\n", h[i * 64 + 63]); }
    printf("B source lane (+1) for D[i][lane], lanes 0..19:\n");
    for (int i = 0; i < 4; ++i) { printf(" i=%d:", i); for (int l = 0; l < 20; ++l) printf(" %2.0f", h[256 + i * 64 + l]); printf(" ... l=63: %2.0f\n", h[256 + i * 64 + 63]); }
    run<6, 4>(256); run<6, 4>(512);
    run<5, 4>(256); run<3, 4>(256); run<2, 4>(256); run<2, 4>(512);
    run<6, 2>(512);
    return 0;
}
