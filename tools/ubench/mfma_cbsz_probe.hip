// Probe: A-operand block broadcast of v_mfma_f32_4x4x1_16b_f32 (cbsz = 4, abid = t): do all 16 blocks take block t's A values
// (lanes 4t .. 4t+3)?  Expected print per t: D[i] at every lane = (4 t + i) * (lane + 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int T>
__device__ void one(float a, float b, float* out) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, T, 0);
    for (int i = 0; i < 4; ++i) out[(T * 4 + i) * 64 + threadIdx.x] = c[i];
}
__global__ void k(float* out) {
    const float a = (float)threadIdx.x, b = (float)(threadIdx.x + 1);
    one<0>(a, b, out); one<1>(a, b, out); one<2>(a, b, out); one<3>(a, b, out); one<5>(a, b, out); one<15>(a, b, out);
}
int main() {
    float* d; hipMalloc(&d, 16 * 4 * 64 * 4); hipMemset(d, 0, 16 * 4 * 64 * 4);
    k<<<1, 64>>>(d); hipDeviceSynchronize();
    static float h[16 * 4 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t : {0, 1, 2, 3, 5, 15})
        for (int i = 0; i < 4; ++i)
            for (int l = 0; l < 64; ++l) {
                const float want = (float)(4 * t + i) * (float)(l + 1), got = h[(t * 4 + i) * 64 + l];
                if (want != got) { if (bad < 10) printf("t %d i %d lane %d: got %g want %g\n", t, i, l, got, want); ++bad; }
            }
    printf("cbsz probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
