// Stand-alone timing of the fused generator data gradient (dmc-net_amd/csrc/gen_fused_bwd.hip), random inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDMC_MEASURE [-DBZ_NO_FEAT] -I include -I dmc-net_amd/csrc tools/ubench/gen_fused_bwd_time.hip
#include "../../dmc-net_amd/csrc/gen_fused_bwd.hip"
namespace dmc { bool gen_fused_supported(int H, int W) { return W <= 224; } }
#include <vector>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 120, H = argc > 2 ? atoi(argv[2]) : 224, W = argc > 3 ? atoi(argv[3]) : 224;
    const size_t HW = (size_t)H * W;
    float *gout, *feat, *gbuf, *pk;
    CK(hipMalloc(&gout, N * 2 * HW * 4)); CK(hipMalloc(&feat, N * NFEAT * HW * 4)); CK(hipMalloc(&gbuf, N * NFEAT * HW * 4));
    CK(hipMalloc(&pk, (PACKED_TOTAL + ZERO_PAD) * 4));
    std::vector<float> h(N * NFEAT * HW);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(gout, h.data(), N * 2 * HW * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(feat, h.data(), N * NFEAT * HW * 4, hipMemcpyHostToDevice));
    std::vector<float> hp(PACKED_TOTAL + ZERO_PAD);
    for (auto& v : hp) v = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(pk, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) if (gen_fused_bwd_data(gout, feat, gbuf, pk, N, H, W, 0)) { printf("launch failed\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < R; ++i) gen_fused_bwd_data(gout, feat, gbuf, pk, N, H, W, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("gen_fused_bwd_data %d x %d x %d: %.4f ms per launch (%.1f TFLOP/s of 3,204 MAC/px)\n", N, H, W, ms / R, (double)N * HW * 6408 / (ms / R) / 1e9);
    return 0;
}
