// Stand-alone timing of the row-sliding generator weight gradient (dmc-net_amd/csrc/gen_wgrad.hip), random inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DWR_NO_LOAD | -DWR_NO_MFMA] -I include -I dmc-net_amd/csrc tools/ubench/gen_wgrad_time.hip
#include "../../dmc-net_amd/csrc/gen_wgrad.hip"
#include <vector>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    using namespace dmc;
    const int N = argc > 1 ? atoi(argv[1]) : 120, H = argc > 2 ? atoi(argv[2]) : 224, W = argc > 3 ? atoi(argv[3]) : 224;
    const size_t HW = (size_t)H * W;
    float *mv, *res, *feat, *gout, *gbuf, *part, *zero;
    CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
    CK(hipMalloc(&mv, N * 2 * HW * 4)); CK(hipMalloc(&res, N * 3 * HW * 4)); CK(hipMalloc(&gout, N * 2 * HW * 4));
    CK(hipMalloc(&feat, N * NFEAT * HW * 4)); CK(hipMalloc(&gbuf, N * NFEAT * HW * 4));
    std::vector<float> h(N * NFEAT * HW);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(mv, h.data(), N * 2 * HW * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, h.data(), N * 3 * HW * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gout, h.data(), N * 2 * HW * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(feat, h.data(), N * NFEAT * HW * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(gbuf, h.data() + 1234, (N * NFEAT * HW - 1234) * 4, hipMemcpyHostToDevice));
    const int groups = gen_wgrad_rs_groups(N, H, W, 256);
    CK(hipMalloc(&part, (size_t)groups * WR_WPART * 4));
    for (int i = 0; i < 3; ++i) if (gen_wgrad_rs(mv, res, feat, gout, gbuf, zero, part, N, H, W, groups, 0)) { printf("launch failed: %s\n", err_buf()); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < R; ++i) gen_wgrad_rs(mv, res, feat, gout, gbuf, zero, part, N, H, W, groups, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double steps = (double)N * ((W + 31) / 32) * ((H + 1) / 2) / groups;
#ifdef WR_PROF
    {
        unsigned long long* dp; CK(hipMalloc(&dp, (size_t)groups * 32 * 8)); CK(hipMemset(dp, 0, (size_t)groups * 32 * 8));
        g_wr_prof = dp;
        gen_wgrad_rs(mv, res, feat, gout, gbuf, zero, part, N, H, W, groups, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hp((size_t)groups * 32);
        CK(hipMemcpy(hp.data(), dp, hp.size() * 8, hipMemcpyDeviceToHost));
        printf("wave: busy / total clocks (s_memtime), mean over groups\n");
        for (int w = 0; w < 16; ++w) {
            double b = 0, tt = 0;
            for (int g = 0; g < groups; ++g) { b += hp[((size_t)g * 16 + w) * 2]; tt += hp[((size_t)g * 16 + w) * 2 + 1]; }
            if (w == 0) printf("  wave  0: %.0f shader clocks in %.0f ticks of 100 MHz = %.2f GHz\n", tt / groups, b / groups, tt / b * 0.1);
            else printf("  wave %2d (%s): busy %.0f of %.0f = %.2f\n", w, w >= 8 ? "splitter" : "consumer", b / groups, tt / groups, b / tt);
        }
        g_wr_prof = nullptr;
    }
#endif
    printf("gen_wgrad_rs %d x %d x %d, %d groups: %.4f ms per launch, %.0f clocks per position at 2.4 GHz\n", N, H, W, groups, ms / R, ms / R * 2.4e6 / steps);
    return 0;
}
