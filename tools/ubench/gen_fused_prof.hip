// Stand-alone timing / per-wave busy profile of the fused generator forward (dmc-net_amd/csrc/gen_fused.hip compiled with
// -DDMC_MEASURE): random inputs, N frames of H x W;  prints the launch time and, per wave role, the share of the step time it
// spent between two barriers and the SIMD it ran on.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DDMC_MEASURE -I include -I dmc-net_amd/csrc tools/ubench/gen_fused_prof.hip
#include "../../dmc-net_amd/csrc/gen_fused.hip"
#include <vector>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 120, H = argc > 2 ? atoi(argv[2]) : 224, W = argc > 3 ? atoi(argv[3]) : 224;
    const int save = argc > 4 ? atoi(argv[4]) : 1;      // 0: nothing saved; 1: features saved; 2: saved, every frame into frame 0's planes
    g_fz_feat_one_frame = save == 2;
    const size_t HW = (size_t)H * W;
    float *mv, *res, *feat, *out, *pk, *flow; double* part;
    CK(hipMalloc(&mv, N * 2 * HW * 4)); CK(hipMalloc(&res, N * 3 * HW * 4)); CK(hipMalloc(&feat, N * NFEAT * HW * 4));
    CK(hipMalloc(&out, N * 2 * HW * 4)); CK(hipMalloc(&flow, N * 2 * HW * 4)); CK(hipMalloc(&pk, (PACKED_TOTAL + ZERO_PAD) * 4));
    CK(hipMalloc(&part, 4096 * 8));
    std::vector<float> h(N * 3 * HW);
    srand(1);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(mv, h.data(), N * 2 * HW * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, h.data(), N * 3 * HW * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(flow, h.data(), N * 2 * HW * 4, hipMemcpyHostToDevice));
    std::vector<float> hp(PACKED_TOTAL + ZERO_PAD);
    for (auto& v : hp) v = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(pk, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    ParamPtrs P;
    { int off = 0; for (int k = 0; k < NL; ++k) { P.w[k] = pk + off; off += cin_of(k) * 9 * cout_of(k); } for (int k = 0; k < NL; ++k) { P.b[k] = pk + off; off += cout_of(k); } }
    int nparts = 0;
    for (int i = 0; i < 3; ++i) if (gen_fused_fwd(mv, res, save ? feat : nullptr, out, P, flow, part, &nparts, N, H, W, 1, 0)) { printf("launch failed\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < R; ++i) gen_fused_fwd(mv, res, save ? feat : nullptr, out, P, flow, part, &nparts, N, H, W, 1, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double px = (double)N * HW;
    printf("gen_fused %d x %d x %d save %d: %.4f ms per launch, %.1f TFLOP/s (%.3f of 157.3)\n", N, H, W, save, ms / R, px * 9108 / (ms / R) / 1e9,
           px * 9108 / (ms / R) / 1e9 / 157.3);
    const int wgs = nparts / 2;
    unsigned long long* prof; CK(hipMalloc(&prof, (size_t)wgs * 12 * 4 * 8)); CK(hipMemset(prof, 0, (size_t)wgs * 12 * 4 * 8));
    g_fz_prof = prof;
    gen_fused_fwd(mv, res, save ? feat : nullptr, out, P, flow, part, &nparts, N, H, W, 1, 0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hq((size_t)wgs * 12 * 4);
    CK(hipMemcpy(hq.data(), prof, hq.size() * 8, hipMemcpyDeviceToHost));
    const char* role[12] = {"L2 a", "L3 a", "L1 a", "L5 a", "L2 b", "L3 b", "L1 b", "L5 b", "load", "L4 a", "L4 b", "L0 a+b"};
    printf("wave role      busy/total  busy clk/step  total clk/step   SIMD histogram (of %d workgroups)\n", wgs);
    for (int w = 0; w < 12; ++w) {
        double busy = 0, tot = 0, steps = 0; int simd[4] = {0, 0, 0, 0};
        for (int g = 0; g < wgs; ++g) {
            const unsigned long long* q = &hq[((size_t)g * 12 + w) * 4];
            busy += q[0]; tot += q[1]; steps += q[3]; ++simd[(q[2] >> 4) & 3];
        }
        printf("%2d   %-8s  %.3f       %8.0f       %8.0f          %d %d %d %d\n", w, role[w], busy / tot, busy / steps, tot / steps, simd[0], simd[1], simd[2], simd[3]);
    }
    return 0;
}
