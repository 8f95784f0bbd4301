// Standalone check + timing of the pre-split bf16x3 convolutions (conv_x3s.hip) through the C ABI of libdmcnet_hip.so,
// next to the in-loop-split kernels (conv_nhwc.hip) on the same tensors.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/conv_x3s_bench.hip -I include -L dmc-net_amd -ldmcnet_hip -Wl,-rpath,'$ORIGIN/../../../dmc-net_amd' -o tools/ubench/bin/conv_x3s_bench
//   conv_x3s_bench N H W Cin Cout [iters] [what: 1 fwd | 2 dgrad | 4 wgrad, bit mask]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "dmcnet_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define DK(x) do { int r_ = (x); if (r_ != 0) { printf("dmc error %d (%s) at %s:%d\n", r_, dmc_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

static float frand() { return (float)((rand() & 0xffff) / 32768.0 - 1.0); }

template <class F>
static float time_ms(F f, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    if (argc < 6) { printf("usage: %s N H W Cin Cout [iters] [what]\n", argv[0]); return 1; }
    const int N = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
    const int iters = argc > 6 ? atoi(argv[6]) : 20, what = argc > 7 ? atoi(argv[7]) : 7;
    const long M = (long)N * H * W;
    if (getenv("ABL")) DK(dmc_set_option("conv_ablate", atoi(getenv("ABL"))));
    if (getenv("CFG")) DK(dmc_set_option("conv_cfg", atoi(getenv("CFG"))));
    srand(7);
    std::vector<float> hx(M * Cin), hw((size_t)Cout * 9 * Cin), hdy(M * Cout);
    // FILL=0: all-zero operands (same instruction stream, minimal switching power: shows how much of the gap to the
    // peak is the chip's power limit on random data -- results are then trivially zero)
    const int fill = getenv("FILL") ? atoi(getenv("FILL")) : 1;
    for (auto& v : hx) v = fill ? frand() : 0.f;
    for (auto& v : hw) v = fill ? frand() * 0.1f : 0.f;
    for (auto& v : hdy) v = fill ? frand() : 0.f;
    float *x, *w, *y, *dy, *dx, *y3, *dw, *dw3, *wsp;
    void *xs, *dys, *wf, *wt, *w3f, *w3t;
    double* part;
    CK(hipMalloc(&x, M * Cin * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&y, M * Cout * 4)); CK(hipMalloc(&y3, M * Cout * 4));
    CK(hipMalloc(&dy, M * Cout * 4)); CK(hipMalloc(&dx, M * Cin * 4));
    CK(hipMalloc(&xs, dmc_x3s_slices_bytes(M, Cin))); CK(hipMalloc(&dys, dmc_x3s_slices_bytes(M, Cout)));
    CK(hipMalloc(&wf, dmc_x3s_wpack_bytes(Cin, Cout))); CK(hipMalloc(&wt, dmc_x3s_wpack_bytes(Cin, Cout)));
    CK(hipMalloc(&w3f, dmc_conv_nhwc_wt_bytes(Cin, Cout, 3, 3))); CK(hipMalloc(&w3t, dmc_conv_nhwc_wt_bytes(Cin, Cout, 3, 3)));
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dw3, hw.size() * 4));
    const size_t wsb = dmc_conv_nhwc_wgrad_bytes(N, H, W, Cin, Cout, 3, 3, 1, 1);
    CK(hipMalloc(&wsp, wsb > 0 ? wsb : 16));
    CK(hipMemcpy(x, hx.data(), M * Cin * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, hdy.data(), M * Cout * 4, hipMemcpyHostToDevice));
    if (!dmc_x3s_conv_supported(N, H, W, Cin, Cout)) { printf("shape not supported by x3s\n"); return 2; }
    const int nblk = dmc_x3s_conv_stat_blocks(N, H, W, Cout);
    CK(hipMalloc(&part, (size_t)nblk * Cout * 2 * sizeof(double)));
    DK(dmc_x3s_split(x, xs, M, Cin, 0));
    DK(dmc_x3s_split(dy, dys, M, Cout, 0));
    DK(dmc_x3s_pack_weights(w, wf, wt, Cin, Cout, 0));
    DK(dmc_conv_nhwc_split(w, w3f, w3t, Cin, Cout, 3, 3, 0));
    const double gf = 2.0 * M * Cout * Cin * 9 / 1e9;
    printf("shape N=%d H=%d W=%d Cin=%d Cout=%d  M=%ld  %.1f GFLOP  stat blocks %d\n", N, H, W, Cin, Cout, M, gf, nblk);

    // ---- merge(split(x)) == x bit for bit ----
    {
        float* xm; CK(hipMalloc(&xm, M * Cin * 4));
        DK(dmc_x3s_merge(xs, xm, M, Cin, 0));
        std::vector<float> back(M * Cin);
        CK(hipMemcpy(back.data(), xm, M * Cin * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (long i = 0; i < M * Cin; ++i) bad += back[i] != hx[i];
        printf("split/merge round trip: %ld mismatches\n", bad);
        CK(hipFree(xm));
    }
    std::vector<float> hy(M * Cout), hy3(M * Cout), hdx(M * Cin), hdx3(M * Cin);
    if (what & 1) {
        DK(dmc_x3s_conv_fwd(xs, wf, y, part, dmc_x3s_conv_stat_blocks(N, H, W, Cout), N, H, W, Cin, Cout, 0));
        DK(dmc_conv_nhwc_fwd(x, nullptr, w3f, nullptr, nullptr, y3, nullptr, 0, N, H, W, Cin, Cout, 3, 3, 1, 1, 0, 0));
        CK(hipMemcpy(hy.data(), y, M * Cout * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hy3.data(), y3, M * Cout * 4, hipMemcpyDeviceToHost));
        double ymax = 0, e4 = 0, e3 = 0;
        for (int t = 0; t < 4000; ++t) {
            const long m = (t < 64 ? t : t < 128 ? M - 1 - (t - 64) : (long)(((unsigned long long)rand() * 2654435761ull) % M));
            const int co = rand() % Cout;
            const int n = (int)(m / (H * W)), yy = (int)(m % (H * W)) / W, xx = (int)(m % W);
            double r = 0;
            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float* xp = &hx[(((long)n * H + iy) * W + ix) * Cin];
                const float* wp = &hw[((size_t)co * 9 + ky * 3 + kx) * Cin];
                for (int c = 0; c < Cin; ++c) r += (double)xp[c] * wp[c];
            }
            ymax = fmax(ymax, fabs(r));
            e4 = fmax(e4, fabs(r - hy[m * Cout + co]));
            e3 = fmax(e3, fabs(r - hy3[m * Cout + co]));
        }
        long ndiff = 0; double dmax = 0;
        for (long i = 0; i < M * Cout; ++i) { const double d = fabs((double)hy[i] - hy3[i]); if (d > 0) ++ndiff; dmax = fmax(dmax, d); }
        // statistics partials against the stored outputs
        std::vector<double> hp((size_t)nblk * Cout * 2);
        CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
        double serr = 0;
        for (int c = 0; c < Cout; c += 7) {
            double s = 0, ss = 0, ps = 0, pss = 0;
            for (long m = 0; m < M; ++m) { const double v = hy[m * Cout + c]; s += v; ss += v * v; }
            for (int b = 0; b < nblk; ++b) { ps += hp[((size_t)b * Cout + c) * 2]; pss += hp[((size_t)b * Cout + c) * 2 + 1]; }
            serr = fmax(serr, fmax(fabs(s - ps) / (fabs(s) + 1.0), fabs(ss - pss) / ss));
        }
        printf("fwd   : max|y| %.3f  err vs fp64: x3s %.3e  conv3 %.3e (rel %.2e / %.2e)   x3s vs conv3: %ld differing, max %.3e   stats rel err %.2e\n",
               ymax, e4, e3, e4 / ymax, e3 / ymax, ndiff, dmax, serr);
        const float t4 = time_ms([&] { DK(dmc_x3s_conv_fwd(xs, wf, y, part, dmc_x3s_conv_stat_blocks(N, H, W, Cout), N, H, W, Cin, Cout, 0)); }, iters);
        const float t3 = time_ms([&] { DK(dmc_conv_nhwc_fwd(x, nullptr, w3f, nullptr, nullptr, y3, part, dmc_conv_nhwc_stat_blocks(N, H, W, Cin, Cout, 3, 1, 1), N, H, W, Cin, Cout, 3, 3, 1, 1, 0, 0)); }, iters);
        const float t4b = time_ms([&] { DK(dmc_x3s_conv_fwd(xs, wf, y, part, dmc_x3s_conv_stat_blocks(N, H, W, Cout), N, H, W, Cin, Cout, 0)); }, iters);
        printf("fwd   : x3s %.3f ms (%.1f TF)  conv3 %.3f ms (%.1f TF)  x3s again %.3f ms (%.1f TF)\n", t4, gf / t4, t3, gf / t3, t4b, gf / t4b);
    }
    if (what & 2) {
        DK(dmc_x3s_conv_dgrad(dys, wt, nullptr, dx, N, H, W, Cin, Cout, 0));
        CK(hipMemcpy(hdx.data(), dx, M * Cin * 4, hipMemcpyDeviceToHost));
        double ymax = 0, e4 = 0;
        for (int t = 0; t < 3000; ++t) {
            const long m = (t < 64 ? t : t < 128 ? M - 1 - (t - 64) : (long)(((unsigned long long)rand() * 2654435761ull) % M));
            const int ci = rand() % Cin;
            const int n = (int)(m / (H * W)), yy = (int)(m % (H * W)) / W, xx = (int)(m % W);
            double r = 0;
            for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
                const int oy = yy - (ky - 1), ox = xx - (kx - 1);
                if (oy < 0 || oy >= H || ox < 0 || ox >= W) continue;
                const float* dp = &hdy[(((long)n * H + oy) * W + ox) * Cout];
                for (int c = 0; c < Cout; ++c) r += (double)dp[c] * hw[((size_t)c * 9 + ky * 3 + kx) * Cin + ci];
            }
            ymax = fmax(ymax, fabs(r));
            e4 = fmax(e4, fabs(r - hdx[m * Cin + ci]));
        }
        printf("dgrad : max|dx| %.3f  err vs fp64: x3s %.3e (rel %.2e)\n", ymax, e4, e4 / ymax);
        const float t4 = time_ms([&] { DK(dmc_x3s_conv_dgrad(dys, wt, nullptr, dx, N, H, W, Cin, Cout, 0)); }, iters);
        const float t3 = time_ms([&] { DK(dmc_conv_nhwc_dgrad(dy, nullptr, (float*)w3t, dx, N, H, W, Cin, Cout, 3, 3, 1, 1, 0)); }, iters);
        printf("dgrad : x3s %.3f ms (%.1f TF)  conv3 %.3f ms (%.1f TF)\n", t4, gf / t4, t3, gf / t3);
    }
#if 1
    if (what & 4) {
        const size_t wb = dmc_x3s_conv_wgrad_bytes(N, H, W, Cin, Cout);
        float* wsp4; CK(hipMalloc(&wsp4, wb > 0 ? wb : 16));
        DK(dmc_x3s_conv_wgrad(xs, dys, dw, wsp4, N, H, W, Cin, Cout, 0));
        DK(dmc_conv_nhwc_wgrad(x, dy, dw3, wsp, N, H, W, Cin, Cout, 3, 3, 1, 1, 0));
        std::vector<float> hdw(hw.size()), hdw3(hw.size());
        CK(hipMemcpy(hdw.data(), dw, hw.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hdw3.data(), dw3, hw.size() * 4, hipMemcpyDeviceToHost));
        double ymax = 0, e4 = 0, e3 = 0;
        for (int t = 0; t < 300; ++t) {
            const int co = rand() % Cout, tap = t < 9 ? t : rand() % 9, ci = rand() % Cin;
            const int ky = tap / 3, kx = tap % 3;
            double r = 0;
            for (long m = 0; m < M; ++m) {
                const int n = (int)(m / (H * W)), yy = (int)(m % (H * W)) / W, xx = (int)(m % W);
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                r += (double)hdy[m * Cout + co] * hx[(((long)n * H + iy) * W + ix) * Cin + ci];
            }
            ymax = fmax(ymax, fabs(r));
            e4 = fmax(e4, fabs(r - hdw[((size_t)co * 9 + tap) * Cin + ci]));
            e3 = fmax(e3, fabs(r - hdw3[((size_t)co * 9 + tap) * Cin + ci]));
        }
        printf("wgrad : max|dw| %.3f  err vs fp64: x3s %.3e  conv3 %.3e (rel %.2e / %.2e)\n", ymax, e4, e3, e4 / ymax, e3 / ymax);
        const float t4 = time_ms([&] { DK(dmc_x3s_conv_wgrad(xs, dys, dw, wsp4, N, H, W, Cin, Cout, 0)); }, iters);
        const float t3 = time_ms([&] { DK(dmc_conv_nhwc_wgrad(x, dy, dw3, wsp, N, H, W, Cin, Cout, 3, 3, 1, 1, 0)); }, iters);
        printf("wgrad : x3s %.3f ms (%.1f TF)  conv3 %.3f ms (%.1f TF)\n", t4, gf / t4, t3, gf / t3);
    }
#endif
    if (what & 8) {   // stride-2 data gradient: here N H W are the dy dims (OH, OW); dx is [N][2H][2W][Cin]
        if (!dmc_x3s_conv_dgrad_s2_supported(N, H, W, Cin, Cout)) { printf("dgrad_s2: shape not supported\n"); return 0; }
        const long M2 = M * 4;
        float *dx2, *dx3; void* wt2;
        CK(hipMalloc(&dx2, M2 * Cin * 4)); CK(hipMalloc(&dx3, M2 * Cin * 4)); CK(hipMalloc(&wt2, dmc_x3s_wpack_bytes(Cin, Cout)));
        DK(dmc_x3s_pack_weights_s2(w, wt2, Cin, Cout, 0));
        DK(dmc_x3s_conv_dgrad_s2(dys, wt2, dx2, N, H, W, Cin, Cout, 0));
        DK(dmc_conv_nhwc_dgrad(dy, nullptr, (float*)w3t, dx3, N, 2 * H, 2 * W, Cin, Cout, 3, 3, 2, 1, 0));
        std::vector<float> h2(M2 * Cin), h3(M2 * Cin);
        CK(hipMemcpy(h2.data(), dx2, M2 * Cin * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h3.data(), dx3, M2 * Cin * 4, hipMemcpyDeviceToHost));
        double ymax = 0, e4 = 0, e3 = 0;
        const int H2 = 2 * H, W2 = 2 * W;
        for (int t = 0; t < 4000; ++t) {
            const long m = (t < 128 ? t : t < 256 ? M2 - 1 - (t - 128) : (long)(((unsigned long long)rand() * 2654435761ull) % M2));
            const int ci = rand() % Cin;
            const int n = (int)(m / (H2 * W2)), iy = (int)(m % (H2 * W2)) / W2, ix = (int)(m % W2);
            double r = 0;
            for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
                const int ny = iy + 1 - ty, nx = ix + 1 - tx;
                if (ny < 0 || nx < 0 || (ny & 1) || (nx & 1)) continue;
                const int oy = ny / 2, ox = nx / 2;
                if (oy >= H || ox >= W) continue;
                const float* dp = &hdy[(((long)n * H + oy) * W + ox) * Cout];
                for (int c = 0; c < Cout; ++c) r += (double)dp[c] * hw[((size_t)c * 9 + ty * 3 + tx) * Cin + ci];
            }
            ymax = fmax(ymax, fabs(r));
            e4 = fmax(e4, fabs(r - h2[m * Cin + ci]));
            e3 = fmax(e3, fabs(r - h3[m * Cin + ci]));
        }
        printf("dgrad2: max|dx| %.3f  err vs fp64: x3s %.3e  conv3 %.3e (rel %.2e / %.2e)\n", ymax, e4, e3, e4 / ymax, e3 / ymax);
        const double gf2 = gf;   // 2 * M(dy) * Cout * Cin * 9
        const float t4 = time_ms([&] { DK(dmc_x3s_conv_dgrad_s2(dys, wt2, dx2, N, H, W, Cin, Cout, 0)); }, iters);
        const float t3 = time_ms([&] { DK(dmc_conv_nhwc_dgrad(dy, nullptr, (float*)w3t, dx3, N, 2 * H, 2 * W, Cin, Cout, 3, 3, 2, 1, 0)); }, iters);
        printf("dgrad2: x3s %.3f ms (%.1f TF)  conv3 %.3f ms (%.1f TF)\n", t4, gf2 / t4, t3, gf2 / t3);
    }
    return 0;
}
