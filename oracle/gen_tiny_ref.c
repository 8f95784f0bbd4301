/* Plain-C restatement of EstimatorDenseNetTiny's forward (TEST INFRASTRUCTURE: an arithmetic check
 * that does not go through torch's convolution kernels).
 *
 * Follows the reference's definition, not its code: code/dmcnet/model.py:111-119 (conv = 3x3,
 * stride 1, zero padding 1, bias, LeakyReLU(0.1); predict_flow = the same without activation),
 * :172-194 (five dense units, x <- cat(conv_i(x), x): new features are PREPENDED), :345-346
 * (optional + input_mv).  Weights in PyTorch layout [Cout][Cin][3][3], inputs NCHW fp32.
 * Accumulation in double, rounded to float once per output -- so it brackets both fp32 orders.
 */
#include <stdlib.h>
#include <string.h>

static const int WIDTH[5] = {8, 8, 6, 4, 2};

static void conv3x3(const float* x, int cin, const float* w, const float* b, int cout, int H, int W,
                    int lrelu, float* y) {
    for (int co = 0; co < cout; ++co)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                double acc = b[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int ky = 0; ky < 3; ++ky) {
                        const int ii = i + ky - 1;
                        if (ii < 0 || ii >= H) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int jj = j + kx - 1;
                            if (jj < 0 || jj >= W) continue;
                            acc += (double)w[((co * cin + ci) * 3 + ky) * 3 + kx] *
                                   (double)x[((size_t)ci * H + ii) * W + jj];
                        }
                    }
                float v = (float)acc;
                if (lrelu && v < 0.f) v *= 0.1f;
                y[((size_t)co * H + i) * W + j] = v;
            }
}

/* x5: [N][5][H][W] = cat(mv, residual); w[6], b[6]: conv_0..conv_4, predict_flow; out [N][2][H][W] */
int dmc_oracle_gen_tiny_forward(const float* x5, const float* const* w, const float* const* b,
                                float* out, int N, int H, int W, int add_mv) {
    const size_t HW = (size_t)H * W;
    float* cur = (float*)malloc(33 * HW * sizeof(float));
    float* nxt = (float*)malloc(33 * HW * sizeof(float));
    if (!cur || !nxt) return -1;
    for (int n = 0; n < N; ++n) {
        int c = 5;
        memcpy(cur, x5 + (size_t)n * 5 * HW, 5 * HW * sizeof(float));
        for (int k = 0; k < 5; ++k) {
            conv3x3(cur, c, w[k], b[k], WIDTH[k], H, W, 1, nxt);            /* new features first */
            memcpy(nxt + (size_t)WIDTH[k] * HW, cur, (size_t)c * HW * sizeof(float));
            float* t = cur; cur = nxt; nxt = t;
            c += WIDTH[k];
        }
        conv3x3(cur, c, w[5], b[5], 2, H, W, 0, out + (size_t)n * 2 * HW);
        if (add_mv)
            for (size_t i = 0; i < 2 * HW; ++i) out[(size_t)n * 2 * HW + i] += x5[(size_t)n * 5 * HW + i];
    }
    free(cur);
    free(nxt);
    return 0;
}
