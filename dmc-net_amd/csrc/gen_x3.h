// EstimatorDenseNetTiny hidden layers in bf16x3 arithmetic on v_mfma_f32_16x16x32_bf16 (gen_x3.hip): interface to gen_tiny.hip.
#pragma once
#include "dmc_common.h"

namespace dmc {

constexpr int GX_LAYERS = 3;          // hidden layers 0, 1, 2 (Cout 8, 8, 6: two row tiles of (dy, co))

// weight fragments of layer K: 2 row tiles x k-blocks x 3 slices, 1 KB each (k-block = four (dx, 8-channel chunk) groups)
__host__ __device__ constexpr int gx_kb(int K) { return (3 * ((cin_of(K) + 7) / 8) + 3) / 4; }
__host__ __device__ constexpr int gx_nfrag(int K) { return 2 * gx_kb(K) * 3; }
__host__ __device__ constexpr int gx_frag_off(int K) {      // in fragments of 1 KB
    int o = 0;
    for (int i = 0; i < K; ++i) o += gx_nfrag(i);
    return o;
}
constexpr int GX_PACK_THREADS = gx_frag_off(GX_LAYERS) / 3 * 512;     // one thread per (fragment triple, lane, j)

// bytes of the packed weight fragments (all layers this file serves), kept behind the fp32 parameter block of the workspace
inline size_t gen_x3_frag_bytes() { return (size_t)gx_frag_off(GX_LAYERS) * 1024; }

#ifdef __HIPCC__
// One thread of the fragment pack (called by gen_tiny.hip's parameter-pack kernel for its threads beyond the fp32 block: no
// launch of its own).  A fragments: [layer][row tile][k-block][slice][lane][8 bf16]; lane (i, kq): row R = 16 rt + i = (dy, co) =
// (R / 8, R % 8), k = 8 kq + j of k-block kb = group g = 4 kb + kq = (dx, chunk) = (g / NCH, g % NCH), channel 8 chunk + j (physical
// channel order: dmc_common.h); truncation split into three bf16 slices as everywhere (conv_nhwc.hip)
__device__ __forceinline__ void gen_x3_pack_thread(const ParamPtrs& P, unsigned short* __restrict__ frags, int t) {
    int K = 0, base = 0;
    while (K < GX_LAYERS && t >= (base + gx_nfrag(K) / 3) * 512) { base += gx_nfrag(K) / 3; ++K; }
    if (K >= GX_LAYERS) return;
    const int cin = cin_of(K), cout = cout_of(K), nch = (cin + 7) / 8, G = 3 * nch, KB = (G + 3) / 4;
    const int u = t - base * 512, j = u & 7, lane = (u >> 3) & 63, f = u >> 9;     // f = rt * KB + kb
    const int kb = f % KB, rt = f / KB;
    const int R = 16 * rt + (lane & 15), g = 4 * kb + (lane >> 4);
    const int dy = R >> 3, co = R & 7, dx = g / nch, ci = 8 * (g % nch) + j;
    float w = 0.f;
    if (R < 24 && co < cout && g < G && ci < cin) w = P.w[K][(co * cin + logical_of(K, ci)) * 9 + dy * 3 + dx];
    const unsigned u0 = __float_as_uint(w);
    const float r1 = w - __uint_as_float(u0 & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
    unsigned short* dst = frags + ((size_t)(gx_frag_off(K) + f * 3) * 64 + lane) * 8 + j;
    dst[0] = (unsigned short)(u0 >> 16);
    dst[512] = (unsigned short)(u1 >> 16);
    dst[1024] = (unsigned short)(u2 >> 16);
}
#endif

// layers this path serves (forward hidden layers): K in [0, GX_LAYERS)
bool gen_x3_supported(int K, int H, int W);
// y_K = LeakyReLU(0.1)(conv3x3(cat(mv, res, y_0 .. y_{K-1})) + b_K) for frames [0, N): feat is [N][28][H][W]; pk = the packed
// fp32 parameters (bias), frags = the fragments gen_x3_pack_thread() wrote
int gen_x3_layer(int K, const float* mv, const float* res, float* feat, const float* pk, const void* frags, int N, int H, int W,
                 hipStream_t s);

}  // namespace dmc
