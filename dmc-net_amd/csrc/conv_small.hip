// 3x3 / stride-1 / padding-1 NHWC convolutions with 16 or 32 channels in bf16x3 arithmetic (gfx950).
//
// The PatchGAN discriminator's high-resolution blocks -- Conv2d(16, 16, 3, 1, 1) on 112 x 112 maps and Conv2d(32, 32, 3, 1, 1)
// on 56 x 56 maps, code/dmcnet_GAN/model.py:254-279 as chained in Discriminator3, :332-366 -- at 240 / 120 frames per step:
// 386 / 193 MB of activations per launch against 13.9 GFLOP.  On the fp32 matrix cores (v_mfma_f32_16x16x4_f32, K = 144:
// conv_nhwc_kernel / conv2_kernel) they sit at the fp32-MFMA roof (0.36 ms for a 240-frame block: 0.49 of 157 TFLOP/s); in
// bf16x3 arithmetic (conv_nhwc.hip: an fp32 value = the exact sum of three bf16 slices, six slice products per product
// block, fp32 accumulate: fp32-level error) the matrix work shrinks 2.67x and the layers become HBM streams.
//
// v_mfma_f32_16x16x32_bf16: rows = the 16 output channels of a row tile, columns = 16 consecutive output pixels of an image row,
// a k-block = 32 contraction values = two taps x 16 channels (C = 16: five k-blocks, the last half empty) or one tap x 32
// channels (C = 32: nine).  Lane (n = lane % 16, kq = lane / 16) of the B operand holds eight consecutive channels of pixel
// n's tap.  The fp32 NHWC activation is loaded straight from global memory (range-checked buffer loads: zeros outside the
// image), split into the three slices in registers ONCE per input piece and parked in a wave-private mini patch in LDS; the
// weights are packed once per call in fragment order and live in LDS for the whole launch (15 / 54 fragments of 1 KB,
// lane-linear, conflict-free reads).  Persistent waves walk the tiles; no workgroup barrier in the loop.
// Epilogue as conv_tile_epilogue of conv_nhwc.hip: + bias, LeakyReLU(0.2), Dropout2d keep mask, per-channel (sum, sum of
// squares) of the result in fp64 for the BatchNorm that follows, 16-byte stores (a wave writes 1 KB of consecutive memory).
// The data gradient is the same kernel on dy with the weights packed transposed and mirrored.
#include "dmc_common.h"
#include "x3s_common.h"
#include "conv_small.h"

using namespace dmc;
using namespace dmc::x3;

namespace {

// 16 bytes through a buffer descriptor: an offset beyond num_records returns zeros -- the zero padding of the convolution
// costs neither a branch nor a register move (the branchy form spent a quarter of the kernel's instructions on it)
__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t srd, unsigned voff) {
    const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, voff, 0, 0));
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

struct CsmArgs {
    const float* x;        // [M][C] fp32 NHWC
    const void* wp;        // packed weights [3 slices][KB][MT][16 rows][32 k] bf16
    const float* bias;     // [C] or null
    const float* keep;     // [N][C] or null
    float* y;              // [M][C]
    double* stat_part;     // [gridDim.x][C][2] or null
    int N, H, W, M, act;
    float rW, rH;          // 1 / W, 1 / H (pixel -> row / image by a float multiply + correction: M < 2^24)
};

template <int C> struct Csm {
    static constexpr int KB = C == 16 ? 5 : 9;          // k-blocks of 32 contraction values
    static constexpr int MT = C / 16;                   // row tiles of 16 output channels
    static constexpr int NFRAG = 3 * KB * MT;           // weight fragments (1 KB each)
    static constexpr int NW = C == 16 ? 4 : 8;          // waves per workgroup (C = 32: eight waves share the 54 KB of weights)
};

// Tiles are 16-pixel segments of one image row ((W + 15) / 16 per row; the last one of a row may be partial).  The nine taps
// of a tile touch 3 rows x 18 columns of the input: every lane loads and splits its share of those 54 x C / 8 eight-channel
// pieces ONCE (C = 16: 1.7 pieces per lane instead of 5 k-blocks' worth, C = 32: 3.4 instead of 9 -- the in-register split is
// 45 VALU instructions per piece and bounded the first version of this kernel) into a wave-private mini patch in LDS
// ([3 slices][3 rows][18 pixels][C] bf16, zeros where the window leaves the image: range-checked loads), from which the B
// fragments of all taps are 16-byte reads.  No workgroup barrier in the tile loop.
template <int C>
__global__ __launch_bounds__(Csm<C>::NW * 64) void csm_conv_kernel(CsmArgs a) {
    using G = Csm<C>;
    constexpr int KB = G::KB, MT = G::MT, NW = G::NW;
    constexpr int OC = C / 8;                                 // 16-byte pieces (8 bf16) per pixel and slice
    constexpr int NPC = 3 * 18 * OC;                          // pieces of a tile's patch
    constexpr int SS = NPC * 16;                              // bytes of one slice of a wave's patch
    __shared__ __attribute__((aligned(16))) u32x4 wlds[G::NFRAG * 64];       // 15 / 54 KB: every weight fragment, lane-linear
    __shared__ __attribute__((aligned(16))) char patch[NW][3 * SS];
    __shared__ double red[NW][C][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp);
    for (int i = tid; i < G::NFRAG * 64; i += NW * 64) wlds[i] = wsrc[i];
    __syncthreads();
    float bias_r[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_r[mt][i] = a.bias ? a.bias[mt * 16 + 4 * kq + i] : 0.f;
    double ssum[MT][4], ssq[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) ssum[mt][i] = ssq[mt][i] = 0.0;

    const int segs = (a.W + 15) >> 4;
    const int ntiles = a.N * a.H * segs;
    const int stride = (int)gridDim.x * NW;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
    char* const mp = patch[wave];
    // this lane's pieces of the patch: piece f = lane + 64 k -> (row f / (18 OC), column (f / OC) % 18, octet f % OC)
    constexpr int NPL = (NPC + 63) / 64;
    int prow[NPL], pcol[NPL], poct[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int f = lane + 64 * k;
        prow[k] = f / (18 * OC);
        pcol[k] = (f / OC) % 18;
        poct[k] = f % OC;
    }
    // fragment read offsets: k-block kb, lane (n, kq): C = 16: tap 2 kb + (kq >> 1), octet kq & 1; C = 32: tap kb, octet kq.
    // (tap 9 of the last C = 16 k-block has zero weights: it re-reads tap 8's finite values)
    int boff[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        int tap = C == 16 ? 2 * kb + (kq >> 1) : kb;
        tap = tap < 9 ? tap : 8;
        const int dy = tap / 3, dx = tap - dy * 3;            // patch row dy (= image row y + dy - 1), patch column n + dx
        boff[kb] = ((dy * 18 + n + dx) * OC + (C == 16 ? (kq & 1) : kq)) * 16;
    }
    const float rS = 1.0f / (float)segs;
    // tile -> (image, row, first column): float reciprocal + one correction step each (exact below 2^24)
    auto geom = [&](int tile, int& img, int& yy, int& x0) {
        int r = (int)((float)tile * rS);
        r -= (r * segs > tile);
        r += ((r + 1) * segs <= tile);
        x0 = (tile - r * segs) * 16;
        img = (int)((float)r * a.rH);
        img -= (img * a.H > r);
        img += ((img + 1) * a.H <= r);
        yy = r - img * a.H;
    };
    // this lane's pieces of a tile's patch, as loaded (fp32): the NEXT tile's are in flight while the current tile computes --
    // one tile per wave at a time left the kernel bound by the memory round trip (4 waves per SIMD x ~2 us per tile)
    float4 plo[2][NPL], phi[2][NPL];
    auto load_patch = [&](int tile, float4 (&lo)[NPL], float4 (&hi)[NPL]) {
        int img, yy, x0;
        geom(tile, img, yy, x0);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int f = lane + 64 * k;
            const int iy = yy + prow[k] - 1, ix = x0 + pcol[k] - 1;
            const bool ok = (NPC % 64 == 0 || f < NPC) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + iy) * a.W + ix) * C + 8 * poct[k]) * 4u : 0x80000000u;
            lo[k] = buf_load16(srd, off);
            hi[k] = buf_load16(srd, off + 16u);
        }
    };
    int it = 0;
    const int tile0 = (int)blockIdx.x * NW + wave;
    if (tile0 < ntiles) load_patch(tile0, plo[0], phi[0]);
#pragma unroll 1
    for (int tile = tile0; tile < ntiles; tile += 2 * stride) {
        // two tiles per iteration so that the register double buffer has compile-time indices
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tcur = tile + half * stride;
            if (tcur >= ntiles) break;
            asm volatile("" ::: "memory");                   // the weight fragments are re-read from LDS per tile, not hoisted into registers
            if (tcur + stride < ntiles) load_patch(tcur + stride, plo[half ^ 1], phi[half ^ 1]);
            int img, yy, x0;
            geom(tcur, img, yy, x0);
            // ---- the patch: split, store ----
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int f = lane + 64 * k;
                u32x4 s0, s1, s2;
                split8(plo[half][k], phi[half][k], s0, s1, s2);
                if (NPC % 64 == 0 || f < NPC) {
                    *reinterpret_cast<u32x4*>(mp + f * 16) = s0;
                    *reinterpret_cast<u32x4*>(mp + SS + f * 16) = s1;
                    *reinterpret_cast<u32x4*>(mp + 2 * SS + f * 16) = s2;
                }
            }
        __builtin_amdgcn_wave_barrier();                     // same wave: its LDS instructions complete in order
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const u32x4 b0 = *reinterpret_cast<const u32x4*>(mp + boff[kb]);
            const u32x4 b1 = *reinterpret_cast<const u32x4*>(mp + SS + boff[kb]);
            const u32x4 b2 = *reinterpret_cast<const u32x4*>(mp + 2 * SS + boff[kb]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const u32x4 w0 = wlds[((0 * KB + kb) * MT + mt) * 64 + lane];
                const u32x4 w1 = wlds[((1 * KB + kb) * MT + mt) * 64 + lane];
                const u32x4 w2 = wlds[((2 * KB + kb) * MT + mt) * 64 + lane];
                // slice products (weight, input): small terms first
                acc[mt] = mfma16(w0, b2, acc[mt]);
                acc[mt] = mfma16(w2, b0, acc[mt]);
                acc[mt] = mfma16(w1, b1, acc[mt]);
                acc[mt] = mfma16(w0, b1, acc[mt]);
                acc[mt] = mfma16(w1, b0, acc[mt]);
                acc[mt] = mfma16(w0, b0, acc[mt]);
            }
        }
        __builtin_amdgcn_wave_barrier();                     // the next tile's stores follow this tile's reads
        // epilogue: lane holds channels mt * 16 + 4 kq + i of pixel (yy, x0 + n)
        const bool pvalid = x0 + n < a.W;
        const size_t p = (size_t)(img * a.H + yy) * a.W + x0 + n;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = acc[mt][i] + bias_r[mt][i];
                if (a.act) t = t > 0.f ? t : 0.2f * t;
                v[i] = t;
            }
            if (a.keep) {
                const float4 k = *reinterpret_cast<const float4*>(a.keep + (size_t)img * C + mt * 16 + 4 * kq);
                v[0] *= k.x; v[1] *= k.y; v[2] *= k.z; v[3] *= k.w;
            }
            if (pvalid) {
                *reinterpret_cast<float4*>(a.y + p * C + mt * 16 + 4 * kq) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.stat_part) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        ssum[mt][i] += (double)v[i];
                        ssq[mt][i] += (double)v[i] * (double)v[i];
                    }
                }
            }
        }
        }
    }
    (void)it;
    if (!a.stat_part) return;
    // over the 16 pixel lanes of a channel group (fixed order), then over the four waves (fixed order)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ssum[mt][i] = row16_sum(ssum[mt][i]);
            ssq[mt][i] = row16_sum(ssq[mt][i]);
            if (n == 0) {
                red[wave][mt * 16 + 4 * kq + i][0] = ssum[mt][i];
                red[wave][mt * 16 + 4 * kq + i][1] = ssq[mt][i];
            }
        }
    __syncthreads();
    if (tid < 2 * C) {
        const int c = tid >> 1, which = tid & 1;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][c][which];
        a.stat_part[((size_t)blockIdx.x * C + c) * 2 + which] = t;
    }
}

// ---- forward of the 16 -> 32 stride-2 block ---------------------------------------------------------------------------------
// csm_conv_kernel's scheme for Conv2d(16, 32, 3, 2, 1) on even-sized maps (the second block of every PatchGAN discriminator,
// code/dmcnet_GAN/model.py:254-279; conv_nhwc_kernel<256, 16, ...>, fp32 MFMA: 87 us per launch in the I3D recipe): rows = the 32
// output channels (two row tiles), columns = 16 output pixels of an output row, five k-blocks of two taps x 16 channels.  The
// tile's patch keeps, per input row 2 oy - 1 .. 2 oy + 1, the even columns Ev[i] = x[2 (ox0 + i)] (16 slots) and the odd columns
// Od[i] = x[2 (ox0 + i) - 1] (17 slots) apart: tap kx = 1 reads Ev[n], kx = 0 reads Od[n], kx = 2 reads Od[n + 1] for output pixel
// n -- unit-stride fragment reads as in the stride-1 kernel.
struct Csm2 {
    static constexpr int CIN = 16, COUT = 32, KB = 5, MT = 2, NW = 4, NFRAG = 3 * KB * MT;
};

__global__ __launch_bounds__(256) void csm_pack_s2_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp) {
    constexpr int total = Csm2::KB * Csm2::MT * 16 * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int t = i;
        const int kk = t & 31; t >>= 5;
        const int row = t & 15; t >>= 4;
        const int mt = t % Csm2::MT;
        const int kb = t / Csm2::MT;
        const int tap = 2 * kb + (kk >> 4), ci = kk & 15, co = mt * 16 + row;
        float v = 0.f;
        if (tap < 9) v = w[((size_t)co * 9 + tap) * Csm2::CIN + ci];
        unsigned u0, u1, u2;
        split3(v, u0, u1, u2);
        const size_t e = (size_t)((kk >> 3) * 16 + row) * 8 + (kk & 7);
        const size_t f = (size_t)(kb * Csm2::MT + mt) * 512;
        const size_t sl = (size_t)Csm2::KB * Csm2::MT * 512;
        wp[f + e] = (unsigned short)(u0 >> 16);
        wp[sl + f + e] = (unsigned short)(u1 >> 16);
        wp[2 * sl + f + e] = (unsigned short)(u2 >> 16);
    }
}

struct Csm2Args {
    const float* x;        // [N][H][W][16]
    const void* wp;
    const float* bias;     // [32] or null
    const float* keep;     // [N][32] or null
    float* y;              // [N][H/2][W/2][32]
    double* stat_part;     // [gridDim.x][32][2] or null
    int N, H, W, OH, OW, act;
    float rOH;
};

__global__ __launch_bounds__(Csm2::NW * 64) void csm_conv_s2_kernel(Csm2Args a) {
    constexpr int KB = Csm2::KB, MT = Csm2::MT, NW = Csm2::NW, CIN = Csm2::CIN, COUT = Csm2::COUT;
    constexpr int OC = CIN / 8;
    constexpr int NPC = 3 * 33 * OC;                          // pieces of a tile's patch: [row][slot: Ev 0..15 | Od 16..32][octet]
    constexpr int SS = NPC * 16;
    constexpr int NPL = (NPC + 63) / 64;
    __shared__ __attribute__((aligned(16))) u32x4 wlds[Csm2::NFRAG * 64];    // 30 KB
    __shared__ __attribute__((aligned(16))) char patch[NW][3 * SS];
    __shared__ double red[NW][COUT][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp);
    for (int i = tid; i < Csm2::NFRAG * 64; i += NW * 64) wlds[i] = wsrc[i];
    __syncthreads();
    float bias_r[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias_r[mt][i] = a.bias ? a.bias[mt * 16 + 4 * kq + i] : 0.f;
    double ssum[MT][4], ssq[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) ssum[mt][i] = ssq[mt][i] = 0.0;
    const int segs = (a.OW + 15) >> 4;
    const int ntiles = a.N * a.OH * segs;
    const int stride = (int)gridDim.x * NW;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
    char* const mp = patch[wave];
    int prow[NPL], pslot[NPL], poct[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int f = lane + 64 * k;
        prow[k] = f / (33 * OC);
        pslot[k] = (f / OC) % 33;
        poct[k] = f % OC;
    }
    int boff[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        int tap = 2 * kb + (kq >> 1);
        tap = tap < 9 ? tap : 8;                               // (tap 9 has zero weights: it re-reads tap 8's finite values)
        const int dy = tap / 3, dx = tap - dy * 3;
        const int slot = dx == 1 ? n : dx == 0 ? 16 + n : 17 + n;
        boff[kb] = ((dy * 33 + slot) * OC + (kq & 1)) * 16;
    }
    const float rS = 1.0f / (float)segs;
    auto geom = [&](int tile, int& img, int& oy, int& ox0) {
        int r = (int)((float)tile * rS);
        r -= (r * segs > tile);
        r += ((r + 1) * segs <= tile);
        ox0 = (tile - r * segs) * 16;
        img = (int)((float)r * a.rOH);
        img -= (img * a.OH > r);
        img += ((img + 1) * a.OH <= r);
        oy = r - img * a.OH;
    };
    float4 plo[2][NPL], phi[2][NPL];
    auto load_patch = [&](int tile, float4 (&lo)[NPL], float4 (&hi)[NPL]) {
        int img, oy, ox0;
        geom(tile, img, oy, ox0);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int f = lane + 64 * k;
            const int iy = 2 * oy + prow[k] - 1;
            const int ix = pslot[k] < 16 ? 2 * (ox0 + pslot[k]) : 2 * (ox0 + pslot[k] - 16) - 1;
            const bool ok = f < NPC && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const unsigned off = ok ? (unsigned)(((img * a.H + iy) * a.W + ix) * CIN + 8 * poct[k]) * 4u : 0x80000000u;
            lo[k] = buf_load16(srd, off);
            hi[k] = buf_load16(srd, off + 16u);
        }
    };
    const int tile0 = (int)blockIdx.x * NW + wave;
    if (tile0 < ntiles) load_patch(tile0, plo[0], phi[0]);
#pragma unroll 1
    for (int tile = tile0; tile < ntiles; tile += 2 * stride) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tcur = tile + half * stride;
            if (tcur >= ntiles) break;
            asm volatile("" ::: "memory");                   // the weight fragments are re-read from LDS per tile, not hoisted into registers
            if (tcur + stride < ntiles) load_patch(tcur + stride, plo[half ^ 1], phi[half ^ 1]);
            int img, oy, ox0;
            geom(tcur, img, oy, ox0);
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int f = lane + 64 * k;
                u32x4 s0, s1, s2;
                split8(plo[half][k], phi[half][k], s0, s1, s2);
                if (f < NPC) {
                    *reinterpret_cast<u32x4*>(mp + f * 16) = s0;
                    *reinterpret_cast<u32x4*>(mp + SS + f * 16) = s1;
                    *reinterpret_cast<u32x4*>(mp + 2 * SS + f * 16) = s2;
                }
            }
            __builtin_amdgcn_wave_barrier();
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(mp + boff[kb]);
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(mp + SS + boff[kb]);
                const u32x4 b2 = *reinterpret_cast<const u32x4*>(mp + 2 * SS + boff[kb]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u32x4 w0 = wlds[((0 * KB + kb) * MT + mt) * 64 + lane];
                    const u32x4 w1 = wlds[((1 * KB + kb) * MT + mt) * 64 + lane];
                    const u32x4 w2 = wlds[((2 * KB + kb) * MT + mt) * 64 + lane];
                    acc[mt] = mfma16(w0, b2, acc[mt]);
                    acc[mt] = mfma16(w2, b0, acc[mt]);
                    acc[mt] = mfma16(w1, b1, acc[mt]);
                    acc[mt] = mfma16(w0, b1, acc[mt]);
                    acc[mt] = mfma16(w1, b0, acc[mt]);
                    acc[mt] = mfma16(w0, b0, acc[mt]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const bool pvalid = ox0 + n < a.OW;
            const size_t p = (size_t)(img * a.OH + oy) * a.OW + ox0 + n;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = acc[mt][i] + bias_r[mt][i];
                    if (a.act) t = t > 0.f ? t : 0.2f * t;
                    v[i] = t;
                }
                if (a.keep) {
                    const float4 k = *reinterpret_cast<const float4*>(a.keep + (size_t)img * COUT + mt * 16 + 4 * kq);
                    v[0] *= k.x; v[1] *= k.y; v[2] *= k.z; v[3] *= k.w;
                }
                if (pvalid) {
                    *reinterpret_cast<float4*>(a.y + p * COUT + mt * 16 + 4 * kq) = make_float4(v[0], v[1], v[2], v[3]);
                    if (a.stat_part) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            ssum[mt][i] += (double)v[i];
                            ssq[mt][i] += (double)v[i] * (double)v[i];
                        }
                    }
                }
            }
        }
    }
    if (!a.stat_part) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ssum[mt][i] = row16_sum(ssum[mt][i]);
            ssq[mt][i] = row16_sum(ssq[mt][i]);
            if (n == 0) {
                red[wave][mt * 16 + 4 * kq + i][0] = ssum[mt][i];
                red[wave][mt * 16 + 4 * kq + i][1] = ssq[mt][i];
            }
        }
    __syncthreads();
    if (tid < 2 * COUT) {
        const int c = tid >> 1, which = tid & 1;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][c][which];
        a.stat_part[((size_t)blockIdx.x * COUT + c) * 2 + which] = t;
    }
}

// ---- data gradient of the 16 -> 32 stride-2 block --------------------------------------------------------------------------------
// dx[2 oy + py][2 ox + px][ci] = sum over the taps whose stride-2 window reaches that pixel and over co of g[oy'][ox'][co] w[co][tap][ci]:
// per parity class (py, px) the taps and their output pixels are STATIC -- (0, 0): tap (1, 1) of g(oy, ox); (0, 1): (1, 0) of
// g(oy, ox + 1), (1, 2) of g(oy, ox); (1, 0): (0, 1) of g(oy + 1, ox), (2, 1) of g(oy, ox); (1, 1): (0, 0) of g(oy + 1, ox + 1), (0, 2) of
// g(oy + 1, ox), (2, 0) of g(oy, ox + 1), (2, 2) of g(oy, ox) -- nine k-blocks of one tap x 32 output channels into four accumulator
// tiles, from ONE patch of g (2 rows x 17 pixels): one launch where conv_nhwc.hip ran one fp32-MFMA launch per class, each reading
// g (4 x 87 us in the I3D recipe).  Tile = 16 positions ox of one row oy = 64 input pixels; rows of the MFMA = the 16 input channels.
constexpr int CSM2D_TAP[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};      // k-block j: tap ky * 3 + kx
constexpr int CSM2D_A[9] = {0, 0, 0, 1, 0, 1, 1, 0, 0};        //            row oy + a
constexpr int CSM2D_B[9] = {0, 1, 0, 0, 0, 1, 0, 1, 0};        //            pixel ox + b
constexpr int CSM2D_CLS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};      //            class 2 py + px

__global__ __launch_bounds__(256) void csm_pack_d2_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp) {
    // fragment (slice, j): lane (row = ci, kq) holds k = 8 kq + e = co: w[co][tap_j][ci]  (w is [32][9][16], OHWI)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 9 * 16 * 32) return;
    const int kk = i & 31, row = (i >> 5) & 15, j = i >> 9;
    const float v = w[((size_t)kk * 9 + CSM2D_TAP[j]) * 16 + row];
    unsigned u0, u1, u2;
    split3(v, u0, u1, u2);
    const size_t e = (size_t)((kk >> 3) * 16 + row) * 8 + (kk & 7), f = (size_t)j * 512, sl = (size_t)9 * 512;
    wp[f + e] = (unsigned short)(u0 >> 16);
    wp[sl + f + e] = (unsigned short)(u1 >> 16);
    wp[2 * sl + f + e] = (unsigned short)(u2 >> 16);
}

struct Csm2dArgs {
    const float* g;        // [N][OH][OW][32]
    const void* wp;
    float* dx;             // [N][2 OH][2 OW][16]
    int N, OH, OW;
    float rOH;
};

__global__ __launch_bounds__(256) void csm_dgrad_s2_kernel(Csm2dArgs a) {
    constexpr int NW = 4, OCG = 4;                            // 8-channel pieces per g pixel
    constexpr int NPC = 2 * 17 * OCG;                         // pieces of a tile's patch: [row a][pixel 0 .. 16][octet]
    constexpr int SS = NPC * 16;
    constexpr int NPL = (NPC + 63) / 64;
    __shared__ __attribute__((aligned(16))) u32x4 wlds[27 * 64];             // 27 KB
    __shared__ __attribute__((aligned(16))) char patch[NW][3 * SS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wp);
    for (int i = tid; i < 27 * 64; i += NW * 64) wlds[i] = wsrc[i];
    __syncthreads();
    const int segs = (a.OW + 15) >> 4;
    const int ntiles = a.N * a.OH * segs;
    const int stride = (int)gridDim.x * NW;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, 0x7fffffff, 0x00020000);
    char* const mp = patch[wave];
    int prow[NPL], ppx[NPL], poct[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int f = lane + 64 * k;
        prow[k] = f / (17 * OCG);
        ppx[k] = (f / OCG) % 17;
        poct[k] = f % OCG;
    }
    int boff[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) boff[j] = ((CSM2D_A[j] * 17 + n + CSM2D_B[j]) * OCG + kq) * 16;
    const float rS = 1.0f / (float)segs;
    auto geom = [&](int tile, int& img, int& oy, int& ox0) {
        int r = (int)((float)tile * rS);
        r -= (r * segs > tile);
        r += ((r + 1) * segs <= tile);
        ox0 = (tile - r * segs) * 16;
        img = (int)((float)r * a.rOH);
        img -= (img * a.OH > r);
        img += ((img + 1) * a.OH <= r);
        oy = r - img * a.OH;
    };
    float4 plo[2][NPL], phi[2][NPL];
    auto load_patch = [&](int tile, float4 (&lo)[NPL], float4 (&hi)[NPL]) {
        int img, oy, ox0;
        geom(tile, img, oy, ox0);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int f = lane + 64 * k;
            const int gy = oy + prow[k], gx = ox0 + ppx[k];
            const bool ok = f < NPC && gy < a.OH && gx < a.OW;
            const unsigned off = ok ? (unsigned)(((img * a.OH + gy) * a.OW + gx) * 32 + 8 * poct[k]) * 4u : 0x80000000u;
            lo[k] = buf_load16(srd, off);
            hi[k] = buf_load16(srd, off + 16u);
        }
    };
    const int H = 2 * a.OH, W = 2 * a.OW;
    const int tile0 = (int)blockIdx.x * NW + wave;
    if (tile0 < ntiles) load_patch(tile0, plo[0], phi[0]);
#pragma unroll 1
    for (int tile = tile0; tile < ntiles; tile += 2 * stride) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tcur = tile + half * stride;
            if (tcur >= ntiles) break;
            asm volatile("" ::: "memory");                   // the weight fragments are re-read from LDS per tile, not hoisted into registers
            if (tcur + stride < ntiles) load_patch(tcur + stride, plo[half ^ 1], phi[half ^ 1]);
            int img, oy, ox0;
            geom(tcur, img, oy, ox0);
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int f = lane + 64 * k;
                u32x4 s0, s1, s2;
                split8(plo[half][k], phi[half][k], s0, s1, s2);
                if (f < NPC) {
                    *reinterpret_cast<u32x4*>(mp + f * 16) = s0;
                    *reinterpret_cast<u32x4*>(mp + SS + f * 16) = s1;
                    *reinterpret_cast<u32x4*>(mp + 2 * SS + f * 16) = s2;
                }
            }
            __builtin_amdgcn_wave_barrier();
            f32x4 acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(mp + boff[j]);
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(mp + SS + boff[j]);
                const u32x4 b2 = *reinterpret_cast<const u32x4*>(mp + 2 * SS + boff[j]);
                const u32x4 w0 = wlds[(0 * 9 + j) * 64 + lane], w1 = wlds[(1 * 9 + j) * 64 + lane], w2 = wlds[(2 * 9 + j) * 64 + lane];
                constexpr int dummy = 0; (void)dummy;
                f32x4& t = acc[CSM2D_CLS[j]];
                t = mfma16(w0, b2, t);
                t = mfma16(w2, b0, t);
                t = mfma16(w1, b1, t);
                t = mfma16(w0, b1, t);
                t = mfma16(w1, b0, t);
                t = mfma16(w0, b0, t);
            }
            __builtin_amdgcn_wave_barrier();
            // lane holds input channels 4 kq + i of position n: pixel (2 oy + py, 2 (ox0 + n) + px) of class c = 2 py + px
            if (ox0 + n < a.OW) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const size_t p = ((size_t)(img * H + 2 * oy + (c >> 1)) * W + 2 * (ox0 + n) + (c & 1));
                    *reinterpret_cast<float4*>(a.dx + p * 16 + 4 * kq) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
                }
            }
        }
    }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
// dw[co][tap][ci] = sum over pixels p of g[p][co] * x[p + d(tap)][ci] -- a GEMM over PIXELS (K) with M = co, N = ci per tap, on
// v_mfma_f32_16x16x32_bf16 in bf16x3 arithmetic.  A k-block = the 32 pixels of TWO 16-pixel row-segment tiles (k-quarters
// 0, 1 -> tile a, 2, 3 -> tile b).  Per tile pair a wave loads the two 3 x 18-pixel neighbourhoods of x and the two 16-pixel
// segments of g (range-checked buffer loads: zeros outside the image / beyond the row), splits every 8-channel piece once
// and parks the slices PIXEL-MAJOR in its mini patch in LDS ([slice][chunk of 16 channels][pixel][16 channels]): both MFMA
// operands then need "eight consecutive pixels of one channel per lane" -- ds_read_b64_tr_b16 (x3s_common.h) transposes
// on the way out of LDS, the nine taps are nine pixel offsets into the same patch.  The 9 x (C/16)^2 accumulator tiles
// stay in registers for the whole launch; partials per workgroup (waves summed in fixed order) and a fixed-order reduction
// over the workgroups: deterministic, no atomics.  (conv_wgrad_small_kernel, fp32 MFMA: 0.25-0.28 ms per launch at 240 / 120
// frames, at 0.7 of the fp32-MFMA peak.)
struct CsmWgArgs {
    const float* x;        // [M][C]
    const float* g;        // [M][C] gradient of the convolution output
    float* part;           // [gridDim.x][C][9][C]
    int N, H, W;
    float rH;
};

template <int C>
__global__ __launch_bounds__(256, C == 16 ? 3 : 1) void csm_wgrad_kernel(CsmWgArgs a) {
    constexpr int OC = C / 8, NCH = C / 16, MT = C / 16;
    constexpr int XPX = 2 * 54, GPX = 2 * 16;                 // pixels of a tile pair's x patches / g segments
    constexpr int XPL = XPX * 32, GPL = GPX * 32;             // bytes of one (slice, chunk) plane
    constexpr int XS = NCH * XPL, GS = NCH * GPL;             // ... of one slice
    constexpr int WB = 3 * (XS + GS);                         // a wave's mini patch: C = 16: 13,440 B, C = 32: 26,880 B
    constexpr int NPX_ = XPX * OC, NPG = GPX * OC;            // 8-channel pieces: x, g
    constexpr int NLX = (NPX_ + 63) / 64, NLG = NPG / 64, NPL = NLX + NLG;      // loads per lane: 4 + 1 (C = 16), 7 + 2 (C = 32)
    static_assert(NPG % 64 == 0, "g pieces fill whole waves");
    __shared__ __attribute__((aligned(16))) char patch[4][WB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = lane & 15, kq = lane >> 4;
    const int segs = (a.W + 15) >> 4;
    const int ntiles = a.N * a.H * segs;
    const int npairs = (ntiles + 1) >> 1;
    const int stride = (int)gridDim.x * 4;
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, 0x7fffffff, 0x00020000);
    char* const mp = patch[wave];
    const float rS = 1.0f / (float)segs;
    auto geom = [&](int tile, int& img, int& yy, int& x0) {
        int r = (int)((float)tile * rS);
        r -= (r * segs > tile);
        r += ((r + 1) * segs <= tile);
        x0 = (tile - r * segs) * 16;
        img = (int)((float)r * a.rH);
        img -= (img * a.H > r);
        img += ((img + 1) * a.H <= r);
        yy = r - img * a.H;
    };
    // this lane's pieces (loads 0 .. NLX-1: x piece = (tile, patch row, patch column, octet); NLX ..: g piece = (tile, pixel,
    // octet)), packed: tile | row << 1 | column << 3 | octet << 8; p_lds = byte offset within slice 0
    int p_geo[NPL], p_lds[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const bool isx = k < NLX;
        int f = lane + 64 * (isx ? k : k - NLX);
        if (isx && f >= NPX_) f = 0;                             // idle lanes of the last x load repeat piece 0 (never stored)
        const int q = f / OC, oct = f % OC;                      // pixel of the pair's patch / segment array
        const int tile = isx ? q / 54 : q / 16;
        const int row = isx ? (q % 54) / 18 : 1;                 // g: the tile's own row = patch row 1
        const int col = isx ? q % 18 : q % 16 + 1;               //    and pixel px = patch column px + 1
        p_geo[k] = tile | row << 1 | col << 3 | oct << 8;
        p_lds[k] = (isx ? 0 : 3 * XS) + (oct >> 1) * (isx ? XPL : GPL) + q * 32 + (oct & 1) * 16;
    }
    // D pairs in flight per wave (registers): a pair's loads are issued D - 1 pairs of split + MFMA work ahead of their use --
    // with D = 1 the 16-channel kernel (54 MFMAs per pair) ran at one memory round trip per pair
    constexpr int D = C == 16 ? 2 : 1;
    float4 plo[D][NPL], phi[D][NPL];
    auto load_pair = [&](int pair, float4 (&lo)[NPL], float4 (&hi)[NPL]) {
        int img[2], yy[2], x0[2];
        geom(2 * pair, img[0], yy[0], x0[0]);
        const bool second = 2 * pair + 1 < ntiles;
        geom(second ? 2 * pair + 1 : 2 * pair, img[1], yy[1], x0[1]);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int t = p_geo[k] & 1, row = (p_geo[k] >> 1) & 3, col = (p_geo[k] >> 3) & 31, oct = p_geo[k] >> 8;
            const int iy = (t ? yy[1] : yy[0]) + row - 1, ix = (t ? x0[1] : x0[0]) + col - 1;
            const bool ok = (t == 0 || second) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const unsigned off = ok ? (unsigned)((((t ? img[1] : img[0]) * a.H + iy) * a.W + ix) * C + 8 * oct) * 4u : 0x80000000u;
            if (k < NLX) {
                lo[k] = buf_load16(srd_x, off);
                hi[k] = buf_load16(srd_x, off + 16u);
            } else {
                lo[k] = buf_load16(srd_g, off);
                hi[k] = buf_load16(srd_g, off + 16u);
            }
        }
    };
    f32x4 acc[9][MT][MT];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // transposing reads: lane (L, kq) of a 16-lane group hands in pixel 8 (kq & 1) + L / 4 (+ 4) of tile kq >> 1, channels
    // 4 (L % 4) ..; it receives channel L of the four pixels = eight k-values of operand row L after two reads
    const int trb = (L & 3) * 8 + (8 * (kq & 1) + (L >> 2)) * 32;
    const int g_off = 3 * XS + (kq >> 1) * 16 * 32 + trb;                     // g: segment pixel 8 (kq & 1) + L / 4 of tile kq >> 1
    const int x_off = (kq >> 1) * 54 * 32 + trb;                              // x: patch row 0, column 0 of that tile (+ tap offset below)
    lds_cptr const LB = (lds_cptr)mp;
    const int pair0 = (int)blockIdx.x * 4 + wave;
#pragma unroll
    for (int h = 0; h < D; ++h)
        if (pair0 + h * stride < npairs) load_pair(pair0 + h * stride, plo[h], phi[h]);
#pragma unroll 1
    for (int pb = pair0; pb < npairs; pb += D * stride) {
#pragma unroll
        for (int h = 0; h < D; ++h) {
            const int pc = pb + h * stride;
            if (pc >= npairs) break;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                u32x4 s0, s1, s2;
                split8(plo[h][k], phi[h][k], s0, s1, s2);
                if (k != NLX - 1 || NPX_ % 64 == 0 || lane + 64 * k < NPX_) {
                    const int ss = k < NLX ? XS : GS;
                    *reinterpret_cast<u32x4*>(mp + p_lds[k]) = s0;
                    *reinterpret_cast<u32x4*>(mp + p_lds[k] + ss) = s1;
                    *reinterpret_cast<u32x4*>(mp + p_lds[k] + 2 * ss) = s2;
                }
            }
            if (pc + D * stride < npairs) load_pair(pc + D * stride, plo[h], phi[h]);   // into the registers just consumed
            __builtin_amdgcn_wave_barrier();
            u32x4 A[MT][3];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int sl = 0; sl < 3; ++sl)
                    tr_read2(LB + g_off + sl * GS + i * GPL, LB + g_off + sl * GS + i * GPL + 4 * 32, A[i][sl]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3) * 18 + (t % 3)) * 32;                // patch row dy, column dx of the tap
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    u32x4 B[3];
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl)
                        tr_read2(LB + x_off + toff + sl * XS + j * XPL, LB + x_off + toff + sl * XS + j * XPL + 4 * 32, B[sl]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        acc[t][i][j] = mfma16(A[i][0], B[2], acc[t][i][j]);
                        acc[t][i][j] = mfma16(A[i][2], B[0], acc[t][i][j]);
                        acc[t][i][j] = mfma16(A[i][1], B[1], acc[t][i][j]);
                        acc[t][i][j] = mfma16(A[i][0], B[1], acc[t][i][j]);
                        acc[t][i][j] = mfma16(A[i][1], B[0], acc[t][i][j]);
                        acc[t][i][j] = mfma16(A[i][0], B[0], acc[t][i][j]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();                 // the next pair's stores follow this pair's reads
        }
    }
    // waves summed in fixed order through LDS (the patches are dead), one partial per workgroup:
    // lane (L = ci column, kq) holds rows co = 16 i + 4 kq + e of column ci = 16 j + L
    __syncthreads();
    float* red = reinterpret_cast<float*>(&patch[0][0]);        // [C][9][C] floats: 9.2 / 36.9 KB
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int idx = ((16 * i + 4 * kq + e) * 9 + t) * C + 16 * j + L;
                            red[idx] = (w == 0 ? 0.f : red[idx]) + acc[t][i][j][e];
                        }
        }
        __syncthreads();
    }
    float* out = a.part + (size_t)blockIdx.x * C * 9 * C;
    for (int i = tid; i < C * 9 * C; i += 256) out[i] = red[i];
}

// dw = sum over workgroups of the partials, fixed order: 16 group lanes x 16 float4 columns per workgroup (group lane sl sums
// groups sl, sl + 16, ..; the 16 lane sums are added in lane order)
__global__ __launch_bounds__(256) void csm_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int ngroup, int numel) {
    __shared__ float4 red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = (int)blockIdx.x * 64 + 4 * o;                    // numel = 9 C^2 is a multiple of 64
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = sl; k < ngroup; k += 16) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * numel + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0) {
        float4 t = red[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(dw + i) = t;
    }
}

// ---- weight gradient of the stride-2 blocks (16 -> 32 on 112 x 112 outputs... of 224 x 224 inputs, 32 -> 64) --------------------------
// dw[co][ky][kx][ci] = sum over output pixels (n, oy, ox) of g[n][oy][ox][co] * x[n][2 oy + ky - 1][2 ox + kx - 1][ci] (3 x 3, stride 2,
// padding 1, even H and W; Cout = 2 Cin): the same GEMM over pixels as csm_wgrad_kernel, with the stride taken out of the operand
// reads by the STAGING: of the three input rows 2 oy - 1 .. 2 oy + 1 a tile's patch keeps the even columns Ev[i] = x[2 (ox0 + i)]
// (16 slots) and the odd columns Od[i] = x[2 (ox0 + i) - 1] (17 slots) apart, so that tap kx = 1 reads Ev[i], kx = 0 reads Od[i] and
// kx = 2 reads Od[i + 1] for output pixel i -- eight consecutive output pixels are eight consecutive slots, which is what the
// transposing LDS read wants.  (conv_wgrad_small_kernel<., ., 2>, fp32 MFMA: 0.43 / 0.25 ms per launch for the I3D recipe's 192
// frames, 0.28 / 0.15 at 240 / 120 frames of config 3.)
// A WORKGROUP works on one tile pair at a time (k-block = 32 output pixels): all threads load and split the pair's pieces (double-
// buffered patch, one barrier per pair), and the waves divide the OUTPUT -- Cin 32: wave = (16 output channels, 16 input channels)
// x all nine taps; Cin 16: wave = (16 output channels, taps 0-4 or 5-8) -- so no wave-level reduction is needed and the
// per-workgroup partials go straight to the fixed-order reduction.
struct CsmWg2Args {
    const float* x;        // [N][H][W][CIN]
    const float* g;        // [N][H/2][W/2][2 CIN]
    float* part;           // [gridDim.x][2 CIN][9][CIN]
    int N, H, W, OH, OW;
    float rOH;
};

template <int CIN>
__global__ __launch_bounds__(CIN == 32 ? 512 : 256) void csm_wgrad_s2_kernel(CsmWg2Args a) {
    constexpr int COUT = 2 * CIN, NT = CIN == 32 ? 512 : 256;
    constexpr int OCX = CIN / 8, OCG = COUT / 8;
    constexpr int XPL = 2 * 99 * 32, GPL = 32 * 32;          // bytes of one (slice, 16-channel chunk) plane: x (two tiles x 99 slots), g (32 pixels)
    constexpr int XS = (CIN / 16) * XPL, GS = (COUT / 16) * GPL;
    constexpr int BUF = 3 * (XS + GS);                        // Cin 32: 50,304 B; Cin 16: 25,152 B
    constexpr int NPX = 2 * 99 * OCX, NPG = 32 * OCG;         // pieces of a pair: x, g
    constexpr int NLX = (NPX + NT - 1) / NT, NLG = (NPG + NT - 1) / NT, NPL = NLX + NLG;
    constexpr int NACC = CIN == 32 ? 9 : 5;
    __shared__ __attribute__((aligned(16))) char lds[2][BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = lane & 15, kq = lane >> 4;
    const int segs = (a.OW + 15) >> 4;
    const int ntiles = a.N * a.OH * segs;
    const int npairs = (ntiles + 1) >> 1;
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, 0x7fffffff, 0x00020000);
    const float rS = 1.0f / (float)segs;
    auto geom = [&](int tile, int& img, int& oy, int& ox0) {
        int r = (int)((float)tile * rS);
        r -= (r * segs > tile);
        r += ((r + 1) * segs <= tile);
        ox0 = (tile - r * segs) * 16;
        img = (int)((float)r * a.rOH);
        img -= (img * a.OH > r);
        img += ((img + 1) * a.OH <= r);
        oy = r - img * a.OH;
    };
    // this thread's pieces: loads 0 .. NLX-1 = x piece (tile, row, slot, octet), NLX .. = g piece (pixel of the pair, octet);
    // packed tile | row << 1 | slot << 3 | octet << 9 (x) resp. pixel | octet << 9 (g); p_lds = byte offset within slice 0
    int p_geo[NPL], p_lds[NPL];
    bool p_on[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const bool isx = k < NLX;
        const int f = tid + NT * (isx ? k : k - NLX);
        p_on[k] = f < (isx ? NPX : NPG);
        const int ff = p_on[k] ? f : 0;
        if (isx) {
            const int q = ff / OCX, oct = ff % OCX;             // q = tile * 99 + row * 33 + slot
            const int tile = q / 99, row = (q % 99) / 33, slot = q % 33;
            p_geo[k] = tile | row << 1 | slot << 3 | oct << 9;
            p_lds[k] = (oct >> 1) * XPL + q * 32 + (oct & 1) * 16;
        } else {
            const int px = ff / OCG, oct = ff % OCG;
            p_geo[k] = px | oct << 9;
            p_lds[k] = 3 * XS + (oct >> 1) * GPL + px * 32 + (oct & 1) * 16;
        }
    }
    float4 plo[NPL], phi[NPL];
    auto load_pair = [&](int pair) {
        int img[2], oy[2], ox0[2];
        geom(2 * pair, img[0], oy[0], ox0[0]);
        const bool second = 2 * pair + 1 < ntiles;
        geom(second ? 2 * pair + 1 : 2 * pair, img[1], oy[1], ox0[1]);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            unsigned off = 0x80000000u;
            if (k < NLX) {
                const int t = p_geo[k] & 1, row = (p_geo[k] >> 1) & 3, slot = (p_geo[k] >> 3) & 63, oct = p_geo[k] >> 9;
                const int iy = 2 * (t ? oy[1] : oy[0]) + row - 1;
                const int ix = slot < 16 ? 2 * ((t ? ox0[1] : ox0[0]) + slot) : 2 * ((t ? ox0[1] : ox0[0]) + slot - 16) - 1;
                if (p_on[k] && (t == 0 || second) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                    off = (unsigned)((((t ? img[1] : img[0]) * a.H + iy) * a.W + ix) * CIN + 8 * oct) * 4u;
                plo[k] = buf_load16(srd_x, off);
                phi[k] = buf_load16(srd_x, off + 16u);
            } else {
                const int px = p_geo[k] & 511, oct = p_geo[k] >> 9, t = px >> 4;
                const int ox = (t ? ox0[1] : ox0[0]) + (px & 15);
                if (p_on[k] && (t == 0 || second) && ox < a.OW)
                    off = (unsigned)((((t ? img[1] : img[0]) * a.OH + (t ? oy[1] : oy[0])) * a.OW + ox) * COUT + 8 * oct) * 4u;
                plo[k] = buf_load16(srd_g, off);
                phi[k] = buf_load16(srd_g, off + 16u);
            }
        }
    };
    auto store_pair = [&](int b) {
        char* const mp = lds[b];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            u32x4 s0, s1, s2;
            split8(plo[k], phi[k], s0, s1, s2);
            if (p_on[k]) {
                const int ss = k < NLX ? XS : GS;
                *reinterpret_cast<u32x4*>(mp + p_lds[k]) = s0;
                *reinterpret_cast<u32x4*>(mp + p_lds[k] + ss) = s1;
                *reinterpret_cast<u32x4*>(mp + p_lds[k] + 2 * ss) = s2;
            }
        }
    };
    // this wave's share of dw: output-channel tile cot, input-channel tile cit, taps t0 .. t0 + nt - 1
    const int cot = CIN == 32 ? (wave & 3) : (wave & 1);
    const int cit = CIN == 32 ? (wave >> 2) : 0;
    const int t0 = CIN == 32 ? 0 : (wave >> 1) * 5;
    const int nt = CIN == 32 ? 9 : ((wave >> 1) ? 4 : 5);
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // transposing reads (x3s_common.h): lane (L, kq) hands in pixel 8 (kq & 1) + L / 4 (+ 4) of tile kq >> 1, channels 4 (L % 4) ..
    const int trb = (L & 3) * 8 + (8 * (kq & 1) + (L >> 2)) * 32;
    const int g_off = 3 * XS + cot * GPL + (kq >> 1) * 16 * 32 + trb;
    const int x_off = cit * XPL + (kq >> 1) * 99 * 32 + trb;
    int toff[NACC];                                            // tap (ky, kx): row ky, slot base Od (kx = 0), Ev (1), Od + 1 (2)
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int t = t0 + (i < nt ? i : 0), ky = t / 3, kx = t - 3 * ky;
        toff[i] = (ky * 33 + (kx == 1 ? 0 : kx == 0 ? 16 : 17)) * 32;
    }

    int pair = (int)blockIdx.x, it = 0;
    if (pair < npairs) { load_pair(pair); store_pair(0); }
    __syncthreads();
#pragma unroll 1
    for (; pair < npairs; pair += (int)gridDim.x, ++it) {
        const int next = pair + (int)gridDim.x;
        if (next < npairs) load_pair(next);                     // in flight under this pair's MFMAs
        lds_cptr const LB = (lds_cptr)lds[it & 1];
        u32x4 A[3];
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) tr_read2(LB + g_off + sl * GS, LB + g_off + sl * GS + 4 * 32, A[sl]);
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (i >= nt) break;                                 // (wave-uniform)
            u32x4 B[3];
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) tr_read2(LB + x_off + toff[i] + sl * XS, LB + x_off + toff[i] + sl * XS + 4 * 32, B[sl]);
            acc[i] = mfma16(A[0], B[2], acc[i]);
            acc[i] = mfma16(A[2], B[0], acc[i]);
            acc[i] = mfma16(A[1], B[1], acc[i]);
            acc[i] = mfma16(A[0], B[1], acc[i]);
            acc[i] = mfma16(A[1], B[0], acc[i]);
            acc[i] = mfma16(A[0], B[0], acc[i]);
        }
        if (next < npairs) store_pair((it + 1) & 1);             // the buffer the pair before this one was read from
        __syncthreads();
    }
    // lane (L = ci column, kq) holds rows co = 16 cot + 4 kq + e of column ci = 16 cit + L
    float* out = a.part + (size_t)blockIdx.x * COUT * 9 * CIN;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        if (i >= nt) break;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[((16 * cot + 4 * kq + e) * 9 + t0 + i) * CIN + 16 * cit + L] = acc[i][e];
    }
}

// w [Cout][9][Cin] fp32 (OHWI) -> fragment order [3 slices][KB][MT][16 rows][32 k] bf16
//   forward:        row = co, k-block value (tap, ci) = w[co][tap][ci]
//   data gradient:  row = ci, k-block value (tap, co) = w[co][8 - tap][ci]          (mirrored taps)
template <int C>
__global__ __launch_bounds__(256) void csm_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int transposed) {
    using G = Csm<C>;
    const int total = G::KB * G::MT * 16 * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int t = i;
        const int kk = t & 31; t >>= 5;
        const int row = t & 15; t >>= 4;
        const int mt = t % G::MT;
        const int kb = t / G::MT;
        const int tap = C == 16 ? 2 * kb + (kk >> 4) : kb;
        const int kc = C == 16 ? (kk & 15) : kk;                    // contraction channel
        const int r = mt * 16 + row;                                // result channel of this direction
        float v = 0.f;
        if (tap < 9) v = transposed ? w[((size_t)kc * 9 + (8 - tap)) * C + r] : w[((size_t)r * 9 + tap) * C + kc];
        unsigned u0, u1, u2;
        split3(v, u0, u1, u2);
        // fragment (slice, kb, mt): lane (row, kq = kk / 8) holds eight k-values: element index within the fragment
        const size_t e = (size_t)((kk >> 3) * 16 + row) * 8 + (kk & 7);
        const size_t f = (size_t)(kb * G::MT + mt) * 512;
        const size_t sl = (size_t)G::KB * G::MT * 512;
        wp[f + e] = (unsigned short)(u0 >> 16);
        wp[sl + f + e] = (unsigned short)(u1 >> 16);
        wp[2 * sl + f + e] = (unsigned short)(u2 >> 16);
    }
}

template <int C>
int csm_launch(const float* x, const float* w, void* wpack, const float* bias, const float* keep, float* y, double* stat_part,
               int stat_blocks, int N, int H, int W, int act, int transposed, hipStream_t s) {
    csm_pack_kernel<C><<<8, 256, 0, s>>>(w, static_cast<unsigned short*>(wpack), transposed);
    int rc = check_launch("csm_pack");
    if (rc) return rc;
    CsmArgs a;
    a.x = x; a.wp = wpack; a.bias = bias; a.keep = keep; a.y = y; a.stat_part = stat_part;
    a.N = N; a.H = H; a.W = W; a.M = N * H * W; a.act = act;
    a.rW = 1.0f / (float)W; a.rH = 1.0f / (float)H;
    const int grid = dmc::csm_stat_blocks(N, H, W, C);
    if (stat_part && stat_blocks != grid)
        return fail(DMC_E_INVALID, "conv_small: statistics partials have %d rows but this launch writes %d", stat_blocks, grid);
    csm_conv_kernel<C><<<grid, Csm<C>::NW * 64, 0, s>>>(a);
    return check_launch("csm_conv");
}

}  // namespace

namespace dmc {

bool csm_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (option(OPT_CONV_ARITH) != 1 || option(OPT_CONV_PATH) != 1 || option(OPT_CONV_CFG) == 301) return false;   // 301: off (A/B runs)
    if (Cin != Cout || (Cin != 16 && Cin != 32) || KH != 3 || KW != 3 || stride != 1 || pad != 1) return false;
    return N > 0 && H > 0 && W > 0 && (long)N * H * W < (1L << 24) && (long)N * H * W * Cin * 4 < (1L << 31);
}

// persistent waves: enough workgroups for ~8 per CU, at most one 16-pixel tile per wave and round below that
int csm_stat_blocks(int N, int H, int W, int C) {
    const long tiles = (long)N * H * ((W + 15) / 16);
    const int nw = C == 16 ? 4 : 8;
    long g = (tiles + nw - 1) / nw;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

size_t csm_wpack_bytes(int C) { return (size_t)3 * (C == 16 ? 5 : 9) * (C / 16) * 512 * 2; }

int csm_fwd(const float* x, const float* w, void* wpack, const float* bias, const float* keep, float* y, double* stat_part,
            int stat_blocks, int N, int H, int W, int C, int act, hipStream_t s) {
    return C == 16 ? csm_launch<16>(x, w, wpack, bias, keep, y, stat_part, stat_blocks, N, H, W, act, 0, s)
                   : csm_launch<32>(x, w, wpack, bias, keep, y, stat_part, stat_blocks, N, H, W, act, 0, s);
}

bool csm_fwd_s2_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (option(OPT_CONV_ARITH) != 1 || option(OPT_CONV_PATH) != 1 || option(OPT_CONV_CFG) == 301 || option(OPT_CONV_CFG) == 303) return false;   // 303: off (A/B)
    if (Cin != 16 || Cout != 32 || KH != 3 || KW != 3 || stride != 2 || pad != 1 || (H & 1) || (W & 1)) return false;
    return N > 0 && H > 0 && W > 0 && (long)N * H * W < (1L << 24) && (long)N * H * W * Cin * 4 < (1L << 31);
}

int csm_fwd_s2_stat_blocks(int N, int H, int W) {
    const long tiles = (long)N * (H / 2) * ((W / 2 + 15) / 16);
    long g = (tiles + Csm2::NW - 1) / Csm2::NW;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

size_t csm_fwd_s2_wpack_bytes() { return (size_t)Csm2::NFRAG * 1024; }

int csm_fwd_s2(const float* x, const float* w, void* wpack, const float* bias, const float* keep, float* y, double* stat_part,
               int stat_blocks, int N, int H, int W, int act, hipStream_t s) {
    csm_pack_s2_kernel<<<8, 256, 0, s>>>(w, static_cast<unsigned short*>(wpack));
    int rc = check_launch("csm_pack_s2");
    if (rc) return rc;
    Csm2Args a;
    a.x = x; a.wp = wpack; a.bias = bias; a.keep = keep; a.y = y; a.stat_part = stat_part;
    a.N = N; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.act = act; a.rOH = 1.0f / (float)a.OH;
    const int grid = csm_fwd_s2_stat_blocks(N, H, W);
    if (stat_part && stat_blocks != grid)
        return fail(DMC_E_INVALID, "conv_small: statistics partials have %d rows but this launch writes %d", stat_blocks, grid);
    csm_conv_s2_kernel<<<grid, Csm2::NW * 64, 0, s>>>(a);
    return check_launch("csm_conv_s2");
}

bool csm_dgrad_s2_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (option(OPT_CONV_CFG) == 304) return false;          // 304: off (A/B)
    return csm_fwd_s2_supported(N, H, W, Cin, Cout, KH, KW, stride, pad) && (long)N * (H / 2) * (W / 2) * 32 * 4 < (1L << 31);
}

size_t csm_dgrad_s2_wpack_bytes() { return (size_t)27 * 1024; }

// dx [N][H][W][16] from dy [N][H/2][W/2][32], w [32][9][16] (OHWI); wpack >= csm_dgrad_s2_wpack_bytes()
int csm_dgrad_s2(const float* dy, const float* w, void* wpack, float* dx, int N, int H, int W, hipStream_t s) {
    csm_pack_d2_kernel<<<(9 * 16 * 32 + 255) / 256, 256, 0, s>>>(w, static_cast<unsigned short*>(wpack));
    int rc = check_launch("csm_pack_d2");
    if (rc) return rc;
    Csm2dArgs a;
    a.g = dy; a.wp = wpack; a.dx = dx; a.N = N; a.OH = H / 2; a.OW = W / 2; a.rOH = 1.0f / (float)a.OH;
    const long tiles = (long)N * a.OH * ((a.OW + 15) / 16);
    long g = (tiles + 3) / 4;
    g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    csm_dgrad_s2_kernel<<<(int)g, 256, 0, s>>>(a);
    return check_launch("csm_dgrad_s2");
}

int csm_wgrad_groups(int N, int H, int W, int C) {
    const long pairs = ((long)N * H * ((W + 15) / 16) + 1) / 2;
    const long cap = C == 16 ? 768 : 256;                  // resident workgroups: 3 / 1 per CU (53.8 / 107.5 KB of LDS)
    long g = (pairs + 3) / 4;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

int csm_wgrad(const float* x, const float* g, float* dw, float* workspace, int N, int H, int W, int C, hipStream_t s) {
    CsmWgArgs a;
    a.x = x; a.g = g; a.part = workspace; a.N = N; a.H = H; a.W = W; a.rH = 1.0f / (float)H;
    const int groups = csm_wgrad_groups(N, H, W, C);
    if (C == 16) csm_wgrad_kernel<16><<<groups, 256, 0, s>>>(a);
    else csm_wgrad_kernel<32><<<groups, 256, 0, s>>>(a);
    int rc = check_launch("csm_wgrad");
    if (rc) return rc;
    const int numel = C * 9 * C;
    csm_wgrad_reduce_kernel<<<numel / 64, 256, 0, s>>>(workspace, dw, groups, numel);
    return check_launch("csm_wgrad_reduce");
}

bool csm_wgrad_s2_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (option(OPT_CONV_ARITH) != 1 || option(OPT_CONV_PATH) != 1 || option(OPT_CONV_CFG) == 301 || option(OPT_CONV_CFG) == 302) return false;
    if ((Cin != 16 && Cin != 32) || Cout != 2 * Cin || KH != 3 || KW != 3 || stride != 2 || pad != 1 || (H & 1) || (W & 1)) return false;
    return N > 0 && H > 0 && W > 0 && (long)N * H * W < (1L << 24) && (long)N * H * W * Cin * 4 < (1L << 31);
}

int csm_wgrad_s2_groups(int N, int H, int W, int Cin) {
    const long pairs = ((long)N * (H / 2) * ((W / 2 + 15) / 16) + 1) / 2;
    const long cap = Cin == 16 ? 512 : 256;                 // resident workgroups (50 / 101 KB of LDS); <= 512 partials fit the workspace
    return (int)(pairs < 1 ? 1 : (pairs > cap ? cap : pairs));
}

int csm_wgrad_s2(const float* x, const float* g, float* dw, float* workspace, int N, int H, int W, int Cin, hipStream_t s) {
    CsmWg2Args a;
    a.x = x; a.g = g; a.part = workspace; a.N = N; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.rOH = 1.0f / (float)a.OH;
    const int groups = csm_wgrad_s2_groups(N, H, W, Cin);
    if (Cin == 16) csm_wgrad_s2_kernel<16><<<groups, 256, 0, s>>>(a);
    else csm_wgrad_s2_kernel<32><<<groups, 512, 0, s>>>(a);
    int rc = check_launch("csm_wgrad_s2");
    if (rc) return rc;
    const int numel = 2 * Cin * 9 * Cin;
    csm_wgrad_reduce_kernel<<<numel / 64, 256, 0, s>>>(workspace, dw, groups, numel);
    return check_launch("csm_wgrad_s2_reduce");
}

int csm_dgrad(const float* dy, const float* w, void* wpack, float* dx, int N, int H, int W, int C, hipStream_t s) {
    return C == 16 ? csm_launch<16>(dy, w, wpack, nullptr, nullptr, dx, nullptr, 0, N, H, W, 0, 1, s)
                   : csm_launch<32>(dy, w, wpack, nullptr, nullptr, dx, nullptr, 0, N, H, W, 0, 1, s);
}

}  // namespace dmc
