// EstimatorDenseNetTiny hidden layers 0-2 in bf16x3 arithmetic on v_mfma_f32_16x16x32_bf16 (gfx950).
//
// Reference behaviour: code/dmcnet/model.py:172-194 (EstimatorDenseNetTiny: conv 3x3 + LeakyReLU(0.1), dense concatenation),
// :111-119 (conv).  The fp32 kernels of gen_tiny.hip run these layers on v_mfma_f32_4x4x1 -- the only fp32 MFMA shape that fits
// 6 / 8 output channels -- and are bound by their consumer loop (LDS operand reads + 9-cycle MFMAs + the push epilogue), not by
// memory.  Here an fp32 value is the exact sum of three bf16 slices (conv_nhwc.hip), six slice products per
// product block, fp32 accumulate: fp32-level error at 2.67x fewer matrix cycles, and the 16 x 16 x 32 shape is filled by
// putting the VERTICAL taps into the rows:
//
//   D[(dy, co)][pixel] += A[(dy, co)][(dx, ci)] * B[(dx, ci)][pixel]      for one INPUT row r
//
//   rows    = (dy, co): 3 x 8 = 24 -> two row tiles (tile 0: dy 0 | dy 1, tile 1: dy 2 | empty); Cout 6 pads to 8
//   columns = 16 consecutive pixels of the row
//   k       = (dx, ci): lane (pixel n, quarter kq) of the B operand holds eight consecutive channels of pixel n + dx - 1 --
//             one 16-byte read of a [pixel][channel] mini patch; a k-block = four (dx, 8-channel chunk) groups
//
// Input row r contributes to output rows r + 1 (dy 0), r (dy 1), r - 1 (dy 2): a wave walks DOWN its strip of 32 columns
// and chains the accumulators -- after the pass over row r, tile 0 holds [partial of row r + 1 | partial of row r]; its
// halves swap into the C operands of the next pass ([0 | partial of r + 1] for tile 0, [partial of r | 0] for tile 1), whose
// tile 1 then completes row r: every output row receives its three vertical taps in three consecutive passes without
// leaving the matrix core's registers.  Cross-lane traffic: one lane-half exchange per accumulator register and pass.
// The weights (A fragments: 6 / 12 / 18 of 1 KB) stay in REGISTERS for the whole strip, so every B fragment read from LDS
// feeds 12 MFMAs (two row tiles x six slice products): the LDS pipe runs at half of the matrix pipe's demand limit.
//
// Everything is wave-private: a wave loads the 34 pixels x CIN planes of its next input row straight from the planar fp32
// tensors (one dword per lane and plane: 136 contiguous bytes, scalar plane bases, one per-lane offset for all planes),
// splits each 8-channel piece ONCE into the three slices and parks them in its own [slice][pixel][channel] row buffer in LDS
// (pixel stride = an odd number of 16-byte words: conflict-free) -- no workgroup barrier anywhere.  Output: + bias,
// LeakyReLU(0.1), planar stores (64-byte runs per plane and instruction).
#include "dmc_common.h"
#include "x3s_common.h"
#include "gen_x3.h"

using namespace dmc;
using namespace dmc::x3;

namespace {

constexpr int GX_RS = 32;             // output rows per strip
constexpr int GX_SW = 28;             // output columns per strip: 28 + 2 halo pixels = 30 of a half wave's 32 lanes
constexpr int GX_WAVES = 4;

template <int K> struct GX {
    static constexpr int CIN = cin_of(K), COUT = cout_of(K);
    static constexpr int NCH = (CIN + 7) / 8;               // 8-channel chunks: 1, 2, 3
    static constexpr int G = 3 * NCH, KB = (G + 3) / 4;      // (dx, chunk) groups; k-blocks of four groups: 1, 2, 3
    static constexpr int RT = 2;
    static constexpr int PS = 16 * NCH + (NCH % 2 == 0 ? 16 : 0);      // pixel stride in bytes: 16, 48, 48
    static constexpr int PL = 34 * PS, RB = 3 * PL;          // slice plane / row buffer of a wave
    static_assert(RT * KB * 3 == gx_nfrag(K), "fragment table of gen_x3.h");
};
struct GenX3Args {
    const float* mv;       // [N][2][H][W]
    const float* res;      // [N][3][H][W]
    float* feat;           // [N][28][H][W]: y_0 .. y_{K-1} read, y_K written
    const u32x4* frag;     // this layer's A fragments
    const float* bias;     // [COUT]
    int H, W, cstrips, rstrips, ntasks, nwg, wg_per_xcd;
};

// the value of lane ^ 32
__device__ __forceinline__ float other_half(float v) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v)));
}

template <int K>
__global__ __launch_bounds__(GX_WAVES * 64, 2) void gen_x3_kernel(GenX3Args a) {
    using X = GX<K>;
    constexpr int CIN = X::CIN, COUT = X::COUT, NCH = X::NCH, KB = X::KB, PS = X::PS, PL = X::PL, RB = X::RB;
    __shared__ __attribute__((aligned(16))) char lds[GX_WAVES][2][RB];
    const int lane = threadIdx.x & 63, L = lane & 15, kq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // consecutive workgroups (neighbouring strips of one frame) on ONE XCD: its L2 serves the halo columns and rows
    const int wg = ((int)blockIdx.x & 7) * a.wg_per_xcd + ((int)blockIdx.x >> 3);
    if (wg >= a.nwg) return;
    const int task = wg * GX_WAVES + wv;
    if (task >= a.ntasks) return;
    const int cs = task % a.cstrips, t2 = task / a.cstrips;
    const int rs = t2 % a.rstrips, n = t2 / a.rstrips;
    const int x0 = cs * GX_SW, y0 = rs * GX_RS;
    const int rend = y0 + GX_RS < a.H ? y0 + GX_RS : a.H;          // output rows [y0, rend)
    const size_t HW = (size_t)a.H * a.W;
    char* const mp = &lds[wv][0][0];
    // patch slots 30 .. 33 are read by the last column tile's unused columns and never written: finite values once
    for (int i = lane; i < 2 * RB / 16; i += 64) reinterpret_cast<u32x4*>(mp)[i] = (u32x4){0u, 0u, 0u, 0u};

    const float* pb[CIN];                                    // plane bases (wave-uniform)
#pragma unroll
    for (int p = 0; p < CIN; ++p)
        pb[p] = p < 2 ? a.mv + ((size_t)n * 2 + p) * HW : p < NIN ? a.res + ((size_t)n * 3 + (p - 2)) * HW
                                                                  : a.feat + ((size_t)n * NFEAT + (p - NIN)) * HW;
    u32x4 A[2][KB][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) A[rt][kb][sl] = a.frag[((rt * KB + kb) * 3 + sl) * 64 + lane];

    // loads: TWO input rows per pass -- lane = (row hl = lane / 32, patch pixel pl = lane % 32 < 30: x0 - 1 + pl); one dword per
    // plane; lanes outside the image read pixel 0 of the plane (branch-free) and their patch slots are zeroed afterwards
    const int hl = lane >> 5, pl = lane & 31;
    const int xl = x0 - 1 + pl;
    const bool xok = pl < 30 && xl >= 0 && xl < a.W;
    char* const wr = mp + hl * RB + pl * PS;                 // this lane's slots: row buffer hl, pixel pl
    float v[NCH][8];
    bool vfix = false;
    auto load_rows = [&](int r) {
        const int rr = r + hl;
        const bool ok = xok && rr >= 0 && rr < a.H;
        const int off = ok ? rr * a.W + xl : 0;
        vfix = pl < 30 && !ok;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = (8 * c + j < CIN) ? pb[8 * c + j < CIN ? 8 * c + j : 0][off] : 0.f;
    };
    auto store_rows = [&]() {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            u32x4 s0, s1, s2;
            split8(make_float4(v[c][0], v[c][1], v[c][2], v[c][3]), make_float4(v[c][4], v[c][5], v[c][6], v[c][7]), s0, s1, s2);
            if (pl < 30) {
                *reinterpret_cast<u32x4*>(wr + 16 * c) = s0;
                *reinterpret_cast<u32x4*>(wr + 16 * c + PL) = s1;
                *reinterpret_cast<u32x4*>(wr + 16 * c + 2 * PL) = s2;
            }
        }
        if (__builtin_amdgcn_ballot_w64(vfix) != 0ull) {           // strips at the image border only
            if (vfix)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) *reinterpret_cast<u32x4*>(wr + 16 * c + sl * PL) = (u32x4){0u, 0u, 0u, 0u};
        }
    };
    // B fragment of k-block kb, column tile ct: 16 bytes at patch pixel 16 ct + L + dx, chunk c; groups beyond G (zero weights)
    // read group 0's data (finite)
    int boff[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int g = 4 * kb + kq;
        boff[kb] = g < X::G ? (L + g / NCH) * PS + 16 * (g % NCH) : L * PS;
    }
    float bq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bq[e] = a.bias[(4 * (kq & 1) + e) < COUT ? 4 * (kq & 1) + e : 0];
    // output: lanes 0 .. 31 = (channel group kq, column L); one pointer per lane, channels / column tiles are constant offsets
    float* const outp = a.feat + ((size_t)n * NFEAT + (yoff(K) - NIN) + 4 * (kq & 1)) * HW + x0 + L;
    const bool st0 = lane < 32 && x0 + L < a.W, st1 = lane < 32 && L < GX_SW - 16 && x0 + 16 + L < a.W;

    f32x4 C0[2], C1[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        C0[ct] = lane < 32 ? (f32x4){bq[0], bq[1], bq[2], bq[3]} : (f32x4){0.f, 0.f, 0.f, 0.f};     // the bias enters with the dy-0 partial
        C1[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    load_rows(y0 - 1);
#pragma unroll 1
    for (int rb = y0 - 1; rb <= rend; rb += 2) {
        store_rows();
        if (rb + 2 <= rend) load_rows(rb + 2);                  // the next pair of rows: in flight under this pair's MFMAs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = rb + h;
            if (r > rend) break;
            // both column tiles' fragments of a k-block first: the MFMAs of a slice product then rotate over four accumulators
            // (a dependent MFMA on the same accumulator would wait for its predecessor)
            f32x4 T0[2], T1[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) { T0[ct] = C0[ct]; T1[ct] = C1[ct]; }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                u32x4 b[2][3];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl)
                        b[ct][sl] = *reinterpret_cast<const u32x4*>(mp + h * RB + sl * PL + 16 * ct * PS + boff[kb]);
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    constexpr int SA[6] = {0, 2, 1, 0, 1, 0}, SB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        T0[ct] = mfma16(A[0][kb][SA[pr]], b[ct][SB[pr]], T0[ct]);
                        T1[ct] = mfma16(A[1][kb][SA[pr]], b[ct][SB[pr]], T1[ct]);
                    }
                }
            }
            // tile 1, rows 0 .. 7 (lanes 0 .. 31: channel 4 (lane / 16) + e, pixel L) = output row r - 1, complete (the bias
            // entered with the dy-0 partial)
            if (r > y0) {
                float* const orow = outp + (size_t)(r - 1) * a.W;
                float o0[4], o1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o0[e] = T1[0][e] > 0.f ? T1[0][e] : 0.1f * T1[0][e];
                    o1[e] = T1[1][e] > 0.f ? T1[1][e] : 0.1f * T1[1][e];
                }
                if (st0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 + e < COUT || kq == 0) orow[(size_t)e * HW] = o0[e];      // Cout 6: lanes 16 .. 31 hold channels 4, 5 only
                }
                if (st1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 + e < COUT || kq == 0) orow[(size_t)e * HW + 16] = o1[e];
                }
            }
            // the halves of tile 0 become the next pass's C operands: [bias | dy-0 partial], [dy-0 + dy-1 partial | 0]
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sw = other_half(T0[ct][e]);
                    C0[ct][e] = lane >= 32 ? sw : bq[e];
                    C1[ct][e] = lane < 32 ? sw : 0.f;
                }
        }
    }
}

template <int K>
int launch_gx(const float* mv, const float* res, float* feat, const float* pk, const void* frags, int N, int H, int W, hipStream_t s) {
    GenX3Args a;
    a.mv = mv; a.res = res; a.feat = feat;
    a.frag = static_cast<const u32x4*>(frags) + (size_t)gx_frag_off(K) * 64;
    a.bias = pk + bf_off(K);
    a.H = H; a.W = W;
    a.cstrips = (W + GX_SW - 1) / GX_SW;
    a.rstrips = (H + GX_RS - 1) / GX_RS;
    a.ntasks = N * a.cstrips * a.rstrips;
    a.nwg = (a.ntasks + GX_WAVES - 1) / GX_WAVES;
    a.wg_per_xcd = (a.nwg + 7) / 8;
    gen_x3_kernel<K><<<a.wg_per_xcd * 8, GX_WAVES * 64, 0, s>>>(a);
    return check_launch("gen_x3_layer");
}

}  // namespace

namespace dmc {

bool gen_x3_supported(int K, int H, int W) {
    return K >= 0 && K < GX_LAYERS && H > 0 && W > 0 && (long)H * W < (1l << 30);
}

int gen_x3_layer(int K, const float* mv, const float* res, float* feat, const float* pk, const void* frags, int N, int H, int W,
                 hipStream_t s) {
    switch (K) {
        case 0: return launch_gx<0>(mv, res, feat, pk, frags, N, H, W, s);
        case 1: return launch_gx<1>(mv, res, feat, pk, frags, N, H, W, s);
        case 2: return launch_gx<2>(mv, res, feat, pk, frags, N, H, W, s);
        default: return fail(DMC_E_INVALID, "gen_x3_layer: layer %d", K);
    }
}

}  // namespace dmc
