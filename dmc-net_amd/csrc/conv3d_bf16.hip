// NDHWC bf16 3-D convolutions of the I3D trunk on the gfx950 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate) -- BASELINE config 5, SURVEY 8(f)4.
//
// Replaces nn.Conv3d inside the reference's Unit3Dpy (code/dmcnet_I3D/network/i3d.py:328-403) for the
// trunk's stride-1 "SAME" convolutions: every 1x1x1 and 3x3x3 Unit3Dpy of conv3d_2b / 2c and of the nine
// Mixed blocks (:421-455), i.e. all of the trunk except the 2-channel 7x7x7 stride-2 stem.
//
//   forward        y[p][co]  = sum_{tap,ci} x[p + off(tap)][ci] * w[co][ci][tap]       (zero outside the volume)
//   data gradient  the same kernel on dy with the weights packed as wt[ci][mirror(tap)][co]
//   weight grad.   dw[co][ci][tap] = sum_p dy[p][co] * x[p + off(tap)][ci]             (conv3d_wgrad_kernel below)
//
// Activations are the memory of a channels_last_3d bf16 tensor, [N][D][H][W][C]; the fp32 master weights are packed
// per call into bf16 [rows_pad][taps][Cp] (rows = output channels of the GEMM padded to 128, Cp = contraction
// channels padded to 32, zero filled -- so the weight tile needs no masks and odd channel counts (16, 24, 48, 112,
// 144, 208, 528 ... the Inception branch widths) only cost zero columns in their last 32-channel chunk).
//
// GEMM view: rows = output channels (A operand = weights), columns = output pixels (B operand = the input at the
// tap-shifted pixel), K = (tap, ci).  A workgroup owns BM pixels x BN channels; a K-step is one tap x 32 channels =
// two MFMA k-blocks.  Both operand tiles are rows of 64 bytes (32 bf16) and go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: 64 lanes x 16 bytes = sixteen rows per instruction), double-buffered; the 16-byte quad q
// of tile row r is stored in slot q ^ ((r >> 2) & 3) -- applied to the source address, and again by the fragment
// reads -- so that a ds_read_b128 of 16 consecutive rows is conflict-free.  Per pixel row a lane keeps one address
// and a 27-bit mask of the taps that fall inside the volume; masked taps and channels >= Cin read 16 zero bytes.
// Epilogue: fp32 -> bf16 (round to nearest even), 8-byte stores of 4 consecutive channels, and, when asked, the
// per-channel (sum, sum of squares) of the ROUNDED values in fp32 per workgroup for the BatchNorm3d that follows.
#include "dmc_common.h"
#include <type_traits>

using namespace dmc;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

__device__ __attribute__((aligned(16))) unsigned g_zeros3d[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void dma16(const void* src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"((unsigned long long)src), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

__device__ __forceinline__ unsigned f2bf(float v) {        // round to nearest even, as torch's .to(bfloat16)
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }

// Sum over the 32 lanes of a half wave (the pixel columns of an accumulator tile) on the VALU's data-parallel primitives: four
// rotations within the rows of 16 and one broadcast of a row's lane 15 into the next row; lanes 16 .. 31 (48 .. 63) end up with
// the sum of lanes 0 .. 31 (32 .. 63).  (A butterfly of __shfl_xor is five ds_bpermute_b32 per value -- 160 LDS-crossbar round
// trips for the 2 x 16 statistics of a tile, each waited for: half of a 1 x 1 x 1 launch, 11-26 % of a 3 x 3 x 3 job.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float c3d_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float c3d_sum32(float v) {
    v += c3d_dpp<0x128, 0xf>(v);                           // row_ror:8
    v += c3d_dpp<0x124, 0xf>(v);                           // row_ror:4
    v += c3d_dpp<0x122, 0xf>(v);                           // row_ror:2
    v += c3d_dpp<0x121, 0xf>(v);                           // row_ror:1
    v += c3d_dpp<0x142, 0xa>(v);                           // row_bcast:15 into rows 1 and 3
    return v;
}

struct C3dArgs {
    const bf16_t* x;       // [N][D][H][W][Cin]
    const bf16_t* w;       // packed [rows_pad][T][Cp]
    bf16_t* y;             // [N][D][H][W][Cout]
    float* stat_part;      // [gridDim.x][Cout][2] or null
    int N, D, H, W, Cin, Cout, Cp;
    int KD, KH, KW;        // odd extents; tap (kz, ky, kx) reads the pixel at offset (kz - KD/2, ky - KH/2, kx - KW/2)
    long M;                // N * D * H * W
};

// NS = stages of the transfer ring.  A step is 32 input channels of one tap: 2 x TM x TN MFMAs per wave (2 for the 64 x 64
// tile), far less than a global -> LDS round trip; with two stages every step waited for its successor's transfer to land
// (~1 us each: 108 steps = 100 us for a 3x3x3 layer of 128 channels on a 9,408-pixel map, 83 such launches per I3D
// micro-step).  With NS stages NS - 1 steps are in flight and a step only waits for the OLDEST of them
// (s_waitcnt vmcnt(<transfers of the NS - 2 younger steps>)).
template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(WM * WN * 64) void conv3d_bf16_kernel(C3dArgs a) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && BM % 16 == 0 && BN % 16 == 0, "tile layout");
    constexpr int IP = BM / 16, IW = BN / 16;              // DMA instructions per step: pixel rows, weight rows
    constexpr int NDP = (IP + NW - 1) / NW, NDW = (IW + NW - 1) / NW;
    constexpr int PIXB = BM * 64, BUF = (BM + BN) * 64;
    static_assert(NS == 2 || (IP % NW == 0 && IW % NW == 0), "a deeper ring counts transfers per wave: every wave must issue the same number");
    static_assert(NS * BUF <= 64 * 1024, "static LDS");
    constexpr int PER_STEP = NDP + NDW;                     // transfers per wave and step
    __shared__ __attribute__((aligned(1024))) char lds[NS * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const long m0 = (long)blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;
    const unsigned lds0 = lds_addr_of(lds);
    const unsigned long long zeros = (unsigned long long)g_zeros3d;
    const int T = a.KD * a.KH * a.KW;

    // ---- transfers: instruction e moves tile rows 16 e .. 16 e + 15; lane -> (row 16 e + lane / 4, slot lane % 4),
    // source quad = slot ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3) ----
    const int rr = lane >> 2;
    const int q = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned long long p_addr[NDP];
    unsigned p_mask[NDP];
#pragma unroll
    for (int j = 0; j < NDP; ++j) {
        const int e = wave + NW * j;
        const long m = m0 + 16 * e + rr;
        p_mask[j] = 0; p_addr[j] = zeros;
        if (e < IP && m < a.M) {
            const long hw = (long)a.H * a.W;
            const long nd = m / hw;
            const int rem = (int)(m - nd * hw);
            const int hh = rem / a.W, ww = rem - hh * a.W;
            const int dd = (int)(nd % a.D);
            p_addr[j] = (unsigned long long)a.x + ((unsigned long long)m * a.Cin + 8 * q) * 2;
            int t = 0;
            for (int kz = 0; kz < a.KD; ++kz)
                for (int ky = 0; ky < a.KH; ++ky)
                    for (int kx = 0; kx < a.KW; ++kx, ++t) {
                        const int z = dd + kz - a.KD / 2, yv = hh + ky - a.KH / 2, xv = ww + kx - a.KW / 2;
                        if (z >= 0 && z < a.D && yv >= 0 && yv < a.H && xv >= 0 && xv < a.W) p_mask[j] |= 1u << t;
                    }
        }
    }
    unsigned long long w_addr[NDW];
#pragma unroll
    for (int j = 0; j < NDW; ++j) {
        const int e = wave + NW * j;
        w_addr[j] = (unsigned long long)a.w + ((unsigned long long)(co0 + 16 * e + rr) * T * a.Cp + 8 * q) * 2;
    }

    // step state of the transfers being issued: tap (kz, ky, kx) = index tap_n, channel chunk ci_n; all scalar
    int tap_n = 0, kz_n = 0, ky_n = 0, kx_n = 0, ci_n = 0;
    auto issue = [&](int buf) {
        const long toff = ((((long)(kz_n - a.KD / 2) * a.H + (ky_n - a.KH / 2)) * a.W + (kx_n - a.KW / 2)) * a.Cin + ci_n) * 2;
        const long woff = ((long)tap_n * a.Cp + ci_n) * 2;
        const bool cok = ci_n + 8 * q < a.Cin;
        const unsigned base = lds0 + buf * BUF;
#pragma unroll
        for (int j = 0; j < NDP; ++j) {
            if (IP % NW != 0 && wave + NW * j >= IP) continue;
            const bool ok = ((p_mask[j] >> tap_n) & 1) && cok;
            const unsigned long long src = ok ? p_addr[j] + (unsigned long long)toff : zeros;
            dma16(reinterpret_cast<const void*>(src), base + (wave + NW * j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < NDW; ++j) {
            if (IW % NW != 0 && wave + NW * j >= IW) continue;
            dma16(reinterpret_cast<const void*>(w_addr[j] + (unsigned long long)woff), base + PIXB + (wave + NW * j) * 1024);
        }
    };
    auto advance = [&]() {
        ci_n += 32;
        if (ci_n >= a.Cp) {
            ci_n = 0; ++tap_n;
            if (++kx_n == a.KW) { kx_n = 0; if (++ky_n == a.KH) { ky_n = 0; ++kz_n; } }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment reads: lane -> tile row l31, quad 2 kb + khalf, stored in slot quad ^ ((row >> 2) & 3)
    const int l31 = lane & 31, khalf = lane >> 5;
    const int prow0 = wm * (BM / WM), crow0 = wn * (BN / WN);
    int xoff[2], wfo[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int slot = ((2 * kb + khalf) ^ ((l31 >> 2) & 3)) << 4;
        xoff[kb] = (prow0 + l31) * 64 + slot;
        wfo[kb] = PIXB + (crow0 + l31) * 64 + slot;
    }

    const int T_steps = T * (a.Cp >> 5);
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < T_steps) { issue(p); advance(); }
#ifdef C3D_TIMING
    long long tm[4] = {0, 0, 0, 0}, tq = (long long)__builtin_amdgcn_s_memtime();
#define C3D_LAP(k) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); tm[k] += now_ - tq; tq = now_; }
#else
#define C3D_LAP(k)
#endif
    if (NS - 1 <= T_steps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_STEP * (NS - 2)) : "memory");   // step 0 has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    C3D_LAP(0)
    int buf = 0, nbuf = NS - 1;                              // ring slots of step t / of the step issued in iteration t
#pragma unroll 1
    for (int t = 0; t < T_steps; ++t) {
        const bool more = t + NS - 1 < T_steps;
        if (more) { issue(nbuf); advance(); }
        const char* base = lds + buf * BUF;
        u32x4 xf[2][TM], wf[2][TN];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[kb][i] = *reinterpret_cast<const u32x4*>(base + xoff[kb] + i * 32 * 64);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[kb][j] = *reinterpret_cast<const u32x4*>(base + wfo[kb] + j * 32 * 64);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[kb][j]),
                                                                        __builtin_bit_cast(bf16x8, xf[kb][i]), acc[i][j], 0, 0, 0);
        // step t + 1 must have landed: the NS - 2 steps younger than it may stay in flight while the ring is full
        C3D_LAP(1)
        if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PER_STEP * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        C3D_LAP(2)
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }

    // ---- epilogue: lane holds pixel column l31 of tile i, channels 8 gq + 4 khalf + e of tile j in acc[4 gq + e] ----
    float* red = reinterpret_cast<float*>(lds);               // [WM][BN][2]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float s1[16], s2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long m = m0 + prow0 + 32 * i + l31;
            const bool mok = m < a.M;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = co0 + crow0 + 32 * j + 8 * gq + 4 * khalf;
                if (co >= a.Cout || !mok) continue;
                unsigned h[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = f2bf(acc[i][j][4 * gq + e]);
                    const float r = bf2f(h[e]);
                    s1[4 * gq + e] += r; s2[4 * gq + e] += r * r;
                }
                uint2 pk = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(a.y + (unsigned long long)m * a.Cout + co) = pk;
            }
        }
        if (a.stat_part) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d1 = c3d_sum32(s1[e]), d2 = c3d_sum32(s2[e]);
                if (l31 == 31) {
                    const int c = crow0 + 32 * j + 8 * (e >> 2) + 4 * khalf + (e & 3);
                    red[(wm * BN + c) * 2 + 0] = d1;
                    red[(wm * BN + c) * 2 + 1] = d2;
                }
            }
        }
    }
    if (a.stat_part) {
        __syncthreads();
        for (int c = tid; c < BN; c += NW * 64)
            if (co0 + c < a.Cout) {
                float d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { d1 += red[(w * BN + c) * 2 + 0]; d2 += red[(w * BN + c) * 2 + 1]; }
                float* dst = a.stat_part + ((size_t)blockIdx.x * a.Cout + co0 + c) * 2;
                dst[0] = d1; dst[1] = d2;
            }
    }
#ifdef C3D_TIMING
    C3D_LAP(3)
    __syncthreads();
    if (tid == 0 && a.stat_part)
        for (int k = 0; k < 4; ++k) a.stat_part[(size_t)blockIdx.x * a.Cout * 2 + blockIdx.y * 8 + k] = (float)tm[k];
#endif
}

// ------------------------------------------------------------------------------------------
// Weight gradient: dw[co][ci][tap] = sum_p dy[p][co] * x[p + off(tap)][ci], a GEMM over pixels.  Both MFMA operands
// must hold 8 consecutive k (= pixels) of one channel per lane, while memory is pixel-major, so the tiles are
// transposed on their way into LDS: every thread loads the 8 channels (16 bytes) of one pixel and writes them with
// eight ds_write_b16 into [channel][pixel] rows of 32 pixels (64 bytes, the forward's slot swizzle), from which the
// fragments are read with ds_read_b128.  A workgroup owns CT x CT channels (CT = 32 TM: 64 or 128) of dw for NT taps --
// the 9 in-plane taps of one kz for the 3x3x3 layers (the dy tile is staged once per 9 x tiles), or the single tap of a
// 1x1x1 layer with 128 x 128 tiles -- and a contiguous run of pixels; split-K partials [split][Cout][T][Cin] are
// summed in fixed order by conv3d_wgrad_reduce_kernel (deterministic, no atomics), which also writes the
// parameter's [Cout][Cin][T] layout.  Out-of-volume taps and pixels beyond the run contribute zeros.
// ------------------------------------------------------------------------------------------
struct C3dWgradArgs {
    const bf16_t* x;       // [N][D][H][W][Cin]
    const bf16_t* dy;      // [N][D][H][W][Cout]
    float* part;           // [splits][Cout][T][Cin]
    int N, D, H, W, Cin, Cout;
    int KD, KH, KW;
    long M;
    long per_split;        // pixels per split (multiple of 32)
    int tiles_ci;          // ceil(Cin / CT)
};

template <int NT, int TM>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(C3dWgradArgs a) {
    constexpr int CT = 64 * TM;                            // channels per tile side (2 x 2 waves of TM x TM 32 x 32 tiles)
    constexpr int LPT = CT / 64;                           // 16-byte loads per thread and tile (32 pixels x CT channels)
    __shared__ __attribute__((aligned(1024))) char lds[3 * CT * 64];   // dyT [CT][32 px] | xT [CT][32 px] x 2 (double-buffered over the taps)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wc = wave >> 1;               // wave tile: rows (co) 32 TM wr.., columns (ci) 32 TM wc..
    const int tile = blockIdx.x;
    const int co0 = (tile / a.tiles_ci) * CT, ci0 = (tile % a.tiles_ci) * CT;
    const int kz = blockIdx.y;                             // NT == 9: the depth tap; NT == 1: 0
    const int T = a.KD * a.KH * a.KW;
    const long p_begin = (long)blockIdx.z * a.per_split;
    long p_end = p_begin + a.per_split;
    if (p_end > a.M) p_end = a.M;

    // staging map: load l of this thread covers pixel (tid >> 3) + 32 * 0 .. and channel octet (tid & 7) + 8 l
    const int sp = tid >> 3;                               // pixel within the 32-pixel step
    const int so = tid & 7;                                // channel octet
    auto lds_elem = [&](int row, int px) -> int {          // byte offset of (channel row, pixel) in a [CT][32] tile
        return row * 64 + ((((px >> 3) ^ ((row >> 2) & 3) ^ ((row >> 4) & 3)) & 3) << 4) + (px & 7) * 2;
    };
    auto put = [&](char* tileb, int oct, const u32x4& v) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned short h = (unsigned short)((j & 1) ? (v[j >> 1] >> 16) : (v[j >> 1] & 0xffffu));
            *reinterpret_cast<unsigned short*>(tileb + lds_elem(8 * oct + j, sp)) = h;
        }
    };

    f32x16 acc[NT][TM][TM];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

    const int l31 = lane & 31, khalf = lane >> 5;
    int foff[2][2][TM];                                    // [operand][k-block][tile] byte offset of this lane's fragment
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ra = wr * 32 * TM + 32 * i + l31, rb = wc * 32 * TM + 32 * i + l31;
            foff[0][kb][i] = ra * 64 + ((((2 * kb + khalf) ^ ((ra >> 2) & 3) ^ ((ra >> 4) & 3)) & 3) << 4);
            foff[1][kb][i] = CT * 64 + rb * 64 + ((((2 * kb + khalf) ^ ((rb >> 2) & 3) ^ ((rb >> 4) & 3)) & 3) << 4);
        }
    const long hw = (long)a.H * a.W;
    const int dzo = (NT == 9 ? kz - a.KD / 2 : 0);

#pragma unroll 1
    for (long p0 = p_begin; p0 < p_end; p0 += 32) {
        const long m = p0 + sp;
        const bool mok = m < p_end;
        int dd = 0, hh = 0, ww = 0;
        if (mok) {
            const long nd = m / hw;
            const int rem = (int)(m - nd * hw);
            hh = rem / a.W; ww = rem - hh * a.W; dd = (int)(nd % a.D);
        }
        // dy tile (shared by the NT taps)
        u32x4 gv[LPT];
#pragma unroll
        for (int l = 0; l < LPT; ++l) {
            const int c = co0 + 8 * (so + 8 * l);
            gv[l] = u32x4{0u, 0u, 0u, 0u};
            if (mok && c < a.Cout) gv[l] = *reinterpret_cast<const u32x4*>(a.dy + (unsigned long long)m * a.Cout + c);
        }
        // the x tile is double-buffered over the taps: tap t + 1 is loaded while tap t's MFMAs run and written to the
        // other buffer behind them -- one barrier per tap, no global-load latency in front of the MFMAs (the first
        // version loaded, waited, wrote and synchronised twice per tap: 19 barriers and 10 exposed loads per 18 MFMAs)
        u32x4 xv4[LPT];
        auto load_x = [&](int t) {
            const int dyo = (NT == 9 ? t / 3 - 1 : 0), dxo = (NT == 9 ? t % 3 - 1 : 0);
            const int z = dd + dzo, yv = hh + dyo, xv = ww + dxo;
            const bool ok = mok && z >= 0 && z < a.D && yv >= 0 && yv < a.H && xv >= 0 && xv < a.W;
            const long ms = m + (long)dzo * hw + dyo * a.W + dxo;
#pragma unroll
            for (int l = 0; l < LPT; ++l) {
                const int c = ci0 + 8 * (so + 8 * l);
                xv4[l] = u32x4{0u, 0u, 0u, 0u};
                if (ok && c < a.Cin) xv4[l] = *reinterpret_cast<const u32x4*>(a.x + (unsigned long long)ms * a.Cin + c);
            }
        };
        load_x(0);
        __syncthreads();                                   // previous step's fragment reads are done
#pragma unroll
        for (int l = 0; l < LPT; ++l) { put(lds, so + 8 * l, gv[l]); put(lds + CT * 64, so + 8 * l, xv4[l]); }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) load_x(t + 1);
            __syncthreads();                               // dyT and xT[t & 1] visible; xT[(t + 1) & 1] free
            const int xoff = (t & 1) * CT * 64;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u32x4 af[TM], bf[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(lds + foff[0][kb][i]);
#pragma unroll
                for (int j = 0; j < TM; ++j) bf[j] = *reinterpret_cast<const u32x4*>(lds + xoff + foff[1][kb][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),
                                                                               __builtin_bit_cast(bf16x8, bf[j]), acc[t][i][j], 0, 0, 0);
            }
            if (t + 1 < NT) {
#pragma unroll
                for (int l = 0; l < LPT; ++l) put(lds + CT * 64 + ((t + 1) & 1) * CT * 64, so + 8 * l, xv4[l]);
            }
        }
    }

    // partials: lane holds column l31 (ci) of tile j, rows 8 gq + 4 khalf + e (co) of tile i
    float* part = a.part + (size_t)blockIdx.z * a.Cout * T * a.Cin;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = (NT == 9 ? kz * 9 + t : 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int ci = ci0 + wc * 32 * TM + 32 * j + l31;
                if (ci >= a.Cin) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = co0 + wr * 32 * TM + 32 * i + 8 * (e >> 2) + 4 * khalf + (e & 3);
                    if (co < a.Cout) part[((size_t)co * T + tap) * a.Cin + ci] = acc[t][i][j][e];
                }
            }
    }
}

// dw[co][ci][t] (the parameter's contiguous layout) = sum over splits of part[s][co][t][ci], fixed order: a workgroup
// owns 64 consecutive elements (16 float4 columns); its 16 thread rows sum the splits s = row, row + 16, .. in
// parallel and are then added in row order.  (One thread per element walking all <= 256 splits serially was
// latency-bound: 15.8 us per launch on average, 56 launches per I3D micro-step.)
__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                  int splits, int Cout, int T, int Cin) {
    __shared__ float4 red[16][17];
    const long total = (long)Cout * T * Cin;                 // a multiple of 8 (Cin % 8 == 0)
    const int o = threadIdx.x & 15, row = threadIdx.x >> 4;
    for (long i0 = (long)blockIdx.x * 64; i0 < total; i0 += (long)gridDim.x * 64) {
        const long i = i0 + 4 * o;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total)
            for (int k = row; k < splits; k += 16) {
                const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * total + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[row][o] = s;
        __syncthreads();
        if (row == 0 && i < total) {
            float4 t = red[0][o];
#pragma unroll
            for (int k = 1; k < 16; ++k) { const float4 v = red[k][o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long ie = i + e;
                const int ci = (int)(ie % Cin);
                const int tt = (int)((ie / Cin) % T);
                const int co = (int)(ie / ((long)Cin * T));
                dw[((size_t)co * Cin + ci) * T + tt] = tv[e];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// 3x3x3 weight gradient over a RING of input rows (option conv3d_wgrad = 1, the default where the geometry fits: W in
// {56, 28, 14, 7}; the scheme of x3s_wgrad_kernel, conv_x3s.hip, on single bf16 NDHWC tensors).
//
// conv3d_wgrad_kernel above stages, per 32 pixels and tap, a transposed x tile (global loads, 16-bit LDS stores, a
// barrier): 9 dependent tap steps per 32 pixels -- 126 of them per workgroup on the 9,408-pixel maps of mixed_4*, which
// is where its time goes.  Here the pixel space is walked in PADDED rows (H + 2 per (n, d) plane); a step covers R
// rows: its dy tile [R x W pixels][64 co] and the R + 2 input rows around it, which live in a ring of input rows in LDS
// ([ring row][W + 2 pixels][64 ci], 128-byte pixel rows); each step transfers only the R rows that entered the window
// (LDS-DMA, hardware zero fill for padding rows / halo columns / planes outside the volume / channels beyond C), and all
// nine in-plane taps read their operands from the ring at shifted addresses through ds_read_b64_tr_b16 (the transposition
// a GEMM over PIXELS needs from channel-contiguous tensors).  One barrier per step (56..63 pixels x 9 taps).
// blockIdx.z = the depth tap kz: the ring holds rows of plane d + kz - 1.  12 waves = 4 quarters (32 co x 32 ci) x 3 tap
// rows ky, three accumulators (kx) per wave.  The 64-byte halves of a pixel row are swapped when (pixel >> 1) & 1: the four
// pixels a 16-lane group transposes then fall into different banks.  Same partial layout and reduction as above.
// ------------------------------------------------------------------------------------------
constexpr unsigned C3R_OOB = 0x80000000u;
typedef const __attribute__((address_space(3))) char* c3r_lds_cptr;
typedef short c3r_s16x4 __attribute__((ext_vector_type(4)));
typedef float c3r_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void c3r_dma16(const u32x4& srd, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(srd), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ u32x4 c3r_srd(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 srd;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    srd[2] = 0x7fffffffu;
    srd[3] = 0x00020000u;
    return srd;
}
__device__ __forceinline__ u32x4 c3r_tr2(c3r_lds_cptr p0, c3r_lds_cptr p1) {   // eight k-values (pixels) of this lane's channel
    const c3r_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3r_s16x4*)p0);
    const c3r_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) c3r_s16x4*)p1);
    typedef unsigned c3r_u32x2 __attribute__((ext_vector_type(2)));
    const c3r_u32x2 ua = __builtin_bit_cast(c3r_u32x2, a), ub = __builtin_bit_cast(c3r_u32x2, b);
    return u32x4{ua[0], ua[1], ub[0], ub[1]};
}

struct C3dRingArgs {
    const bf16_t* x;       // [N][D][H][W][Cin]
    const bf16_t* dy;      // [N][D][H][W][Cout]
    float* part;           // [splits][Cout][27][Cin]
    int N, D, H, Cin, Cout;
    int steps, per_group, tiles_ci;
};

template <int W_, int R>
struct C3rGeom {
    static constexpr int PW = W_ + 2;
    static constexpr int NGRP = R >= W_ + 2 ? 2 : 4, LA = NGRP / 2;
    static constexpr int NRING = NGRP * R;
    static constexpr int XPX = ((NRING * PW + 7) / 8) * 8;             // ring pixels (transfers move 8 pixel rows of 128 B)
    static constexpr int NPX = R * W_;                                 // pixel slots per step
    static constexpr int NKB = (NPX + 15) / 16;
    static constexpr int DYT = ((NKB * 16 + 7) / 8) * 8;
    static constexpr int XI = (R * PW + 7) / 8;                        // transfers per step: input rows / dy tile
    static constexpr int DI = DYT / 8;
    static constexpr int XBYTES = ((XPX * 128 + 255) / 256) * 256, DYBYTES = ((DYT * 128 + 255) / 256) * 256;
    static constexpr int LDS = XBYTES + 2 * DYBYTES;
};

// KT = 3: the 3x3x3 layers.  KT = 1: the 1x1x1 layers on the same machinery (centre tap only: the waves of tap rows 0 and 2
// only help with the transfers) -- a GEMM over pixels with 56..63-pixel steps and transposing LDS reads instead of
// conv3d_wgrad_kernel's 32-pixel steps with transposing 16-bit stores.
template <int W_, int R, int KT>
__global__ __launch_bounds__(768) void conv3d_wgrad_ring_kernel(C3dRingArgs a) {
    using G = C3rGeom<W_, R>;
    constexpr int PW = G::PW, NRING = G::NRING, NKB = G::NKB;
    __shared__ __attribute__((aligned(1024))) char lds_r[G::LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5, L = lane & 15, grp = (lane >> 4) & 1;
    const int quarter = wave & 3, trow = wave >> 2;
    const int wi = quarter & 1, wj = quarter >> 1;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const int dz = KT == 3 ? (int)blockIdx.z - 1 : 0;
    const unsigned lds0 = lds_addr_of(lds_r);
    const int HP = a.H + 2, NPL = a.N * a.D;
    const int s_begin = blockIdx.y * a.per_group;
    const int s_end = s_begin + a.per_group < a.steps ? s_begin + a.per_group : a.steps;
    const u32x4 srd_x = c3r_srd(a.x), srd_d = c3r_srd(a.dy);
    // transfer lane map: pixel lane >> 3 of the instruction's 8, 16-byte piece lane & 7 of its 128-byte row
    const int tpx = lane >> 3, tpc = lane & 7;
    const int ci0 = tci * 64, co0 = tco * 64;

    // byte offset of (padded row g, column x, logical piece) in x / dy, or OOB
    auto x_off = [&](int g, int x, int piece) -> unsigned {
        const int pl = g / HP, y = g - pl * HP - 1;
        const int d = pl % a.D;
        const int c = ci0 + piece * 8;
        return (g >= 0 && pl < NPL && y >= 0 && y < a.H && x >= 0 && x < W_ && d + dz >= 0 && d + dz < a.D && c < a.Cin)
                   ? (unsigned)(((((long)(pl + dz) * a.H + y) * W_ + x) * a.Cin + c) * 2) : C3R_OOB;
    };
    auto d_off = [&](int g, int x, int piece) -> unsigned {
        const int pl = g / HP, y = g - pl * HP - 1;
        const int c = co0 + piece * 8;
        return (g >= 0 && pl < NPL && y >= 0 && y < a.H && c < a.Cout) ? (unsigned)(((((long)pl * a.H + y) * W_ + x) * a.Cout + c) * 2) : C3R_OOB;
    };
    // input rows of group q (padded rows [R q, R q + R)) -> ring slots R (q mod NGRP) ..; instruction i by wave i % 12
    auto issue_x = [&](int q) {
        const int slot0 = (((q % G::NGRP) + G::NGRP) % G::NGRP) * R;
#pragma unroll
        for (int i = 0; i < G::XI; ++i) {
            if (i % 12 != wave % 12) continue;
            const int px = 8 * i + tpx;                                // pixel of the R x PW range
            const int lpx = slot0 * PW + px;                           // its LDS pixel
            const int piece = tpc ^ (((lpx >> 1) & 1) << 2);           // the row's 64-byte halves swapped on odd pixel pairs
            const int r = px / PW, col = px - r * PW;
            // lanes beyond the group's R x PW pixels stay out of the transfer: their LDS rows belong to the next group
            if (px < R * PW) c3r_dma16(srd_x, x_off(R * q + r, col - 1, piece), lds0 + (unsigned)(slot0 * PW + 8 * i) * 128u);
        }
    };
    auto issue_dy = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < G::DI; ++i) {
            if ((i + G::XI) % 12 != wave % 12) continue;
            const int p = 8 * i + tpx;
            const int piece = tpc ^ (((p >> 1) & 1) << 2);
            const int r = p / W_, x = p - r * W_;
            c3r_dma16(srd_d, p < G::NPX ? d_off(R * s + r, x, piece) : C3R_OOB, lds0 + G::XBYTES + (unsigned)buf * G::DYBYTES + (unsigned)(8 * i) * 128u);
        }
    };

    c3r_f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (s_begin < s_end) {
        // every ring row the first step can touch holds zeros or data (never uninitialised LDS: 0 x NaN pattern = NaN)
#pragma unroll
        for (int q = -1; q < G::LA; ++q) issue_x(s_begin + q);
        issue_dy(s_begin, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    c3r_lds_cptr const LB = (c3r_lds_cptr)lds_r;
    // Loop-invariant part of the fragment addresses (round 4; 14-17 VALU instructions per MFMA before, profiles/
    // r3_pmc_i3d_kernels.csv: two modulo operations and a swizzle per unit and tap inside the step loop).  Unit (kb, u) of this
    // lane is pixel slot P = 16 kb + 8 khalf + 4 u + L / 4 of EVERY step: row r, column x and, per tap column t, the byte offset
    // of LDS pixel x + t with its half-row swizzle are fixed; only the ring slot of row R s + r + trow - 1 depends on s, and
    // (R s + trow - 1) mod NRING is wave-uniform.  With an even row pitch the swizzle bit of pixel slot PW + x + t is that of
    // x + t, flipped when slot (PW / 2) is odd: one xor.
    constexpr bool EVEN = (PW % 2) == 0;
    int rr[NKB][2], dyoff[NKB][2], pre[NKB][2][3], xx[NKB][2];
    const int cd = 2 * wi + grp, cx = 2 * wj + grp;                  // 16-channel chunk of this lane's group: co / ci
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int P = 16 * kb + 8 * khalf + 4 * u + (L >> 2);
            dyoff[kb][u] = G::XBYTES + (L & 3) * 8 + P * 128 + ((cd ^ (((P >> 1) & 1) << 1)) << 5);
            if (P >= G::NPX) P = G::NPX - 1;                         // beyond the tile: dy is zero there, any valid input address
            rr[kb][u] = P / W_;
            xx[kb][u] = P - rr[kb][u] * W_;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int lp = xx[kb][u] + t;                        // LDS pixel of tap column t within its ring row
                pre[kb][u][t] = lp * 128 + ((cx ^ (((lp >> 1) & 1) << 1)) << 5) + (L & 3) * 8;
            }
        }
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        issue_x(s + G::LA);                                          // the group entering the window
        if (s + 1 < s_end) issue_dy(s + 1, buf ^ 1);
        const int sb = (((R * s + trow - 1) % NRING) + NRING) % NRING;   // ring row of step row 0 for this wave's tap row (wave-uniform)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            int dyo[2], xo[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dyo[u] = dyoff[kb][u] + buf * G::DYBYTES;
                int slot = sb + rr[kb][u];
                slot = slot >= NRING ? slot - NRING : slot;
                if (EVEN) {
                    const int base = slot * (PW * 128), flip = ((slot * (PW / 2)) & 1) << 6;
#pragma unroll
                    for (int t = 0; t < 3; ++t) xo[u][t] = base + (pre[kb][u][t] ^ flip);
                } else {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int lp = slot * PW + xx[kb][u] + t;    // LDS pixel of tap column t
                        xo[u][t] = lp * 128 + ((cx ^ (((lp >> 1) & 1) << 1)) << 5) + (L & 3) * 8;
                    }
                }
            }
            if (KT == 3 || trow == 1) {                              // (wave-uniform)
                const u32x4 A = c3r_tr2(LB + dyo[0], LB + dyo[1]);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    if (KT == 1 && t != 1) continue;
                    const u32x4 B = c3r_tr2(LB + xo[0][t], LB + xo[1][t]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[t], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    constexpr int T = KT == 3 ? 27 : 1;
    float* out = a.part + (size_t)blockIdx.y * a.Cout * T * a.Cin;
    const int ci = ci0 + 32 * wj + l31;
    if (ci < a.Cin && (KT == 3 || trow == 1))
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (KT == 1 && t != 1) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + 32 * wi + 8 * (e >> 2) + 4 * khalf + (e & 3);
                const int tap = KT == 3 ? (int)blockIdx.z * 9 + 3 * trow + t : 0;
                if (co < a.Cout) out[((size_t)co * T + tap) * a.Cin + ci] = acc[t][e];
            }
        }
}

struct C3rPlan { int R, steps, tiles, groups, per_group; };
bool c3r_plan(int N, int D, int H, int W, int Cin, int Cout, C3rPlan& p, int nz = 3) {
    p.R = W == 56 ? 1 : W == 28 ? 2 : W == 14 ? 4 : W == 7 ? 9 : 0;
    if (!p.R || (W == 7 && H != 7) || (long)N * D * H * W * (Cin > Cout ? Cin : Cout) * 2 >= 0x7fffffffL) return false;
    const long rows = (long)N * D * (H + 2);
    p.steps = (int)((rows + p.R - 1) / p.R);
    p.tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
    int groups = 512 / (p.tiles * nz);
    if (groups < 1) groups = 1;
    const int max_groups = p.steps / 4 > 0 ? p.steps / 4 : 1;          // at least four steps per workgroup
    if (groups > max_groups) groups = max_groups;
    p.per_group = (p.steps + groups - 1) / groups;
    p.groups = (p.steps + p.per_group - 1) / p.per_group;
    return true;
}

template <int W_, int R, int KT>
int launch_c3r(const C3rPlan& p, C3dRingArgs a, hipStream_t s) {
    static_assert(C3rGeom<W_, R>::LDS <= 64 * 1024, "ring + dy tiles within the static LDS limit");
    conv3d_wgrad_ring_kernel<W_, R, KT><<<dim3(p.tiles, p.groups, KT), 768, 0, s>>>(a);
    return check_launch("conv3d_wgrad_ring");
}

struct C3dWgradPlan { int nt, ct, tiles_co, tiles_ci, groups, splits; long per_split; };
C3dWgradPlan c3d_wgrad_plan(long M, int Cin, int Cout, int KD, int KH, int KW) {
    C3dWgradPlan p;
    const bool k3 = KD == 3 && KH == 3 && KW == 3;
    p.nt = k3 ? 9 : 1;
    p.ct = k3 ? 64 : 128;
    if (!k3 && (Cin <= 64 || Cout <= 64)) p.ct = 64;
    p.tiles_co = (Cout + p.ct - 1) / p.ct; p.tiles_ci = (Cin + p.ct - 1) / p.ct;
    p.groups = k3 ? 3 : 1;
    const long base = (long)p.tiles_co * p.tiles_ci * p.groups;
    long splits = (512 + base - 1) / base;                  // ~2 rounds of one workgroup per CU (144 accumulator registers: one wave per SIMD); more splits only
                                                            // grow the partials (conv3d_2c: 171 splits = 227 MB written and read back)
    const long max_splits = (M + 511) / 512;                // at least 512 pixels per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.per_split = ((M + splits - 1) / splits + 31) / 32 * 32;
    p.splits = (int)((M + p.per_split - 1) / p.per_split);
    return p;
}

// index of (row r, tap t, contraction channel c) in the MFMA-fragment-order copy of a 27-tap weight array:
// [r / 32][c / 32][t][k-block (c / 16) % 2][lane = 32 ((c / 8) % 2) + r % 32][c % 8]  (same element count as [rows_pad][27][Cp])
__device__ __forceinline__ long frag_index(int r, int t, int c, int Cp) {
    const int nch = Cp >> 5;
    return ((((long)(r >> 5) * nch + (c >> 5)) * 27 + t) * 2 + ((c >> 4) & 1)) * 512 + (((c >> 3) & 1) * 32 + (r & 31)) * 8 + (c & 7);
}

// fp32 weights (any strides) -> packed bf16 [rows_pad][T][Cp], zero padded; tap index mirrored for the data gradient
__global__ __launch_bounds__(256) void conv3d_pack_w_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp, int rows,
                                                            int rows_pad, int cols, int Cp, int T, long s_row, long s_col,
                                                            long s_tap, int mirror) {
    const long total = (long)rows_pad * T * Cp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const int t = (int)((i / Cp) % T);
        const int r = (int)(i / ((long)Cp * T));
        unsigned v = 0;
        if (r < rows && c < cols) v = f2bf(w[r * s_row + c * s_col + (mirror ? T - 1 - t : t) * s_tap]);
        wp[i] = (bf16_t)v;
        if (T == 27) wp[total + frag_index(r, t, c, Cp)] = (bf16_t)v;       // the patch-resident kernel's fragment-order copy
    }
}

// both layouts in one launch: blockIdx.y = 0 the forward's [Cout_pad][T][Cin_p], 1 the data gradient's mirrored [Cin_pad][T][Cout_p]
__global__ __launch_bounds__(256) void conv3d_pack_w2_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wb,
                                                             int Cout, int Cin, int T, long s_co, long s_ci, long s_tap) {
    const bool bwd = blockIdx.y == 1;
    const int rows = bwd ? Cin : Cout, cols = bwd ? Cout : Cin;
    const int rows_pad = (rows + 127) / 128 * 128, Cp = (cols + 31) / 32 * 32;
    const long s_row = bwd ? s_ci : s_co, s_col = bwd ? s_co : s_ci;
    bf16_t* wp = bwd ? wb : wf;
    const long total = (long)rows_pad * T * Cp;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const int t = (int)((i / Cp) % T);
        const int r = (int)(i / ((long)Cp * T));
        unsigned v = 0;
        if (r < rows && c < cols) v = f2bf(w[r * s_row + c * s_col + (bwd ? T - 1 - t : t) * s_tap]);
        wp[i] = (bf16_t)v;
        if (T == 27) wp[total + frag_index(r, t, c, Cp)] = (bf16_t)v;       // the patch-resident kernel's fragment-order copy
    }
}

// ------------------------------------------------------------------------------------------
// 3 x 3 x 3, stride 1: PATCH-RESIDENT form (round 6).  The tap-stepping kernel above re-fetches the pixel rows of every
// tap from L2 (27 transfers of the same data) and meets a barrier every 2 - 8 MFMAs per wave.  Here a workgroup owns
// P = 32 TM consecutive positions of ONE (n, d) plane in PADDED flat coordinates f = hp (W + 2) + wp -- so that every
// in-plane tap is the constant offset (ky - 1) (W + 2) + (kx - 1) -- and 128 output channels, 32 per wave: every wave
// multiplies ALL P positions (TM accumulator tiles: a weight fragment is loaded once per TM MFMAs).  Per 32-channel chunk
// of the contraction the workgroup loads the patch ONCE: the rows [t P, t P + P + 2 (W + 2) + 2) of the three depth planes
// d - 1, d, d + 1 (LDS-DMA, 16 bytes per lane; rows outside the volume are never written and stay zero from a one-off
// clear), then runs all 27 taps x 2 k-blocks on it: 54 TM MFMAs per wave between two barriers.
// LDS rows have a pitch of 80 bytes (64 of data + 16 of padding): 20 banks per row, so the sixteen lanes of a
// ds_read_b128 group -- rows b + {0..3, 12..15, 20..27}, every residue mod 16 once -- hit disjoint banks for ANY base
// row b: no swizzle, the address of a tap is one scalar added to a per-lane constant.  The weight fragments come
// straight from global memory (L2-resident), from a copy of the packed weights in MFMA-fragment order
// [32-row tile][chunk][tap][k-block][lane][8] -- one contiguous KB per wave-load -- prefetched one tap ahead in registers.
// The patch is single-buffered: the workgroups resident on a CU (up to three at W = 28: 48 KB of LDS each) cover each
// other's load phases.  Pad positions (wp = 0, W + 1) are computed and dropped: W / (W + 2) of the MFMAs are useful.
// Workgroups that share a patch (the channel blocks of one tile) are mapped to the same XCD, one after the other.
struct P3Args {
    const bf16_t* x; const bf16_t* wfrag; bf16_t* y; float* stat_part;
    int N, D, H, W, Cin, Cout, Cp;
    int Wp, R, tiles_pp, nx, ny, ni;         // ni: DMA instructions per chunk (12 patch rows each)
    unsigned magic_R, magic_Wp;              // floor(2^32 / d) + 1: n / d = umulhi(n, magic) for the small n used here
    int ablate;                              // -DDMC_MEASURE build only (option conv_ablate): 16 = patch transfers of chunk 0 only, 32 = one weight fragment for every tap,
                                             // 64 = one position fragment per tap, 128 = s_memtime clocks per phase written over stat_part
};

constexpr int P3_PITCH = 80;
constexpr int P3_MAXK = 22;                  // DMA instructions per depth plane and chunk (12 patch rows each): R <= 264
constexpr int P3_LOADERS = 4;
constexpr int P3_THREADS = 256 + 64 * P3_LOADERS;   // waves 0 .. 3: the consumers; the rest: loaders (transfer instruction k of a plane -> loader k % P3_LOADERS)

typedef __bf16 p3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float p3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned p3_cvt2(float lo, float hi) {     // two fp32 -> packed bf16, round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(p3_f32x2{lo, hi}, p3_bf16x2));
}

__device__ __forceinline__ void p3_dma16(unsigned long long src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

// one consumer wave's share of a job: TMW position tiles (32 positions each) starting at tile i0 of the workgroup's P = 32 TM
// positions, the 32 output channels from co0; `job` counts chunk jobs (buffer parity) and is advanced by nch
template <int TM, int TMW>
__device__ __forceinline__ void p3_consume(const P3Args& a, const char* p3_lds, int bufsz, int nch, int bx, int co0, int i0, bool active,
                                           int& job, long long (&tim)[3]) {
    [[maybe_unused]] long long tq = DMC_ABL(a.ablate & 128) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    auto lap = [&](int slot) {                                         // measurement build, conv_ablate bit 7: clocks per phase
        if (DMC_ABL(a.ablate & 128)) {
            const long long now = (long long)__builtin_amdgcn_s_memtime();
            tim[slot] += now - tq;
            tq = now;
        }
    };
    constexpr int P = 32 * TM;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int lb = (32 * i0 + l31) * P3_PITCH + khalf * 16;            // B fragment: row of this lane's position, quad khalf (+ 2 kb)
    const int plane = bx / a.tiles_pp, t = bx - plane * a.tiles_pp;      // plane = n * D + d
    // A fragments of this wave's channel tile: [chunk][tap][kb][lane][8]
    const bf16_t* wl = a.wfrag + ((size_t)(active ? co0 >> 5 : 0) * nch * 27 * 2 * 64 + lane) * 8;

    f32x16 acc[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    for (int c = 0; c < nch; ++c, ++job) {
        if (active) {
            // Software pipeline, pinned with scheduling barriers (left alone, the scheduler sinks every load to just in
            // front of its first use and each MFMA waits out a full LDS / L2 latency): while the MFMAs of tap t run, the
            // position fragments of tap t + 1 (LDS) and the weight fragments of tap t + 2 (global) are in flight.
            const bf16_t* wc = wl + (size_t)c * 27 * 2 * 64 * 8;
            const char* pb = p3_lds + (job & 1) * bufsz + lb;
            // prefetch distances in taps: position fragments (LDS, ~130 - 300 clocks) PB ahead, weight fragments (L2, 500+ clocks
            // under load) PA ahead; one wave per SIMD has registers to spare (LDS bounds the occupancy, not the register file)
            constexpr int PB = 2, PA = 4;
            u32x4 af[PA + 1][2], bf[PB + 1][2][TMW];
            auto lda1 = [&](int tap, int kb) {
                af[tap % (PA + 1)][kb] = *reinterpret_cast<const u32x4*>(wc + ((size_t)tap * 2 + kb) * 64 * 8);
            };
            auto ldb1 = [&](int tap, int kb, int i) {
                const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
                bf[tap % (PB + 1)][kb][i] = *reinterpret_cast<const u32x4*>(pb + (kz * a.R + ky * a.Wp + kx) * P3_PITCH + i * 32 * P3_PITCH + kb * 32);
            };
#pragma unroll
            for (int tap = 0; tap < PA; ++tap)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    if (tap == 0 || !DMC_ABL(a.ablate & 32)) lda1(tap, kb);
#pragma unroll
            for (int tap = 0; tap < PB; ++tap)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < TMW; ++i)
                        if (tap == 0 || !DMC_ABL(a.ablate & 64)) ldb1(tap, kb, i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                // one load behind every MFMA (the wave issues in order: a block of loads in front of the MFMAs would leave
                // the matrix pipe idle while it issues): position fragment (kb, i) of tap + PB, and behind the first MFMA of
                // each k-block a weight fragment of tap + PA
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < TMW; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[DMC_ABL(a.ablate & 32) ? 0 : tap % (PA + 1)][kb]),
                            __builtin_bit_cast(bf16x8, bf[DMC_ABL(a.ablate & 64) ? 0 : tap % (PB + 1)][kb][i]), acc[i], 0, 0, 0);
                        if (tap + PB < 27 && !DMC_ABL(a.ablate & 64)) ldb1(tap + PB, kb, i);
                        if (i == 0 && tap + PA < 27 && !DMC_ABL(a.ablate & 32)) lda1(tap + PA, kb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
        }
        lap(0);                                                            // compute
        if (c + 1 < nch) __syncthreads();                                  // this chunk is consumed / the next one has landed
        lap(1);                                                            // waiting for the loaders / the other waves
    }

    // ---- epilogue (before the job's closing barrier: the loader is already fetching the next job's first chunk): lane
    // holds position 32 (i0 + i) + l31, channels 8 gq + 4 khalf + e of this wave's tile in acc[i][4 gq + e] ----
    if (active) {
        float s1[16], s2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const unsigned f = (unsigned)(t * P + a.Wp + 1 + 32 * (i0 + i) + l31);
            const unsigned hp = __umulhi(f, a.magic_Wp);
            const int wp = (int)(f - hp * a.Wp);
            const bool ok = hp >= 1u && (int)hp <= a.H && wp >= 1 && wp <= a.W;
            const size_t m = ((size_t)plane * a.H + (hp - 1)) * a.W + (wp - 1);
            // 16-byte stores: a lane holds channels 8 gq + 4 khalf .. + 3 of its position; lanes l and l + 32 (the two khalf)
            // swap halves so that each stores 8 consecutive channels for two of the four gq (eight dwordx4 stores per lane and
            // tile set instead of sixteen dwordx2: the store tail of a job is issue-bound)
            unsigned pk[4][2];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                // v_cvt_pk_bf16_f32 (round to nearest even, as f2bf): one instruction per two values
                pk[gq][0] = p3_cvt2(acc[i][4 * gq + 0], acc[i][4 * gq + 1]);
                pk[gq][1] = p3_cvt2(acc[i][4 * gq + 2], acc[i][4 * gq + 3]);
                if (a.stat_part && ok && co0 + 8 * gq + 4 * khalf < a.Cout) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = bf2f((e & 1) ? pk[gq][e >> 1] >> 16 : pk[gq][e >> 1] & 0xffffu);
                        s1[4 * gq + e] += r; s2[4 * gq + e] += r * r;
                    }
                }
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                // this lane stores gq = 2 pr + khalf and hands gq = 2 pr + 1 - khalf to its partner (selects, not indexed registers)
                const unsigned t0 = khalf ? pk[2 * pr][0] : pk[2 * pr + 1][0], t1 = khalf ? pk[2 * pr][1] : pk[2 * pr + 1][1];
                const unsigned m0 = khalf ? pk[2 * pr + 1][0] : pk[2 * pr][0], m1 = khalf ? pk[2 * pr + 1][1] : pk[2 * pr][1];
                const unsigned r0 = __shfl_xor(t0, 32, 64), r1 = __shfl_xor(t1, 32, 64);
                const int co = co0 + 8 * (2 * pr + khalf);
                if (co < a.Cout && ok) {
                    const u32x4 v = khalf ? u32x4{r0, r1, m0, m1} : u32x4{m0, m1, r0, r1};
                    *reinterpret_cast<u32x4*>(a.y + m * a.Cout + co) = v;
                }
            }
        }
        if (a.stat_part) {
            // TM rows of partials per position tile bx (one per 32 positions); this wave owns rows i0 .. i0 + TMW - 1 of its
            // channels: its sums go to the first, zeros to the rest -- every (row, channel) is written exactly once
            // whichever way the job was shared among the waves
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d1 = c3d_sum32(s1[e]), d2 = c3d_sum32(s2[e]);
                const int co = co0 + 8 * (e >> 2) + 4 * khalf + (e & 3);
                if (l31 == 31 && co < a.Cout) {
#pragma unroll
                    for (int q = 0; q < TMW; ++q) {
                        float* dst = a.stat_part + (((size_t)bx * TM + i0 + q) * a.Cout + co) * 2;
                        dst[0] = q == 0 ? d1 : 0.f; dst[1] = q == 0 ? d2 : 0.f;
                    }
                }
            }
        }
    }
    lap(2);                                                                // epilogue
    __syncthreads();                                                       // the job's last chunk is consumed / the next job's first has landed
    lap(1);
}

// PERSISTENT: gridDim.x workgroups (one per CU) walk the (position tile, channel block) jobs with stride gridDim.x; the
// loader streams the chunks of job after job, so the first chunk of the next job lands while the epilogue of this one runs.
template <int TM>
__global__ __launch_bounds__(P3_THREADS) void conv3d_p3_kernel(P3Args a) {
    constexpr int P = 32 * TM;
    extern __shared__ __attribute__((aligned(1024))) char p3_lds[];      // two patch buffers of ni x 960 bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const unsigned lds0 = lds_addr_of(p3_lds);
    const int bufsz = a.ni * 960;
    const int nch = a.Cp >> 5;
    const int total = ((a.nx + 7) / 8) * 8 * a.ny;
    const int npi = a.R / 12;

    if (loader) {
        // ---- the loader waves: one instruction moves 12 patch rows (60 lanes: row 12 k + lane / 5, quad lane % 5; quad 4 is
        // the row's padding and lanes 60 .. 63 idle) = 960 contiguous bytes of LDS.  Every depth plane of the patch starts at
        // an instruction boundary (R is a multiple of 12), so ONE plan of R / 12 offsets serves the three planes, which differ
        // by a scalar; the plan walks the rows incrementally (no division per row).  Rows outside the volume and channels
        // beyond Cin are fetched from a page of zeros: every slot a consumer reads is rewritten by every chunk. ----
        const int quad = lane % 5, rsub = lane / 5;
        const bool lane_on = lane < 60 && quad < 4;
        const unsigned long long zeros = (unsigned long long)g_zeros3d;
        const size_t plane_bytes = (size_t)a.H * a.W * a.Cin * 2;
        int job = 0;
        for (int id = blockIdx.x; id < total; id += gridDim.x) {
            const int slot = id >> 3;
            const int bx = (slot / a.ny) * 8 + (id & 7);
            if (bx >= a.nx) continue;
            const int plane = bx / a.tiles_pp, t = bx - plane * a.tiles_pp;
            const int n = plane / a.D, d = plane - n * a.D;
            unsigned off[P3_MAXK];
            {
                const unsigned f0 = (unsigned)(t * P + rsub);
                unsigned hp = __umulhi(f0, a.magic_Wp);
                int wp = (int)(f0 - hp * a.Wp);
#pragma unroll
                for (int k = 0; k < P3_MAXK; ++k) {
                    off[k] = 0xffffffffu;                                  // = the page of zeros
                    if (k < npi && hp >= 1u && (int)hp <= a.H && wp >= 1 && wp <= a.W)
                        off[k] = (unsigned)((((int)(hp - 1) * a.W + (wp - 1)) * a.Cin + quad * 8) * 2);
                    wp += 12;
                    if (wp >= a.Wp) { wp -= a.Wp; ++hp; }
                    if (wp >= a.Wp) { wp -= a.Wp; ++hp; }                  // (W + 2 >= 6)
                }
            }
            for (int c = 0; c < nch; ++c, ++job) {
                // chunk job -> buffer job & 1, while the other four waves multiply the previous one (its readers left this
                // buffer at the barrier that closed the job before that)
                if (lane_on && !(DMC_ABL(a.ablate & 16) && job > 0)) {
                    const bool cok = c * 32 + quad * 8 < a.Cin;
                    const unsigned cbytes = (unsigned)c * 64u;
                    for (int z = 0; z < 3; ++z) {
                        const int dz = d + z - 1;
                        const bool zok = cok && dz >= 0 && dz < a.D;
                        const unsigned long long xb = (unsigned long long)a.x + (size_t)(n * a.D + (zok ? dz : 0)) * plane_bytes + cbytes;
                        const unsigned dst = lds0 + (job & 1) * bufsz + z * npi * 960;
                        // (one wave issues a transfer every ~100 clocks: 63 of them per chunk at W = 56 took longer than the
                        // consumers' 216 MFMAs -- P3_LOADERS waves share them)
#pragma unroll
                        for (int k = 0; k < P3_MAXK; ++k)
                            if (k < npi && k % P3_LOADERS == wave - 4) p3_dma16(zok && off[k] != 0xffffffffu ? xb + off[k] : zeros, dst + k * 960);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                           // this chunk has landed / the previous one is consumed
            }
        }
        __syncthreads();                                                   // (the consumers' last chunk)
        return;
    }

    // ---- the four consumer waves.  A job has nt = 1 .. 4 tiles of 32 output channels; the waves share them so that none
    // idles: nt = 4 (or 3): one tile each, all P positions; nt = 2: two waves per tile, half of the positions each; nt = 1:
    // four waves on the one tile, a quarter of the positions each (a weight fragment then serves fewer MFMAs, but the
    // alternative is an idle SIMD for the whole job). ----
    int job = 0;
    long long tim[3] = {0, 0, 0};
    const long long t_begin = DMC_ABL(a.ablate & 128) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    __syncthreads();                                                       // the first chunk has landed
    for (int id = blockIdx.x; id < total; id += gridDim.x) {
        const int slot = id >> 3;
        const int bx = (slot / a.ny) * 8 + (id & 7), by = slot % a.ny;
        if (bx >= a.nx) continue;
        const int nt = min(4, (a.Cout - by * 128 + 31) / 32);
        if (nt >= 3) p3_consume<TM, TM>(a, p3_lds, bufsz, nch, bx, by * 128 + wave * 32, 0, wave < nt, job, tim);
        else if (nt == 2) p3_consume<TM, TM / 2>(a, p3_lds, bufsz, nch, bx, by * 128 + (wave & 1) * 32, (wave >> 1) * (TM / 2), true, job, tim);
        else if constexpr (TM >= 4) p3_consume<TM, TM / 4>(a, p3_lds, bufsz, nch, bx, by * 128, wave * (TM / 4), true, job, tim);
        else p3_consume<TM, TM / 2>(a, p3_lds, bufsz, nch, bx, by * 128, (wave & 1) * (TM / 2), wave < 2, job, tim);
    }
    if (DMC_ABL(a.ablate & 128) && a.stat_part != nullptr && lane == 0) {  // measurement build: clocks per phase of this wave, over the statistics
        float* d = a.stat_part + (blockIdx.x * 4 + wave) * 4;
        d[0] = (float)tim[0]; d[1] = (float)tim[1]; d[2] = (float)tim[2];
        d[3] = (float)((long long)__builtin_amdgcn_s_memtime() - t_begin);
    }
}

// geometry of the patch-resident kernel for a 3 x 3 x 3 layer with K contraction channels; tm = 0: not served
struct P3Plan { int tm, Wp, R, tiles_pp, nx, ny, ni, lds; };
P3Plan p3_plan(int N, int D, int H, int W, int K, int Cout, int KD, int KH, int KW) {
    P3Plan p{};
    const int cfg = option(OPT_CONV_CFG);
    if ((cfg >= 1 && cfg <= 6) || !(KD == 3 && KH == 3 && KW == 3) || K % 8 != 0) return p;   // conv_cfg 1 .. 5: a tap-stepping tile, 6: its automatic choice (A/B)
    if ((long)N * D * H * W * K * 2 >= 0x7fffffffL || (long)(H + 2) * (W + 2) + 130 >= (1L << 16)) return p;   // 32-bit offsets; exact magic division
    p.Wp = W + 2;
    p.ny = (Cout + 127) / 128;
    const int count = (H - 1) * p.Wp + W;                  // flat positions first real .. last real
    for (int tm = 4; tm >= 2; tm -= 2) {
        const int P = 32 * tm, R = (P + 2 * p.Wp + 2 + 11) / 12 * 12;   // rows per depth plane, a whole number of 12-row transfers
        const int ni = 3 * R / 12;
        if (R / 12 > P3_MAXK || p.Wp < 6) continue;
        const int tiles = (count + P - 1) / P;
        if (tm == 4 && cfg != 7 && (cfg == 8 || count <= 64)) continue;   // one 64-position tile covers a 7 x 7 plane; conv_cfg 7 / 8: always 128 / 64 (A/B)
        p.tm = tm; p.R = R; p.tiles_pp = tiles; p.nx = N * D * tiles; p.ni = ni; p.lds = 2 * ni * 960 + 64;
        return p;
    }
    return p;
}

int launch_p3(const C3dArgs& a, const P3Plan& p, hipStream_t s) {
    P3Args q;
    const int T = 27;
    q.x = a.x; q.y = a.y; q.stat_part = a.stat_part;
    q.wfrag = a.w + (size_t)((a.Cout + 127) / 128 * 128) * T * a.Cp;     // behind the [rows_pad][T][Cp] copy
    q.N = a.N; q.D = a.D; q.H = a.H; q.W = a.W; q.Cin = a.Cin; q.Cout = a.Cout; q.Cp = a.Cp;
    q.Wp = p.Wp; q.R = p.R; q.tiles_pp = p.tiles_pp; q.nx = p.nx; q.ny = p.ny; q.ni = p.ni;
    q.magic_R = (unsigned)(0x100000000ull / (unsigned)p.R) + 1u;
    q.magic_Wp = (unsigned)(0x100000000ull / (unsigned)p.Wp) + 1u;
    q.ablate = option(OPT_CONV_ABLATE);
    const unsigned total = (unsigned)(((p.nx + 7) / 8) * 8 * p.ny);
    const unsigned cus = (unsigned)persistent_cus(256);                   // persistent: one workgroup per CU (a multiple of 8: XCD mapping)
    const unsigned grid = total < cus ? total : cus;
    static LdsLimit lim4, lim2;
    hipError_t e = p.tm == 4 ? lim4.raise(reinterpret_cast<const void*>(&conv3d_p3_kernel<4>), 160 * 1024)
                             : lim2.raise(reinterpret_cast<const void*>(&conv3d_p3_kernel<2>), 160 * 1024);
    if (e != hipSuccess) return fail(DMC_E_LAUNCH, "conv3d_p3: dynamic LDS limit: %s", hipGetErrorString(e));
    if (p.tm == 4) conv3d_p3_kernel<4><<<grid, P3_THREADS, p.lds, s>>>(q);
    else conv3d_p3_kernel<2><<<grid, P3_THREADS, p.lds, s>>>(q);
    return check_launch("conv3d_p3");
}

template <int BM, int BN, int WM, int WN, int NS>
int launch_c3d(const C3dArgs& a, hipStream_t s) {
    dim3 grid((unsigned)((a.M + BM - 1) / BM), (a.Cout + BN - 1) / BN);
    conv3d_bf16_kernel<BM, BN, WM, WN, NS><<<grid, WM * WN * 64, 0, s>>>(a);
    return check_launch("conv3d_bf16");
}

// tile choice: 0 = 128 x 128 (2 x 2 waves, 64 x 64 per wave), 1 = 128 x 64 (2 x 2), 2 = 64 x 64 (2 x 2), 3 = 128 x 32 (4 x 1),
// 4 = 256 x 128 (4 x 2)
int c3d_choice(int cout, long M) {
    const int cfg = option(OPT_CONV_CFG);
    if (cfg >= 1 && cfg <= 5) return cfg - 1;
    if (cout <= 32) return 3;
    const long need = 512;
    if (cout > 64 && ((M + 255) / 256) * ((cout + 127) / 128) >= need && cout % 128 == 0) return 4;
    if (cout > 64 && (cout % 128 == 0 || cout % 128 > 64) && ((M + 127) / 128) * ((cout + 127) / 128) >= need) return 0;
    if (((M + 127) / 128) * ((cout + 63) / 64) >= need) return 1;
    return 2;
}
int c3d_block_pixels(int cout, long M) {
    const int c = c3d_choice(cout, M);
    return c == 4 ? 256 : c == 2 ? 64 : 128;
}

int launch_conv3d(const C3dArgs& a, hipStream_t s) {
    if (a.M <= 0) return DMC_OK;
    const P3Plan p3 = p3_plan(a.N, a.D, a.H, a.W, a.Cin, a.Cout, a.KD, a.KH, a.KW);
    if (p3.tm) return launch_p3(a, p3, s);
    switch (c3d_choice(a.Cout, a.M)) {
        // transfer-ring depth: 4 where every wave issues the same number of transfers and 4 stages fit the static LDS
        // (8 / 5 stages for the two small tiles: 20.11 vs 19.90 ms per I3D micro-step, three same-box pairs -- the LDS they take
        // costs resident workgroups)
        case 0: return launch_c3d<128, 128, 2, 2, 4>(a, s);
        case 1: return launch_c3d<128, 64, 2, 2, 4>(a, s);
        case 2: return launch_c3d<64, 64, 2, 2, 4>(a, s);
        case 3: return launch_c3d<128, 32, 4, 1, 2>(a, s);
        default: return launch_c3d<256, 128, 4, 2, 2>(a, s);
    }
}

bool c3d_supported(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return false;
    if (Cin % 8 != 0 || Cout % 8 != 0) return false;
    if (!((KD == 1 || KD == 3) && (KH == 1 || KH == 3) && (KW == 1 || KW == 3))) return false;
    return (long)N * D * H * W * (Cin > Cout ? Cin : Cout) < (1L << 40);
}

int pad_to(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" {

int dmc_conv3d_bf16_supported(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW) {
    return c3d_supported(N, D, H, W, Cin, Cout, KD, KH, KW) ? 1 : 0;
}

// bytes of the packed-weight workspace of one forward or data-gradient call
size_t dmc_conv3d_bf16_wpack_bytes(int Cin, int Cout, int KD, int KH, int KW) {
    const size_t T = (size_t)KD * KH * KW;
    const size_t f = (size_t)pad_to(Cout, 128) * T * pad_to(Cin, 32), b = (size_t)pad_to(Cin, 128) * T * pad_to(Cout, 32);
    return (T == 27 ? 4 : 2) * (f > b ? f : b) + 16;      // 3 x 3 x 3: a second copy in MFMA-fragment order behind the first
}

// pack the weights for the forward (wpack_f) and the data gradient (wpack_b) in one launch; each workspace has
// dmc_conv3d_bf16_wpack_bytes() bytes.  The fwd / dgrad entry points take w == NULL to use such a workspace as it is.
int dmc_conv3d_bf16_pack(const float* w, long w_s_co, long w_s_ci, long w_s_tap, void* wpack_f, void* wpack_b, int Cin, int Cout,
                         int KD, int KH, int KW, dmc_stream_t stream) {
    if (!w || !wpack_f || !wpack_b) return fail(DMC_E_INVALID, "dmc_conv3d_bf16_pack: null pointer");
    const int T = KD * KH * KW;
    const long tf = (long)pad_to(Cout, 128) * T * pad_to(Cin, 32), tb = (long)pad_to(Cin, 128) * T * pad_to(Cout, 32);
    const long total = tf > tb ? tf : tb;
    dim3 grid((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256), 2);
    if (DMC_ABL(option(OPT_CONV_ABLATE) & 256)) return DMC_OK;   // measurement build: what the per-layer pack launches cost a step
    conv3d_pack_w2_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w, (bf16_t*)wpack_f, (bf16_t*)wpack_b, Cout, Cin, T, w_s_co, w_s_ci, w_s_tap);
    return check_launch("conv3d_pack_w2");
}

// number of [Cout][2] float partial rows the forward writes when asked for statistics: the 1 x 1 x 1 / general kernels ...
int dmc_conv3d_bf16_stat_blocks(int N, int D, int H, int W, int Cout) {
    const long M = (long)N * D * H * W;
    const int bm = c3d_block_pixels(Cout, M);
    return (int)((M + bm - 1) / bm);
}
// ... and for a given layer (the 3 x 3 x 3 layers take the patch-resident kernel: one row per position tile)
int dmc_conv3d_bf16_stat_blocks_k(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW) {
    const P3Plan p3 = p3_plan(N, D, H, W, Cin, Cout, KD, KH, KW);
    return p3.tm ? p3.nx * p3.tm : dmc_conv3d_bf16_stat_blocks(N, D, H, W, Cout);
}

// y [N,D,H,W,Cout] bf16 = conv3d(x [N,D,H,W,Cin] bf16, w fp32 [Cout][Cin][KD][KH][KW] given by its element strides),
// stride 1, "SAME" zero padding (odd kernel extents).  wpack: dmc_conv3d_bf16_wpack_bytes().
int dmc_conv3d_bf16_fwd(const void* x, const float* w, long w_s_co, long w_s_ci, long w_s_tap, void* wpack, void* y,
                        float* stat_partials, int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW,
                        dmc_stream_t stream) {
    if (!x || !wpack || !y) return fail(DMC_E_INVALID, "dmc_conv3d_bf16_fwd: null pointer");
    if (!c3d_supported(N, D, H, W, Cin, Cout, KD, KH, KW))
        return fail(DMC_E_INVALID, "dmc_conv3d_bf16_fwd: unsupported shape N=%d D=%d H=%d W=%d Cin=%d Cout=%d k=%dx%dx%d",
                    N, D, H, W, Cin, Cout, KD, KH, KW);
    hipStream_t s = (hipStream_t)stream;
    const int T = KD * KH * KW, Cp = pad_to(Cin, 32), rows_pad = pad_to(Cout, 128);
    const long total = (long)rows_pad * T * Cp;
    int rc = DMC_OK;
    if (w) {                                               // w == NULL: wpack already holds the packed weights (dmc_conv3d_bf16_pack)
        conv3d_pack_w_kernel<<<(int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256), 256, 0, s>>>(
            w, (bf16_t*)wpack, Cout, rows_pad, Cin, Cp, T, w_s_co, w_s_ci, w_s_tap, 0);
        if ((rc = check_launch("conv3d_pack_w"))) return rc;
    }
    C3dArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)wpack; a.y = (bf16_t*)y; a.stat_part = stat_partials;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cp = Cp; a.KD = KD; a.KH = KH; a.KW = KW;
    a.M = (long)N * D * H * W;
    return launch_conv3d(a, s);
}

// dx [N,D,H,W,Cin] bf16 from dy [N,D,H,W,Cout] bf16
int dmc_conv3d_bf16_dgrad(const void* dy, const float* w, long w_s_co, long w_s_ci, long w_s_tap, void* wpack, void* dx,
                          int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW, dmc_stream_t stream) {
    if (!dy || !wpack || !dx) return fail(DMC_E_INVALID, "dmc_conv3d_bf16_dgrad: null pointer");
    if (!c3d_supported(N, D, H, W, Cin, Cout, KD, KH, KW)) return fail(DMC_E_INVALID, "dmc_conv3d_bf16_dgrad: unsupported shape");
    hipStream_t s = (hipStream_t)stream;
    const int T = KD * KH * KW, Cp = pad_to(Cout, 32), rows_pad = pad_to(Cin, 128);
    const long total = (long)rows_pad * T * Cp;
    int rc = DMC_OK;
    if (w) {
        conv3d_pack_w_kernel<<<(int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256), 256, 0, s>>>(
            w, (bf16_t*)wpack, Cin, rows_pad, Cout, Cp, T, w_s_ci, w_s_co, w_s_tap, 1);
        if ((rc = check_launch("conv3d_pack_w"))) return rc;
    }
    C3dArgs a;
    a.x = (const bf16_t*)dy; a.w = (const bf16_t*)wpack; a.y = (bf16_t*)dx; a.stat_part = nullptr;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cout; a.Cout = Cin; a.Cp = Cp; a.KD = KD; a.KH = KH; a.KW = KW;
    a.M = (long)N * D * H * W;
    return launch_conv3d(a, s);
}

// bytes of the split-K partials of the weight gradient
size_t dmc_conv3d_bf16_wgrad_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW) {
    const C3dWgradPlan p = c3d_wgrad_plan((long)N * D * H * W, Cin, Cout, KD, KH, KW);
    size_t splits = (size_t)p.splits;
    C3rPlan r;
    const bool k3 = KD == 3 && KH == 3 && KW == 3, k1 = KD == 1 && KH == 1 && KW == 1;
    if ((k3 || k1) && c3r_plan(N, D, H, W, Cin, Cout, r, k3 ? 3 : 1) && (size_t)r.groups > splits) splits = (size_t)r.groups;
    return splits * Cout * KD * KH * KW * Cin * sizeof(float) + 16;
}

// dw fp32 [Cout][Cin][KD][KH][KW] (contiguous, the parameter's layout) from x, dy bf16 NDHWC; deterministic
int dmc_conv3d_bf16_wgrad(const void* x, const void* dy, float* dw, float* workspace, int N, int D, int H, int W, int Cin,
                          int Cout, int KD, int KH, int KW, dmc_stream_t stream) {
    if (!x || !dy || !dw || !workspace) return fail(DMC_E_INVALID, "dmc_conv3d_bf16_wgrad: null pointer");
    if (!c3d_supported(N, D, H, W, Cin, Cout, KD, KH, KW) || !((KD == 3 && KH == 3 && KW == 3) || (KD == 1 && KH == 1 && KW == 1)))
        return fail(DMC_E_INVALID, "dmc_conv3d_bf16_wgrad: unsupported shape");
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)N * D * H * W;
    C3rPlan rp;
    const bool k3 = KD == 3 && KH == 3 && KW == 3;
    const int ring = option(OPT_CONV3D_WGRAD);               // 0: tap-stepping kernels; 1: row ring for 3x3x3; 2 (default): and for 1x1x1
    if (((k3 && ring >= 1) || (!k3 && ring >= 2)) && c3r_plan(N, D, H, W, Cin, Cout, rp, k3 ? 3 : 1)) {
        C3dRingArgs ra;
        ra.x = (const bf16_t*)x; ra.dy = (const bf16_t*)dy; ra.part = workspace;
        ra.N = N; ra.D = D; ra.H = H; ra.Cin = Cin; ra.Cout = Cout;
        ra.steps = rp.steps; ra.per_group = rp.per_group; ra.tiles_ci = (Cin + 63) / 64;
        int rc;
        if (k3) rc = W == 56 ? launch_c3r<56, 1, 3>(rp, ra, s) : W == 28 ? launch_c3r<28, 2, 3>(rp, ra, s)
                   : W == 14 ? launch_c3r<14, 4, 3>(rp, ra, s) : launch_c3r<7, 9, 3>(rp, ra, s);
        else rc = W == 56 ? launch_c3r<56, 1, 1>(rp, ra, s) : W == 28 ? launch_c3r<28, 2, 1>(rp, ra, s)
                : W == 14 ? launch_c3r<14, 4, 1>(rp, ra, s) : launch_c3r<7, 9, 1>(rp, ra, s);
        if (rc) return rc;
        const int T = k3 ? 27 : 1;
        const long total = (long)Cout * T * Cin;
        if (DMC_ABL(option(OPT_CONV_ABLATE) & 512)) return DMC_OK;   // measurement build: the partial-sum launches
        conv3d_wgrad_reduce_kernel<<<(int)((total + 63) / 64 > 8192 ? 8192 : (total + 63) / 64), 256, 0, s>>>(
            workspace, dw, rp.groups, Cout, T, Cin);
        return check_launch("conv3d_wgrad_reduce");
    }
    const C3dWgradPlan p = c3d_wgrad_plan(M, Cin, Cout, KD, KH, KW);
    C3dWgradArgs a;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.part = workspace;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KD = KD; a.KH = KH; a.KW = KW;
    a.M = M; a.per_split = p.per_split; a.tiles_ci = p.tiles_ci;
    dim3 grid(p.tiles_co * p.tiles_ci, p.groups, p.splits);
    if (p.nt == 9) conv3d_wgrad_kernel<9, 1><<<grid, 256, 0, s>>>(a);
    else if (p.ct == 128) conv3d_wgrad_kernel<1, 2><<<grid, 256, 0, s>>>(a);
    else conv3d_wgrad_kernel<1, 1><<<grid, 256, 0, s>>>(a);
    int rc = check_launch("conv3d_wgrad");
    if (rc) return rc;
    const int T = KD * KH * KW;
    const long total = (long)Cout * T * Cin;
    if (DMC_ABL(option(OPT_CONV_ABLATE) & 512)) return DMC_OK;
    conv3d_wgrad_reduce_kernel<<<(int)((total + 63) / 64 > 8192 ? 8192 : (total + 63) / 64), 256, 0, s>>>(
        workspace, dw, p.splits, Cout, T, Cin);
    return check_launch("conv3d_wgrad_reduce");
}

}  // extern "C"
