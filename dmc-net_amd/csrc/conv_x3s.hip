// 3x3 / stride-1 NHWC convolutions in bf16x3 arithmetic on PRE-SPLIT operands (gfx950).
//
// Fourth generation of the classifier's convolution path (torchvision BasicBlock 3x3 convolutions behind
// code/dmcnet/model.py:305, run at :352; the same arithmetic as conv3_kernel in conv_nhwc.hip: an fp32 value is the
// exact sum of three bf16 slices, a product is formed from six of the nine slice products, fp32 accumulate).
// What changed: the PRODUCER of an activation (BatchNorm apply / pooling / the BatchNorm backward) writes the three
// slices, so the convolutions neither read fp32 fragments nor split them -- the main loop is ds_read_b128 + MFMA only.
//
// Activation slices ("x3s" tensor):  bf16 [3 slices][C / 16 chunks][M pixels][16 channels]   (6 bytes per value)
//   -- chunk-planar: one (slice, chunk) plane is a dense array of 32-byte pixel rows, so a run of consecutive pixels is
//   one contiguous run of memory whatever the channel count (LDS-DMA transfers of 32 pixels x 32 B = 1 KB, no partial
//   cache lines).
//
// Forward / data gradient = direct convolution from an LDS-resident PATCH:
//   a workgroup owns BM consecutive output pixels x 64 output channels.  Per 16-channel chunk it stages, ONCE, the input
//   patch that those pixels' 3x3 windows touch -- the image rows of the tile plus a halo row above and below, W + 2
//   columns, zero rows/columns where the window leaves the image (written by the hardware range check of
//   buffer_load ... lds: no zero source, no select) -- and all nine taps read their B fragments from it at shifted
//   addresses.  Against the per-tap tiles of conv3_kernel this is 5x less L2 -> LDS traffic and 3x fewer transfers per
//   MFMA.  The weights of a step (one tap row x 16 channels x 3 slices x 64 channels = 18 KB) are packed by
//   x3s_pack_w_kernel in exactly the LDS image the fragments are read from, so their transfers are linear copies.
//   Steps (chunk, tap row) are double-buffered: the transfers of step s + 1 (its weights and a third of the next chunk's
//   patch) are issued at the top of step s and waited for at its end -- one barrier per 36 MFMAs per wave.
//   LDS rows are 32 bytes (16 channels); the 16-byte half h of row r lives in half h ^ ((r >> 3) & 1), which makes a
//   ds_read_b128 of any 32 consecutive rows conflict-free at every tap shift (applied to the transfer's source address,
//   the LDS image stays lane-linear).
//   Data gradient = the same kernel on the dy slices with the weights packed transposed and the taps mirrored.
#include "dmc_common.h"
#include "x3s_common.h"
#include <type_traits>

using namespace dmc;
using namespace dmc::x3;

namespace {

// ---- producers of the slice tensors ------------------------------------------------------------------
// x [M][C] fp32 (channels_last memory) -> xs [3][C/16][M][16] bf16.  One thread = 8 channels of one pixel.
__global__ __launch_bounds__(256) void x3s_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ xs, long M, int C) {
    const int c8 = C >> 3;
    const long total = M * c8;
    const int nchunk = C >> 4;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const long m = o / c8;
        const int g = (int)(o - m * c8);                       // 8-channel group
        const float4 a = reinterpret_cast<const float4*>(x)[2 * o], b = reinterpret_cast<const float4*>(x)[2 * o + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        x3s_store8(xs, v, (size_t)M, nchunk, (size_t)m, g);
    }
}

// xs -> x (exact: s0 + s1 + s2 reproduces the fp32 value bit for bit); tests and fallbacks
__global__ __launch_bounds__(256) void x3s_merge_kernel(const unsigned short* __restrict__ xs, float* __restrict__ x, long M, int C) {
    const long total = M * C;
    const size_t plane = (size_t)M * 16;
    const int nchunk = C >> 4;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const long m = o / C;
        const int c = (int)(o - m * C);
        const size_t w = ((size_t)(c >> 4) * M + m) * 16 + (c & 15);
        const float a = __uint_as_float((unsigned)xs[w] << 16);
        const float b = __uint_as_float((unsigned)xs[(size_t)nchunk * plane + w] << 16);
        const float d = __uint_as_float((unsigned)xs[(size_t)2 * nchunk * plane + w] << 16);
        x[o] = (a + b) + d;
    }
}

// ---- weights: w [R][9][K'] fp32 OHWI -> the LDS image of every step ----------------------------------------
// GEMM rows r (output channels of the launch) x k (its input channels):
//   forward:        r = co, k = ci, element = w[co][tap][ci]
//   data gradient:  r = ci, k = co, element = w[co][8 - tap][ci]          (mirrored taps)
// packed [r / 64][k / 16][tap row 3][tap 3][slice 3][r % 64][16], the 8-channel half of a row swapped when
// ((r % 64) >> 3) & 1 (the LDS bank swizzle of the fragment reads).  blockIdx.y selects the direction.
__global__ __launch_bounds__(256) void x3s_pack_w_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                                         unsigned short* __restrict__ wt, int Cout, int Cin, int mirror) {
    const bool transposed = blockIdx.y == 1;                  // mirror = 0: the stride-2 data gradient's image (taps as they are)
    unsigned short* dst = transposed ? wt : wf;
    if (!dst) return;
    const int R = transposed ? Cin : Cout, K = transposed ? Cout : Cin;
    const int nchunk = K >> 4;
    const long total = (long)R * 9 * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        // destination-major enumeration: i = ((((rt * nchunk + ch) * 9 + tap) * 3 [slice below]) ...) -- enumerate (rt, ch, tap, rl, kk)
        long t = i;
        const int kk = (int)(t & 15); t >>= 4;
        const int rl = (int)(t & 63); t >>= 6;
        const int tap = (int)(t % 9); t /= 9;
        const int ch = (int)(t % nchunk);
        const int rt = (int)(t / nchunk);
        const int r = rt * 64 + rl, k = ch * 16 + kk;
        const float v = transposed ? w[((long)k * 9 + (mirror ? 8 - tap : tap)) * Cin + r] : w[((long)r * 9 + tap) * Cin + k];
        unsigned u0, u1, u2;
        split3(v, u0, u1, u2);
        const size_t base = ((((size_t)(rt * nchunk + ch) * 9 + tap) * 3) * 64 + rl) * 16 + (kk ^ (((rl >> 3) & 1) << 3));
        dst[base] = (unsigned short)(u0 >> 16);
        dst[base + 64 * 16] = (unsigned short)(u1 >> 16);
        dst[base + 2 * 64 * 16] = (unsigned short)(u2 >> 16);
    }
}

// ---- forward / data gradient ---------------------------------------------------------------------------
struct X3Args {
    const void* xs;        // activation slices [3][K/16][M][16]
    const void* wp;        // packed weights (this direction)
    float* y;              // [M][R] fp32
    const float* addend;   // [M][R] or null: added before the store
    double* stat_part;     // [gridDim.x][R][2] or null
    int N, H, W, K, R;     // K input channels, R output channels of this launch
    int M;                 // N * H * W
    unsigned plane_bytes;  // M * 32
    // data gradient only: BatchNorm-backward sums of the UPSTREAM unit (the one whose output gradient this launch writes):
    // per channel sum(d) and sum(d * xhat), d = dx [* relu'], from the tile in LDS -- that unit's bn_partial pass disappears
    const float* bnb_y = nullptr;        // its convolution output [M][R]
    const float* bnb_stats = nullptr;    // mean[R], invstd[R]
    const float* bnb_gamma = nullptr;
    const float* bnb_beta = nullptr;
    const unsigned char* bnb_mask = nullptr;   // ReLU sign bits (4 per float4) or null: recomputed
    int bnb_relu = 0;
    double* bnb_part = nullptr;          // [gridDim.x][R][2]
    int part_blocks = -1;  // host only: rows of stat_part / bnb_part the caller allocated (checked against the grid; -1 = no check)
    int ablate;            // measurement only (option conv_ablate): 16 = no wait for transfers, 32 = no statistics, 64 = no transfers,
                           // 128 = no output stores, 256 = no main loop (results wrong / missing)
};

// WM waves along pixels x WN along channels; a wave owns TM x TN tiles of 32 pixels x 32 channels.
// BN = 32 TN WN must be 64 (the packed weight image); PPMAX = patch pixels the LDS is laid out for.
// STAG: the transfers are issued by HALF of the waves at a time, one behind each pair of MFMAs: waves 0 .. NW/2-1 (one per
// SIMD) issue the patch transfers in the tap behind the step's barrier, waves NW/2 .. (their SIMD partners) the next step's
// weights one tap later -- a SIMD always has one wave whose MFMA stream is not interrupted by transfer issue.
// CLS = 4: the data gradient of a STRIDE-2 3x3 convolution (padding 1, even input size) in one launch.  The launch's
// pixels are the positions (a, b) of dy; input pixel (2a + py, 2b + px) of parity class (py, px) receives only the taps
// with ty = 1 (py = 0) or ty in {0, 2} (py = 1), likewise in x: 1 + 2 + 2 + 4 = 9 (tap, class) pairs, every tap feeds
// exactly one class's accumulators, reading dy at offset (+1 for t = 0, else 0) -- the same step structure, four accumulator
// sets, no multiply on structural zeros (the in-loop-split path issued four small launches per convolution).
template <int WM, int WN, int TM, int TN, int PPMAX, bool STAG = false, int CLS = 1>
__global__ __launch_bounds__(WM * WN * 64) void x3s_conv_kernel(X3Args a) {
    constexpr int NW = WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
    static_assert(BN == 64, "packed weight image is 64 rows wide");
    // patch transfers move 32 pixel rows: the staged pixels (<= PPMAX - 1) end on a transfer boundary or the zero row is the
    // first row behind the last full transfer (PPMAX = 32 k + 1: the tightest fit, configuration 4)
    static_assert(PPMAX % 32 == 0 || PPMAX % 32 == 1, "patch transfers move 32 pixel rows");
    constexpr int PSL = PPMAX * 32;                        // bytes of one slice of a patch buffer
    constexpr int PB = 3 * PSL;                            // one patch buffer
    constexpr int WB = 3 * 3 * BN * 32;                    // one weight buffer: tap row x 3 slices x 64 rows x 32 B
    constexpr int WOFF = 2 * PB;
    static_assert(!STAG || NW >= 4, "two transfer groups need four waves");
    constexpr int NA = STAG ? NW / 2 : NW, NB = NW - (STAG ? NA : 0);   // waves issuing patch / weight transfers
    constexpr int NDW = (PPMAX / 32 + NA - 1) / NA;        // patch transfers per issuing wave and slice
    constexpr int NEW = WB / 1024;                         // weight transfers per step (18)
    constexpr int NEWW = (NEW + NB - 1) / NB;
    extern __shared__ __attribute__((aligned(1024))) char lds_x3[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, khalf = lane >> 5;
    const unsigned lds0 = lds_addr_of(lds_x3);

    // ---- patch geometry (wave-uniform) ----
    // patch rows have NO halo columns (pitch W): the 32 pixels of an MFMA tile are then 32 consecutive 32-byte LDS rows, the
    // pattern the half-swap swizzle makes conflict-free (with W + 2 columns every image-row end shifted the later lanes by
    // two rows: 13 % of the LDS cycles were bank conflicts at W = 56, 37 % at W = 7).  A tap that leaves the image sideways
    // reads the patch's last row instead, which is kept zero.
    // The patch is a CONTIGUOUS run of real pixels: from the start of the image row above the tile's first pixel to the end of
    // the row below its last one (rows of consecutive images follow each other in memory).  No padding rows are staged: a
    // tap that leaves the image upwards / downwards reads the zero row too (the row it would otherwise hit belongs to the
    // neighbouring image and is simply not used) -- a 128-pixel tile of a 28-wide map needs 224 rows instead of 280, which
    // is what lets two workgroups share a CU there.
    const int HW = a.H * a.W, PW = a.W;
    constexpr int ZROW = PPMAX - 1;
    const int m0 = blockIdx.x * BM;
    const int mlast = (m0 + BM < a.M ? m0 + BM : a.M) - 1;
    const int r_first = m0 / PW, r_last = mlast / PW;      // global pixel rows (n * H + y) of the tile's first / last pixel
    const int f0 = (r_first > 0 ? r_first - 1 : 0) * PW;   // first staged pixel
    const int f1 = (r_last + 2) * PW < a.M ? (r_last + 2) * PW : a.M;   // one past the last staged pixel
    const int PP = f1 - f0;
    const int ND = (PP + 31) >> 5;                         // transfers per slice

    // ---- this lane's patch transfers: instruction d = wave + NW k moves patch rows 32 d .. 32 d + 31 ----
    unsigned pvoff[NDW];
#pragma unroll
    for (int k = 0; k < NDW; ++k) {
        const int pr = 32 * (wave + NA * k) + (lane >> 1), h = lane & 1;
        pvoff[k] = pr < PP ? (unsigned)(f0 + pr) * 32u + (unsigned)((h ^ ((pr >> 3) & 1)) << 4) : OOB;
    }
    // ---- fragment addresses: B operand (activations) per tile and tap; A operand (weights) per lane ----
    int xaddr[TM][9];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = m0 + (wm * TM + i) * 32 + l31;
        if (m > mlast) m = mlast;                          // rows beyond M: a valid address, result not stored
        const int n = m / HW, rem = m - n * HW, yy = rem / a.W, xx = rem - yy * a.W;
        const int pp = m - f0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dx = CLS == 4 ? (t % 3 == 0 ? 1 : 0) : t % 3 - 1, dy = CLS == 4 ? (t / 3 == 0 ? 1 : 0) : t / 3 - 1;
            const int row = (xx + dx >= 0 && xx + dx < a.W && yy + dy >= 0 && yy + dy < a.H) ? pp + dy * PW + dx : ZROW;
            xaddr[i][t] = row * 32 + ((khalf ^ ((row >> 3) & 1)) << 4);
        }
    }
    const int waddr = (wn * TN * 32 + l31) * 32 + ((khalf ^ ((l31 >> 3) & 1)) << 4);

    const u32x4 srd_x = make_srd(a.xs);
    const int nchunk = a.K >> 4, S = 3 * nchunk;
    const u32x4 srd_w = make_srd(reinterpret_cast<const char*>(a.wp) + (size_t)blockIdx.y * S * WB);
    const unsigned lane16 = (unsigned)lane * 16u;

    const bool grp_a = wave < NA, grp_b = !STAG || wave >= NA;
    const int wb = STAG ? wave - NA : wave;
    auto dma_weight = [&](int step, int wbuf, int k) {      // transfer k of this wave for step's weight image
        const int e = wb + NB * k;
        if (e < NEW) dma_buf16(srd_w, lane16, (unsigned)(step * WB + e * 1024), lds0 + WOFF + wbuf * WB + e * 1024);
    };
    auto dma_patch = [&](int chunk, int slice, int pbuf, int k) {   // transfer k of this wave for one slice of chunk's patch
        const int d = wave + NA * k;
        if (d < ND) dma_buf16(srd_x, pvoff[k], (unsigned)(slice * nchunk + chunk) * a.plane_bytes, lds0 + pbuf * PB + slice * PSL + d * 1024);
    };
    auto issue_weights = [&](int step, int wbuf) {
        if (grp_b) {
#pragma unroll
            for (int k = 0; k < NEWW; ++k) dma_weight(step, wbuf, k);
        }
    };
    auto issue_patch = [&](int chunk, int slice, int pbuf) {
        if (grp_a) {
#pragma unroll
            for (int k = 0; k < NDW; ++k) dma_patch(chunk, slice, pbuf, k);
        }
    };

    f32x16 acc[CLS][TM][TN];
#pragma unroll
    for (int cl = 0; cl < CLS; ++cl)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[cl][i][j][e] = 0.f;

    lds_cptr const L = (lds_cptr)lds_x3;
    struct Frag { u32x4 X[TM][3], W[TN][3]; };              // one tap's operands: 3 (TM + TN) reads, 6 TM TN MFMAs
    auto load_frags = [&](auto pbufc, auto wbufc, auto tc, Frag& f) {
        constexpr int pbuf = decltype(pbufc)::value, wbuf = decltype(wbufc)::value, t = decltype(tc)::value;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                f.W[j][s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(
                    L + waddr + (WOFF + wbuf * WB + (((t % 3) * 3 + s) * BN + 32 * j) * 32));
#pragma unroll
            for (int i = 0; i < TM; ++i)
                f.X[i][s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(L + xaddr[i][t] + (pbuf * PB + s * PSL));
        }
    };
    // slice products (weight, input): (0,2) (0,1) (0,0) (1,1) (1,0) (2,0); `between(p)` runs behind product p
    // (tc = the tap: its parity class selects the accumulator set when CLS = 4)
    auto mfma_frags = [&](const Frag& f, auto tc, auto&& between) {
        constexpr int t = decltype(tc)::value;
        constexpr int cl = CLS == 4 ? (t / 3 != 1) * 2 + (t % 3 != 1) : 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            constexpr int WS[6] = {0, 0, 0, 1, 1, 2}, XS[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[cl][i][j] = mfma_bf16(f.W[j][WS[p]], f.X[i][XS[p]], acc[cl][i][j]);
            between(p);
        }
    };
    auto nothing = [](int) {};
    constexpr int NRD = 3 * (TM + TN), NMF = 6 * TM * TN;
    auto interleave = [&]() {                                // one fragment read behind each of the first NRD MFMAs
#pragma unroll
        for (int k = 0; k < NRD; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
    };
    const bool dma_on = !(DMC_ABL(a.ablate) & 64);
    // One step = tap row q of chunk c (operands in patch buffer pbuf / weight buffer wbuf).  On entry FA holds the
    // fragments of its first tap.  The loop is software-pipelined over taps -- tap t + 1's fragments are read behind tap
    // t's MFMAs -- and the step's barrier stands in front of its LAST tap's MFMAs: by then every fragment of the step is
    // in registers (its buffers are dead: the transfers of step s + 2 go into them), and the next step's first fragments
    // are read behind that last tap's MFMAs, so nobody waits for LDS after the barrier.
    auto step = [&](auto pbufc, auto wbufc, auto qc, Frag& FA, Frag& FB, int c) {
        constexpr int pbuf = decltype(pbufc)::value, wbuf = decltype(wbufc)::value, q = decltype(qc)::value;
        using NP = std::integral_constant<int, q == 2 ? (pbuf ^ 1) : pbuf>;     // next step's buffers and first tap
        using NWB = std::integral_constant<int, wbuf ^ 1>;
        using NT = std::integral_constant<int, q == 2 ? 0 : 3 * (q + 1)>;
        using T0 = std::integral_constant<int, 3 * q>;
        using T1 = std::integral_constant<int, 3 * q + 1>;
        using T2 = std::integral_constant<int, 3 * q + 2>;
        const int s = 3 * c + q;
        load_frags(pbufc, wbufc, std::integral_constant<int, 3 * q + 1>{}, FB);
        if constexpr (STAG) {       // group B: the next step's weights (their buffer was released by the previous step's barrier)
            const bool go = grp_b && dma_on && s + 1 < S;
            mfma_frags(FA, T0{}, [&](int p) {
                if (go) {
#pragma unroll
                    for (int k = p; k < NEWW; k += 6) dma_weight(s + 1, wbuf ^ 1, k);
                }
            });
        } else {
            mfma_frags(FA, T0{}, nothing);
            interleave();
        }
        load_frags(pbufc, wbufc, std::integral_constant<int, 3 * q + 2>{}, FA);
        mfma_frags(FB, T1{}, nothing);
        interleave();
        if (DMC_ABL(a.ablate) & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's transfers for step s + 1 have landed
        __builtin_amdgcn_s_barrier();                                  // ... everyone's; step s's buffers are dead
        // the patch slice whose buffer the barrier released: q = 0, 1: slice q + 1 of chunk c + 1; q = 2: slice 0 of chunk c + 2
        const int vc = q == 2 ? c + 2 : c + 1;
        if constexpr (STAG) {
            if (s + 1 < S) load_frags(NP{}, NWB{}, NT{}, FB);
            const bool go = grp_a && dma_on && vc < nchunk;
            mfma_frags(FA, T2{}, [&](int p) {
                if (go) {
#pragma unroll
                    for (int k = p; k < NDW; k += 6) dma_patch(vc, q == 2 ? 0 : q + 1, q == 2 ? pbuf : (pbuf ^ 1), k);
                }
            });
        } else {
            if (dma_on) {
                if (s + 2 < S) issue_weights(s + 2, wbuf);
                if (vc < nchunk) issue_patch(vc, q == 2 ? 0 : q + 1, q == 2 ? pbuf : (pbuf ^ 1));
            }
            if (s + 1 < S) load_frags(NP{}, NWB{}, NT{}, FB);
            mfma_frags(FA, T2{}, nothing);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;

    if (tid < 12)                                          // the zero row of the 2 x 3 slice regions (never a transfer target that carries data)
        *reinterpret_cast<float4*>(lds_x3 + (tid >> 1) * PSL + ZROW * 32 + (tid & 1) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    issue_weights(0, 0);
    issue_patch(0, 0, 0); issue_patch(0, 1, 0); issue_patch(0, 2, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (dma_on) {
        if (!STAG && S > 1) issue_weights(1, 1);           // STAG: step 0's first tap issues them
        if (nchunk > 1) issue_patch(1, 0, 1);
    }
    Frag F0, F1;
    load_frags(I0{}, I0{}, Q0{}, F0);
    // chunk pairs (K % 32 == 0): patch buffers 0, 1; weight buffers alternate per step: 0 1 0 | 1 0 1
#pragma unroll 1
    for (int c = 0; c < ((DMC_ABL(a.ablate) & 256) ? 0 : nchunk); c += 2) {
        step(I0{}, I0{}, Q0{}, F0, F1, c);
        step(I0{}, I1{}, Q1{}, F1, F0, c);
        step(I0{}, I0{}, Q2{}, F0, F1, c);
        step(I1{}, I1{}, Q0{}, F1, F0, c + 1);
        step(I1{}, I0{}, Q1{}, F0, F1, c + 1);
        step(I1{}, I1{}, Q2{}, F1, F0, c + 1);
    }

    // ---- epilogue ----
    // lane holds pixel l31 of tile i, channels 32 j + 8 g + 4 khalf + e in acc[i][j][4 g + e].  The wave writes its tile
    // as fp32 [32 TM pixels][64 channels] into its own piece of the (dead) operand LDS, rows padded to 272 B, and reads it
    // back (i) row-major, 16 bytes per lane: the global stores are whole 256-byte pixel rows, 1 KB contiguous per
    // instruction (+ the optional addend, read the same way); (ii) column-wise, lane = channel: the BatchNorm sums of the
    // tile in fp64, fixed order.  (Statistics are those of the convolution result; the forward passes no addend.)
    static_assert(WN == 1 && TN == 2, "epilogue: a wave owns all 64 channels of its pixels");
    constexpr int EP = 272, ETILE = 32 * TM * EP;
    char* etile = lds_x3 + wave * ETILE;
    const int mw0 = m0 + wm * TM * 32;                            // first pixel of this wave
    const int rbase = blockIdx.y * BN;
#pragma unroll
    for (int cl = 0; cl < CLS; ++cl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bool ok = mw0 + 32 * i + l31 <= mlast;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = make_float4(acc[cl][i][j][4 * g], acc[cl][i][j][4 * g + 1], acc[cl][i][j][4 * g + 2], acc[cl][i][j][4 * g + 3]);
                    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(etile + (32 * i + l31) * EP + (32 * j + 8 * g + 4 * khalf) * 4) = v;
                }
        }
        if (!(DMC_ABL(a.ablate) & 128)) {
#pragma unroll
            for (int it = 0; it < 8 * TM; ++it) {
                const int prow = 4 * it + (lane >> 4);
                const int m = mw0 + prow;
                float4 v = *reinterpret_cast<const float4*>(etile + prow * EP + (lane & 15) * 16);
                if (m <= mlast) {
                    size_t opix = (size_t)m;
                    if (CLS == 4) {                               // position (n, ya, xb) of dy -> input pixel (2 ya + py, 2 xb + px)
                        const int n = m / HW, rem = m - n * HW, ya = rem / a.W, xb = rem - ya * a.W;
                        opix = ((size_t)n * (2 * a.H) + 2 * ya + (cl >> 1)) * (2 * a.W) + 2 * xb + (cl & 1);
                    }
                    const size_t o = opix * a.R + rbase + (lane & 15) * 4;
                    if (a.addend) {
                        const float4 av = *reinterpret_cast<const float4*>(a.addend + o);
                        v.x += av.x; v.y += av.y; v.z += av.z; v.w += av.w;
                    }
                    *reinterpret_cast<float4*>(a.y + o) = v;
                    if (CLS == 1 && a.bnb_part) {
                        // upstream BatchNorm sums: this lane's four channels of the pixel, read like the addend (coalesced,
                        // all iterations' loads in flight together): d = dx [* relu'] and xhat go back to LDS for the column sums
                        const int c4 = rbase + (lane & 15) * 4;
                        const float4 yv = *reinterpret_cast<const float4*>(a.bnb_y + o);
                        const float4 mean = *reinterpret_cast<const float4*>(a.bnb_stats + c4);
                        const float4 istd = *reinterpret_cast<const float4*>(a.bnb_stats + a.R + c4);
                        const float4 xh = make_float4((yv.x - mean.x) * istd.x, (yv.y - mean.y) * istd.y, (yv.z - mean.z) * istd.z, (yv.w - mean.w) * istd.w);
                        if (a.bnb_relu) {
                            unsigned mb;
                            if (a.bnb_mask) {
                                mb = a.bnb_mask[o >> 2];
                            } else {
                                const float4 g = *reinterpret_cast<const float4*>(a.bnb_gamma + c4), b = *reinterpret_cast<const float4*>(a.bnb_beta + c4);
                                mb = (fmaf(xh.x, g.x, b.x) > 0.f ? 1u : 0u) | (fmaf(xh.y, g.y, b.y) > 0.f ? 2u : 0u) |
                                     (fmaf(xh.z, g.z, b.z) > 0.f ? 4u : 0u) | (fmaf(xh.w, g.w, b.w) > 0.f ? 8u : 0u);
                            }
                            v.x = (mb & 1) ? v.x : 0.f; v.y = (mb & 2) ? v.y : 0.f; v.z = (mb & 4) ? v.z : 0.f; v.w = (mb & 8) ? v.w : 0.f;
                        }
                        *reinterpret_cast<float4*>(etile + prow * EP + (lane & 15) * 16) = v;
                        *reinterpret_cast<float4*>(etile + NW * ETILE + 8192 + prow * EP + (lane & 15) * 16) = xh;
                    }
                } else if (CLS == 1 && a.bnb_part) {                // pixels beyond M: no contribution
                    *reinterpret_cast<float4*>(etile + prow * EP + (lane & 15) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(etile + NW * ETILE + 8192 + prow * EP + (lane & 15) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    }
    if (CLS == 1 && a.bnb_part) {
        // lane = channel: column sums of d and d * xhat over the wave's pixels, fp64, fixed order
        double* red = reinterpret_cast<double*>(lds_x3 + NW * ETILE);   // [WM][64][2]
        const char* xtile = etile + NW * ETILE + 8192;                  // this wave's xhat tile (behind all d tiles and `red`)
        double d1 = 0.0, d2 = 0.0;
#pragma unroll 8
        for (int p = 0; p < 32 * TM; ++p) {
            const double d = (double)*reinterpret_cast<const float*>(etile + p * EP + lane * 4);
            const double xh = (double)*reinterpret_cast<const float*>(xtile + p * EP + lane * 4);
            d1 += d;
            d2 += d * xh;
        }
        red[(wm * 64 + lane) * 2 + 0] = d1;
        red[(wm * 64 + lane) * 2 + 1] = d2;
        __syncthreads();
        for (int cc = tid; cc < BN; cc += NW * 64) {
            double e1 = 0.0, e2 = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { e1 += red[(w * 64 + cc) * 2 + 0]; e2 += red[(w * 64 + cc) * 2 + 1]; }
            double* dst = a.bnb_part + ((size_t)blockIdx.x * a.R + rbase + cc) * 2;
            dst[0] = e1; dst[1] = e2;
        }
    }
    if (CLS == 1 && a.stat_part) {
        double* red = reinterpret_cast<double*>(lds_x3 + NW * ETILE);   // [WM][64][2]
        double d1 = 0.0, d2 = 0.0;
#pragma unroll 8
        for (int p = 0; p < 32 * TM; ++p) {
            const double v = (double)*reinterpret_cast<const float*>(etile + p * EP + lane * 4);
            d1 += v;
            d2 += v * v;
        }
        red[(wm * 64 + lane) * 2 + 0] = d1;
        red[(wm * 64 + lane) * 2 + 1] = d2;
        __syncthreads();
        for (int c = tid; c < BN; c += NW * 64) {
            double e1 = 0.0, e2 = 0.0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { e1 += red[(w * 64 + c) * 2 + 0]; e2 += red[(w * 64 + c) * 2 + 1]; }
            double* dst = a.stat_part + ((size_t)blockIdx.x * a.R + rbase + c) * 2;
            dst[0] = e1; dst[1] = e2;
        }
    }
}

// patch pixels the tiles of BM consecutive output pixels need at most: rows spanned + halo + the zero rows between images
int patch_pixels_max(int N, int H, int W, int BM) {
    const long M = (long)N * H * W;
    int worst = 0;
    // tile starts repeat with period lcm(BM, W) pixels; scanning W tiles (or all of them) is exact
    long tiles = (M + BM - 1) / BM;
    if (tiles > W) tiles = W;
    for (long t = 0; t < tiles; ++t) {
        const long m0 = t * BM, ml = (m0 + BM < M ? m0 + BM : M) - 1;
        const long r0 = m0 / W, r1 = ml / W;
        const long f0 = (r0 > 0 ? r0 - 1 : 0) * W, f1 = (r1 + 2) * W;       // (interior tiles: the clipped ones are smaller)
        if ((int)(f1 - f0) + 1 > worst) worst = (int)(f1 - f0) + 1;        // staged pixels + the zero row
    }
    return worst;
}

struct X3Cfg { int id, BM, PPMAX, threads; };
// configurations: 0 = 256 pixels, 8 waves (one tile of 32 x 64 each), 1 = 128 pixels, 4 waves, 2 = 64 pixels, 2 waves
// 4: 128 pixels, 4 waves, a patch of <= 224 pixels (the 28-, 14- and 7-wide maps): 80,064 B of LDS -- TWO workgroups per CU, so that
// one's prologue / epilogue (20 % of a tile, profiles/r3_x3s_ablation.txt) runs under the other's main loop
constexpr X3Cfg X3CFGS[] = {{0, 256, 608, 512}, {1, 128, 320, 256}, {2, 64, 192, 128}, {3, 256, 608, 256}, {4, 128, 225, 256}};   // 3 (measurement): 4 waves of 64 x 64

bool x3s_shape_ok(int N, int H, int W, int K, int R) {
    if (N <= 0 || H <= 0 || W <= 0 || K % 32 != 0 || R % 64 != 0 || K <= 0 || R <= 0) return false;
    const long M = (long)N * H * W;
    return M * (K > R ? K : R) * 6 < (1L << 31);             // every slice tensor stays below the descriptor's 2 GB
}

int x3s_choose(int N, int H, int W, int R) {
    const long M = (long)N * H * W;
    const int forced = option(OPT_CONV_CFG);                 // measurement switch: 101 + configuration; 106 = never configuration 4
    if (forced >= 101 && forced <= 105 && patch_pixels_max(N, H, W, X3CFGS[forced - 101].BM) <= X3CFGS[forced - 101].PPMAX) return forced - 101;
    // the largest tile whose workgroups still cover most of the 256 CUs (measured, 120 frames: 8-wave workgroups of 256
    // pixels win down to layer4's 184 workgroups -- 160-187 TFLOP/s against 123-133 with 64-pixel tiles); else the
    // smallest tile whose patch fits
    // two 4-wave workgroups per CU where the patch of a 128-pixel tile is small enough (14- and 7-wide maps) and there are at
    // least as many workgroups as the chip has slots for them: layer3 (736 workgroups) 0.160 -> 0.141 ms, -12 %; layer4's 368
    // workgroups lose 2 % against 184 8-wave ones
    if (forced != 106 && patch_pixels_max(N, H, W, X3CFGS[4].BM) <= X3CFGS[4].PPMAX && ((M + X3CFGS[4].BM - 1) / X3CFGS[4].BM) * (R / 64) >= 512)
        return 4;
    int fallback = -1;
    for (const X3Cfg& c : X3CFGS) {
        if (c.id > 2 || patch_pixels_max(N, H, W, c.BM) > c.PPMAX) continue;                // (3: by option only)
        const long wgs = ((M + c.BM - 1) / c.BM) * (R / 64);
        if (wgs >= 160) return c.id;
        fallback = c.id;
    }
    return fallback;
}

// Measured and not kept (profiles/r3_x3s_persistent_ab.txt): a persistent form -- one workgroup per CU walking tiles, the
// transfer pipeline running across tiles (no tile starts by waiting for HBM), transfers issued by half of the waves at a
// time.  Correct, but 3-6 % SLOWER on every layer (layer1 forward 0.150 vs 0.141 ms): two tile geometries and the
// cross-tile cases cost 108 spilled SGPRs and 250 wave-uniform branches in the step bodies, more than the hidden prologue.
template <int WM, int TM, int PPMAX>
int launch_x3s(const X3Args& a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM;
    constexpr size_t op_bytes = 2 * 3 * PPMAX * 32 + 2 * 3 * 3 * 64 * 32, ep_bytes = 2 * (size_t)BM * 272 + 8192;   // d tiles, sums, xhat tiles
    constexpr size_t lds_bytes = op_bytes > ep_bytes ? op_bytes : ep_bytes;
    dim3 grid((a.M + BM - 1) / BM, a.R / 64);
    // the partials are [gridDim.x][R][2]: sized by the caller with dmc_x3s_conv_stat_blocks(), while the tile configuration is
    // chosen again here (option conv_cfg): refuse a buffer of another height instead of writing past it
    if ((a.stat_part || a.bnb_part) && a.part_blocks >= 0 && (unsigned)a.part_blocks != grid.x)
        return fail(DMC_E_INVALID, "x3s_conv: statistics partials have %d rows but this launch writes %u (dmc_x3s_conv_stat_blocks was "
                                   "called under another conv_cfg option?)", a.part_blocks, grid.x);
    if constexpr (WM >= 4) {
        if (!(option(OPT_CONV_ABLATE) & 512)) {               // 512: measurement, every wave issues its share behind the barrier
            static LdsLimit lim_attr2;
    const hipError_t attr2 = lim_attr2.raise(reinterpret_cast<const void*>(&x3s_conv_kernel<WM, 1, TM, 2, PPMAX, true>), (int)lds_bytes);
            if (attr2 != hipSuccess) return fail(DMC_E_LAUNCH, "x3s_conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr2));
            x3s_conv_kernel<WM, 1, TM, 2, PPMAX, true><<<grid, WM * 64, lds_bytes, s>>>(a);
            return check_launch("x3s_conv");
        }
    }
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&x3s_conv_kernel<WM, 1, TM, 2, PPMAX>), (int)lds_bytes);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "x3s_conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    x3s_conv_kernel<WM, 1, TM, 2, PPMAX><<<grid, WM * 64, lds_bytes, s>>>(a);
    return check_launch("x3s_conv");
}

struct X3Bnb { const float *y, *stats, *gamma, *beta; const unsigned char* mask; int relu; double* part; };

int run_x3s(const void* xs, const void* wp, const float* addend, float* y, double* stat_part, int part_blocks, int N, int H, int W,
            int K, int R, hipStream_t s, const X3Bnb* bnb = nullptr) {
    if (!xs || !wp || !y) return fail(DMC_E_INVALID, "x3s_conv: null pointer");
    if (!x3s_shape_ok(N, H, W, K, R)) return fail(DMC_E_INVALID, "x3s_conv: unsupported shape N=%d H=%d W=%d K=%d R=%d", N, H, W, K, R);
    const int cfg = x3s_choose(N, H, W, R);
    X3Args a;
    a.xs = xs; a.wp = wp; a.y = y; a.addend = addend; a.stat_part = stat_part;
    a.N = N; a.H = H; a.W = W; a.K = K; a.R = R; a.M = N * H * W;
    a.plane_bytes = (unsigned)a.M * 32u;
    a.part_blocks = part_blocks;
    if ((stat_part || bnb) && part_blocks <= 0) return fail(DMC_E_INVALID, "x3s_conv: stat_blocks must be the row count of the partials");
    if (bnb) {
        a.bnb_y = bnb->y; a.bnb_stats = bnb->stats; a.bnb_gamma = bnb->gamma; a.bnb_beta = bnb->beta; a.bnb_mask = bnb->mask;
        a.bnb_relu = bnb->relu; a.bnb_part = bnb->part;
    }
    a.ablate = option(OPT_CONV_ABLATE);
    if (DMC_ABL(a.ablate) & 32) a.stat_part = nullptr;
    switch (cfg) {
        case 0: return launch_x3s<8, 1, 608>(a, s);
        case 1: return launch_x3s<4, 1, 320>(a, s);
        case 2: return launch_x3s<2, 1, 192>(a, s);
        case 3: return launch_x3s<4, 2, 608>(a, s);
        case 4: return launch_x3s<4, 1, 225>(a, s);
        default: return fail(DMC_E_INVALID, "x3s_conv: the patch of a %d x %d image does not fit the LDS", H, W);
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------
// dw[co][tap][ci] = sum over pixels of dy[p][co] * x[p + tap][ci]: a GEMM over PIXELS, while both slice tensors are
// channel-contiguous.  ds_read_b64_tr_b16 does the transposition on the way out of LDS: a 16-lane group hands in the
// addresses of a [4 pixels][16 channels] block (lane L: pixel L / 4, channels 4 (L % 4) ..) and lane i receives channel i
// of the four pixels -- two such reads are the eight consecutive k-values (pixels) of one MFMA operand row (channel).
// In a chunk plane four consecutive pixels are 128 contiguous bytes and the two groups of a half-wave read planes whose
// strides are = 128 (mod 256) bytes: conflict-free, no swizzle, at every tap shift.
//   workgroup = 64 co x 64 ci block of dw, all nine taps, a run of steps; 12 waves = 4 quarters (32 co x 32 ci) x 3 tap
//   rows, three per SIMD; 3 x 16 accumulators per wave, kept for the whole run.
//   The pixel space is walked in PADDED image rows (every image has a zero row above and below, H + 2 rows): step s
//   covers padded rows [R s, R s + R) -- its dy tile [R x W pixel slots][64 co] (pad rows arrive as zeros: range check)
//   and the R + 2 input rows around it, which live in a RING of input rows in LDS ([ring row][W + 2 pixels][64 ci]): each
//   step only transfers the R rows that entered the window, 3x less staging than a patch per step, and image borders
//   need no special case.  Transfers of step s + 1 are issued at the top of step s (dy double-buffered), one barrier per
//   step.  Partials [group][Cout][9][Cin] are summed in group order by x3s_wgrad_reduce_kernel: deterministic.
struct X3WgArgs {
    const void* xs;        // [3][Cin/16][M][16]
    const void* dys;       // [3][Cout/16][M][16]
    float* part;           // [groups][Cout][9][Cin]
    int N, H, Cin, Cout;
    unsigned plane_bytes;  // M * 32
    int steps, per_group, tiles_ci;
};

template <int W_, int R>
struct WgGeom {
    static constexpr int PW = W_ + 2;
    // ring = NGRP groups of R rows.  A step reads its own group and the last / first row of its neighbours: groups s - 1,
    // s, s + 1 resident, s + 2 arriving (4 groups); when a group is a whole padded image (R = H + 2) the neighbours are only
    // touched by pad-row pixel slots, whose dy is zero: 2 groups, one arriving.
    static constexpr int NGRP = R >= W_ + 2 ? 2 : 4, LA = NGRP / 2;
    static constexpr int NRING = NGRP * R;
    static constexpr int XPX = NRING * PW;                             // ring pixels
    static constexpr int XPL = ((XPX + 3) / 8) * 8 + 4;                // plane stride in pixels: = 4 (mod 8) -> 128 (mod 256) bytes
    static constexpr int NPX = R * W_;                                 // pixel slots per step
    static constexpr int NKB = (NPX + 15) / 16;
    static constexpr int DYP = NKB * 16;
    static constexpr int DYT = ((DYP + 31) / 32) * 32;                 // transfers move 32 pixel rows
    static constexpr int DYPL = ((DYT + 3) / 8) * 8 + 4;
    static constexpr int XI = (R * PW + 31) / 32;                      // transfers per plane and step: input rows / dy tile
    static constexpr int DI = DYT / 32;
    static constexpr int XBYTES = 12 * XPL * 32, DYBYTES = 12 * DYPL * 32;
    static constexpr int LDS = XBYTES + 2 * DYBYTES;
};

template <int W_, int R>
__global__ __launch_bounds__(768) void x3s_wgrad_kernel(X3WgArgs a) {
    using G = WgGeom<W_, R>;
    constexpr int PW = G::PW, NRING = G::NRING, XPL = G::XPL, DYPL = G::DYPL, NKB = G::NKB;
    extern __shared__ __attribute__((aligned(1024))) char lds_wg[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5, L = lane & 15, grp = (lane >> 4) & 1;
    const int quarter = wave & 3, trow = wave >> 2;
    const int wi = quarter & 1, wj = quarter >> 1;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const unsigned lds0 = lds_addr_of(lds_wg);
    const int HP = a.H + 2;
    const int s_begin = blockIdx.y * a.per_group;
    const int s_end = s_begin + a.per_group < a.steps ? s_begin + a.per_group : a.steps;
    const u32x4 srd_x = make_srd(a.xs), srd_d = make_srd(a.dys);
    // this wave's plane of both tiles: slice = wave / 4, chunk = wave % 4
    const unsigned soff_x = (unsigned)((wave >> 2) * (a.Cin >> 4) + tci * 4 + (wave & 3)) * a.plane_bytes;
    const unsigned soff_d = (unsigned)((wave >> 2) * (a.Cout >> 4) + tco * 4 + (wave & 3)) * a.plane_bytes;
    const unsigned xplane = lds0 + (unsigned)wave * XPL * 32, dplane = lds0 + G::XBYTES + (unsigned)wave * DYPL * 32;
    const int lpx = lane >> 1;
    const unsigned lhalf = (unsigned)(lane & 1) << 4;

    // global byte offset of pixel (padded row g, column x) in a plane, or OOB
    auto pix_off = [&](int g, int x) -> unsigned {
        const int n = g / HP, y = g - n * HP - 1;
        return (g >= 0 && n < a.N && y >= 0 && y < a.H && x >= 0 && x < W_) ? (unsigned)((n * a.H + y) * W_ + x) * 32u + lhalf : OOB;
    };
    // input rows of group q (padded rows [R q, R q + R)) -> ring slots R (q mod NGRP) ..
    auto issue_x = [&](int q) {
        const int slot0 = (((q % G::NGRP) + G::NGRP) % G::NGRP) * R;
#pragma unroll
        for (int i = 0; i < G::XI; ++i) {
            const int px = 32 * i + lpx;                            // pixel of the R x PW range
            const int r = px / PW, col = px - r * PW;
            if (px < R * PW) dma_buf16(srd_x, pix_off(R * q + r, col - 1), soff_x, xplane + (unsigned)(slot0 * PW + 32 * i) * 32u);
        }
    };
    // dy tile of step s: pixel slot p = r W + x of padded rows R s + r
    auto issue_dy = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < G::DI; ++i) {
            const int p = 32 * i + lpx;
            const int r = p / W_, x = p - r * W_;
            dma_buf16(srd_d, p < G::NPX ? pix_off(R * s + r, x) : OOB, soff_d, dplane + (unsigned)buf * G::DYBYTES + (unsigned)(32 * i) * 32u);
        }
    };

    f32x16 acc[3];
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
    lds_cptr const LB = (lds_cptr)lds_wg;
    // loop-invariant part of the fragment addresses: unit (kb, u) of this lane is pixel slot P = 16 kb + 8 khalf + 4 u + L / 4 of
    // EVERY step -- its row r and column x within the step do not depend on s; only the ring slot of row R s + r + trow - 1
    // does, and (R s + trow - 1) mod NRING is wave-uniform: one compare + select per unit instead of two modulo operations
    // (VALU instructions per MFMA were 2.7-5.5, profiles/r3_pmc_mfma_busy.csv)
    int rr[NKB][2], xcol[NKB][2], dyoff[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int P = 16 * kb + 8 * khalf + 4 * u + (L >> 2);
            dyoff[kb][u] = (2 * wi + grp) * DYPL * 32 + G::XBYTES + (L & 3) * 8 + P * 32;
            if (P >= G::NPX) P = G::NPX - 1;                         // beyond the tile: dy is zero there, any valid input address
            rr[kb][u] = P / W_;
            xcol[kb][u] = (2 * wj + grp) * XPL * 32 + (L & 3) * 8 + (P - rr[kb][u] * W_) * 32;   // + tap column dx * 32 below
        }
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        issue_x(s + G::LA);                                          // the group entering the window
        if (s + 1 < s_end) issue_dy(s + 1, buf ^ 1);
        const int sb = (((R * s + trow - 1) % NRING) + NRING) % NRING;   // ring row of step row 0 for this wave's tap row (wave-uniform)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            int dyo[2], xo[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dyo[u] = dyoff[kb][u] + buf * G::DYBYTES;
                int slot = sb + rr[kb][u];
                slot = slot >= NRING ? slot - NRING : slot;
                xo[u] = xcol[kb][u] + slot * PW * 32;
            }
            u32x4 A[3], B[3][3];
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                tr_read2(LB + dyo[0] + sl * 4 * DYPL * 32, LB + dyo[1] + sl * 4 * DYPL * 32, A[sl]);
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    tr_read2(LB + xo[0] + (sl * 4 * XPL + t) * 32, LB + xo[1] + (sl * 4 * XPL + t) * 32, B[t][sl]);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[t] = mfma_bf16(A[0], B[t][2], acc[t]);
                acc[t] = mfma_bf16(A[2], B[t][0], acc[t]);
                acc[t] = mfma_bf16(A[1], B[t][1], acc[t]);
                acc[t] = mfma_bf16(A[0], B[t][1], acc[t]);
                acc[t] = mfma_bf16(A[1], B[t][0], acc[t]);
                acc[t] = mfma_bf16(A[0], B[t][0], acc[t]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float* out = a.part + (size_t)blockIdx.y * a.Cout * 9 * a.Cin;
    const int ci = tci * 64 + 32 * wj + l31;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = tco * 64 + 32 * wi + 8 * (e >> 2) + 4 * khalf + (e & 3);
            out[((size_t)co * 9 + 3 * trow + t) * a.Cin + ci] = acc[t][e];
        }
}

// dw = sum over groups of the partials, fixed order (16 group lanes x 16 float4 columns per workgroup)
__global__ __launch_bounds__(256) void x3s_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int ngroup, long numel) {
    __shared__ float4 red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (long i0 = (long)blockIdx.x * 64; i0 < numel; i0 += (long)gridDim.x * 64) {
        const long i = i0 + 4 * o;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < numel)
            for (int k = sl; k < ngroup; k += 16) {
                const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * numel + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        red[sl][o] = s;
        __syncthreads();
        if (sl == 0 && i < numel) {
            float4 t = red[0][o];
#pragma unroll
            for (int k = 1; k < 16; ++k) { const float4 v = red[k][o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(dw + i) = t;
        }
        __syncthreads();
    }
}

struct X3WgPlan { int R, steps, tiles, groups, per_group; };
bool x3s_wgrad_plan(int N, int H, int W, int Cin, int Cout, X3WgPlan& p) {
    p.R = W == 56 ? 1 : W == 28 ? 2 : W == 14 ? 4 : W == 7 ? 9 : 0;
    if (!p.R || (H + 2) % p.R != 0 || (W == 7 && H != 7) || Cin % 64 != 0 || Cout % 64 != 0) return false;
    p.steps = N * (H + 2) / p.R;
    p.tiles = (Cout / 64) * (Cin / 64);
    int groups = 256 / p.tiles;
    if (groups < 1) groups = 1;
    if (groups > p.steps) groups = p.steps;
    p.per_group = (p.steps + groups - 1) / groups;
    p.groups = (p.steps + p.per_group - 1) / p.per_group;
    return true;
}

template <int W_, int R>
int launch_x3s_wgrad(const X3WgPlan& p, X3WgArgs a, float* dw, hipStream_t s) {
    using G = WgGeom<W_, R>;
    static_assert(G::LDS <= 160 * 1024, "weight-gradient tiles exceed the LDS");
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&x3s_wgrad_kernel<W_, R>), G::LDS);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "x3s_wgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    x3s_wgrad_kernel<W_, R><<<dim3(p.tiles, p.groups), 768, G::LDS, s>>>(a);
    int rc = check_launch("x3s_wgrad");
    if (rc || p.groups == 1) return rc;
    const long numel = (long)a.Cout * 9 * a.Cin;
    const long blocks = (numel + 63) / 64;
    x3s_wgrad_reduce_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(a.part, dw, p.groups, numel);
    return check_launch("x3s_wgrad_reduce");
}

// dgamma / dbeta of the upstream BatchNorm from the data gradient's per-workgroup sums (fixed order, one workgroup per channel)
__global__ __launch_bounds__(256) void x3s_bnb_final_kernel(const double* __restrict__ part, int nblk, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[2][4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, ss = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const double2 v = *reinterpret_cast<const double2*>(part + ((size_t)b * C + c) * 2);
        s += v.x;
        ss += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); ss += __shfl_down(ss, o, 64); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    dbeta[c] = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    dgamma[c] = (float)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
}

int stream_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" {

size_t dmc_x3s_slices_bytes(long M, int C) { return (size_t)M * (size_t)C * 6; }

int dmc_x3s_split(const float* x, void* xs, long M, int C, dmc_stream_t stream) {
    if (!x || !xs || M <= 0 || C <= 0 || C % 16 != 0) return fail(DMC_E_INVALID, "dmc_x3s_split: bad argument");
    x3s_split_kernel<<<stream_blocks(M * (C / 8)), 256, 0, (hipStream_t)stream>>>(x, static_cast<unsigned short*>(xs), M, C);
    return check_launch("x3s_split");
}

int dmc_x3s_merge(const void* xs, float* x, long M, int C, dmc_stream_t stream) {
    if (!x || !xs || M <= 0 || C <= 0 || C % 16 != 0) return fail(DMC_E_INVALID, "dmc_x3s_merge: bad argument");
    x3s_merge_kernel<<<stream_blocks(M * C), 256, 0, (hipStream_t)stream>>>(static_cast<const unsigned short*>(xs), x, M, C);
    return check_launch("x3s_merge");
}

size_t dmc_x3s_wpack_bytes(int Cin, int Cout) { return (size_t)Cin * (size_t)Cout * 9 * 6; }

int dmc_x3s_pack_weights(const float* w, void* wpack_f, void* wpack_t, int Cin, int Cout, dmc_stream_t stream) {
    if (!w || (!wpack_f && !wpack_t) || Cin % 64 != 0 || Cout % 64 != 0 || Cin <= 0 || Cout <= 0)
        return fail(DMC_E_INVALID, "dmc_x3s_pack_weights: bad argument");
    const long total = (long)Cin * Cout * 9;
    x3s_pack_w_kernel<<<dim3(stream_blocks(total) > 1024 ? 1024 : stream_blocks(total), 2), 256, 0, (hipStream_t)stream>>>(
        w, static_cast<unsigned short*>(wpack_f), static_cast<unsigned short*>(wpack_t), Cout, Cin, 1);
    return check_launch("x3s_pack_w");
}

int dmc_x3s_conv_supported(int N, int H, int W, int Cin, int Cout) {
    return Cin % 64 == 0 && Cout % 64 == 0 && x3s_shape_ok(N, H, W, Cin, Cout) && x3s_shape_ok(N, H, W, Cout, Cin) &&
           x3s_choose(N, H, W, Cout) >= 0 && x3s_choose(N, H, W, Cin) >= 0 ? 1 : 0;
}

int dmc_x3s_conv_stat_blocks(int N, int H, int W, int Cout) {
    const int cfg = x3s_choose(N, H, W, Cout);
    if (cfg < 0) return 0;
    const long M = (long)N * H * W;
    return (int)((M + X3CFGS[cfg].BM - 1) / X3CFGS[cfg].BM);
}

int dmc_x3s_conv_fwd(const void* xs, const void* wpack_f, float* y, double* stat_partials, int stat_blocks, int N, int H, int W,
                     int Cin, int Cout, dmc_stream_t stream) {
    return run_x3s(xs, wpack_f, nullptr, y, stat_partials, stat_blocks, N, H, W, Cin, Cout, (hipStream_t)stream);
}

int dmc_x3s_conv_dgrad(const void* dys, const void* wpack_t, const float* addend, float* dx, int N, int H, int W, int Cin, int Cout,
                       dmc_stream_t stream) {
    return run_x3s(dys, wpack_t, addend, dx, nullptr, -1, N, H, W, Cout, Cin, (hipStream_t)stream);
}

int dmc_x3s_conv_wgrad_supported(int N, int H, int W, int Cin, int Cout) {
    X3WgPlan p;
    return x3s_wgrad_plan(N, H, W, Cin, Cout, p) && x3s_shape_ok(N, H, W, Cin, Cout) && x3s_shape_ok(N, H, W, Cout, Cin) ? 1 : 0;
}

size_t dmc_x3s_conv_wgrad_bytes(int N, int H, int W, int Cin, int Cout) {
    X3WgPlan p;
    if (!x3s_wgrad_plan(N, H, W, Cin, Cout, p) || p.groups <= 1) return 0;
    return (size_t)p.groups * Cout * 9 * Cin * sizeof(float);
}

int dmc_x3s_conv_wgrad(const void* xs, const void* dys, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout,
                       dmc_stream_t stream) {
    X3WgPlan p;
    if (!xs || !dys || !dw) return fail(DMC_E_INVALID, "dmc_x3s_conv_wgrad: null pointer");
    if (!dmc_x3s_conv_wgrad_supported(N, H, W, Cin, Cout) || !x3s_wgrad_plan(N, H, W, Cin, Cout, p))
        return fail(DMC_E_INVALID, "dmc_x3s_conv_wgrad: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d", N, H, W, Cin, Cout);
    if (p.groups > 1 && !workspace) return fail(DMC_E_INVALID, "dmc_x3s_conv_wgrad: workspace required");
    X3WgArgs a;
    a.xs = xs; a.dys = dys; a.part = p.groups > 1 ? workspace : dw;
    a.N = N; a.H = H; a.Cin = Cin; a.Cout = Cout;
    a.plane_bytes = (unsigned)((long)N * H * W) * 32u;
    a.steps = p.steps; a.per_group = p.per_group; a.tiles_ci = Cin / 64;
    hipStream_t s = (hipStream_t)stream;
    switch (W) {
        case 56: return launch_x3s_wgrad<56, 1>(p, a, dw, s);
        case 28: return launch_x3s_wgrad<28, 2>(p, a, dw, s);
        case 14: return launch_x3s_wgrad<14, 4>(p, a, dw, s);
        default: return launch_x3s_wgrad<7, 9>(p, a, dw, s);
    }
}

/* Data gradient of a 3x3 / stride-2 / padding-1 convolution on pre-split operands, ONE launch (the four input-parity
 * classes keep their own accumulators): dys = slice tensor of dy [N][OH][OW][Cout], dx [N][2 OH][2 OW][Cin] fp32;
 * wpack_t2 from dmc_x3s_pack_weights_s2 (dmc_x3s_wpack_bytes).  Even input sizes only (H = 2 OH, W = 2 OW). */
int dmc_x3s_conv_dgrad_s2_supported(int N, int OH, int OW, int Cin, int Cout) {
    return Cin % 64 == 0 && Cout % 64 == 0 && x3s_shape_ok(N, OH, OW, Cout, Cin) && (long)N * OH * OW * 4 * Cin * 4 < (1L << 40) &&
           patch_pixels_max(N, OH, OW, 128) <= 320 ? 1 : 0;
}

int dmc_x3s_pack_weights_s2(const float* w, void* wpack_t2, int Cin, int Cout, dmc_stream_t stream) {
    if (!w || !wpack_t2 || Cin % 64 != 0 || Cout % 64 != 0 || Cin <= 0 || Cout <= 0)
        return fail(DMC_E_INVALID, "dmc_x3s_pack_weights_s2: bad argument");
    const long total = (long)Cin * Cout * 9;
    x3s_pack_w_kernel<<<dim3(stream_blocks(total) > 1024 ? 1024 : stream_blocks(total), 2), 256, 0, (hipStream_t)stream>>>(
        w, nullptr, static_cast<unsigned short*>(wpack_t2), Cout, Cin, 0);
    return check_launch("x3s_pack_w_s2");
}

int dmc_x3s_conv_dgrad_s2(const void* dys, const void* wpack_t2, float* dx, int N, int OH, int OW, int Cin, int Cout,
                          dmc_stream_t stream) {
    if (!dys || !wpack_t2 || !dx) return fail(DMC_E_INVALID, "dmc_x3s_conv_dgrad_s2: null pointer");
    if (!dmc_x3s_conv_dgrad_s2_supported(N, OH, OW, Cin, Cout))
        return fail(DMC_E_INVALID, "dmc_x3s_conv_dgrad_s2: unsupported shape N=%d OH=%d OW=%d Cin=%d Cout=%d", N, OH, OW, Cin, Cout);
    X3Args a;
    a.xs = dys; a.wp = wpack_t2; a.y = dx; a.addend = nullptr; a.stat_part = nullptr;
    a.N = N; a.H = OH; a.W = OW; a.K = Cout; a.R = Cin; a.M = N * OH * OW;
    a.plane_bytes = (unsigned)a.M * 32u;
    a.ablate = option(OPT_CONV_ABLATE);
    constexpr size_t lds_bytes = 2 * 3 * 320 * 32 + 2 * 3 * 3 * 64 * 32;
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&x3s_conv_kernel<4, 1, 1, 2, 320, true, 4>), (int)lds_bytes);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "x3s_conv_dgrad_s2: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    dim3 grid((a.M + 127) / 128, Cin / 64);
    x3s_conv_kernel<4, 1, 1, 2, 320, true, 4><<<grid, 256, lds_bytes, (hipStream_t)stream>>>(a);
    return check_launch("x3s_conv_dgrad_s2");
}

/* dmc_x3s_conv_dgrad that ALSO reduces the BatchNorm-backward sums of the unit whose output gradient it writes (dx is that
 * unit's dout): bn_y / bn_stats / bn_gamma / bn_beta / bn_relu_mask (nullable) / bn_relu describe that unit's BatchNorm
 * [+ ReLU]; partials = stat_blocks = dmc_x3s_conv_stat_blocks(N, H, W, Cin) rows x Cin x 2 doubles of workspace; dgamma / dbeta [Cin] receive
 * sum(d * xhat) / sum(d) with d = dx [zeroed where the ReLU was off] -- what dmc_bn_act_bwd's first pass computes. */
int dmc_x3s_conv_dgrad_bnb(const void* dys, const void* wpack_t, const float* addend, float* dx, const float* bn_y,
                           const float* bn_stats, const float* bn_gamma, const float* bn_beta, const unsigned char* bn_relu_mask,
                           int bn_relu, double* partials, int stat_blocks, float* dgamma, float* dbeta, int N, int H, int W, int Cin,
                           int Cout, dmc_stream_t stream) {
    if (!bn_y || !bn_stats || !bn_gamma || !bn_beta || !partials || !dgamma || !dbeta)
        return fail(DMC_E_INVALID, "dmc_x3s_conv_dgrad_bnb: null pointer");
    X3Bnb b = {bn_y, bn_stats, bn_gamma, bn_beta, bn_relu_mask, bn_relu, partials};
    int rc = run_x3s(dys, wpack_t, addend, dx, nullptr, stat_blocks, N, H, W, Cout, Cin, (hipStream_t)stream, &b);
    if (rc) return rc;
    x3s_bnb_final_kernel<<<Cin, 256, 0, (hipStream_t)stream>>>(partials, stat_blocks, Cin, dgamma, dbeta);
    return check_launch("x3s_bnb_final");
}

}  // extern "C"
