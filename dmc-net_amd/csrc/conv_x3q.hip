// The STRIDE-2 residual blocks of the classifier on PRE-SPLIT bf16x3 operands (gfx950):
// a 3x3 / stride-2 / padding-1 convolution and the 1x1 / stride-2 shortcut convolution that reads the same input
// (torchvision BasicBlock `conv1` + `downsample[0]` of layer2.0 / layer3.0 / layer4.0 behind code/dmcnet/model.py:305,
// run at :352), forward, data gradient and weight gradient, the shortcut FUSED into the 3x3 launch.
//
// Same arithmetic as conv_x3s.hip (an fp32 value = the exact sum of three bf16 slices, a product = six slice products on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate).  What is new is the input layout.  A stride-2 window touches every input pixel
// 2.25 times on average instead of 9 -- a patch in LDS would be 4x the output tile -- so the PRODUCER of the block input
// writes its slice tensor SPACE-TO-DEPTH ("s2d"):
//     bf16 [3 slices][4 parity classes (py, px)][C / 16][Mq = N (H/2) (W/2) pixels][16 channels]
// class (py, px) holds the pixels (2a + py, 2b + px): each class plane is an ordinary chunk plane over the OUTPUT grid, and
// tap (ky, kx) of output pixel (a, b) reads class ((ky - 1) & 1, (kx - 1) & 1) at (a - [ky == 0], b - [kx == 0]):
//     class (1,1): taps (0,0) (0,2) (2,0) (2,2)   class (0,1): (1,0) (1,2)   class (1,0): (0,1) (2,1)   class (0,0): (1,1)
// and the 1x1 / stride-2 shortcut reads class (0,0) at (a, b): the centre tap's operand, a second set of accumulators.
// Forward = per 16-channel chunk four UNITS (one class plane each, a patch of the tile's pixels + one halo row in LDS)
// and five STEPS of two tap slots; the data gradient runs the same skeleton over the output gradients (nine taps feeding
// four parity-class accumulator sets, as x3s_conv_kernel<CLS = 4>, plus one slot for the shortcut's gradient).
#include "dmc_common.h"
#include "x3s_common.h"
#include <type_traits>
#include <utility>

using namespace dmc;
using namespace dmc::x3;

namespace {

// ---- tile geometry ------------------------------------------------------------------------------------------
constexpr int QBM = 256, QBN = 64;                  // the full-size tile: 256 pixels x 64 output rows (channels) per workgroup
constexpr int QWSLOT = 3 * QBN * 32;                // one tap slot's weights: 3 slices x 64 rows x 16 k = 6 KB
constexpr int QWB = 3 * QWSLOT;                     // a weight buffer holds a step of up to three slots
constexpr int QEP = 272, QETILE = 32 * QEP;         // epilogue: a wave's [32 pixels][64 channels] fp32 tile, rows padded

// Geometry of a kernel variant: NW waves, NPB patch buffers, NT patch transfers (32 pixel rows each) per slice.
//   QG<8, 3, 12>: 8 waves, 256 (TN = 2) / 128 (TN = 1) pixels, three patch buffers: 147.7 KB, one workgroup per CU;
//   QG<4, 2, 7>:  4 waves x (32 pixels x 64 rows) = 128 pixels, two patch buffers of <= 224 pixels: 80,064 B -- TWO workgroups
//                 per CU, one's prologue / epilogue under the other's main loop (the x3s configuration-4 recipe): the short K
//                 loops of these layers (4-32 chunks) spend a third of a workgroup's time outside the main loop.
template <int NW_, int NPB_, int NT_>
struct QG {
    static constexpr int NW = NW_, NPB = NPB_, NT = NT_;
    static constexpr int NA = NW / 2;                   // waves that move patches (the others move weights)
    static constexpr int KP = (NT + NA - 1) / NA;       // patch transfers per issuing wave and slice
    static constexpr int PT = 3 * KP;                   // ... per unit
    static constexpr int KW = (18 + NA - 1) / NA;       // weight transfers per issuing wave and step (<= 18)
    static constexpr int ZROW = 32 * NT;                // the zero row behind the staged rows
    static constexpr int PSL = (ZROW + 1) * 32;         // bytes of one slice region of a patch buffer
    static constexpr int PB = 3 * PSL;
    static constexpr int WOFF = NPB * PB;
    static constexpr int LDS_OP = WOFF + 2 * QWB;
    static constexpr int LDS_EP = NW * QETILE + 8192;
    static constexpr int LDS = LDS_OP > LDS_EP ? LDS_OP : LDS_EP;
    static constexpr bool COUNTED = NPB == 3;           // three buffers: the newest patch may stay in flight (vmcnt(PT))
    static_assert(!COUNTED || NT % NA == 0, "counted waits need the same number of transfers from every patch wave");
};
using QGBig = QG<8, 3, 12>;
using QGTwo = QG<4, 2, 7>;

struct QSlot { int sh, acc; };                      // address shift (index into the program's shift table), accumulator set
struct QStep { int unit, nslot; QSlot s[3]; };

// Forward.  Shifts 0: (-1,-1)  1: (-1,0)  2: (0,-1)  3: (0,0).  Units = class planes (1,1), (0,1), (1,0), (0,0) of chunk c.
// Slot order (= order of the packed weights): taps 0, 2 | 6, 8 | 3, 5 | 1, 7 | 4, shortcut.
struct ProgFwd {
    static constexpr int NUNIT = 4, NSTEP = 5, NACC = 2, NSH = 4, HALO_UP = 1, HALO_DN = 0, NSLOTS = 10;
    static constexpr int SHY[4] = {-1, -1, 0, 0}, SHX[4] = {-1, 0, -1, 0};
    static constexpr int UNIT_T[4] = {0, 0, 0, 0};            // source tensor of the unit
    static constexpr int UNIT_G[4] = {3, 1, 2, 0};            // plane group: plane = UNIT_G * nchunk + c (class 2 py + px)
    static constexpr QStep STEP[5] = {{0, 2, {{0, 0}, {1, 0}, {0, 0}}}, {0, 2, {{2, 0}, {3, 0}, {0, 0}}}, {1, 2, {{2, 0}, {3, 0}, {0, 0}}},
                                      {2, 2, {{1, 0}, {3, 0}, {0, 0}}}, {3, 2, {{3, 0}, {3, 1}, {0, 0}}}};
};
// Data gradient (the launch's pixels are the positions (a, b) of dy).  Shifts 0: (0,0)  1: (0,+1)  2: (+1,0)  3: (+1,+1):
// tap (ky, kx) reads dy at (a + [ky == 0], b + [kx == 0]) and feeds input parity class (ky != 1, kx != 1).
// Unit 0 = plane c of dy (3x3 branch), three steps of one tap row; unit 1 = plane c of the shortcut's dy, one slot.
struct ProgDgrad {
    static constexpr int NUNIT = 2, NSTEP = 4, NACC = 4, NSH = 4, HALO_UP = 0, HALO_DN = 1, NSLOTS = 10;
    static constexpr int SHY[4] = {0, 0, 1, 1}, SHX[4] = {0, 1, 0, 1};
    static constexpr int UNIT_T[2] = {0, 1};
    static constexpr int UNIT_G[2] = {0, 0};
    static constexpr QStep STEP[4] = {{0, 3, {{3, 3}, {2, 2}, {2, 3}}}, {0, 3, {{1, 1}, {0, 0}, {0, 1}}}, {0, 3, {{1, 3}, {0, 2}, {0, 3}}},
                                      {1, 1, {{0, 0}, {0, 0}, {0, 0}}}};
};

template <class P> constexpr int slots_before(int j) { int n = 0; for (int i = 0; i < j; ++i) n += P::STEP[i].nslot; return n; }

struct X3qArgs {
    const void* t0;        // slice tensor 0 (forward: the s2d input; data gradient: dy of the 3x3 branch)
    const void* t1;        // slice tensor 1 (data gradient: dy of the shortcut branch)
    const void* wp;        // packed weights of this direction
    float* y0;             // forward: 3x3 result [M][R]; data gradient: dx [N][2H][2W][R]
    float* y1;             // forward: shortcut result [M][R]
    double* part0;         // forward: BatchNorm partials of y0 / y1 [gridDim.x][R][2] or null
    double* part1;
    int N, H, W;           // the OUTPUT grid of the stride-2 convolution (= the grid of every class plane / of dy)
    int K, R, M;           // contraction channels, rows (result channels) of this launch, N * H * W
    unsigned plane_bytes;  // M * 32
    int pps0, pps1;        // planes per slice of t0 / t1
};

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

template <class P, int TN, class GEO>
__global__ __launch_bounds__(GEO::NW * 64) void x3q_conv_kernel(X3qArgs a) {
    constexpr int NSH = P::NSH, NACC = P::NACC, NSTEP = P::NSTEP, NUNIT = P::NUNIT;
    constexpr int QNW = GEO::NW, QNPB = GEO::NPB, QZROW = GEO::ZROW, QPSL = GEO::PSL, QPB = GEO::PB, QWOFF = GEO::WOFF, QPT = GEO::PT;
    constexpr int NA = GEO::NA, KP = GEO::KP, KW = GEO::KW;
    constexpr int WN = 2 / TN, WM = QNW / WN, BM = 32 * WM;  // waves along rows (channels) / pixels; pixels per workgroup
    static_assert(TN == 1 || TN == 2, "a wave owns 32 or 64 of the workgroup's 64 rows");
    constexpr int CHB = P::NSLOTS * QWSLOT;                  // packed weight bytes per (64-row block, chunk)
    static_assert(P::NSLOTS % 2 == 0, "the fragment registers alternate per slot: an even number per chunk keeps the roles fixed");
    extern __shared__ __attribute__((aligned(1024))) char lds_q[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, khalf = lane >> 5;
    const unsigned lds0 = lds_addr_of(lds_q);

    // ---- patch geometry (wave-uniform): a contiguous run of pixels of the class plane, from the start of the row above
    // (forward) the tile's first pixel to the end of the row of (below: data gradient) its last one; no halo columns,
    // taps that leave the image read the zero row (conv_x3s.hip) ----
    const int HW = a.H * a.W, PW = a.W;
    const int m0 = blockIdx.x * BM;
    const int mlast = (m0 + BM < a.M ? m0 + BM : a.M) - 1;
    const int r_first = m0 / PW, r_last = mlast / PW;
    const int f0 = (r_first - P::HALO_UP > 0 ? r_first - P::HALO_UP : 0) * PW;
    const int f1 = (r_last + 1 + P::HALO_DN) * PW < a.M ? (r_last + 1 + P::HALO_DN) * PW : a.M;
    const int PP = f1 - f0;                                  // <= QZROW (checked by the host)

    // ---- transfers.  Waves 0-3 move patches (wave w: rows 32 (w + 4 k) .., k = 0..2, of each slice: ALWAYS nine
    // transfers per unit -- rows beyond the patch arrive as zeros -- so that s_waitcnt vmcnt(9) means "all but the newest
    // patch"); waves 4-7 move the weights ----
    const bool grp_a = wave < NA;
    const int wq = wave % NA;
    unsigned pvoff[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int pr = 32 * (wq + NA * k) + (lane >> 1), h = lane & 1;
        pvoff[k] = pr < PP ? (unsigned)(f0 + pr) * 32u + (unsigned)((h ^ ((pr >> 3) & 1)) << 4) : OOB;
    }
    const int nchunk = a.K >> 4;
    const int total_units = nchunk * NUNIT, total_steps = nchunk * NSTEP;
    const u32x4 srd0 = make_srd(a.t0), srd1 = make_srd(a.t1 ? a.t1 : a.t0);
    const u32x4 srd_w = make_srd(reinterpret_cast<const char*>(a.wp) + (size_t)blockIdx.y * nchunk * CHB);
    const unsigned lane16 = (unsigned)lane * 16u;

    // transfer i (0..8) of this wave for unit (c, u) into patch buffer pbuf
    auto dma_patch = [&](int c, auto uc, int pbuf, int i) {
        constexpr int u = decltype(uc)::value;
        constexpr int ut = P::UNIT_T[u], ug = P::UNIT_G[u];
        const int s = i / KP, k = i - KP * s;
        const unsigned soff = (unsigned)(s * (ut ? a.pps1 : a.pps0) + ug * nchunk + c) * a.plane_bytes;
        if (GEO::COUNTED || wq + NA * k < GEO::NT)            // (counted waits: always issued, rows beyond the patch arrive as zeros)
            dma_buf16(ut ? srd1 : srd0, pvoff[k], soff, lds0 + pbuf * QPB + s * QPSL + (wq + NA * k) * 1024);
    };
    // transfer e of step (c, j)'s weights into weight buffer wbuf
    auto dma_weight = [&](int c, auto jc, int wbuf, int e) {
        constexpr int j = decltype(jc)::value;
        constexpr int n = P::STEP[j].nslot * 6, off = slots_before<P>(j) * QWSLOT;
        if (e < n) dma_buf16(srd_w, lane16, (unsigned)(c * CHB + off + e * 1024), lds0 + QWOFF + wbuf * QWB + e * 1024);
    };

    // ---- fragment addresses ----
    int xaddr[NSH];
    {
        int m = m0 + wm * 32 + l31;
        if (m > mlast) m = mlast;                            // rows beyond M: a valid address, result not stored
        const int n = m / HW, rem = m - n * HW, yy = rem / a.W, xx = rem - yy * a.W;
        const int pp = m - f0;
        static_for<NSH>([&](auto sc) {
            constexpr int sh = decltype(sc)::value;
            constexpr int dy = P::SHY[sh], dx = P::SHX[sh];
            const int row = (xx + dx >= 0 && xx + dx < a.W && yy + dy >= 0 && yy + dy < a.H) ? pp + dy * PW + dx : QZROW;
            xaddr[sh] = row * 32 + ((khalf ^ ((row >> 3) & 1)) << 4);
        });
    }
    const int waddr = QWOFF + (wn * TN * 32 + l31) * 32 + ((khalf ^ ((l31 >> 3) & 1)) << 4);

    f32x16 acc[NACC][TN];
#pragma unroll
    for (int s = 0; s < NACC; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][j][e] = 0.f;

    lds_cptr const L = (lds_cptr)lds_q;
    struct Frag { u32x4 X[3], W[TN][3]; };                   // one tap slot's operands: 3 + 3 TN reads, 6 TN MFMAs
    auto load_frags = [&](Frag& f, int xa, int wa, auto kc) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                f.W[j][s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(L + wa + ((k * 3 + s) * QBN + 32 * j) * 32);
            f.X[s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(L + xa + s * QPSL);
        }
    };
    // slice products (weight, input): (0,2) (0,1) (0,0) (1,1) (1,0) (2,0); `between(p)` runs behind product p
    auto mfma_slot = [&](const Frag& f, auto ac, auto&& between) {
        constexpr int A = decltype(ac)::value;
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            constexpr int WS[6] = {0, 0, 0, 1, 1, 2}, XS[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[A][j] = mfma_bf16(f.W[j][WS[p]], f.X[XS[p]], acc[A][j]);
            between(p);
        }
    };
    auto nothing = [](int) {};
    constexpr int NRD = 3 + 3 * TN, NMF = 6 * TN;
    auto interleave = [&]() {                                // one fragment read behind each of the first MFMAs
#pragma unroll
        for (int k = 0; k < (NRD < NMF ? NRD : NMF); ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NMF > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
    };

    // ---- prologue: zero rows, the first three patches, the first two steps' weights ----
    if (tid < 2 * 3 * QNPB)
        *reinterpret_cast<float4*>(lds_q + (tid >> 1) / 3 * QPB + ((tid >> 1) % 3) * QPSL + QZROW * 32 + (tid & 1) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grp_a) {
        static_for<QNPB>([&](auto uc) {
            constexpr int U = decltype(uc)::value;           // global unit U = chunk U / NUNIT, unit U % NUNIT
            if (U < total_units) {
#pragma unroll
                for (int i = 0; i < QPT; ++i) dma_patch(U / NUNIT, std::integral_constant<int, U % NUNIT>{}, U, i);
            }
        });
    } else {
        static_for<2>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (g < total_steps) {
#pragma unroll
                for (int i = 0; i < KW; ++i) dma_weight(g / NSTEP, std::integral_constant<int, g % NSTEP>{}, g, wq + NA * i);
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    Frag F0, F1;
    int pb = 0, wb = 0;                                      // patch buffer of the unit in use, weight buffer of the step in use
    int xcur[NSH];
#pragma unroll
    for (int sh = 0; sh < NSH; ++sh) xcur[sh] = xaddr[sh];
    int wcur = waddr;
    {
        constexpr int sh0 = P::STEP[0].s[0].sh;
        load_frags(F0, xcur[sh0], wcur, std::integral_constant<int, 0>{});
    }

    // One step (compile-time index J within the chunk).  On entry the fragments of its first slot are in registers (loaded
    // behind the previous step's last MFMAs).  The loop is software-pipelined over slots; the step's ONE barrier stands in
    // front of its last slot's MFMAs -- by then every fragment of the step is in registers, so its weight buffer (and,
    // at the end of a unit, its patch buffer) is dead: the transfers of step + 2 / unit + 3 go into them behind those
    // MFMAs, and the next step's first fragments are read there too (nobody waits for LDS after a barrier).
    auto frag_of = [&](auto ic) -> Frag& {
        if constexpr (decltype(ic)::value == 0) return F0; else return F1;
    };
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        static_for<NSTEP>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int JN = (J + 1) % NSTEP;
            constexpr int unit = P::STEP[J].unit, nslot = P::STEP[J].nslot;
            constexpr bool unit_last = (P::STEP[JN].unit != unit) || (NUNIT == 1 && J == NSTEP - 1);
            constexpr int par = slots_before<P>(J) & 1;
            const int g = c * NSTEP + J, U = c * NUNIT + unit;
            static_for<nslot - 1>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                constexpr int sh_next = P::STEP[J].s[k + 1].sh, acc_k = P::STEP[J].s[k].acc;
                load_frags(frag_of(std::integral_constant<int, (par + k + 1) & 1>{}), xcur[sh_next], wcur, std::integral_constant<int, k + 1>{});
                mfma_slot(frag_of(std::integral_constant<int, (par + k) & 1>{}), std::integral_constant<int, acc_k>{}, nothing);
                interleave();
            });
            // this wave's transfers for the next step have landed (patch waves: all but the newest patch, which is two units
            // ahead; at the end of the run nothing newer is in flight)
            if (GEO::COUNTED && grp_a && U + 2 < total_units) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(QPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int pb_old = pb, wb_old = wb;
            if constexpr (unit_last) {
                pb = pb == QNPB - 1 ? 0 : pb + 1;
#pragma unroll
                for (int sh = 0; sh < NSH; ++sh) xcur[sh] = xaddr[sh] + pb * QPB;
            }
            wb ^= 1;
            wcur = waddr + wb * QWB;
            constexpr int kl = nslot - 1;
            constexpr int sh_first = P::STEP[JN].s[0].sh, acc_last = P::STEP[J].s[kl].acc;
            if (g + 1 < total_steps)
                load_frags(frag_of(std::integral_constant<int, (par + kl + 1) & 1>{}), xcur[sh_first], wcur, std::integral_constant<int, 0>{});
            // step g + 2's weights -> the buffer this step used; unit U + 3's patch -> the buffer this unit used
            constexpr int J2 = (J + 2) % NSTEP, C2 = (J + 2) / NSTEP;
            constexpr int U3 = (unit + QNPB) % NUNIT, CU3 = (unit + QNPB) / NUNIT;   // the unit whose patch takes the freed buffer
            const bool wgo = !grp_a && g + 2 < total_steps;
            const bool pgo = grp_a && unit_last && U + QNPB < total_units;
            mfma_slot(frag_of(std::integral_constant<int, (par + kl) & 1>{}), std::integral_constant<int, acc_last>{}, [&](int p) {
                if (wgo) {
#pragma unroll
                    for (int i = p; i < KW; i += 6) dma_weight(c + C2, std::integral_constant<int, J2>{}, wb_old, wq + NA * i);
                }
                if constexpr (unit_last) {
                    if (pgo) {
#pragma unroll
                        for (int i = p; i < QPT; i += 6) dma_patch(c + CU3, std::integral_constant<int, U3>{}, pb_old, i);
                    }
                }
            });
        });
    }

    // ---- epilogue (as x3s_conv_kernel): per accumulator set the wave's tile [32 pixels][CW = 32 TN channels] goes through its
    // piece of the (dead) operand LDS -- stored row-major (whole 128- / 256-byte pixel rows), summed column-wise in fp64 for
    // the BatchNorm statistics ----
    __syncthreads();                                          // every wave has read its last fragments
    constexpr int CW = 32 * TN, LPR = CW / 4, RPI = 64 / LPR; // channels per wave; lanes per pixel row; pixel rows per iteration
    char* etile = lds_q + wave * QETILE;
    const int mw0 = m0 + wm * 32;
    const int rbase = blockIdx.y * QBN + wn * CW;
    const bool ok = mw0 + l31 <= mlast;
#pragma unroll
    for (int A = 0; A < NACC; ++A) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 v = make_float4(acc[A][j][4 * g4], acc[A][j][4 * g4 + 1], acc[A][j][4 * g4 + 2], acc[A][j][4 * g4 + 3]);
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(etile + l31 * QEP + (32 * j + 8 * g4 + 4 * khalf) * 4) = v;
            }
        float* const yout = (NACC == 2 && A == 1) ? a.y1 : a.y0;
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int prow = RPI * it + lane / LPR;
            const int m = mw0 + prow;
            const float4 v = *reinterpret_cast<const float4*>(etile + prow * QEP + (lane % LPR) * 16);
            if (m <= mlast) {
                size_t opix = (size_t)m;
                if (NACC == 4) {                              // position (n, ya, xb) of dy -> input pixel (2 ya + py, 2 xb + px)
                    const int n = m / HW, rem = m - n * HW, ya = rem / a.W, xb = rem - ya * a.W;
                    opix = ((size_t)n * (2 * a.H) + 2 * ya + (A >> 1)) * (2 * a.W) + 2 * xb + (A & 1);
                }
                *reinterpret_cast<float4*>(yout + opix * a.R + rbase + (lane % LPR) * 4) = v;
            }
        }
        if (NACC == 2) {
            double* const part = A == 1 ? a.part1 : a.part0;
            if (part) {
                double* red = reinterpret_cast<double*>(lds_q + QNW * QETILE);   // [wm][64 channels][2]
                if (lane < CW) {
                    double d1 = 0.0, d2 = 0.0;
#pragma unroll 8
                    for (int p = 0; p < 32; ++p) {
                        const double v = (double)*reinterpret_cast<const float*>(etile + p * QEP + lane * 4);
                        d1 += v;
                        d2 += v * v;
                    }
                    red[(wm * 64 + wn * CW + lane) * 2 + 0] = d1;
                    red[(wm * 64 + wn * CW + lane) * 2 + 1] = d2;
                }
                __syncthreads();
                for (int cc = tid; cc < QBN; cc += QNW * 64) {
                    double e1 = 0.0, e2 = 0.0;
#pragma unroll
                    for (int w = 0; w < WM; ++w) { e1 += red[(w * 64 + cc) * 2 + 0]; e2 += red[(w * 64 + cc) * 2 + 1]; }
                    double* dst = part + ((size_t)blockIdx.x * a.R + blockIdx.y * QBN + cc) * 2;
                    dst[0] = e1; dst[1] = e2;
                }
                __syncthreads();                              // `red` and the tiles are rewritten by the next set
            }
        }
    }
}

// ---- weights ---------------------------------------------------------------------------------------------------
// w3 [Cout][9][Cin] (OHWI: the memory of the channels_last 3x3 weight), w1 [Cout][Cin] (the 1x1 shortcut weight) ->
//   [rows / 64][k / 16][10 slots][3 slices][64 rows][16 k], the 8-channel half of a row swapped when ((row % 64) >> 3) & 1.
// Forward (blockIdx.y = 0): rows = Cout, k = Cin, slots = taps 0 2 6 8 3 5 1 7 4 shortcut (ProgFwd's order);
// data gradient (blockIdx.y = 1): rows = Cin, k = Cout, slots = taps 0..8, shortcut (taps NOT mirrored: ProgDgrad's shifts).
__global__ __launch_bounds__(256) void x3q_pack_w_kernel(const float* __restrict__ w3, const float* __restrict__ w1,
                                                         unsigned short* __restrict__ wf, unsigned short* __restrict__ wt, int Cout, int Cin) {
    const bool transposed = blockIdx.y == 1;
    unsigned short* dst = transposed ? wt : wf;
    if (!dst) return;
    const int R = transposed ? Cin : Cout, K = transposed ? Cout : Cin;
    const int nchunk = K >> 4;
    const long total = (long)R * 10 * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long t = i;
        const int kk = (int)(t & 15); t >>= 4;
        const int rl = (int)(t & 63); t >>= 6;
        const int slot = (int)(t % 10); t /= 10;
        const int ch = (int)(t % nchunk);
        const int rt = (int)(t / nchunk);
        const int r = rt * 64 + rl, k = ch * 16 + kk;
        constexpr int FWD_TAP[10] = {0, 2, 6, 8, 3, 5, 1, 7, 4, 9};
        const int tap = transposed ? slot : FWD_TAP[slot];
        const int co = transposed ? k : r, ci = transposed ? r : k;
        const float v = tap < 9 ? w3[((long)co * 9 + tap) * Cin + ci] : w1[(long)co * Cin + ci];
        unsigned u0, u1, u2;
        split3(v, u0, u1, u2);
        const size_t base = ((((size_t)(rt * nchunk + ch) * 10 + slot) * 3) * 64 + rl) * 16 + (kk ^ (((rl >> 3) & 1) << 3));
        dst[base] = (unsigned short)(u0 >> 16);
        dst[base + 64 * 16] = (unsigned short)(u1 >> 16);
        dst[base + 2 * 64 * 16] = (unsigned short)(u2 >> 16);
    }
}

// ---- s2d slice tensors: stand-alone producer / inverse (tests, inputs nobody split) ------------------------------------
// x [N][H][W][C] fp32 -> xq [3][4][C/16][Mq][16] bf16 (H, W even); one thread = 8 channels of one pixel
__global__ __launch_bounds__(256) void x3q_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ xq, int N, int H, int W, int C) {
    const int c8 = C >> 3, nchunk = C >> 4, H2 = H >> 1, W2 = W >> 1;
    const long total = (long)N * H * W * c8;
    const size_t Mq = (size_t)N * H2 * W2;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const long m = o / c8;
        const int g = (int)(o - m * c8);
        const int xx = (int)(m % W), yy = (int)((m / W) % H), n = (int)(m / ((long)W * H));
        const float4 a = reinterpret_cast<const float4*>(x)[2 * o], b = reinterpret_cast<const float4*>(x)[2 * o + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const int cls = (yy & 1) * 2 + (xx & 1);
        x3s_store8(xq, v, Mq, 4 * nchunk, ((size_t)n * H2 + (yy >> 1)) * W2 + (xx >> 1), g + cls * 2 * nchunk);
    }
}

__global__ __launch_bounds__(256) void x3q_merge_kernel(const unsigned short* __restrict__ xq, float* __restrict__ x, int N, int H, int W, int C) {
    const int nchunk = C >> 4, H2 = H >> 1, W2 = W >> 1;
    const long total = (long)N * H * W * C;
    const size_t Mq = (size_t)N * H2 * W2, plane = Mq * 16;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const long m = o / C;
        const int c = (int)(o - m * C);
        const int xx = (int)(m % W), yy = (int)((m / W) % H), n = (int)(m / ((long)W * H));
        const int cls = (yy & 1) * 2 + (xx & 1);
        const size_t mq = ((size_t)n * H2 + (yy >> 1)) * W2 + (xx >> 1);
        const size_t w = ((size_t)(cls * nchunk + (c >> 4)) * Mq + mq) * 16 + (c & 15);
        const float s0 = __uint_as_float((unsigned)xq[w] << 16);
        const float s1 = __uint_as_float((unsigned)xq[(size_t)4 * nchunk * plane + w] << 16);
        const float s2 = __uint_as_float((unsigned)xq[(size_t)8 * nchunk * plane + w] << 16);
        x[o] = (s0 + s1) + s2;
    }
}

int stream_blocks_q(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// staged pixels the tiles of QBM consecutive pixels of a W-wide grid need at most (tile rows + one halo row); the 128-pixel
// tiles of the TN = 1 variant need fewer
int q_patch_pixels_max(long M, int W) {
    int worst = 0;
    long tiles = (M + QBM - 1) / QBM;
    if (tiles > W) tiles = W;                                 // tile starts repeat with period lcm(QBM, W)
    for (long t = 0; t < tiles; ++t) {
        const long m0 = t * QBM, ml = (m0 + QBM < M ? m0 + QBM : M) - 1;
        const int pp = (int)((ml / W - m0 / W + 2) * W);
        if (pp > worst) worst = pp;
    }
    return worst;
}

int q_patch_pixels_max128(long M, int W) {
    int worst = 0;
    long tiles = (M + 127) / 128;
    if (tiles > W) tiles = W;
    for (long t = 0; t < tiles; ++t) {
        const long m0 = t * 128, ml = (m0 + 128 < M ? m0 + 128 : M) - 1;
        const int pp = (int)((ml / W - m0 / W + 2) * W);
        if (pp > worst) worst = pp;
    }
    return worst;
}

bool q_shape_ok(int N, int OH, int OW, int Cin, int Cout) {
    if (N <= 0 || OH <= 0 || OW <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0 || Cout % 64 != 0) return false;
    const long M = (long)N * OH * OW;
    if (M * 4 * Cin * 6 >= (1L << 31) || M * Cout * 6 >= (1L << 31)) return false;     // slice tensors below the descriptor's 2 GB
    return q_patch_pixels_max(M, OW) <= QGBig::ZROW;
}

// Tile quantisation: one workgroup per CU (147 KB of LDS), so a launch takes ceil(workgroups / 256) rounds.  The half-size
// workgroups (TN = 1: 128 pixels, the same 8 waves on 32 x 32 tiles, 4/3 of the LDS reads per MFMA) cost ~0.73 of a full-size
// one instead of 0.5 (measured, 120 frames: layer4.0's data gradient -- 92 full-size workgroups on 256 CUs -- 158 -> 115 us;
// but layer3.0's forward, 368 -> 736 workgroups, 107 -> 116 us): they are taken only where they save more than that.
// option conv_cfg = 201 / 202 forces TN = 2 / 1 (measurement).
int q_choose_tn(long M, int R) {
    const int forced = option(OPT_CONV_CFG);
    if (forced == 201) return 2;
    if (forced == 202) return 1;
    const long n2 = ((M + 255) / 256) * (R / QBN), n1 = ((M + 127) / 128) * (R / QBN);
    const double t2 = (double)((n2 + 255) / 256), t1 = 0.75 * (double)((n1 + 255) / 256);
    return t1 < t2 ? 1 : 2;
}

template <class P, int TN, class GEO>
int launch_q(const X3qArgs& a, hipStream_t s) {
    static LdsLimit lim;
    const hipError_t attr = lim.raise(reinterpret_cast<const void*>(&x3q_conv_kernel<P, TN, GEO>), GEO::LDS);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "x3q_conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    constexpr int BM = 32 * GEO::NW / (2 / TN);
    dim3 grid((a.M + BM - 1) / BM, a.R / QBN);
    x3q_conv_kernel<P, TN, GEO><<<grid, GEO::NW * 64, GEO::LDS, s>>>(a);
    return check_launch("x3q_conv");
}

// variant of a launch: 0 = QGBig, TN = 2 (256 pixels); 1 = QGBig, TN = 1 (128 pixels, 32 x 32 wave tiles); 2 = QGTwo (128 pixels,
// 4 waves, two workgroups per CU).  option conv_cfg: 201 / 202 / 203 force variant 0 / 1 / 2 (203 only where the patch fits).
int q_patch_pixels_max128(long M, int W);
int q_variant(long M, int W, int R) {
    const int forced = option(OPT_CONV_CFG);
    const bool two_ok = q_patch_pixels_max128(M, W) <= QGTwo::ZROW;
    if (forced == 201) return 0;
    if (forced == 202) return 1;
    if (forced == 203 && two_ok) return 2;
    // two workgroups per CU wherever the patch fits and the launch has at least as many workgroups as the chip has slots
    if (forced != 204 && two_ok && ((M + 127) / 128) * (R / QBN) >= 512) return 2;
    return q_choose_tn(M, R) == 2 ? 0 : 1;
}
int q_block_pixels(int variant) { return variant == 0 ? 256 : 128; }


// ---- weight gradient of the 3x3 / stride-2 convolution -------------------------------------------------------------
// dw3[co][ky][kx][ci] = sum over output pixels (n, a, b) of dy[n][a][b][co] * X_cls[n][a + da][b + db][ci], cls / shift of tap
// (ky, kx) as above: a GEMM over PIXELS with both operands channel-contiguous, ds_read_b64_tr_b16 transposing on the way out
// of LDS -- the scheme of x3s_wgrad_kernel (conv_x3s.hip) over the four class planes of the s2d input:
//   workgroup = 64 co x 64 ci block of dw3, all nine taps, a run of steps; 12 waves = 4 quarters (32 co x 32 ci) x 3 tap
//   rows ky (one class-row parity py and row shift per wave), three accumulators (kx = 0, 1, 2) per wave for the whole run.
//   The pixel space is walked in PADDED rows of the output grid (HP = OH + 1 or OH + 2 rows per image: a zero row above,
//   for OW = 14 also one below so that R divides HP); step s covers padded rows [R s, R s + R): its dy tile [R x OW pixel
//   slots][64 co] (pad rows arrive as zeros) and, per class plane, a RING of rows [row][OW + 1 pixels][64 ci] (column 0 = the
//   zero column left of the image) -- py = 1 classes keep three groups of R rows (taps ky = 0 read the row above), py = 0
//   classes two; only the group that enters the window is transferred per step.  One barrier per step; partials
//   [group][Cout][9][Cin] summed in group order by x3s-style reduce: deterministic.
struct X3qWgArgs {
    const void* xq;        // s2d slices [3][4][Cin/16][Mq][16]
    const void* dys;       // [3][Cout/16][Mq][16]
    float* part;           // [groups][Cout][9][Cin]
    int N, OH, Cin, Cout;
    unsigned plane_bytes;  // Mq * 32
    int steps, per_group, tiles_ci;
};

template <int OW_, int R, int HP_>
struct QWgGeom {
    static constexpr int PW = OW_ + 1;
    static constexpr int D1 = 3, D0 = 2;                               // ring groups of the py = 1 / py = 0 classes
    static constexpr int XPL1 = ((D1 * R * PW + 3) / 8) * 8 + 4;       // plane strides in pixels: = 4 (mod 8) -> 128 (mod 256) bytes
    static constexpr int XPL0 = ((D0 * R * PW + 3) / 8) * 8 + 4;
    static constexpr int NPX = R * OW_;                                // pixel slots per step
    static constexpr int NKB = (NPX + 15) / 16;
    static constexpr int DYT = ((NKB * 16 + 31) / 32) * 32;            // transfers move 32 pixel rows
    static constexpr int DYPL = ((DYT + 3) / 8) * 8 + 4;
    static constexpr int XI = (R * PW + 31) / 32;                      // transfers per class plane and step
    static constexpr int DI = DYT / 32;
    // x region: [slice 3][class 4 = 2 py + px][chunk 4][XPL(py) pixels][32 B]
    static constexpr int CLS_OFF[4] = {0, 4 * XPL0 * 32, 8 * XPL0 * 32, 8 * XPL0 * 32 + 4 * XPL1 * 32};
    static constexpr int SLB = 8 * XPL0 * 32 + 8 * XPL1 * 32;          // one slice of the x region
    static constexpr int XBYTES = 3 * SLB, DYBYTES = 12 * DYPL * 32;
    static constexpr int LDS = XBYTES + 2 * DYBYTES;
    static_assert(HP_ % R == 0, "steps are whole groups of padded rows");
};

template <int OW_, int R, int HP_>
__global__ __launch_bounds__(768) void x3q_wgrad_kernel(X3qWgArgs a) {
    using G = QWgGeom<OW_, R, HP_>;
    constexpr int PW = G::PW, DYPL = G::DYPL, NKB = G::NKB;
    extern __shared__ __attribute__((aligned(1024))) char lds_qw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5, L = lane & 15, grp = (lane >> 4) & 1;
    const int quarter = wave & 3, trow = wave >> 2;                      // trow = ky
    const int wi = quarter & 1, wj = quarter >> 1;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const unsigned lds0 = lds_addr_of(lds_qw);
    const int s_begin = blockIdx.y * a.per_group;
    const int s_end = s_begin + a.per_group < a.steps ? s_begin + a.per_group : a.steps;
    const u32x4 srd_x = make_srd(a.xq), srd_d = make_srd(a.dys);
    const int nch_in = a.Cin >> 4;
    // this wave's (slice, chunk) of both operands for the transfers: slice = wave / 4, chunk = wave % 4
    const int tsl = wave >> 2, tch = wave & 3;
    const unsigned soff_d = (unsigned)(tsl * (a.Cout >> 4) + tco * 4 + tch) * a.plane_bytes;
    const unsigned dplane = lds0 + G::XBYTES + (unsigned)wave * DYPL * 32;
    const int lpx = lane >> 1;
    const unsigned lhalf = (unsigned)(lane & 1) << 4;

    // global byte offset of pixel (padded row g, column x) in a plane of the OUTPUT grid, or OOB
    auto pix_off = [&](int g, int x) -> unsigned {
        const int n = g / HP_, y = g - n * HP_ - 1;
        return (g >= 0 && n < a.N && y >= 0 && y < a.OH && x >= 0 && x < OW_) ? (unsigned)((n * a.OH + y) * OW_ + x) * 32u + lhalf : OOB;
    };
    // rows of group q (padded rows [R q, R q + R)) of the four class planes -> their ring slots
    auto issue_x = [&](int q) {
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int D = (cls >> 1) ? G::D1 : G::D0, xpl = (cls >> 1) ? G::XPL1 : G::XPL0;
            const int slot0 = (((q % D) + D) % D) * R;
            const unsigned soff_x = (unsigned)(tsl * 4 * nch_in + cls * nch_in + tci * 4 + tch) * a.plane_bytes;
            const unsigned xplane = lds0 + (unsigned)(tsl * G::SLB + G::CLS_OFF[cls] + tch * xpl * 32);
#pragma unroll
            for (int i = 0; i < G::XI; ++i) {
                const int px = 32 * i + lpx;                            // pixel of the R x PW range
                const int r = px / PW, col = px - r * PW;
                if (px < R * PW) dma_buf16(srd_x, pix_off(R * q + r, col - 1), soff_x, xplane + (unsigned)(slot0 * PW + 32 * i) * 32u);
            }
        }
    };
    auto issue_dy = [&](int s, int buf) {
#pragma unroll
        for (int i = 0; i < G::DI; ++i) {
            const int p = 32 * i + lpx;
            const int r = p / OW_, x = p - r * OW_;
            dma_buf16(srd_d, p < G::NPX ? pix_off(R * s + r, x) : OOB, soff_d, dplane + (unsigned)buf * G::DYBYTES + (unsigned)(32 * i) * 32u);
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (s_begin < s_end) {
        issue_x(s_begin - 1);                                            // (the py = 0 rings do not need it: harmless, overwritten by s_begin + 1)
        issue_x(s_begin);
        issue_dy(s_begin, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // this wave's tap row: class-row parity, row shift, ring depth; its three taps kx = 0, 1, 2: classes (py, 1) (py, 0) (py, 1)
    // at columns x - 1 + 1, x + 1, x + 1 of the padded row (column 0 = the zero column)
    const int py = trow != 1, da = trow == 0 ? -1 : 0;
    const int DR = (py ? G::D1 : G::D0) * R, xpl = py ? G::XPL1 : G::XPL0;
    const int cls_a = py ? G::CLS_OFF[3] : G::CLS_OFF[1];                // class (py, 1)
    const int cls_b = py ? G::CLS_OFF[2] : G::CLS_OFF[0];                // class (py, 0)
    lds_cptr const LB = (lds_cptr)lds_qw;
    // Loop-invariant part of the fragment addresses.  Unit (kb, u) of this lane is pixel slot P = 16 kb + 8 khalf + 4 u + L / 4
    // of every step: its row r = P / OW and column x = P % OW within the step do not depend on s; only the ring slot of row
    // R s + r + da does, and (R s + da) mod DR is wave-uniform: one compare + select per unit instead of two modulo operations.
    // The class bases are folded into two per-unit bases (taps kx = 0, 2 read class (py, 1), kx = 1 class (py, 0)), so that the
    // tap / slice offsets of the 18 B-fragment reads per k-block are instruction immediates.
    int rr[NKB][2], xcol[NKB][2], dyoff[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int P = 16 * kb + 8 * khalf + 4 * u + (L >> 2);
            dyoff[kb][u] = (2 * wi + grp) * DYPL * 32 + G::XBYTES + (L & 3) * 8 + P * 32;
            if (P >= G::NPX) P = G::NPX - 1;                             // beyond the tile: dy is zero there, any valid input address
            rr[kb][u] = P / OW_;
            xcol[kb][u] = (2 * wj + grp) * xpl * 32 + (L & 3) * 8 + (P - rr[kb][u] * OW_) * 32;
        }
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        issue_x(s + 1);                                                  // the group entering the window
        if (s + 1 < s_end) issue_dy(s + 1, buf ^ 1);
        const int sb = (((R * s + da) % DR) + DR) % DR;                  // ring row of step row 0 for this wave's tap row (wave-uniform)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            int dyo[2], xa[2], xb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dyo[u] = dyoff[kb][u] + buf * G::DYBYTES;
                int slot = sb + rr[kb][u];
                slot = slot >= DR ? slot - DR : slot;
                const int xo = xcol[kb][u] + slot * PW * 32;
                xa[u] = xo + cls_a;
                xb[u] = xo + cls_b;
            }
            u32x4 A[3], B[3][3];
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                tr_read2(LB + dyo[0] + sl * 4 * DYPL * 32, LB + dyo[1] + sl * 4 * DYPL * 32, A[sl]);
                tr_read2(LB + xa[0] + sl * G::SLB, LB + xa[1] + sl * G::SLB, B[0][sl]);
                tr_read2(LB + xb[0] + sl * G::SLB + 32, LB + xb[1] + sl * G::SLB + 32, B[1][sl]);
                tr_read2(LB + xa[0] + sl * G::SLB + 32, LB + xa[1] + sl * G::SLB + 32, B[2][sl]);
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

// ---- weight gradient of the 1x1 / stride-2 shortcut: dw1[co][ci] = sum over pixels of dy1[p][co] * X_00[p][ci] ------------------
// A plain GEMM over the Mq pixels of two ordinary chunk-planar tensors (class (0,0) of the s2d input has the output grid's
// pixel order).  Workgroup = 64 co x 64 ci, 4 waves (one 32 x 32 quarter each), steps of 64 pixels, both tiles
// [12 planes][64 pixels] double-buffered; partials per pixel group, fixed-order reduce.  1/10 of the block's FLOPs.
struct X3qWg1Args {
    const void* xq; const void* dys; float* part;
    int Cin, Cout; long Mq; unsigned plane_bytes; int steps, per_group, tiles_ci;
};
constexpr int W1PX = 64, W1PL = 68;                                    // pixels per step; plane stride = 4 (mod 8)
constexpr int W1TILE = 12 * W1PL * 32, W1LDS = 4 * W1TILE;             // x, dy tiles x 2 buffers = 104,448 B

__global__ __launch_bounds__(256) void x3q_wgrad1_kernel(X3qWg1Args a) {
    extern __shared__ __attribute__((aligned(1024))) char lds_w1[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, khalf = lane >> 5, L = lane & 15, grp = (lane >> 4) & 1;
    const int wi = wave & 1, wj = wave >> 1;
    const int tci = blockIdx.x % a.tiles_ci, tco = blockIdx.x / a.tiles_ci;
    const unsigned lds0 = lds_addr_of(lds_w1);
    const int s_begin = blockIdx.y * a.per_group;
    const int s_end = s_begin + a.per_group < a.steps ? s_begin + a.per_group : a.steps;
    const u32x4 srd_x = make_srd(a.xq), srd_d = make_srd(a.dys);
    const int nch_in = a.Cin >> 4, nch_out = a.Cout >> 4;
    const int lpx = lane >> 1;
    const unsigned lhalf = (unsigned)(lane & 1) << 4;
    // wave w moves planes 3 w .. 3 w + 2 (plane = slice * 4 + chunk) of both tiles: two transfers of 32 pixels each
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int pl = 3 * wave + k, sl = pl >> 2, ch = pl & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long p = (long)s * W1PX + 32 * i + lpx;
                const unsigned voff = p < a.Mq ? (unsigned)p * 32u + lhalf : OOB;
                dma_buf16(srd_x, voff, (unsigned)(sl * 4 * nch_in + tci * 4 + ch) * a.plane_bytes,            // class (0,0): plane group 0
                          lds0 + (unsigned)(buf * 2 * W1TILE + pl * W1PL * 32 + 32 * i * 32));
                dma_buf16(srd_d, voff, (unsigned)(sl * nch_out + tco * 4 + ch) * a.plane_bytes,
                          lds0 + (unsigned)(buf * 2 * W1TILE + W1TILE + pl * W1PL * 32 + 32 * i * 32));
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (s_begin < s_end) issue(s_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    lds_cptr const LB = (lds_cptr)lds_w1;
#pragma unroll 1
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) issue(s + 1, buf ^ 1);
        const int xb = buf * 2 * W1TILE + (2 * wj + grp) * W1PL * 32 + (L & 3) * 8;
        const int db = buf * 2 * W1TILE + W1TILE + (2 * wi + grp) * W1PL * 32 + (L & 3) * 8;
#pragma unroll
        for (int kb = 0; kb < W1PX / 16; ++kb) {
            u32x4 A[3], B[3];
            const int P0 = 16 * kb + 8 * khalf + (L >> 2), P1 = P0 + 4;
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                tr_read2(LB + db + sl * 4 * W1PL * 32 + P0 * 32, LB + db + sl * 4 * W1PL * 32 + P1 * 32, A[sl]);
                tr_read2(LB + xb + sl * 4 * W1PL * 32 + P0 * 32, LB + xb + sl * 4 * W1PL * 32 + P1 * 32, B[sl]);
            }
            acc = mfma_bf16(A[0], B[2], acc);
            acc = mfma_bf16(A[2], B[0], acc);
            acc = mfma_bf16(A[1], B[1], acc);
            acc = mfma_bf16(A[0], B[1], acc);
            acc = mfma_bf16(A[1], B[0], acc);
            acc = mfma_bf16(A[0], B[0], acc);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float* out = a.part + (size_t)blockIdx.y * a.Cout * a.Cin;
    const int ci = tci * 64 + 32 * wj + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int co = tco * 64 + 32 * wi + 8 * (e >> 2) + 4 * khalf + (e & 3);
        out[(size_t)co * a.Cin + ci] = acc[e];
    }
}

// dw = sum over groups of the partials, fixed order (16 group lanes x 16 float4 columns per workgroup)
__global__ __launch_bounds__(256) void x3q_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int ngroup, long numel) {
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

struct QWgPlan { int R, HP, steps, tiles, groups, per_group, steps1, groups1, per_group1; };
bool q_wgrad_plan(int N, int OH, int OW, int Cin, int Cout, QWgPlan& p) {
    if (Cin % 64 != 0 || Cout % 64 != 0 || N <= 0) return false;
    if (OW == 28 && OH == 28) { p.R = 1; p.HP = 29; }
    else if (OW == 14 && OH == 14) { p.R = 2; p.HP = 16; }
    else if (OW == 7 && OH == 7) { p.R = 4; p.HP = 8; }
    else return false;
    p.steps = N * p.HP / p.R;
    p.tiles = (Cout / 64) * (Cin / 64);
    int groups = 256 / p.tiles;
    if (groups < 1) groups = 1;
    if (groups > p.steps) groups = p.steps;
    p.per_group = (p.steps + groups - 1) / groups;
    p.groups = (p.steps + p.per_group - 1) / p.per_group;
    const long Mq = (long)N * OH * OW;
    p.steps1 = (int)((Mq + W1PX - 1) / W1PX);
    int g1 = 512 / p.tiles;                                   // 4-wave workgroups: two per CU
    if (g1 < 1) g1 = 1;
    if (g1 > p.steps1) g1 = p.steps1;
    p.per_group1 = (p.steps1 + g1 - 1) / g1;
    p.groups1 = (p.steps1 + p.per_group1 - 1) / p.per_group1;
    return true;
}

template <int OW_, int R, int HP_>
int launch_q_wgrad(const QWgPlan& p, const X3qWgArgs& a, hipStream_t s) {
    using G = QWgGeom<OW_, R, HP_>;
    static_assert(G::LDS <= 160 * 1024, "weight-gradient tiles exceed the LDS");
    static LdsLimit lim;
    const hipError_t attr = lim.raise(reinterpret_cast<const void*>(&x3q_wgrad_kernel<OW_, R, HP_>), G::LDS);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "x3q_wgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    x3q_wgrad_kernel<OW_, R, HP_><<<dim3(p.tiles, p.groups), 768, G::LDS, s>>>(a);
    return check_launch("x3q_wgrad");
}

}  // namespace

extern "C" {

size_t dmc_x3q_wpack_bytes(int Cin, int Cout) { return (size_t)Cin * (size_t)Cout * 10 * 6; }

int dmc_x3q_supported(int N, int OH, int OW, int Cin, int Cout) { return q_shape_ok(N, OH, OW, Cin, Cout) ? 1 : 0; }

int dmc_x3q_stat_blocks(int N, int OH, int OW, int Cout) {
    const long M = (long)N * OH * OW;
    const int bm = q_block_pixels(q_variant(M, OW, Cout));
    return (int)((M + bm - 1) / bm);
}

int dmc_x3q_split(const float* x, void* xq, int N, int H, int W, int C, dmc_stream_t stream) {
    if (!x || !xq || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || C % 16 != 0)
        return fail(DMC_E_INVALID, "dmc_x3q_split: bad argument");
    x3q_split_kernel<<<stream_blocks_q((long)N * H * W * (C / 8)), 256, 0, (hipStream_t)stream>>>(x, static_cast<unsigned short*>(xq), N, H, W, C);
    return check_launch("x3q_split");
}

int dmc_x3q_merge(const void* xq, float* x, int N, int H, int W, int C, dmc_stream_t stream) {
    if (!x || !xq || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || C % 16 != 0)
        return fail(DMC_E_INVALID, "dmc_x3q_merge: bad argument");
    x3q_merge_kernel<<<stream_blocks_q((long)N * H * W * C), 256, 0, (hipStream_t)stream>>>(static_cast<const unsigned short*>(xq), x, N, H, W, C);
    return check_launch("x3q_merge");
}

int dmc_x3q_pack_weights(const float* w3, const float* w1, void* wpack_f, void* wpack_t, int Cin, int Cout, dmc_stream_t stream) {
    if (!w3 || !w1 || (!wpack_f && !wpack_t) || Cin % 64 != 0 || Cout % 64 != 0 || Cin <= 0 || Cout <= 0)
        return fail(DMC_E_INVALID, "dmc_x3q_pack_weights: bad argument");
    const long total = (long)Cin * Cout * 10;
    const int blocks = stream_blocks_q(total) > 1024 ? 1024 : stream_blocks_q(total);
    x3q_pack_w_kernel<<<dim3(blocks, 2), 256, 0, (hipStream_t)stream>>>(w3, w1, static_cast<unsigned short*>(wpack_f),
                                                                           static_cast<unsigned short*>(wpack_t), Cout, Cin);
    return check_launch("x3q_pack_w");
}

int dmc_x3q_conv_fwd(const void* xq, const void* wpack_f, float* y3, float* y1, double* stat_partials3, double* stat_partials1,
                     int stat_blocks, int N, int OH, int OW, int Cin, int Cout, dmc_stream_t stream) {
    if (!xq || !wpack_f || !y3 || !y1) return fail(DMC_E_INVALID, "dmc_x3q_conv_fwd: null pointer");
    if (!q_shape_ok(N, OH, OW, Cin, Cout))
        return fail(DMC_E_INVALID, "dmc_x3q_conv_fwd: unsupported shape N=%d OH=%d OW=%d Cin=%d Cout=%d", N, OH, OW, Cin, Cout);
    const int var = q_variant((long)N * OH * OW, OW, Cout);
    const int rows = (int)(((long)N * OH * OW + q_block_pixels(var) - 1) / q_block_pixels(var));
    if ((stat_partials3 || stat_partials1) && stat_blocks != rows)
        return fail(DMC_E_INVALID, "dmc_x3q_conv_fwd: statistics partials have %d rows but this launch writes %d (dmc_x3q_stat_blocks "
                                   "was called under another conv_cfg option?)", stat_blocks, rows);
    X3qArgs a;
    a.t0 = xq; a.t1 = nullptr; a.wp = wpack_f; a.y0 = y3; a.y1 = y1; a.part0 = stat_partials3; a.part1 = stat_partials1;
    a.N = N; a.H = OH; a.W = OW; a.K = Cin; a.R = Cout; a.M = N * OH * OW;
    a.plane_bytes = (unsigned)a.M * 32u;
    a.pps0 = 4 * (Cin / 16); a.pps1 = 0;
    return var == 0 ? launch_q<ProgFwd, 2, QGBig>(a, (hipStream_t)stream) : var == 1 ? launch_q<ProgFwd, 1, QGBig>(a, (hipStream_t)stream)
                                                                                      : launch_q<ProgFwd, 2, QGTwo>(a, (hipStream_t)stream);
}

int dmc_x3q_conv_dgrad(const void* dys3, const void* dys1, const void* wpack_t, float* dx, int N, int OH, int OW, int Cin, int Cout,
                       dmc_stream_t stream) {
    if (!dys3 || !dys1 || !wpack_t || !dx) return fail(DMC_E_INVALID, "dmc_x3q_conv_dgrad: null pointer");
    if (!q_shape_ok(N, OH, OW, Cin, Cout))
        return fail(DMC_E_INVALID, "dmc_x3q_conv_dgrad: unsupported shape N=%d OH=%d OW=%d Cin=%d Cout=%d", N, OH, OW, Cin, Cout);
    X3qArgs a;
    a.t0 = dys3; a.t1 = dys1; a.wp = wpack_t; a.y0 = dx; a.y1 = nullptr; a.part0 = a.part1 = nullptr;
    a.N = N; a.H = OH; a.W = OW; a.K = Cout; a.R = Cin; a.M = N * OH * OW;
    a.plane_bytes = (unsigned)a.M * 32u;
    a.pps0 = a.pps1 = Cout / 16;
    const int var = q_variant(a.M, OW, Cin);
    return var == 0 ? launch_q<ProgDgrad, 2, QGBig>(a, (hipStream_t)stream) : var == 1 ? launch_q<ProgDgrad, 1, QGBig>(a, (hipStream_t)stream)
                                                                                        : launch_q<ProgDgrad, 2, QGTwo>(a, (hipStream_t)stream);
}

int dmc_x3q_conv_wgrad_supported(int N, int OH, int OW, int Cin, int Cout) {
    QWgPlan p;
    return q_shape_ok(N, OH, OW, Cin, Cout) && q_wgrad_plan(N, OH, OW, Cin, Cout, p) ? 1 : 0;
}

size_t dmc_x3q_conv_wgrad_bytes(int N, int OH, int OW, int Cin, int Cout) {
    QWgPlan p;
    if (!q_wgrad_plan(N, OH, OW, Cin, Cout, p)) return 0;
    return ((size_t)p.groups * 9 + (size_t)p.groups1) * Cout * Cin * sizeof(float);
}

int dmc_x3q_conv_wgrad(const void* xq, const void* dys3, const void* dys1, float* dw3, float* dw1, float* workspace, int N, int OH,
                       int OW, int Cin, int Cout, dmc_stream_t stream) {
    QWgPlan p;
    if (!xq || !dys3 || !dys1 || !dw3 || !dw1 || !workspace) return fail(DMC_E_INVALID, "dmc_x3q_conv_wgrad: null pointer");
    if (!dmc_x3q_conv_wgrad_supported(N, OH, OW, Cin, Cout) || !q_wgrad_plan(N, OH, OW, Cin, Cout, p))
        return fail(DMC_E_INVALID, "dmc_x3q_conv_wgrad: unsupported shape N=%d OH=%d OW=%d Cin=%d Cout=%d", N, OH, OW, Cin, Cout);
    hipStream_t s = (hipStream_t)stream;
    const long Mq = (long)N * OH * OW;
    X3qWgArgs a;
    a.xq = xq; a.dys = dys3; a.part = workspace;
    a.N = N; a.OH = OH; a.Cin = Cin; a.Cout = Cout;
    a.plane_bytes = (unsigned)Mq * 32u;
    a.steps = p.steps; a.per_group = p.per_group; a.tiles_ci = Cin / 64;
    int rc = OW == 28 ? launch_q_wgrad<28, 1, 29>(p, a, s) : OW == 14 ? launch_q_wgrad<14, 2, 16>(p, a, s) : launch_q_wgrad<7, 4, 8>(p, a, s);
    if (rc) return rc;
    const long numel3 = (long)Cout * 9 * Cin, numel1 = (long)Cout * Cin;
    long blocks = (numel3 + 63) / 64;
    x3q_wgrad_reduce_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(workspace, dw3, p.groups, numel3);
    rc = check_launch("x3q_wgrad_reduce");
    if (rc) return rc;
    // the shortcut's weight gradient
    float* part1 = workspace + (size_t)p.groups * numel3;
    X3qWg1Args b;
    b.xq = xq; b.dys = dys1; b.part = part1; b.Cin = Cin; b.Cout = Cout; b.Mq = Mq; b.plane_bytes = (unsigned)Mq * 32u;
    b.steps = p.steps1; b.per_group = p.per_group1; b.tiles_ci = Cin / 64;
    static LdsLimit lim1;
    const hipError_t attr = lim1.raise(reinterpret_cast<const void*>(&x3q_wgrad1_kernel), W1LDS);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "x3q_wgrad1: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    x3q_wgrad1_kernel<<<dim3(p.tiles, p.groups1), 256, W1LDS, s>>>(b);
    rc = check_launch("x3q_wgrad1");
    if (rc) return rc;
    blocks = (numel1 + 63) / 64;
    x3q_wgrad_reduce_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(part1, dw1, p.groups1, numel1);
    return check_launch("x3q_wgrad_reduce1");
}

}  // extern "C"
