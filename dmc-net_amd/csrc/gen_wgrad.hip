// Weight gradient of EstimatorDenseNetTiny, row-sliding form (autograd of /root/reference/code/dmcnet/model.py:187-194; the GEMM
// and its bf16x3 arithmetic are those of gen_bwd_weight_pc_kernel<3> in gen_tiny.hip):
//
//   dW[(k, co)][(ci, dy, dx)] = sum over pixels of g_k[co][y][x] * in[ci][y + dy - 1][x + dx - 1]
//
// gen_bwd_weight_pc_kernel<3> splits every fp32 operand into its three bf16 slices IN the consumer waves: the window of a (ci, dy)
// column is split by the three tile rows that use it, 4.8 vector instructions per MFMA next to 16 matrix clocks, two waves per SIMD:
// 0.41 matrix-pipe busy.  Here every value is split ONCE, by four splitter waves, into bf16-slice rings in LDS; the eight consumer
// waves read ready 16-byte fragments and issue MFMAs:
//
//   * a workgroup walks down a 32-column strip two image rows per step (position); the X ring holds 6 rows x 3 slices x 34 planes
//     (33 inputs + a plane of ones for the bias column), the G ring the same for 32 gradient planes (30 + two zero rows);
//   * the three horizontal taps share ONE aligned B fragment (the strip's own 32 columns of an input row, no column halo): the
//     K slots of tap dx are the gradient pixels x + 1 - dx, so it is the A fragment that shifts -- aligned for dx = 1, and for
//     dx = 0 / 2 formed with v_alignbit from the aligned 16 bytes + one edge dword (pixels 8 kq - 1 and 8 kq + 8, the E ring);
//   * columns are dy-major (g = 33 dy + ci): a 16-lane column tile reads 16 consecutive planes of ONE row -- conflict-free
//     ds_read_b128 at a plane pitch of 24 dwords (tools/ubench/wgrad_lds_banks.py);
//   * consumer wave = (row of the pair, accumulator tiles): tile A (layers 0, 1: inputs < 13 -> column tiles gt = 0, 2, 4) +
//     its bias | tile B gt 0, 1 | tile B gt 2, 3, 4 | tile B gt 5, 6 -- 93 / 90 / 93 / 90 MFMAs per step on the four SIMDs;
//   * the splitters load three positions ahead into registers (saddr loads, one array per wave-instruction), split (22 vector
//     instructions per 4 pixels) and write b64 pairs; one barrier per position;
//   * work = the global sequence of (frame, strip, row pair) steps cut into equal contiguous ranges, one per workgroup (a range
//     that starts inside a strip pays one extra position for the row above): all CUs end together.
//
// Deterministic: fixed ranges, fixed summation order.  Partial layout: gen_wgrad.h.
#include "gen_wgrad.h"

#include <type_traits>

#include "gen_fused_inl.h"

namespace dmc {
namespace {
using namespace fz;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WR_THREADS = 1024;                                       // 8 consumer + 8 splitter waves
constexpr int SW = 32;                                                  // strip width = one MFMA k-block per image row
// rings, in dwords (a dword = two horizontally adjacent pixels of one bf16 slice)
constexpr int XP = 24, XPL = 34, XSL = XPL * XP, XROW = 3 * XSL + 8, XRING = 6 * XROW;      // 16 dwords used per plane row
constexpr int GP = XP, GSL = XSL, GROW = XROW, GRING = XRING;           // the gradient ring: the same geometry (32 of the 34 planes used)
constexpr int ESL = 128, EROW = 3 * ESL, ERING = 6 * EROW;              // [M tile][kq][row of the tile]: lo16 = pixel 8 kq + 8, hi16 = 8 kq - 1
constexpr int WR_LDS = XRING + GRING + ERING;
static_assert(WR_LDS * 4 <= 160 * 1024, "LDS");
static_assert(WR_WPART <= XRING, "the final reduction reuses the X ring");
#ifndef WR_PRIO
#define WR_PRIO 2
#endif
#ifndef WR_BAND
#define WR_BAND 14                                                      // row pairs per band
#endif
constexpr unsigned ONE_PAIR = 0x3F803F80u;                              // two bf16 ones

struct WrArgs {
    const float* mv;
    const float* res;
    const float* feat;
    const float* gout;
    const float* gbuf;
    const float* zero;              // >= 16 bytes of zeros
    float* partials;
    int N, H, W, nstr, HS;          // strips per frame, steps (row pairs) per strip
    int BS, nb;                     // row pairs per band, bands
    int T, groups;                  // steps in all, workgroups
    unsigned long long* prof;       // (-DWR_PROF harness: [group][16 waves][busy, total] clocks)
};

// per-wave clock bookkeeping of the stand-alone harness (-DWR_PROF): cycles between barriers = busy, the rest = waiting at them
#ifdef WR_PROF
struct WrProf {
    unsigned long long busy = 0, t0 = 0, mark = 0, r0 = 0;
    __device__ __forceinline__ void begin() { mark = __builtin_amdgcn_s_memtime(); if (!t0) { t0 = mark; r0 = __builtin_amdgcn_s_memrealtime(); } }
    __device__ __forceinline__ void end() { busy += __builtin_amdgcn_s_memtime() - mark; }
    __device__ __forceinline__ void flush(unsigned long long* prof, int wave, int lane) {
        if (!prof || lane) return;
        unsigned long long* q = prof + ((size_t)blockIdx.x * 16 + wave) * 2;
        q[0] = busy; q[1] = __builtin_amdgcn_s_memtime() - t0;
        if (wave == 0) q[0] = __builtin_amdgcn_s_memrealtime() - r0;      // (wave 0 reports the 100 MHz clock instead of its busy time)
    }
};
#else
struct WrProf {
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void end() {}
    __device__ __forceinline__ void flush(unsigned long long*, int, int) {}
};
#endif
#define WR_BARRIER(prof) do { (prof).end(); step_barrier(); (prof).begin(); } while (0)

// ---- the positions of one workgroup: [PRE] STEP STEP ... per segment ------------------------------------------------------------
// The steps of a frame are ordered (row band, strip, row pair): a workgroup finishes a band of BS row pairs in one strip, then takes
// the same band of the next strip.  The 16-byte edge chunks of the gradient rows pull the neighbouring strip's whole 128-byte line;
// that line is the next segment's own data (and the previous segment's), BS positions away: close enough for the Infinity Cache
// to serve every line's second and third use (strip-major order -- 112 positions between them -- fetched 2.8 GB for 1.5).
struct Sched {
    int t, i, n, strip, band, end;      // step index, row pair in the strip, frame, strip, band, first row pair past this segment
    bool pre;
    __device__ __forceinline__ static int seg_len(const WrArgs& a, int band) { return band == a.nb - 1 ? a.HS - band * a.BS : a.BS; }
    // (segment number, offset in the segment) of step t
    __device__ __forceinline__ static int locate(const WrArgs& a, int t, int& n, int& band, int& strip, int& off) {
        const int FS = a.nstr * a.HS;
        n = t / FS;
        const int u = t - n * FS;
        band = u / (a.nstr * a.BS);
        band = band < a.nb - 1 ? band : a.nb - 1;
        const int v = u - band * a.nstr * a.BS, L = seg_len(a, band);
        strip = v / L;
        off = v - strip * L;
        return (n * a.nb + band) * a.nstr + strip;
    }
    __device__ __forceinline__ void init(const WrArgs& a, int t0) {
        int n_, band_, strip_, off_;
        locate(a, t0, n_, band_, strip_, off_);
        t = t0;
        n = __builtin_amdgcn_readfirstlane(n_); band = __builtin_amdgcn_readfirstlane(band_);
        strip = __builtin_amdgcn_readfirstlane(strip_);
        i = band * a.BS + __builtin_amdgcn_readfirstlane(off_);
        end = band * a.BS + seg_len(a, band);
        pre = true;
    }
    __device__ __forceinline__ void advance(const WrArgs& a) {
        if (pre) { pre = false; return; }
        ++t; ++i;
        if (i == end) {
            pre = true;
            if (++strip == a.nstr) {
                strip = 0;
                if (++band == a.nb) { band = 0; ++n; }
            }
            i = band * a.BS;
            end = i + seg_len(a, band);
        }
    }
};

// ---- splitter ------------------------------------------------------------------------------------------------------------------
// A task = one 16-byte chunk (4 pixels of one plane and row) of a position: 2 rows x 33 input planes x 8 chunks, then 2 rows x 30
// gradient planes x 10 chunks (columns tx0 - 4 .. tx0 + 35: the edge pixels of the shifted fragments) = 1,128 tasks on 8 waves x
// 2 slots (+ a third slot on two of the waves).  Everything that tells the arrays apart is per-lane data (64-bit plane base, bytes
// per frame), so the code has no branches on it; a chunk outside the image -- and every gradient chunk of a PRE position -- is read
// from a.zero: zeros arrive, no masking afterwards.  Per position and task: 6 vector instructions to issue the load (the validity
// of a lane is a scalar mask: four row classes x a column mask kept per strip), 18 to split, 3 + 3 LDS writes.
constexpr int NSPLIT = 8, NTASK = 2 * 33 * 8 + 2 * 30 * 10;
struct Task {
    unsigned long long base;   // its plane in frame 0
    unsigned fbytes;           // bytes per frame of its array
    unsigned voff;             // ((rowp + (input plane ? 1 : 0)) * W + 4 c) * 4: byte offset from (gradient row 0 of the pair, column tx0), wrapping
    unsigned lds;              // dword offset of the chunk's first pair in ring slot 0, slice 0
    int eh;                    // halfword index of this chunk's edge pixel in the E ring (slot 0, slice 0), or -1
    int c4;                    // image column of the chunk relative to tx0
    int flags;                 // bit 0 rowp, 1 live, 2 writes its pairs, 3 edge pixel = element 3 (else element 0), 4 gradient plane
};

__device__ __forceinline__ Task make_task(const WrArgs& a, int t) {
    Task s;
    const unsigned HW = (unsigned)a.H * (unsigned)a.W;
    const bool isx = t < 528, live = t < NTASK;
    const int u = isx ? t : live ? t - 528 : 0;
    const int chunks = isx ? 8 : 10, per_row = isx ? 264 : 300;
    const int rowp = u / per_row, rem = u - rowp * per_row;
    const int P = rem / chunks, c = rem - P * chunks - (isx ? 0 : 1);   // P: plane in its ring
    const float* arr;
    int pl, chan;
    if (isx) {
        if (P < 2) { arr = a.mv; pl = P; chan = 2; }
        else if (P < 5) { arr = a.res; pl = P - 2; chan = 3; }
        else { arr = a.feat; pl = P - 5; chan = NFEAT; }
    } else {
        if (P < NFEAT) { arr = a.gbuf; pl = P; chan = NFEAT; }
        else { arr = a.gout; pl = P - NFEAT; chan = 2; }
    }
    s.base = (unsigned long long)arr + (unsigned long long)pl * HW * 4ull;
    s.fbytes = (unsigned)chan * HW * 4u;
    s.c4 = 4 * c;
    s.voff = (unsigned)(((rowp + (isx ? 1 : 0)) * a.W + 4 * c) * 4);
    const bool main = live && c >= 0 && c < 8;
    s.lds = (unsigned)((isx ? 0 : XRING) + rowp * XROW + P * XP + 2 * (main ? c : 0));
    s.eh = -1;
    int el3 = 0;
    if (!isx && live) {
        int kq = -1, half = 0;
        if (c == -1 || c == 1 || c == 3 || c == 5) { kq = (c + 1) / 2; half = 1; el3 = 1; }
        else if (c == 2 || c == 4 || c == 6 || c == 8) { kq = c / 2 - 1; half = 0; }
        if (kq >= 0) s.eh = 2 * (2 * XRING + rowp * EROW + (P >> 4) * 64 + kq * 16 + (P & 15)) + half;
    }
    s.flags = rowp | (live ? 2 : 0) | (main ? 4 : 0) | (el3 ? 8 : 0) | (isx ? 0 : 16);
    return s;
}

typedef unsigned long long mask_t;
// lane-wise select by a scalar mask (one v_cndmask with the mask in an SGPR pair)
__device__ __forceinline__ unsigned sel_mask(unsigned if0, unsigned if1, mask_t m) {
    unsigned r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m));
    return r;
}

template <int RING>
__device__ __forceinline__ void commit_task(const Task& s, f32x4 v, unsigned* lds) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    unsigned u[3][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                       // (pairs: the subtractions become v_pk_add_f32)
        const f32x2 x = {v[2 * h], v[2 * h + 1]};
        u[0][2 * h] = __float_as_uint(x.x); u[0][2 * h + 1] = __float_as_uint(x.y);
        const f32x2 t0 = {__uint_as_float(u[0][2 * h] & 0xffff0000u), __uint_as_float(u[0][2 * h + 1] & 0xffff0000u)};
        const f32x2 r1 = x - t0;
        u[1][2 * h] = __float_as_uint(r1.x); u[1][2 * h + 1] = __float_as_uint(r1.y);
        const f32x2 t1 = {__uint_as_float(u[1][2 * h] & 0xffff0000u), __uint_as_float(u[1][2 * h + 1] & 0xffff0000u)};
        const f32x2 r2 = r1 - t1;
        u[2][2 * h] = __float_as_uint(r2.x); u[2][2 * h + 1] = __float_as_uint(r2.y);
    }
    if (s.flags & 4) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x2 w;
            w.x = __builtin_amdgcn_perm(u[q][1], u[q][0], 0x07060302u);
            w.y = __builtin_amdgcn_perm(u[q][3], u[q][2], 0x07060302u);
            *reinterpret_cast<u32x2*>(lds + s.lds + RING * 2 * XROW + q * XSL) = w;
        }
    }
    if (s.eh >= 0) {
        unsigned short* e16 = reinterpret_cast<unsigned short*>(lds);
        const bool el3 = s.flags & 8;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            e16[s.eh + 2 * (RING * 2 * EROW + q * ESL)] = (unsigned short)((el3 ? u[q][3] : u[q][0]) >> 16);
    }
}

template <int NS>
__device__ __forceinline__ void run_splitter(const WrArgs& a, unsigned* lds, int sw, int lane, int t0, int iters) {
    Task tk[NS];
    mask_t mrow[NS][4], mcol[NS];            // lanes of (input row 0, input row 1, gradient row 0, gradient row 1); lanes whose column is inside
    unsigned long long fp[NS];               // the task's plane in the current frame
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        tk[s] = make_task(a, s * (NSPLIT * 64) + sw * 64 + lane);
        const int cls = (tk[s].flags & 1) | ((tk[s].flags & 16) >> 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) mrow[s][c] = __builtin_amdgcn_ballot_w64((tk[s].flags & 2) && cls == c);
        mcol[s] = 0; fp[s] = 0;
    }
    Sched is;
    is.init(a, t0);
    int cur_n = -1, cur_strip = -1;
    struct Stage { f32x4 r[NS]; } st[3];
    // (loads are issued at every position, also past the end of the range, so that the number in flight is the same everywhere:
    // the compiler's waits count them = three positions ahead)
    auto issue = [&](Stage& g) {
        const int n = is.n < a.N ? is.n : a.N - 1, tx0 = is.strip * SW;   // (positions past the end of the range are loaded, never used)
        if (n != cur_n || is.strip != cur_strip) {                       // a new strip: plane pointers and column masks
            cur_n = n; cur_strip = is.strip;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                fp[s] = tk[s].base + (unsigned long long)n * tk[s].fbytes;
                mcol[s] = __builtin_amdgcn_ballot_w64((unsigned)(tx0 + tk[s].c4) < (unsigned)a.W);
            }
        }
        // input rows rowg - 1, rowg (PRE) or rowg + 1, rowg + 2 (STEP); gradient rows rowg, rowg + 1 (STEP only)
        const int rowg = 2 * is.i, rowx = is.pre ? rowg - 1 : rowg + 1;
        const bool okx0 = (unsigned)rowx < (unsigned)a.H, okx1 = (unsigned)(rowx + 1) < (unsigned)a.H;
        const bool okg0 = !is.pre && rowg < a.H, okg1 = !is.pre && rowg + 1 < a.H;
        const unsigned delta = (unsigned)(((is.pre ? rowg - 2 : rowg) * a.W + tx0) * 4);   // (task offsets count input rows from rowg + 1)
        const unsigned long long zero = (unsigned long long)a.zero;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const mask_t ok = ((okx0 ? mrow[s][0] : 0) | (okx1 ? mrow[s][1] : 0) | (okg0 ? mrow[s][2] : 0) | (okg1 ? mrow[s][3] : 0)) & mcol[s];
            const unsigned long long addr = fp[s] + (unsigned)(tk[s].voff + delta);
            const unsigned lo = sel_mask((unsigned)zero, (unsigned)addr, ok), hi = sel_mask((unsigned)(zero >> 32), (unsigned)(addr >> 32), ok);
            typedef __attribute__((address_space(1))) f32x4 gf4;
#ifdef WR_NO_LOAD                 // (tools/ubench/gen_wgrad_time.hip: the kernel without its global loads)
            g.r[s] = (f32x4){1.f, 2.f, 3.f, (float)(lo + hi)};
#else
            g.r[s] = *(const gf4*)(((unsigned long long)hi << 32) | lo);
#endif
        }
        is.advance(a);
    };
    auto commit = [&](const Stage& g, auto ringc) {
        constexpr int RING = decltype(ringc)::value;
#ifdef WR_NO_GSPLIT               // (harness: what the splitters cost without the gradient planes' share -- results wrong)
        commit_task<RING>(tk[0], g.r[0], lds);
        if (g.r[NS - 1].x == 12345.f) commit_task<RING>(tk[NS - 1], g.r[NS - 1], lds);
#else
#pragma unroll
        for (int s = 0; s < NS; ++s) commit_task<RING>(tk[s], g.r[s], lds);
#endif
    };
    WrProf prof;
    // the splitters are the pole of a position (0.94 busy against 0.45-0.6 of the consumers, tools/ubench/gen_wgrad_time.hip -DWR_PROF):
    // their vector instructions go first, the consumers' MFMAs fill in (0.709 -> 0.650 ms)
    __builtin_amdgcn_s_setprio(WR_PRIO);
    issue(st[0]);
    issue(st[1]);
    issue(st[2]);
    prof.begin();
    // whole triples of positions, nothing conditional (a position past the end commits loaded-but-unused data into ring slots that
    // no consumer reads any more): a loop the compiler's wait-count analysis follows exactly
    for (int it = 0; it < iters; ++it) {
        commit(st[0], std::integral_constant<int, 0>{});
        issue(st[0]);
        WR_BARRIER(prof);
        commit(st[1], std::integral_constant<int, 1>{});
        issue(st[1]);
        WR_BARRIER(prof);
        commit(st[2], std::integral_constant<int, 2>{});
        issue(st[2]);
        WR_BARRIER(prof);
    }
    prof.flush(a.prof, 8 + sw, lane);
}

// ---- consumers -----------------------------------------------------------------------------------------------------------------
struct Frag3 { u32x4 s[3]; };
__device__ __forceinline__ f32x4 mfma_x3(const Frag3& x, const Frag3& y, f32x4 c) {
    // small terms first: (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
    auto mm = [&](int i, int j, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x.s[i]), __builtin_bit_cast(bf16x8, y.s[j]), acc, 0, 0, 0);
    };
    c = mm(0, 2, c); c = mm(2, 0, c); c = mm(1, 1, c); c = mm(0, 1, c); c = mm(1, 0, c); c = mm(0, 0, c);
    return c;
}

// M: accumulator row tile (0 = planes 0..15, 1 = planes 16..31); NW column groups GT0, GT1, GT2; ONES: the bias tile of tile A
template <int M_, int NW_, int GT0, int GT1, int GT2, bool ONES_>
struct Cons {
    static constexpr int M = M_, NW = NW_, NT = 3 * NW_ + (ONES_ ? 1 : 0);
    static constexpr bool ONES = ONES_;
    static constexpr int gt(int w) { return w == 0 ? GT0 : w == 1 ? GT1 : GT2; }
    static constexpr int slot(int w, int dx) { return M_ == 0 ? 3 * w + dx : WR_NA + 3 * gt(w) + dx; }
};

template <typename C>
__device__ __forceinline__ void run_consumer(const WrArgs& a, unsigned* lds, int wave_id, int lane, int t0, int P, int iters, f32x4 (&acc)[10]) {
    const int k = (wave_id >> 1) & 1;
    const int j = lane & 15, kq = lane >> 4;
    // window offsets per phase (the row slot of a lane depends on its dy and wraps in the ring of 6)
    unsigned offx[3][C::NW];
#pragma unroll
    for (int w = 0; w < C::NW; ++w) {
        int g = 16 * C::gt(w) + j;
        g = g < 99 ? g : 99;
        const int dy = g == 99 ? 0 : g / 33, ci = g == 99 ? 33 : g - dy * 33;
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int rs = (2 * ((ph + 2) % 3) + k + dy) % 6;
            offx[ph][w] = (unsigned)(rs * XROW + ci * XP + 4 * kq);
        }
    }
    const unsigned offg = (unsigned)(XRING + k * GROW + (16 * C::M + j) * GP + 4 * kq);
    const unsigned offe = (unsigned)(XRING + GRING + k * EROW + C::M * 64 + lane);
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Sched cs;
    cs.init(a, t0);
    auto work = [&](auto phc) {
        constexpr int PH = decltype(phc)::value;
        Frag3 am, a0, ap;                      // gradient pixels x - 1 (tap dx = 2), x (dx = 1), x + 1 (dx = 0)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const u32x4 d = *reinterpret_cast<const u32x4*>(lds + offg + PH * 2 * GROW + s * GSL);
            const unsigned e = lds[offe + PH * 2 * EROW + s * ESL];
            a0.s[s] = d;
            ap.s[s][0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
            ap.s[s][1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
            ap.s[s][2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
            ap.s[s][3] = __builtin_amdgcn_alignbit(e, d[3], 16);
            am.s[s][0] = __builtin_amdgcn_alignbit(d[0], e, 16);
            am.s[s][1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
            am.s[s][2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
            am.s[s][3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
        }
#ifdef WR_NO_MFMA                 // (the kernel without its consumers' LDS reads and MFMAs)
        if (a.N > 0) return;
#endif
#pragma unroll
        for (int w = 0; w < C::NW; ++w) {
            Frag3 b;
#pragma unroll
            for (int s = 0; s < 3; ++s) b.s[s] = *reinterpret_cast<const u32x4*>(lds + offx[PH][w] + s * XSL);
            acc[3 * w + 0] = mfma_x3(ap, b, acc[3 * w + 0]);
            acc[3 * w + 1] = mfma_x3(a0, b, acc[3 * w + 1]);
            acc[3 * w + 2] = mfma_x3(am, b, acc[3 * w + 2]);
        }
        if constexpr (C::ONES) {               // bias of tile A: the aligned fragment against ones (slices 1, 2 of one are zero)
            const u32x4 one = (u32x4){ONE_PAIR, ONE_PAIR, ONE_PAIR, ONE_PAIR};
            f32x4 c = acc[9];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0.s[2]), __builtin_bit_cast(bf16x8, one), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0.s[1]), __builtin_bit_cast(bf16x8, one), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0.s[0]), __builtin_bit_cast(bf16x8, one), c, 0, 0, 0);
            acc[9] = c;
        }
    };
    // position p computes what the splitters committed at position p - 1 (ring slot (p - 1) % 3); 3 * iters >= P + 1 positions
    WrProf prof;
    prof.begin();
    for (int p = 0; p < 3 * iters; p += 3) {
        if (p >= 1 && p <= P && !cs.pre) work(std::integral_constant<int, 2>{});
        if (p >= 1) cs.advance(a);
        WR_BARRIER(prof);
        if (p + 1 <= P && !cs.pre) work(std::integral_constant<int, 0>{});
        cs.advance(a);
        WR_BARRIER(prof);
        if (p + 2 <= P && !cs.pre) work(std::integral_constant<int, 1>{});
        cs.advance(a);
        WR_BARRIER(prof);
    }
    prof.flush(a.prof, wave_id, lane);
}

typedef Cons<0, 3, 0, 2, 4, true> ConsA;
typedef Cons<1, 2, 0, 1, 0, false> ConsB;
typedef Cons<1, 3, 2, 3, 4, false> ConsC;
typedef Cons<1, 2, 5, 6, 0, false> ConsD;

template <typename C>
__device__ __forceinline__ void reduce_pair(const WrArgs& a, unsigned* lds, int k, int lane, const f32x4 (&acc)[10]) {
    float* sc = reinterpret_cast<float*>(lds);
    const int j = lane & 15, kq = lane >> 4;
    auto slot_of = [](int t) { return t == 9 ? 9 : C::slot(t / 3, t % 3); };
    if (k == 1) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
            if (t < 3 * C::NW || (C::ONES && t == 9))
#pragma unroll
                for (int q = 0; q < 4; ++q) sc[slot_of(t) * 256 + (kq * 4 + q) * 16 + j] = acc[t][q];
    }
    __syncthreads();
    if (k == 0) {
        float* part = a.partials + (size_t)blockIdx.x * WR_WPART;
#pragma unroll
        for (int t = 0; t < 10; ++t)
            if (t < 3 * C::NW || (C::ONES && t == 9))
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = slot_of(t) * 256 + (kq * 4 + q) * 16 + j;
                    part[idx] = acc[t][q] + sc[idx];
                }
    }
}

__global__ __launch_bounds__(WR_THREADS) void gen_wgrad_rs_kernel(WrArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned lds[WR_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x;
    const int t0 = (int)((long)b * a.T / a.groups), t1 = (int)((long)(b + 1) * a.T / a.groups);
    int P;                                                                 // steps + one PRE per segment
    {
        int n_, b_, s_, o_;
        P = __builtin_amdgcn_readfirstlane((t1 - t0) + Sched::locate(a, t1 - 1, n_, b_, s_, o_) - Sched::locate(a, t0, n_, b_, s_, o_) + 1);
    }
    // constant planes: ones (X plane 33), zero gradient rows 30, 31 and their edge dwords
    for (int i = threadIdx.x; i < 6 * 3 * XP; i += WR_THREADS) {
        const int row = i / (3 * XP), s = (i / XP) % 3, d = i % XP;
        lds[row * XROW + s * XSL + 33 * XP + d] = s == 0 ? ONE_PAIR : 0u;
    }
    for (int i = threadIdx.x; i < 6 * 3 * 2 * GP; i += WR_THREADS) {
        const int row = i / (3 * 2 * GP), s = (i / (2 * GP)) % 3, d = i % (2 * GP);
        lds[XRING + row * GROW + s * GSL + 30 * GP + d] = 0u;
    }
    for (int i = threadIdx.x; i < 6 * 3 * 8; i += WR_THREADS) {
        const int rs = i / 8, kq = (i >> 1) & 3, r = 14 + (i & 1);
        lds[XRING + GRING + rs * ESL + 64 + kq * 16 + r] = 0u;
    }
    __syncthreads();
    f32x4 acc[10];
    const int k = (wave >> 1) & 1;
    const int iters = (P + 3) / 3;                                         // positions 0 .. P in whole triples
    if (wave >= 8 + 2) run_splitter<2>(a, lds, wave - 8, lane, t0, iters);
    else if (wave >= 8) run_splitter<3>(a, lds, wave - 8, lane, t0, iters);   // (tasks 1,024 .. 1,127)
    else if (wave == 0 || wave == 2) run_consumer<ConsA>(a, lds, wave, lane, t0, P, iters, acc);
    else if (wave == 1 || wave == 3) run_consumer<ConsC>(a, lds, wave, lane, t0, P, iters, acc);
    else if (wave == 4 || wave == 6) run_consumer<ConsB>(a, lds, wave, lane, t0, P, iters, acc);
    else run_consumer<ConsD>(a, lds, wave, lane, t0, P, iters, acc);
    __syncthreads();                                                       // rings are free: the X ring becomes the scratch
    if (wave == 0 || wave == 2) reduce_pair<ConsA>(a, lds, k, lane, acc);
    else if (wave == 1 || wave == 3) reduce_pair<ConsC>(a, lds, k, lane, acc);
    else if (wave == 4 || wave == 6) reduce_pair<ConsB>(a, lds, k, lane, acc);
    else if (wave < 8) reduce_pair<ConsD>(a, lds, k, lane, acc);
    else __syncthreads();
}

}  // namespace

#ifdef WR_PROF
unsigned long long* g_wr_prof = nullptr;
#endif
bool gen_wgrad_rs_supported(int H, int W) { return W % 4 == 0 && H >= 2 && (long)H * W * NFEAT * 4 < (1l << 31); }

int gen_wgrad_rs_groups(int N, int H, int W, int max_groups) {
    const long T = (long)N * ((W + SW - 1) / SW) * ((H + 1) / 2);
    const long g = fz_num_cus() < max_groups ? fz_num_cus() : max_groups;
    return (int)(T < g ? T : g);
}

int gen_wgrad_rs(const float* mv, const float* res, const float* feat, const float* gout, const float* gbuf, const float* zero,
                 float* partials, int N, int H, int W, int groups, hipStream_t s) {
    WrArgs a;
    a.mv = mv; a.res = res; a.feat = feat; a.gout = gout; a.gbuf = gbuf; a.zero = zero; a.partials = partials;
    a.N = N; a.H = H; a.W = W;
    a.nstr = (W + SW - 1) / SW;
    a.HS = (H + 1) / 2;
    const long T = (long)N * a.nstr * a.HS;
    if (T >= (1l << 31) || groups < 1 || groups > T) return fail(DMC_E_INVALID, "gen_wgrad_rs: bad shape");
    a.T = (int)T; a.groups = groups;
    a.BS = a.HS < WR_BAND ? a.HS : WR_BAND;
    a.nb = (a.HS + a.BS - 1) / a.BS;
    a.prof = nullptr;
#ifdef WR_PROF
    a.prof = g_wr_prof;
#endif
    gen_wgrad_rs_kernel<<<groups, WR_THREADS, 0, s>>>(a);
    return check_launch("gen_wgrad_rs");
}

}  // namespace dmc
