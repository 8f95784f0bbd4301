// Weight gradient of EstimatorDenseNetTiny, row-sliding form (autograd of /root/reference/code/dmcnet/model.py:187-194; the GEMM
// and its bf16x3 arithmetic are those of gen_bwd_weight_pc_kernel<3> in gen_tiny.hip):
//
//   dW[(k, co)][(ci, dy, dx)] = sum over pixels of g_k[co][y][x] * in[ci][y + dy - 1][x + dx - 1]
//
// gen_bwd_weight_pc_kernel<3> splits every fp32 operand into its three bf16 slices IN the consumer waves: the window of a (ci, dy)
// column is split by the three tile rows that use it, 4.8 vector instructions per MFMA next to 16 matrix clocks, two waves per SIMD:
// 0.41 matrix-pipe busy.  Here every value is split ONCE, by eight splitter waves, into bf16-slice rings in LDS; the eight consumer
// waves read ready 16-byte fragments and issue MFMAs:
//
//   * a workgroup walks down a 32-column strip two image rows per step (position); the X ring holds 6 rows x 3 slices x 34 planes
//     (33 inputs + a plane of ones for the bias column), the G ring the same for 32 gradient planes (30 + two zero rows);
//   * the K slots of a tap are chosen so that only LEFT neighbours are needed: dx = 1 pairs aligned fragments; dx = 0 pairs the
//     aligned gradient fragment with the input fragment shifted one pixel left; dx = 2 the gradient fragment shifted one pixel left
//     with the aligned input fragment.  A shifted fragment = v_alignbit over the aligned 16 bytes + one edge halfword (pixel
//     8 kq - 1, the E rings).  The edge of kq = 0 is the last pixel of the strip to the left: the workgroup took that strip in its
//     previous segment and kept its last column (63 planes x 30 rows of 8-byte entries), so NO halo column is ever loaded --
//     an earlier form of this kernel fetched 16-byte halo chunks and with them whole 128-byte lines: 2.8 GB for 1.5;
//   * columns are dy-major (g = 33 dy + ci): a 16-lane column tile reads 16 consecutive planes of ONE row -- conflict-free
//     ds_read_b128 at a plane pitch of 24 dwords (tools/ubench/wgrad_lds_banks.py);
//   * consumer wave = (row of the pair, accumulator tiles): tile A (layers 0, 1: inputs < 13 -> column tiles gt = 0, 2, 4) +
//     its bias | tile B gt 0, 1 | tile B gt 2, 3, 4 | tile B gt 5, 6 -- 93 / 90 / 93 / 90 MFMAs per step on the four SIMDs;
//   * the splitters load three positions ahead into registers, split (18 vector instructions per 4 pixels) and write b64 pairs;
//     one barrier per position; they run at priority 2 (they are the pole of a position, the consumers' MFMAs fill in);
//   * work = segments (frame, band of <= 14 row pairs, strip), strips of a band consecutive; each workgroup takes a contiguous range
//     of them.  A range that starts at a strip > 0 reads that strip's left column from memory once (prologue).
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
#ifndef WR_PRIO
#define WR_PRIO 2
#endif
#ifndef WR_BAND
#define WR_BAND 14                                                      // row pairs per band
#endif
// rings, in dwords (a dword = two horizontally adjacent pixels of one bf16 slice)
constexpr int XP = 24, XPL = 34, XSL = XPL * XP, XROW = 3 * XSL + 8, XRING = 6 * XROW;      // 16 dwords used per plane row
constexpr int GP = XP, GSL = XSL, GROW = XROW, GRING = XRING;           // the gradient ring: the same geometry (32 of the 34 planes used)
// edge rings: 8 bytes per (row, plane, kq) = the three slices' halfwords of pixel 8 kq - 1 (+ pad): one b64 write, one b64 read.
// Gradient planes: [M tile][kq][row of the tile] (the lane order of an A fragment); input planes: [kq][plane] (the 16 lanes of a
// column tile = 16 consecutive planes)
constexpr int EROW = 2 * 192, ERING = 6 * EROW;
// first entry of the input planes' kq block: 0, 48, 104, 152 -- a read (lanes of kq 0, 1 or 2, 3 together) wants the two blocks 32
// dwords apart mod 64, a write (the four kq of one plane together) wants them on different banks mod 32: 2-way at best
__host__ __device__ constexpr int ex_kq(int kq) { return kq * 48 + (kq >> 1) * 8; }
constexpr int EG0 = XRING + GRING, EX0 = EG0 + ERING;
// last column of the strip to the left, the same 8-byte entries: [input / gradient][34 planes][32 rows of the segment]
constexpr int CROWS = 32, CPITCH = 33, CKIND = 2 * XPL * CPITCH;       // (dwords; 33 entries per plane: planes two apart on different banks)
constexpr int C0 = EX0 + ERING;
constexpr int WR_LDS = C0 + 2 * CKIND;
static_assert(WR_LDS * 4 <= 160 * 1024, "LDS");
static_assert(WR_WPART <= XRING, "the final reduction reuses the X ring");
static_assert(2 * WR_BAND + 2 <= CROWS, "rows of a segment in the left-column cache");
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
    int S, groups;                  // segments in all, workgroups
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

// ---- the positions of one workgroup: PRE STEP STEP ... per segment --------------------------------------------------------------
// Segment = (frame, band, strip), in this order: a workgroup finishes a band of <= BS row pairs in one strip, then takes the same
// band of the strip to its right.  PRE brings input rows 2 i - 1, 2 i of the segment's first row pair i; STEP i brings input rows
// 2 i + 1, 2 i + 2 and gradient rows 2 i, 2 i + 1.
struct Sched {
    int seg, n, band, strip, i, end, pos;   // segment, its frame / band / strip, row pair, first row pair past the segment, position in it (PRE = 0)
    bool pre;
    __device__ __forceinline__ static int seg_len(const WrArgs& a, int band) { return band == a.nb - 1 ? a.HS - band * a.BS : a.BS; }
    __device__ __forceinline__ void init(const WrArgs& a, int s0) {
        seg = s0;
        n = __builtin_amdgcn_readfirstlane(s0 / (a.nb * a.nstr));
        const int r = s0 - n * a.nb * a.nstr;
        band = __builtin_amdgcn_readfirstlane(r / a.nstr);
        strip = r - band * a.nstr;
        i = band * a.BS; end = i + seg_len(a, band);
        pre = true; pos = 0;
    }
    __device__ __forceinline__ void next_segment(const WrArgs& a) {
        ++seg;
        if (++strip == a.nstr) {
            strip = 0;
            if (++band == a.nb) { band = 0; ++n; }
        }
        i = band * a.BS; end = i + seg_len(a, band);
        pre = true; pos = 0;
    }
    __device__ __forceinline__ void advance(const WrArgs& a) {
        if (pre) { pre = false; pos = 1; return; }
        ++i; ++pos;
        if (i == end) next_segment(a);
    }
};

// ---- splitter ------------------------------------------------------------------------------------------------------------------
// A task = one 16-byte chunk (4 pixels of one plane and row) of a position: 2 rows x 33 input planes x 8 chunks, then 2 rows x 30
// gradient planes x 8 chunks = 1,008 tasks = two per lane of the 8 waves.  Everything that tells the arrays apart is per-lane data
// (64-bit plane base, bytes per frame), so the code has no branches on it; a chunk outside the image -- and every gradient chunk of
// a PRE position -- is read from a.zero: zeros arrive, no masking afterwards.  Per position and task: 6 vector instructions to
// issue the load (the validity of a lane is a scalar mask: four row classes x a column mask kept per strip), 18 to split, 3 b64
// writes of pairs + one 8-byte edge entry (odd chunks: the chunk's last pixel; chunk 7: the cached left neighbour to kq = 0 and
// its own last pixel into the cache).
constexpr int NSPLIT = 8, NS = 2, NTASK = 2 * 33 * 8 + 2 * 30 * 8;
static_assert(NTASK <= NSPLIT * 64 * NS, "tasks per position");
struct Task {
    unsigned long long base;   // its plane in frame 0
    unsigned fbytes;           // bytes per frame of its array
    unsigned voff;             // ((rowp + (input plane ? 1 : 0)) * W + 4 c) * 4: byte offset from (gradient row 0 of the pair, column tx0)
    unsigned lds;              // dword offset of the chunk's first pair in ring slot 0, slice 0
    int eh;                    // odd chunks: dword index of the edge entry this chunk supplies (ring slot 0), else -1
    int ch;                    // chunk 7: dword index of (its plane, row 0 of the pair) in the left-column cache, else -1
    int c4;                    // image column of the chunk relative to tx0
    int flags;                 // bit 0 rowp, 1 live, 4 gradient plane
};

__device__ __forceinline__ Task make_task(const WrArgs& a, int t) {
    Task s;
    const unsigned HW = (unsigned)a.H * (unsigned)a.W;
    const bool isx = t < 528, live = t < NTASK;
    const int u = isx ? t : live ? t - 528 : 0;
    // groups of 8 lanes = the chunks of one (row, plane); a 16-lane group of ds_write_b64 = two of them.  At the plane pitch of 24 dwords
    // neighbouring planes overlap in 8 of the 32 write banks (2-way: SQ_LDS_BANK_CONFLICT 0.43 of the LDS cycles); planes two apart
    // do not: within every four (row, plane) groups the middle two swap places
    const int planes = isx ? 33 : 30, ngroups = 2 * planes;
    int gi = u >> 3;
    const int c = u & 7;
    if ((gi | 3) < ngroups) gi = (gi & ~3) | (((gi & 1) << 1) | ((gi >> 1) & 1));
    const int rowp = gi / planes, P = gi - rowp * planes;             // P: plane in its ring
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
    s.c4 = live ? 4 * c : (1 << 28);                                    // (an idle lane: never inside the image)
    s.voff = (unsigned)(((rowp + (isx ? 1 : 0)) * a.W + 4 * c) * 4);
    s.lds = (unsigned)((isx ? 0 : XRING) + rowp * XROW + P * XP + 2 * c);
    s.eh = -1; s.ch = -1;
    if (live && (c & 1)) {
        const int kq = c == 7 ? 0 : (c + 1) / 2;                        // chunk 7 hands on the cached left neighbour of kq = 0
        s.eh = isx ? EX0 + rowp * EROW + 2 * (ex_kq(kq) + P) : EG0 + rowp * EROW + 2 * ((P >> 4) * 64 + kq * 16 + (P & 15));
        if (c == 7) s.ch = C0 + (isx ? 0 : CKIND) + 2 * (P * CPITCH + rowp);
    }
    s.flags = rowp | (live ? 2 : 0) | (isx ? 0 : 16);
    return s;
}

typedef unsigned long long mask_t;
// lane-wise select by a scalar mask (one v_cndmask with the mask in an SGPR pair)
__device__ __forceinline__ unsigned sel_mask(unsigned if0, unsigned if1, mask_t m) {
    unsigned r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(m));
    return r;
}

// the three bf16 slices of four pixels, as 16-bit values in the high halves
struct Split4 { unsigned u[3][4]; };
__device__ __forceinline__ Split4 split4(f32x4 v) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    Split4 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                       // (pairs: the subtractions become v_pk_add_f32)
        const f32x2 x = {v[2 * h], v[2 * h + 1]};
        r.u[0][2 * h] = __float_as_uint(x.x); r.u[0][2 * h + 1] = __float_as_uint(x.y);
        const f32x2 t0 = {__uint_as_float(r.u[0][2 * h] & 0xffff0000u), __uint_as_float(r.u[0][2 * h + 1] & 0xffff0000u)};
        const f32x2 r1 = x - t0;
        r.u[1][2 * h] = __float_as_uint(r1.x); r.u[1][2 * h + 1] = __float_as_uint(r1.y);
        const f32x2 t1 = {__uint_as_float(r.u[1][2 * h] & 0xffff0000u), __uint_as_float(r.u[1][2 * h + 1] & 0xffff0000u)};
        const f32x2 r2 = r1 - t1;
        r.u[2][2 * h] = __float_as_uint(r2.x); r.u[2][2 * h + 1] = __float_as_uint(r2.y);
    }
    return r;
}

__device__ __forceinline__ void run_splitter(const WrArgs& a, unsigned* lds, int sw, int lane, int s0, int iters) {
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
    Sched is, cs;
    is.init(a, s0); cs.init(a, s0);
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
    // ridx2 = 2 x (position in the segment) = the cache row of the pair's row 0; border: strip 0 (pixel tx0 - 1 is outside the
    // image).  Order: the cache reads of both tasks first (their latency passes under the splits), the pairs, then the edge
    // entries -- chunk 7's hand-over: cached left neighbour -> edge of kq = 0, own last pixel -> cache
    auto commit = [&](const Stage& g, auto ringc) {
        constexpr int RING = decltype(ringc)::value;
        const int ridx2 = 2 * cs.pos;
        const bool border = cs.strip == 0;
        u32x2 cached[NS], last[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cached[s] = (u32x2){0u, 0u};
            if (tk[s].ch >= 0) cached[s] = *reinterpret_cast<const u32x2*>(lds + tk[s].ch + 2 * ridx2);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const Split4 sp = split4(g.r[s]);
            if (tk[s].flags & 2) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    u32x2 w;
                    w.x = __builtin_amdgcn_perm(sp.u[q][1], sp.u[q][0], 0x07060302u);
                    w.y = __builtin_amdgcn_perm(sp.u[q][3], sp.u[q][2], 0x07060302u);
                    *reinterpret_cast<u32x2*>(lds + tk[s].lds + RING * 2 * XROW + q * XSL) = w;
                }
            }
            last[s].x = __builtin_amdgcn_perm(sp.u[1][3], sp.u[0][3], 0x07060302u);   // the last pixel's slices 0 | 1
            last[s].y = sp.u[2][3] >> 16;                                              // slice 2
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (tk[s].eh >= 0) {
                const bool c7 = tk[s].ch >= 0;
                u32x2 e = last[s];
                if (c7) e = border ? (u32x2){0u, 0u} : cached[s];
                *reinterpret_cast<u32x2*>(lds + tk[s].eh + RING * 2 * EROW) = e;
                if (c7) *reinterpret_cast<u32x2*>(lds + tk[s].ch + 2 * ridx2) = last[s];
            }
        }
        cs.advance(a);
    };
    WrProf prof;
    // the splitters are the pole of a position (tools/ubench/gen_wgrad_time.hip -DWR_PROF): their vector instructions go first,
    // the consumers' MFMAs fill in
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

// fragment of the pixels one to the left: dword k = (pixel 2 k - 1, pixel 2 k) of the lane's eight; e's high half = the pixel before them
__device__ __forceinline__ u32x4 shift_left1(u32x4 d, unsigned e) {
    u32x4 r;
    r[0] = __builtin_amdgcn_alignbit(d[0], e, 16);
    r[1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
    r[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
    r[3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
    return r;
}

template <typename C>
__device__ __forceinline__ void run_consumer(const WrArgs& a, unsigned* lds, int wave_id, int lane, int s0, int P, int iters, f32x4 (&acc)[10]) {
    const int k = (wave_id >> 1) & 1;
    const int j = lane & 15, kq = lane >> 4;
    // window / edge offsets per phase (the row slot of a lane depends on its dy and wraps in the ring of 6)
    unsigned offx[3][C::NW], offex[3][C::NW];
#pragma unroll
    for (int w = 0; w < C::NW; ++w) {
        int g = 16 * C::gt(w) + j;
        g = g < 99 ? g : 99;
        const int dy = g == 99 ? 0 : g / 33, ci = g == 99 ? 33 : g - dy * 33;
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int rs = (2 * ((ph + 2) % 3) + k + dy) % 6;
            offx[ph][w] = (unsigned)(rs * XROW + ci * XP + 4 * kq);
            offex[ph][w] = (unsigned)(EX0 + rs * EROW + 2 * (ex_kq(kq) + ci));
        }
    }
    const unsigned offg = (unsigned)(XRING + k * GROW + (16 * C::M + j) * GP + 4 * kq);
    const unsigned offeg = (unsigned)(EG0 + k * EROW + 2 * (C::M * 64 + lane));
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Sched cs;
    cs.init(a, s0);
    auto work = [&](auto phc) {
        constexpr int PH = decltype(phc)::value;
        Frag3 a0, am;                          // gradient pixels x (taps dx = 0, 1) and x - 1 (dx = 2)
        {
            const u32x2 e = *reinterpret_cast<const u32x2*>(lds + offeg + PH * 2 * EROW);   // halfwords: slice 0 | 1, 2
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const u32x4 d = *reinterpret_cast<const u32x4*>(lds + offg + PH * 2 * GROW + s * GSL);
                a0.s[s] = d;
                am.s[s] = shift_left1(d, s == 0 ? e.x << 16 : s == 1 ? e.x : e.y << 16);
            }
        }
#ifdef WR_NO_MFMA                 // (the kernel without its consumers' window reads and MFMAs)
        if (a.N > 0) return;
#endif
#pragma unroll
        for (int w = 0; w < C::NW; ++w) {
            Frag3 b0, bm;                      // input pixels x (dx = 1, 2) and x - 1 (dx = 0)
            const u32x2 e = *reinterpret_cast<const u32x2*>(lds + offex[PH][w]);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const u32x4 d = *reinterpret_cast<const u32x4*>(lds + offx[PH][w] + s * XSL);
                b0.s[s] = d;
                bm.s[s] = shift_left1(d, s == 0 ? e.x << 16 : s == 1 ? e.x : e.y << 16);
            }
            acc[3 * w + 0] = mfma_x3(a0, bm, acc[3 * w + 0]);            // sum g[x] in[x - 1]
            acc[3 * w + 1] = mfma_x3(a0, b0, acc[3 * w + 1]);            // sum g[x] in[x]
            acc[3 * w + 2] = mfma_x3(am, b0, acc[3 * w + 2]);            // sum g[x - 1] in[x]
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
#ifdef WR_CPRIO                   // (harness: a static priority for the second-dispatched consumer waves)
    if (wave_id >= 4) __builtin_amdgcn_s_setprio(WR_CPRIO);
#endif
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

// the left column of a range's first segment (strip > 0), from memory: pixel tx0 - 1 of the 63 planes x the rows of the segment
__device__ __forceinline__ void cold_fill(const WrArgs& a, unsigned* lds, const Sched& sg) {
    const size_t HW = (size_t)a.H * a.W;
    const int col = sg.strip * SW - 1;
    for (int idx = threadIdx.x; idx < 63 * CROWS; idx += WR_THREADS) {
        const int Pl = idx / CROWS, ridx = idx - Pl * CROWS;
        const bool isx = Pl < 33;
        const int P = isx ? Pl : Pl - 33;
        const int row = 2 * sg.i + ridx - (isx ? 1 : 2);                // (input rows count from 2 i - 1, gradient rows from 2 i - 2)
        const float* pl = isx ? (P < 2 ? a.mv + ((size_t)sg.n * 2 + P) * HW : P < 5 ? a.res + ((size_t)sg.n * 3 + (P - 2)) * HW
                                                                                   : a.feat + ((size_t)sg.n * NFEAT + (P - 5)) * HW)
                              : (P < NFEAT ? a.gbuf + ((size_t)sg.n * NFEAT + P) * HW : a.gout + ((size_t)sg.n * 2 + (P - NFEAT)) * HW);
        const float v = row >= 0 && row < a.H ? pl[(size_t)row * a.W + col] : 0.f;
        const unsigned u0 = __float_as_uint(v);
        const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = __float_as_uint(r1);
        const unsigned u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
        const int e = C0 + (isx ? 0 : CKIND) + 2 * (P * CPITCH + ridx);
        lds[e] = (u0 >> 16) | (u1 & 0xffff0000u);
        lds[e + 1] = u2 >> 16;
    }
}

__global__ __launch_bounds__(WR_THREADS) void gen_wgrad_rs_kernel(WrArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned lds[WR_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x;
    const int s0 = (int)((long)b * a.S / a.groups), s1 = (int)((long)(b + 1) * a.S / a.groups);
    Sched sg;
    sg.init(a, s0);
    const Sched first = sg;
    int P = 0;                                                             // positions: one PRE + the steps of every segment
    for (; sg.seg < s1; sg.next_segment(a)) P += 1 + (sg.end - sg.i);
    // edge rings and left-column cache: zero (gradient rows 30, 31 and the ones plane never get an edge); constant planes: ones
    // (X plane 33), zero gradient planes 30, 31
    for (int i = EG0 + threadIdx.x; i < WR_LDS; i += WR_THREADS) lds[i] = 0u;
    for (int i = threadIdx.x; i < 6 * 3 * XP; i += WR_THREADS) {
        const int row = i / (3 * XP), s = (i / XP) % 3, d = i % XP;
        lds[row * XROW + s * XSL + 33 * XP + d] = s == 0 ? ONE_PAIR : 0u;
    }
    for (int i = threadIdx.x; i < 6 * 3 * 2 * GP; i += WR_THREADS) {
        const int row = i / (3 * 2 * GP), s = (i / (2 * GP)) % 3, d = i % (2 * GP);
        lds[XRING + row * GROW + s * GSL + 30 * GP + d] = 0u;
    }
    __syncthreads();
    if (first.strip > 0) cold_fill(a, lds, first);
    __syncthreads();
    f32x4 acc[10];
    const int k = (wave >> 1) & 1;
    const int iters = (P + 3) / 3;                                         // positions 0 .. P in whole triples
    if (wave >= 8) run_splitter(a, lds, wave - 8, lane, s0, iters);
    else if (wave == 0 || wave == 2) run_consumer<ConsA>(a, lds, wave, lane, s0, P, iters, acc);
    else if (wave == 1 || wave == 3) run_consumer<ConsC>(a, lds, wave, lane, s0, P, iters, acc);
    else if (wave == 4 || wave == 6) run_consumer<ConsB>(a, lds, wave, lane, s0, P, iters, acc);
    else run_consumer<ConsD>(a, lds, wave, lane, s0, P, iters, acc);
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

namespace {
struct Geo { int nstr, HS, BS, nb; long S; };
Geo geo_of(int N, int H, int W) {
    Geo g;
    g.nstr = (W + SW - 1) / SW;
    g.HS = (H + 1) / 2;
    g.BS = g.HS < WR_BAND ? g.HS : WR_BAND;
    g.nb = (g.HS + g.BS - 1) / g.BS;
    g.S = (long)N * g.nb * g.nstr;
    return g;
}
}  // namespace

int gen_wgrad_rs_groups(int N, int H, int W, int max_groups) {
    const long S = geo_of(N, H, W).S;
    const long g = fz_num_cus() < max_groups ? fz_num_cus() : max_groups;
    return (int)(S < g ? S : g);
}

int gen_wgrad_rs(const float* mv, const float* res, const float* feat, const float* gout, const float* gbuf, const float* zero,
                 float* partials, int N, int H, int W, int groups, hipStream_t s) {
    WrArgs a;
    a.mv = mv; a.res = res; a.feat = feat; a.gout = gout; a.gbuf = gbuf; a.zero = zero; a.partials = partials;
    a.N = N; a.H = H; a.W = W;
    const Geo g = geo_of(N, H, W);
    a.nstr = g.nstr; a.HS = g.HS; a.BS = g.BS; a.nb = g.nb;
    if (g.S >= (1l << 31) || groups < 1 || groups > g.S) return fail(DMC_E_INVALID, "gen_wgrad_rs: bad shape");
    a.S = (int)g.S; a.groups = groups;
    a.prof = nullptr;
#ifdef WR_PROF
    a.prof = g_wr_prof;
#endif
    gen_wgrad_rs_kernel<<<groups, WR_THREADS, 0, s>>>(a);
    return check_launch("gen_wgrad_rs");
}

}  // namespace dmc
