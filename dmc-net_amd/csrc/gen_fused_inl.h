// Shared pieces of the fused generator kernels (gen_fused.hip: forward; gen_fused_bwd.hip: data gradient): strip geometry,
// pixel halves, scalar-base global accesses, the step barrier.  Included inside an anonymous namespace user.
#pragma once
#include "dmc_common.h"

namespace dmc { namespace fz {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FZ_RS = 120;                     // floats per LDS row (>= strip width + 2 zero columns)
constexpr int FZ_MAXSW = 118;                  // widest strip, halo included
constexpr int FZ_HALO = 6;                     // columns recomputed on the interior side of a strip (one per layer)
constexpr int FZ_WAVES = 12, FZ_THREADS = FZ_WAVES * 64;
// one strip of one frame, as a workgroup sees it
struct Strip {
    int n;                // frame
    int c0;               // image column of LDS column 0
    int v0, v1;           // LDS columns [v0, v1) are this strip's to store
    int w0, w1;           // image rows [w0, w1): the rows this work item sees (everything outside counts as zero padding)
    int s0, s1;           // image rows [s0, s1) are this item's to store (the window = these + the halo rows inside the image)
};

// global access = scalar plane base + 32-bit BYTE offset per lane (the form the saddr encodings take: no 64-bit vector adds).
// The plane base passes through readfirstlane: opaque to the reassociation that would otherwise fold it into per-lane
// 64-bit adds; the rebuilt pointer is given the global address space explicitly (a generic one would become flat_*).
typedef __attribute__((address_space(1))) float gfloat;
__device__ __forceinline__ gfloat* scalar_plane(const float* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned long long r = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)u);
    return (gfloat*)r;
}
// (the same pointer, scalar, for inline assembly operands)
__device__ __forceinline__ const float* scalar_plane_generic(const float* p) {
    const unsigned long long u = (unsigned long long)p;
    return (const float*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)u));
}
__device__ __forceinline__ void store_at(float* plane, unsigned byte_off, float v) {
    *(gfloat*)((__attribute__((address_space(1))) char*)scalar_plane(plane) + byte_off) = v;
}
__device__ __forceinline__ float load_at(const float* plane, unsigned byte_off) {
    return *(const gfloat*)((const __attribute__((address_space(1))) char*)scalar_plane(plane) + byte_off);
}

__device__ __forceinline__ float dpp_shr0(float cur) {      // lane i <- cur[i-1]; lane 0 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cur), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl0(float cur) {      // lane i <- cur[i+1]; lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cur), 0x130, 0xf, 0xf, false));
}
// end-of-step barrier: LDS traffic drained, global stores left in flight
__device__ __forceinline__ void step_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// what a wave knows about its pixel half
struct Half {
    int col;              // LDS column of this lane
    unsigned ucol;        // the same, unsigned: global accesses take a scalar row base + this 32-bit lane offset
    bool own;             // this lane's column is one the half produces
    bool store;           // ... and one the strip stores to HBM
};

// item -> (frame, strip, vertical part).  A frame's strips may be cut into a.vsplit row bands (more, shorter work items when
// frames x strips does not fill the CUs in whole rounds); a band recomputes a.vhalo rows of every layer above and below
// (where the image continues) exactly as a strip recomputes halo columns.
template <typename Args>
__device__ __forceinline__ Strip strip_of(const Args& a, int item) {
    Strip st;
    // (readfirstlane: the divisions run on the vector unit; everything derived from the item should be scalar again)
    const int fs = __builtin_amdgcn_readfirstlane(item / a.vsplit), vp = item - fs * a.vsplit;
    st.n = __builtin_amdgcn_readfirstlane(fs / a.nstrips);
    const int s = fs - st.n * a.nstrips;
    if (a.nstrips == 1) { st.c0 = 0; st.v0 = 0; st.v1 = a.W; }
    else if (s == 0) { st.c0 = 0; st.v0 = 0; st.v1 = a.m; }
    else { st.c0 = a.W - a.sw; st.v0 = a.sw - (a.W - a.m); st.v1 = a.sw; }
    const int rp = (a.H + a.vsplit - 1) / a.vsplit;
    st.s0 = vp * rp;
    st.s1 = min(a.H, st.s0 + rp);
    st.w0 = max(0, st.s0 - a.vhalo);
    st.w1 = min(a.H, st.s1 + a.vhalo);
    return st;
}

// pixel half hf of a strip sw columns wide: half 0 = LDS columns [0, 64), produces [0, ha); half 1 = columns [sw - 62, sw + 2),
// produces [ha, sw) (columns sw, sw + 1 are never written: zeros).  A strip of <= 62 columns has no second half.
template <typename Args>
__device__ __forceinline__ Half half_of(const Args& a, const Strip& st, int hf, int lane) {
    const int ha = a.sw <= 62 ? a.sw : (a.sw + 1) / 2;
    Half h;
    h.col = hf == 0 ? lane : a.sw - 62 + lane;
    if (a.sw <= 62 && hf == 1) h.col = lane;                 // (idle half: reads valid LDS, produces nothing)
    h.ucol = (unsigned)h.col;
    h.own = hf == 0 ? h.col < ha : (a.sw > 62 && h.col >= ha && h.col < a.sw);
    h.store = h.own && h.col >= st.v0 && h.col < st.v1;
    return h;
}

inline int fz_num_cus_hw() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}
// the CUs a persistent grid fills (option grid_reserve_cus); buffer SIZES use fz_num_cus_hw()
inline int fz_num_cus() { return dmc::persistent_cus(fz_num_cus_hw()); }


// strip geometry of a launch: one strip up to FZ_MAXSW columns, else two strips with `halo` recomputed columns on the interior side
struct StripGeo { int nstrips, sw, m, vsplit; };
// vsplit: the number of row bands per strip that minimises (rounds of workgroups) x (steps of a band); `lag` = pipeline depth in steps
inline StripGeo strip_geo(int N, int H, int W, int halo, int lag) {
    StripGeo g;
    if (W <= FZ_MAXSW) { g.nstrips = 1; g.sw = W; g.m = W; }
    else { g.nstrips = 2; g.m = (W + 1) / 2; g.sw = g.m + halo; }
    long best = -1;
    g.vsplit = 1;
    for (int vs = 1; vs <= 4 && (vs == 1 || (H + vs - 1) / vs >= 4 * halo); ++vs) {
        const long items = (long)N * g.nstrips * vs, rounds = (items + fz_num_cus() - 1) / fz_num_cus();
        const int rows = (H + vs - 1) / vs + (vs > 1 ? 2 * halo : 0);
        const long cost = rounds * ((rows + lag + 1 + 2) / 3 * 3);
        if (best < 0 || cost < best) { best = cost; g.vsplit = vs; }
    }
    return g;
}

} }  // namespace dmc::fz
