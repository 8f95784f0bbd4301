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

template <typename Args>
__device__ __forceinline__ Strip strip_of(const Args& a, int item) {
    Strip st;
    // (readfirstlane: the division runs on the vector unit; everything derived from the frame index should be scalar again)
    st.n = __builtin_amdgcn_readfirstlane(item / a.nstrips);
    const int s = item - st.n * a.nstrips;
    if (a.nstrips == 1) { st.c0 = 0; st.v0 = 0; st.v1 = a.W; }
    else if (s == 0) { st.c0 = 0; st.v0 = 0; st.v1 = a.m; }
    else { st.c0 = a.W - a.sw; st.v0 = a.sw - (a.W - a.m); st.v1 = a.sw; }
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

inline int fz_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}


// strip geometry of a launch: one strip up to FZ_MAXSW columns, else two strips with `halo` recomputed columns on the interior side
struct StripGeo { int nstrips, sw, m; };
inline StripGeo strip_geo(int W, int halo) {
    StripGeo g;
    if (W <= FZ_MAXSW) { g.nstrips = 1; g.sw = W; g.m = W; }
    else { g.nstrips = 2; g.m = (W + 1) / 2; g.sw = g.m + halo; }
    return g;
}

} }  // namespace dmc::fz
