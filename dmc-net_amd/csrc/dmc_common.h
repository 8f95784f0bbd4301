// Shared host/device helpers for libdmcnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <atomic>

#include "dmcnet_hip.h"

namespace dmc {

// ---- error reporting (thread-local, no global mutable state shared between threads) --------
inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DMC_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DMC_OK;
}

// ---- dynamic LDS above 64 KB -----------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): one function-static per launch site
// remembers on which devices of this process it has been raised (a process that drives a second GPU would otherwise
// fail every > 64 KB launch there).  Thread-safe: a relaxed bit mask; setting the attribute twice is harmless.
struct LdsLimit {
    std::atomic<unsigned long long> raised{0};
    hipError_t raise(const void* kernel, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && ((raised.load(std::memory_order_relaxed) >> dev) & 1ull)) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && dev >= 0 && dev < 64) raised.fetch_or(1ull << dev, std::memory_order_relaxed);
        return e;
    }
};

// ---- kernel-selection options (dmc_set_option / dmc_get_option, storage in losses.hip) ----------
// The library never reads the environment.  These few integers select between kernel variants for
// A/B measurements; the defaults are the fastest measured path.  They are relaxed atomics read
// once per entry-point call -- the only process-wide state of the library.
enum Option { OPT_GEN_LAYER_PATH = 0, OPT_GEN_GATHER, OPT_GEN_FUSE45, OPT_GEN_WGRAD_PATH, OPT_GEN_FUSE_FWD,
              OPT_GEN_FUSE_BWD, OPT_GEN_FRAMES, OPT_CONV_PATH, OPT_CONV_CFG, OPT_CONV_ABLATE, OPT_GEN_ABLATE, OPT_CONV_ARITH, OPT_CONV3D_WGRAD, OPT_GEN_X3, OPT_GEN_WINO, OPT_GEN_STAGGER, OPT_GEN_FUSED, OPT_GRID_RESERVE_CUS, OPT_COUNT };
int option(Option o);
// CUs a PERSISTENT grid (one workgroup per CU walking a work list: gen_fused, the generator ring / gather / Winograd kernels,
// gen_wgrad_rs, conv3d_p3) may fill, out of `hw`: option "grid_reserve_cus" (default 0) leaves that many idle -- room for RCCL's
// channel kernels when gradients are exchanged while the backward pass runs; a multiple of 8 stays (one CU per XCD granularity).
inline int persistent_cus(int hw) {
    const int r = option(OPT_GRID_RESERVE_CUS);
    int n = hw - (r > 0 ? r : 0);
    n -= n % 8;
    return n < 8 ? (hw < 8 ? hw : 8) : n;
}
// Measurement-only options ("gen_ablate", "conv_ablate", "gen_stagger": parts of a kernel switched off, RESULTS WRONG) exist only
// in a -DDMC_MEASURE build (dmc-net_amd/build.py --measure -> libdmcnet_hip_measure.so, which tools/ load through DMC_HIP_LIB):
// the product library refuses to set them, reads them as 0, and DMC_ABL() folds every ablated path out of its kernels.
#ifdef DMC_MEASURE
#define DMC_ABL(x) (x)
constexpr bool MEASURE_BUILD = true;
#else
#define DMC_ABL(x) 0
constexpr bool MEASURE_BUILD = false;
#endif
constexpr bool measure_only(int o) { return o == OPT_GEN_ABLATE || o == OPT_CONV_ABLATE || o == OPT_GEN_STAGGER; }
// Kernel variants that LOST their A/B measurement are compiled into the -DDMC_MEASURE build only (round 6): the product library
// refuses the option values that select them -- the one-launch generator data gradient (gen_fused bit 1, gen_fused_bwd.hip),
// the generator layer variants (ii) / (iii)-on-layer-3 / all-three-stage (gen_layer_path 3, 4, 5), the weight-gradient
// predecessors (gen_wgrad_path 0 .. 3).
constexpr bool product_value(int o, int v) {
    return o == OPT_GEN_FUSED ? (v == 0 || v == 1) : o == OPT_GEN_LAYER_PATH ? (v == 0 || v == 1)
         : o == OPT_GEN_WGRAD_PATH ? (v == 4 || v == 5) : true;
}

// ---- EstimatorDenseNetTiny geometry (code/dmcnet/model.py:172-194) --------------------------
// Physical channel order used by every kernel: [mv0 mv1 r0 r1 r2 | y0(8) | y1(8) | y2(6) | y3(4)
// | y4(2)] -- features are APPENDED, so layer k reads physical channels [0, CIN[k]).  The
// reference PREPENDS (torch.cat((conv(x), x), 1)); the weight repack maps between the two.
constexpr int NL = 6;
__host__ __device__ constexpr int cin_of(int k) {
    return k == 0 ? 5 : k == 1 ? 13 : k == 2 ? 21 : k == 3 ? 27 : k == 4 ? 31 : 33;
}
__host__ __device__ constexpr int cout_of(int k) {
    return k == 0 ? 8 : k == 1 ? 8 : k == 2 ? 6 : k == 3 ? 4 : 2;
}
// first physical channel of y_k (k = 0..4); yoff(5) = 33
__host__ __device__ constexpr int yoff(int k) { return cin_of(k); }
constexpr int NFEAT = 28;   // y0..y4 channels kept in the `saved` / `gbuf` buffers
constexpr int NIN = 5;

// packed parameter block (floats):  WF | BF | WB
//   WF[k][p][tap][co]   forward weights, p = physical input channel           (4554)
//   BF[k][co]                                                                  (30)
//   WB[j][c][tap][cd]   j = 0..4, data-gradient weights BY OUTPUT GROUP: the gradient of feature
//                       group y_j (cd in [0, COUT[j])) gathers from every later layer's g:
//                       input channel c runs over [g_{j+1} .. g_4, g_5], taps flipped  (3204)
__host__ __device__ constexpr int wf_off(int k) {
    int o = 0;
    for (int i = 0; i < k; ++i) o += cin_of(i) * 9 * cout_of(i);
    return o;
}
constexpr int WF_TOTAL = wf_off(6);   // 4554
__host__ __device__ constexpr int bf_off(int k) {
    int o = WF_TOTAL;
    for (int i = 0; i < k; ++i) o += cout_of(i);
    return o;
}
constexpr int NPARAM = bf_off(6);     // 4584
// number of gradient channels feeding feature group j: all g_k with k > j (30 - yoff(j+1) + 5)
__host__ __device__ constexpr int gin_of(int j) { return 35 - yoff(j + 1); }
__host__ __device__ constexpr int wb_off(int j) {
    int o = NPARAM;
    for (int i = 0; i < j; ++i) o += gin_of(i) * 9 * cout_of(i);
    return o;
}
constexpr int PACKED_TOTAL = wb_off(5);   // 7788
constexpr int ZERO_PAD = 256;             // zero words (1 KB: one LDS-DMA row) kept behind the packed parameters
// row-splits of the per-channel BatchNorm reductions: partial sums live in scratch as [channel][BN_MAX_SPLIT][2] doubles
// (dmc_bn_act_scratch_bytes); producers that reduce statistics in their epilogue (stem.hip) write the same layout
constexpr int BN_MAX_SPLIT = 1024;

// logical (reference, prepend order) input-channel index of physical channel p in layer k
__host__ __device__ inline int logical_of(int k, int p) {
    if (p < NIN) return (cin_of(k) - NIN) + p;
    int j = 0;
    while (p >= yoff(j) + cout_of(j)) ++j;
    int l = 0;
    for (int i = j + 1; i < k; ++i) l += cout_of(i);
    return l + (p - yoff(j));
}

struct ParamPtrs {
    const float* w[NL];
    const float* b[NL];
};
struct GradPtrs {
    float* w[NL];
    float* b[NL];
};

}  // namespace dmc

// ---- LDS-DMA helpers shared by kernels outside gen_tiny.hip -----------------------------------
// global_load_lds_dwordx4 through inline assembly (the compiler then inserts no vmcnt(0) in front
// of unrelated ds_reads; completion is tracked by hand with s_waitcnt vmcnt).  Address = 64-bit
// scalar base + 32-bit per-lane byte offset; the LDS destination is lds_byte_addr + 16 * lane.
namespace dmc {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(lds_ptr_t)p; }
__device__ __forceinline__ void lds_dma16(unsigned long long sbase, unsigned voff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}
}  // namespace dmc

// ---- lane sums on the VALU's data-parallel primitives (DPP) -----------------------------------------------
// A butterfly of __shfl_xor is one ds_bpermute_b32 per dword and level -- an LDS-crossbar round trip each, waited for: the fp64
// BatchNorm sums of an accumulator tile (32 values x 5 levels x 2 dwords) cost more than a short convolution's main loop.
// Rotations within a row of 16 lanes (row_ror) and the broadcast of a row's lane 15 into the next row (row_bcast:15) are plain
// vector instructions.  The order of the additions is fixed (not the butterfly's).
namespace dmc {
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// every lane of a row of 16 (lanes 16 k .. 16 k + 15) gets the row's sum
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0x128, 0xf>(v);                           // row_ror:8
    v += dpp_f64<0x124, 0xf>(v);                           // row_ror:4
    v += dpp_f64<0x122, 0xf>(v);                           // row_ror:2
    v += dpp_f64<0x121, 0xf>(v);                           // row_ror:1
    return v;
}
// lanes 16 .. 31 (48 .. 63) get the sum of lanes 0 .. 31 (32 .. 63); the other lanes hold their row's sum
__device__ __forceinline__ double half32_sum_hi(double v) {
    v = row16_sum(v);
    v += dpp_f64<0x142, 0xa>(v);                           // row_bcast:15 into rows 1 and 3 (rows 0 and 2 add 0)
    return v;
}
}  // namespace dmc

// ---- bf16x3 slice tensors (conv_x3s.hip and their producers in bn_act.hip) -------------------------------
// An fp32 value is the exact sum of three bf16 values: s0 = its upper 16 bits, s1 = the upper 16 bits of the exact
// remainder, s2 = the second remainder.  A slice tensor of a [M][C] fp32 activation is bf16 [3][C / 16][M][16].
namespace dmc {
typedef unsigned x3s_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void x3s_split3(float v, unsigned& u0, unsigned& u1, unsigned& u2) {
    u0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
    u1 = __float_as_uint(r1);
    u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
}
// the three slices of 8 consecutive channels (8 g .. 8 g + 7) of pixel m -> xs
__device__ __forceinline__ void x3s_store8(unsigned short* __restrict__ xs, const float (&v)[8], size_t M, int nchunk, size_t m, int g) {
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x3s_split3(v[e], u0[e], u1[e], u2[e]);
    x3s_u32x4 s0, s1, s2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s0[e] = __builtin_amdgcn_perm(u0[2 * e + 1], u0[2 * e], 0x07060302u);
        s1[e] = __builtin_amdgcn_perm(u1[2 * e + 1], u1[2 * e], 0x07060302u);
        s2[e] = __builtin_amdgcn_perm(u2[2 * e + 1], u2[2 * e], 0x07060302u);
    }
    const size_t plane = M * 16;
    const size_t w = ((size_t)(g >> 1) * M + m) * 16 + (size_t)(g & 1) * 8;
    *reinterpret_cast<x3s_u32x4*>(xs + w) = s0;
    *reinterpret_cast<x3s_u32x4*>(xs + (size_t)nchunk * plane + w) = s1;
    *reinterpret_cast<x3s_u32x4*>(xs + (size_t)2 * nchunk * plane + w) = s2;
}
}  // namespace dmc
