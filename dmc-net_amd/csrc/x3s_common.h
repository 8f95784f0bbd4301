// Device helpers shared by the pre-split bf16x3 convolution kernels (conv_x3s.hip, conv_x3q.hip); gfx950 only.
#pragma once
#include "dmc_common.h"

namespace dmc {
namespace x3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) char* lds_cptr;

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// fp32 -> three bf16 slices (truncation split: s0 = upper 16 bits, s1 = upper 16 bits of the exact remainder, s2 = the rest)
__device__ __forceinline__ void split3(float v, unsigned& u0, unsigned& u1, unsigned& u2) {
    u0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
    u1 = __float_as_uint(r1);
    u2 = __float_as_uint(r1 - __uint_as_float(u1 & 0xffff0000u));
}
__device__ __forceinline__ unsigned pack_hi(unsigned hi_of_second, unsigned hi_of_first) {   // (upper half of b, upper half of a)
    return __builtin_amdgcn_perm(hi_of_second, hi_of_first, 0x07060302u);
}

constexpr unsigned OOB = 0x80000000u;          // a transfer offset beyond num_records: the lane's 16 bytes arrive as zeros

// 16 bytes per lane global -> LDS through a buffer descriptor (range-checked: offsets >= num_records write zeros).
// LDS destination = lds_byte_addr + 16 * lane.  srd / soff / lds_byte_addr must be wave-uniform; the s_nops cover the
// M0 -> LDS-DMA wait state and an SGPR written by v_readfirstlane just in front of the statement.
__device__ __forceinline__ void dma_buf16(const u32x4& srd, unsigned voff, unsigned soff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :: "v"(voff), "s"(srd), "s"(lds_byte_addr), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 make_srd(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 srd;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    srd[2] = 0x7fffffffu;
    srd[3] = 0x00020000u;
    return srd;
}

// ds_read_b64_tr_b16: a 16-lane group hands in the addresses of a [4 pixels][16 channels] block (lane L: pixel L / 4,
// channels 4 (L % 4) ..) and lane i receives channel i of the four pixels; two reads = eight consecutive k-values
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tr_read2(lds_cptr p0, lds_cptr p1, u32x4& f) {   // eight k-values = two transposed reads
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    f[0] = ua[0]; f[1] = ua[1]; f[2] = ub[0]; f[3] = ub[1];
}


// ---- v_mfma_f32_16x16x32_bf16 users (conv_small.hip, gen_x3.hip) -----------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// eight fp32 values -> three bf16x8 fragments (truncation split, as split3)
__device__ __forceinline__ void split8(const float4& lo, const float4& hi, u32x4& s0, u32x4& s1, u32x4& s2) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(v[e], u0[e], u1[e], u2[e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s0[e] = pack_hi(u0[2 * e + 1], u0[2 * e]);
        s1[e] = pack_hi(u1[2 * e + 1], u1[2 * e]);
        s2[e] = pack_hi(u2[2 * e + 1], u2[2 * e]);
    }
}

}  // namespace x3
}  // namespace dmc
