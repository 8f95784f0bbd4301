// MaxPool3dTFPadding of the I3D trunk on bf16 NDHWC tensors (BASELINE config 5).
//
// Replaces the reference's MaxPool3dTFPadding (code/dmcnet_I3D/network/i3d.py:406-418): ConstantPad3d with ZEROS to
// the TF-"SAME" extent followed by nn.MaxPool3d(kernel, stride, ceil_mode=True) -- the four down-sampling pools of
// the trunk (:482,:488,:494,:505) and the 3x3x3 stride-1 pool of every Mixed block's branch_3 (:441-443) -- and its
// autograd.  Pad and pool are one pass: a window position inside the padded extent but outside the volume
// contributes the value 0 (as the explicit zero padding does; the activations are post-ReLU so it never wins
// against a positive value), positions beyond the padded extent (ceil_mode) are skipped.  The scan order and the
// strict `>` comparison are nn.MaxPool3d's, so ties -- frequent among rectified zeros -- resolve to the same
// element and the gradient goes where the stock op sends it (nowhere, when a padding zero comes first).
//
// HBM-bound streaming kernels: a thread owns 8 channels (16 bytes) of four consecutive output columns (forward) or four
// consecutive input columns (backward) -- one column each in the generic kernels for strides > 2 or kernels wider than 3.  The forward stores the winning tap per output value as one byte (255 = a padding zero); the
// backward GATHERS: every input pixel visits the <= kd kh kw windows that contain it and adds dy where the stored
// tap is its own -- no atomics (the stock backward scatters with bf16 atomic adds: 0.7 ms per call, and rounds
// after every add), fp32 sums rounded once, deterministic.
#include "dmc_common.h"

using namespace dmc;

namespace {

typedef unsigned short bf16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Pool3dArgs {
    const bf16_t* x;       // [N][D][H][W][C]   (backward: unused)
    bf16_t* y;             // [N][OD][OH][OW][C] (backward: dy, read)
    unsigned char* code;   // [N][OD][OH][OW][C]
    bf16_t* dx;            // backward: [N][D][H][W][C]
    int N, D, H, W, C;
    int OD, OH, OW;
    int kd, kh, kw, sd, sh, sw;
    int fd, fh, fw;        // front padding
    int PD, PH, PW;        // padded extents
};

__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ unsigned f2bf(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

constexpr int P3_OWT = 4;                                  // output columns per thread (forward) / input columns (backward)
constexpr int P3_MAXCOL = (P3_OWT - 1) * 2 + 3;           // input columns a thread touches per (kz, ky): strides <= 2, kw <= 3

// Forward.  A thread owns 8 channels of P3_OWT consecutive output columns of one (n, od, oh) row: per (kz, ky) it loads the
// (P3_OWT - 1) sw + kw input columns those outputs share once (6 instead of 12 vectors for the 3x3x3 stride-1 pools of
// the Inception blocks) and scans them per output in nn.MaxPool3d's order (kz, ky, kx ascending, strict >).
template <int SW>                                          // the stride along W when it is 1 or 2 (column cache indexed at compile time); 0 = generic
__global__ __launch_bounds__(256) void pool3d_fwd_kernel(Pool3dArgs a) {
    const int C8 = a.C >> 3;
    const int OWG = (a.OW + P3_OWT - 1) / P3_OWT;
    const long total = (long)a.N * a.OD * a.OH * OWG * C8;
    constexpr bool wide = SW != 0;                         // the column cache below holds P3_MAXCOL vectors (kw <= 3)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long p = i / C8;
        const int owg = (int)(p % OWG); p /= OWG;
        const int oh = (int)(p % a.OH); p /= a.OH;
        const int od = (int)(p % a.OD);
        const int n = (int)(p / a.OD);
        const int ow0 = owg * P3_OWT;
        float best[P3_OWT][8];
        unsigned char cd[P3_OWT][8];
#pragma unroll
        for (int o = 0; o < P3_OWT; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) { best[o][e] = -INFINITY; cd[o][e] = 255; }
        const int ncol = wide ? (P3_OWT - 1) * SW + a.kw : 0;
        for (int kz = 0; kz < a.kd; ++kz)
            for (int ky = 0; ky < a.kh; ++ky) {
                const int z = od * a.sd + kz, yv = oh * a.sh + ky;
                if (z >= a.PD || yv >= a.PH) continue;                             // beyond the padded extent (ceil_mode)
                const int iz = z - a.fd, iy = yv - a.fh;
                const bool row_in = iz >= 0 && iz < a.D && iy >= 0 && iy < a.H;
                const bf16_t* rowp = a.x + (((long)n * a.D + (row_in ? iz : 0)) * a.H + (row_in ? iy : 0)) * a.W * a.C + 8 * c8;
                u32x4 col[P3_MAXCOL];
                if constexpr (wide) {
#pragma unroll
                    for (int q = 0; q < P3_MAXCOL; ++q) {
                        col[q] = u32x4{0u, 0u, 0u, 0u};
                        const int ix = ow0 * SW + q - a.fw;
                        if (q < ncol && row_in && ix >= 0 && ix < a.W) col[q] = *reinterpret_cast<const u32x4*>(rowp + (long)ix * a.C);
                    }
                }
#pragma unroll
                for (int o = 0; o < P3_OWT; ++o) {
                    if (ow0 + o >= a.OW) continue;
                    auto take = [&](int kx, const u32x4& v, bool in) {
                        const int tap = (kz * a.kh + ky) * a.kw + kx;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = bf2f((e & 1) ? (v[e >> 1] >> 16) : (v[e >> 1] & 0xffffu));
                            if (f > best[o][e] || f != f) { best[o][e] = f; cd[o][e] = in ? (unsigned char)tap : (unsigned char)255; }
                        }
                    };
                    if constexpr (wide) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int xv = (ow0 + o) * SW + kx, ix = xv - a.fw;
                            if (kx >= a.kw || xv >= a.PW) continue;
                            take(kx, col[o * SW + kx], row_in && ix >= 0 && ix < a.W);
                        }
                    } else {
                        for (int kx = 0; kx < a.kw; ++kx) {
                            const int xv = (ow0 + o) * a.sw + kx, ix = xv - a.fw;
                            if (xv >= a.PW) continue;
                            const bool in = row_in && ix >= 0 && ix < a.W;
                            u32x4 v = u32x4{0u, 0u, 0u, 0u};
                            if (in) v = *reinterpret_cast<const u32x4*>(rowp + (long)ix * a.C);
                            take(kx, v, in);
                        }
                    }
                }
            }
#pragma unroll
        for (int o = 0; o < P3_OWT; ++o) {
            if (ow0 + o >= a.OW) continue;
            u32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (__float_as_uint(best[o][2 * e]) >> 16) | (__float_as_uint(best[o][2 * e + 1]) & 0xffff0000u);
            const long off = ((((long)n * a.OD + od) * a.OH + oh) * a.OW + ow0 + o) * a.C + 8 * c8;
            *reinterpret_cast<u32x4*>(a.y + off) = ov;
            u32x2 cc;
            cc[0] = cd[o][0] | (cd[o][1] << 8) | (cd[o][2] << 16) | ((unsigned)cd[o][3] << 24);
            cc[1] = cd[o][4] | (cd[o][5] << 8) | (cd[o][6] << 16) | ((unsigned)cd[o][7] << 24);
            *reinterpret_cast<u32x2*>(a.code + off) = cc;
        }
    }
}

// Forward, key form (round 6; kw <= 3, stride 1 or 2 along W).  The scan above spends ~6 vector instructions per (tap,
// channel) on compare / select pairs (value and tap code) and is bound by them, not by memory (57 us for a 29 MB tensor).
// Here every candidate becomes ONE 32-bit key, (order-preserving 16-bit image of the bf16 value) << 16 | (255 - tap), and the
// window's winner is an unsigned max: the larger value wins, among equal values the EARLIER tap (nn.MaxPool3d's strict >
// in scan order), in 2 instructions per (tap, channel) plus 6 per loaded word for the image:
//   x >= +0:  x | 0x8000          x < 0 (sign bit set):  (0 - x) mod 2^16     (-0 and +0 share 0x8000; -inf -> 0x0080; +NaN on top)
// Padding positions inside the padded extent are candidates with the value +0; whether the winning tap was one of them
// (code 255: the gradient goes nowhere) is read from a per-output bit mask of the taps that lie in the volume.
// Deviation from the scan kernel: a NaN with the sign bit set orders below everything instead of winning (the hardware's
// default NaN, and torch's, is positive), and a winning -0 is returned as +0.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned p3k_image(unsigned v) {            // two bf16 -> two order-preserving 16-bit images
    const unsigned neg = __builtin_bit_cast(unsigned, (u16x2)(-__builtin_bit_cast(u16x2, v)));
    const unsigned m = ((v >> 15) & 0x00010001u) * 0xffffu;
    return (neg & m) | ((v | 0x80008000u) & ~m);
}
__device__ __forceinline__ unsigned p3k_value(unsigned k16) {          // image -> bf16 bits
    return (k16 & 0x8000u) ? (k16 & 0x7fffu) : ((0x10000u - k16) & 0xffffu);
}

template <int SW>
__global__ __launch_bounds__(256) void pool3d_fwd_key_kernel(Pool3dArgs a) {
    const int C8 = a.C >> 3;
    const int OWG = (a.OW + P3_OWT - 1) / P3_OWT;
    const long total = (long)a.N * a.OD * a.OH * OWG * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long p = i / C8;
        const int owg = (int)(p % OWG); p /= OWG;
        const int oh = (int)(p % a.OH); p /= a.OH;
        const int od = (int)(p % a.OD);
        const int n = (int)(p / a.OD);
        const int ow0 = owg * P3_OWT;
        unsigned best[P3_OWT][8], inm[P3_OWT];
#pragma unroll
        for (int o = 0; o < P3_OWT; ++o) {
            inm[o] = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) best[o][e] = 0;
        }
        const int ncol = (P3_OWT - 1) * SW + a.kw;
        for (int kz = 0; kz < a.kd; ++kz)
            for (int ky = 0; ky < a.kh; ++ky) {
                const int z = od * a.sd + kz, yv = oh * a.sh + ky;
                if (z >= a.PD || yv >= a.PH) continue;                             // beyond the padded extent (ceil_mode)
                const int iz = z - a.fd, iy = yv - a.fh;
                const bool row_in = iz >= 0 && iz < a.D && iy >= 0 && iy < a.H;
                const bf16_t* rowp = a.x + (((long)n * a.D + (row_in ? iz : 0)) * a.H + (row_in ? iy : 0)) * a.W * a.C + 8 * c8;
                u32x4 col[P3_MAXCOL];
                unsigned colin = 0;
#pragma unroll
                for (int q = 0; q < P3_MAXCOL; ++q) {
                    col[q] = u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};   // the image of +0 (padding)
                    const int ix = ow0 * SW + q - a.fw;
                    if (q < ncol && row_in && ix >= 0 && ix < a.W) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(rowp + (long)ix * a.C);
#pragma unroll
                        for (int e = 0; e < 4; ++e) col[q][e] = p3k_image(v[e]);
                        colin |= 1u << q;
                    }
                }
                const int tzy = (kz * a.kh + ky) * a.kw;
#pragma unroll
                for (int o = 0; o < P3_OWT; ++o) {
                    if (ow0 + o >= a.OW) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        if (kx >= a.kw || (ow0 + o) * SW + kx >= a.PW) continue;
                        const int tap = tzy + kx;
                        const unsigned T = 255u - (unsigned)tap;
                        const u32x4 v = col[o * SW + kx];
                        inm[o] |= ((colin >> (o * SW + kx)) & 1u) << tap;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned lo = (v[e] << 16) | T, hi = (v[e] & 0xffff0000u) | T;
                            best[o][2 * e] = best[o][2 * e] > lo ? best[o][2 * e] : lo;
                            best[o][2 * e + 1] = best[o][2 * e + 1] > hi ? best[o][2 * e + 1] : hi;
                        }
                    }
                }
            }
#pragma unroll
        for (int o = 0; o < P3_OWT; ++o) {
            if (ow0 + o >= a.OW) continue;
            u32x4 ov;
            unsigned cd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned tap = 255u - (best[o][e] & 0xffu);
                cd[e] = ((inm[o] >> tap) & 1u) ? tap : 255u;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = p3k_value(best[o][2 * e] >> 16) | (p3k_value(best[o][2 * e + 1] >> 16) << 16);
            const long off = ((((long)n * a.OD + od) * a.OH + oh) * a.OW + ow0 + o) * a.C + 8 * c8;
            *reinterpret_cast<u32x4*>(a.y + off) = ov;
            u32x2 cc;
            cc[0] = cd[0] | (cd[1] << 8) | (cd[2] << 16) | (cd[3] << 24);
            cc[1] = cd[4] | (cd[5] << 8) | (cd[6] << 16) | (cd[7] << 24);
            *reinterpret_cast<u32x2*>(a.code + off) = cc;
        }
    }
}

__device__ __forceinline__ int ceil_div_floor0(int a, int b) { return a <= 0 ? 0 : (a + b - 1) / b; }

__global__ __launch_bounds__(256) void pool3d_bwd_kernel(Pool3dArgs a) {
    const int C8 = a.C >> 3;
    const long total = (long)a.N * a.D * a.H * a.W * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long p = i / C8;
        const int ix = (int)(p % a.W); p /= a.W;
        const int iy = (int)(p % a.H); p /= a.H;
        const int iz = (int)(p % a.D);
        const int n = (int)(p / a.D);
        const int z = iz + a.fd, yv = iy + a.fh, xv = ix + a.fw;                 // padded coordinates
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        // windows od with od * sd <= z <= od * sd + kd - 1
        int od_hi = z / a.sd; if (od_hi > a.OD - 1) od_hi = a.OD - 1;
        int oh_hi = yv / a.sh; if (oh_hi > a.OH - 1) oh_hi = a.OH - 1;
        int ow_hi = xv / a.sw; if (ow_hi > a.OW - 1) ow_hi = a.OW - 1;
        for (int od = ceil_div_floor0(z - a.kd + 1, a.sd); od <= od_hi; ++od)
            for (int oh = ceil_div_floor0(yv - a.kh + 1, a.sh); oh <= oh_hi; ++oh)
                for (int ow = ceil_div_floor0(xv - a.kw + 1, a.sw); ow <= ow_hi; ++ow) {
                    const unsigned tap = (unsigned)(((z - od * a.sd) * a.kh + (yv - oh * a.sh)) * a.kw + (xv - ow * a.sw));
                    const long off = ((((long)n * a.OD + od) * a.OH + oh) * a.OW + ow) * a.C + 8 * c8;
                    const u32x2 cc = *reinterpret_cast<const u32x2*>(a.code + off);
                    const unsigned t4 = tap * 0x01010101u;
                    if (((cc[0] ^ t4) & 0xff) && ((cc[0] ^ t4) & 0xff00) && ((cc[0] ^ t4) & 0xff0000) && ((cc[0] ^ t4) & 0xff000000u) &&
                        ((cc[1] ^ t4) & 0xff) && ((cc[1] ^ t4) & 0xff00) && ((cc[1] ^ t4) & 0xff0000) && ((cc[1] ^ t4) & 0xff000000u))
                        continue;                                                 // none of the 8 channels points here
                    const u32x4 g = *reinterpret_cast<const u32x4*>(a.y + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned ce = (cc[e >> 2] >> (8 * (e & 3))) & 0xffu;
                        if (ce == tap) s[e] += bf2f((e & 1) ? (g[e >> 1] >> 16) : (g[e >> 1] & 0xffffu));
                    }
                }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(s[2 * e]) | (f2bf(s[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4*>(a.dx + ((((long)n * a.D + iz) * a.H + iy) * a.W + ix) * a.C + 8 * c8) = o;
    }
}

// Backward for strides 1 / 2 along W and kw <= 3: a thread owns 8 channels of P3_OWT consecutive INPUT columns of one
// (n, iz, iy) row; per window row pair (od, oh) it loads the (code, dy) vectors of the <= 6 window columns those inputs
// fall into once (12 loads for four single-column threads) and adds dy where the stored tap points at each of its columns.
template <int SW>
__global__ __launch_bounds__(256) void pool3d_bwd_wide_kernel(Pool3dArgs a) {
    constexpr int NWC = (P3_OWT + 1) / SW + 1;             // window columns touching P3_OWT consecutive inputs (kw <= 3): 6 / 3
    const int C8 = a.C >> 3;
    const int WG = (a.W + P3_OWT - 1) / P3_OWT;
    const long total = (long)a.N * a.D * a.H * WG * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long p = i / C8;
        const int wg = (int)(p % WG); p /= WG;
        const int iy = (int)(p % a.H); p /= a.H;
        const int iz = (int)(p % a.D);
        const int n = (int)(p / a.D);
        const int ix0 = wg * P3_OWT;
        const int z = iz + a.fd, yv = iy + a.fh, xv0 = ix0 + a.fw;                // padded coordinates
        float s[P3_OWT][8];
#pragma unroll
        for (int o = 0; o < P3_OWT; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[o][e] = 0.f;
        int od_hi = z / a.sd; if (od_hi > a.OD - 1) od_hi = a.OD - 1;
        int oh_hi = yv / a.sh; if (oh_hi > a.OH - 1) oh_hi = a.OH - 1;
        const int ow_lo = ceil_div_floor0(xv0 - a.kw + 1, SW);
        int ow_hi = (xv0 + P3_OWT - 1) / SW; if (ow_hi > a.OW - 1) ow_hi = a.OW - 1;
        for (int od = ceil_div_floor0(z - a.kd + 1, a.sd); od <= od_hi; ++od)
            for (int oh = ceil_div_floor0(yv - a.kh + 1, a.sh); oh <= oh_hi; ++oh) {
                const unsigned tzy = (unsigned)(((z - od * a.sd) * a.kh + (yv - oh * a.sh)) * a.kw);
                const long rowoff = (((long)n * a.OD + od) * a.OH + oh) * a.OW;
#pragma unroll
                for (int wc = 0; wc < NWC; ++wc) {
                    const int ow = ow_lo + wc;
                    if (ow > ow_hi) continue;
                    const long off = (rowoff + ow) * a.C + 8 * c8;
                    const u32x2 cc = *reinterpret_cast<const u32x2*>(a.code + off);
                    const u32x4 g = *reinterpret_cast<const u32x4*>(a.y + off);
#pragma unroll
                    for (int o = 0; o < P3_OWT; ++o) {
                        const int kx = xv0 + o - ow * SW;
                        if (kx < 0 || kx >= a.kw || ix0 + o >= a.W) continue;
                        const unsigned tap = tzy + (unsigned)kx;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const unsigned ce = (cc[e >> 2] >> (8 * (e & 3))) & 0xffu;
                            if (ce == tap) s[o][e] += bf2f((e & 1) ? (g[e >> 1] >> 16) : (g[e >> 1] & 0xffffu));
                        }
                    }
                }
            }
#pragma unroll
        for (int o = 0; o < P3_OWT; ++o) {
            if (ix0 + o >= a.W) continue;
            u32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = f2bf(s[o][2 * e]) | (f2bf(s[o][2 * e + 1]) << 16);
            *reinterpret_cast<u32x4*>(a.dx + ((((long)n * a.D + iz) * a.H + iy) * a.W + ix0 + o) * a.C + 8 * c8) = ov;
        }
    }
}

int out_extent(int L, int k, int s, int* front, int* padded) {
    const int total = k - s > 0 ? k - s : 0;
    *front = total / 2;
    *padded = L + total;
    int o = (*padded - k + s - 1) / s + 1;                 // ceil_mode
    if ((o - 1) * s >= *padded) --o;                       // the last window must start inside the (padded) input
    return o;
}

bool pool_args(Pool3dArgs& a, int N, int D, int H, int W, int C, int kd, int kh, int kw, int sd, int sh, int sw) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0) return false;
    if (kd < 1 || kh < 1 || kw < 1 || sd < 1 || sh < 1 || sw < 1 || kd * kh * kw > 254) return false;
    a.N = N; a.D = D; a.H = H; a.W = W; a.C = C;
    a.kd = kd; a.kh = kh; a.kw = kw; a.sd = sd; a.sh = sh; a.sw = sw;
    a.OD = out_extent(D, kd, sd, &a.fd, &a.PD);
    a.OH = out_extent(H, kh, sh, &a.fh, &a.PH);
    a.OW = out_extent(W, kw, sw, &a.fw, &a.PW);
    return a.OD > 0 && a.OH > 0 && a.OW > 0;
}

int grid_for(long threads) {
    const long b = (threads + 255) / 256;
    return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

// output extents of MaxPool3dTFPadding(kernel, stride) on a [D,H,W] volume; returns 0 if unsupported
int dmc_maxpool3d_tf_out_shape(int D, int H, int W, int C, int kd, int kh, int kw, int sd, int sh, int sw, int* od, int* oh,
                               int* ow) {
    Pool3dArgs a;
    if (!pool_args(a, 1, D, H, W, C, kd, kh, kw, sd, sh, sw)) return 0;
    if (od) *od = a.OD;
    if (oh) *oh = a.OH;
    if (ow) *ow = a.OW;
    return 1;
}

// y [N,OD,OH,OW,C] bf16, code [N,OD,OH,OW,C] bytes (the winning tap, for the backward)
int dmc_maxpool3d_tf_bf16_fwd(const void* x, void* y, void* code, int N, int D, int H, int W, int C, int kd, int kh, int kw,
                              int sd, int sh, int sw, dmc_stream_t stream) {
    Pool3dArgs a;
    if (!x || !y || !code) return fail(DMC_E_INVALID, "dmc_maxpool3d_tf_bf16_fwd: null pointer");
    if (!pool_args(a, N, D, H, W, C, kd, kh, kw, sd, sh, sw)) return fail(DMC_E_INVALID, "dmc_maxpool3d_tf_bf16_fwd: unsupported shape");
    a.x = (const bf16_t*)x; a.y = (bf16_t*)y; a.code = (unsigned char*)code; a.dx = nullptr;
    const int grid = grid_for((long)N * a.OD * a.OH * ((a.OW + P3_OWT - 1) / P3_OWT) * (C / 8));
    const bool keys = kd * kh * kw <= 27 && option(OPT_CONV_CFG) != 9;   // (the mask of in-volume taps has 32 bits; conv_cfg 9: the scan kernels, A/B)
    // measured (tools/pool3d_microbench.py, us, key / scan): 3x3x3 s1 @28^2 x192 79 / 96, @14^2 x480 39 / 46; 1x3x3 s2 @112^2 72 / 60,
    // 3x3x3 s2 @28^2 51 / 51 -- the key form serves the stride-1 pools (the nine Mixed blocks), the scan form the rest
    if (kw <= 3 && sw == 1 && keys) pool3d_fwd_key_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else if (kw <= 3 && sw == 2 && keys && option(OPT_CONV_CFG) == 10) pool3d_fwd_key_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(a);   // (A/B)
    else if (kw <= 3 && sw == 1) pool3d_fwd_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else if (kw <= 3 && sw == 2) pool3d_fwd_kernel<2><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    else pool3d_fwd_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    return check_launch("pool3d_fwd");
}

// dx [N,D,H,W,C] bf16 from dy [N,OD,OH,OW,C] bf16 and the forward's codes
int dmc_maxpool3d_tf_bf16_bwd(const void* dy, const void* code, void* dx, int N, int D, int H, int W, int C, int kd, int kh,
                              int kw, int sd, int sh, int sw, dmc_stream_t stream) {
    Pool3dArgs a;
    if (!dy || !dx || !code) return fail(DMC_E_INVALID, "dmc_maxpool3d_tf_bf16_bwd: null pointer");
    if (!pool_args(a, N, D, H, W, C, kd, kh, kw, sd, sh, sw)) return fail(DMC_E_INVALID, "dmc_maxpool3d_tf_bf16_bwd: unsupported shape");
    a.x = nullptr; a.y = (bf16_t*)const_cast<void*>(dy); a.code = (unsigned char*)const_cast<void*>(code); a.dx = (bf16_t*)dx;
    const int gridw = grid_for((long)N * D * H * ((W + P3_OWT - 1) / P3_OWT) * (C / 8));
    if (kw <= 3 && sw == 1) pool3d_bwd_wide_kernel<1><<<gridw, 256, 0, (hipStream_t)stream>>>(a);
    else if (kw <= 3 && sw == 2) pool3d_bwd_wide_kernel<2><<<gridw, 256, 0, (hipStream_t)stream>>>(a);
    else pool3d_bwd_kernel<<<grid_for((long)N * D * H * W * (C / 8)), 256, 0, (hipStream_t)stream>>>(a);
    return check_launch("pool3d_bwd");
}

}  // extern "C"
