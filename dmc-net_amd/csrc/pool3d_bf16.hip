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
// HBM-bound streaming kernels: a thread owns 8 channels (16 bytes) of one output pixel (forward) or one input pixel
// (backward).  The forward stores the winning tap per output value as one byte (255 = a padding zero); the
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

__global__ __launch_bounds__(256) void pool3d_fwd_kernel(Pool3dArgs a) {
    const int C8 = a.C >> 3;
    const long total = (long)a.N * a.OD * a.OH * a.OW * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long p = i / C8;
        const int ow = (int)(p % a.OW); p /= a.OW;
        const int oh = (int)(p % a.OH); p /= a.OH;
        const int od = (int)(p % a.OD);
        const int n = (int)(p / a.OD);
        float best[8];
        unsigned char cd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; cd[e] = 255; }
        int tap = 0;
        for (int kz = 0; kz < a.kd; ++kz)
            for (int ky = 0; ky < a.kh; ++ky)
                for (int kx = 0; kx < a.kw; ++kx, ++tap) {
                    const int z = od * a.sd + kz, yv = oh * a.sh + ky, xv = ow * a.sw + kx;
                    if (z >= a.PD || yv >= a.PH || xv >= a.PW) continue;        // beyond the padded extent (ceil_mode)
                    const int iz = z - a.fd, iy = yv - a.fh, ix = xv - a.fw;
                    const bool in = iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    u32x4 v = u32x4{0u, 0u, 0u, 0u};
                    if (in) v = *reinterpret_cast<const u32x4*>(a.x + ((((long)n * a.D + iz) * a.H + iy) * a.W + ix) * a.C + 8 * c8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = bf2f((e & 1) ? (v[e >> 1] >> 16) : (v[e >> 1] & 0xffffu));
                        if (f > best[e] || f != f) { best[e] = f; cd[e] = in ? (unsigned char)tap : (unsigned char)255; }
                    }
                }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__float_as_uint(best[2 * e]) >> 16) | (__float_as_uint(best[2 * e + 1]) & 0xffff0000u);
        const long off = ((((long)n * a.OD + od) * a.OH + oh) * a.OW + ow) * a.C + 8 * c8;
        *reinterpret_cast<u32x4*>(a.y + off) = o;
        u32x2 cc;
        cc[0] = cd[0] | (cd[1] << 8) | (cd[2] << 16) | ((unsigned)cd[3] << 24);
        cc[1] = cd[4] | (cd[5] << 8) | (cd[6] << 16) | ((unsigned)cd[7] << 24);
        *reinterpret_cast<u32x2*>(a.code + off) = cc;
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
    pool3d_fwd_kernel<<<grid_for((long)N * a.OD * a.OH * a.OW * (C / 8)), 256, 0, (hipStream_t)stream>>>(a);
    return check_launch("pool3d_fwd");
}

// dx [N,D,H,W,C] bf16 from dy [N,OD,OH,OW,C] bf16 and the forward's codes
int dmc_maxpool3d_tf_bf16_bwd(const void* dy, const void* code, void* dx, int N, int D, int H, int W, int C, int kd, int kh,
                              int kw, int sd, int sh, int sw, dmc_stream_t stream) {
    Pool3dArgs a;
    if (!dy || !dx || !code) return fail(DMC_E_INVALID, "dmc_maxpool3d_tf_bf16_bwd: null pointer");
    if (!pool_args(a, N, D, H, W, C, kd, kh, kw, sd, sh, sw)) return fail(DMC_E_INVALID, "dmc_maxpool3d_tf_bf16_bwd: unsupported shape");
    a.x = nullptr; a.y = (bf16_t*)const_cast<void*>(dy); a.code = (unsigned char*)const_cast<void*>(code); a.dx = (bf16_t*)dx;
    pool3d_bwd_kernel<<<grid_for((long)N * D * H * W * (C / 8)), 256, 0, (hipStream_t)stream>>>(a);
    return check_launch("pool3d_bwd");
}

}  // extern "C"
