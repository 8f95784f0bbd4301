// Forward of the I3D stem on the bf16 matrix cores (BASELINE config 5).
//
// Replaces the convolution of `conv3d_1a_7x7 = Unit3Dpy(in_channels, 64, (7, 7, 7), (2, 2, 2))`, built at
// code/dmcnet_I3D/network/i3d.py:480-481, run through Unit3Dpy.forward (:390-393: ConstantPad3d to the TF-"SAME"
// extent, front 2 / back 3 per dimension, then nn.Conv3d with padding 0) on the 2-channel DMC cue, as the reference
// runs it in 16-bit mixed precision: 106 GFLOP for 3 clips x 64 frames x 224^2, K = 2 x 343 = 686.
//
// With 2 input channels the contraction is arranged by (kz, ky): one MFMA k-block of 16 = the 7 taps kx x 2 channels
// of an input row segment (14 values) + 2 zero-weight slots -- 49 k-blocks, 87.5 % useful.  The cue is first
// converted to a zero-padded bf16 volume [n][T+5][H+5][Wp][2] (stem3d_prep_kernel; Wp a multiple of 4 pixels), in
// which the 16 values of a k-block for output pixel ox are 32 CONTIGUOUS bytes starting at pixel 2 ox of row
// (2 od + kz, 2 oh + ky): lane (pixel, half) loads its 16 bytes straight from global memory (the volume is 44 MB
// and stays in L2 / MALL), no LDS staging of activations.  The packed weights [49][64][16] (100 KB) are loaded into
// LDS once per workgroup; rows = the 64 output channels (two 32-row tiles), columns = 32 consecutive output pixels.
// Persistent workgroups of 8 waves walk the pixel tiles.  Epilogue: bf16 NDHWC stores and, per wave, the per-channel
// (sum, sum of squares) of the rounded outputs for the BatchNorm3d that follows (dmc_bn3d_bf16_fwd).
// The stem's data and weight gradients stay on MIOpen (dmc-net_amd/ops.py: _Stem3d).
#include "dmc_common.h"

using namespace dmc;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;

__device__ __forceinline__ unsigned f2bf(float v) {
    unsigned u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }

constexpr int S3_CO = 64, S3_K = 7, S3_KB = 49, S3_WAVES = 8;
constexpr int S3_WLDS = S3_KB * S3_CO * 16 * 2;           // 100,352 bytes

// x fp32 [N][2][T][H][W] -> xq [N][Tp][Hp][Wp] dwords, dword = (bf16 channel 0) | (bf16 channel 1) << 16, zero borders
// (one padded row (n, zz, yy) per workgroup iteration, threads along xx: the row's coordinates are wave-uniform 32-bit divisions;
// a flat 64-bit index per element cost three 64-bit divisions each -- 291 us for a 44 MB volume)
__global__ __launch_bounds__(256) void stem3d_prep_kernel(const float* __restrict__ x, unsigned* __restrict__ xq, int N, int T, int H,
                                                          int W, int Tp, int Hp, int Wp) {
    const int rows = N * Tp * Hp;
    const long plane = (long)T * H * W;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const int yy = r % Hp, q = r / Hp;
        const int zz = q % Tp, n = q / Tp;
        const int t = zz - 2, h = yy - 2;
        const bool row_in = t >= 0 && t < T && h >= 0 && h < H;
        const float* src = x + ((long)n * 2) * plane + ((long)t * H + h) * W;
        unsigned* dst = xq + (long)r * Wp;
        for (int xx = threadIdx.x; xx < Wp; xx += 256) {
            const int w = xx - 2;
            unsigned v = 0;
            if (row_in && w >= 0 && w < W) v = f2bf(src[w]) | (f2bf(src[w + plane]) << 16);
            dst[xx] = v;
        }
    }
}

// w fp32 [64][2][7][7][7] (contiguous) -> wp bf16 [49][64][16]: slot 2 kx + c, slots 14, 15 zero
__global__ __launch_bounds__(256) void stem3d_pack_w_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S3_KB * S3_CO * 16) return;
    const int j = i & 15, co = (i >> 4) % S3_CO, kb = i / (16 * S3_CO);
    unsigned v = 0;
    if (j < 14) {
        const int kx = j >> 1, c = j & 1;
        v = f2bf(w[((co * 2 + c) * 343) + kb * 7 + kx]);       // kb = kz * 7 + ky
    }
    wp[i] = (bf16_t)v;
}

struct Stem3dArgs {
    const unsigned* xq;    // [N][Tp][Hp][Wp] dwords
    const bf16_t* wp;      // [49][64][16]
    bf16_t* y;             // [N][OD][OH][OW][64]
    float* stat_part;      // [gridDim.x * 8][64][2] or null
    int N, OD, OH, OW, Tp, Hp, Wp, tiles_x;
};

__global__ __launch_bounds__(S3_WAVES * 64) void stem3d_fwd_kernel(Stem3dArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < S3_WLDS / 16; i += S3_WAVES * 64)
        reinterpret_cast<u32x4*>(wlds)[i] = reinterpret_cast<const u32x4*>(a.wp)[i];
    __syncthreads();

    const long ntiles = (long)a.N * a.OD * a.OH * a.tiles_x;
    const long stride = (long)gridDim.x * S3_WAVES;
    const int woff = l31 * 32 + half * 16;                 // this lane's A fragment inside a [64][16] slab (+ 1024 for tile 1)
    float s1[2][16], s2[2][16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1[ct][e] = 0.f; s2[ct][e] = 0.f; }

    for (long tile = (long)blockIdx.x * S3_WAVES + wave; tile < ntiles; tile += stride) {
        const int xt = (int)(tile % a.tiles_x);
        long r = tile / a.tiles_x;                          // (n * OD + od) * OH + oh
        const int oh = (int)(r % a.OH);
        const long nd = r / a.OH;
        const int od = (int)(nd % a.OD);
        const int n = (int)(nd / a.OD);
        const int ox = 32 * xt + l31;
        const int oxc = ox < a.OW ? ox : a.OW - 1;          // clipped lanes read a valid pixel and do not store
        // dword index of (row z = 2 od, y = 2 oh, pixel 2 ox + 4 half)
        const unsigned* base = a.xq + (((long)n * a.Tp + 2 * od) * a.Hp + 2 * oh) * a.Wp + 2 * oxc + 4 * half;
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
        // one kz plane (7 rows x 16 bytes per lane) in flight ahead of the 14 MFMAs of the current one; the loop is NOT
        // unrolled over kz: fully unrolled, the compiler hoisted all 98 loads and spilled (256 VGPRs + 1 KB of scratch)
        u32x4 cur[S3_K], nxt[S3_K];
        auto load_plane = [&](int kz, u32x4 (&xb)[S3_K]) {
#pragma unroll
            for (int ky = 0; ky < S3_K; ++ky) {
                const unsigned* p = base + ((long)kz * a.Hp + ky) * a.Wp;
                const u32x2 lo = *reinterpret_cast<const u32x2*>(p), hi = *reinterpret_cast<const u32x2*>(p + 2);
                xb[ky] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        };
        load_plane(0, cur);
#pragma unroll 1
        for (int kz = 0; kz < S3_K; ++kz) {
            if (kz + 1 < S3_K) load_plane(kz + 1, nxt);
#pragma unroll
            for (int ky = 0; ky < S3_K; ++ky) {
                const char* wk = wlds + (kz * S3_K + ky) * (S3_CO * 32) + woff;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const u32x4 wf = *reinterpret_cast<const u32x4*>(wk + ct * 1024);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, cur[ky]),
                                                                      acc[ct], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ky = 0; ky < S3_K; ++ky) cur[ky] = nxt[ky];
        }
        // lane holds pixel column l31, channels 32 ct + 8 g + 4 half + e in acc[ct][4 g + e]
        if (ox < a.OW) {
            bf16_t* dst = a.y + (r * a.OW + ox) * S3_CO;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned h[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] = f2bf(acc[ct][4 * g + e]);
                        const float rv = bf2f(h[e]);
                        s1[ct][4 * g + e] += rv; s2[ct][4 * g + e] += rv * rv;
                    }
                    *reinterpret_cast<u32x2*>(dst + 32 * ct + 8 * g + 4 * half) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
                }
        }
    }
    if (a.stat_part) {
        float* dst = a.stat_part + ((size_t)blockIdx.x * S3_WAVES + wave) * S3_CO * 2;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float d1 = s1[ct][e], d2 = s2[ct][e];
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
                if (l31 == 0) {
                    const int c = 32 * ct + 8 * (e >> 2) + 4 * half + (e & 3);
                    dst[2 * c] = d1; dst[2 * c + 1] = d2;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the stem: dw[co][c][kz][ky][kx] = sum over output pixels of dy[px][co] * x[2 od + kz][2 oh + ky][2 ow + kx][c],
// a GEMM over pixels with rows = the forward's k index (49 k-blocks x 16 slots = 784 rows, 25 tiles of 32), columns = the 64
// channels.  Both operands must hold 8 consecutive pixels per lane, so a 32-pixel row segment is transposed on its way into
// LDS: dy [32 px][64] -> dyT [64][32 px] (eight ds_write_b16 per 16-byte load) and, from the padded volume of the forward,
// xT [784][32 px]: every input dword (both channels of one input pixel) is written to the <= 4 (output pixel, tap kx) positions
// that use it -- the (k-block, input pixel) work items and their LDS targets are fixed per thread for the whole launch.  Four
// waves share the 25 row tiles (6-7 tiles x 2 column tiles of fp32 accumulators each); a workgroup walks a contiguous run of
// segments and writes one partial [800][64]; stem3d_wgrad_reduce_kernel sums the partials in fixed order (deterministic).
// ------------------------------------------------------------------------------------------
constexpr int S3_ROWS = 800;                               // 25 x 32 (rows 784 .. 799 stay zero)
constexpr int S3_ITEMS = S3_KB * 70;                       // (k-block, input pixel q = 0 .. 69) work items per segment
constexpr int S3_IPT = (S3_ITEMS + 255) / 256;             // 14 per thread

struct Stem3dWgArgs {
    const unsigned* xq;    // [N][Tp][Hp][Wp] dwords
    const bf16_t* dy;      // [N][OD][OH][OW][64]
    float* part;           // [gridDim.x][800][64]
    int N, OD, OH, OW, Tp, Hp, Wp, tiles_x;
    long nseg, per_wg;
};

__device__ __forceinline__ int s3_lds(int row, int px) { return row * 64 + ((((px >> 3) ^ ((row >> 2) & 3)) & 3) << 4) + (px & 7) * 2; }

__global__ __launch_bounds__(256) void stem3d_wgrad_kernel(Stem3dWgArgs a) {
    __shared__ __attribute__((aligned(1024))) char lds[(S3_ROWS + 64) * 64];   // xT [800][32 px] | dyT [64][32 px]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < (S3_ROWS + 64) * 4; i += 256) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0u, 0u, 0u, 0u};
    char* dyT = lds + S3_ROWS * 64;

    // this thread's work items: item = tid + 256 i -> (kb, q); input pixel q of the segment's row feeds output pixel
    // ow = (q - kx) / 2 for the taps kx = q & 1, (q & 1) + 2, ... with 0 <= ow < 32
    int it_kb[S3_IPT], it_q[S3_IPT];
#pragma unroll
    for (int i = 0; i < S3_IPT; ++i) {
        const int item = tid + 256 * i;
        it_kb[i] = item < S3_ITEMS ? item / 70 : -1;
        it_q[i] = item % 70;
    }
    const int dpx = tid >> 3, doct = tid & 7;              // dy staging: pixel, channel octet

    f32x16 acc[7][2];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int fa[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) fa[i] = (32 * (wave + 4 * i) + l31) * 64;       // row of tile wave + 4 i (tile 28 does not exist: i = 6 only for wave 0)
    const int swz = (l31 >> 2) & 3;

    long seg = (long)blockIdx.x * a.per_wg;
    long seg_end = seg + a.per_wg;
    if (seg_end > a.nseg) seg_end = a.nseg;
    __syncthreads();
#pragma unroll 1
    for (; seg < seg_end; ++seg) {
        const int xt = (int)(seg % a.tiles_x);
        const long r = seg / a.tiles_x;                    // (n * OD + od) * OH + oh
        const int oh = (int)(r % a.OH);
        const long nd = r / a.OH;
        const int od = (int)(nd % a.OD);
        const int n = (int)(nd / a.OD);
        const int ow0 = 32 * xt;
        // loads: dy chunk and the work items' dwords
        u32x4 dv = u32x4{0u, 0u, 0u, 0u};
        if (ow0 + dpx < a.OW) dv = *reinterpret_cast<const u32x4*>(a.dy + (r * a.OW + ow0 + dpx) * S3_CO + 8 * doct);
        unsigned xv[S3_IPT];
        const unsigned* rowbase = a.xq + (((long)n * a.Tp + 2 * od) * a.Hp + 2 * oh) * a.Wp + 2 * ow0;
#pragma unroll
        for (int i = 0; i < S3_IPT; ++i) {
            xv[i] = 0u;
            if (it_kb[i] >= 0 && 2 * ow0 + it_q[i] < a.Wp) {
                const int kz = it_kb[i] / 7, ky = it_kb[i] - kz * 7;
                xv[i] = rowbase[((long)kz * a.Hp + ky) * a.Wp + it_q[i]];
            }
        }
        __syncthreads();                                   // the previous segment's fragment reads are done
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned short h = (unsigned short)((j & 1) ? (dv[j >> 1] >> 16) : (dv[j >> 1] & 0xffffu));
            *reinterpret_cast<unsigned short*>(dyT + s3_lds(8 * doct + j, dpx)) = h;
        }
#pragma unroll
        for (int i = 0; i < S3_IPT; ++i) {
            if (it_kb[i] < 0) continue;
            const unsigned short c0 = (unsigned short)(xv[i] & 0xffffu), c1 = (unsigned short)(xv[i] >> 16);
            const int q = it_q[i];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kx = (q & 1) + 2 * u, ow = (q - kx) >> 1;
                if (kx < 7 && ow >= 0 && ow < 32 && q >= kx) {
                    const int row = it_kb[i] * 16 + 2 * kx;
                    *reinterpret_cast<unsigned short*>(lds + s3_lds(row, ow)) = c0;
                    *reinterpret_cast<unsigned short*>(lds + s3_lds(row + 1, ow)) = c1;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int slot = ((2 * kb + half) ^ swz) << 4;
            u32x4 bf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const u32x4*>(dyT + (32 * j + l31) * 64 + slot);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (wave + 4 * i >= 25) continue;
                const u32x4 af = *reinterpret_cast<const u32x4*>(lds + fa[i] + slot);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf[j]),
                                                                        acc[i][j], 0, 0, 0);
            }
        }
    }
    // partial: lane holds column l31 (co) of column tile j, rows 8 g + 4 half + e of row tile
    float* part = a.part + (size_t)blockIdx.x * S3_ROWS * S3_CO;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (wave + 4 * i >= 25) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * (wave + 4 * i) + 8 * (e >> 2) + 4 * half + (e & 3);
                part[(size_t)row * S3_CO + 32 * j + l31] = acc[i][j][e];
            }
    }
}

// dw [64][2][7][7][7] = sum over workgroups of part[g][kb * 16 + 2 kx + c][co], fixed order
// (thread = (used partial row, co) with co fastest: a wave reads 256 contiguous bytes of every partial.  Indexed by the OUTPUT element,
// co slowest, every lane of a wave read its own line, and one load at a time was in flight per thread: 267 us per launch for 52 MB of
// partials; the sums and their order are the same)
__global__ __launch_bounds__(256) void stem3d_wgrad_reduce_kernel(const float* __restrict__ part, int groups, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S3_CO * 686) return;
    const int co = i & (S3_CO - 1), rj = i / S3_CO;         // rj = kb * 14 + (2 kx + c)
    const int kb = rj / 14, j = rj - kb * 14, kx = j >> 1, c = j & 1;
    const size_t off = (size_t)(kb * 16 + j) * S3_CO + co;
    float s = 0.f;
    int g = 0;
    for (; g + 16 <= groups; g += 16) {                      // 16 loads in flight, added in group order (172 workgroups: latency-bound)
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = part[(size_t)(g + k) * S3_ROWS * S3_CO + off];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    }
    for (; g < groups; ++g) s += part[(size_t)g * S3_ROWS * S3_CO + off];
    dw[(size_t)co * 686 + c * 343 + kb * 7 + kx] = s;
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the stem, second form (round 6): the kernel above spends its time on the 2-byte LDS scatter that turns every
// 32-pixel segment into a [784][32 px] operand (112 ds_write_b16 per thread and segment against 28 matrix instructions per
// wave: matrix pipe 9.5 % busy, 0.94 ms for the 3 x 64-frame micro-step).  Here nothing is scattered:
//   * the cue is prepared as PLANES, xp [n][z][y][c][parity p][128] bf16 with plane[i] = x[2 i + p - 2] (stem3d_prep_planes_kernel):
//     the operand row of tap (kz, ky, kx = 2 u + p, c) for output row (od, oh) is plane (c, p) of input row (2 od + kz, 2 oh + ky)
//     SHIFTED by u -- dw[kz,ky,2u+p,c][co] = sum_i plane[i] * dy[i - u][co] -- so the four taps u share ONE operand, the
//     1 KB input row as it lies in memory, and the shift moves to dy.  GEMM per output row: M = 49 input rows x 4 planes
//     = 196 (7 tiles of 32), N = 64, K = OW + 3 <= 128 positions i;
//   * a workgroup walks consecutive output rows of one (n, od) plane and keeps the 7 x 7 input rows of the current output row in
//     a ring of 16 slots per kz (LDS, filled by global_load_lds: no registers, no scatter): the next output row needs TWO new
//     input rows per kz -- 14 KB instead of 49 KB -- and they land while the current row is multiplied;
//   * dy rows are transposed once into dyT [co][i] (pixel pairs as dwords: 16 conflict-free ds_write_b32 per thread and row),
//     double-buffered; wave u takes its fragment for positions i - u from two aligned reads and a funnel shift by the
//     compile-time u.
// Wave u owns taps kx = 2 u, 2 u + 1: 7 M tiles x 2 column tiles = 14 accumulators (224 registers); one barrier per output row.
// Partials [workgroup][u][224][64] fp32, summed in workgroup order by stem3d_w2_reduce_kernel (deterministic).
// ------------------------------------------------------------------------------------------
constexpr int W2_SLOTS = 16;
constexpr int W2_A_BYTES = 7 * W2_SLOTS * 1024;            // 114,688: [kz][slot][c][p][128 i] bf16
constexpr int W2_BPITCH = 272;                             // dyT row: 16 zero bytes (i = -8 .. -1) + 128 positions; 272 = 16 (mod 32): fragment reads conflict-free
constexpr int W2_B_BYTES = 64 * W2_BPITCH;                 // 17,408
constexpr int W2_LDS = W2_A_BYTES + 2 * W2_B_BYTES;        // 149,504
constexpr int W2_MROWS = 224;

// x fp32 [N][2][T][H][W] -> xp [N][Tp][Hp][c][p][128] bf16: xp[..][c][p][i] = x[c][z - 2][y - 2][2 i + p - 2], zero outside
__global__ __launch_bounds__(256) void stem3d_prep_planes_kernel(const float* __restrict__ x, unsigned* __restrict__ xp, int N, int T, int H,
                                                                 int W, int Tp, int Hp) {
    const int rows = N * Tp * Hp;
    const long plane = (long)T * H * W;
    const int e0 = 2 * threadIdx.x;                        // two consecutive i of one (c, p) plane per thread
    const int c = e0 >> 8, par = (e0 >> 7) & 1, i = e0 & 127;
    const int w0 = 2 * i + par - 2, w1 = w0 + 2;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const int yy = r % Hp, q = r / Hp;
        const int zz = q % Tp, n = q / Tp;
        const int t = zz - 2, h = yy - 2;
        unsigned v = 0;
        if (t >= 0 && t < T && h >= 0 && h < H) {
            const float* src = x + ((long)n * 2 + c) * plane + ((long)t * H + h) * W;
            if (w0 >= 0 && w0 < W) v = f2bf(src[w0]);
            if (w1 >= 0 && w1 < W) v |= f2bf(src[w1]) << 16;
        }
        xp[(long)r * 256 + threadIdx.x] = v;
    }
}

struct Stem3dW2Args {
    const bf16_t* xp;      // [N][Tp][Hp][2][2][128]
    const bf16_t* dy;      // [N][OD][OH][OW][64]
    float* part;           // [gridDim.x][4][224][64]
    int N, OD, OH, OW, Tp, Hp, ksteps;
    long nrows, per_wg;
};

__device__ __forceinline__ void w2_dma16(const void* src, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"((unsigned long long)src), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

template <int U>
__device__ __forceinline__ void w2_wave(const Stem3dW2Args& a, char* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned lds0 = lds_addr_of(lds);
    char* bT = lds + W2_A_BYTES;

    f32x16 acc[7][2];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // operand rows of this lane: M row 32 mt + l31 = (kz * 7 + ky) * 4 + (c, p); rows >= 196 (tile 6) read row 192 + (c, p): never stored
    int kz_[7], ky_[7];
#pragma unroll
    for (int mt = 0; mt < 7; ++mt) {
        int kzky = 8 * mt + (l31 >> 2);
        if (kzky > 48) kzky = 48;
        kz_[mt] = kzky / 7;
        ky_[mt] = kzky - 7 * kz_[mt];
    }
    const int cp = l31 & 3;
    // transfers: lane -> plane lane >> 4 of the 1 KB input row, 16-byte unit lane & 15 of it
    const int dcp = lane >> 4, dpos = lane & 15;
    // dy transposition: thread -> pixel pair (tid >> 1) & 63, channel octets (tid & 1) + 4 (tid >> 7) + 2 it, it = 0, 1
    const int pair = (tid >> 1) & 63, oct0 = (tid & 1) + 4 * (tid >> 7);
    const int npairs = (a.OW + 1) >> 1;

    const int r0 = blockIdx.x * (int)a.per_wg;
    int r_end = r0 + (int)a.per_wg;
    if (r_end > (int)a.nrows) r_end = (int)a.nrows;
    if (r0 >= r_end) return;                               // (whole workgroup: the grid is sized so that this does not happen)
    int oh = r0 % a.OH;                                    // the row's coordinates advance with it (no division per row)
    int nd = r0 / a.OH;                                    // n * OD + od
    auto row_base = [&](int nd_, int oh_) -> const bf16_t* {   // input row (2 od, 2 oh) of output row (nd_, oh_)
        const int od = nd_ % a.OD, n = nd_ / a.OD;
        return a.xp + (((long)n * a.Tp + 2 * od) * a.Hp + 2 * oh_) * 512;
    };
    const bf16_t* base = row_base(nd, oh);
    auto dma = [&](const bf16_t* base, int kz, int ky, int S) {   // input row (kz, ky) of the output row at `base` into slot S of ring kz
        const int swz = (((S - kz) & 3) << 2) | dcp;
        const bf16_t* src = base + ((long)kz * a.Hp + ky) * 512 + dcp * 128 + ((dpos ^ swz) << 3);
        w2_dma16(src, lds0 + (unsigned)((kz * W2_SLOTS + (S & (W2_SLOTS - 1))) * 1024));
    };
    u32x4 de[2], dd[2];                                    // dy of pixels 2 pair, 2 pair + 1, two octets
    auto dy_load = [&](int row) {
        const bf16_t* src = a.dy + (long)row * a.OW * S3_CO;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int oct = oct0 + 2 * it;
            de[it] = u32x4{0u, 0u, 0u, 0u};
            dd[it] = u32x4{0u, 0u, 0u, 0u};
            if (pair < npairs) {
                de[it] = *reinterpret_cast<const u32x4*>(src + (2 * pair) * S3_CO + 8 * oct);
                if (2 * pair + 1 < a.OW) dd[it] = *reinterpret_cast<const u32x4*>(src + (2 * pair + 1) * S3_CO + 8 * oct);
            }
        }
    };
    auto dy_store = [&](char* buf) {
        if (pair >= npairs) return;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int oct = oct0 + 2 * it;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned e = de[it][j >> 1], o = dd[it][j >> 1];
                const unsigned v = (j & 1) ? ((e >> 16) | (o & 0xffff0000u)) : ((e & 0xffffu) | (o << 16));
                *reinterpret_cast<unsigned*>(buf + (8 * oct + j) * W2_BPITCH + 16 + 4 * pair) = v;
            }
        }
    };

#ifdef W2_TIMING
    long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = (long long)__builtin_amdgcn_s_memtime();
#define W2_LAP(k) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); tm[k] += now_ - tq; tq = now_; }
#else
#define W2_LAP(k)
#endif
    // ---- first output row of the run: its 49 input rows, its dy ----
    int sb = 0;
    {
#pragma unroll 1
        for (int q = U; q < 49; q += 4) {
            const int kz = q / 7, ky = q - 7 * kz;
            dma(base, kz, ky, sb + ky);
        }
        dy_load(r0);
        dy_store(bT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    W2_LAP(0)

#pragma unroll 1
    for (int r = r0; r < r_end; ++r) {
        const int buf = (r - r0) & 1;
        const bool more = r + 1 < r_end;
        const bool same_plane = oh + 1 < a.OH;
        if (same_plane) { ++oh; base += 1024; }
        else { oh = 0; ++nd; if (more) base = row_base(nd, 0); }
        if (more) dy_load(r + 1);
        W2_LAP(6)
        // this row's operand addresses
        int arow[7], axor[7];
#pragma unroll
        for (int mt = 0; mt < 7; ++mt) {
            const int S = sb + ky_[mt];
            arow[mt] = (kz_[mt] * W2_SLOTS + (S & (W2_SLOTS - 1))) * 1024 + cp * 256;
            axor[mt] = ((((S - kz_[mt]) & 3) << 2) | cp) << 4;
        }
        const char* bb = bT + buf * W2_B_BYTES + l31 * W2_BPITCH + 16 + half * 16;
        // One register set for the operand fragments: the read of tile mt for the NEXT k-step is issued right behind the two matrix
        // instructions that use tile mt in this one (it then has the other twelve to land), and the next step's dy fragment -- raw
        // reads first, funnel shift after the fifth tile -- goes into the other of two sets.  With one wave per SIMD nothing else
        // fills the matrix pipe: every other instruction has to sit BETWEEN matrix instructions, not in a block before them
        // (a block of ~20 reads and address instructions in front of 14 matrix instructions cost ~170 of 620 clocks per k-step).
        auto readA = [&](int ks, int mt) -> u32x4 {
            return *reinterpret_cast<const u32x4*>(lds + arow[mt] + (((2 * ks + half) << 4) ^ axor[mt]));
        };
        auto readB = [&](int ks, u32x4 (&C)[2], u32x2 (&P)[2]) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const char* p = bb + nt * 32 * W2_BPITCH + ks * 32;
                C[nt] = *reinterpret_cast<const u32x4*>(p);
                if (U == 1 || U == 2) P[nt] = u32x2{0u, *reinterpret_cast<const unsigned*>(p - 4)};
                else if (U == 3) P[nt] = *reinterpret_cast<const u32x2*>(p - 8);
                else P[nt] = u32x2{0u, 0u};
            }
        };
        auto shifted = [&](const u32x4& c, const u32x2& pq) -> u32x4 {
            if (U == 0) return c;
            if (U == 2) return u32x4{pq[1], c[0], c[1], c[2]};
            if (U == 1)
                return u32x4{__builtin_amdgcn_alignbit(c[0], pq[1], 16), __builtin_amdgcn_alignbit(c[1], c[0], 16),
                             __builtin_amdgcn_alignbit(c[2], c[1], 16), __builtin_amdgcn_alignbit(c[3], c[2], 16)};
            return u32x4{__builtin_amdgcn_alignbit(pq[1], pq[0], 16), __builtin_amdgcn_alignbit(c[0], pq[1], 16),
                         __builtin_amdgcn_alignbit(c[1], c[0], 16), __builtin_amdgcn_alignbit(c[2], c[1], 16)};
        };
        u32x4 af[7], b0[2], b1[2];
        auto step = [&](int ksn, const u32x4 (&B)[2], u32x4 (&Bn)[2]) {   // k-step with fragments (af, B); fetches k-step ksn into (af, Bn)
            u32x4 C[2];
            u32x2 P[2];
#pragma unroll
            for (int mt = 0; mt < 7; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mt]), __builtin_bit_cast(bf16x8, B[nt]),
                                                                          acc[mt][nt], 0, 0, 0);
                af[mt] = readA(ksn, mt);
                if (mt == 0) readB(ksn, C, P);
                if (mt == 5) {
                    Bn[0] = shifted(C[0], P[0]);
                    Bn[1] = shifted(C[1], P[1]);
                }
                __builtin_amdgcn_sched_barrier(0);         // source order, group by group (the scheduler otherwise bunches the reads)
            }
        };
        {
            u32x4 C[2];
            u32x2 P[2];
#pragma unroll
            for (int mt = 0; mt < 7; ++mt) af[mt] = readA(0, mt);
            readB(0, C, P);
            b0[0] = shifted(C[0], P[0]);
            b0[1] = shifted(C[1], P[1]);
        }
        // The next row's transfers go out HERE, behind the wait for the first fragments: global_load_lds counts in lgkmcnt as well
        // as vmcnt (measured: any lgkmcnt(0) behind them waits for the data, ~1,100 clocks), so nothing that drains lgkmcnt may
        // follow them closely -- the k-steps' waits are all lgkmcnt(>= 5)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) {
            if (same_plane) {                              // input rows 2 (oh + 1) + 5, + 6 of every kz: slots sb + 7, sb + 8
#pragma unroll 1
                for (int q = U; q < 14; q += 4) dma(base, q >> 1, 5 + (q & 1), sb + 7 + (q & 1));
            } else {
#pragma unroll 1
                for (int q = U; q < 49; q += 4) {
                    const int kz = q / 7, ky = q - 7 * kz;
                    dma(base, kz, ky, sb + 7 + ky);
                }
            }
        }
        W2_LAP(1)
        // k-steps in pairs, no branch inside (an odd count runs one more step: its dy positions are zero, its operand bytes valid)
#pragma unroll 1
        for (int ks = 0; ks < a.ksteps; ks += 2) {
            __builtin_amdgcn_sched_barrier(0);
            step(ks + 1, b0, b1);
            __builtin_amdgcn_sched_barrier(0);
            step(ks + 2 < 8 ? ks + 2 : 7, b1, b0);
            __builtin_amdgcn_sched_barrier(0);
        }
        W2_LAP(2)
        if (more) dy_store(bT + (buf ^ 1) * W2_B_BYTES);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        W2_LAP(3)
        __syncthreads();
        W2_LAP(4)
        sb = (sb + (same_plane ? 2 : 7)) & (W2_SLOTS - 1);
    }

    // partial: lane holds column l31 (co) of column tile nt, rows 8 g + 4 half + e of row tile mt
    float* part = a.part + ((size_t)blockIdx.x * 4 + U) * W2_MROWS * S3_CO;
#pragma unroll
    for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 32 * mt + 8 * (e >> 2) + 4 * half + (e & 3);
                if (U == 3 && (e & 1)) continue;           // kx = 7 does not exist: the reduce never reads these rows
                if (row < 196) part[(size_t)row * S3_CO + 32 * nt + l31] = acc[mt][nt][e];
            }
#ifdef W2_TIMING
    if (lane == 0)
        for (int k = 0; k < 8; ++k) part[k] = (float)tm[k];
#endif
}

__global__ __launch_bounds__(256) void stem3d_w2_kernel(Stem3dW2Args a) {
    extern __shared__ __attribute__((aligned(1024))) char w2_lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // dyT buffers: zero once (the zero unit in front of every row and the positions >= OW are never written)
    for (int i = threadIdx.x; i < 2 * W2_B_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(w2_lds + W2_A_BYTES)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (wave == 0) w2_wave<0>(a, w2_lds);
    else if (wave == 1) w2_wave<1>(a, w2_lds);
    else if (wave == 2) w2_wave<2>(a, w2_lds);
    else w2_wave<3>(a, w2_lds);
}

// dw [64][2][7][7][7] = sum over workgroups of part[g][u][(kz * 7 + ky) * 4 + 2 c + p][co] with kx = 2 u + p, fixed order
__global__ __launch_bounds__(256) void stem3d_w2_reduce_kernel(const float* __restrict__ part, int groups, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S3_CO * 686) return;
    const int co = i & (S3_CO - 1), rj = i / S3_CO;         // rj = kb * 14 + (2 kx + c)
    const int kb = rj / 14, j = rj - kb * 14, kx = j >> 1, c = j & 1;
    const size_t off = ((size_t)(kx >> 1) * W2_MROWS + kb * 4 + 2 * c + (kx & 1)) * S3_CO + co;
    const size_t gs = (size_t)4 * W2_MROWS * S3_CO;
    float s = 0.f;
    int g = 0;
    for (; g + 16 <= groups; g += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = part[(size_t)(g + k) * gs + off];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    }
    for (; g < groups; ++g) s += part[(size_t)g * gs + off];
    dw[(size_t)co * 686 + c * 343 + kb * 7 + kx] = s;
}

// ------------------------------------------------------------------------------------------
// Data gradient of the stem (the gradient of the 2-channel cue): per input row (n, t, h)
//     Q[ow][(kx, c)] = sum over the (kz, ky) whose stride-2 window reaches (t, h), and over co, of dy[od][oh][ow][co] * w[co][c][kz][ky][kx]
//     dx[t][h][w][c] = sum over kx = w (mod 2) of Q[(w + 2 - kx) / 2][(kx, c)]
// The first line is a GEMM with M = the OW output pixels of a row, N = 14 (+ 2 zero) columns and K = 64 channels x up to 16
// window rows -- A fragments are 16 contiguous bytes of dy (NDHWC), B fragments 16 contiguous bytes of the weights packed as
// [kz][ky][(kx, c)][co]: no transposition anywhere, v_mfma_f32_16x16x32_bf16, 87.5 % useful columns (a per-pixel gather
// formulation would fill 2 of 16 columns).  The second line is a 1-D fold of the row through LDS.  One wave per input row,
// fp32 output in the cue's NCDHW layout, deterministic.
// ------------------------------------------------------------------------------------------
typedef float s3_f32x4 __attribute__((ext_vector_type(4)));

// w fp32 [64][2][7][7][7] -> wq bf16 [49][16][64]: row (kz * 7 + ky) * 16 + 2 kx + c, 64 channels contiguous; rows 14, 15 zero
__global__ __launch_bounds__(256) void stem3d_pack_wq_kernel(const float* __restrict__ w, bf16_t* __restrict__ wq) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S3_KB * 16 * S3_CO) return;
    const int co = i & 63, j = (i >> 6) & 15, kb = i >> 10;
    unsigned v = 0;
    if (j < 14) v = f2bf(w[((co * 2 + (j & 1)) * 343) + kb * 7 + (j >> 1)]);
    wq[i] = (bf16_t)v;
}

struct Stem3dDgArgs {
    const bf16_t* dy;      // [N][OD][OH][OW][64]
    const bf16_t* wq;      // [49][16][64]
    float* dx;             // [N][2][T][H][W]
    int N, T, H, W, OD, OH, OW;
};

constexpr int S3_DG_WAVES = 4;
constexpr int S3_DG_MT = 8;                                // up to 128 output pixels per row

template <int MTC>                                         // number of 16-pixel row tiles when known at compile time (the loads of a
                                                           // k-block are then issued back to back); 0 = runtime
__global__ __launch_bounds__(S3_DG_WAVES * 64) void stem3d_dgrad_kernel(Stem3dDgArgs a) {
    __shared__ float qlds[S3_DG_WAVES][S3_DG_MT * 16 * 16];   // Q tile of each wave: [ow][16]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, kg = lane >> 4;
    float* Q = qlds[wave];
    const long nrows = (long)a.N * a.T * a.H;
    const int mtiles = MTC ? MTC : (a.OW + 15) / 16;
    const long plane = (long)a.T * a.H * a.W;
    for (long row = (long)blockIdx.x * S3_DG_WAVES + wave; row < nrows; row += (long)gridDim.x * S3_DG_WAVES) {
        const int h = (int)(row % a.H);
        const long nt = row / a.H;
        const int t = (int)(nt % a.T);
        const int n = (int)(nt / a.T);
        s3_f32x4 acc[S3_DG_MT];
#pragma unroll
        for (int m = 0; m < S3_DG_MT; ++m) acc[m] = s3_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kz = (t & 1); kz < S3_K; kz += 2) {
            const int od = (t + 2 - kz) >> 1;
            if (t + 2 - kz < 0 || od >= a.OD) continue;
            for (int ky = (h & 1); ky < S3_K; ky += 2) {
                const int oh = (h + 2 - ky) >> 1;
                if (h + 2 - ky < 0 || oh >= a.OH) continue;
                const bf16_t* drow = a.dy + (((long)n * a.OD + od) * a.OH + oh) * a.OW * S3_CO;
                const bf16_t* wrow = a.wq + ((kz * S3_K + ky) * 16 + l15) * S3_CO;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const u32x4 bf = *reinterpret_cast<const u32x4*>(wrow + 32 * kb + 8 * kg);
                    u32x4 af[S3_DG_MT];
#pragma unroll
                    for (int m = 0; m < S3_DG_MT; ++m) {
                        if (MTC ? m >= MTC : m >= mtiles) continue;
                        int ow = 16 * m + l15;
                        ow = ow < a.OW ? ow : a.OW - 1;            // clipped rows: valid memory, their Q rows are never read
                        af[m] = *reinterpret_cast<const u32x4*>(drow + (long)ow * S3_CO + 32 * kb + 8 * kg);
                    }
#pragma unroll
                    for (int m = 0; m < S3_DG_MT; ++m) {
                        if (MTC ? m >= MTC : m >= mtiles) continue;
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), __builtin_bit_cast(bf16x8, bf), acc[m], 0, 0, 0);
                    }
                }
            }
        }
        // C layout of 16x16: lane (column l15 = j, rows 4 kg + q = ow within the tile)
#pragma unroll
        for (int m = 0; m < S3_DG_MT; ++m) {
            if (m >= mtiles) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) Q[(16 * m + 4 * kg + q) * 16 + l15] = acc[m][q];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // fold: dx[w][c] = sum_{kx = w (mod 2)} Q[(w + 2 - kx) / 2][2 kx + c]
        float* dst = a.dx + ((long)n * 2 * a.T + t) * a.H * a.W + (long)h * a.W;
        for (int i = lane; i < 2 * a.W; i += 64) {
            const int c = i / a.W, w = i - c * a.W;
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kx = (w & 1) + 2 * u;
                const int ow2 = w + 2 - kx;
                if (kx < S3_K && ow2 >= 0 && (ow2 >> 1) < a.OW) sum += Q[(ow2 >> 1) * 16 + 2 * kx + c];
            }
            dst[c * plane + w] = sum;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// The same computation with FOUR consecutive input rows per wave (h0 .. h0 + 3, h0 a multiple of 4).  With one row per wave
// every dy row was fetched once per (input row, ky) that uses it -- 12 times on average, 7.4 GB through the L2 for the 64-frame
// clips, matrix pipe 5 % busy (profiles/r4_pmc_i3d_kernels.csv).  The four rows' windows overlap: the five dy rows
// oh = h0 / 2 + d, d = -2 .. 2, serve 1 + 3 + 4 + 4 + 2 = 14 (row, ky) pairs (ky = r + 2 - 2 d), so each A fragment is
// loaded once and multiplied into up to four accumulator sets: 2.8x fewer dy bytes, same MFMAs.  Fold as above, row by row;
// Q rows are 20 floats apart (16 made the fold's reads 16-way bank conflicts).
constexpr int S3_QS = 20;
template <int MTC>
__global__ __launch_bounds__(S3_DG_WAVES * 64) void stem3d_dgrad4_kernel(Stem3dDgArgs a) {
    static_assert(MTC >= 1 && MTC <= S3_DG_MT, "compile-time tile count");
    __shared__ float qlds[S3_DG_WAVES][S3_DG_MT * 16 * S3_QS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, kg = lane >> 4;
    float* Q = qlds[wave];
    const int hgroups = (a.H + 3) >> 2;
    const long ngroups = (long)a.N * a.T * hgroups;
    const long plane = (long)a.T * a.H * a.W;
    for (long grp = (long)blockIdx.x * S3_DG_WAVES + wave; grp < ngroups; grp += (long)gridDim.x * S3_DG_WAVES) {
        const int h0 = (int)(grp % hgroups) * 4;
        const long nt = grp / hgroups;
        const int t = (int)(nt % a.T);
        const int n = (int)(nt / a.T);
        const int nrow = a.H - h0 < 4 ? a.H - h0 : 4;       // rows of this group inside the image
        s3_f32x4 acc[4][MTC];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MTC; ++m) acc[r][m] = s3_f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kz = (t & 1); kz < S3_K; kz += 2) {
            const int od = (t + 2 - kz) >> 1;
            if (t + 2 - kz < 0 || od >= a.OD) continue;
            const bf16_t* wkz = a.wq + (long)kz * S3_K * 16 * S3_CO + l15 * S3_CO + 8 * kg;
#pragma unroll
            for (int d = -2; d <= 2; ++d) {
                const int oh = (h0 >> 1) + d;
                if (oh < 0 || oh >= a.OH) continue;                // (wave-uniform)
                const bf16_t* drow = a.dy + (((long)n * a.OD + od) * a.OH + oh) * a.OW * S3_CO + 8 * kg;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    u32x4 af[MTC];
#pragma unroll
                    for (int m = 0; m < MTC; ++m) {
                        int ow = 16 * m + l15;
                        ow = ow < a.OW ? ow : a.OW - 1;            // clipped rows: valid memory, their Q rows are never read
                        af[m] = *reinterpret_cast<const u32x4*>(drow + (long)ow * S3_CO + 32 * kb);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ky = r + 2 - 2 * d;              // compile-time
                        if (ky < 0 || ky >= S3_K) continue;
                        if (r >= nrow) continue;                   // (wave-uniform)
                        const u32x4 bf = *reinterpret_cast<const u32x4*>(wkz + ky * 16 * S3_CO + 32 * kb);
#pragma unroll
                        for (int m = 0; m < MTC; ++m)
                            acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), __builtin_bit_cast(bf16x8, bf), acc[r][m], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nrow) break;
            // C layout of 16x16: lane (column l15 = j, rows 4 kg + q = ow within the tile)
#pragma unroll
            for (int m = 0; m < MTC; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) Q[(16 * m + 4 * kg + q) * S3_QS + l15] = acc[r][m][q];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // fold: dx[w][c] = sum_{kx = w (mod 2)} Q[(w + 2 - kx) / 2][2 kx + c]
            float* dst = a.dx + ((long)n * 2 * a.T + t) * a.H * a.W + (long)(h0 + r) * a.W;
            for (int i = lane; i < 2 * a.W; i += 64) {
                const int c = i / a.W, w = i - c * a.W;
                float sum = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kx = (w & 1) + 2 * u;
                    const int ow2 = w + 2 - kx;
                    if (kx < S3_K && ow2 >= 0 && (ow2 >> 1) < a.OW) sum += Q[(ow2 >> 1) * S3_QS + 2 * kx + c];
                }
                dst[c * plane + w] = sum;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// ------------------------------------------------------------------------------------------
// Data gradient of the stem, block form (round 6).  The row kernels above read both operands of every matrix instruction from
// global memory (two 1 KB loads per 16-clock instruction: matrix pipe 8.5 % busy, 0.60 ms for the 3 x 64-frame micro-step).
// Here a workgroup owns a BLOCK of the cue's gradient -- 4 frames t x 8 rows h, full width: wave w owns frame 4 tb + w and keeps
// its eight rows' Q[ow][(kx, c)] in accumulators (8 x 7 tiles of 16 x 16 = 224 registers) -- all 49 weight blocks
// [(kx, c)][co] sit in LDS for the whole (persistent) launch, and the dy rows the block needs (od = 2 tb - 2 .. 2 tb + 2,
// oh = 4 hb - 2 .. 4 hb + 4: at most 35) pass through a double-buffered LDS row, staged by all four waves (global -> registers ->
// ds_write_b128: vmcnt only, no lgkmcnt coupling) while the previous row is multiplied.  A staged row serves every (frame,
// row) pair of the block it reaches: wave w uses it with kz = w + 6 - 2 od_l (if that is a tap) and, for its row h_l, with
// ky = h_l + 6 - 2 oh_l -- both static in the unrolled code, so accumulator indices are compile-time.  A dy row is read from
// L2 once per ~0.9 rows of dx (the row kernel: 2.6 GB through the L2 per launch, here 0.55 GB).  Pixel rows and weight rows
// are 128 bytes with their 16-byte pieces XOR-swizzled (d3_swz): fragment reads and staging stores touch every bank once.  The fold of Q
// into dx goes through slots in LDS (see below).  OW <= 112.
// ------------------------------------------------------------------------------------------
constexpr int D3_PITCH = 128;                              // rows of 128 bytes = eight 16-byte pieces; piece p of row r lives in piece p ^ d3_swz(r)
constexpr int D3_W_BYTES = S3_KB * 16 * D3_PITCH;          // 100,352
constexpr int D3_ROW_BYTES = 128 * D3_PITCH;               // 16,384 (128 pixels)
constexpr int D3_LDS = D3_W_BYTES + 2 * D3_ROW_BYTES;      // 133,120: weights | two dy rows (the fold's rows, 4 x 7,424, at a block's end)
// The 16 lanes a ds_read_b128 serves together are {0-3, 12-15, 20-27} (+ 4 / + 32 / + 36 for the other three groups): in a
// 16 x 16 x 32 fragment read, rows 0-3 and 12-15 of one k-group and rows 4-11 of the next.  With 128-byte rows (bank = 8 (r & 1) +
// piece) the XOR below sends them to sixteen different 16-byte bank groups, and the staging stores (8 pieces of 8 pixels per wave) as
// well; a 144-byte pitch was conflict-free within a k-group only (LDS bank-conflict share 0.45, profiles/r6_pmc_i3d_kernels.csv).
__device__ __forceinline__ int d3_swz(int row) { return ((row >> 1) & 3) << 1; }
constexpr int D3_MT = 7;

__global__ __launch_bounds__(256) void stem3d_dgrad_blk_kernel(Stem3dDgArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char d3_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    for (int i = tid; i < S3_KB * 16 * 8; i += 256) {      // the packed weights [49][16][64] -> rows of 144 bytes
        const int row = i >> 3, piece = i & 7;
        *reinterpret_cast<u32x4*>(d3_lds + row * D3_PITCH + (piece ^ d3_swz(row)) * 16) = *reinterpret_cast<const u32x4*>(a.wq + row * S3_CO + piece * 8);
    }
    char* rows = d3_lds + D3_W_BYTES;
    for (int i = tid; i < 2 * D3_ROW_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(rows)[i] = u32x4{0u, 0u, 0u, 0u};   // pixels >= OW stay zero
    const int tblocks = (a.T + 3) >> 2, hblocks = (a.H + 7) >> 3;
    const int total = a.N * tblocks * hblocks;
    const int pieces = a.OW * 8;                           // 16-byte pieces of a dy row
    const long plane = (long)a.T * a.H * a.W;
    const int wfrag = l15 * D3_PITCH + (kg ^ d3_swz(l15)) * 16;   // this lane's 16 bytes of a weight block / of a 16-pixel tile, first k-step
    const int dk = (d3_swz(l15) & 4) ? -64 : 64;           // ... to the second k-step (piece ^ 4)
    __syncthreads();
#ifdef D3_TIMING
    long long tm[6] = {0, 0, 0, 0, 0, 0}, tq = (long long)__builtin_amdgcn_s_memtime();
#define D3_LAP(k) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); tm[k] += now_ - tq; tq = now_; }
#else
#define D3_LAP(k)
#endif

#pragma unroll 1
    for (int blk = blockIdx.x; blk < total; blk += gridDim.x) {
        const int hb = blk % hblocks, q = blk / hblocks;
        const int tb = q % tblocks, n = q / tblocks;
        const int t = 4 * tb + wave;
        s3_f32x4 acc[8][D3_MT];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int m = 0; m < D3_MT; ++m) acc[r][m] = s3_f32x4{0.f, 0.f, 0.f, 0.f};

        auto row_ok = [&](int od_l, int oh_l) -> bool {
            const int od = 2 * tb - 2 + od_l, oh = 4 * hb - 2 + oh_l;
            return od_l < 5 && od >= 0 && od < a.OD && oh >= 0 && oh < a.OH;
        };
        // Staging: row s + 3 is requested at the top of step s into register set s & 1 and written to LDS two steps later (a load
        // takes ~2 us under this traffic, a step ~0.5: with one register set, loaded and stored within a step, every step waited
        // for memory).  Eight steps per od_l (the eighth is empty) keep the parities compile-time.
        u32x4 sv[2][4];
        // (unconditional, coordinates clamped into the volume: with loads under a branch the compiler cannot count what is in flight
        // and waits with vmcnt(0), which is the one-step coverage again; a row outside the volume is staged and not used)
        auto g_load = [&](int od_l, int oh_l, u32x4 (&R)[4]) {
            int od = 2 * tb - 2 + od_l, oh = 4 * hb - 2 + oh_l;
            od = od < 0 ? 0 : (od >= a.OD ? a.OD - 1 : od);
            oh = oh < 0 ? 0 : (oh >= a.OH ? a.OH - 1 : oh);
            const bf16_t* src = a.dy + ((((long)n * a.OD + od) * a.OH + oh) * a.OW) * S3_CO;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int p = tid + 256 * k;
                p = p < pieces ? p : pieces - 1;           // (beyond the row: the last piece again, no branch)
                R[k] = *reinterpret_cast<const u32x4*>(src + (long)p * 8);
            }
        };
        auto l_store = [&](int buf, const u32x4 (&R)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int p = tid + 256 * k;
                p = p < pieces ? p : pieces - 1;
                *reinterpret_cast<u32x4*>(rows + buf * D3_ROW_BYTES + (p >> 3) * D3_PITCH + ((p & 7) ^ d3_swz(p >> 3)) * 16) = R[k];
            }
        };
        g_load(0, 0, sv[0]);
        l_store(0, sv[0]);
        g_load(0, 1, sv[0]);                               // row 1 -> set 0, row 2 -> set 1 (written to LDS at the top of steps 0 and 1)
        g_load(0, 2, sv[1]);
        __syncthreads();
        D3_LAP(0)
#pragma unroll 1
        for (int od_l = 0; od_l < 5; ++od_l) {
            const int kz = wave + 6 - 2 * od_l;
            const bool wave_on = kz >= 0 && kz < S3_K && t < a.T;
            const char* wkz = d3_lds + kz * (S3_K * 16 * D3_PITCH) + wfrag;
#pragma unroll
            for (int oh_l = 0; oh_l < 8; ++oh_l) {
                const int buf = oh_l & 1;
                // row s + 1 (requested two steps ago) into the other buffer; then request row s + 3 into the freed registers
                l_store(buf ^ 1, sv[oh_l & 1]);
                g_load(oh_l + 3 < 8 ? od_l : od_l + 1, (oh_l + 3) & 7, sv[oh_l & 1]);
                D3_LAP(1)
                if (oh_l < 7 && wave_on && row_ok(od_l, oh_l)) {
                    const char* rb = rows + buf * D3_ROW_BYTES + wfrag;
                    // all of the step's weight fragments and the first k-step's pixel fragments up front (one exposed LDS round trip
                    // per step instead of one per group of seven matrix instructions: with one wave per SIMD nothing else covers it);
                    // the second k-step's pixel fragments are read between the first's matrix instructions
                    u32x4 bf[8][2], af0[D3_MT], af1[D3_MT];
#pragma unroll
                    for (int m = 0; m < D3_MT; ++m) af0[m] = *reinterpret_cast<const u32x4*>(rb + m * 16 * D3_PITCH);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int ky = r + 6 - 2 * oh_l;           // compile-time
                        if (ky < 0 || ky >= S3_K) continue;
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) bf[r][ks] = *reinterpret_cast<const u32x4*>(wkz + ky * 16 * D3_PITCH + ks * dk);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    int nread = 0;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int ky = r + 6 - 2 * oh_l;
                        if (ky < 0 || ky >= S3_K) continue;
#pragma unroll
                        for (int m = 0; m < D3_MT; ++m) {
                            acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af0[m]),
                                                                                 __builtin_bit_cast(bf16x8, bf[r][0]), acc[r][m], 0, 0, 0);
                            if ((m & 1) == 0 && nread < D3_MT) {
                                af1[nread] = *reinterpret_cast<const u32x4*>(rb + nread * 16 * D3_PITCH + dk);
                                ++nread;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
#pragma unroll
                    for (; nread < D3_MT; ++nread) af1[nread] = *reinterpret_cast<const u32x4*>(rb + nread * 16 * D3_PITCH + dk);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const int ky = r + 6 - 2 * oh_l;
                        if (ky < 0 || ky >= S3_K) continue;
#pragma unroll
                        for (int m = 0; m < D3_MT; ++m)
                            acc[r][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af1[m]),
                                                                                 __builtin_bit_cast(bf16x8, bf[r][1]), acc[r][m], 0, 0, 0);
                    }
                }
                D3_LAP(2)
                __syncthreads();
                D3_LAP(3)
            }
        }
        // Fold: dx[w][c] = sum over kx = w (mod 2) of Q[(w + 2 - kx) / 2][(kx, c)].  A lane holds column (kx, c) = l15 of Q for 28
        // output pixels ow = 16 m + 4 kg + q, and its value is term u = kx / 2 of dx pixel w = 2 ow + kx - 2: it is WRITTEN to slot
        // (c, w, u) of a [2][232][4] float row in LDS (one base address per lane + compile-time offsets, every slot has one
        // writer), and an output pixel is one ds_read_b128 and three additions in the row kernels' order (kx ascending).  Slots no
        // pixel writes (ow < 0 or >= 112 at the two ends of a row) are zeroed once per block; columns 14 / 15 carry zero weights.
        // (The row kernels' fold -- per output pixel four conditional LDS reads and an integer division -- is latency that
        // other waves hide there; with ONE wave per SIMD it was 46 % of this kernel.  ds_add_f32 into one float per pixel was worse
        // still: ~780 clocks per instruction.)  The rows live in the (then idle) dy buffers.
        int lane_f = lane;                                 // (opaque: the fold's 56 row / column addresses are NOT to be computed -- and
        asm volatile("" : "+v"(lane_f));                   // spilled -- in front of the staging loop, where the compiler hoists them)
        float* F = reinterpret_cast<float*>(rows) + wave * (2 * 232 * 4);
        float* fw = F + ((l15 & 1) * 232 + 8 * kg + (l15 >> 1)) * 4 + (l15 >> 2);   // slot (c, w, u) at F[((c * 232) + 2 + w) * 4 + u]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = lane + 64 * j;                   // 2 planes x (8 + 12) edge pixels x 4 slots
            const int c = k / 80, rem = k - 80 * c, e = rem >> 2;
            if (k < 160) F[((c * 232) + (e < 8 ? e : 212 + e)) * 4 + (rem & 3)] = 0.f;
        }
        if (t < a.T) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int h = 8 * hb + r;
                if (h >= a.H) break;
#pragma unroll
                for (int m = 0; m < D3_MT; ++m)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) fw[(32 * m + 2 * qq) * 4] = acc[r][m][qq];
                __builtin_amdgcn_wave_barrier();
                float* dst = a.dx + ((long)n * 2 * a.T + t) * a.H * a.W + (long)h * a.W;
#pragma unroll
                for (int it = 0; it < 7; ++it) {                   // 2 W <= 448 = 7 x 64
                    int i = lane_f + 64 * it;
                    i = i < 2 * a.W ? i : 2 * a.W - 1;             // (beyond the row: the last element again -- the same value to the same
                                                                   // address; under a branch the compiler sinks the read into it and the seven
                                                                   // LDS round trips of a row happen one after the other)
                    const int c = i >= a.W ? 1 : 0, w = i - c * a.W;
                    const float4 v = *reinterpret_cast<const float4*>(F + ((c * 232) + 2 + w) * 4);
                    dst[c * plane + w] = ((v.x + v.y) + v.z) + v.w;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * (112 - a.OW) * 8; i += 256) {    // narrow frames: pixels OW .. 111 of the dy buffers are zeros again
            const int bufi = i / ((112 - a.OW) * 8), rem = i - bufi * (112 - a.OW) * 8;
            *reinterpret_cast<u32x4*>(rows + bufi * D3_ROW_BYTES + (a.OW + (rem >> 3)) * D3_PITCH + (rem & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        D3_LAP(4)
        __syncthreads();                                   // the dy buffers are the next block's again
        D3_LAP(5)
    }
#ifdef D3_TIMING
    if (lane == 0)
        for (int k = 0; k < 6; ++k) a.dx[(blockIdx.x * 4 + wave) * 8 + k] = (float)tm[k];
#endif
}

int s3_wg_groups(long nseg) { return (int)(nseg < 256 ? nseg : 256); }

int s3_blocks(long tiles) {
    long b = (tiles + S3_WAVES - 1) / S3_WAVES;
    return (int)(b > 256 ? 256 : (b < 1 ? 1 : b));         // one 100 KB workgroup per CU
}

}  // namespace

extern "C" {

// bytes of the workspace: padded bf16 volume + packed weights
size_t dmc_stem3d_bf16_workspace_bytes(int N, int T, int H, int W) {
    const long Wp = (W + 5 + 1 + 3) / 4 * 4;
    return (size_t)N * (T + 5) * (H + 5) * Wp * 4 + (size_t)S3_KB * S3_CO * 16 * 2 + 64;
}
// rows of [64][2] float partials the forward writes when asked for statistics
int dmc_stem3d_bf16_stat_blocks(int N, int T, int H, int W) {
    const int OD = (T + 5 - 7) / 2 + 1, OH = (H + 5 - 7) / 2 + 1, OW = (W + 5 - 7) / 2 + 1;
    return s3_blocks((long)N * OD * OH * ((OW + 31) / 32)) * S3_WAVES;
}

// y [N,OD,OH,OW,64] bf16 (NDHWC) = conv3d(pad_SAME(x [N,2,T,H,W] fp32), w [64,2,7,7,7] fp32 contiguous, stride 2);
// OD = (T + 5 - 7) / 2 + 1 etc.  x and w are rounded to bf16 (nearest even), fp32 accumulation.
int dmc_stem3d_bf16_fwd(const float* x, const float* w, void* workspace, void* y, float* stat_partials, int N, int T, int H, int W,
                        dmc_stream_t stream) {
    if (!x || !w || !workspace || !y) return fail(DMC_E_INVALID, "dmc_stem3d_bf16_fwd: null pointer");
    if (N <= 0 || T < 2 || H < 2 || W < 2) return fail(DMC_E_INVALID, "dmc_stem3d_bf16_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int Tp = T + 5, Hp = H + 5, Wp = (W + 5 + 1 + 3) / 4 * 4;
    unsigned* xq = (unsigned*)workspace;
    bf16_t* wp = (bf16_t*)((char*)workspace + (size_t)N * Tp * Hp * Wp * 4);
    const long total = (long)N * Tp * Hp * Wp;
    stem3d_prep_kernel<<<(int)(total / Wp > 16384 ? 16384 : total / Wp), 256, 0, s>>>(x, xq, N, T, H, W, Tp, Hp, Wp);
    int rc = check_launch("stem3d_prep");
    if (rc) return rc;
    stem3d_pack_w_kernel<<<(S3_KB * S3_CO * 16 + 255) / 256, 256, 0, s>>>(w, wp);
    if ((rc = check_launch("stem3d_pack_w"))) return rc;
    Stem3dArgs a;
    a.xq = xq; a.wp = wp; a.y = (bf16_t*)y; a.stat_part = stat_partials;
    a.N = N; a.OD = (T + 5 - 7) / 2 + 1; a.OH = (H + 5 - 7) / 2 + 1; a.OW = (W + 5 - 7) / 2 + 1;
    a.Tp = Tp; a.Hp = Hp; a.Wp = Wp; a.tiles_x = (a.OW + 31) / 32;
    static LdsLimit lim_attr;
    const hipError_t attr = lim_attr.raise(reinterpret_cast<const void*>(&stem3d_fwd_kernel), S3_WLDS);
    if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "stem3d: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    const long tiles = (long)N * a.OD * a.OH * a.tiles_x;
    stem3d_fwd_kernel<<<s3_blocks(tiles), S3_WAVES * 64, S3_WLDS, s>>>(a);
    return check_launch("stem3d_fwd");
}

// bytes of the weight gradient's workspace (prepared cue + partials; the larger of the two forms)
size_t dmc_stem3d_bf16_wgrad_workspace_bytes(int N, int T, int H, int W) {
    const long Wp = (W + 5 + 1 + 3) / 4 * 4;
    const size_t v1 = (size_t)N * (T + 5) * (H + 5) * Wp * 4 + (size_t)256 * S3_ROWS * S3_CO * sizeof(float) + 64;
    const size_t v2 = (size_t)N * (T + 5) * (H + 5) * 1024 + (size_t)256 * 4 * W2_MROWS * S3_CO * sizeof(float) + 64;
    return v1 > v2 ? v1 : v2;
}

// dw [64,2,7,7,7] fp32 contiguous from x [N,2,T,H,W] fp32 (rounded to bf16 as in the forward) and dy [N,OD,OH,OW,64] bf16
// NDHWC; deterministic
int dmc_stem3d_bf16_wgrad(const float* x, const void* dy, float* dw, void* workspace, int N, int T, int H, int W, dmc_stream_t stream) {
    if (!x || !dy || !dw || !workspace) return fail(DMC_E_INVALID, "dmc_stem3d_bf16_wgrad: null pointer");
    if (N <= 0 || T < 2 || H < 2 || W < 2) return fail(DMC_E_INVALID, "dmc_stem3d_bf16_wgrad: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int Tp = T + 5, Hp = H + 5, Wp = (W + 5 + 1 + 3) / 4 * 4;
    const int OWn = (W + 5 - 7) / 2 + 1;
    if (OWn + 3 <= 128 && option(OPT_CONV_CFG) != 11) {   // the plane form (rows of up to 125 output pixels); conv_cfg 11: the first form, for A/B
        unsigned* xp = (unsigned*)workspace;
        float* part2 = (float*)((char*)workspace + (size_t)N * Tp * Hp * 1024);
        const int rows = N * Tp * Hp;
        stem3d_prep_planes_kernel<<<rows > 16384 ? 16384 : rows, 256, 0, s>>>(x, xp, N, T, H, W, Tp, Hp);
        int rc2 = check_launch("stem3d_prep_planes");
        if (rc2) return rc2;
        Stem3dW2Args b;
        b.xp = (const bf16_t*)xp; b.dy = (const bf16_t*)dy; b.part = part2;
        b.N = N; b.OD = (T + 5 - 7) / 2 + 1; b.OH = (H + 5 - 7) / 2 + 1; b.OW = OWn;
        b.Tp = Tp; b.Hp = Hp; b.ksteps = (OWn + 3 + 15) / 16;
        b.nrows = (long)N * b.OD * b.OH;
        const int cus = persistent_cus(256);
        const int groups2 = (int)(b.nrows < cus ? b.nrows : cus);
        b.per_wg = (b.nrows + groups2 - 1) / groups2;
        const int used2 = (int)((b.nrows + b.per_wg - 1) / b.per_wg);
        static LdsLimit lim_w2;
        const hipError_t attr = lim_w2.raise(reinterpret_cast<const void*>(&stem3d_w2_kernel), W2_LDS);
        if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "stem3d wgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
        stem3d_w2_kernel<<<used2, 256, W2_LDS, s>>>(b);
        if ((rc2 = check_launch("stem3d_w2"))) return rc2;
        stem3d_w2_reduce_kernel<<<(S3_CO * 686 + 255) / 256, 256, 0, s>>>(part2, used2, dw);
        return check_launch("stem3d_w2_reduce");
    }
    unsigned* xq = (unsigned*)workspace;
    float* part = (float*)((char*)workspace + (((size_t)N * Tp * Hp * Wp * 4 + 63) / 64) * 64);
    const long total = (long)N * Tp * Hp * Wp;
    stem3d_prep_kernel<<<(int)(total / Wp > 16384 ? 16384 : total / Wp), 256, 0, s>>>(x, xq, N, T, H, W, Tp, Hp, Wp);
    int rc = check_launch("stem3d_prep");
    if (rc) return rc;
    Stem3dWgArgs a;
    a.xq = xq; a.dy = (const bf16_t*)dy; a.part = part;
    a.N = N; a.OD = (T + 5 - 7) / 2 + 1; a.OH = (H + 5 - 7) / 2 + 1; a.OW = (W + 5 - 7) / 2 + 1;
    a.Tp = Tp; a.Hp = Hp; a.Wp = Wp; a.tiles_x = (a.OW + 31) / 32;
    a.nseg = (long)N * a.OD * a.OH * a.tiles_x;
    const int groups = s3_wg_groups(a.nseg);
    a.per_wg = (a.nseg + groups - 1) / groups;
    const int used = (int)((a.nseg + a.per_wg - 1) / a.per_wg);
    stem3d_wgrad_kernel<<<used, 256, 0, s>>>(a);
    if ((rc = check_launch("stem3d_wgrad"))) return rc;
    stem3d_wgrad_reduce_kernel<<<(S3_CO * 686 + 255) / 256, 256, 0, s>>>(part, used, dw);
    return check_launch("stem3d_wgrad_reduce");
}

// bytes of the data gradient's workspace (packed weights)
size_t dmc_stem3d_bf16_dgrad_workspace_bytes(void) { return (size_t)S3_KB * 16 * S3_CO * 2 + 64; }

// dx [N,2,T,H,W] fp32 (the gradient of the cue) from dy [N,OD,OH,OW,64] bf16 NDHWC and w [64,2,7,7,7] fp32 contiguous (rounded to
// bf16); OW <= 128; deterministic
int dmc_stem3d_bf16_dgrad(const void* dy, const float* w, float* dx, void* workspace, int N, int T, int H, int W, dmc_stream_t stream) {
    if (!dy || !w || !dx || !workspace) return fail(DMC_E_INVALID, "dmc_stem3d_bf16_dgrad: null pointer");
    const int OW = (W + 5 - 7) / 2 + 1;
    if (N <= 0 || T < 2 || H < 2 || W < 2 || OW > 16 * S3_DG_MT)
        return fail(DMC_E_INVALID, "dmc_stem3d_bf16_dgrad: unsupported shape N=%d T=%d H=%d W=%d (W <= 256)", N, T, H, W);
    hipStream_t s = (hipStream_t)stream;
    stem3d_pack_wq_kernel<<<(S3_KB * 16 * S3_CO + 255) / 256, 256, 0, s>>>(w, (bf16_t*)workspace);
    int rc = check_launch("stem3d_pack_wq");
    if (rc) return rc;
    Stem3dDgArgs a;
    a.dy = (const bf16_t*)dy; a.wq = (const bf16_t*)workspace; a.dx = dx;
    a.N = N; a.T = T; a.H = H; a.W = W; a.OD = (T + 5 - 7) / 2 + 1; a.OH = (H + 5 - 7) / 2 + 1; a.OW = OW;
    const long rows = (long)N * T * H;
    long blocks = (rows + S3_DG_WAVES - 1) / S3_DG_WAVES;
    if (blocks > 2048) blocks = 2048;
    const int mt = (OW + 15) / 16;
    if (mt <= D3_MT && option(OPT_CONV_CFG) != 12) {          // the block form (conv_cfg 12: the row kernels, for A/B)
        const int total = N * ((T + 3) / 4) * ((H + 7) / 8);
        const int cus = persistent_cus(256);
        static LdsLimit lim_d3;
        const hipError_t attr = lim_d3.raise(reinterpret_cast<const void*>(&stem3d_dgrad_blk_kernel), D3_LDS);
        if (attr != hipSuccess) return fail(DMC_E_LAUNCH, "stem3d dgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
        stem3d_dgrad_blk_kernel<<<total < cus ? total : cus, 256, D3_LDS, s>>>(a);
        return check_launch("stem3d_dgrad_blk");
    }
    if (mt == 7 && option(OPT_CONV3D_WGRAD) != 10) {          // 224-wide frames: four input rows per wave (option value 10: one row, A/B)
        long b4 = ((long)N * T * ((H + 3) / 4) + S3_DG_WAVES - 1) / S3_DG_WAVES;
        if (b4 > 2048) b4 = 2048;
        stem3d_dgrad4_kernel<7><<<(int)b4, S3_DG_WAVES * 64, 0, s>>>(a);
    }
    else if (mt == 7) stem3d_dgrad_kernel<7><<<(int)blocks, S3_DG_WAVES * 64, 0, s>>>(a);
    else if (mt == 8) stem3d_dgrad_kernel<8><<<(int)blocks, S3_DG_WAVES * 64, 0, s>>>(a);
    else stem3d_dgrad_kernel<0><<<(int)blocks, S3_DG_WAVES * 64, 0, s>>>(a);
    return check_launch("stem3d_dgrad");
}

}  // extern "C"
