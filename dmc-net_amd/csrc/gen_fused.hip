// EstimatorDenseNetTiny forward as ONE launch for gfx950: line-buffered in LDS, the six layers pipelined across waves.
//
// Reference behaviour: code/dmcnet/model.py:172-194 (the dense stack: x_{k+1} = cat(conv_k(x_k), x_k)), :111-119 (conv =
// Conv2d(3x3, pad 1) + LeakyReLU(0.1); predict_flow = bare conv), :341-346 (cat(mv, res), + input_mv), train.py:245 (MSE
// against the flow target).  Nothing here is derived from reference source text; the reference has no kernels.
//
// Why: the layer-by-layer kernels of gen_tiny.hip move 528 B/px through HBM (every layer re-reads every earlier feature
// plane) where the algorithm needs 28 B/px + the 112 B/px of saved features, and each of them sits at 3 - 4 TB/s of
// combined traffic.  Here every feature value is produced once, kept in LDS for exactly as long as a later layer needs
// it, and written to HBM once (for the backward pass): 20 B/px read, 120 B/px written.
//
// Scheme.  A workgroup owns a vertical STRIP of one frame (<= 118 columns: the 224-wide frames are two strips of 112
// columns + 6 halo columns on the interior side, recomputed: 4.5 % of the work) and walks DOWN it one image row per step.
//   * Vertical and horizontal taps are both PUSHED: the MFMA rows of layer k are rho = (dy, dx, co), 9 Cout of them, the
//     K dimension is the input channel alone.  Consuming input row i, accumulator set S_dy receives w[.][ci][dy][dx] *
//     in[ci][i][x]: S_0 belongs to output row i + 1 (first contribution), S_1 to row i, S_2 to row i - 1, which is
//     complete after this step: out[co][x] = P_0[x-1] + P_1[x] + P_2[x+1] of its dx rows (two DPP lane shifts), then
//     S_2 <- S_1 <- S_0 <- bias.  One LDS read per (input channel, step) feeds ceil(9 Cout / 4) MFMAs
//     (v_mfma_f32_4x4x1: 4 rows x 64 pixels, one pixel per lane), row-tile efficiency 72/72, 72/72, 54/56, 36/36, 18/20,
//     18/20: 1,181 MFMAs per 64 pixels against 1,266 with gathered vertical taps.
//   * The weights live in REGISTERS for the whole launch: with cbsz = 4 the MFMA broadcasts the A values of block `abid`
//     to all 16 blocks, so ONE register holds the 4-row A operands of 16 row tiles (lane 4 t + i = row i of tile t) and
//     layer k needs Cin ceil(NT / 16) of them (10 / 26 / 21 / 27 / 31 / 33).  No weight traffic at all.
//   * Layer k consumes input row t - 2k in step t and completes output row t - 2k - 1, which layer k + 1 consumes in step
//     t + 1: every value read in a step was written in an EARLIER step, so the layers run CONCURRENTLY on different waves
//     with one barrier per step.  Ring of group g (g = 0: the 5 input planes; g = 1 .. 5: y0 .. y4) holds rows
//     [t - 10, t - 2g + 1]: 12 / 10 / 8 / 6 / 4 / 2 rows x 5 / 8 / 8 / 6 / 4 / 2 planes = 260 rows of 120 floats =
//     124.8 KB.  Rows outside the image are never stored: a layer simply skips the accumulation of an out-of-range input row.
//   * 12 waves = 6 layers x 2 pixel halves (<= 62 output columns each, 64 lanes), placed so that each SIMD carries about a
//     quarter of the 2 x 1,181 MFMAs per step (588 / 576 / 623 / 575); the twelfth wave stages the next input row.
// Arithmetic: exact fp32 (one fused multiply-add per MFMA element, fixed order): same class as the layer kernels, not
// bit-identical to them (different summation order).
#include "gen_fused.h"

using namespace dmc;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FZ_RS = 120;                     // floats per LDS row (>= strip width + 2 zero columns)
constexpr int FZ_MAXSW = 118;                  // widest strip, halo included
constexpr int FZ_HALO = 6;                     // columns recomputed on the interior side of a strip (one per layer)
constexpr int FZ_WAVES = 12, FZ_THREADS = FZ_WAVES * 64;
constexpr int FZ_LAG = 2 * (NL - 1);           // steps between a row entering layer 0 and entering layer 5

// ring groups: 0 = the five input planes (mv, res), g = 1 .. 5: y_{g-1}
__host__ __device__ constexpr int fz_planes(int g) { return g == 0 ? NIN : cout_of(g - 1); }
__host__ __device__ constexpr int fz_len(int g) { return FZ_LAG + 2 - 2 * g; }
__host__ __device__ constexpr int fz_base(int g) {
    int o = 0;
    for (int i = 0; i < g; ++i) o += fz_planes(i) * fz_len(i) * FZ_RS;
    return o;
}
constexpr int FZ_LDS = fz_base(NL);            // 31,200 floats = 124,800 B
// physical input channel p (dmc_common.h) -> ring group / plane within it
__host__ __device__ constexpr int fz_group_of(int p) {
    if (p < NIN) return 0;
    int j = 0;
    while (p >= yoff(j) + cout_of(j)) ++j;
    return j + 1;
}
__host__ __device__ constexpr int fz_plane_of(int p) { return p < NIN ? p : p - yoff(fz_group_of(p) - 1); }

struct FusedArgs {
    const float* mv;      // [N,2,H,W]
    const float* res;     // [N,3,H,W]
    float* feat;          // [N,28,H,W] or null
    float* out;           // [N,2,H,W]
    const float* pk;      // packed parameters (dmc_common.h)
    const float* flow;    // [N,2,H,W] or null
    double* mse_part;     // [gridDim.x * 2]
    int H, W, add_mv;
    int nstrips, sw, m;   // strips per frame; strip width in LDS columns; first image column the second strip owns
    int nitems;           // N * nstrips
};

// one strip of one frame, as a workgroup sees it
struct Strip {
    int n;                // frame
    int c0;               // image column of LDS column 0
    int v0, v1;           // LDS columns [v0, v1) are this strip's to store
};

__device__ __forceinline__ float dpp_shr0(float cur) {      // lane i <- cur[i-1]; lane 0 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cur), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl0(float cur) {      // lane i <- cur[i+1]; lane 63 <- 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cur), 0x130, 0xf, 0xf, false));
}
// end-of-step barrier: LDS traffic drained, global stores left in flight
__device__ __forceinline__ void step_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int T, int NT, int NA>
struct TileLoop {
    static __device__ __forceinline__ void run(f32x4 (&acc)[NT], const float (&a)[NA], float b) {
        acc[T] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[T / 16], b, acc[T], 4, T % 16, 0);
        TileLoop<T + 1, NT, NA>::run(acc, a, b);
    }
};
template <int NT, int NA>
struct TileLoop<NT, NT, NA> {
    static __device__ __forceinline__ void run(f32x4 (&)[NT], const float (&)[NA], float) {}
};

// what a wave knows about its pixel half
struct Half {
    int col;              // LDS column of this lane
    bool own;             // this lane's column is one the half produces
    bool store;           // ... and one the strip stores to HBM
};

// ------------------------------------------------------------------------------------------------------------
// Layer K as a wave sees it: weights and accumulators in registers, one step() per image row.
// ------------------------------------------------------------------------------------------------------------
template <int K>
struct FzLayer {
    static constexpr int CIN = cin_of(K), C = cout_of(K), NROW = 9 * C, NT = (NROW + 3) / 4, NA = (NT + 15) / 16;
    float A[CIN][NA];                 // lane 4 t' + i of A[ci][a]: row 4 (16 a + t') + i = (dy, dx, co) of input channel ci
    f32x4 acc[NT];                    // flat rows (dy, dx, co); element [r / 4][r % 4], one pixel per lane
    double sq;                        // K == 5: sum of squared differences to the flow target

    __device__ __forceinline__ void load_weights(const float* __restrict__ pk, int lane) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int r = 4 * (16 * a + (lane >> 2)) + (lane & 3);
                const int dy = r / (3 * C), dx = (r / C) % 3, co = r % C;
                A[ci][a] = r < NROW ? pk[wf_off(K) + (ci * 9 + dy * 3 + dx) * C + co] : 0.f;
            }
    }
    // accumulator set dy starts as the bias in its centre-tap rows
    __device__ __forceinline__ void init_set(const float* __restrict__ pk, int dy) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int co = 0; co < C; ++co) {
                const int r = dy * 3 * C + dx * C + co;
                acc[r / 4][r % 4] = dx == 1 ? pk[bf_off(K) + co] : 0.f;
            }
    }
    __device__ __forceinline__ void reset(const float* __restrict__ pk) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        init_set(pk, 0); init_set(pk, 1); init_set(pk, 2);
    }

    // step t of the strip: consume input row t - 2K, complete output row t - 2K - 1
    __device__ __forceinline__ void step(const FusedArgs& a, const Strip& st, const Half& h, float* lds, int t) {
        const int i = t - 2 * K, o = i - 1;
        const bool emit = o >= 0 && o < a.H;
        const size_t HW = (size_t)a.H * a.W;
        // layer 5: the delta input and the flow target of the row it completes, requested ahead of the MFMAs
        float mvv[2] = {0.f, 0.f}, flw[2] = {0.f, 0.f};
        if (K == NL - 1 && emit && h.store) {
            const size_t px = (size_t)o * a.W + st.c0 + h.col;
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                if (a.add_mv) mvv[co] = a.mv[((size_t)st.n * 2 + co) * HW + px];
                if (a.flow) flw[co] = a.flow[((size_t)st.n * 2 + co) * HW + px];
            }
        }
        if (i >= 0 && i < a.H) {
            const float* pg[K + 1];
#pragma unroll
            for (int g = 0; g <= K; ++g) pg[g] = lds + fz_base(g) + (i % fz_len(g)) * (fz_planes(g) * FZ_RS) + h.col;
            float b[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) b[ci] = pg[fz_group_of(ci)][fz_plane_of(ci) * FZ_RS];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) TileLoop<0, NT, NA>::run(acc, A[ci], b[ci]);
        }
        if (emit) {
            float* ring = lds + h.col;
            if constexpr (K < NL - 1) ring += fz_base(K + 1) + (o % fz_len(K + 1)) * (C * FZ_RS);
#pragma unroll
            for (int co = 0; co < C; ++co) {
                constexpr int S2 = 6 * C;
                const int r0 = S2 + co, r1 = S2 + C + co, r2 = S2 + 2 * C + co;
                // (the shifts run with all lanes active; only the stores are masked)
                float v = acc[r1 / 4][r1 % 4] + dpp_shr0(acc[r0 / 4][r0 % 4]) + dpp_shl0(acc[r2 / 4][r2 % 4]);
                if constexpr (K < NL - 1) {
                    v = v > 0.f ? v : 0.1f * v;
                    if (h.own) ring[co * FZ_RS] = v;
                    if (h.store && a.feat)
                        a.feat[((size_t)st.n * NFEAT + (yoff(K) - NIN) + co) * HW + (size_t)o * a.W + st.c0 + h.col] = v;
                } else {
                    v += mvv[co];
                    if (h.store) {
                        a.out[((size_t)st.n * 2 + co) * HW + (size_t)o * a.W + st.c0 + h.col] = v;
                        if (a.flow) { const float d = v - flw[co]; sq += (double)d * (double)d; }
                    }
                }
            }
        }
        // S_2 <- S_1 <- S_0 <- bias
#pragma unroll
        for (int r = NROW - 1; r >= 3 * C; --r) acc[r / 4][r % 4] = acc[(r - 3 * C) / 4][(r - 3 * C) % 4];
        init_set(a.pk, 0);
    }
};

__device__ __forceinline__ Strip strip_of(const FusedArgs& a, int item) {
    Strip st;
    st.n = item / a.nstrips;
    const int s = item - st.n * a.nstrips;
    if (a.nstrips == 1) { st.c0 = 0; st.v0 = 0; st.v1 = a.W; }
    else if (s == 0) { st.c0 = 0; st.v0 = 0; st.v1 = a.m; }
    else { st.c0 = a.W - a.sw; st.v0 = a.sw - (a.W - a.m); st.v1 = a.sw; }
    return st;
}

// pixel half hf of a strip sw columns wide: half 0 = LDS columns [0, 64), produces [0, ha); half 1 = columns [sw - 62, sw + 2),
// produces [ha, sw) (columns sw, sw + 1 are never written: zeros).  A strip of <= 62 columns has no second half.
__device__ __forceinline__ Half half_of(const FusedArgs& a, const Strip& st, int hf, int lane) {
    const int ha = a.sw <= 62 ? a.sw : (a.sw + 1) / 2;
    Half h;
    h.col = hf == 0 ? lane : a.sw - 62 + lane;
    if (a.sw <= 62 && hf == 1) h.col = lane;                 // (idle half: reads valid LDS, produces nothing)
    h.own = hf == 0 ? h.col < ha : (a.sw > 62 && h.col >= ha && h.col < a.sw);
    h.store = h.own && h.col >= st.v0 && h.col < st.v1;
    return h;
}

// a wave that runs layer KA (and, when KB >= 0, layer KB after it) on pixel half hf
template <int KA, int KB>
__device__ __forceinline__ void run_layers(const FusedArgs& a, float* lds, int hf, int lane) {
    FzLayer<KA> la;
    FzLayer<(KB >= 0 ? KB : 0)> lb;
    la.load_weights(a.pk, lane);
    la.sq = 0.0;
    if (KB >= 0) { lb.load_weights(a.pk, lane); lb.sq = 0.0; }
    const int steps = a.H + FZ_LAG + 1;
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        const Half h = half_of(a, st, hf, lane);
        la.reset(a.pk);
        if (KB >= 0) lb.reset(a.pk);
        step_barrier();                                        // input row 0 is staged
#pragma unroll 1
        for (int t = 0; t < steps; ++t) {
            la.step(a, st, h, lds, t);
            if (KB >= 0) lb.step(a, st, h, lds, t);
            step_barrier();
        }
    }
    if (KA == NL - 1 && a.flow) {
        double s = la.sq;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) a.mse_part[blockIdx.x * 2 + hf] = s;
    }
}

// the staging wave: input row t + 1 of the five input planes -> ring group 0, during step t
__device__ __forceinline__ void run_loader(const FusedArgs& a, float* lds, int lane) {
    const int steps = a.H + FZ_LAG + 1;
    const size_t HW = (size_t)a.H * a.W;
    auto stage = [&](const Strip& st, int row) {
        if (row >= a.H) return;
        float v[NIN][2];
#pragma unroll
        for (int p = 0; p < NIN; ++p) {
            const float* src = (p < 2 ? a.mv + ((size_t)st.n * 2 + p) * HW : a.res + ((size_t)st.n * 3 + (p - 2)) * HW) +
                               (size_t)row * a.W + st.c0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int col = lane + 64 * q;
                v[p][q] = col < a.sw ? src[col] : 0.f;
            }
        }
        float* dst = lds + fz_base(0) + (row % fz_len(0)) * (NIN * FZ_RS);
#pragma unroll
        for (int p = 0; p < NIN; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int col = lane + 64 * q;
                if (col < a.sw) dst[p * FZ_RS + col] = v[p][q];
            }
    };
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        stage(st, 0);
        step_barrier();
#pragma unroll 1
        for (int t = 0; t < steps; ++t) {
            stage(st, t + 1);
            step_barrier();
        }
    }
}

__global__ __launch_bounds__(FZ_THREADS) void gen_fused_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[FZ_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < FZ_LDS; i += FZ_THREADS) lds[i] = 0.f;      // (columns >= the strip width stay zero)
    __syncthreads();
    // wave w runs on SIMD w % 4: MFMAs per step and SIMD 588 (layer 2 twice) / 576 / 623 / 575
    switch (wave) {
        case 0: run_layers<2, -1>(a, lds, 0, lane); break;
        case 4: run_layers<2, -1>(a, lds, 1, lane); break;
        case 8: run_loader(a, lds, lane); break;
        case 1: run_layers<3, -1>(a, lds, 0, lane); break;
        case 5: run_layers<3, -1>(a, lds, 1, lane); break;
        case 9: run_layers<0, -1>(a, lds, 0, lane); break;
        case 2: run_layers<1, -1>(a, lds, 0, lane); break;
        case 6: run_layers<1, -1>(a, lds, 1, lane); break;
        case 10: run_layers<4, -1>(a, lds, 0, lane); break;
        case 3: run_layers<5, -1>(a, lds, 0, lane); break;
        case 7: run_layers<5, -1>(a, lds, 1, lane); break;
        default: run_layers<4, 0>(a, lds, 1, lane); break;
    }
}

int fz_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}

}  // namespace

namespace dmc {

bool gen_fused_supported(int H, int W) { return H >= 1 && W >= 1 && W <= 2 * (FZ_MAXSW - FZ_HALO); }

int gen_fused_max_partials() { return 2 * fz_num_cus(); }

int gen_fused_fwd(const float* mv, const float* res, float* feat, float* out, const float* pk, const float* flow,
                  double* mse_part, int* nparts, int N, int H, int W, int add_mv, hipStream_t s) {
    if (!gen_fused_supported(H, W)) return fail(DMC_E_INVALID, "gen_fused_fwd: shape %d x %d not served", H, W);
    FusedArgs a;
    a.mv = mv; a.res = res; a.feat = feat; a.out = out; a.pk = pk;
    a.flow = flow && mse_part ? flow : nullptr;
    a.mse_part = mse_part;
    a.H = H; a.W = W; a.add_mv = add_mv;
    if (W <= FZ_MAXSW) { a.nstrips = 1; a.sw = W; a.m = W; }
    else { a.nstrips = 2; a.m = (W + 1) / 2; a.sw = a.m + FZ_HALO; }
    a.nitems = N * a.nstrips;
    const int wgs = a.nitems < fz_num_cus() ? a.nitems : fz_num_cus();
    if (nparts) *nparts = a.flow ? 2 * wgs : 0;
    gen_fused_kernel<<<wgs, FZ_THREADS, 0, s>>>(a);
    return check_launch("gen_fused");
}

}  // namespace dmc
