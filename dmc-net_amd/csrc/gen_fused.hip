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
//     layer k needs Cin ceil(NT / 16) of them (10 / 26 / 21 / 27 / 31 / 33), read once per wave straight from the twelve PyTorch
//     parameter tensors (the reference's prepend channel order is mapped on the fly: no repack launch).  No weight traffic at all.
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
#include "gen_fused_inl.h"

using namespace dmc;
using namespace dmc::fz;

namespace {

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
    ParamPtrs prm;        // the twelve parameter tensors as PyTorch holds them (w[k]: [Cout][Cin_logical][3][3]): read once per wave
    const float* flow;    // [N,2,H,W] or null
    double* mse_part;     // [gridDim.x * 2]
    int H, W, add_mv;
    int nstrips, sw, m;   // strips per frame; strip width in LDS columns; first image column the second strip owns
    int vsplit, vhalo;    // row bands per strip; rows a band recomputes above / below
    int nitems;           // N * nstrips * vsplit
#ifdef DMC_MEASURE
    int feat_one_frame;           // every frame's features land in frame 0's planes (isolates the cost of the store INSTRUCTIONS from HBM write traffic)
    unsigned long long* prof;     // [gridDim.x][12 waves][4]: busy clocks, total clocks, HW_ID, steps (tools/ubench/gen_fused_prof.hip)
#endif
};

#if defined(DMC_MEASURE) && !defined(FZ_NOPROF)
struct FzProf {
    unsigned long long busy = 0, t_first = 0, t_mark = 0, steps = 0;
    __device__ __forceinline__ void begin() { t_mark = __builtin_amdgcn_s_memtime(); if (!t_first) t_first = t_mark; }
    __device__ __forceinline__ void end() { busy += __builtin_amdgcn_s_memtime() - t_mark; ++steps; }
    __device__ __forceinline__ void flush(const FusedArgs& a, int wave, int lane) {
        if (!a.prof || lane) return;
        unsigned long long* q = a.prof + ((size_t)blockIdx.x * 12 + wave) * 4;
        q[0] = busy; q[1] = __builtin_amdgcn_s_memtime() - t_first; q[2] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); q[3] = steps;
    }
};
#else
struct FzProf {
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void end() {}
    __device__ __forceinline__ void flush(const FusedArgs&, int, int) {}
};
#endif

#ifdef DMC_MEASURE
__device__ __forceinline__ int fz_feat_frame(const FusedArgs& a, int n) { return a.feat_one_frame ? 0 : n; }
#else
__device__ __forceinline__ int fz_feat_frame(const FusedArgs&, int n) { return n; }
#endif

// steps per work item of H window rows: H + 11 (the last output row leaves layer 5 in step H + 10), rounded up to whole phase
// triples; the extra steps find every row out of range and only meet the barrier
__device__ __forceinline__ int fz_steps(int H) { return (H + FZ_LAG + 1 + 2) / 3 * 3; }

// ------------------------------------------------------------------------------------------------------------
// Layer K as a wave sees it: weights and accumulators in registers, one step() per image row.
//
// Accumulator rows are (set, dx, co).  Where a set (3 Cout rows) is a whole number of 4-row tiles (Cout 8 and 4) the three
// sets ROTATE BY RENAMING: in phase J = step mod 3 the set that receives tap dy is physical set (J - dy) mod 3, the step
// code exists three times and no accumulator ever moves.  Otherwise (Cout 6, 2) the rows stay flat (dy, dx, co) and are
// shifted by 3 Cout rows with register moves after each step.  A set starts its life with the first MFMA of a step taking
// a ZERO C operand (whole fresh tiles; the rows of a shared tile are zeroed by the shift), the bias joins in the epilogue.
// ------------------------------------------------------------------------------------------------------------
template <int K>
struct FzG {
    static constexpr int CIN = cin_of(K), C = cout_of(K), NROW = 9 * C, NT = (NROW + 3) / 4, NA = (NT + 15) / 16;
    static constexpr bool ROT = (3 * C) % 4 == 0;
    static constexpr int TPS = 3 * C / 4;              // ROT: tiles per set; flat: whole tiles at the head of set 0
    static constexpr int OLD = K == 0 ? 0 : yoff(K - 1);   // input channels [0, OLD) were complete a step ago: prefetched across the barrier
};

template <int K, int J, int T, bool FIRST>
struct FzTiles {
    using G = FzG<K>;
    static __device__ __forceinline__ void run(f32x4 (&acc)[G::NT], const float (&a)[G::NA], float b) {
        constexpr int dy = G::ROT ? T / G::TPS : 0;
        constexpr int P = G::ROT ? ((J + 3 - dy) % 3) * G::TPS + T % G::TPS : T;
        constexpr bool fresh = FIRST && (G::ROT ? dy == 0 : T < G::TPS);
        if constexpr (fresh) acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[T / 16], b, (f32x4){0.f, 0.f, 0.f, 0.f}, 4, T % 16, 0);
        else acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[T / 16], b, acc[P], 4, T % 16, 0);
        if constexpr (T + 1 < G::NT) FzTiles<K, J, T + 1, FIRST>::run(acc, a, b);
    }
};

template <int K>
struct FzLayer {
    using G = FzG<K>;
    static constexpr int CIN = G::CIN, C = G::C, NROW = G::NROW, NT = G::NT, NA = G::NA, OLD = G::OLD;
    float A[CIN][NA];                 // lane 4 t' + i of A[ci][a]: row 4 (16 a + t') + i = (dy, dx, co) of input channel ci
    f32x4 acc[NT];                    // one pixel per lane
    float b[CIN];                     // the step's B operands (channels [0, OLD) requested before the previous barrier)
    float bias[C];
    double sq;                        // K == 5: sum of squared differences to the flow target
    static constexpr bool ROT = G::ROT;

    __device__ __forceinline__ void load_weights(const ParamPtrs& prm, int lane) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int r = 4 * (16 * a + (lane >> 2)) + (lane & 3);
                const int dy = r / (3 * C), dx = (r / C) % 3, co = r % C;
                // (physical channel ci of this kernel = logical channel logical_of(K, ci) of the reference's prepend order)
                A[ci][a] = r < NROW ? prm.w[K][(co * CIN + logical_of(K, ci)) * 9 + dy * 3 + dx] : 0.f;
            }
#pragma unroll
        for (int co = 0; co < C; ++co) bias[co] = prm.b[K][co];
    }
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // B operands of input row i, channels [C0, C1)
    template <int C0, int C1>
    __device__ __forceinline__ void load_b(const float* lds, int col, int i) {
        if constexpr (C0 < C1) {
            const float* pg[K + 1];
#pragma unroll
            for (int g = fz_group_of(C0); g <= fz_group_of(C1 - 1); ++g)
                pg[g] = lds + fz_base(g) + (i % fz_len(g)) * (fz_planes(g) * FZ_RS) + col;
#pragma unroll
            for (int ci = C0; ci < C1; ++ci) b[ci] = pg[fz_group_of(ci)][fz_plane_of(ci) * FZ_RS];
        }
    }
    // requested before the barrier that opens step t: the B operands whose rows are already complete and, for layer 5, the
    // delta input and the flow target of the row that step completes (a global load takes about a step under load)
    float mvv[2], flw[2];
    __device__ __forceinline__ void prefetch(const FusedArgs& a, const Strip& st, const Half& h, const float* lds, int t) {
        const int i = t - 2 * K, o = i - 1;
        if (i >= st.w0 && i < st.w1) load_b<0, OLD>(lds, h.col, i);
        if constexpr (K == NL - 1) {
            mvv[0] = mvv[1] = flw[0] = flw[1] = 0.f;
            if (o >= st.s0 && o < st.s1 && h.store) {
                const unsigned hw = (unsigned)(a.H * a.W);
                const size_t plane = (size_t)st.n * 2 * hw;
                const unsigned pix = ((unsigned)(o * a.W + st.c0) + h.ucol) * 4u;
#pragma unroll
                for (int co = 0; co < 2; ++co) {
                    if (a.add_mv) mvv[co] = load_at(a.mv + plane + co * hw, pix);
                    if (a.flow) flw[co] = load_at(a.flow + plane + co * hw, pix);
                }
            }
        }
    }

    // step t of the strip (phase J = t mod 3 for the rotating layers): consume input row t - 2K, complete output row t - 2K - 1
    template <int J>
    __device__ __forceinline__ void step(const FusedArgs& a, const Strip& st, const Half& h, float* lds, int t) {
        const int i = t - 2 * K, o = i - 1;
        const bool emit = o >= st.w0 && o < st.w1, keep = o >= st.s0 && o < st.s1;      // complete a row of the window / one the item stores
        if (i >= st.w0 && i < st.w1) {
            load_b<OLD, CIN>(lds, h.col, i);
            FzTiles<K, J, 0, true>::run(acc, A[0], b[0]);
#pragma unroll
            for (int ci = 1; ci < CIN; ++ci) FzTiles<K, J, 0, false>::run(acc, A[ci], b[ci]);
        } else {
            // nothing enters the fresh set: it is zero
#pragma unroll
            for (int j = 0; j < G::TPS; ++j) acc[(G::ROT ? (J % 3) * G::TPS : 0) + j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (emit) {
            constexpr int S2 = G::ROT ? ((J + 1) % 3) * 3 * C : 6 * C;            // the set that has seen all three taps
            float v[C];
#pragma unroll
            for (int co = 0; co < C; ++co) {
                const int r0 = S2 + co, r1 = S2 + C + co, r2 = S2 + 2 * C + co;
                // (the shifts run with all lanes active; only the stores are masked)
                v[co] = acc[r1 / 4][r1 % 4] + dpp_shr0(acc[r0 / 4][r0 % 4]);
                v[co] += dpp_shl0(acc[r2 / 4][r2 % 4]);
                v[co] += bias[co];
                if constexpr (K < NL - 1) v[co] = fmaxf(v[co], 0.1f * v[co]);     // LeakyReLU(0.1)
                else v[co] += mvv[co];
            }
            // (uniform row pointers: the stores take a scalar base and one 32-bit lane offset)
            // (scalar plane bases that do not change with the step + one 32-bit lane offset that does)
            const unsigned hw = (unsigned)(a.H * a.W), pix = ((unsigned)(o * a.W + st.c0) + h.ucol) * 4u;
            if constexpr (K < NL - 1) {
                if (h.own) {
                    float* ring = lds + fz_base(K + 1) + (o % fz_len(K + 1)) * (C * FZ_RS) + h.col;
#pragma unroll
                    for (int co = 0; co < C; ++co) ring[co * FZ_RS] = v[co];
                }
                if (h.store && keep && a.feat) {
                    float* plane = a.feat + ((size_t)fz_feat_frame(a, st.n) * NFEAT + (yoff(K) - NIN)) * hw;
#pragma unroll
                    for (int co = 0; co < C; ++co) store_at(plane + co * hw, pix, v[co]);
                }
            } else if (h.store && keep) {
                float* plane = a.out + (size_t)st.n * 2 * hw;
#pragma unroll
                for (int co = 0; co < C; ++co) {
                    store_at(plane + co * hw, pix, v[co]);
                    if (a.flow) { const float d = v[co] - flw[co]; sq += (double)d * (double)d; }
                }
            }
        }
        if constexpr (!G::ROT) {
            // S_2 <- S_1 <- S_0; the rows of set 0 that share a tile with set 1 restart at zero
#pragma unroll
            for (int r = NROW - 1; r >= 3 * C; --r) acc[r / 4][r % 4] = acc[(r - 3 * C) / 4][(r - 3 * C) % 4];
#pragma unroll
            for (int r = 4 * G::TPS; r < 3 * C; ++r) acc[r / 4][r % 4] = 0.f;
        }
        prefetch(a, st, h, lds, t + 1);
    }
};

// Layer 0 (5 -> 8) with GATHERED vertical taps: K = (ci, dy), rows (dx, co) = 6 tiles -- the same 90 MFMAs as the push form
// with 24 accumulator registers instead of 72 and nothing to rotate, so that it can share a wave with another layer.
// Output row t - 1 reads input rows t - 2 .. t (the row the staging wave delivered before this step's opening barrier).
template <>
struct FzLayer<0> {
    static constexpr int CIN = NIN, C = 8, NT = 6;
    static constexpr bool ROT = false;
    float A[CIN * 3];                 // lane 4 t + i of A[ci * 3 + dy]: row 4 t + i = (dx, co)
    f32x4 acc[NT];
    float bias[C];
    double sq;

    __device__ __forceinline__ void load_weights(const ParamPtrs& prm, int lane) {
#pragma unroll
        for (int q = 0; q < CIN * 3; ++q) {
            const int r = lane, dx = r / C, co = r % C;
            A[q] = r < 3 * C ? prm.w[0][(co * CIN + logical_of(0, q / 3)) * 9 + (q % 3) * 3 + dx] : 0.f;
        }
#pragma unroll
        for (int co = 0; co < C; ++co) bias[co] = prm.b[0][co];
    }
    __device__ __forceinline__ void reset() {}
    __device__ __forceinline__ void prefetch(const FusedArgs&, const Strip&, const Half&, const float*, int) {}

    template <int T, bool FRESH>
    __device__ __forceinline__ void tiles(float a, float b) {
        if constexpr (FRESH) acc[T] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, (f32x4){0.f, 0.f, 0.f, 0.f}, 4, T, 0);
        else acc[T] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[T], 4, T, 0);
        if constexpr (T + 1 < NT) tiles<T + 1, FRESH>(a, b);
    }
    template <int DY, bool FRESH>
    __device__ __forceinline__ void tap_row(const float (&b)[3][CIN]) {
        tiles<0, FRESH>(A[DY], b[DY][0]);
#pragma unroll
        for (int ci = 1; ci < CIN; ++ci) tiles<0, false>(A[ci * 3 + DY], b[DY][ci]);
    }
    template <int J>
    __device__ __forceinline__ void step(const FusedArgs& a, const Strip& st, const Half& h, float* lds, int t) {
        const int o = t - 1;
        if (o < st.w0 || o >= st.w1) return;
        // all fifteen operands first (a row outside the image is read from the nearest one inside and not used)
        float b[3][CIN];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            int r = o + dy - 1;
            r = r < st.w0 ? st.w0 : r >= st.w1 ? st.w1 - 1 : r;
            const float* p = lds + fz_base(0) + (r % fz_len(0)) * (NIN * FZ_RS) + h.col;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) b[dy][ci] = p[ci * FZ_RS];
        }
        __builtin_amdgcn_sched_barrier(0);           // (the compiler otherwise sinks each read to its MFMA, one exposed LDS latency per pair)
        tap_row<1, true>(b);
        if (o - 1 >= st.w0) tap_row<0, false>(b);
        if (o + 1 < st.w1) tap_row<2, false>(b);
        float v[C];
#pragma unroll
        for (int co = 0; co < C; ++co) {
            const int r0 = co, r1 = C + co, r2 = 2 * C + co;
            v[co] = acc[r1 / 4][r1 % 4] + dpp_shr0(acc[r0 / 4][r0 % 4]);
            v[co] += dpp_shl0(acc[r2 / 4][r2 % 4]);
            v[co] += bias[co];
            v[co] = fmaxf(v[co], 0.1f * v[co]);
        }
        if (h.own) {
            float* ring = lds + fz_base(1) + (o % fz_len(1)) * (C * FZ_RS) + h.col;
#pragma unroll
            for (int co = 0; co < C; ++co) ring[co * FZ_RS] = v[co];
        }
        if (h.store && o >= st.s0 && o < st.s1 && a.feat) {
            const unsigned hw = (unsigned)(a.H * a.W), pix = ((unsigned)(o * a.W + st.c0) + h.ucol) * 4u;
            float* plane = a.feat + (size_t)fz_feat_frame(a, st.n) * NFEAT * hw;
#pragma unroll
            for (int co = 0; co < C; ++co) store_at(plane + co * hw, pix, v[co]);
        }
    }
};

// a wave that runs layer KA on pixel half hf (and, when KB >= 0, layer KB on half hfb after it)
template <int KA, int KB>
__device__ __forceinline__ void run_layers(const FusedArgs& a, float* lds, int hf, int hfb, int lane, int wave) {
    FzProf prof;
    FzLayer<KA> la;
    FzLayer<(KB >= 0 ? KB : 0)> lb;
    la.load_weights(a.prm, lane);
    la.sq = 0.0;
    if (KB >= 0) { lb.load_weights(a.prm, lane); lb.sq = 0.0; }
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        const Half h = half_of(a, st, hf, lane), hb = half_of(a, st, hfb, lane);
        const int t0 = st.w0, steps = fz_steps(st.w1 - st.w0);        // step t0 + k consumes the window's row k in layer 0
        la.reset();
        if (KB >= 0) lb.reset();
        step_barrier();                                        // the window's first input row is staged
        la.prefetch(a, st, h, lds, t0);
        if (KB >= 0) lb.prefetch(a, st, hb, lds, t0);
        if constexpr (KB < 0 && FzLayer<KA>::ROT) {
            // a rotating layer: three steps per trip, one per phase (straight-line code: the accumulators never meet a phi)
#pragma unroll 1
            for (int t = t0; t < t0 + steps; t += 3) {
                prof.begin(); la.template step<0>(a, st, h, lds, t); prof.end();
                step_barrier();
                prof.begin(); la.template step<1>(a, st, h, lds, t + 1); prof.end();
                step_barrier();
                prof.begin(); la.template step<2>(a, st, h, lds, t + 2); prof.end();
                step_barrier();
            }
        } else {
            static_assert(KB < 0 || (!FzLayer<KA>::ROT && !FzLayer<(KB >= 0 ? KB : 0)>::ROT), "a rotating layer has its wave to itself");
#pragma unroll 1
            for (int t = t0; t < t0 + steps; ++t) {
                prof.begin();
                la.template step<0>(a, st, h, lds, t);
                if (KB >= 0) lb.template step<0>(a, st, hb, lds, t);
                prof.end();
                step_barrier();
            }
        }
    }
    prof.flush(a, wave, lane);
    if (KA == NL - 1 && a.flow) {
        double s = la.sq;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) a.mse_part[blockIdx.x * 2 + hf] = s;
    }
}

// the staging wave: input row t + 1 of the five input planes -> ring group 0 during step t; its global loads were issued
// a step earlier (the values wait in registers across the barrier).  Branch-free: lanes beyond the strip re-read its last
// column and do not write.
__device__ __forceinline__ void run_loader(const FusedArgs& a, float* lds, int lane, int wave) {
    FzProf prof;
    const unsigned hw = (unsigned)(a.H * a.W);
    float v[NIN][2];
    const unsigned c[2] = {(unsigned)min(lane, a.sw - 1) * 4u, (unsigned)min(lane + 64, a.sw - 1) * 4u};
    auto request = [&](const Strip& st, int row) {
        if (row >= st.w1) return;
        const unsigned rowoff = (unsigned)(row * a.W + st.c0) * 4u;
#pragma unroll
        for (int p = 0; p < NIN; ++p) {
            const float* plane = p < 2 ? a.mv + ((size_t)st.n * 2 + p) * hw : a.res + ((size_t)st.n * 3 + (p - 2)) * hw;
#pragma unroll
            for (int q = 0; q < 2; ++q) v[p][q] = load_at(plane, rowoff + c[q]);
        }
    };
    auto park = [&](const Strip& st, int row) {
        if (row >= st.w1) return;
        float* dst = lds + fz_base(0) + (row % fz_len(0)) * (NIN * FZ_RS) + lane;
        if (lane < a.sw) {
#pragma unroll
            for (int p = 0; p < NIN; ++p) dst[p * FZ_RS] = v[p][0];
        }
        if (lane + 64 < a.sw) {
#pragma unroll
            for (int p = 0; p < NIN; ++p) dst[p * FZ_RS + 64] = v[p][1];
        }
    };
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        const int t0 = st.w0, steps = fz_steps(st.w1 - st.w0);
        request(st, t0);
        park(st, t0);
        request(st, t0 + 1);
        step_barrier();
#pragma unroll 1
        for (int t = t0; t < t0 + steps; ++t) {
            prof.begin();
            park(st, t + 1);
            request(st, t + 2);
            prof.end();
            step_barrier();
        }
    }
    prof.flush(a, wave, lane);
}

__global__ __launch_bounds__(FZ_THREADS) void gen_fused_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[FZ_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < FZ_LDS; i += FZ_THREADS) lds[i] = 0.f;      // (columns >= the strip width stay zero)
    __syncthreads();
    // waves w, w + 4, w + 8 share a SIMD (measured: HW_ID of tools/ubench/gen_fused_prof.hip).  Cost of a wave's step = 8 clocks per
    // MFMA + 4 per other vector instruction (they do not overlap); per SIMD: layer 2 twice + staging / 3, 3, 4 / 1, 1, 4 / 5, 5, 0 + 0
    switch (wave) {
        case 0: case 4: run_layers<2, -1>(a, lds, wave >> 2, 0, lane, wave); break;
        case 8: run_loader(a, lds, lane, wave); break;
        case 1: case 5: run_layers<3, -1>(a, lds, wave >> 2, 0, lane, wave); break;
        case 9: run_layers<4, -1>(a, lds, 0, 0, lane, wave); break;
        case 2: case 6: run_layers<1, -1>(a, lds, wave >> 2, 0, lane, wave); break;
        case 10: run_layers<4, -1>(a, lds, 1, 0, lane, wave); break;
        case 3: case 7: run_layers<5, -1>(a, lds, wave >> 2, 0, lane, wave); break;
        default: run_layers<0, 0>(a, lds, 0, 1, lane, wave); break;       // layer 0, both halves
    }
}

}  // namespace

#ifdef DMC_MEASURE
static int g_fz_feat_one_frame = 0;
static unsigned long long* g_fz_prof = nullptr;      // measurement builds only (tools/ubench/gen_fused_prof.hip sets it)
#endif

namespace dmc {

bool gen_fused_supported(int H, int W) { return H >= 1 && W >= 1 && W <= 2 * (FZ_MAXSW - FZ_HALO); }

int gen_fused_max_partials() { return 2 * fz_num_cus_hw(); }

int gen_fused_fwd(const float* mv, const float* res, float* feat, float* out, const ParamPtrs& prm, const float* flow,
                  double* mse_part, int* nparts, int N, int H, int W, int add_mv, hipStream_t s) {
    if (!gen_fused_supported(H, W)) return fail(DMC_E_INVALID, "gen_fused_fwd: shape %d x %d not served", H, W);
    FusedArgs a;
    a.mv = mv; a.res = res; a.feat = feat; a.out = out; a.prm = prm;
    a.flow = flow && mse_part ? flow : nullptr;
    a.mse_part = mse_part;
    a.H = H; a.W = W; a.add_mv = add_mv;
    const StripGeo geo = strip_geo(N, H, W, FZ_HALO, FZ_LAG);
    a.nstrips = geo.nstrips; a.sw = geo.sw; a.m = geo.m;
    a.vsplit = geo.vsplit; a.vhalo = FZ_HALO;
    a.nitems = N * a.nstrips * a.vsplit;
#ifdef DMC_MEASURE
    a.prof = g_fz_prof;
    a.feat_one_frame = g_fz_feat_one_frame;
#endif
    const int wgs = a.nitems < fz_num_cus() ? a.nitems : fz_num_cus();
    if (nparts) *nparts = a.flow ? 2 * wgs : 0;
    gen_fused_kernel<<<wgs, FZ_THREADS, 0, s>>>(a);
    return check_launch("gen_fused");
}

}  // namespace dmc
