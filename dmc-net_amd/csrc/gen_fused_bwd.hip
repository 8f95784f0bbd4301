// EstimatorDenseNetTiny data gradient (all five feature groups) as ONE launch for gfx950: the scheme of gen_fused.hip run
// backwards through the dense stack.
//
// Reference behaviour: autograd of code/dmcnet/model.py:172-194 (x_{k+1} = cat(conv_k(x_k), x_k); conv = Conv2d(3x3, pad 1) +
// LeakyReLU(0.1), :111-119).  With g_5 = dL/d(out) and g_k = dL/d(pre-activation of layer k), k = 4 .. 0:
//     g_j = LeakyReLU'(y_j) (.) sum_{k > j} corr(g_k, flipped w_k[:, channels of y_j])
// -- feature group j gathers from EVERY later layer's gradient, so the chain g_5 -> g_4 -> ... -> g_0 is again a dense stack
// (stage s = 4 - j reads gin = 2 / 4 / 8 / 14 / 22 gradient planes and writes 2 / 4 / 6 / 8 / 8): 3,204 MAC per pixel.
// Nothing here is derived from reference source text; the reference has no kernels.
//
// The layer-by-layer kernels move 426 B/px for this (every group re-reads every later gradient plane: five launches); here
// each gradient plane is produced once, line-buffered in LDS for the stages that read it, and written once for the
// weight-gradient kernel: read 8 B/px (dL/dout) + 112 (the saved features: LeakyReLU' needs their signs), write 112.
//
// Scheme (gen_fused.hip has the long form): a workgroup walks a strip of <= 118 columns down the frame one row per step; MFMA
// rows are (dy, dx, cd) -- both taps PUSHED, K = input gradient channel -- with the three accumulator sets rotating by renaming
// where a set is whole tiles; the flipped weights (the WB block of the packed parameters) stay in registers as cbsz-broadcast
// A operands; stage s consumes gradient row t - 2s in step t and completes row t - 2s - 1, so the stages run concurrently on
// different waves with one barrier per step.  Rings: g_5 (staged from HBM) 10 rows x 2 planes, g_4 8 x 2, g_3 6 x 4, g_2 4 x 6,
// g_1 2 x 8, g_0 2 x 8 (for the storing wave), the saved-feature rows 2 x 28 = 172 rows of 120 floats = 82.6 KB.  Stages 4 (22 -> 8: half of all MFMAs) and 3 (14 -> 8) are each split over two
// waves by output channels (4 + 4: 9 tiles each, nothing padded), so the waves load the four SIMDs with 396 / 396 / 410 / 410
// MFMAs per step: (4a lo, 4a hi, staging) (4b lo, 4b hi, storing) (3a lo, 3a hi, 2a+1a+0a) (3b lo, 3b hi, 2b+1b+0b).
// Global memory is touched by TWO waves only: a staging wave (row t + 1 of dL/dout, and for every stage the saved-feature row
// it will complete two steps later, global -> registers -> LDS one step behind) and a storing wave (every gradient row
// completed in the previous step, LDS -> global).  The ten computing waves issue LDS reads, MFMAs and LDS writes, nothing
// else: with their own loads and stores each waited a memory round trip per step (vmcnt counts loads and stores in one
// queue) and the launch took 0.93 ms; without any memory instruction the same structure takes 0.52 ms
// (tools/ubench/gen_fused_bwd_time.hip).
// Arithmetic: exact fp32, fixed order; not bit-identical to the layer kernels (different summation order).
#include "gen_fused.h"
#include "gen_fused_inl.h"

using namespace dmc;
using namespace dmc::fz;

namespace {

// measurement only (tools/ubench/gen_fused_bwd_time.hip, -DDMC_MEASURE): no feature staging / no stores
#if defined(DMC_MEASURE) && defined(BZ_NO_FEAT)
constexpr bool BZ_ABL_FEAT = true;
#else
constexpr bool BZ_ABL_FEAT = false;
#endif
#if defined(DMC_MEASURE) && defined(BZ_NO_STORE)
constexpr bool BZ_ABL_STORE = true;
#else
constexpr bool BZ_ABL_STORE = false;
#endif
constexpr int BZ_STAGES = 5;
constexpr int BZ_HALO = BZ_STAGES;             // columns recomputed on the interior side of a strip (one per stage)
constexpr int BZ_LAG = 2 * (BZ_STAGES - 1);
constexpr int BZ_WAVES = 12, BZ_THREADS = BZ_WAVES * 64;

// ring r: 0 = g_5 (dL/dout), r = 1 .. 5: g_{5-r}; ring 5 (g_0) has no reader but the storing wave
__host__ __device__ constexpr int bz_planes(int r) { return r == 0 ? 2 : cout_of(5 - r); }
__host__ __device__ constexpr int bz_len(int r) { return r == 5 ? 2 : r == 0 ? BZ_LAG + 4 : BZ_LAG + 2 - 2 * r; }   // (ring 0: rows t - 8 .. t + 3, two in flight)
__host__ __device__ constexpr int bz_base(int r) {
    int o = 0;
    for (int i = 0; i < r; ++i) o += bz_planes(i) * bz_len(i) * FZ_RS;
    return o;
}
// saved-feature rows: [3 slots][28 planes], slot = row mod 3: one being read, one landed, one in flight
constexpr int BZ_FSLOTS = 3;
constexpr int BZ_FEAT = bz_base(BZ_STAGES + 1);
constexpr int BZ_LDS = BZ_FEAT + BZ_FSLOTS * NFEAT * FZ_RS;     // 24,480 floats = 97,920 B
// input channel c of the group-j gather (order g_{j+1}, ..., g_4, g_5: dmc_common.h WB) -> ring / plane within it
__host__ __device__ constexpr int bz_ring_of(int j, int c) {
    int k = j + 1;
    while (k < 5 && c >= cout_of(k)) { c -= cout_of(k); ++k; }
    return 5 - k;
}
__host__ __device__ constexpr int bz_plane_of(int j, int c) {
    int k = j + 1;
    while (k < 5 && c >= cout_of(k)) { c -= cout_of(k); ++k; }
    return c;
}

struct BwdArgs {
    const float* gout;    // [N,2,H,W]  dL/d(out)
    const float* feat;    // [N,28,H,W] saved features y0 .. y4
    float* gbuf;          // [N,28,H,W] g_0 .. g_4 (what the weight-gradient kernel reads)
    const float* pk;      // packed parameters (WB block)
    int H, W;
    int nstrips, sw, m, nitems;
    int vsplit, vhalo;    // (row bands: not used by this kernel -- 1, 0)
};

__device__ __forceinline__ int bz_steps(int H) { return (H + BZ_LAG + 1 + 2) / 3 * 3; }

// ------------------------------------------------------------------------------------------------------------
// Stage S (feature group j = 4 - S), output channels [CLO, CLO + C) of the group's D, as a wave sees it.
// ROT: the three accumulator sets rotate by renaming (needs 3 C % 4 == 0 and the wave to itself); else flat rows + moves.
// ------------------------------------------------------------------------------------------------------------
template <int S, int CLO, int C_, bool ROT_>
struct BzGeo {
    static constexpr int J = 4 - S, CIN = gin_of(J), D = cout_of(J), C = C_, NROW = 9 * C, NT = (NROW + 3) / 4, NA = (NT + 15) / 16;
    static constexpr bool ROT = ROT_;
    static_assert(!ROT || (3 * C) % 4 == 0, "rotating sets are whole tiles");
    static constexpr int TPS = 3 * C / 4;
    static constexpr int NEW = S == 0 ? CIN : cout_of(J + 1);   // channels [0, NEW): the row the previous stage completed a step ago (stage 0: staged a step ago)
};

template <typename G, int JP, int T, bool FIRST>
struct BzTiles {
    static __device__ __forceinline__ void run(f32x4 (&acc)[G::NT], const float (&a)[G::NA], float b) {
        constexpr int dy = G::ROT ? T / G::TPS : 0;
        constexpr int P = G::ROT ? ((JP + 3 - dy) % 3) * G::TPS + T % G::TPS : T;
        constexpr bool fresh = FIRST && (G::ROT ? dy == 0 : T < G::TPS);
        if constexpr (fresh) acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[T / 16], b, (f32x4){0.f, 0.f, 0.f, 0.f}, 4, T % 16, 0);
        else acc[P] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[T / 16], b, acc[P], 4, T % 16, 0);
        if constexpr (T + 1 < G::NT) BzTiles<G, JP, T + 1, FIRST>::run(acc, a, b);
    }
};

template <int S, int CLO, int C_, bool ROT_>
struct BzStage {
    using G = BzGeo<S, CLO, C_, ROT_>;
    static constexpr int J = G::J, CIN = G::CIN, D = G::D, C = G::C, NROW = G::NROW, NT = G::NT, NA = G::NA, NEW = G::NEW;
    static constexpr bool ROT = G::ROT;
    float A[CIN][NA];                 // lane 4 t' + i of A[c][a]: row 4 (16 a + t') + i = (dy, dx, cd - CLO) of input channel c
    f32x4 acc[NT];
    float b[CIN];                     // the step's B operands (channels [NEW, CIN) requested before the previous barrier)
    float y[C];                       // the saved feature values of the row the next step completes (their signs: LeakyReLU')

    __device__ __forceinline__ void load_weights(const float* __restrict__ pk, int lane) {
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int r = 4 * (16 * a + (lane >> 2)) + (lane & 3);
                const int dy = r / (3 * C), dx = (r / C) % 3, cd = CLO + r % C;
                A[c][a] = r < NROW ? pk[wb_off(J) + (c * 9 + dy * 3 + dx) * D + cd] : 0.f;
            }
    }
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    template <int C0, int C1>
    __device__ __forceinline__ void load_b(const float* lds, int col, int i) {
        if constexpr (C0 < C1) {
            const float* pr[BZ_STAGES];
#pragma unroll
            for (int r = 0; r <= S; ++r) pr[r] = lds + bz_base(r) + (i % bz_len(r)) * (bz_planes(r) * FZ_RS) + col;
#pragma unroll
            for (int c = C0; c < C1; ++c) b[c] = pr[bz_ring_of(J, c)][bz_plane_of(J, c) * FZ_RS];
        }
    }
    // requested before the barrier that opens step t: the operands of rows that are already complete
    __device__ __forceinline__ void prefetch(const BwdArgs& a, const Strip&, const Half& h, const float* lds, int t) {
        const int i = t - 2 * S;
        if (i >= 0 && i < a.H) load_b<NEW, CIN>(lds, h.col, i);
    }

    template <int JP>
    __device__ __forceinline__ void step(const BwdArgs& a, const Strip& st, const Half& h, float* lds, int t) {
        const int i = t - 2 * S, o = i - 1;
        const bool emit = o >= 0 && o < a.H;
        if (emit) {
            // the saved features of the row this step completes (the staging wave parked them in the previous step)
            const float* f = lds + BZ_FEAT + ((o % BZ_FSLOTS) * NFEAT + (yoff(J) - NIN) + CLO) * FZ_RS + h.col;
#pragma unroll
            for (int cd = 0; cd < C; ++cd) y[cd] = f[cd * FZ_RS];
        }
        if (i >= 0 && i < a.H) {
            load_b<0, NEW>(lds, h.col, i);
            // the channels requested a step ago first, the row that has just been completed last
            if constexpr (NEW < CIN) {
                BzTiles<G, JP, 0, true>::run(acc, A[NEW], b[NEW]);
#pragma unroll
                for (int c = NEW + 1; c < CIN; ++c) BzTiles<G, JP, 0, false>::run(acc, A[c], b[c]);
#pragma unroll
                for (int c = 0; c < NEW; ++c) BzTiles<G, JP, 0, false>::run(acc, A[c], b[c]);
            } else {
                BzTiles<G, JP, 0, true>::run(acc, A[0], b[0]);
#pragma unroll
                for (int c = 1; c < CIN; ++c) BzTiles<G, JP, 0, false>::run(acc, A[c], b[c]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < G::TPS; ++j) acc[(ROT ? (JP % 3) * G::TPS : 0) + j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (emit) {
            constexpr int S2 = ROT ? ((JP + 1) % 3) * 3 * C : 6 * C;
            float v[C];
#pragma unroll
            for (int cd = 0; cd < C; ++cd) {
                const int r0 = S2 + cd, r1 = S2 + C + cd, r2 = S2 + 2 * C + cd;
                v[cd] = acc[r1 / 4][r1 % 4] + dpp_shr0(acc[r0 / 4][r0 % 4]);
                v[cd] += dpp_shl0(acc[r2 / 4][r2 % 4]);
                v[cd] *= y[cd] > 0.f ? 1.f : 0.1f;                     // LeakyReLU'(0.1) of the saved feature
            }
            if (h.own) {
                float* ring = lds + bz_base(S + 1) + (o % bz_len(S + 1)) * (D * FZ_RS) + CLO * FZ_RS + h.col;
#pragma unroll
                for (int cd = 0; cd < C; ++cd) ring[cd * FZ_RS] = v[cd];
            }
        }
        if constexpr (!ROT) {
#pragma unroll
            for (int r = NROW - 1; r >= 3 * C; --r) acc[r / 4][r % 4] = acc[(r - 3 * C) / 4][(r - 3 * C) % 4];
#pragma unroll
            for (int r = 4 * G::TPS; r < 3 * C; ++r) acc[r / 4][r % 4] = 0.f;
        }
        prefetch(a, st, h, lds, t + 1);
    }
};

// a wave that runs one rotating stage
template <typename ST>
__device__ __forceinline__ void run_rot(const BwdArgs& a, float* lds, int hf, int lane) {
    ST s;
    s.load_weights(a.pk, lane);
    const int steps = bz_steps(a.H);
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        const Half h = half_of(a, st, hf, lane);
        s.reset();
        step_barrier();
        s.prefetch(a, st, h, lds, 0);
#pragma unroll 1
        for (int t = 0; t < steps; t += 3) {
            s.template step<0>(a, st, h, lds, t);
            step_barrier();
            s.template step<1>(a, st, h, lds, t + 1);
            step_barrier();
            s.template step<2>(a, st, h, lds, t + 2);
            step_barrier();
        }
    }
}

// a wave that runs stages 2, 1 and 0 (flat rows) one after the other
__device__ __forceinline__ void run_tail(const BwdArgs& a, float* lds, int hf, int lane) {
    BzStage<2, 0, 6, false> s2;
    BzStage<1, 0, 4, false> s1;
    BzStage<0, 0, 2, false> s0;
    s2.load_weights(a.pk, lane); s1.load_weights(a.pk, lane); s0.load_weights(a.pk, lane);
    const int steps = bz_steps(a.H);
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        const Half h = half_of(a, st, hf, lane);
        s2.reset(); s1.reset(); s0.reset();
        step_barrier();
        s2.prefetch(a, st, h, lds, 0); s1.prefetch(a, st, h, lds, 0); s0.prefetch(a, st, h, lds, 0);
#pragma unroll 1
        for (int t = 0; t < steps; ++t) {
            s2.template step<0>(a, st, h, lds, t);
            s1.template step<0>(a, st, h, lds, t);
            s0.template step<0>(a, st, h, lds, t);
            step_barrier();
        }
    }
}

// The staging wave: LDS-DMA (global_load_lds_dword: global -> LDS without registers), two steps ahead.  During step t it
// requests what step t + 2 reads -- row t + 3 of dL/dout (ring 0; stage 0 consumes row t + 2 then, this keeps one more in
// flight) and, for every stage s, the saved-feature row that stage completes in step t + 2 -- then waits until everything
// but this step's requests has landed (vmcnt counts in order) and meets the barrier.  A load has between one and two steps
// to arrive (measured: ~2.5 us under this kernel's own write traffic; a step is ~2.2 us).  Lanes beyond the strip are masked
// (their LDS columns stay zero).
constexpr int BZ_DMA = 2 * (2 + NFEAT);             // transfers per step: (2 + 28) planes x 2 column batches
static_assert(BZ_DMA <= 63, "vmcnt is a 6-bit counter");
__device__ __forceinline__ void bz_dma4(const float* plane, unsigned byte_off, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :: "v"(byte_off), "s"(plane), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void run_stage_in(const BwdArgs& a, float* lds, int lane) {
    const int steps = bz_steps(a.H);
    const unsigned hw = (unsigned)(a.H * a.W);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned lane4 = (unsigned)lane * 4u;
    // one plane row: LDS columns [0, sw) <- image columns [c0, c0 + sw) of `row`; always BOTH transfers are issued (vmcnt
    // bookkeeping), the second with no lane active when the strip has <= 64 columns
    auto row_in = [&](const float* plane, const Strip& st, int row, unsigned lds_row_float) {
        const float* sp = scalar_plane_generic(plane);
        const unsigned off = (unsigned)(row * a.W + st.c0) * 4u + lane4;
        const unsigned dst = lds0 + lds_row_float * 4u;
        if (lane < a.sw) bz_dma4(sp, off, dst);
        if (lane + 64 < a.sw) bz_dma4(sp, off + 256u, dst + 256u);
    };
    auto request = [&](const Strip& st, int t) {          // what step t + 2 reads
        const int grow = t + 3;
        if (grow < a.H) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
                row_in(a.gout + ((size_t)st.n * 2 + p) * hw, st, grow, bz_base(0) + ((grow % bz_len(0)) * 2 + p) * FZ_RS);
        }
#pragma unroll
        for (int j = 0; j < BZ_STAGES; ++j) {
            const int row = (t + 2) - 2 * (4 - j) - 1;    // the row stage 4 - j completes in step t + 2
            if (row >= 0 && row < a.H && !BZ_ABL_FEAT) {
#pragma unroll
                for (int p = yoff(j) - NIN; p < yoff(j) - NIN + cout_of(j); ++p)
                    row_in(a.feat + ((size_t)st.n * NFEAT + p) * hw, st, row, BZ_FEAT + ((row % BZ_FSLOTS) * NFEAT + p) * FZ_RS);
            }
        }
    };
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        // rows 0 .. 2 of dL/dout before the first step (request(t) covers row t + 3), then what steps 0 and 1 read
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (r < a.H) {
#pragma unroll
                for (int p = 0; p < 2; ++p) row_in(a.gout + ((size_t)st.n * 2 + p) * hw, st, r, bz_base(0) + (r * 2 + p) * FZ_RS);
            }
        request(st, -2);
        request(st, -1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        step_barrier();
#pragma unroll 1
        for (int t = 0; t < steps; ++t) {
            request(st, t);
            // everything but this step's requests has landed (a request issues at most BZ_DMA transfers; fewer near the
            // image's top and bottom, where waiting for a smaller count than needed would be wrong: wait for all then)
            const bool full = !BZ_ABL_FEAT && t + 3 < a.H && (t + 2) - 2 * 4 - 1 >= 0 && (t + 2) - 1 < a.H && a.sw > 64;
            if (full) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BZ_DMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            step_barrier();
        }
    }
}

// The storing wave.  During step t: every gradient row completed in step t - 1 (stage s: row t - 2s - 2, ring s + 1), LDS -> gbuf.
__device__ __forceinline__ void run_store_out(const BwdArgs& a, float* lds, int lane) {
    const int steps = bz_steps(a.H);
    const unsigned hw = (unsigned)(a.H * a.W);
#pragma unroll 1
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
        const Strip st = strip_of(a, item);
        const bool s0 = lane >= st.v0 && lane < st.v1, s1 = lane + 64 >= st.v0 && lane + 64 < st.v1;
        step_barrier();
#pragma unroll 1
        for (int t = 0; t < steps + 1; ++t) {
#pragma unroll
            for (int s = 0; s < BZ_STAGES; ++s) {
                const int row = t - 2 * s - 2;
                if (row >= 0 && row < a.H) {
                    const int j = 4 - s;
                    const float* src = lds + bz_base(s + 1) + (row % bz_len(s + 1)) * (bz_planes(s + 1) * FZ_RS) + lane;
                    const unsigned pix = ((unsigned)(row * a.W + st.c0) + (unsigned)lane) * 4u;
                    float* plane = a.gbuf + ((size_t)st.n * NFEAT + (yoff(j) - NIN)) * hw;
                    float v[8][2];
#pragma unroll
                    for (int cd = 0; cd < cout_of(j); ++cd) { v[cd][0] = src[cd * FZ_RS]; v[cd][1] = src[cd * FZ_RS + 64]; }
#pragma unroll
                    for (int cd = 0; cd < cout_of(j) && !BZ_ABL_STORE; ++cd) {
                        if (s0) store_at(plane + cd * hw, pix, v[cd][0]);
                        if (s1) store_at(plane + cd * hw, pix + 256u, v[cd][1]);
                    }
                }
            }
            if (t < steps) step_barrier();
        }
    }
}

__global__ __launch_bounds__(BZ_THREADS) void gen_fused_bwd_kernel(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[BZ_LDS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < BZ_LDS; i += BZ_THREADS) lds[i] = 0.f;      // (columns >= the strip width stay zero)
    __syncthreads();
    // waves w, w + 4, w + 8 share a SIMD: (4a lo, 4a hi, staging) (4b lo, 4b hi, storing) (3a lo, 3a hi, tail a) (3b lo, 3b hi, tail b)
    const int hf = wave & 1;
    switch (wave) {
        case 0: case 1: run_rot<BzStage<4, 0, 4, true>>(a, lds, hf, lane); break;
        case 4: case 5: run_rot<BzStage<4, 4, 4, true>>(a, lds, hf, lane); break;
        case 2: case 3: run_rot<BzStage<3, 0, 4, true>>(a, lds, hf, lane); break;
        case 6: case 7: run_rot<BzStage<3, 4, 4, true>>(a, lds, hf, lane); break;
        case 8: run_stage_in(a, lds, lane); break;
        case 9: run_store_out(a, lds, lane); break;
        default: run_tail(a, lds, hf, lane); break;              // 10, 11
    }
}

}  // namespace

namespace dmc {

int gen_fused_bwd_data(const float* gout, const float* feat, float* gbuf, const float* pk, int N, int H, int W, hipStream_t s) {
    if (!gen_fused_supported(H, W)) return fail(DMC_E_INVALID, "gen_fused_bwd_data: shape %d x %d not served", H, W);
    BwdArgs a;
    a.gout = gout; a.feat = feat; a.gbuf = gbuf; a.pk = pk; a.H = H; a.W = W;
    const StripGeo geo = strip_geo(N, H, W, BZ_HALO, BZ_LAG);
    a.nstrips = geo.nstrips; a.sw = geo.sw; a.m = geo.m;
    a.vsplit = 1; a.vhalo = 0;
    a.nitems = N * a.nstrips;
    const int wgs = a.nitems < fz_num_cus() ? a.nitems : fz_num_cus();
    gen_fused_bwd_kernel<<<wgs, BZ_THREADS, 0, s>>>(a);
    return check_launch("gen_fused_bwd");
}

}  // namespace dmc
