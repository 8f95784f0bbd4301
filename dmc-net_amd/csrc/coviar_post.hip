// Post-decode motion-vector / residual extraction: the three integer loops of the reference's
// data loader, code/dmcnet/data_loader/coviar_data_loader.c:71-175, on arrays the decoder hands over.
//
// The reference walks the AVMotionVector list in order and writes the pixels of every block; where blocks overlap the
// LATER vector wins.  On the device that order becomes a per-pixel "owner": the largest index among the vectors whose
// block covers the pixel and passes the reference's four bounds tests -- the maximum of a set does not depend on the
// order of the updates, so the result is the sequential loop's, bit for bit.  The owner maps are resolved tile by tile
// in LDS (mv_owner_tile_kernel below); a second pass reads each pixel's owner and does what the reference's innermost
// statement does for it.
//
// A whole accumulated chain needs no ping-pong between frames: accu_t[p] = accu_{t-1}[p + (src - dst) of p's owner in
// frame t] (or accu_{t-1}[p] when no vector covers p; :101-105 reads accu_src_old, writes accu_src, and :125-127 copies
// back), and accu_0 is the identity (:311-318) -- so accu_T[p] is the position reached by walking p back through the
// owner maps of frames T, T-1, ..., 1.  dmc_mv_gop_batch builds the owner maps of every frame of every chain in one
// launch and walks every pixel back in a second one.
#include "dmc_common.h"

namespace {

using namespace dmc;

// AVMotionVector (libavutil/motion_vector.h, public ABI): int32 source @0, uint8 w @4, h @5, int16 src_x @6, src_y @8,
// dst_x @10, dst_y @12, uint64 flags @16 [, int32 motion_x @24, motion_y @28, uint16 motion_scale @32].  sizeof = 24
// before libavutil 55.63 and 40 since; the caller passes its own sizeof.
struct MvRec {
    int w, h, sx, sy, dx, dy, source;
};
// records are 8-byte aligned (sizeof 24 / 40): the 14 bytes that matter are read as four dwords (one 16-byte load; a scalar
// load when the index is wave-uniform) instead of seven byte / short loads
__device__ __forceinline__ MvRec load_mv(const unsigned char* __restrict__ mvs, int stride, int i) {
    const uint2* p = reinterpret_cast<const uint2*>(mvs + (size_t)i * stride);
    const uint2 lo = p[0], hi = p[1];
    MvRec r;
    r.source = (int)lo.x;
    r.w = (int)(lo.y & 0xffu);
    r.h = (int)((lo.y >> 8) & 0xffu);
    r.sx = (short)(lo.y >> 16);
    r.sy = (short)(hi.x & 0xffffu);
    r.dx = (short)(hi.x >> 16);
    r.dy = (short)(hi.y & 0xffffu);
    return r;
}

// a pixel's entry in the maps the owner pass writes: the displacement (src - dst) of the LAST vector that covers it
constexpr int DISP_NONE = (int)0x80008000u;                // (-32768, -32768): no displacement between two in-frame positions
__device__ __forceinline__ int pack_disp(int ddx, int ddy) { return (int)(((unsigned)ddx & 0xffffu) | ((unsigned)ddy << 16)); }
__device__ __forceinline__ int disp_x(int d) { return (int)(short)(d & 0xffff); }
__device__ __forceinline__ int disp_y(int d) { return d >> 16; }

// Owner pass, tile by tile in LDS.  (The first form did one GLOBAL atomicMax per (vector, block pixel) into owner maps that do
// not fit the L2 -- 105 M read-modify-writes against HBM for the 120-chain batch, 240 us plus a 38 us clear.)  A workgroup owns
// one 32 x 128 pixel tile of one frame (16 KB of LDS; 0.278 ms per batch against 0.291 for 64 x 64, 0.332 for 64 x 32 and 0.367 for
// 32 x 32: fewer tiles re-read the vector list, and 340 columns waste less of a 32-wide tile): its four waves walk the frame's vector list 64 vectors at a time (one record per lane;
// vectors that miss the tile are dropped by a ballot), then the lanes of a wave cover the pixels of each block that hits the
// tile and do the atomicMax in LDS, and the finished tile -- -1 where no vector
// landed -- is written out with coalesced stores: every owner word is written once, nothing is cleared, no global atomics.
// A vector is looked at by every tile of its frame (22 tiles at 340 x 256) and
// rasterised by the 1 - 4 tiles it overlaps.  frame_off [n_frames + 1]: vector index ranges per frame (NULL: one frame, all
// n_mv vectors); the owner plane of frame f is owner + f * H * W, row-major [y][x].  The block bounds are the reference's
// (-1 * w / 2 .. w / 2: C integer division, so an odd w covers 2 * (w / 2) columns), :91-92.
#ifndef DMC_OT_W
#define DMC_OT_W 32
#define DMC_OT_H 128
#endif
constexpr int OT_W = DMC_OT_W, OT_H = DMC_OT_H;
__global__ __launch_bounds__(256) void mv_owner_tile_kernel(const unsigned char* __restrict__ mvs, int stride,
                                                            const int* __restrict__ frame_off, int n_mv, int* __restrict__ owner,
                                                            int H, int W, int tiles_x, int* __restrict__ bad_source) {
    __shared__ int tile[OT_W * OT_H];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y;
    const int x0 = (blockIdx.x % tiles_x) * OT_W, y0 = (blockIdx.x / tiles_x) * OT_H;
    const int i0 = frame_off != nullptr ? __builtin_amdgcn_readfirstlane(frame_off[f]) : 0;
    const int i1 = frame_off != nullptr ? __builtin_amdgcn_readfirstlane(frame_off[f + 1]) : n_mv;
    for (int k = tid; k < OT_W * OT_H; k += 256) tile[k] = -1;
    __syncthreads();
    // the scan is vectorised: a wave tests 64 vectors at a time, one per lane (a scalar walk spent ~200 clocks per vector on
    // its dependent record load: 1.25 ms per batch), and rasterises the few that hit the tile one after the other
    int nbad = 0;
    for (int base = i0 + wave * 64; base < i1; base += 256) {
        const int i = base + lane;
        const bool valid = i < i1;
        MvRec m = load_mv(mvs, stride, valid ? i : i0);
        nbad += valid && m.source != -1;                                                    // the reference asserts, :86
        const int hw = m.w / 2, hh = m.h / 2;
        const bool hit = valid && !(m.dx - m.sx == 0 && m.dy - m.sy == 0) &&                 // :88
                         m.dx + hw > x0 && m.dx - hw < x0 + OT_W && m.dy + hh > y0 && m.dy - hh < y0 + OT_H;
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int vi = base + src;
            const int vdx = __builtin_amdgcn_readlane(m.dx, src), vdy = __builtin_amdgcn_readlane(m.dy, src);
            const int vsx = __builtin_amdgcn_readlane(m.sx, src), vsy = __builtin_amdgcn_readlane(m.sy, src);
            const int vhw = __builtin_amdgcn_readlane(hw, src), vhh = __builtin_amdgcn_readlane(hh, src);
            // the part of the block inside this tile (scalar bounds), walked 16 columns x 4 rows per step: no division per pixel
            const int ox_lo = max(-vhw, x0 - vdx), ox_hi = min(vhw, x0 + OT_W - vdx);
            const int oy_lo = max(-vhh, y0 - vdy), oy_hi = min(vhh, y0 + OT_H - vdy);
            for (int oy0 = oy_lo; oy0 < oy_hi; oy0 += 4)
                for (int ox0 = ox_lo; ox0 < ox_hi; ox0 += 16) {
                    const int ox = ox0 + (lane & 15), oy = oy0 + (lane >> 4);
                    const int pdx = vdx + ox, pdy = vdy + oy, psx = vsx + ox, psy = vsy + oy;
                    if (ox < ox_hi && oy < oy_hi &&
                        pdy >= 0 && pdy < H && pdx >= 0 && pdx < W && psy >= 0 && psy < H && psx >= 0 && psx < W)   // :100-103
                        atomicMax(&tile[(pdy - y0) * OT_W + (pdx - x0)], vi);
                }
        }
    }
    for (int o = 32; o > 0; o >>= 1) nbad += __shfl_xor(nbad, o, 64);
    if (nbad != 0 && bad_source != nullptr && blockIdx.x == 0 && lane == 0) atomicAdd(bad_source, nbad);
    __syncthreads();
    // What goes out is not the owner's index but what every consumer wants of it: its displacement (src - dst), two int16 in one
    // word (both positions lie inside the frame, :100-103, so it fits), DISP_NONE where no vector landed.  The walk through a
    // chain then makes ONE dependent load per frame instead of two (owner, then its record): MV + residual 0.367 -> 0.334 ms per batch.
    int* __restrict__ plane = owner + (size_t)f * H * W;
    for (int k = tid; k < OT_W * OT_H; k += 256) {
        const int px = x0 + k % OT_W, py = y0 + k / OT_W;
        if (px < W && py < H) {
            const int o = tile[k];
            int d = DISP_NONE;
            if (o >= 0) {
                const MvRec m = load_mv(mvs, stride, o);
                d = pack_disp(m.sx - m.dx, m.sy - m.dy);
            }
            plane[py * W + px] = d;
        }
    }
}

// non-accumulating branch, :111-113: covered pixels get (dst - src) of their owner, the others keep what mv_out holds
__global__ __launch_bounds__(256) void mv_rasterise_kernel(const unsigned char* __restrict__ mvs, int stride,
                                                           const int* __restrict__ owner, int* __restrict__ mv_out, int H, int W) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int d = owner[p];
    if (d == DISP_NONE) return;
    reinterpret_cast<int2*>(mv_out)[p] = make_int2(-disp_x(d), -disp_y(d));
}

// accumulating branch, :105-110, one frame: accu_new[dst] = accu_old[src] for covered pixels, accu_old[dst] for the
// rest (the reference gets that from :125-127's copy).  Both in the reference's [x][y][2] layout.
__global__ __launch_bounds__(256) void mv_accumulate_kernel(const unsigned char* __restrict__ mvs, int stride,
                                                            const int* __restrict__ owner, const int* __restrict__ accu_old,
                                                            int* __restrict__ accu_new, int H, int W) {
    // threads run along y (the contiguous index of the accumulator); the owner read is the strided one
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * W) return;
    const int x = q / H, y = q % H;
    const int d = owner[y * W + x];
    int sx = x, sy = y;
    if (d != DISP_NONE) {
        sx = x + disp_x(d);
        sy = y + disp_y(d);
    }
    reinterpret_cast<int2*>(accu_new)[q] = reinterpret_cast<const int2*>(accu_old)[sx * H + sy];
}

__global__ __launch_bounds__(256) void accu_init_kernel(int* __restrict__ accu, int H, int W) {   // :311-318
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * W) return;
    reinterpret_cast<int2*>(accu)[q] = make_int2(q / H, q % H);
}

__global__ __launch_bounds__(256) void mv_from_accu_kernel(const int* __restrict__ accu, int* __restrict__ mv_out, int H, int W) {
    const int p = blockIdx.x * 256 + threadIdx.x;   // :130-139
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    const int2 a = reinterpret_cast<const int2*>(accu)[x * H + y];
    reinterpret_cast<int2*>(mv_out)[p] = make_int2(x - a.x, y - a.y);
}

// :141-175.  src from the accumulator (accumulate) or x - mv (not); the reference reads bgr[0] at the source position
// and bgr[1] at the pixel itself.
__global__ __launch_bounds__(256) void residual_kernel(const unsigned char* __restrict__ ref, const unsigned char* __restrict__ cur,
                                                       const int* __restrict__ accu, const int* __restrict__ mv,
                                                       int* __restrict__ res, int H, int W) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p % W;
    int sx, sy;
    if (accu != nullptr) {
        const int2 a = reinterpret_cast<const int2*>(accu)[x * H + y];
        sx = a.x;
        sy = a.y;
    } else {
        const int2 v = reinterpret_cast<const int2*>(mv)[p];
        sx = x - v.x;
        sy = y - v.y;
    }
    const int ls = (sy * W + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) res[p * 3 + c] = (int)cur[p * 3 + c] - (int)ref[ls + c];
}

// Whole batch: one thread per (chain, pixel) walks back through the chain's owner planes.  chain_off [n_chains + 1]
// indexes frames; emit [n_chains] (nullable = all 1): 0 leaves the chain's outputs untouched (the reference's
// `cur_pos > 0` / `if (sd)` gates, :128 and :363).
// TRACE_PPT pixels per thread, their walks interleaved (a walk is a chain of dependent gathers, one per frame).  Measured on the
// 120-chain batch: 1 pixel 0.278 ms, 2 pixels 0.274-0.284, 4 pixels 0.305 -- the kernel is not short of gathers in flight; 1 it is.
constexpr int TRACE_PPT = 1;
__global__ __launch_bounds__(256) void gop_trace_kernel(const unsigned char* __restrict__ mvs, int stride, const int* __restrict__ chain_off,
                                                        const int* __restrict__ emit, const int* __restrict__ owner,
                                                        const unsigned char* __restrict__ ref, const unsigned char* __restrict__ cur,
                                                        int* __restrict__ accu_out, int* __restrict__ mv_out, int* __restrict__ res_out,
                                                        int H, int W) {
    const int c = blockIdx.y;
    if (emit != nullptr && emit[c] == 0) return;
    const int npx = H * W;
    int p[TRACE_PPT], x[TRACE_PPT], y[TRACE_PPT];
#pragma unroll
    for (int j = 0; j < TRACE_PPT; ++j) {
        p[j] = (blockIdx.x * TRACE_PPT + j) * 256 + threadIdx.x;
        const int q = p[j] < npx ? p[j] : npx - 1;         // (beyond the frame: the last pixel's walk again, nothing stored)
        x[j] = q % W;
        y[j] = q / W;
    }
    const int f0 = chain_off[c], f1 = chain_off[c + 1];
    const size_t hw = (size_t)H * W;
    for (int f = f1 - 1; f >= f0; --f) {
        const int* __restrict__ plane = owner + f * hw;
        int d[TRACE_PPT];
#pragma unroll
        for (int j = 0; j < TRACE_PPT; ++j) d[j] = plane[y[j] * W + x[j]];
#pragma unroll
        for (int j = 0; j < TRACE_PPT; ++j)
            if (d[j] != DISP_NONE) {
                x[j] += disp_x(d[j]);
                y[j] += disp_y(d[j]);
            }
    }
#pragma unroll
    for (int j = 0; j < TRACE_PPT; ++j) {
        if (p[j] >= npx) continue;
        const int x0 = p[j] % W, y0 = p[j] / W;
        if (accu_out != nullptr) reinterpret_cast<int2*>(accu_out + (size_t)c * hw * 2)[x0 * H + y0] = make_int2(x[j], y[j]);
        if (mv_out != nullptr) reinterpret_cast<int2*>(mv_out + (size_t)c * hw * 2)[p[j]] = make_int2(x0 - x[j], y0 - y[j]);
        if (res_out != nullptr) {
            const unsigned char* __restrict__ r = ref + (size_t)c * hw * 3 + ((size_t)y[j] * W + x[j]) * 3;
            const unsigned char* __restrict__ k = cur + (size_t)c * hw * 3 + (size_t)p[j] * 3;
            int* __restrict__ out = res_out + (size_t)c * hw * 3 + (size_t)p[j] * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out[ch] = (int)k[ch] - (int)r[ch];
        }
    }
}

int check_dims(const char* who, int H, int W, int stride) {
    if (H <= 0 || W <= 0 || (long)H * W > (1l << 28)) return fail(DMC_E_INVALID, "%s: bad frame size %d x %d", who, H, W);
    if (stride < 16 || (stride & 7)) return fail(DMC_E_INVALID, "%s: mv_stride %d is not a sizeof(AVMotionVector) (24 or 40: a multiple of 8, >= 16)", who, stride);
    return DMC_OK;
}

int owner_pass(const char* who, const void* mvs, int stride, const int* frame_off, int n_frames, int n_mv, int* owner, int H, int W,
               int* bad_source, hipStream_t s) {
    const int tiles_x = (W + OT_W - 1) / OT_W, tiles_y = (H + OT_H - 1) / OT_H;
    if (n_frames > 65535) return fail(DMC_E_INVALID, "%s: at most 65535 frames per call", who);
    mv_owner_tile_kernel<<<dim3(tiles_x * tiles_y, n_frames), 256, 0, s>>>(static_cast<const unsigned char*>(mvs), stride, frame_off, n_mv,
                                                                            owner, H, W, tiles_x, bad_source);
    return check_launch(who);
}

}  // namespace

extern "C" {

size_t dmc_mv_owner_bytes(int n_frames, int H, int W) {
    if (n_frames <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)n_frames * H * W * sizeof(int);
}

int dmc_mv_accu_init(int32_t* accu, int H, int W, dmc_stream_t stream) {
    if (accu == nullptr) return fail(DMC_E_INVALID, "dmc_mv_accu_init: null pointer");
    if (int rc = check_dims("dmc_mv_accu_init", H, W, 24)) return rc;
    accu_init_kernel<<<(H * W + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(accu, H, W);
    return check_launch("dmc_mv_accu_init");
}

int dmc_mv_rasterise(const void* mvs, int mv_stride, int n_mv, int32_t* owner_ws, int32_t* mv_out, int32_t* bad_source, int H,
                     int W, dmc_stream_t stream) {
    if (owner_ws == nullptr || mv_out == nullptr || (n_mv > 0 && mvs == nullptr) || n_mv < 0)
        return fail(DMC_E_INVALID, "dmc_mv_rasterise: null pointer or negative count");
    if (int rc = check_dims("dmc_mv_rasterise", H, W, mv_stride)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_mv == 0) return DMC_OK;
    if (int rc = owner_pass("dmc_mv_rasterise", mvs, mv_stride, nullptr, 1, n_mv, owner_ws, H, W, bad_source, s)) return rc;
    mv_rasterise_kernel<<<(H * W + 255) / 256, 256, 0, s>>>(static_cast<const unsigned char*>(mvs), mv_stride, owner_ws, mv_out, H, W);
    return check_launch("dmc_mv_rasterise");
}

int dmc_mv_accumulate(const void* mvs, int mv_stride, int n_mv, int32_t* owner_ws, const int32_t* accu_old, int32_t* accu_new,
                      int32_t* bad_source, int H, int W, dmc_stream_t stream) {
    if (owner_ws == nullptr || accu_old == nullptr || accu_new == nullptr || accu_old == accu_new || (n_mv > 0 && mvs == nullptr) ||
        n_mv < 0)
        return fail(DMC_E_INVALID, "dmc_mv_accumulate: null pointer, negative count, or accu_old == accu_new");
    if (int rc = check_dims("dmc_mv_accumulate", H, W, mv_stride)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = owner_pass("dmc_mv_accumulate", mvs, mv_stride, nullptr, 1, n_mv, owner_ws, H, W, bad_source, s)) return rc;
    mv_accumulate_kernel<<<(H * W + 255) / 256, 256, 0, s>>>(static_cast<const unsigned char*>(mvs), mv_stride, owner_ws, accu_old,
                                                             accu_new, H, W);
    return check_launch("dmc_mv_accumulate");
}

int dmc_mv_from_accu(const int32_t* accu, int32_t* mv_out, int H, int W, dmc_stream_t stream) {
    if (accu == nullptr || mv_out == nullptr) return fail(DMC_E_INVALID, "dmc_mv_from_accu: null pointer");
    if (int rc = check_dims("dmc_mv_from_accu", H, W, 24)) return rc;
    mv_from_accu_kernel<<<(H * W + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(accu, mv_out, H, W);
    return check_launch("dmc_mv_from_accu");
}

int dmc_residual(const uint8_t* bgr_ref, const uint8_t* bgr_cur, const int32_t* accu, const int32_t* mv, int32_t* res, int H, int W,
                 dmc_stream_t stream) {
    if (bgr_ref == nullptr || bgr_cur == nullptr || res == nullptr || ((accu == nullptr) == (mv == nullptr)))
        return fail(DMC_E_INVALID, "dmc_residual: null pointer, or not exactly one of accu / mv given");
    if (int rc = check_dims("dmc_residual", H, W, 24)) return rc;
    residual_kernel<<<(H * W + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(bgr_ref, bgr_cur, accu, mv, res, H, W);
    return check_launch("dmc_residual");
}

int dmc_mv_gop_batch(const void* mvs, int mv_stride, int n_mv, const int32_t* frame_off, int n_frames, const int32_t* chain_off,
                     int n_chains, const int32_t* emit, int32_t* owner_ws, const uint8_t* bgr_ref, const uint8_t* bgr_cur,
                     int32_t* accu_out, int32_t* mv_out, int32_t* res_out, int32_t* bad_source, int H, int W, dmc_stream_t stream) {
    if (n_chains <= 0 || n_frames < 0 || n_mv < 0 || chain_off == nullptr || (n_frames > 0 && (frame_off == nullptr || owner_ws == nullptr)) ||
        (n_mv > 0 && mvs == nullptr))
        return fail(DMC_E_INVALID, "dmc_mv_gop_batch: null pointer or bad count");
    if (res_out != nullptr && (bgr_ref == nullptr || bgr_cur == nullptr))
        return fail(DMC_E_INVALID, "dmc_mv_gop_batch: res_out needs bgr_ref and bgr_cur");
    if (n_chains > 65535) return fail(DMC_E_INVALID, "dmc_mv_gop_batch: at most 65535 chains per call");
    if (int rc = check_dims("dmc_mv_gop_batch", H, W, mv_stride)) return rc;
    if ((size_t)n_frames * H * W > (size_t)1 << 40) return fail(DMC_E_INVALID, "dmc_mv_gop_batch: owner maps too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_frames > 0)
        if (int rc = owner_pass("dmc_mv_gop_batch", mvs, mv_stride, frame_off, n_frames, n_mv, owner_ws, H, W, bad_source, s)) return rc;
    gop_trace_kernel<<<dim3((H * W + 256 * TRACE_PPT - 1) / (256 * TRACE_PPT), n_chains), 256, 0, s>>>(static_cast<const unsigned char*>(mvs), mv_stride, chain_off, emit,
                                                                         owner_ws, bgr_ref, bgr_cur, accu_out, mv_out, res_out, H, W);
    return check_launch("dmc_mv_gop_batch");
}

}  // extern "C"
