// GPU-side input preparation: uint8 HWC frames [flow_x flow_y mv_x mv_y r g b]
// -> the three normalised fp32 NCHW tensors the hot path consumes, with the optional horizontal
// flip (x components of flow and MV change sign around 128) and the 16x16 flow blockification.
//
// Reference behaviour: code/dmcnet/dataset.py:215-263 (channel split, block_reduce mean + repeat,
// /255, (x-0.5)/std) and code/dmcnet/transforms.py:47-58 (GroupRandomHorizontalFlip).  The
// arithmetic order is the reference's (fp32: u/255, -0.5, /std; block means in fp64 then cast),
// so the result is bit-identical to CoviarDataSet.__getitem__'s tensors.
// Pure HBM streaming: 7 B read + 28 B written per pixel; 4x less host->device traffic than
// shipping fp32.
#include "dmc_common.h"

using namespace dmc;

namespace {

struct PrepArgs {
    const unsigned char* frames;   // [N][H][W][7]
    const unsigned char* flip;     // [N] or null
    float* flow;                   // [N][2][H][W]
    float* mv;                     // [N][2][H][W]
    float* res;                    // [N][3][H][W]
    float* block_mean;             // [N][2][bh][bw] (factor > 0)
    int N, H, W, factor, bh, bw;
    float inv_std_mean_unused;
    float std_mean, std_r, std_g, std_b;
};

__device__ __forceinline__ int flip_value(int v, int ch, bool flipped) {
    return (flipped && (ch == 0 || ch == 2)) ? 256 - v : v;
}

// mean of the (flipped) flow over factor x factor blocks; ragged edge blocks are zero-padded,
// i.e. divided by factor^2 as skimage.measure.block_reduce does
__global__ __launch_bounds__(256) void block_mean_kernel(PrepArgs a) {
    const long total = (long)a.N * 2 * a.bh * a.bw;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int bx = (int)(i % a.bw), by = (int)((i / a.bw) % a.bh);
    const int c = (int)((i / ((long)a.bw * a.bh)) % 2), n = (int)(i / ((long)a.bw * a.bh * 2));
    const bool flipped = a.flip && a.flip[n];
    long sum = 0;
    for (int dy = 0; dy < a.factor; ++dy) {
        const int y = by * a.factor + dy;
        if (y >= a.H) break;
        for (int dx = 0; dx < a.factor; ++dx) {
            const int x = bx * a.factor + dx;
            if (x >= a.W) break;
            const int xs = flipped ? a.W - 1 - x : x;
            sum += flip_value(a.frames[(((size_t)n * a.H + y) * a.W + xs) * 7 + c], c, flipped);
        }
    }
    a.block_mean[i] = (float)((double)sum / (double)(a.factor * a.factor));
}

__global__ __launch_bounds__(256) void prepare_kernel(PrepArgs a) {
    const size_t HW = (size_t)a.H * a.W;
    const size_t total = (size_t)a.N * HW;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const int n = (int)(p / HW);
        const int rem = (int)(p - (size_t)n * HW);
        const int y = rem / a.W, x = rem - y * a.W;
        const bool flipped = a.flip && a.flip[n];
        const int xs = flipped ? a.W - 1 - x : x;
        const unsigned char* src = a.frames + (((size_t)n * a.H + y) * a.W + xs) * 7;
        int v[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) v[c] = flip_value(src[c], c, flipped);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float f;
            if (a.factor > 0)
                f = a.block_mean[(((size_t)n * 2 + c) * a.bh + y / a.factor) * a.bw + x / a.factor];
            else
                f = (float)v[c];
            a.flow[((size_t)n * 2 + c) * HW + rem] = (f / 255.0f - 0.5f) / a.std_mean;
            a.mv[((size_t)n * 2 + c) * HW + rem] = ((float)v[2 + c] / 255.0f - 0.5f) / a.std_mean;
        }
        a.res[((size_t)n * 3 + 0) * HW + rem] = ((float)v[4] / 255.0f - 0.5f) / a.std_r;
        a.res[((size_t)n * 3 + 1) * HW + rem] = ((float)v[5] / 255.0f - 0.5f) / a.std_g;
        a.res[((size_t)n * 3 + 2) * HW + rem] = ((float)v[6] / 255.0f - 0.5f) / a.std_b;
    }
}


// ------------------------------------------------------------------------------------------
// Crop + bilinear resize + flip + blockify + normalise in one go (reference:
// GroupMultiScaleCrop / GroupCenterCrop / GroupScale, code/dmcnet/transforms.py:36-46,62-78,117-139,
// then GroupRandomHorizontalFlip :47-58, then dataset.py:215-263).  The host only draws the random
// crop box and the flip bit; the uint8 frames travel as decoded (7 B/px instead of 28).
//
// Resize arithmetic = this package's transforms.resize_bilinear (cv2.INTER_LINEAR's half-pixel-centre
// convention), operation for operation in fp32 with contraction off, so results are bit-identical:
//   pos = (i + 0.5) * float(n_in / n_out) - 0.5 in fp32;  lo = floor(pos);  f = double(pos) - lo
//   (numpy promotes float32 - int64 to float64, and everything downstream with it);  indices clamped;
//   top = a*(1-fx) + b*fx;  bot = c*(1-fx) + d*fx;  v = rint(top*(1-fy) + bot*fy) clipped to [0,255], in fp64.
// A box of the output size is copied (no arithmetic), as resize_bilinear does.  A crop AFTER the resize
// (GroupScale + GroupCenterCrop, the validation pipeline) is a window (cy, cx) of the resized image.
// ------------------------------------------------------------------------------------------
struct CropArgs {
    const unsigned char* frames;   // [N][H0][W0][7]
    const int* boxes;              // [N][8] = y0, x0, h, w, rh, rw, cy, cx (device), or null = whole frame
    const unsigned char* flip;     // [N] or null
    float* flow; float* mv; float* res;
    float* block_mean;             // [N][2][bh][bw] (factor > 0)
    int N, H0, W0, OH, OW, factor, bh, bw;
    float std_mean, std_r, std_g, std_b;
};

// box (y0, x0, h, w) of the frame is resized to rh x rw, of which the window at (cy, cx) of the output
// size is kept (GroupScale + GroupCenterCrop); rh x rw == h x w means "no resampling"
struct Box { int y0, x0, h, w, rh, rw, cy, cx; float sy, sx; };      // sy, sx: the fp32 scales n_in / n_out of tap_of, once per box
__device__ __forceinline__ Box box_of(const CropArgs& a, int n) {
    Box b = {0, 0, a.H0, a.W0, a.OH, a.OW, 0, 0, 1.f, 1.f};
    if (a.boxes) {
        const int* p = a.boxes + (size_t)n * 8;
        b.y0 = p[0]; b.x0 = p[1]; b.h = p[2]; b.w = p[3]; b.rh = p[4]; b.rw = p[5]; b.cy = p[6]; b.cx = p[7];
    }
    if (b.h != b.rh || b.w != b.rw) {
        b.sy = (float)((double)b.h / (double)b.rh);
        b.sx = (float)((double)b.w / (double)b.rw);
    }
    return b;
}

struct Tap { int lo, hi; double f; };
__device__ __forceinline__ Tap tap_scaled(int i, float scale, int n_in);
__device__ __forceinline__ Tap tap_of(int i, int n_out, int n_in) {
    return tap_scaled(i, (float)((double)n_in / (double)n_out), n_in);
}
// tap_of with the scale given (an fp64 division per PIXEL was a tenth of the resize path's instructions)
__device__ __forceinline__ Tap tap_scaled(int i, float scale, int n_in) {
    const float pos = ((float)i + 0.5f) * scale - 0.5f;
    const float fl = floorf(pos);
    const int lo = (int)fl;
    Tap t;
    t.f = (double)pos - (double)lo;
    t.lo = lo < 0 ? 0 : (lo > n_in - 1 ? n_in - 1 : lo);
    t.hi = lo + 1 < 0 ? 0 : (lo + 1 > n_in - 1 ? n_in - 1 : lo + 1);
    return t;
}

// value of channel c at output pixel (y, x) BEFORE the flip's sign change (x already un-mirrored by the caller)
__device__ __forceinline__ int resized_u8(const CropArgs& a, const unsigned char* fr, const Box& b, int y, int x, int c) {
    if (b.h == b.rh && b.w == b.rw)
        return fr[((size_t)(b.y0 + b.cy + y) * a.W0 + (b.x0 + b.cx + x)) * 7 + c];
    const Tap ty = tap_of(y + b.cy, b.rh, b.h), tx = tap_of(x + b.cx, b.rw, b.w);
    const unsigned char* r0 = fr + ((size_t)(b.y0 + ty.lo) * a.W0 + b.x0) * 7 + c;
    const unsigned char* r1 = fr + ((size_t)(b.y0 + ty.hi) * a.W0 + b.x0) * 7 + c;
    const double p00 = (double)r0[(size_t)tx.lo * 7], p01 = (double)r0[(size_t)tx.hi * 7];
    const double p10 = (double)r1[(size_t)tx.lo * 7], p11 = (double)r1[(size_t)tx.hi * 7];
    const double top = p00 * (1.0 - tx.f) + p01 * tx.f;
    const double bot = p10 * (1.0 - tx.f) + p11 * tx.f;
    double v = rint(top * (1.0 - ty.f) + bot * ty.f);
    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    return (int)v;
}

// Values of 4 consecutive output pixels (y, x0 .. x0 + 3), all 7 channels, after crop / resize / flip:
// identity-size boxes read their 28 source bytes as aligned dwords + byte alignment (no byte loads).
__device__ __forceinline__ void quad_values(const CropArgs& a, const unsigned char* fr, const Box& b, int y, int x0,
                                            bool flipped, int (&v)[4][7]) {
    const bool identity = b.h == b.rh && b.w == b.rw;
    if (identity && x0 + 3 < a.OW) {
        // source pixels xs .. xs+3 (ascending in memory): 28 contiguous bytes
        const int xs = flipped ? a.OW - 4 - x0 : x0;
        const size_t off = ((size_t)(b.y0 + b.cy + y) * a.W0 + (b.x0 + b.cx + xs)) * 7;
        const size_t base = (size_t)(fr - a.frames) + off;
        const unsigned* w32 = reinterpret_cast<const unsigned*>(a.frames + (base & ~(size_t)3));
        const unsigned sh = (unsigned)(base & 3);
        unsigned d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = w32[k];              // (the 8th dword is inside the buffer's 16-byte tail pad)
        unsigned char bytes[28];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const unsigned u = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
            bytes[4 * k + 0] = u & 255; bytes[4 * k + 1] = (u >> 8) & 255;
            bytes[4 * k + 2] = (u >> 16) & 255; bytes[4 * k + 3] = u >> 24;
        }
        // (two unrolled copies: with a run-time `flipped` the index js * 7 + c is a dynamic index into a register array, which the
        // compiler lowers to a 28-way select chain per value -- 1,000 of the kernel's 1,500 VALU instructions per quad)
        if (flipped) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 7; ++c) v[j][c] = flip_value(bytes[(3 - j) * 7 + c], c, true);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 7; ++c) v[j][c] = bytes[j * 7 + c];
        }
    } else if (identity) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j < a.OW ? x0 + j : a.OW - 1;
            const int xs = flipped ? a.OW - 1 - x : x;
#pragma unroll
            for (int c = 0; c < 7; ++c) v[j][c] = flip_value(resized_u8(a, fr, b, y, xs, c), c, flipped);
        }
    } else {
        // bilinear resize: the taps are computed once per pixel (not per channel) and the two horizontally
        // adjacent source pixels of a row (14 contiguous bytes) arrive as 5 aligned dwords + byte
        // alignment instead of 14 byte loads -- the byte gathers were the kernel's bottleneck.  The
        // arithmetic is resized_u8's, operation for operation.
        const Tap ty = tap_scaled(y + b.cy, b.sy, b.h);
        const unsigned char* row0 = fr + ((size_t)(b.y0 + ty.lo) * a.W0 + b.x0) * 7;
        const unsigned char* row1 = fr + ((size_t)(b.y0 + ty.hi) * a.W0 + b.x0) * 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j < a.OW ? x0 + j : a.OW - 1;
            const int xs = flipped ? a.OW - 1 - x : x;
            const Tap tx = tap_scaled(xs + b.cx, b.sx, b.w);
            const int hi_off = (tx.hi - tx.lo) * 7;               // 7, or 0 at the clamped right edge
            unsigned char pb[2][16];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const size_t base = (size_t)((r ? row1 : row0) + (size_t)tx.lo * 7 - a.frames);
                const unsigned* w32 = reinterpret_cast<const unsigned*>(a.frames + (base & ~(size_t)3));
                const unsigned sh = (unsigned)(base & 3);
                unsigned d[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) d[k] = w32[k];          // (at most 6 bytes past the pair: inside the tail pad)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned u = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
                    pb[r][4 * k + 0] = u & 255; pb[r][4 * k + 1] = (u >> 8) & 255;
                    pb[r][4 * k + 2] = (u >> 16) & 255; pb[r][4 * k + 3] = u >> 24;
                }
            }
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const double p00 = (double)pb[0][c], p01 = (double)(hi_off ? pb[0][7 + c] : pb[0][c]);
                const double p10 = (double)pb[1][c], p11 = (double)(hi_off ? pb[1][7 + c] : pb[1][c]);
                const double top = p00 * (1.0 - tx.f) + p01 * tx.f;
                const double bot = p10 * (1.0 - tx.f) + p11 * tx.f;
                double r = rint(top * (1.0 - ty.f) + bot * ty.f);
                r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);
                v[j][c] = flip_value((int)r, c, flipped);
            }
        }
    }
}

// normalised outputs of a quad: channels 2..6 (mv, residual) from the table, the two flow channels either from
// the table (factor == 0) or from the given block means
struct Lut { float t[4][257]; };
__device__ __forceinline__ void fill_lut(const CropArgs& a, Lut& lut) {
    // (u / 255 - 0.5) / std has 257 possible results per standard deviation (256 - v for a flipped x-component,
    // code/dmcnet/transforms.py:54-56, reaches 256): one table per std, filled with exactly that expression,
    // replaces two IEEE divisions per value (56 per quad)
    for (int i = threadIdx.x; i < 257; i += blockDim.x) {
        const float u = (float)i;
        lut.t[0][i] = (u / 255.0f - 0.5f) / a.std_mean;
        lut.t[1][i] = (u / 255.0f - 0.5f) / a.std_r;
        lut.t[2][i] = (u / 255.0f - 0.5f) / a.std_g;
        lut.t[3][i] = (u / 255.0f - 0.5f) / a.std_b;
    }
}

__device__ __forceinline__ void store_plane_quad(const CropArgs& a, float* dst, size_t pix, int x0, const float (&o)[4]) {
    if ((a.OW & 3) == 0) {
        *reinterpret_cast<float4*>(dst + pix) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x0 + j < a.OW) dst[pix + j] = o[j];
    }
}

__device__ __forceinline__ float* plane_of(const CropArgs& a, int n, int c, size_t OHW) {
    return c < 2 ? a.flow + ((size_t)n * 2 + c) * OHW
         : c < 4 ? a.mv + ((size_t)n * 2 + (c - 2)) * OHW
                 : a.res + ((size_t)n * 3 + (c - 4)) * OHW;
}

// flow_ds_factor == 0: 4 consecutive output pixels per thread, float4 stores into the seven planes
__global__ __launch_bounds__(256) void prepare_crop_kernel(CropArgs a) {
    __shared__ Lut lut;
    fill_lut(a, lut);
    __syncthreads();
    const int qw = (a.OW + 3) / 4;                                  // 4-pixel groups per row
    const size_t OHW = (size_t)a.OH * a.OW;
    const size_t total = (size_t)a.N * a.OH * qw;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int n = (int)(g / ((size_t)a.OH * qw));
        const int rem = (int)(g - (size_t)n * a.OH * qw);
        const int y = rem / qw, x0 = (rem - y * qw) * 4;
        const bool flipped = a.flip && a.flip[n];
        const Box b = box_of(a, n);
        int v[4][7];
        quad_values(a, a.frames + (size_t)n * a.H0 * a.W0 * 7, b, y, x0, flipped, v);
        const size_t pix = (size_t)y * a.OW + x0;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const int k = c < 4 ? 0 : c - 3;
            const float o[4] = {lut.t[k][v[0][c]], lut.t[k][v[1][c]], lut.t[k][v[2][c]], lut.t[k][v[3][c]]};
            store_plane_quad(a, plane_of(a, n, c, OHW), pix, x0, o);
        }
    }
}

// flow_ds_factor > 0 (the dmcnet recipes' 16): one workgroup per (frame, band of `factor` output rows), ONE pass
// over the source: every quad's mv / residual planes are written at once, its flow values are added to
// the band's per-block integer sums in LDS, and after a barrier the flow planes are written from the block
// means (ragged edge blocks divide by factor^2, as skimage.measure.block_reduce's zero padding does).
// HBM traffic = the algorithmic 7 B per sampled source pixel + 28 B per output pixel; the earlier two-kernel
// form read the flow channels twice and round-tripped the block means.
__global__ __launch_bounds__(256) void prepare_crop_band_kernel(CropArgs a) {
    __shared__ Lut lut;
    extern __shared__ int bsum[];                                    // [bw][2]
    fill_lut(a, lut);
    for (int i = threadIdx.x; i < 2 * a.bw; i += 256) bsum[i] = 0;
    __syncthreads();
    const int n = blockIdx.x / a.bh, by = blockIdx.x - n * a.bh;
    const int qw = (a.OW + 3) / 4;
    const size_t OHW = (size_t)a.OH * a.OW;
    const bool flipped = a.flip && a.flip[n];
    const Box b = box_of(a, n);
    const unsigned char* fr = a.frames + (size_t)n * a.H0 * a.W0 * 7;
    const int y_begin = by * a.factor, rows = (a.OH - y_begin < a.factor) ? a.OH - y_begin : a.factor;
    for (int g = threadIdx.x; g < rows * qw; g += 256) {
        const int y = y_begin + g / qw, x0 = (g % qw) * 4;
        int v[4][7];
        quad_values(a, fr, b, y, x0, flipped, v);
        const size_t pix = (size_t)y * a.OW + x0;
#pragma unroll
        for (int c = 2; c < 7; ++c) {
            const int k = c < 4 ? 0 : c - 3;
            const float o[4] = {lut.t[k][v[0][c]], lut.t[k][v[1][c]], lut.t[k][v[2][c]], lut.t[k][v[3][c]]};
            store_plane_quad(a, plane_of(a, n, c, OHW), pix, x0, o);
        }
        if ((a.factor & 3) == 0 && x0 + 3 < a.OW) {                    // the quad lies in one block column; integer sums: any grouping
            atomicAdd(&bsum[2 * (x0 / a.factor) + 0], v[0][0] + v[1][0] + v[2][0] + v[3][0]);
            atomicAdd(&bsum[2 * (x0 / a.factor) + 1], v[0][1] + v[1][1] + v[2][1] + v[3][1]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < a.OW) {
                    atomicAdd(&bsum[2 * ((x0 + j) / a.factor) + 0], v[j][0]);
                    atomicAdd(&bsum[2 * ((x0 + j) / a.factor) + 1], v[j][1]);
                }
        }
    }
    __syncthreads();
    const double denom = (double)(a.factor * a.factor);
    for (int g = threadIdx.x; g < rows * qw; g += 256) {
        const int y = y_begin + g / qw, x0 = (g % qw) * 4;
        const size_t pix = (size_t)y * a.OW + x0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + j < a.OW ? x0 + j : a.OW - 1;
                const float mean = (float)((double)bsum[2 * (x / a.factor) + c] / denom);
                o[j] = (mean / 255.0f - 0.5f) / a.std_mean;
            }
            store_plane_quad(a, plane_of(a, n, c, OHW), pix, x0, o);
        }
    }
}

}  // namespace

extern "C" {

size_t dmc_prepare_inputs_workspace_bytes(int N, int H, int W, int flow_ds_factor) {
    if (flow_ds_factor <= 0) return 16;
    const size_t bh = (H + flow_ds_factor - 1) / flow_ds_factor, bw = (W + flow_ds_factor - 1) / flow_ds_factor;
    return (size_t)N * 2 * bh * bw * sizeof(float) + 16;
}

int dmc_prepare_inputs(const unsigned char* frames_u8, const unsigned char* flip, float* out_flow,
                       float* out_mv, float* out_res, float* workspace, int N, int H, int W,
                       int flow_ds_factor, const float* std4_host, dmc_stream_t stream) {
    if (!frames_u8 || !out_flow || !out_mv || !out_res || !workspace || !std4_host)
        return fail(DMC_E_INVALID, "dmc_prepare_inputs: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || flow_ds_factor < 0)
        return fail(DMC_E_INVALID, "dmc_prepare_inputs: bad shape");
    hipStream_t s = (hipStream_t)stream;
    PrepArgs a;
    a.frames = frames_u8; a.flip = flip; a.flow = out_flow; a.mv = out_mv; a.res = out_res;
    a.block_mean = workspace; a.N = N; a.H = H; a.W = W; a.factor = flow_ds_factor;
    a.bh = flow_ds_factor ? (H + flow_ds_factor - 1) / flow_ds_factor : 0;
    a.bw = flow_ds_factor ? (W + flow_ds_factor - 1) / flow_ds_factor : 0;
    a.inv_std_mean_unused = 0.f;
    a.std_mean = std4_host[0]; a.std_r = std4_host[1]; a.std_g = std4_host[2]; a.std_b = std4_host[3];
    int rc;
    if (flow_ds_factor > 0) {
        const long total = (long)N * 2 * a.bh * a.bw;
        block_mean_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(a);
        if ((rc = check_launch("prepare_block_mean"))) return rc;
    }
    size_t want = ((size_t)N * H * W + 255) / 256;
    prepare_kernel<<<(int)(want > 8192 ? 8192 : want), 256, 0, s>>>(a);
    return check_launch("prepare_inputs");
}

size_t dmc_prepare_crop_workspace_bytes(int N, int OH, int OW, int flow_ds_factor) {
    return dmc_prepare_inputs_workspace_bytes(N, OH, OW, flow_ds_factor);
}

int dmc_prepare_inputs_crop(const unsigned char* frames_u8, const int* boxes, const unsigned char* flip,
                            float* out_flow, float* out_mv, float* out_res, float* workspace, int N,
                            int H0, int W0, int OH, int OW, int flow_ds_factor, const float* std4_host,
                            dmc_stream_t stream) {
    if (!frames_u8 || !out_flow || !out_mv || !out_res || !workspace || !std4_host)
        return fail(DMC_E_INVALID, "dmc_prepare_inputs_crop: null pointer");
    if (N <= 0 || H0 <= 0 || W0 <= 0 || OH <= 0 || OW <= 0 || flow_ds_factor < 0)
        return fail(DMC_E_INVALID, "dmc_prepare_inputs_crop: bad shape");
    if (!boxes && (H0 != OH || W0 != OW))
        return fail(DMC_E_INVALID, "dmc_prepare_inputs_crop: without boxes the frames must measure OH x OW");
    hipStream_t s = (hipStream_t)stream;
    CropArgs a;
    a.frames = frames_u8; a.boxes = boxes; a.flip = flip; a.flow = out_flow; a.mv = out_mv; a.res = out_res;
    a.block_mean = workspace; a.N = N; a.H0 = H0; a.W0 = W0; a.OH = OH; a.OW = OW; a.factor = flow_ds_factor;
    a.bh = flow_ds_factor ? (OH + flow_ds_factor - 1) / flow_ds_factor : 0;
    a.bw = flow_ds_factor ? (OW + flow_ds_factor - 1) / flow_ds_factor : 0;
    a.std_mean = std4_host[0]; a.std_r = std4_host[1]; a.std_g = std4_host[2]; a.std_b = std4_host[3];
    if (flow_ds_factor > 0) {
        prepare_crop_band_kernel<<<N * a.bh, 256, (size_t)2 * a.bw * sizeof(int), s>>>(a);
        return check_launch("prepare_inputs_crop_band");
    }
    const size_t want = ((size_t)N * OH * ((OW + 3) / 4) + 255) / 256;
    prepare_crop_kernel<<<(int)(want > 16384 ? 16384 : want), 256, 0, s>>>(a);
    return check_launch("prepare_inputs_crop");
}

}  // extern "C"
