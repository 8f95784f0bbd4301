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

}  // extern "C"
