/*
 * dmcnet_hip.h -- C ABI of libdmcnet_hip.so: the MI355X (gfx950) kernels behind the DMC-Net
 * training hot path.
 *
 * The reference (facebookresearch/dmc-net) has NO FFI seam on this path: its hot path is stock
 * torch.nn modules driven from Python.  Each entry point below therefore
 * names the reference Python it replaces (paths relative to the reference root); the binding a
 * maintainer adds on the reference side is the ctypes stub in INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes, no torch types; every pointer is a DEVICE pointer unless it
 *     says "host";
 *   - the caller owns every buffer, workspaces included; nothing here allocates, frees or
 *     synchronises; every launch is asynchronous on `stream` (a hipStream_t; NULL = the
 *     null stream);
 *   - tensors are contiguous NCHW fp32; weights are in PyTorch's native [Cout][Cin][3][3]
 *     layout with the reference's prepend-concat input-channel order
 *     (code/dmcnet/model.py:187-194: layer k sees [y_{k-1}, ..., y_0, mv(2), residual(3)]);
 *   - return value: DMC_OK (0) or a negative DMC_E_* code; dmc_last_error() gives the text
 *     (thread-local);
 *   - the ONLY process-wide state is the table of kernel-selection options behind dmc_set_option() (A/B
 *     measurement knobs, relaxed atomics, defaults = fastest path; an operator reads them at each of its
 *     entry points, so do not change them while another thread is between two calls of one operator); the
 *     library never reads the environment; otherwise safe from several host threads on distinct streams.
 */
#ifndef DMCNET_HIP_H
#define DMCNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMC_OK 0
#define DMC_E_INVALID (-1) /* bad argument (null pointer, non-positive size, ...) */
#define DMC_E_LAUNCH (-2)  /* hipLaunch / hipGetLastError failure                */

typedef void* dmc_stream_t; /* hipStream_t */

/* Library version: major*10000 + minor*100 + patch. */
int dmc_version(void);
/* Text of the last error on this host thread ("" if none). */
const char* dmc_last_error(void);
/* Kernel-selection options for A/B measurements (tools/, bench.py); every default is the fastest measured path.
 *   "gen_fused"       1 (default): the generator forward as ONE launch (gen_fused.hip: strips of <= 118 columns walked row by row,
 *                     features line-buffered in LDS, the six layers pipelined across the waves of a workgroup; any H, W <= 224);
 *                     0: the layer-by-layer kernels, which also serve wider images.  [2 / 3 = also the one-launch data gradient,
 *                     gen_fused_bwd.hip: measured slower, -DDMC_MEASURE build only]
 *   "gen_layer_path"  1 (default) / 0: matrix-core / VALU layer kernels of the layer-by-layer forward and the data gradient.
 *                     [3, 4, 5: tile variants that lost their measurement, -DDMC_MEASURE build only]
 *   "gen_gather"      0: push form for the Cout-8 layers.        "gen_fuse45"  0: layers 4 and 5 as two launches.
 *   "gen_wgrad_path"  5 (default): the row-sliding weight gradient of gen_wgrad.hip (W % 4 == 0, else path 4's rules); 4: the bf16x3
 *                     tile kernel.  [0 .. 3: predecessors, -DDMC_MEASURE build only]
 *   "gen_fuse_fwd" / "gen_fuse_bwd"  0: the layer-by-layer forward / data-gradient launches instead of the fused groups.
 *   "gen_x3"          bit K: hidden layer K (0 .. 2) of the layer-by-layer forward in bf16x3 arithmetic (gen_x3.hip); default 2.
 *   "gen_wino"        bit K: hidden layer K (0 .. 3) of the layer-by-layer forward, bit 8 + K: data-gradient group K (0 .. 4), on the
 *                     Winograd F(2x2, 3x3) ring kernel (fp32, results within rounding of the direct kernels'; W % 4 == 0,
 *                     64 <= W <= 224); default 768 = gradient groups 0 and 1.
 *   "conv_path" / "conv_arith" / "conv_cfg"  classifier / discriminator / I3D convolutions: second-generation kernels (1), bf16x3
 *                     arithmetic (1) or fp32 MFMA (0), and a forced tile configuration (0 = automatic; 1 .. 5 tiles of the
 *                     tap-stepping 3-D kernel, 6 = never the patch-resident 3x3x3 kernel, 7 / 8 = its 128- / 64-position tiles,
 *                     9 = the scan form of the 3-D max pool, 11 / 12 = the first forms of the I3D stem's weight / data gradient (LDS scatter; one or four rows per wave),
 *                     101 .. 305 = 2-D tile choices named in the kernels).
 *   "conv3d_wgrad"    2 (default): row-ring weight gradient for 3x3x3 and 1x1x1; 1: 3x3x3 only; 0: tap-stepping kernels.
 *   "grid_reserve_cus" 0 (default) .. 128: CUs every PERSISTENT grid (gen_fused, the generator's ring / gather / Winograd kernels,
 *                     gen_wgrad_rs, conv3d_p3) leaves idle -- room for RCCL's channel kernels while gradients are exchanged
 *                     during the backward pass; buffer-size queries do not depend on it; no result changes.
 *   "gen_ablate" / "conv_ablate" (parts of a kernel switched off, results wrong), "gen_stagger": MEASUREMENT BUILD ONLY
 *                     (-DDMC_MEASURE, dmc-net_amd/build.py --measure -> libdmcnet_hip_measure.so): the product library refuses
 *                     them, they read 0 and the ablated paths are not compiled into its kernels.
 * The product library also refuses the option VALUES in [brackets] above (kernel variants that lost their A/B measurement are
 * compiled into the measurement build only).  dmc_get_option("measure_build") reads 1 in that build, 0 in the product library.
 * dmc_set_option returns DMC_E_INVALID for an unknown name or a refused value; dmc_get_option returns -1 for an unknown name. */
int dmc_set_option(const char* name, int value);
int dmc_get_option(const char* name);
/* Launches an empty kernel named dmc_profile_mark_kernel on `stream`: a marker that brackets
 * a region of interest in a rocprofv3 kernel trace (no reference counterpart; tooling only). */
int dmc_profile_mark(dmc_stream_t stream);

/* ---- EstimatorDenseNetTiny (the DMC generator every shipped recipe uses) ----------------
 * Replaces: EstimatorDenseNetTiny.forward            code/dmcnet/model.py:187-194
 *           + torch.cat((mv, residual), 1)            code/dmcnet/model.py:341
 *           + torch.add(gen_flow, input_mv)           code/dmcnet/model.py:345-346
 * Layer widths 5->8, 13->8, 21->6, 27->4, 31->2, 33->2 (last one without LeakyReLU(0.1)).
 *
 * w[6], b[6] (host arrays of device pointers): conv_0..conv_4 `.0.weight/.0.bias`, then
 * predict_flow.weight/.bias.
 */

/* Bytes of `workspace` the generator entry points need (repacked weights, zero words, bf16x3 weight fragments). */
size_t dmc_gen_tiny_workspace_bytes(void);
/* Bytes of the saved-activation buffer for N frames of H x W (28 channels fp32). */
size_t dmc_gen_tiny_saved_bytes(int N, int H, int W);

/*
 * Forward.  mv [N,2,H,W], res [N,3,H,W] -> out [N,2,H,W].
 * saved: buffer of dmc_gen_tiny_saved_bytes(); it receives y0..y4 (the post-LeakyReLU features, physical channel order) for
 * the backward pass.  NULL = inference (nothing is kept: the forward then writes 8 instead of 120 bytes per pixel): accepted
 * when the one-launch forward serves the shape (option "gen_fused" bit 0, W <= 224), DMC_E_INVALID otherwise -- the
 * layer-by-layer kernels pass the features from launch to launch through this buffer.
 * workspace: dmc_gen_tiny_workspace_bytes() (repacked weights + 256 zero words + the bf16x3 weight fragments of gen_x3.hip).
 * add_mv_delta != 0 adds input_mv to the result (gen_flow_or_delta == 1).
 * Any H, W >= 1; W % 4 == 0 takes the vectorised path.
 */
int dmc_gen_tiny_fwd(const float* mv, const float* res, const float* const* w,
                     const float* const* b, float* out, float* saved, float* workspace, int N,
                     int H, int W, int add_mv_delta, dmc_stream_t stream);

/*
 * Forward + flow-reconstruction loss in one go: as dmc_gen_tiny_fwd, and additionally
 * *loss_out = mean((out - flow)^2) = nn.MSELoss()(gen_flow, input_flow), code/dmcnet/train.py:167,245,
 * reduced in the epilogue of the kernel that produces `out` (the separate loss kernel would read `out`
 * back from HBM).  flow [N,2,H,W]; loss_out: one float on the device; mse_partials: workspace of
 * dmc_gen_tiny_mse_partials_bytes().  Shapes that do not take the fused kernel (W % 4 != 0, W > 224)
 * fall back to dmc_flow_mse_fwd internally.  The gradient is dmc_flow_mse_bwd's, unchanged.
 */
size_t dmc_gen_tiny_mse_partials_bytes(void);
int dmc_gen_tiny_fwd_mse(const float* mv, const float* res, const float* const* w, const float* const* b,
                         const float* flow, float* out, float* saved, float* workspace, float* loss_out,
                         void* mse_partials, int N, int H, int W, int add_mv_delta, dmc_stream_t stream);

/* Bytes of the gradient-feature buffer (`gbuf`) and of the per-workgroup weight-gradient
 * partials (`partials`) that the backward needs. */
size_t dmc_gen_tiny_gbuf_bytes(int N, int H, int W);
size_t dmc_gen_tiny_partials_bytes(int N, int H, int W);

/*
 * Backward (autograd of the stack above; replaces loss.backward() through
 * code/dmcnet/model.py:187-194).  grad_out [N,2,H,W] is dL/d(out).
 * Writes dw[6], db[6] (host arrays of device pointers, same shapes as w, b; overwritten, not
 * accumulated).  Gradients w.r.t. mv/res through the convolutions are not produced (the
 * reference never asks for them); in delta mode dL/d(mv) through the skip is grad_out itself.
 */
int dmc_gen_tiny_bwd(const float* mv, const float* res, const float* const* w,
                     const float* saved, const float* grad_out, float* const* dw,
                     float* const* db, float* gbuf, float* partials, float* workspace, int N,
                     int H, int W, dmc_stream_t stream);

/* ---- flow reconstruction loss --------------------------------------------------------------
 * Replaces: nn.MSELoss()(gen_flow, input_flow)        code/dmcnet/train.py:167,245
 * (mean over all `numel` elements).  partials: workspace of dmc_flow_mse_partials_bytes().
 * loss_out: one float on the device.
 */
size_t dmc_flow_mse_partials_bytes(void);
int dmc_flow_mse_fwd(const float* gen_flow, const float* flow, float* loss_out, float* partials,
                     size_t numel, dmc_stream_t stream);
/* grad_gen = 2 (gen - flow) / numel * (*grad_loss) ; grad_loss is a device scalar. */
int dmc_flow_mse_bwd(const float* gen_flow, const float* flow, const float* grad_loss,
                     float* grad_gen, size_t numel, dmc_stream_t stream);

/* ---- TSN segment consensus + cross entropy ---------------------------------------------------
 * Replaces: output.view(-1, S, C).mean(dim=1); CrossEntropyLoss()(output, target)
 *           code/dmcnet/train.py:239-241   (also the adversarial CE with S = 1,
 *           code/dmcnet_GAN/train.py:274,346)
 * logits [B*S, C] fp32, target [B] int64 ->
 *   consensus [B, C] (the averaged logits, what accuracy() consumes),
 *   loss_out  one float: mean over B of -log softmax(consensus)[target],
 *   grad_logits [B*S, C] = (softmax - onehot) / (B*S), i.e. dloss/dlogits for upstream grad 1
 *   (may be NULL to skip).
 * A label outside [0, C) is never used as an index: the loss and that clip's gradient rows become
 * NaN (torch's CrossEntropyLoss asserts on the device instead).
 */
int dmc_consensus_ce_fwd_bwd(const float* logits, const int64_t* target, float* consensus,
                             float* loss_out, float* grad_logits, int B, int S, int C,
                             dmc_stream_t stream);

/* ---- discriminator block tail ----------------------------------------------------------------
 * Replaces the three modules that follow the Conv2d of discriminator_block /
 * discriminator_block2:  LeakyReLU(0.2) -> Dropout2d(0.25) -> BatchNorm2d(C, eps=0.8)
 *           code/dmcnet_GAN/model.py:254-279
 * x [N,C,H,W] is the conv output.  keep [N,C] holds the Dropout2d keep-mask ALREADY divided by
 * (1-p) (0 or 1/0.75) -- the caller draws it, so CPU and GPU runs can share one.
 * Training-mode BatchNorm: batch statistics (biased variance for normalisation), running stats
 * updated with momentum (unbiased variance), as torch.nn.BatchNorm2d does.
 * use_bn == 0 (first block): y = keep * lrelu(x) only; gamma..running_var ignored.
 * stats: workspace of dmc_disc_tail_stats_bytes(C) bytes; its first 2*C floats receive
 * (mean, invstd) for the backward pass, the rest is reduction scratch.
 */
size_t dmc_disc_tail_stats_bytes(int C);
int dmc_disc_tail_fwd(const float* x, const float* keep, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float* y, float* stats, int N,
                      int C, int H, int W, int use_bn, int training, float eps, float momentum,
                      dmc_stream_t stream);
/* Backward of the above in training mode: given dy -> dx, dgamma, dbeta (overwritten).
 * `stats` is the buffer the forward filled (its scratch part is reused). */
int dmc_disc_tail_bwd(const float* x, const float* keep, const float* gamma, float* stats,
                      const float* dy, float* dx, float* dgamma, float* dbeta, int N, int C, int H,
                      int W, int use_bn, dmc_stream_t stream);

/* ---- classifier BatchNorm [+ residual] [+ ReLU], NHWC -------------------------------------------
 * Replaces, inside the torchvision ResNet the reference builds at code/dmcnet/model.py:305 and
 * runs at :352 (GAN: code/dmcnet_GAN/model.py:560): nn.BatchNorm2d followed by the residual
 * addition and the in-place nn.ReLU of the stem and of every BasicBlock / Bottleneck:
 *     y = act( BN(x) [+ residual] ),  act = ReLU if relu != 0 else identity.
 * x, residual, y, dy, dx, dresidual: [M][C] row-major, M = N*H*W (the memory of a channels_last
 * tensor).  Training mode uses batch statistics and updates running_mean / running_var exactly
 * as nn.BatchNorm2d does.  dmc_bn_act_supported() tells whether (M, C) is handled
 * (C % 4 == 0, C/4 <= 256, 256 % (C/4) == 0); callers use the stock op otherwise.
 * stats: dmc_bn_act_stats_bytes(C) bytes (2*C floats): the forward leaves (mean, invstd) there for
 * the backward -- the only thing a caller keeps between the two.  scratch: dmc_bn_act_scratch_bytes(C)
 * bytes of reduction workspace, transient (dead when the call's kernels have run; NULL allowed for
 * an eval-mode forward, which reduces nothing).  relu_mask (nullable, M*C/4 bytes): the forward stores the four ReLU
 * sign bits of every float4; a backward given the same buffer reads them instead of re-reading the
 * residual (`residual` may then be NULL).  Without it the backward recomputes the signs from x
 * (and residual).
 */
int dmc_bn_act_supported(int M, int C);
size_t dmc_bn_act_stats_bytes(int C);
size_t dmc_bn_act_scratch_bytes(int C);
int dmc_bn_act_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float* y, float* stats, void* scratch,
                   unsigned char* relu_mask, int M, int C, int relu, int training, float eps,
                   float momentum, dmc_stream_t stream);
int dmc_bn_act_bwd(const float* x, const float* residual, const float* gamma, const float* beta,
                   const float* stats, void* scratch, const float* dy, float* dx, float* dresidual,
                   float* dgamma, float* dbeta, const unsigned char* relu_mask, int M, int C, int relu,
                   dmc_stream_t stream);

/* Pieces of the same kernels for a producer that has already reduced the statistics (the
 * discriminator blocks, whose convolution epilogue does it): the apply pass alone; the backward
 * of  y = BN(z), z = keep[n][c] * LeakyReLU_slope(pre)  (code/dmcnet_GAN/model.py:254-279: Conv ->
 * LeakyReLU(0.2) -> Dropout2d -> BatchNorm2d) down to d(pre) = the gradient of the convolution's
 * output, gamma == NULL meaning "no BatchNorm" (first block); and a per-channel row sum (the
 * convolution's bias gradient).  scratch: dmc_bn_act_scratch_bytes(C).
 */
int dmc_bn_apply_nhwc(const float* x, const float* gamma, const float* beta, const float* stats, float* y,
                      int M, int C, dmc_stream_t stream);
/* the same apply pass with the residual add / ReLU / ReLU sign mask of dmc_bn_act_fwd: the second half of
 * torchvision's `relu(bn(conv(x)) [+ identity])` when the convolution's epilogue reduced the statistics */
int dmc_bn_apply_act_nhwc(const float* x, const float* residual, const float* gamma, const float* beta,
                          const float* stats, float* y, unsigned char* relu_mask, int M, int C, int relu,
                          dmc_stream_t stream);
int dmc_bn_bwd_act_nhwc(const float* z, const float* gamma, const float* beta, const float* stats, void* scratch,
                        const float* dy, float* dpre, float* dgamma, float* dbeta, const float* keep, int hw,
                        float slope, int M, int C, dmc_stream_t stream);
int dmc_channel_sum_nhwc(const float* g, void* scratch, float* out, int M, int C, dmc_stream_t stream);

/* ---- GPU-side input preparation -----------------------------------------------------------------
 * Replaces the tensor side of CoviarDataSet.__getitem__, code/dmcnet/dataset.py:215-263 (channel
 * split, flow block_reduce(mean)+repeat when flow_ds_factor != 0, /255, (x-0.5)/std) and the
 * horizontal flip of code/dmcnet/transforms.py:47-58, bit-identically.
 * frames_u8 [N,H,W,7] uint8 HWC = [flow_x flow_y mv_x mv_y r g b] (already cropped/resized);
 * flip [N] bytes or NULL (non-zero: mirror the frame, x components of flow and MV -> 256 - v);
 * outputs input_flow [N,2,H,W], input_mv [N,2,H,W], input_residual [N,3,H,W] fp32.
 * std4_host: HOST array {mean(std), std_r, std_g, std_b} as fp32 (what torch computes from
 * [0.229, 0.224, 0.225]).  workspace: dmc_prepare_inputs_workspace_bytes() bytes.
 */
size_t dmc_prepare_inputs_workspace_bytes(int N, int H, int W, int flow_ds_factor);
int dmc_prepare_inputs(const unsigned char* frames_u8, const unsigned char* flip, float* out_flow,
                       float* out_mv, float* out_res, float* workspace, int N, int H, int W,
                       int flow_ds_factor, const float* std4_host, dmc_stream_t stream);

/* Crop + bilinear resize + flip + blockify + normalise on the device: the whole of
 * GroupMultiScaleCrop / GroupCenterCrop / GroupScale (code/dmcnet/transforms.py:36-46,62-78,
 * 117-139), GroupRandomHorizontalFlip (:47-58) and dataset.py:215-263 except the random draws,
 * which stay on the host (crop box, flip bit).
 * frames_u8 [N,H0,W0,7] uint8 as decoded, allocated with >= 16 spare bytes behind the last frame
 * (the 4-pixel path reads whole aligned dwords); boxes [N,8] int32 DEVICE array
 * (y0, x0, h, w, rh, rw, cy, cx): the box (y0, x0, h, w) of each frame is resized to rh x rw with the
 * half-pixel-centre bilinear rule (bit-identical to this package's transforms.resize_bilinear;
 * rh x rw == h x w copies) and the OH x OW window at (cy, cx) of that image is produced
 * (crop -> resize: cy = cx = 0, rh x rw = OH x OW; scale -> centre crop: the window); NULL = the
 * whole frame, which must then measure OH x OW; flip as above (applied last); outputs
 * [N,2,OH,OW], [N,2,OH,OW], [N,3,OH,OW] fp32.  workspace: dmc_prepare_crop_workspace_bytes().
 */
size_t dmc_prepare_crop_workspace_bytes(int N, int OH, int OW, int flow_ds_factor);
int dmc_prepare_inputs_crop(const unsigned char* frames_u8, const int* boxes, const unsigned char* flip,
                            float* out_flow, float* out_mv, float* out_res, float* workspace, int N,
                            int H0, int W0, int OH, int OW, int flow_ds_factor, const float* std4_host,
                            dmc_stream_t stream);

/* ---- classifier stem: BatchNorm + ReLU + MaxPool2d(3, stride 2, padding 1) fused -----------------
 * Replaces `self.maxpool(self.relu(self.bn1(x)))` of the torchvision ResNet the reference builds at
 * code/dmcnet/model.py:305 (run at :352) and its autograd.  x [N,H,W,C] fp32 NHWC (conv1 output),
 * y_pool / d_pool [N,PH,PW,C] with PH = (H-1)/2+1, PW = (W-1)/2+1; the rectified full-resolution
 * tensor is never materialised.  Arg-max ties follow PyTorch (first element in row-major window
 * order).  stats: dmc_bn_act_stats_bytes(C) bytes, written by fwd (mean, invstd) and read by bwd;
 * scratch: dmc_bn_act_scratch_bytes(C) bytes, transient (NULL allowed when training == 0).
 * training = 0 normalises with the running statistics (forward only).
 * codes: dmc_bn_relu_pool_codes_bytes() bytes of scratch (the windows' arg-max positions, written by the
 * backward's first pass and read by its second), or NULL to recompute them in the second pass.
 */
int dmc_bn_relu_pool_supported(int N, int H, int W, int C);
int dmc_bn_relu_pool_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* y_pool, float* stats, void* scratch, int N, int H,
                         int W, int C, int training, float eps, float momentum, dmc_stream_t stream);
size_t dmc_bn_relu_pool_codes_bytes(int N, int H, int W, int C);
int dmc_bn_relu_pool_bwd(const float* x, const float* gamma, const float* beta, const float* stats,
                         void* scratch, const float* d_pool, float* dx, float* dgamma, float* dbeta,
                         void* codes, int N, int H, int W, int C, dmc_stream_t stream);

/* ---- NHWC convolutions on the fp32 matrix cores ---------------------------------------------------
 * Replace nn.Conv2d (3x3 padding 1, or 1x1 padding 0; stride 1 or 2; Cin, Cout multiples of 16) and
 * its autograd wherever the hot path runs them:
 *   - the PatchGAN discriminator blocks, code/dmcnet_GAN/model.py:254-279 (Conv2d(in, out, 3, stride,
 *     1) with bias, followed by LeakyReLU(0.2), Dropout2d(0.25), BatchNorm2d(out, eps=0.8)) as chained
 *     in Discriminator..Discriminator5, :282-438;
 *   - the torchvision ResNet the reference builds at code/dmcnet/model.py:305 and runs at :352
 *     (BasicBlock 3x3 convolutions and the 1x1 stride-2 shortcut convolutions, bias-free).
 * Layouts: activations NHWC fp32 ([N][H][W][C], the memory of a channels_last tensor), weights
 * OHWI ([Cout][KH][KW][Cin], the memory of a channels_last weight); OH = (H + 2 pad - KH) / stride + 1.
 * dmc_conv_nhwc_supported() says whether a shape is handled (callers use the stock op otherwise).
 *
 * fwd: y = conv(x, w) [+ bias] [LeakyReLU(0.2) if act == 1] [* keep[n][co]] -- the Dropout2d keep mask
 * [N][Cout], already divided by 1 - p -- and, if stat_partials != NULL, per-channel (sum, sum of squares)
 * of y per workgroup row, [stat_blocks = dmc_conv_nhwc_stat_blocks()][Cout][2] doubles (the launch FAILS if its grid has
 * another number of rows -- e.g. the tile-configuration option changed between the two calls -- instead of overrunning), which
 * dmc_conv_nhwc_stats_final() turns into the (mean, invstd) pair [2*C] and the running-statistics
 * update of the nn.BatchNorm2d that follows (count = N*OH*OW).
 * wpack: workspace of dmc_conv_nhwc_wt_bytes() bytes for the weights as the kernel wants them (needed when the
 * option "conv_arith" is 1; may be NULL otherwise).
 * dgrad: dx = conv_transpose(dy, w); wt = workspace of dmc_conv_nhwc_wt_bytes().
 * bf16x3 shapes (dmc_conv_nhwc_presplit_supported): dmc_conv_nhwc_split() fills the forward's and the data gradient's workspaces
 * in one launch, and fwd / dgrad called with w == NULL use theirs as it is (one split launch per layer and step instead of two).
 * Arithmetic (option "conv_arith"): 0 = v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 accumulate); 1 = "bf16x3" for
 * Cin % 32 == 0, Cout % 64 == 0: every fp32 operand is the exact sum of three bf16 slices, each product is formed
 * from six slice products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the omitted three are below 2^-23 of
 * |a||b|) -- the error against an fp64 evaluation is that of the fp32 instruction or below
 * (tools/conv_x3_check.py, tests), at 2.67x its rate.
 * wgrad: dw (OHWI) = sum over pixels, deterministic (fixed-order split-K reduction, no atomics);
 * workspace of dmc_conv_nhwc_wgrad_bytes().
 */
int dmc_conv_nhwc_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int dmc_conv_nhwc_stat_blocks(int N, int H, int W, int Cin, int Cout, int KH, int stride, int pad);
int dmc_conv_nhwc_fwd(const float* x, const float* w, void* wpack, const float* bias, const float* keep, float* y,
                      double* stat_partials, int stat_blocks, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                      int pad, int act, dmc_stream_t stream);
int dmc_conv_nhwc_stats_final(const double* partials, int nblk, int C, long count, float* stats,
                              float* running_mean, float* running_var, float eps, float momentum,
                              dmc_stream_t stream);
size_t dmc_conv_nhwc_wt_bytes(int Cin, int Cout, int KH, int KW);
int dmc_conv_nhwc_presplit_supported(int Cin, int Cout);
int dmc_conv_nhwc_split(const float* w, void* wpack_f, void* wpack_t, int Cin, int Cout, int KH, int KW, dmc_stream_t stream);
int dmc_conv_nhwc_dgrad(const float* dy, const float* w, float* wt, float* dx, int N, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream);
/* dx = data gradient + addend (the residual branch's gradient of a torchvision BasicBlock, resnet.py BasicBlock.forward:
 * `out += identity`), added in the convolution's epilogue; stride 1 and dmc_conv_nhwc_presplit_supported shapes only. */
int dmc_conv_nhwc_dgrad_add(const float* dy, const float* w, float* wt, const float* addend, float* dx, int N, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream);
size_t dmc_conv_nhwc_wgrad_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int dmc_conv_nhwc_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, dmc_stream_t stream);

/* ---- 3x3 / stride-1 convolutions on PRE-SPLIT bf16x3 operands (conv_x3s.hip) ---------------------------------
 * Replaces: the stride-1 3x3 convolutions of the ResNet-18 classifier (torchvision BasicBlock behind
 *           code/dmcnet/model.py:305, run at :352) and their autograd, in the same bf16x3 arithmetic as
 *           dmc_conv_nhwc_* (fp32 products from three bf16 slices per operand), with the split moved to the PRODUCER of
 *           each activation: every operand arrives as a "slice tensor"
 *               bf16 [3 slices][C / 16][M pixels][16 channels]        (dmc_x3s_slices_bytes(M, C) = 6 M C bytes)
 *           whose three slices sum to the fp32 value exactly.  dmc_x3s_split / dmc_x3s_merge convert from / to the
 *           fp32 [M][C] (channels_last) tensor; the BatchNorm kernels below write slice tensors directly.
 * Weights: dmc_x3s_pack_weights() packs w [Cout][3][3][Cin] (channels_last memory of the PyTorch weight) once per step
 *           into the forward's and / or the data gradient's image (dmc_x3s_wpack_bytes() each; either may be NULL).
 * fwd:     y [M][Cout] fp32 = conv3x3(x, w), padding 1, stride 1; stat_partials (nullable) receives the per-channel
 *           (sum, sum of squares) of y per workgroup row, [stat_blocks = dmc_x3s_conv_stat_blocks()][Cout][2] doubles, for
 *           dmc_conv_nhwc_stats_final(); a launch whose grid has another number of rows is refused (no overrun).
 * dgrad:   dx [M][Cin] fp32 = conv_transpose(dy, w) [+ addend, nullable: the residual branch's gradient].
 * Cin % 64 == 0, Cout % 64 == 0; dmc_x3s_conv_supported() says whether a shape is handled.  Deterministic. */
size_t dmc_x3s_slices_bytes(long M, int C);
int dmc_x3s_split(const float* x, void* xs, long M, int C, dmc_stream_t stream);
int dmc_x3s_merge(const void* xs, float* x, long M, int C, dmc_stream_t stream);
size_t dmc_x3s_wpack_bytes(int Cin, int Cout);
int dmc_x3s_pack_weights(const float* w, void* wpack_f, void* wpack_t, int Cin, int Cout, dmc_stream_t stream);
int dmc_x3s_conv_supported(int N, int H, int W, int Cin, int Cout);
int dmc_x3s_conv_stat_blocks(int N, int H, int W, int Cout);
int dmc_x3s_conv_fwd(const void* xs, const void* wpack_f, float* y, double* stat_partials, int stat_blocks, int N, int H, int W,
                     int Cin, int Cout, dmc_stream_t stream);
int dmc_x3s_conv_dgrad(const void* dys, const void* wpack_t, const float* addend, float* dx, int N, int H, int W, int Cin,
                       int Cout, dmc_stream_t stream);
/* wgrad: dw [Cout][3][3][Cin] fp32 (channels_last memory of the PyTorch gradient) from the slice tensors of x and dy;
 * workspace of dmc_x3s_conv_wgrad_bytes() (per-workgroup-row partials, summed in fixed order).  W in {56, 28, 14, 7}. */
int dmc_x3s_conv_wgrad_supported(int N, int H, int W, int Cin, int Cout);
size_t dmc_x3s_conv_wgrad_bytes(int N, int H, int W, int Cin, int Cout);
int dmc_x3s_conv_wgrad(const void* xs, const void* dys, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout,
                       dmc_stream_t stream);
/* Stride-2 3x3 data gradient on pre-split operands in ONE launch (four input-parity classes, nine (tap, class) pairs):
 * dys = slice tensor of dy [N][OH][OW][Cout]; dx [N][2 OH][2 OW][Cin] fp32; weights from dmc_x3s_pack_weights_s2
 * (dmc_x3s_wpack_bytes()).  Replaces the stride-2 convolutions' backward of torchvision's BasicBlock (layerN.0.conv1). */
int dmc_x3s_conv_dgrad_s2_supported(int N, int OH, int OW, int Cin, int Cout);
int dmc_x3s_pack_weights_s2(const float* w, void* wpack_t2, int Cin, int Cout, dmc_stream_t stream);
int dmc_x3s_conv_dgrad_s2(const void* dys, const void* wpack_t2, float* dx, int N, int OH, int OW, int Cin, int Cout,
                          dmc_stream_t stream);


/* Producers of slice tensors (bn_act.hip): the BatchNorm kernels above with the result written as fp32 (nullable) and / or
 * as a bf16x3 slice tensor (nullable), C % 16 == 0 -- the forward's activation for the next convolution
 * (dmc_bn_apply_act_nhwc / dmc_bn_relu_pool_fwd semantics), the backward's convolution-output gradient (dmc_bn_act_bwd
 * semantics) for that convolution's dmc_x3s_conv_dgrad / dmc_x3s_conv_wgrad. */
int dmc_bn_apply_act_x3s(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                         float* y, void* ys, unsigned char* relu_mask, int M, int C, int relu, dmc_stream_t stream);
int dmc_bn_act_bwd_x3s(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                       void* scratch, const float* dy, float* dx, void* dxs, float* dresidual, float* dgamma, float* dbeta,
                       const unsigned char* relu_mask, int M, int C, int relu, dmc_stream_t stream);
int dmc_bn_relu_pool_fwd_x3s(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float* y_pool, void* ys, float* stats, void* scratch, int N, int H, int W, int C, int training,
                             float eps, float momentum, dmc_stream_t stream);
/* The stem tail with an arg-max record (replaces torchvision's bn1 / relu / maxpool behind
 * /root/reference/code/dmcnet/model.py:305, as dmc_bn_relu_pool_fwd does): dmc_bn_relu_pool_fwd_arg =
 * dmc_bn_relu_pool_fwd_x3s that also writes, when `codes` / `xmax` are given (both or neither), each window's arg-max
 * position per channel (codes: dmc_bn_relu_pool_codes_bytes) and the raw input there (xmax: fp32 [N][PH][PW][C]);
 * dmc_bn_relu_pool_bwd_arg = dmc_bn_relu_pool_bwd from that record: its BatchNorm-backward sums stream over the
 * pooled-size (d_pool, xmax) pair instead of gathering over the 4x larger input.  Same dx / dgamma / dbeta semantics.
 * stat_split: 0 = reduce the batch statistics of x here; > 0 = `scratch` already holds that many partial sums per channel
 * (written by x's producer, dmc_stem_fwd_x3_stats). */
int dmc_bn_relu_pool_fwd_arg(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float* y_pool, void* ys, void* codes, float* xmax, float* stats, void* scratch, int stat_split,
                             int N, int H, int W, int C, int training, float eps, float momentum, dmc_stream_t stream);
int dmc_bn_relu_pool_bwd_arg(const float* x, const float* gamma, const float* beta, const float* stats, void* scratch,
                             const float* d_pool, const void* codes, const float* xmax, float* dx, float* dgamma, float* dbeta,
                             int N, int H, int W, int C, dmc_stream_t stream);
/* ---- the stride-2 residual blocks on PRE-SPLIT operands (conv_x3q.hip) ------------------------------------------------
 * Replaces: torchvision BasicBlock `conv1` (3x3, stride 2, padding 1) together with `downsample[0]` (1x1, stride 2) of
 *           layer2.0 / layer3.0 / layer4.0 -- the two convolutions that read the block input, built at
 *           code/dmcnet/model.py:305 and run at :352 -- and their autograd, in the bf16x3 arithmetic of dmc_x3s_*, ONE
 *           launch per direction for the pair (the shortcut reads x[2a][2b] = the centre tap's operand).
 * Input:    the block input as a SPACE-TO-DEPTH slice tensor ("s2d"; H, W even, OH = H / 2, OW = W / 2, Mq = N OH OW):
 *               bf16 [3 slices][4 parity classes 2 py + px][Cin / 16][Mq pixels][16 channels]       (6 N H W Cin bytes)
 *           class (py, px) holds the pixels (2a + py, 2b + px); written by the input's producer (dmc_bn_apply_act_x3q) or
 *           by dmc_x3q_split; dmc_x3q_merge is the exact inverse.
 * Weights:  dmc_x3q_pack_weights(w3 [Cout][3][3][Cin], w1 [Cout][Cin] -- channels_last memory of the two PyTorch weights)
 *           fills the forward's and / or the data gradient's image (dmc_x3q_wpack_bytes() each; either may be NULL).
 * fwd:      y3 [Mq][Cout] = conv3x3 stride 2 (x, w3), y1 [Mq][Cout] = conv1x1 stride 2 (x, w1), fp32; stat_partials3 / 1
 *           (nullable) receive the per-channel (sum, sum of squares) per workgroup row, [stat_blocks =
 *           dmc_x3q_stat_blocks()][Cout][2] doubles each, for dmc_conv_nhwc_stats_final().
 * dgrad:    dx [N][H][W][Cin] fp32 = conv_transpose(dy3, w3) + conv_transpose(dy1, w1) from the ORDINARY slice tensors of
 *           the two output gradients ([3][Cout / 16][Mq][16], e.g. from dmc_bn_act_bwd_x3s); every element of dx is written.
 * wgrad:    dw3 [Cout][3][3][Cin], dw1 [Cout][Cin] from the s2d input and the two output-gradient slice tensors;
 *           workspace of dmc_x3q_conv_wgrad_bytes(); deterministic (fixed-order reduction of per-workgroup partials).
 * Cin, Cout multiples of 64; dmc_x3q_supported() says whether a shape is handled (OW in {28, 14, 7} for the weight
 * gradient, as the classifier has them). */
size_t dmc_x3q_wpack_bytes(int Cin, int Cout);
int dmc_x3q_supported(int N, int OH, int OW, int Cin, int Cout);
int dmc_x3q_stat_blocks(int N, int OH, int OW, int Cout);
int dmc_x3q_split(const float* x, void* xq, int N, int H, int W, int C, dmc_stream_t stream);
int dmc_x3q_merge(const void* xq, float* x, int N, int H, int W, int C, dmc_stream_t stream);
int dmc_x3q_pack_weights(const float* w3, const float* w1, void* wpack_f, void* wpack_t, int Cin, int Cout, dmc_stream_t stream);
int dmc_x3q_conv_fwd(const void* xq, const void* wpack_f, float* y3, float* y1, double* stat_partials3, double* stat_partials1,
                     int stat_blocks, int N, int OH, int OW, int Cin, int Cout, dmc_stream_t stream);
int dmc_x3q_conv_dgrad(const void* dys3, const void* dys1, const void* wpack_t, float* dx, int N, int OH, int OW, int Cin, int Cout,
                       dmc_stream_t stream);
/* dmc_bn_apply_act_x3s (BatchNorm apply [+ residual] [+ ReLU] with given statistics) whose slice output yq is the s2d
 * tensor over the [N][H][W] grid; y (fp32 NHWC, nullable) as usual. */
int dmc_bn_apply_act_x3q(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                         float* y, void* yq, unsigned char* relu_mask, int N, int H, int W, int C, int relu, dmc_stream_t stream);
int dmc_x3q_conv_wgrad_supported(int N, int OH, int OW, int Cin, int Cout);
size_t dmc_x3q_conv_wgrad_bytes(int N, int OH, int OW, int Cin, int Cout);
int dmc_x3q_conv_wgrad(const void* xq, const void* dys3, const void* dys1, float* dw3, float* dw1, float* workspace, int N, int OH,
                       int OW, int Cin, int Cout, dmc_stream_t stream);

/* BatchNorm-backward sums from the data gradient's epilogue: dmc_x3s_conv_dgrad_bnb = dmc_x3s_conv_dgrad that also
 * reduces, for the unit whose output gradient it writes (dx), dbeta = sum(d) and dgamma = sum(d * xhat) (d = dx, zeroed
 * where that unit's ReLU was off; partials: dmc_x3s_conv_stat_blocks(N, H, W, Cin) x Cin x 2 doubles);
 * dmc_bn_act_bwd_x3s_apply = the second half of dmc_bn_act_bwd_x3s with those sums as inputs (no reduction pass). */
int dmc_x3s_conv_dgrad_bnb(const void* dys, const void* wpack_t, const float* addend, float* dx, const float* bn_y,
                           const float* bn_stats, const float* bn_gamma, const float* bn_beta, const unsigned char* bn_relu_mask,
                           int bn_relu, double* partials, int stat_blocks, float* dgamma, float* dbeta, int N, int H, int W, int Cin,
                           int Cout, dmc_stream_t stream);
int dmc_bn_act_bwd_x3s_apply(const float* x, const float* residual, const float* gamma, const float* beta, const float* stats,
                             const float* dy, float* dx, void* dxs, float* dresidual, const float* dgamma, const float* dbeta,
                             const unsigned char* relu_mask, int M, int C, int relu, dmc_stream_t stream);




/* ---- first discriminator block: Conv2d(2, Cout, 3, stride 2, padding 1) on the NCHW cue ------------
 * Replaces the convolution (+ LeakyReLU(0.2) + Dropout2d keep mask) of `discriminator_block(ch_in, 16,
 * bn=False)`, code/dmcnet_GAN/model.py:254-265 as used at :287,:308,:334,:371,:400, and its autograd.
 * x [M,2,H,W] NCHW fp32 (the generated / TV-L1 flow), w [Cout,2,3,3] (PyTorch's layout), Cout 16 (8 for
 * Discriminator4); z, g: [M,OH,OW,Cout] NHWC, OH = (H-1)/2+1.  keep: [M,Cout] or NULL; act != 0
 * applies LeakyReLU(0.2).  wgrad writes dw [Cout,2,3,3] and db [Cout] (NULL to skip) deterministically;
 * workspace: dmc_disc_first_wgrad_bytes(Cout).
 */
int dmc_disc_first_supported(int Cout);
int dmc_disc_first_fwd(const float* x, const float* w, const float* bias, const float* keep, float* z, int M,
                       int H, int W, int Cout, int act, dmc_stream_t stream);
int dmc_disc_first_dgrad(const float* g, const float* w, float* dx, int M, int H, int W, int Cout,
                         dmc_stream_t stream);
size_t dmc_disc_first_wgrad_bytes(int Cout);
int dmc_disc_first_wgrad(const float* x, const float* g, float* dw, float* db, void* workspace, int M, int H,
                         int W, int Cout, dmc_stream_t stream);

/* ---- classifier stem: weight gradient of conv1 (2 -> 64 channels, 7x7, stride 2, pad 3) ----------
 * Replaces what autograd computes for the conv1 the reference installs for the 2-channel flow
 * input, code/dmcnet/model.py:285-294 (nn.Conv2d(2, 64, 7, stride=2, padding=3, bias=False)),
 * inside loss.backward(), code/dmcnet/train.py:258.  (The forward convolution and, when the input
 * needs a gradient, the data gradient stay on MIOpen.)
 * x [N,2,H,W] fp32 NCHW; dy [N,OH,OW,64] fp32 NHWC (channels_last storage of the [N,64,OH,OW]
 * gradient), OH = (H+1)/2, OW = (W+1)/2; dw [64,2,7,7] fp32 contiguous.  Requires W % 4 == 0
 * (dmc_stem_wgrad_supported).  partials: dmc_stem_wgrad_partials_bytes(N,H,W) bytes.
 * Deterministic (fixed summation order).
 */
int dmc_stem_wgrad_supported(int H, int W);
size_t dmc_stem_wgrad_partials_bytes(int N, int H, int W);
int dmc_stem_wgrad(const float* x, const float* dy, float* dw, float* partials, int N, int H, int W,
                   dmc_stream_t stream);
/* forward of the same convolution (replaces F.conv2d(x, conv1.weight, None, 2, 3) behind base_model(input),
 * code/dmcnet/model.py:352): y [N,OH,OW,64] NHWC fp32; w [64,2,7,7] by element strides (contiguous: 98, 49, 7, 1;
 * channels_last: 98, 1, 14, 2).  Exact fp32 (v_mfma_f32_32x32x2_f32), any H, W. */
int dmc_stem_fwd(const float* x, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, float* y, int N, int H, int W,
                 dmc_stream_t stream);
/* the same forward in bf16x3 arithmetic (fp32 operands as three bf16 slices, six bf16 MFMAs per product block, fp32
 * accumulate: fp32-level error); workspace: dmc_stem_fwd_x3_workspace_bytes(N, H, W) (padded slice volumes + split weights) */
size_t dmc_stem_fwd_x3_workspace_bytes(int N, int H, int W);
int dmc_stem_fwd_x3(const float* x, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, void* workspace, float* y, int N,
                    int H, int W, dmc_stream_t stream);
/* dmc_stem_fwd_x3 that also reduces the BatchNorm batch statistics of its output in the epilogue (the stem's bn1,
 * /root/reference/code/dmcnet/model.py:305): stat_scratch (dmc_bn_act_scratch_bytes(64), nullable = plain forward)
 * receives dmc_stem_fwd_x3_stat_blocks(N, H, W) partial (sum, sum of squares) pairs per channel in the layout the
 * BatchNorm kernels reduce; pass that count as `stat_split` to dmc_bn_relu_pool_fwd_arg, which then skips its own
 * statistics pass over the (4x pooled-size) convolution output. */
int dmc_stem_fwd_x3_stat_blocks(int N, int H, int W);
int dmc_stem_fwd_x3_stats(const float* x, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, void* workspace, float* y,
                          void* stat_scratch, int N, int H, int W, dmc_stream_t stream);
/* Data gradient of the stem convolution (the gradient of the 2-channel cue: only the GAN variant needs it, where the
 * classifier's loss reaches the generator -- /root/reference/code/dmcnet_GAN/model.py:557-561; replaces the library
 * GEMM + col2im of autograd's conv2d backward): dx [N][2][H][W] fp32 contiguous from dy [N][OH][OW][64] fp32 (the memory
 * of a channels_last [N,64,OH,OW] tensor, OH = (H + 1) / 2, OW = (W + 1) / 2) and w [64,2,7,7] addressed by its element
 * strides; stride 2, padding 3; bf16x3 arithmetic (fp32-level error), deterministic; W <= 256.  workspace:
 * dmc_stem_dgrad_workspace_bytes() (the pre-split weights). */
size_t dmc_stem_dgrad_workspace_bytes(void);
int dmc_stem_dgrad_supported(int H, int W);
int dmc_stem_dgrad(const float* dy, const float* w, long ws_co, long ws_ci, long ws_ky, long ws_kx, void* workspace, float* dx, int N,
                   int H, int W, dmc_stream_t stream);

/* ---- I3D trunk: bf16 3-D convolutions on the matrix cores (BASELINE config 5) -------------------------
 * Replace nn.Conv3d and its autograd inside the reference's Unit3Dpy, code/dmcnet_I3D/network/i3d.py:328-403
 * (self.conv3d at :372-388, run at :393), for the trunk's stride-1 TF-"SAME" units: conv3d_2b_1x1, conv3d_2c_3x3
 * (:484-486) and the 1x1x1 / 3x3x3 branches of the nine Mixed blocks (:421-455, built at :491-513), as the
 * reference runs them in bf16/fp16-style mixed precision (fp32 master weights, 16-bit activations).
 * Layouts: activations bf16 NDHWC ([N][D][H][W][C], the memory of a channels_last_3d tensor); weights fp32
 * [Cout][Cin][KD][KH][KW] addressed by element strides (w_s_co, w_s_ci, w_s_tap: 1 for a contiguous parameter),
 * rounded to bf16 (nearest even) on the way into the packed workspace; fp32 accumulation; bf16 results rounded to
 * nearest even.  Kernel extents 1 or 3 per dimension, stride 1, zero padding k/2; Cin, Cout multiples of 8.
 * fwd: y = conv3d(x, w); if stat_partials != NULL also per-channel (sum, sum of squares) of the rounded y per
 * workgroup row, [dmc_conv3d_bf16_stat_blocks()][Cout][2] floats (for the BatchNorm3d that follows).
 * dgrad: dx = conv_transpose3d(dy, w).  wgrad: dw fp32 [Cout][Cin][KD][KH][KW] contiguous = sum over pixels
 * (1x1x1 and 3x3x3 only), deterministic split-K reduction; workspace of dmc_conv3d_bf16_wgrad_bytes().
 * wpack: workspace of dmc_conv3d_bf16_wpack_bytes() bytes; dmc_conv3d_bf16_pack() fills the forward's and the data
 * gradient's workspaces in one launch, and fwd / dgrad called with w == NULL use theirs as it is.
 */
int dmc_conv3d_bf16_supported(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW);
size_t dmc_conv3d_bf16_wpack_bytes(int Cin, int Cout, int KD, int KH, int KW);
int dmc_conv3d_bf16_pack(const float* w, long w_s_co, long w_s_ci, long w_s_tap, void* wpack_f, void* wpack_b, int Cin, int Cout,
                         int KD, int KH, int KW, dmc_stream_t stream);
int dmc_conv3d_bf16_stat_blocks(int N, int D, int H, int W, int Cout);
/* rows of stat_partials dmc_conv3d_bf16_fwd writes for THIS layer (the 3x3x3 layers run a patch-resident kernel whose
 * workgroups are position tiles of one (n, d) plane; everything else: dmc_conv3d_bf16_stat_blocks) */
int dmc_conv3d_bf16_stat_blocks_k(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW);
int dmc_conv3d_bf16_fwd(const void* x, const float* w, long w_s_co, long w_s_ci, long w_s_tap, void* wpack, void* y,
                        float* stat_partials, int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW,
                        dmc_stream_t stream);
int dmc_conv3d_bf16_dgrad(const void* dy, const float* w, long w_s_co, long w_s_ci, long w_s_tap, void* wpack, void* dx,
                          int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW, dmc_stream_t stream);
size_t dmc_conv3d_bf16_wgrad_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW);
int dmc_conv3d_bf16_wgrad(const void* x, const void* dy, float* dw, float* workspace, int N, int D, int H, int W, int Cin,
                          int Cout, int KD, int KH, int KW, dmc_stream_t stream);

/* ---- I3D trunk: MaxPool3dTFPadding on bf16 NDHWC tensors -------------------------------------------
 * Replaces the reference's MaxPool3dTFPadding, code/dmcnet_I3D/network/i3d.py:406-418 (ConstantPad3d with zeros
 * to the TF-"SAME" extent, then nn.MaxPool3d(kernel, stride, ceil_mode=True)), and its autograd, as used at
 * :482,:488,:494,:505 and in every Mixed block (:441-443).  x [N,D,H,W,C] bf16 (channels_last_3d memory), C % 8 == 0;
 * front padding = max(k - s, 0) / 2 per dimension; the scan order and tie rule are nn.MaxPool3d's.  fwd also writes
 * one byte per output value (the winning tap; 255 = a padding zero) which bwd gathers from: no atomics,
 * deterministic, fp32 sums rounded once.  dmc_maxpool3d_tf_out_shape() returns 0 for unsupported arguments.
 */
int dmc_maxpool3d_tf_out_shape(int D, int H, int W, int C, int kd, int kh, int kw, int sd, int sh, int sw, int* od, int* oh,
                               int* ow);
int dmc_maxpool3d_tf_bf16_fwd(const void* x, void* y, void* code, int N, int D, int H, int W, int C, int kd, int kh, int kw,
                              int sd, int sh, int sw, dmc_stream_t stream);
int dmc_maxpool3d_tf_bf16_bwd(const void* dy, const void* code, void* dx, int N, int D, int H, int W, int C, int kd, int kh,
                              int kw, int sd, int sh, int sw, dmc_stream_t stream);

/* ---- I3D trunk: BatchNorm3d (+ ReLU) on bf16 NDHWC tensors, training mode ------------------------------
 * Replaces `self.batch3d(out)` and `F.relu` of the reference's Unit3Dpy.forward, code/dmcnet_I3D/network/i3d.py:394-398
 * (torch.nn.BatchNorm3d, batch statistics, running-statistics update), and their autograd, behind a convolution
 * that ran dmc_conv3d_bf16_fwd with stat_partials: y [M,C] bf16 (M = N*D*H*W), partials [nblk][C][2] floats.
 * fwd: stats [2*C] = (mean, invstd) (biased variance + eps), running_mean / running_var updated with `momentum`
 * (unbiased variance; NULL to skip), out = relu?(gamma * (y - mean) * invstd + beta) in bf16.
 * bwd: dy = gradient of y, dgamma, dbeta (fp32) from dout (pixel stride dout_ld elements: C, or the width of the
 * concatenated Inception output when dout is a channel slice of its gradient); scratch of dmc_bn3d_bf16_scratch_bytes(C).  fp32 per
 * element, fp64 per-channel sums in fixed order (deterministic).  C % 8 == 0, C <= 2048.
 */
int dmc_bn3d_bf16_supported(long M, int C);
size_t dmc_bn3d_bf16_scratch_bytes(int C);
int dmc_bn3d_bf16_fwd(const void* y, const float* partials, int nblk, const float* gamma, const float* beta, float* stats,
                      float* running_mean, float* running_var, void* out, long M, int C, int relu, float eps, float momentum,
                      dmc_stream_t stream);
/* the same with `out` a channel slice of a wider NDHWC tensor: out_ld = elements between consecutive pixels of `out` (a multiple
 * of 8, >= C; `out` 16-byte aligned) -- an Inception branch writes its part of the block's output in place (no concatenation) */
int dmc_bn3d_bf16_fwd_ld(const void* y, const float* partials, int nblk, const float* gamma, const float* beta, float* stats,
                         float* running_mean, float* running_var, void* out, long out_ld, long M, int C, int relu, float eps,
                         float momentum, dmc_stream_t stream);
/* out = ((a + b) + c) + d on bf16 arrays of n elements (n % 8 == 0, pointers 16-byte aligned), every sum rounded to bf16 (nearest
 * even) -- bit for bit what three additions of bf16 tensors give: the gradient of an Inception block's input
 * (network/i3d.py:449-454: four branches read x) in one pass instead of three. */
int dmc_add4_bf16(const void* a, const void* b, const void* c, const void* d, void* out, long n, dmc_stream_t stream);
int dmc_bn3d_bf16_bwd(const void* dout, long dout_ld, const void* y, const float* stats, const float* gamma, const float* beta,
                      float* scratch, void* dy, float* dgamma, float* dbeta, long M, int C, int relu, dmc_stream_t stream);

/* A whole Unit3Dpy (/root/reference/code/dmcnet_I3D/network/i3d.py:328-403: Conv3d -> BatchNorm3d(training) [-> ReLU]) per
 * call: dmc_unit3d_bf16_fwd = dmc_conv3d_bf16_pack + dmc_conv3d_bf16_fwd (statistics in its epilogue) + dmc_bn3d_bf16_fwd,
 * dmc_unit3d_bf16_bwd = dmc_bn3d_bf16_bwd + dmc_conv3d_bf16_dgrad (dx nullable) + dmc_conv3d_bf16_wgrad (dw nullable),
 * with every intermediate carved out of one workspace per direction (dmc_unit3d_bf16_{fwd,bwd}_workspace_bytes; the
 * backward reads the forward's workspace: statistics and the data-gradient weight layout).  w: fp32 [Cout][Cin][KD][KH][KW]
 * contiguous; x, y (convolution output), out, dout (pixel stride dout_ld elements), dy, dx: bf16 NDHWC; dw fp32 in w's
 * layout.  Same kernels in the same order as the separate calls: identical results; the trunk's ~940 launches per
 * micro-step are bound by the host that issues them, and this is a third of the foreign calls and allocations. */
size_t dmc_unit3d_bf16_fwd_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW);
size_t dmc_unit3d_bf16_bwd_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW);
int dmc_unit3d_bf16_fwd(const void* x, const float* w, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, void* workspace, void* y, void* out, int N, int D, int H, int W, int Cin, int Cout,
                        int KD, int KH, int KW, int relu, float eps, float momentum, dmc_stream_t stream);
/* dmc_unit3d_bf16_fwd with `out` a channel slice of a wider NDHWC tensor (see dmc_bn3d_bf16_fwd_ld) */
int dmc_unit3d_bf16_fwd_into(const void* x, const float* w, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, void* workspace, void* y, void* out, long out_ld, int N, int D, int H, int W, int Cin,
                             int Cout, int KD, int KH, int KW, int relu, float eps, float momentum, dmc_stream_t stream);
int dmc_unit3d_bf16_bwd(const void* dout, long dout_ld, const void* x, const void* y, const void* fwd_workspace, const float* gamma,
                        const float* beta, void* bwd_workspace, void* dy, void* dx, float* dw, float* dgamma, float* dbeta, int N,
                        int D, int H, int W, int Cin, int Cout, int KD, int KH, int KW, int relu, dmc_stream_t stream);

/* ---- I3D stem: forward of conv3d_1a_7x7 on the bf16 matrix cores -----------------------------------------
 * Replaces ConstantPad3d(TF-"SAME": front 2, back 3) + nn.Conv3d(2, 64, 7, stride 2) of the stem Unit3Dpy,
 * code/dmcnet_I3D/network/i3d.py:480-481 via :390-393, as run in 16-bit mixed precision: x [N,2,T,H,W] fp32 (the DMC
 * cue) and w [64,2,7,7,7] fp32 contiguous are rounded to bf16, fp32 accumulation, y [N,OD,OH,OW,64] bf16 NDHWC with
 * OD = (T + 5 - 7) / 2 + 1 (likewise OH, OW); stat_partials (NULL to skip): [dmc_stem3d_bf16_stat_blocks()][64][2]
 * floats = per-channel (sum, sum of squares) of the rounded outputs, for dmc_bn3d_bf16_fwd.  workspace:
 * dmc_stem3d_bf16_workspace_bytes().
 */
size_t dmc_stem3d_bf16_workspace_bytes(int N, int T, int H, int W);
int dmc_stem3d_bf16_stat_blocks(int N, int T, int H, int W);
int dmc_stem3d_bf16_fwd(const float* x, const float* w, void* workspace, void* y, float* stat_partials, int N, int T, int H, int W,
                        dmc_stream_t stream);
/* weight gradient of the same convolution: dw [64,2,7,7,7] fp32 contiguous from x (rounded to bf16 as in the forward) and
 * dy [N,OD,OH,OW,64] bf16 NDHWC; GEMM over pixels on the bf16 matrix cores (plane form for rows of up to 125 output pixels:
 * csrc/stem3d_bf16.hip, stem3d_w2_kernel), deterministic split-K reduction; workspace: dmc_stem3d_bf16_wgrad_workspace_bytes(). */
size_t dmc_stem3d_bf16_wgrad_workspace_bytes(int N, int T, int H, int W);
int dmc_stem3d_bf16_wgrad(const float* x, const void* dy, float* dw, void* workspace, int N, int T, int H, int W, dmc_stream_t stream);
/* data gradient of the same convolution (the gradient of the cue): dx [N,2,T,H,W] fp32 from dy [N,OD,OH,OW,64] bf16 NDHWC and
 * w (rounded to bf16); per input row a GEMM Q[ow][(kx,c)] over (window rows, channels) on the bf16 matrix cores and a 1-D fold
 * along x; W <= 256; deterministic; workspace: dmc_stem3d_bf16_dgrad_workspace_bytes(). */
size_t dmc_stem3d_bf16_dgrad_workspace_bytes(void);
int dmc_stem3d_bf16_dgrad(const void* dy, const float* w, float* dx, void* workspace, int N, int T, int H, int W, dmc_stream_t stream);

/* ---- Post-decode motion-vector / residual extraction ---------------------------------------
 * Replaces the integer loops of create_and_load_mv_residual(), code/dmcnet/data_loader/coviar_data_loader.c:71-175, and
 * the accumulator set-up of decode_video(), :306-319, on arrays the decoder hands over (the bitstream decode itself is
 * FFmpeg's and stays on the host).  `mvs`: DEVICE copy of the frame's AV_FRAME_DATA_MOTION_VECTORS side data, an array
 * of AVMotionVector (libavutil/motion_vector.h, public ABI: int32 source @0, uint8 w @4, h @5, int16 src_x @6, src_y @8,
 * dst_x @10, dst_y @12), `mv_stride` = the caller's sizeof(AVMotionVector) (24 before libavutil 55.63, 40 since).
 * Accumulators use the reference's TRANSPOSED layout, accu[x * H * 2 + y * 2 + c] (:107-108); MV planes are int32
 * [H][W][2] and residuals int32 [H][W][3] as its numpy arrays (:292-309); BGR frames uint8 [H][W][3].
 * Overlapping blocks: the LATER vector of the list wins, as in the sequential loop (realised as the per-pixel maximum of
 * the covering vectors' indices, resolved tile by tile in LDS; owner_ws, int32 [H][W], receives the winner's displacement
 * (src - dst) as two int16: order-independent, bit-exact).  Vectors with zero displacement are skipped (:88); a (block pixel) is written only when destination AND source are inside the frame (:100-103).
 * bad_source (nullable): device int32 incremented once per vector whose `source` is not -1 (the reference asserts, :86).
 *
 *   dmc_mv_accu_init   :311-318  accu = identity (x, y)
 *   dmc_mv_rasterise   :88-121, `accumulate == 0` branch: covered pixels of mv_out get (dst - src); the others keep
 *                      their value (the reference's array is zero-initialised once, :292-298)
 *   dmc_mv_accumulate  :88-127, `accumulate != 0` branch, ONE frame: accu_new[dst] = accu_old[src] where covered,
 *                      accu_old[dst] elsewhere (the reference's two buffers + memcpy; here the caller swaps the pointers)
 *   dmc_mv_from_accu   :130-139  mv_out[y][x] = (x, y) - accu[x][y]
 *   dmc_residual       :141-175  res[y][x][c] = bgr_cur[y][x][c] - bgr_ref[src_y][src_x][c], src from `accu`
 *                      (accumulate) or (x, y) - `mv` (not): pass exactly one of the two
 *   dmc_mv_gop_batch   a BATCH of accumulated chains in two launches: chain c owns the frames chain_off[c] ..
 *                      chain_off[c + 1] - 1 (in decode order), frame f the vectors frame_off[f] .. frame_off[f + 1] - 1
 *                      (device int32 arrays of n_chains + 1 / n_frames + 1 entries; owner_ws of dmc_mv_owner_bytes(n_frames,
 *                      H, W)).  Every pixel is walked back through its chain's owner maps -- the same values as n calls of
 *                      dmc_mv_accumulate from the identity -- and whichever of accu_out [n_chains][W][H][2], mv_out
 *                      [n_chains][H][W][2], res_out [n_chains][H][W][3] (bgr_ref / bgr_cur [n_chains][H][W][3]) is non-NULL
 *                      is written.  emit (nullable, device int32 [n_chains]): 0 = leave that chain's outputs untouched (the
 *                      reference's `cur_pos > 0` and `if (sd)` gates, :128, :363).  A one-frame chain is also the
 *                      non-accumulating case (mv = dst - src of the owner, 0 elsewhere).
 */
size_t dmc_mv_owner_bytes(int n_frames, int H, int W);
int dmc_mv_accu_init(int32_t* accu, int H, int W, dmc_stream_t stream);
int dmc_mv_rasterise(const void* mvs, int mv_stride, int n_mv, int32_t* owner_ws, int32_t* mv_out, int32_t* bad_source, int H,
                     int W, dmc_stream_t stream);
int dmc_mv_accumulate(const void* mvs, int mv_stride, int n_mv, int32_t* owner_ws, const int32_t* accu_old, int32_t* accu_new,
                      int32_t* bad_source, int H, int W, dmc_stream_t stream);
int dmc_mv_from_accu(const int32_t* accu, int32_t* mv_out, int H, int W, dmc_stream_t stream);
int dmc_residual(const uint8_t* bgr_ref, const uint8_t* bgr_cur, const int32_t* accu, const int32_t* mv, int32_t* res, int H, int W,
                 dmc_stream_t stream);
int dmc_mv_gop_batch(const void* mvs, int mv_stride, int n_mv, const int32_t* frame_off, int n_frames, const int32_t* chain_off,
                     int n_chains, const int32_t* emit, int32_t* owner_ws, const uint8_t* bgr_ref, const uint8_t* bgr_cur,
                     int32_t* accu_out, int32_t* mv_out, int32_t* res_out, int32_t* bad_source, int H, int W, dmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DMCNET_HIP_H */
