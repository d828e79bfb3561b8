/* C ABI of libvc_hip.so -- hand-written gfx950 (MI355X) kernels for the video-classification hot path.
 *
 * The reference (paci12/video-classification) has no FFI: its boundary is the Python nn.Module API and
 * everything underneath is stock torch.nn.  Each entry point below therefore names the torch operator
 * call site in the reference that it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - Plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (in practice a
 *     torch tensor's data_ptr()).  The library never allocates or frees device memory.
 *   - Every call enqueues work on `stream` and returns immediately; buffers must stay alive until the
 *     stream reaches the call.  Calls are re-entrant per stream; the only library state is the
 *     thread-local error string.
 *   - Return value: 0 on success, negative vc status otherwise (VC_ERR_SHAPE -1, VC_ERR_DTYPE -2,
 *     VC_ERR_HIP -3, VC_ERR_ARG -4); vc_last_error() then describes the failure.  Nothing is thrown.
 *   - dtype: VC_F32 (0) = exact fp32 path (plain FMA kernels), VC_BF16 (1) = bf16 storage, MFMA with
 *     fp32 accumulation.  Activations are channels-last: [N][D][H][W][C] (convolutions) or [M][C]
 *     (linear), C a multiple of 8.
 *   - Convolution descriptor `desc` = 18 ints:
 *       n, di, hi, wi, ci,  do, ho, wo, co,  kd, kh, kw,  sd, sh, sw,  pd, ph, pw
 *     with ci / co the PHYSICAL (padded) channel counts of x / y.  A Linear layer is the descriptor
 *     {M,1,1,1,K, 1,1,1,N, 1,1,1, 1,1,1, 0,0,0}.
 *   - Packed weights: forward [co][taps][ci], data-gradient [ci_true][taps][co]; taps = kd*kh*kw with
 *     kw fastest.  vc_pack_weight produces both from the torch layout [Co][Ci][taps] (fp32).
 */
#ifndef VC_HIP_H
#define VC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* vc_stream_t; /* hipStream_t */

#define VC_F32 0
#define VC_BF16 1

/* ---- library ---- */
int vc_version(void);
const char* vc_last_error(void);
/* Deterministic mode, the counterpart of torch.backends.cudnn.deterministic = True in set_deterministic
 * (utils/common/training_utils.py:27-35).  on != 0: sums that blocks otherwise combine with fp32 atomics in whatever
 * order they finish (weight / bias gradients, BatchNorm statistics and backward reductions, column sums, the mean
 * loss) are added in a fixed order, so that a training step repeats bit for bit.  workspace: device memory of the
 * CURRENT device (at least 2^20 floats; 2^26 cover the ResNet-50 CRNN step at 32 clips without cutting a weight
 * gradient into fewer splits), owned by the caller, kept alive while the mode is on; call once per device, and issue
 * the library's calls on one stream at a time per device.  on == 0 switches the mode off on every device.  The mode's
 * state is safe to switch from another host thread, but a workspace must outlive every call already issued with it:
 * do not switch the mode off (and free the workspace) while another thread is inside a library call.  Covered
 * besides the above: LayerNorm gamma / beta gradients and the relative-position-bias gradient of window attention
 * (bf16 path).  Not covered: the CBAM gradients of the fusion CNN and the fp32 attention kernel. */
int vc_set_deterministic(int on, float* workspace, size_t workspace_floats);
int vc_get_deterministic(void);

/* ---- dense contractions (MFMA implicit GEMM) ------------------------------------------------------
 * Replace nn.Conv2d (utils/common/model_components.py:91,96,101,106; torchvision ResNet convs behind
 * :159-165), nn.Conv3d (models/Conv3D/UCF101_3DCNN_refactored.py:66,72,78,84), nn.Linear
 * (model_components.py:110,114,115,166-170,216-217,273-277,425,484-489; Conv3D :96,99) and the
 * gate products inside nn.LSTM (:209-214, 482-483), forward and backward.
 * ld* = 0 means dense rows.  beta != 0 adds the result to what `y`/`dx` already hold. */
int vc_conv_fwd(const int* desc, int dtype, const void* x, int ldx, const void* w_packed, const float* bias, void* y,
                int ldy, int relu, int out_f32, int beta, vc_stream_t stream);
/* Forward convolution that also produces the BatchNorm statistics of its output (conv -> BN pairs of the ResNet
 * trunk, model_components.py:164-176; torchvision Bottleneck): bn_partial[slab][0][c] / [slab][1][c] receive the sum
 * and sum of squares of y over slab = 64 consecutive output rows, [ceil(M/64)][2][co] fp32, every entry written.
 * vc_conv_fwd_stats_supported() says whether a geometry can take this path (1) or needs vc_conv_fwd + vc_bn_stats (0). */
int vc_conv_fwd_stats_supported(const int* desc, int dtype, int ldx, int out_f32);
int vc_conv_fwd_stats(const int* desc, int dtype, const void* x, int ldx, const void* w_packed, const float* bias,
                      void* y, int ldy, int relu, float* bn_partial, vc_stream_t stream);
/* Forward layer whose reduction is cut into slices: few output tiles and a long reduction (EncoderCNN.fc1,
 * model_components.py:104 / :131: 36864 -> fc_hidden1 on B*T rows) would otherwise occupy a fraction of the chip.
 * vc_conv_fwd_splitk_plan returns the number of slices (1 = not split: use vc_conv_fwd); vc_conv_fwd_splitk needs
 * workspace [slices][M][co] fp32 and writes y = act(conv(x) + bias) as bf16 (fp32 with out_f32). */
int vc_conv_fwd_splitk_plan(const int* desc, int dtype);
int vc_conv_fwd_splitk(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias, void* y,
                       int relu, int out_f32, float* workspace, vc_stream_t stream);
int vc_conv_dgrad(const int* desc, int dtype, const void* dy, int lddy, const void* w_packed_t, void* dx, int lddx,
                  int out_f32, int beta, vc_stream_t stream);
/* vc_conv_dgrad (bf16, stride 1, dense rows, beta as above) for the LAST contribution to the gradient of a residual
 * block's output out = relu(bn(y) + shortcut) (torchvision Bottleneck.forward; the block input feeds conv1 and the
 * next shortcut): after the accumulation the epilogue multiplies dx by the ReLU decisions in tail_bits ([M][ci/8],
 * from vc_bn_apply_mask), and writes per 64-row slab the sums of dz = dx and of dz * tail_y (tail_y = the BatchNorm's
 * input y, same layout as dx) to bn_partial [ceil(M/64)][2][ci], every entry written.  vc_bn_bwd_sums_from_partial
 * turns them into the BatchNorm backward reductions, so no separate pass reads dx and y for them. */
int vc_conv_dgrad_bn_tail_supported(const int* desc, int dtype);
int vc_conv_dgrad_bn_tail(const int* desc, int dtype, const void* dy, const void* w_packed_t, void* dx, int beta,
                          const void* tail_y, const uint8_t* tail_bits, float* bn_partial, vc_stream_t stream);
/* vc_conv_dgrad (bf16, stride 1, dense rows) whose result is the gradient of a = relu(bn(y)) inside a block (conv ->
 * BatchNorm -> ReLU -> conv: torchvision Bottleneck.forward conv1/bn1/conv2/bn2/conv3, EncoderCNN's conv blocks,
 * model_components.py:86-101) with a consumed by this convolution alone: the epilogue zeroes dx where
 * y * scale + shift <= 0 (stat = [4][G][ci] fp32: mean, invstd, scale, shift of the BatchNorm, one group per
 * rows_per_group consecutive rows, a multiple of 64) and writes per 64-row slab the sums of dz = dx and of dz * y to
 * bn_partial [M/64][2][ci]; vc_bn_bwd_sums_from_partial turns them into the BatchNorm backward reductions. */
int vc_conv_dgrad_bn_pre_supported(const int* desc, int dtype, int rows_per_group);
int vc_conv_dgrad_bn_pre(const int* desc, int dtype, const void* dy, const void* w_packed_t, void* dx, const void* y,
                         const float* stat, int rows_per_group, float* bn_partial, vc_stream_t stream);
/* fc1 of a Swin / ViT MLP (timm Mlp: fc1 -> GELU -> fc2, created at model_components.py:254, 412): y = conv(x) + bias
 * and y_gelu = gelu(y) written by the same epilogue, so the activation is not a pass of its own.  Same geometries
 * as vc_conv_fwd_residual (vc_conv_epilogue_side_supported(desc, dtype, 0)). */
int vc_conv_fwd_gelu_dual(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias, void* y,
                          void* y_gelu, vc_stream_t stream);
/* GEMM epilogues that read a side tensor of the output's own layout (bf16, dense rows, more than 64 output columns;
 * vc_conv_epilogue_side_supported(desc, dtype, mode: 0 forward, 1 data gradient) says whether a geometry qualifies):
 * vc_conv_fwd_residual: y = row_scale * (conv(x) + bias) + residual, out of place -- the x + drop_path(attn(x)) /
 * x + drop_path(mlp(x)) of the Swin blocks (timm SwinTransformerBlock behind model_components.py:254, 412); row_scale
 * (nullable = ones) holds one fp32 factor per rows_per_sample consecutive rows, as vc_drop_path_scales writes them; vc_conv_dgrad_gelu: dx = dgrad * gelu'(pre), the
 * gradient through the MLP's GELU taken in the data-gradient GEMM of mlp.fc2 (pre = mlp.fc1's output). */
int vc_conv_epilogue_side_supported(const int* desc, int dtype, int mode);
int vc_conv_fwd_residual(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias,
                         const void* residual, const float* row_scale, int rows_per_sample, void* y,
                         vc_stream_t stream);
/* y = relu(conv(x) + bias + residual): torchvision Bottleneck's out = relu(bn3(conv3(.)) + identity) in EVAL mode, the
 * BatchNorm folded into the convolution's weights and bias (predict.py / validate_epoch callers behind
 * model_components.py:164-191).  Geometries as vc_conv_fwd_residual. */
int vc_conv_fwd_residual_relu(const int* desc, int dtype, const void* x, const void* w_packed, const float* bias,
                              const void* residual, void* y, vc_stream_t stream);
int vc_conv_dgrad_gelu(const int* desc, int dtype, const void* dy, const void* w_packed_t, const void* pre, void* dx,
                       vc_stream_t stream);
/* dw_packed (fp32 [co][taps][ci]) is accumulated into with atomics: the caller zeroes it.  dbias (nullable, fp32
 * [co]) receives += the column sums of dy, i.e. the bias gradient of the same layer, from the same pass over dy. */
int vc_conv_wgrad(const int* desc, int dtype, const void* x, int ldx, const void* dy, int lddy, float* dw_packed,
                  float* dbias, int force_naive, vc_stream_t stream);
/* cross-check used by the tests: the plain-FMA kernel on bf16 data (mode 0 forward, 1 data gradient) */
int vc_conv_fwd_naive_bf16(const int* desc, int mode, const void* src, const void* w_packed, const float* bias,
                           void* out, int relu, int out_f32, vc_stream_t stream);
int vc_pack_weight(int dtype, const float* w, void* w_fwd, void* w_t, int co, int ci, int taps, int ci_pad, int co_pad,
                   vc_stream_t stream);
/* data-gradient pack [taps*ci_pad][co_pad] for a convolution whose kernel covers its whole input, i.e. a Linear
 * layer over the flattened channels-last map (EncoderCNN.fc1 after x.view(B,-1), model_components.py:131-134) */
int vc_pack_weight_flat_t(int dtype, const float* w, void* w_t, int co, int ci, int taps, int ci_pad, int co_pad,
                          vc_stream_t stream);
/* flags: bit 0 = add to dw (gradient accumulation) instead of overwriting it; bit 1 = zero dw_packed after it has
 * been read, so a caller-owned accumulator is ready for the next vc_conv_wgrad without a fill pass */
int vc_unpack_wgrad(float* dw_packed, float* dw, int co, int ci, int taps, int ci_pad, int flags,
                    vc_stream_t stream);

/* ---- BatchNorm with per-group statistics / LayerNorm ----------------------------------------------
 * Replace nn.BatchNorm2d(momentum=0.01) (model_components.py:92,97,102,107), nn.BatchNorm1d(momentum=0.01)
 * (:167,169,274,276), torchvision's BatchNorm2d, nn.BatchNorm3d (Conv3D :67,73,79,85) and the ReLU /
 * residual add that follow them.  Rows are [G groups][rpg rows][C]; G = T reproduces the reference's
 * per-time-step statistics and its T ordered running-stat updates (encoder loops at :119, :174, :281).
 * sum/sumsq/s1/s2 are [G][C] fp32 accumulated with atomics: the caller zeroes them. */
int vc_bn_stats(int dtype, const void* y, float* sum, float* sumsq, int G, int rpg, int C, vc_stream_t stream);
/* the same sums from the slab partials of vc_conv_fwd_stats (rpg = 64 * spg rows per group) */
int vc_bn_stats_from_partial(const float* partial, float* sum, float* sumsq, int G, int spg, int C,
                             vc_stream_t stream);
/* flags: bit 0 = apply the G ordered momentum updates to running_mean/var; bit 1 = zero sum/sumsq after reading */
int vc_bn_finalize(float* sum, float* sumsq, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float* mean, float* invstd, float* scale, float* shift, int G, int C, int rpg,
                   float eps, float momentum, int flags, vc_stream_t stream);
/* the same, and num_batches_tracked (device int64, nullable) += G in the same launch */
int vc_bn_finalize_count(float* sum, float* sumsq, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* mean, float* invstd, float* scale, float* shift, int G, int C,
                         int rpg, float eps, float momentum, int flags, int64_t* num_batches_tracked,
                         vc_stream_t stream);
int vc_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float* scale, float* shift, int C, float eps, vc_stream_t stream);
int vc_bn_apply(int dtype, const void* y, const float* scale, const float* shift, const void* residual, void* out,
                int G, int rpg, int C, int relu, vc_stream_t stream);
/* backward: with relu set, act (the activation output) supplies the ReLU mask; act == NULL recomputes it as
 * y*scale+shift > 0, valid when no residual was added in front of the ReLU (one activation read less) */
int vc_bn_bwd_reduce(int dtype, const void* y, const void* dout, const void* act, const float* mean,
                     const float* invstd, const float* scale, const float* shift, float* s1, float* s2, int G, int rpg,
                     int C, int relu, vc_stream_t stream);
/* the pair for out = relu(bn(y) + residual) (torchvision Bottleneck's out += identity; relu): vc_bn_apply_mask is
 * vc_bn_apply that also records the ReLU decisions, bits[row][C/8] (bit i of a byte = channel 8*cc + i positive);
 * vc_bn_bwd_reduce_mask takes the mask from those bits and writes dz = dout * mask back OVER dout.  dz is the
 * shortcut's gradient as it stands and the input of vc_bn_bwd_apply(relu = 0, act = NULL, dres = NULL): two
 * passes less over the widest tensors than reading the activation twice and writing a second gradient */
int vc_bn_apply_mask(int dtype, const void* y, const float* scale, const float* shift, const void* residual, void* out,
                     uint8_t* bits, int G, int rpg, int C, int relu, vc_stream_t stream);
int vc_bn_bwd_reduce_mask(int dtype, const void* y, void* dout, const uint8_t* bits, const float* mean,
                          const float* invstd, float* s1, float* s2, int G, int rpg, int C, vc_stream_t stream);
/* s1[g][c] += sum dz, s2[g][c] += sum dz * xhat from the slab sums of vc_conv_dgrad_bn_tail (rpg = 64 * spg) */
int vc_bn_bwd_sums_from_partial(const float* partial, const float* mean, const float* invstd, float* s1, float* s2,
                                int G, int spg, int C, vc_stream_t stream);
int vc_bn_bwd_apply(int dtype, const void* y, const void* dout, const void* act, const float* mean,
                    const float* invstd, const float* scale, const float* shift, const float* s1, const float* s2,
                    void* dy, void* dres, int G, int rpg, int C, int relu, vc_stream_t stream);
/* BatchNorm + ReLU + max-pool in one pass each way: torchvision ResNet's bn1 / relu / maxpool(3, 2, 1) behind
 * model_components.py:164-165, and BatchNorm3d / ReLU / MaxPool3d((1,2,2)) of Conv3D (UCF101_3DCNN_refactored.py:
 * 67-69, 73-75, 79-81, 85-87).  y is [N frames][H][W][C], a BatchNorm group is npg = N/G consecutive frames; the
 * full-resolution activation is never written.  vc_bn_pool_fwd writes the pooled tensor and the arg-max taps (one
 * byte per pooled element); vc_bn_pool_bwd gathers the pooled gradient dyp through them: phase 0 accumulates the
 * BatchNorm backward reductions into s1 / s2 [G][C] (zeroed by the caller), phase 1 writes dy. */
int vc_bn_pool_fwd(int dtype, const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, int N,
                   int H, int W, int C, int k, int s, int p, int npg, vc_stream_t stream);
int vc_bn_pool_bwd(int dtype, int phase, const void* y, const void* dyp, const uint8_t* idx, const float* mean,
                   const float* invstd, const float* scale, const float* shift, float* s1, float* s2, void* dy, int N,
                   int H, int W, int C, int k, int s, int p, int G, vc_stream_t stream);
/* flags: bit 0 = add to dgamma/dbeta instead of overwriting; bit 1 = zero s1/s2 after reading (dgamma and dbeta may
 * both be NULL: the call then only clears) */
int vc_bn_bwd_params(float* s1, float* s2, float* dgamma, float* dbeta, int G, int C, int flags,
                     vc_stream_t stream);
/* nn.LayerNorm inside the Swin-T backbone (timm, created at model_components.py:254, 412) */
int vc_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* mean,
                     float* rstd, int M, int C, float eps, vc_stream_t stream);
int vc_layernorm_bwd(int dtype, const void* x, const void* dout, const float* gamma, const float* mean,
                     const float* rstd, void* dx, float* dgamma, float* dbeta, int M, int C, vc_stream_t stream);
/* the same with dx = gradient through the normalisation + add (nullable; same shape and type as x): a Swin block's
 * input feeds norm1 / norm2 and the residual stream, and autograd would otherwise add the two gradients in a pass
 * of its own */
int vc_layernorm_bwd_add(int dtype, const void* x, const void* dout, const float* gamma, const float* mean,
                         const float* rstd, const void* add, void* dx, float* dgamma, float* dbeta, int M, int C,
                         vc_stream_t stream);

/* ---- layout, pooling, elementwise ------------------------------------------------------------------ */
/* x_3d[:, t] slicing + NCHW->channels-last (model_components.py:127, 176, 285, 460; Conv3D input
 * [B,1,T,H,W] from utils/common/data_loaders.py:56).  time_major: frame = t*B + b. */
int vc_clip_to_channels_last(int dtype, const float* x, void* out, int B, int T, int C, int S, int Cp,
                             int time_major, int chan_outer, vc_stream_t stream);
/* First layers with 1 or 3 input channels (ResNet stem models.resnet*.conv1 via model_components.py:164-165; EncoderCNN
 * conv1 :84-88; Conv3D layer 1, UCF101_3DCNN_refactored.py:62-66; Swin patch_embed.proj): the kw taps along W are folded
 * into the channel axis, out[f][h][wo][j*C + c] = x[.., c, h, wo*sw - pw + j] with Jp = pad8(kw*C) channels, so the
 * layer runs as a (kd, kh, 1) convolution with stride (sd, sh, 1) instead of padding C to 8.  Frame order and source
 * layouts as for vc_clip_to_channels_last.  vc_pack_weight_unfold / vc_unpack_wgrad_unfold convert the torch weight
 * [co][ci][taps_outer][kw] to and from the matching packed form [co_pad][taps_outer][jp] (flags as vc_unpack_wgrad). */
int vc_clip_unfold_w(int dtype, const float* x, void* out, int B, int T, int C, int H, int W, int kw, int sw, int pw,
                     int Jp, int time_major, int chan_outer, vc_stream_t stream);
int vc_pack_weight_unfold(int dtype, const float* w, void* w_fwd, int co, int ci, int taps_outer, int kw, int jp,
                          int co_pad, vc_stream_t stream);
int vc_unpack_wgrad_unfold(float* dw_packed, float* dw, int co, int ci, int taps_outer, int kw, int jp, int flags,
                           vc_stream_t stream);
/* nn.MaxPool2d(3,2,1) of the ResNet stem; nn.MaxPool3d((1,2,2)) (Conv3D :69,75,81,87) as a per-frame 2-D pool */
int vc_maxpool2d_fwd(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int k, int s, int p,
                     vc_stream_t stream);
int vc_maxpool2d_bwd(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int k, int s,
                     int p, vc_stream_t stream);
/* nn.AdaptiveAvgPool2d((1,1)) of ResNet, nn.AdaptiveAvgPool3d((1,1,1)) (Conv3D :90), Swin's token mean */
int vc_avgpool_fwd(int dtype, const void* x, void* y, int N, int S, int C, vc_stream_t stream);
int vc_avgpool_bwd(int dtype, const void* dy, void* dx, int N, int S, int C, vc_stream_t stream);
/* op: 0 a+b, 1 a*(b>0) (F.relu backward), 2 relu(a), 3 gelu(a), 4 a*gelu'(b), 5 sigmoid(a) */
int vc_elementwise(int dtype, int op, const void* a, const void* b, void* out, int64_t n, vc_stream_t stream);
int vc_cast(int src_dtype, int dst_dtype, const void* a, void* out, int64_t n, vc_stream_t stream);
/* bias gradients: out[c] += sum_m a[m][c] */
int vc_colsum(int dtype, const void* a, float* out, int M, int C, int lda, vc_stream_t stream);
/* F.dropout (model_components.py:136,184,228,297,302), nn.Dropout (Conv3D :95,98; TCG_LSTM :488) */
/* seed_dev (nullable): device counter folded into the seed at execution time (new masks on graph replays) */
int vc_dropout_fwd(int dtype, const void* x, void* out, uint8_t* mask, int64_t n, float p, uint64_t seed,
                   const int64_t* seed_dev, vc_stream_t stream);
int vc_dropout_bwd(int dtype, const void* dy, const uint8_t* mask, void* dx, int64_t n, float p, vc_stream_t stream);

/* timm's DropPath (stochastic depth) inside the Swin blocks; backward applies the same call to the gradient */
int vc_drop_path(int dtype, const void* x, void* out, int64_t n, int64_t elems_per_sample, float p, uint64_t seed,
                 const int64_t* seed_dev, vc_stream_t stream);
/* the same per-sample factors (0 or 1 / (1 - p)) as a vector scales[samples], for vc_conv_fwd_residual's row_scale */
int vc_drop_path_scales(float* scales, int samples, float p, uint64_t seed, const int64_t* seed_dev,
                        vc_stream_t stream);

/* ---- Channel + spatial attention gate of the image / density fusion CNN
 * (models/swintransformer_RNN/image_density_fusion.py:14-87: ChannelAttention, SpatialAttention and their use in
 * LocalFeatureExtraction.forward :80-86).  x, out, dout, dx: channels-last [N][H*W][C] in `dtype`;
 * W1 [C/r][C], W2 [C][C/r] (ChannelAttention.fc.0 / fc.2), Ws [1][2][k][k] (SpatialAttention.conv), all fp32.
 * out = relu(y * sigmoid(conv(mean_c y, max_c y))), y = x * sigmoid(mlp(avg_p x) + mlp(max_p x)).  The forward call
 * writes what backward needs into caller-allocated fp32 / int32 buffers: avg, mx, gc [N][C]; argp [N][C]; hid
 * [N][2][C/r]; pooled [N][H*W][2]; argc, gs [N][H*W]; workspace: vc_cbam_workspace_floats() floats.  Backward
 * writes dx and ACCUMULATES dW1, dW2, dWs; scratch dpre [N][H*W], dpooled [N][H*W][2], dgc, davg, dmx [N][C]. */
int64_t vc_cbam_workspace_floats(int N, int H, int W, int C);
int vc_cbam_fwd(int dtype, const void* x, const float* W1, const float* W2, const float* Ws, void* out, float* avg,
                float* mx, int* argp, float* hid, float* gc, float* pooled, int* argc, float* gs, float* workspace,
                int N, int H, int W, int C, int Ch, int k, vc_stream_t stream);
int vc_cbam_bwd(int dtype, const void* x, const void* dout, const float* W1, const float* W2, const float* Ws,
                const float* avg, const float* mx, const int* argp, const float* hid, const float* gc,
                const float* pooled, const int* argc, const float* gs, void* dx, float* dW1, float* dW2, float* dWs,
                float* dpre, float* dpooled, float* dgc, float* davg, float* dmx, int N, int H, int W, int C, int Ch,
                int k, vc_stream_t stream);

/* ---- Swin-T window attention / patch merging (timm swin_tiny_patch4_window7_224 behind
 * model_components.py:254, 412).  qkv [B*H*W][3C] in token order; window partition, cyclic shift, relative
 * position bias gather, shift mask, softmax and P.V are one kernel; head_dim is 32, window <= 7. */
int vc_window_attn_fwd(int dtype, const void* qkv, const float* bias_table, void* out, int B, int H, int W, int C,
                       int heads, int ws, int shift, float scale, vc_stream_t stream);
int vc_window_attn_bwd(int dtype, const void* qkv, const void* dout, const float* bias_table, void* dqkv,
                       float* dbias_table, int B, int H, int W, int C, int heads, int ws, int shift, float scale,
                       vc_stream_t stream);
int vc_patch_merge(int dtype, const void* in, void* out, int B, int H, int W, int C, int inverse, vc_stream_t stream);

/* Persistent recurrence of one nn.LSTM layer (DecoderRNN.LSTM, model_components.py:209-214, 221; TCG_LSTM's
 * lstm_short / lstm_long, :482-483): all T steps, ndir = 1 or 2 directions (direction 1 walks the sequence in
 * reverse), in ONE launch with W_hh sharded over workgroups and resident in registers.  Arrays have one entry per
 * direction: gates_x [T][B][4H] fp32 = x W_ih^T + b_ih + b_hh, w_hh [4H][H] bf16 (gate order i,f,g,o), hs receives
 * h_t at hs[(t*B + b)*ldh + j], act [T][B][4H] / c [T][B][H] fp32 are what vc_lstm_cell_bwd reads, xchg is scratch
 * of vc_lstm_persist_xchg_bytes(B, H) bytes (zeroed by the call).  *err (nullable) becomes 1 if a workgroup gave
 * up waiting for its peers.  vc_lstm_persist_supported() = 1 when (dtype, B, H, ndir) can take this path. */
long long vc_lstm_persist_xchg_bytes(int B, int H);
int vc_lstm_persist_supported(int dtype, int B, int H, int ndir);
int vc_lstm_persist_fwd(int dtype, const float* const* gates_x, const void* const* w_hh, void* const* hs, int ldh,
                        float* const* act, float* const* c, void* const* xchg, int* err, int T, int B, int H,
                        int ndir, vc_stream_t stream);
/* No counterpart in the reference.  vc_lstm_persist_fwd for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): same arguments without xchg; the exchange
 * buffer is a 512 KiB array inside the library's code object (one copy per GPU, held until the process ends), and the
 * call allocates and frees nothing.  Refused (VC_ERR_SHAPE) when ndir * ceil(B/32) * H * 256 bytes exceed 512 KiB
 * (both directions of B <= 64 at H = 512, or of B <= 32 at H = 1024, fit).
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU race; calls on one stream are
 * ordered, and each clears what the one before left. */
int vc_lstm_persist_fwd_noalloc(int dtype, const float* const* gates_x, const void* const* w_hh, void* const* hs,
                                int ldh, float* const* act, float* const* c, int* err, int T, int B, int H, int ndir,
                                vc_stream_t stream);
/* Backward recurrence (BPTT) of the same layer in one persistent launch: per direction dh_out (bf16, gradient of the
 * layer output at dh_out[(t*B + b)*ldo + j]), act / c as saved by the forward pass, w_hh_t = W_hh^T [H][4H] bf16 held in
 * registers, dgates [T][B][4H] bf16 receives the gate pre-activation gradients of every step (what the sequence-wide
 * weight- and input-gradient GEMMs read); xchg = scratch of vc_lstm_persist_bwd_xchg_bytes(B, H) bytes (zeroed by the
 * call).  Supported for H a power of two in 32..512 (vc_lstm_persist_bwd_supported). */
long long vc_lstm_persist_bwd_xchg_bytes(int B, int H);
int vc_lstm_persist_bwd_supported(int dtype, int B, int H, int ndir);
int vc_lstm_persist_bwd(int dtype, const void* const* dh_out, int ldo, const float* const* act, const float* const* c,
                        const void* const* w_hh_t, void* const* dgates, void* const* xchg, int* err, int T, int B,
                        int H, int ndir, vc_stream_t stream);
/* ---- LSTM cell (gate order i,f,g,o; nn.LSTM at model_components.py:209-214, 482-483) -------------- */
int vc_lstm_cell_fwd(int dtype, const float* gates_x, int ldx, const float* gates_h, const float* c_prev, void* h_out,
                     int ldh, float* c_out, float* act_out, int B, int H, vc_stream_t stream);
int vc_lstm_cell_bwd(int dtype, const void* dh_out, int ldo, const float* dh_rec, const float* dc_next,
                     const float* act, const float* c, const float* c_prev, void* dgates, float* dc_prev, int B, int H,
                     vc_stream_t stream);

/* ---- TCG-LSTM fusion (model_components.py:494-514) ------------------------------------------------- */
int vc_tcg_inputs(int dtype, const float* g, const float* c, void* short_in, void* long_in, int B, int T, int Tl,
                  int K, int E, int down, vc_stream_t stream);
int vc_tcg_inputs_bwd(int dtype, const void* dshort, const void* dlong, float* dg, float* dc, int B, int T, int Tl,
                      int K, int E, int down, vc_stream_t stream);
int vc_tcg_prep(int dtype, const void* hs, const void* hl, void* cat, int T, int Tl, int B, int H, vc_stream_t stream);
int vc_tcg_fuse_fwd(int dtype, const float* z, const void* cat, float* gate, float* h, float* alpha, void* clip,
                    int T, int B, int H, vc_stream_t stream);
int vc_tcg_fuse_bwd(int dtype, const void* dclip, const void* cat, const float* gate, const float* h,
                    const float* alpha, void* dz, float* dhs, float* dup2, int T, int B, int H, vc_stream_t stream);
int vc_tcg_combine_bwd(int dtype, const void* dcat, const float* dhs, const float* dup2, void* dhs_out, void* dhl,
                       int T, int Tl, int B, int H, vc_stream_t stream);

/* ---- loss and optimizer ----------------------------------------------------------------------------
 * nn.CrossEntropyLoss + argmax + eq().sum() (train.py:554; utils/common/training_utils.py:85, 90-91);
 * loss[0] and correct[0] are accumulated into: the caller zeroes them. */
int vc_ce_fwd_bwd(int grad_dtype, const float* logits, int ld, const int64_t* target, float* loss, int64_t* pred,
                  int* correct, void* dlogits, int ldd, int B, int C, float loss_scale, float grad_scale,
                  vc_stream_t stream);
/* optim.Adam (L2-coupled) / AdamW / SGD(momentum) (training_utils.py:324-329) over one flat fp32 buffer.
 * mode 0 Adam, 1 AdamW, 2 SGD (b1 = momentum).  grad_scale folds the 1/world of the all-reduce. */
int vc_optim_step(int mode, float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                  float eps, float wd, int step, float grad_scale, const float* lr_dev, const int64_t* step_dev,
                  vc_stream_t stream);
/* *counter += inc on the stream: the device-resident step counter that lr_dev/step_dev/seed_dev users read, so
 * that a hipGraph of the whole training step replays with current values */
int vc_counter_add(int64_t* counter, int64_t inc, vc_stream_t stream);

/* ---- per-frame input transform (the reference's host-side PIL + torchvision chain, per frame:
 * utils/common/data_loaders.py:39-51, 88-100 with the transform of train.py:344-388) ---------------------------
 * vc_resample_ksize / vc_resample_coeffs are HOST functions: window bounds [out_size][2] (first source index, tap
 * count) and 22-bit fixed-point weights [out_size][ksize] of Pillow's anti-aliased bilinear filter for a resize
 * of in_size to out_size samples; vc_resample_coeffs returns ksize.  The caller copies both tables to the device.
 * vc_frames_transform: frames uint8 [n][hs][ws][3] (RGB, 16-byte aligned base) -> out [n][C][hd][wd] of out_dtype,
 * C = 1 if gray (convert('L') before the resize, Dataset_3DCNN) else 3: Resize([hd, wd]) -> flip of the frames
 * with flip[i] != 0 (nullable; one flag per frame, the reference draws RandomHorizontalFlip per frame) -> ToTensor
 * -> Normalize(mean[C], std[C]).  Integer-exact with Pillow up to the uint8 frame, fp32 output bit-identical to
 * the host chain.  RandomCrop(size) after Resize to the same size is the identity; ColorJitter: vc_frames_jitter. */
int vc_resample_ksize(int in_size, int out_size);
int vc_resample_coeffs(int in_size, int out_size, int* bounds, int* kk);
int vc_frames_transform(const void* frames, int n, int hs, int ws, int gray, int hd, int wd, const int* xbounds,
                        const int* xk, int ksx, const int* ybounds, const int* yk, int ksy, const void* flip,
                        const float* mean, const float* stdv, void* out, int out_dtype, vc_stream_t stream);
/* ColorJitter(brightness, contrast, saturation, hue) of train.py:377-378 between the flip and ToTensor.
 * vc_frames_transform with out_dtype 2 stops after the flip and writes the uint8 frame, planar [n][C][hd][wd]
 * (mean / stdv unused).  vc_frames_jitter then applies to every frame i of planes_u8 [n][3][h][w] (overwritten)
 * the operations order[4 i .. 4 i + 3] (0 brightness, 1 contrast, 2 saturation, 3 hue, >= 4 nothing; torchvision
 * draws a permutation per call, a second contrast entry is ignored) with the factors factors[3 i .. 3 i + 2]
 * (brightness, contrast, saturation) and hue_shift[i] = (uint8)(hue_factor * 255), followed by ToTensor and
 * Normalize into out [n][3][h][w] of out_dtype 0 / 1.  Byte-identical to torchvision's PIL path (ImageEnhance
 * blends in float32, Pillow's RGB <-> HSV conversions). */
int vc_frames_jitter(void* planes_u8, int n, int h, int w, const void* order, const float* factors,
                     const void* hue_shift, const float* mean, const float* stdv, void* out, int out_dtype,
                     vc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VC_HIP_H */
