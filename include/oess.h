/*
 * liboess -- C-ABI of the MI355X-native OpenESS hot path (gfx950, hand-written HIP).
 *
 * The reference (ldkong1205/OpenESS) is pure Python: it has no FFI.  Its "plugin surface" is a
 * set of Python call contracts; each entry point below is what a ctypes binding
 * inside the cited reference function would call instead of the NumPy / PyTorch-CPU body.
 * INTEGRATION.md shows the stub for each.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host.
 *   - stream-ordered: the call only ENQUEUES work on `stream` (a hipStream_t passed as void*);
 *     the caller synchronises.  No hidden allocation: outputs and workspaces are caller-owned and
 *     must stay alive until the stream has passed the call; sizes come from *_workspace_bytes().
 *   - returns 0 on success, a negative OESS_E* code otherwise; no exceptions cross the boundary.
 *   - no global mutable state; thread-safe for distinct streams/buffers.
 */
#ifndef OESS_H
#define OESS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OESS_OK 0
#define OESS_EINVAL (-22)   /* bad argument (shape, null pointer, unsupported size)            */
#define OESS_ENOMEM (-12)   /* workspace too small                                             */
#define OESS_ELAUNCH (-5)   /* hipGetLastError() != hipSuccess after a launch                  */

typedef void* oess_stream_t; /* hipStream_t */

/* Library / device identification.  OESS_ABI_VERSION is bumped whenever a signature of this header changes or an entry point
 * is removed; oess_abi_version() returns the value the library was built with and the ctypes binding (openess_amd/_lib.py,
 * ABI_VERSION) refuses a library whose value differs. */
#define OESS_ABI_VERSION 10
int oess_abi_version(void);
const char* oess_build_info(void);           /* "liboess <ver> gfx950 hipcc <ver>" */
const char* oess_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * K1  Tri-linear event voxelizer.
 * Replaces VoxelGrid.convert (DSEC/dataset/representations.py:15-54), called once per segment
 * (sub-window) from Sequence.generate_event_tensor (DSEC/dataset/sequence_ov.py:212-223).
 *
 * x,y,p,t: float32 SoA exactly as passed to VoxelGrid.convert (t already normalised to [0,1] by
 * the caller, sequence_ov.py:155-156).  Segment s owns events [seg_offsets[s], seg_offsets[s+1])
 * and writes output channels [s*C, (s+1)*C).  out is (n_seg*C) x (H-crop_rows) x W float32,
 * every element written exactly once (no pre-zeroing needed).  crop_rows fuses the
 * `event_tensor[:, :-40, :]` crop (sequence_ov.py:307).  count_mode != 0 replaces every weight by
 * 1.0 (integer hit histogram; used by the bit-exact index tests).
 * ------------------------------------------------------------------------------------------ */
/* Workspace for any voxelizer call with at most n_events events in n_seg segments of at most
 * max_seg_len events each (C = accumulated channels: bins for tri-linear, 2*bins for nearest). */
size_t oess_voxelize_workspace_bytes(int64_t n_events, int n_seg, int64_t max_seg_len, int C, int H, int W,
                                     int crop_rows);

int oess_voxelize_trilinear_f32(const float* x, const float* y, const float* p, const float* t,
                                const int64_t* seg_offsets, int n_seg, int64_t max_seg_len,
                                int C, int H, int W, int crop_rows, int count_mode,
                                float* out, void* workspace, size_t workspace_bytes, oess_stream_t stream);

/* Same, straight from raw DSEC event columns (events.h5 `events/{x,y,t,p}`, eventslicer.py:32-98):
 * fuses rectify_events (sequence_ov.py:204-210: rectify_map[y, x] -> (x', y'), map is H x W x 2
 * float32; seg_map[s] selects one of n_maps maps, one per sequence), the float64->float32 time
 * normalisation of events_to_voxel_grid (sequence_ov.py:154-157) and K1. */
int oess_voxelize_dsec_raw(const uint16_t* x, const uint16_t* y, const int64_t* t_us, const uint8_t* p,
                           const float* rectify_maps, const int32_t* seg_map, int n_maps,
                           const int64_t* seg_offsets, int n_seg, int64_t max_seg_len,
                           int C, int H, int W, int crop_rows, int count_mode,
                           float* out, void* workspace, size_t workspace_bytes, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K1' Nearest-xy / linear-t voxelizer.
 * Replaces generate_voxel_grid (datasets/data_util.py:51-117) called per chunk from
 * DDD17Events.__getitem__ (datasets/ddd17_events_loader.py:161-173).
 * events: [N x 4] rows (x, y, t, p), int64 (DDD17 memmap loader) or float64.
 * Output channels per segment: separate_pol ? 2*nbins (pos then neg) : nbins (pos - neg).
 * out is (n_seg*ch) x (H-crop_rows) x W float32.
 * ------------------------------------------------------------------------------------------ */
int oess_voxelize_nearest_i64(const int64_t* events, const int64_t* seg_offsets, int n_seg, int64_t max_seg_len,
                              int nbins, int H, int W, int crop_rows, int separate_pol, int count_mode,
                              float* out, void* workspace, size_t workspace_bytes, oess_stream_t stream);
int oess_voxelize_nearest_f64(const double* events, const int64_t* seg_offsets, int n_seg, int64_t max_seg_len,
                              int nbins, int H, int W, int crop_rows, int separate_pol, int count_mode,
                              float* out, void* workspace, size_t workspace_bytes, oess_stream_t stream);

/* generate_event_histogram (datasets/data_util.py:17-35): 2 x H x W [neg, pos] counts per segment.
 * (The reference does no bounds check; out-of-range events are DROPPED here instead of raising.) */
int oess_event_histogram_i64(const int64_t* events, const int64_t* seg_offsets, int n_seg, int64_t max_seg_len,
                             int H, int W, float* out, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K2  Masked (non-zero) normalisation of a whole tensor.
 * Replaces EventPreprocessor.__call__ normalisation (e2vid/utils/inference_utils.py:78-85) and
 * normalize_voxel_grid (datasets/data_util.py:38-48).  in/out: n float32 (may alias).
 * stats: oess_masked_stats_doubles(1) doubles of caller-owned device scratch: the totals {sum, sumsq, nnz, -} first, then one
 * row of partial sums per workgroup, added in a fixed order by a finalize launch (deterministic: no floating-point atomics).
 * No host sync: the "if num_nonzeros > 0" test happens on the device.
 * ------------------------------------------------------------------------------------------ */
size_t oess_masked_stats_doubles(int n_slices);
int oess_masked_normalize_f32(const float* in, float* out, int64_t n, double* stats, oess_stream_t stream);
/* Strided variant for a channel slice [B, c0:c0+Cs, H, W] of a [B, Ctot, H, W] tensor (the 5-bin
 * sub-window slice event[:, 5i:5i+5], training/pretrain_trainer.py:437-440). out is dense B x Cs x HW. */
int oess_masked_normalize_slice_f32(const float* in, float* out, int B, int Ctot, int c0, int Cs, int64_t HW,
                                    double* stats, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K7  Superpixel scatter-mean (segment mean) and its backward.
 * Replaces the inline sparse one-hot matmul of training/pretrain_trainer.py:445-465.
 * feat: pixel-major [P x Cf] (NHWC flattened, P = B*H*W), float32 or bf16 (is_bf16).
 * ids: [P] int64 raw superpixel ids; the per-sample offset b*superpixel_size is added inside
 * (pixels_per_sample = H*W).  k: [S x Cf] float32, count: [S] float32, both fully written.
 * Pixels whose offset id is outside [0, S) are an error in the reference (S = max id + 1) and are
 * ignored here.
 * Deterministic (bit-repeatable): partial sums meet as 64-bit fixed-point integers (2^-32 units; 96-bit global
 * accumulators in `workspace`, oess_segment_mean_fwd_workspace_bytes(S, Cf) bytes, 16-byte aligned).  Range
 * contract: finite features with |x| < 32768; anything else makes the WHOLE of k NaN.
 * ------------------------------------------------------------------------------------------ */
size_t oess_segment_mean_fwd_workspace_bytes(int S, int Cf);
int oess_segment_mean_fwd(const void* feat, int is_bf16, const int64_t* ids, int64_t P, int64_t pixels_per_sample,
                          int superpixel_size, int Cf, int S, float* k, float* count, void* workspace,
                          size_t workspace_bytes, oess_stream_t stream);
/* grad_feat[p, :] = grad_k[id(p), :] / (count[id(p)] + 1e-6)  (float32 or bf16 output).  workspace (nullable): S * Cf *
 * sizeof(output element) bytes, 16-byte aligned: the S x Cf quotients are then formed once and the per-pixel pass is a row gather. */
int oess_segment_mean_bwd(const float* grad_k, const float* count, const int64_t* ids, int64_t P,
                          int64_t pixels_per_sample, int superpixel_size, int Cf, int S,
                          void* grad_feat, int is_bf16, void* workspace, size_t workspace_bytes, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K9  TaskLoss = DiceLoss + CrossEntropyLoss(ignore_index) forward and backward.
 * Replaces utils/loss_functions.py:17-24 (TaskLoss), :114-135 (DiceLoss), :80-90 (BinaryDiceLoss).
 * logits element (pixel p, class c) lives at logits[p*stride_p + c*stride_c] within a sample of
 * `pixels_per_sample` pixels whose base is b*stride_b (covers NCHW: stride_p=1, stride_c=HW,
 * stride_b=K*HW; and NHWC: stride_p=K, stride_c=1, stride_b=HW*K).  float32 or bf16.
 * target: [P] int64.  sums: oess_task_loss_sums_doubles(K) doubles of scratch: {inter[K], psq[K], ysum[K], ce_sum, n_valid}
 * totals (written by the fwd call, consumed by the bwd call) followed by one row of partial sums per forward workgroup, which
 * the fwd call adds in row order (no atomics: bit-repeatable).  loss_out: 3 floats {total, dice, ce}.
 * flags bit0 = dice, bit1 = cross-entropy.
 * ------------------------------------------------------------------------------------------ */
size_t oess_task_loss_sums_doubles(int K);
int oess_task_loss_fwd(const void* logits, int is_bf16, const int64_t* target, int64_t P, int64_t pixels_per_sample,
                       int64_t stride_b, int64_t stride_p, int64_t stride_c, int K, int ignore_index, int flags,
                       double* sums, float* loss_out, oess_stream_t stream);
int oess_task_loss_bwd(const void* logits, int is_bf16, const int64_t* target, int64_t P, int64_t pixels_per_sample,
                       int64_t stride_b, int64_t stride_p, int64_t stride_c, int K, int ignore_index, int flags,
                       const double* sums, float grad_scale, const float* grad_scale_dev /* nullable: multiplies grad_scale */,
                       void* grad_logits, int grad_is_bf16, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a16 Consistency losses of the joint OpenESS stage (training/openess_trainer.py:497-503):
 *   oess_l1_mean_*      cons_feat_loss = nn.L1Loss()(feat_a, feat_b)  (mean |a - b| over n elements, same memory order)
 *   oess_cosine_mean_*  cons_pred_loss = mean(1 - F.cosine_similarity(logits_a, logits_b, dim=1)) over P pixels x C
 *                       channels, NHWC with pixel strides in elements, eps = 1e-8 per-norm clamp (torch 2.x formula).
 * Operands bf16 (is_bf16 != 0) or fp32; gradients are written in the operand dtype (either may be null).
 * partials: scratch of oess_loss_partials_bytes() bytes (deterministic two-stage sum); loss / grad_out: device scalars.
 * ------------------------------------------------------------------------------------------ */
size_t oess_loss_partials_bytes(void);
int oess_l1_mean_fwd(const void* a, const void* b, int64_t n, int is_bf16, void* partials, float* loss, oess_stream_t stream);
int oess_l1_mean_bwd(const void* a, const void* b, int64_t n, int is_bf16, const float* grad_out, void* grad_a, void* grad_b,
                     oess_stream_t stream);
int oess_cosine_mean_fwd(const void* a, long long a_pix_stride, const void* b, long long b_pix_stride, int64_t P, int C,
                         int is_bf16, float eps, void* partials, float* loss, oess_stream_t stream);
int oess_cosine_mean_bwd(const void* a, long long a_pix_stride, const void* b, long long b_pix_stride, int64_t P, int C,
                         int is_bf16, float eps, const float* grad_out, void* grad_a, long long ga_pix_stride, void* grad_b,
                         long long gb_pix_stride, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a14 PointInfoNCE, NCELoss.forward (utils/loss_functions.py:147-154): loss = CrossEntropy(k q^T / T, arange(S)).
 * k, q: fp32 [S x C] (superpixel means).  The forward leaves dL/d(k q^T) = (softmax - I) / (S T) in grad_logits
 * [S x S] fp32, which the backward multiplies out: grad_k = g * G q, grad_q = g * G^T k (either may be null).
 * partials: scratch of >= S doubles; loss / grad_out: device scalars.
 * ------------------------------------------------------------------------------------------ */
int oess_nce_loss_fwd(const float* k, const float* q, int S, int C, float temperature, float* grad_logits, void* partials,
                      size_t partials_bytes, float* loss, oess_stream_t stream);
int oess_nce_loss_bwd(const float* grad_logits, const float* k, const float* q, int S, int C, const float* grad_out, float* grad_k,
                      float* grad_q, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * a18 AdamW step for a whole parameter list in one launch (torch.optim.AdamW as built in
 * training/pretrain_trainer.py:231-243: decoupled weight decay, amsgrad off), ATen's operation order.
 * table: device int64 [n_tensors][5] = {param, grad, exp_avg, exp_avg_sq pointers (fp32), numel};
 * chunk_map: device int32 [n_chunks][2] = {tensor index, chunk index}, chunks of chunk_elems elements.
 * bias_correction1 = 1 - beta1^step, bias_correction2_sqrt = sqrt(1 - beta2^step); all scalars are doubles (the Python
 * optimiser's own values) and are rounded to fp32 once, so 1 - beta2 etc. match the library bit for bit.
 * ------------------------------------------------------------------------------------------ */
int oess_adamw_multi_f32(const int64_t* table, int n_tensors, const int32_t* chunk_map, int n_chunks, int chunk_elems, double lr,
                         double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                         double bias_correction2_sqrt, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K11 Confusion matrix (evaluation/metrics.py:4-23): conf[gt*K + pred] += 1 over gt != ignore.
 * conf: K*K int64, ACCUMULATED into (caller zeroes once per validation epoch).
 * ------------------------------------------------------------------------------------------ */
int oess_confusion_accumulate(const int64_t* pred, const int64_t* label, int64_t n, int K, int ignore_label,
                              int64_t* conf, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MFMA implicit-GEMM convolution (bf16 in, fp32 accumulate), NHWC with explicit pixel strides.
 * Replaces the ATen/cuDNN conv2d calls behind nn.Conv2d in e2vid/model/submodules.py:7-31,175-214
 * (ConvLayer, ConvLSTM.Gates), models/_resnet.py:74-114 (Bottleneck), models/deeplabv3.py:295-348
 * (ASPP) and models/style_networks.py:252-289 (ReLUINSConv2d / INSResBlock).
 *
 * oess_conv2d_pack_weight: OIHW fp32 (Conv2d.weight) -> packed bf16 [Npad][Kpad], K = (r, s, ci);
 *   flip_for_dgrad == 1 packs the data-gradient operator (taps rotated 180 deg, in/out swapped) so
 *   that dX = conv(dY, packed, pad = dil*(R-1) - pad) for stride-1 convolutions; == 2 packs the forward operator
 *   with ConvLSTM gate-interleaved rows for oess_convlstm_fused_bf16.
 * oess_conv2d_fwd_bf16: out = act(conv(in, w) + bias [+ residual]); Cin must be a multiple of 8
 *   (pad the channel dimension; padded weights are zero).  Exactly one of out_bf16 / out_f32 is used
 *   (out_f32 wins when non-null).  Pixel strides are in ELEMENTS.  relu: 0 = none, 1 = ReLU, 2 = GELU (erf form).
 * ------------------------------------------------------------------------------------------ */
size_t oess_conv2d_packed_bytes(int Cout, int Cin, int R, int S, int flip_for_dgrad);
int oess_conv2d_pack_weight(const float* w_oihw, int Cout, int Cin, int R, int S, int flip_for_dgrad, void* packed,
                            size_t packed_bytes, oess_stream_t stream);
/* The same packing for n weights in ONE launch (all trainable convolutions of a model after an optimiser step); Cout % 8 == 0
 * and Cin % 8 == 0.  table_dev: n rows of eight 64-bit words in DEVICE memory, the last one the problem's first workgroup
 * (problem p owns oess_conv2d_pack_multi_blocks(...) workgroups; total_blocks = their sum).
 *   flip_from_packed == 0: rows {w_oihw fp32, packed forward operand, Cout, Cin, R, S, 0, first block};
 *   flip_from_packed == 1: rows {packed forward operand, packed data-gradient operand, Cout, Cin, R, S, 0, first block}: the
 *                          flip_for_dgrad = 1 operand formed from the (already current) forward operand by tile transposes.
 * Only the valid region of a destination is written: it must have been packed once by oess_conv2d_pack_weight (zero padding).
 * The caller uploads the table (and may cache it while the pointers repeat). */
long long oess_conv2d_pack_multi_blocks(int Cout, int Cin, int R, int S, int flip_from_packed);
int oess_conv2d_pack_weight_multi(const long long* table_dev, int n, long long total_blocks, int flip_from_packed,
                                  oess_stream_t stream);
int oess_conv2d_fwd_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int Cin, const void* w_packed,
                         const float* bias, int Cout, int R, int S, int stride, int pad, int dil, int relu,
                         const void* residual, long long res_pix_stride, void* out_bf16, float* out_f32,
                         long long out_pix_stride, float* tile_stats, void* workspace, size_t workspace_bytes,
                         oess_stream_t stream);
/* workspace (nullable): oess_conv2d_fwd_workspace_bytes(...) bytes of scratch for the split-K form that small-M / long-K layers
 * take (ASPP dilated 3x3 at output stride 16, models/deeplabv3.py:295-348: 140 workgroups x 288 K-slabs otherwise); the query
 * returns 0 for layers that run in one pass.  Without a workspace every layer runs in one pass (same result up to the
 * summation order of the fp32 accumulators). */
size_t oess_conv2d_fwd_workspace_bytes(int B, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int dil,
                                       int with_tile_stats, int out_is_f32);
/* tile_stats (nullable): [ceil(M/128)][2][Cout] fp32, per-128-row-tile column sums and sums of squares of the fp32
 * result (BatchNorm batch statistics straight from the accumulators; bias-free, no activation/residual).
 * oess_norm_reduce_finalize_tile_stats turns them into mean / rstd / scale / shift. */

/* Reduce + finalize (G = 1), sums and E[x^2] - E[x]^2 in double, slices added in a FIXED order (bit-repeatable):
 * the tile partials of a conv epilogue -> mean / rstd / scale / shift (+ BatchNorm running statistics;
 * models/image_model.py:113-114 leaves the frozen teacher in .train()).  One launch for <= 64 tiles, otherwise a slice-sum launch
 * and a finalize launch (the kernel boundary is the synchronisation).  `scratch`: caller-owned double[32 * 2 * C] (any
 * content), `counters`: caller-owned uint32[(C + 31) / 32], ZERO on entry and left zero on return (only touched by the opt-in
 * one-launch ticket protocol, environment OESS_BN_ONE_LAUNCH=1, kept for A/B measurements). */
int oess_norm_reduce_finalize_tile_stats(const float* tile_stats, int tiles, int C, double* scratch,
                                         unsigned int* counters, float count, float eps, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float momentum, float* mean, float* rstd,
                                         float* scale, float* shift, oess_stream_t stream);

/* Small maps (tiles <= 512, C % 64 == 0): the reduce / finalize above AND oess_norm_apply_nhwc_bf16 in one launch, no
 * cross-workgroup protocol: every workgroup re-reduces the tile partials of its 64 channels (fixed order, double) and
 * applies out = act(x * scale + shift [+ residual]) to its pixel chunk; x may alias out.  mean / rstd nullable. */
int oess_norm_tile_stats_apply_nhwc_bf16(const float* tile_stats, int tiles, int C, float count, float eps, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float momentum, float* mean,
                                         float* rstd, const void* x, long long x_pix_stride, const void* residual,
                                         long long res_pix_stride, int relu, long long pixels, void* out, long long out_pix_stride,
                                         oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ConvLSTM gate fusion.  Replaces the chunk/sigmoid/tanh/mul/add tail of ConvLSTM.forward
 * (e2vid/model/submodules.py:205-212).  gates: [P x 4C] bf16 (in, remember, out, cell blocks);
 * cell state fp32 [P x C] (prev_cell may be null = zero state; cell may alias prev_cell);
 * hidden bf16 with its own pixel stride (a channel slice of the cat(x, h) buffer).
 * ------------------------------------------------------------------------------------------ */
int oess_convlstm_gates_bf16(const void* gates, long long gates_pix_stride, const float* prev_cell, float* cell,
                             void* hidden, long long hidden_pix_stride, long long n_pixels, int C, oess_stream_t stream);

/* E2VID head + first encoder conv in ONE kernel (e2vid/model/unet.py:137-146, submodules.py ConvLayer: head = 5x5 stride 1,
 * bins -> 32 channels; encoder 0's conv = 5x5 stride 2, 32 -> 64): out = act_e(conv5x5s2(act_h(conv5x5(x8) + head_bias)) + enc_bias).
 * The 32-channel head output never reaches memory: every 8 x 16-pixel output patch computes the 19 x 35 head pixels under it
 * into LDS.  x8: NHWC bf16 with 8 channels (bins zero-padded); head_w_packed = oess_conv2d_pack_weight of the [32, 8, 5, 5] weight,
 * enc_w_packed of the [64, 32, 5, 5] weight; *_relu in {0, 1}; out: NHWC bf16 [B, (H-1)/2+1, (W-1)/2+1, 64] view, pixel stride
 * out_pix_stride (a channel slice of the ConvLSTM's cat(x, h) buffer), 16-byte aligned.  Same result as the two
 * oess_conv2d_fwd_bf16 calls (the head output rounded to bf16 in between, as there). */
/* The same kernel fed from the fp32 event tensor [B, Ctot, H, W]: the EventPreprocessor apply of the slice events[:, c0:c0+Cs]
 * (e2vid/utils/inference_utils.py:80-85; stats = {sum, sumsq, nnz} from oess_masked_stats_slice(s)_f32, normalize = 0 skips it)
 * and the zero-padded 8-channel bf16 packing happen per patch inside the kernel: the NHWC8 tensor of
 * oess_event_slice_to_nhwc8_bf16 is never written either.  Cs <= 5.  Same result as that call followed by
 * oess_e2vid_head_enc0_bf16. */
int oess_e2vid_events_head_enc0_bf16(const float* events, int B, int Ctot, int c0, int Cs, int H, int W, const double* stats,
                                     int normalize, const void* head_w_packed, const float* head_bias, int head_relu,
                                     const void* enc_w_packed, const float* enc_bias, int enc_relu, void* out,
                                     long long out_pix_stride, oess_stream_t stream);
int oess_e2vid_head_enc0_bf16(const void* x8, long long x8_pix_stride, int B, int H, int W, const void* head_w_packed,
                              const float* head_bias, int head_relu, const void* enc_w_packed, const float* enc_bias, int enc_relu,
                              void* out, long long out_pix_stride, oess_stream_t stream);

/* ConvLSTM step in ONE kernel: Gates convolution (submodules.py:202-203) + the cell update above, fused in the MFMA
 * epilogue.  w_packed_gates comes from oess_conv2d_pack_weight(..., flip_for_dgrad = 2): the 4C Conv2d rows are packed
 * gate-interleaved (row 4*hc + gate) so that one lane of the transposed accumulator owns the four gates of a hidden
 * channel; bias stays in Conv2d order [4C].  in = cat(x, h_prev) NHWC bf16 (Cin = Cx + C); hidden (bf16, own pixel
 * stride) must NOT overlap `in` (neighbouring tiles still read h_prev: ping-pong two cat buffers); cell may alias
 * prev_cell (null = zero state).  C_hidden % 32 == 0, stride 1, dilation 1.  The 4C-channel gate tensor is never
 * written (-512 B/pixel of HBM traffic per step at C = 64 versus conv + oess_convlstm_gates_bf16). */
int oess_convlstm_fused_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int Cin,
                             const void* w_packed_gates, const float* bias, int C_hidden, int R, int S, int pad,
                             const float* prev_cell, float* cell, void* hidden, long long hidden_pix_stride,
                             oess_stream_t stream);

/* n <= 3 INDEPENDENT ConvLSTM steps (arguments as oess_convlstm_fused_bf16) in ONE launch: the three levels of E2VID's recurrent
 * encoder (e2vid/model/unet.py:141-150) on the skewed schedule, where level l works on sub-window s - l, so the three Gates
 * convolutions of a stage have no dependency on each other.  Results are those of n separate calls (same tiles, same arithmetic);
 * what the single launch saves is each level's partial last round of tiles and two launch gaps.  No output (hidden, cell) of one
 * problem may overlap any buffer of another (checked).  Geometries the row-halo kernel does not take run as n launches. */
typedef struct {
    const void* in; long long in_pix_stride; int B, H, W, Cin;
    const void* w_packed_gates; const float* bias; int C_hidden, R, S, pad;
    const float* prev_cell; float* cell; void* hidden; long long hidden_pix_stride;
} oess_convlstm_desc_t;
int oess_convlstm_fused_group_bf16(const oess_convlstm_desc_t* problems, int n, oess_stream_t stream);

/* The same n <= 3 ConvLSTM steps on the round-6 kernel (csrc/conv_lstm_w128.h: persistent workgroups, one wave per SIMD on a
 * 128-pixel x 128-gate-column accumulator block, hand-laid MFMA / LDS / LDS-DMA stream).  Replaces ConvLSTM.forward
 * (e2vid/model/submodules.py:199-214) exactly as oess_convlstm_fused_bf16 does; what differs is the STORAGE of the cell state:
 * prev_cell / cell are in the kernel's own "w128-tiled" layout ([tile of 256 pixels x 64 channels][wave][16][lane][4] fp32 = the
 * accumulator layout, oess_convlstm_w128_cell_bytes(B*H*W, C) bytes: the pixel count is padded to a multiple of 256), so the
 * previous cell arrives and the new cell leaves with coalesced 16-byte accesses and no transposition.  The cell state never
 * leaves the recurrent encoder (submodules.py:205-212 only feeds it back); oess_convlstm_w128_cell_relayout converts to and from
 * the reference's [B,H,W,C] order for state import / export and tests.  cell may equal prev_cell (a tile updates its own block)
 * or be disjoint from it; hidden (NHWC bf16, own pixel stride) must not overlap `in`; no output of one problem may overlap
 * another problem's buffers (checked).  Takes 3 x 3 / pad 1 Gates with Cin % 64 == 0, C_hidden % 64 == 0, H >= 8 and 32-bit
 * extents; anything else returns OESS_EINVAL before any launch (callers fall back to oess_convlstm_fused_group_bf16 on a cell
 * state in [B,H,W,C] order).  Gate arithmetic: bias added inside the exponent's FMA, sigmoid / tanh via exp2 + rcp (the fused
 * kernels' fast forms); results agree with oess_convlstm_fused_bf16 to fp32 rounding of that reassociation. */
size_t oess_convlstm_w128_cell_bytes(long long pixels, int C_hidden);      /* 0 when C_hidden % 64 != 0 */
int oess_convlstm_w128_cell_relayout(const float* src, float* dst, long long pixels, int C_hidden, int to_tiled, oess_stream_t stream);
int oess_convlstm_w128_group_bf16(const oess_convlstm_desc_t* problems, int n, oess_stream_t stream);
/* Host-only: the static tile lists the kernel above walks (one list per persistent workgroup; entry = (problem << 24) | tile of 256
 * pixels x 256 gate columns, -1 ends a list; problems longest-K first, greedy onto the least loaded workgroup of the tile's XCD).
 * lists (may be NULL: only *stride is returned) receives grid x *stride ints; capacity = its size in ints.  No device work: the
 * CPU tests check that every tile of every problem appears exactly once. */
int oess_convlstm_w128_tile_lists(const int* tiles_m, const int* tiles_n, const int* cin, int n, int grid, int* lists, int capacity, int* stride);

/* n <= 2 INDEPENDENT 5x5 / stride-2 / pad-2 convolutions (out = act(conv(in, w) + bias), arguments as oess_conv2d_fwd_bf16 with
 * R = S = 5, stride 2, pad 2, relu in {0, 1}) in ONE launch: the encoder ConvLayers of levels 1 and 2 of E2VID's recurrent encoder
 * (e2vid/model/unet.py:141-150, submodules.py:96-115) on the skewed schedule.  Results are those of separate calls.  Neither output
 * may overlap the other problem's input or output (checked); geometries the 2-D-halo kernel does not take run as n launches. */
typedef struct {
    const void* in; long long in_pix_stride; int B, H, W, Cin;
    const void* w_packed; const float* bias; int Cout, relu;
    void* out; long long out_pix_stride;
} oess_conv_s2_desc_t;
int oess_conv5x5s2_group_bf16(const oess_conv_s2_desc_t* problems, int n, oess_stream_t stream);

/* Statistics of a channel slice without the apply pass (first half of K2): stats = {sum, sumsq, nnz, -}. */
/* ... and of n_slices consecutive Cs-channel slices in ONE launch (the 20 sub-windows of a pre-training sample are known up
 * front, pretrain_trainer.py:437-441): stats[4 z ..] = {sum, sumsq, nnz, -} of in[:, z*Cs : (z+1)*Cs].
 * stats holds oess_masked_stats_doubles(n_slices) doubles (1 for the single-slice form): totals first, partial rows behind. */
int oess_masked_stats_slices_f32(const float* in, int B, int Ctot, int Cs, int n_slices, int64_t HW, double* stats,
                                 oess_stream_t stream);
int oess_masked_stats_slice_f32(const float* in, int B, int Ctot, int c0, int Cs, int64_t HW, double* stats,
                                oess_stream_t stream);
/* EventPreprocessor apply (e2vid/utils/inference_utils.py:80-85) fused with the NCHW fp32 -> NHWC bf16
 * (8 channels, zero padded) re-layout that feeds the E2VID head conv.  normalize == 0 only re-lays out. */
int oess_event_slice_to_nhwc8_bf16(const float* in, int B, int Ctot, int c0, int Cs, long long HW, const double* stats,
                                   int normalize, void* out_nhwc8, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation and resampling on NHWC bf16 (pixel strides in elements, C % 8 == 0).
 * Replace nn.BatchNorm2d (train-mode batch statistics; models/_resnet.py:74-114, image_model.py:130-143),
 * nn.InstanceNorm2d + ReLU (models/style_networks.py:252-289), F.interpolate(nearest, x2)
 * (style_networks.py:148-160) and nn.Upsample(x4, bilinear, align_corners=True) + F.normalize
 * (image_model.py:121-143).  G groups of pixels_per_group pixels: G = 1 BatchNorm, G = B InstanceNorm.
 * ------------------------------------------------------------------------------------------ */
/* Statistics are DETERMINISTIC: every workgroup writes its partial sums to `partials` (caller workspace of
 * oess_norm_partials_bytes(G, pixels_per_group, C, backward) bytes; backward = 1 for the two *_bwd entry points) and a second
 * kernel adds the rows in a fixed order in double -- no floating-point atomics anywhere on the training path. */
size_t oess_norm_partials_bytes(int G, long long pixels_per_group, int C, int backward);
/* sum[G x C] = sum of x, sumsq[G x C] = sum of x^2 (overwritten; also the bias gradient: sum of dY over pixels) */
int oess_norm_stats_nhwc_bf16(const void* x, long long x_pix_stride, int G, long long pixels_per_group, int C,
                              float* sum, float* sumsq, float* partials, size_t partials_bytes, oess_stream_t stream);
/* statistics + mean / rstd / scale = gamma*rstd / shift = beta - mean*scale per (group, channel), E[x^2] - E[x]^2 in double;
 * G == 1 and running_mean != NULL: nn.BatchNorm2d's running-statistics update (unbiased variance).  gamma / beta nullable. */
int oess_norm_stats_finalize_nhwc_bf16(const void* x, long long x_pix_stride, int G, long long pixels_per_group, int C, float eps,
                                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                                       float momentum, float* mean, float* rstd, float* scale, float* shift, float* partials,
                                       size_t partials_bytes, oess_stream_t stream);
/* out = act(x*scale + shift [+ residual]) */
int oess_norm_apply_nhwc_bf16(const void* x, long long x_pix_stride, const float* scale, const float* shift,
                              const void* residual, long long res_pix_stride, int relu, int G, long long pixels_per_group,
                              int C, void* out, long long out_pix_stride, oess_stream_t stream);
/* affine-free InstanceNorm (+ReLU) backward; s1/s2 are [G x C] scratch */
int oess_instnorm_bwd_nhwc_bf16(const void* x, long long x_pix_stride, const void* dy, long long dy_pix_stride,
                                const float* mean, const float* rstd, int relu, int G, long long pixels_per_group, int C,
                                float* s1, float* s2, void* dx, long long dx_pix_stride, float* partials, size_t partials_bytes,
                                oess_stream_t stream);
/* nn.BatchNorm2d TRAIN-mode backward fused with the ReLU mask and the residual branch of a Bottleneck
 * (models/_resnet.py:96-114: out = relu(bn3(conv3) + identity)):  g = dy * (y_out > 0 if relu);  d(residual) = g;
 * dbeta[C] = sum g;  dgamma[C] = sum g * xhat;  dx = gamma * rstd * (g - dbeta/N - xhat * dgamma/N).
 * x = the BatchNorm input, y_out = the stored forward output (needed when relu != 0), mean / rstd = the batch statistics
 * of the forward (oess_norm_stats_finalize_nhwc_bf16).  dresidual nullable.  Replaces MIOpenBatchNormBwdSpatial* + the ATen ReLU / add
 * backward kernels on the trainable DeepLabv3 path. */
int oess_batchnorm_bwd_nhwc_bf16(const void* x, long long x_pix_stride, const void* dy, long long dy_pix_stride, const void* y_out,
                                 long long y_pix_stride, const float* mean, const float* rstd, const float* gamma, int relu,
                                 long long pixels, int C, float* dbeta, float* dgamma, void* dx, long long dx_pix_stride,
                                 void* dresidual, long long dres_pix_stride, float* partials, size_t partials_bytes,
                                 oess_stream_t stream);
int oess_upsample_nearest2x_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, void* out,
                                      long long out_pix_stride, oess_stream_t stream);
/* z[b, s*y, s*x, :] = in[b, y, x, :], zero elsewhere on an Hz x Wz grid: turns the data gradient of a stride-s convolution
 * into the stride-1 product dX = conv(z, oess_conv2d_pack_weight(flip_for_dgrad = 1), pad = dil*(R-1) - pad) with
 * Hz = H_in - dil*(R-1) + 2*pad (replaces aten::convolution_backward for the stride-2 3x3 / 1x1 convolutions of
 * models/_resnet.py:74-114; MIOpen resolved those to a naive 9.4 ms kernel on gfx950). */
int oess_zero_insert_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, int stride, int Hz, int Wz,
                               void* out, long long out_pix_stride, oess_stream_t stream);
int oess_downsample_sum2x_nhwc_bf16(const void* gout, long long gout_pix_stride, int B, int H, int W, int C, void* gin,
                                    long long gin_pix_stride, oess_stream_t stream);
/* nn.Upsample(scale, bilinear, align_corners=True) + F.normalize(dim=1) in one pass (models/image_model.py:121-143).  inv_norm
 * (nullable, needs normalize != 0): [B * H*scale * W*scale] fp32, 1 / max(|x|, 1e-12) of every output pixel -- what
 * oess_l2norm_nhwc_bwd needs, so the differentiable head never materialises the un-normalised full-resolution tensor. */
int oess_bilinear_l2norm_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, int scale,
                                   int normalize, void* out, long long out_pix_stride, float* inv_norm, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bilinear resampling (any size, both align_corners modes) and channel L2 normalisation, forward + adjoint.
 * Replace F.interpolate(size=input_shape, mode='bilinear', align_corners=False) on DeepLabv3's logits and features
 * (models/deeplabv3.py:179-189) and nn.Upsample(x4, bilinear, align_corners=True) + F.normalize(p=2, dim=1) of the
 * teacher (models/image_model.py:121-143) including their autograd backward.  NHWC, pixel strides in elements,
 * bf16 (is_bf16 != 0) or fp32 (same dtype in and out).  The backward is a two-pass deterministic gather; workspace =
 * oess_resize_bilinear_bwd_workspace_bytes(B, W, C, Ho) bytes (fp32 [B, Ho, W, C]).
 * inv_norm: fp32 [P] = 1 / max(|x|, eps) per pixel, written by the forward (nullable) and read by the backward.
 * ------------------------------------------------------------------------------------------ */
int oess_resize_bilinear_nhwc_fwd(const void* in, long long in_pix_stride, int B, int H, int W, int C, int is_bf16, int Ho, int Wo,
                                  int align_corners, void* out, long long out_pix_stride, oess_stream_t stream);
size_t oess_resize_bilinear_bwd_workspace_bytes(int B, int W, int C, int Ho);
int oess_resize_bilinear_nhwc_bwd(const void* grad_out, long long gout_pix_stride, int B, int H, int W, int C, int is_bf16, int Ho,
                                  int Wo, int align_corners, void* workspace, size_t workspace_bytes, void* grad_in,
                                  long long gin_pix_stride, oess_stream_t stream);
int oess_l2norm_nhwc_fwd(const void* x, long long x_pix_stride, int64_t P, int C, int is_bf16, float eps, void* y,
                         long long y_pix_stride, float* inv_norm, oess_stream_t stream);
int oess_l2norm_nhwc_bwd(const void* y, long long y_pix_stride, const void* grad_y, long long gy_pix_stride, const float* inv_norm,
                         int64_t P, int C, int is_bf16, float eps, void* grad_x, long long gx_pix_stride, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MaskCLIP ViT-B/16 image tower (models/maskclip_model.py:448-541 TransformerEncoderLayer, :545-851 VisionTransformer):
 * the non-GEMM pieces.  The linear layers (patch embedding, in_proj, out_proj, FFN, proj) run on oess_conv2d_fwd_bf16 as
 * 1x1 / 16x16 convolutions over the token axis (bias, residual and GELU fused in the epilogue).
 *   oess_layernorm_bf16: y = (x - mean) * rsqrt(var + eps) * gamma + beta over C channels of each of `rows` tokens
 *     (nn.LayerNorm, eps 1e-6 in the ViT; build_norm_layer(dict(type='LN', eps=1e-6)) :486-501,:706-718).
 *   oess_attention_d64_bf16: softmax(Q K^T * scale) V for head dimension 64; qkv = nn.MultiheadAttention's packed
 *     in_proj output [B, L, 3 * heads * 64] (q | k | v), out = [B, L, heads * 64] before out_proj
 *     (mmcv MultiheadAttention -> nn.MultiheadAttention.forward, :496-497,:538).
 * Row strides in elements (multiples of 8), pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
int oess_layernorm_bf16(const void* x, long long x_row_stride, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                        void* y, long long y_row_stride, oess_stream_t stream);
int oess_attention_d64_bf16(const void* qkv, long long qkv_row_stride, int B, int L, int heads, float scale, void* out,
                            long long out_row_stride, oess_stream_t stream);

/* Weight gradient of oess_conv2d_fwd_bf16's convolution: dW (OIHW fp32, every element WRITTEN: no pre-zeroing)
 * from x (NHWC bf16, Cin_x >= Cin channels present, Cin_x % 8 == 0) and dy (NHWC bf16, Cout % 8 == 0).
 * Replaces the weight half of ATen's convolution_backward behind nn.Conv2d (models/style_networks.py:252-289). */
int oess_conv2d_wgrad_bf16(const void* x, long long x_pix_stride, int B, int H, int W, int Cin_x, const void* dy,
                           long long dy_pix_stride, int Cout, int Cin, int R, int S, int stride, int pad, int dil,
                           float* dw_oihw, void* workspace, size_t workspace_bytes, oess_stream_t stream);
/* workspace: split-K partial tiles; any size >= one padded tile set works, 64 MiB lets the kernel use full parallelism */

/* ------------------------------------------------------------------------------------------
 * GPU-side decode of 8-bit single-channel PNG maps: pseudo-labels, superpixel ids, ground-truth labels.
 * Replaces np.array(Image.open(path)) + torch.tensor(...).long() [+ torch.flip(..., [1])] of
 * DSEC/dataset/sequence_ov.py:340-358,366-372 and datasets/ddd17_events_loader.py:228-262 for a whole batch:
 *   files    the n_images PNG files back to back (bytes as read from disk), offsets[n_images + 1] their byte offsets (device),
 *            total_file_bytes = offsets[n_images] (host copy: sizes the scratch check)
 *   flip     optional uint8[n_images]: 1 = mirror the map horizontally (the loader's flip augmentation)
 *   out      int64 [n_images][H][W] (the dtype the trainers index with)
 *   scratch  oess_png_decode_scratch_bytes(total bytes, n_images, H, W); scratch_offsets[n_images] (device): byte offset of image
 *            i's region = sum over j < i of (align16(len_j) + H * (W + 1) + 16)
 *   status   int[n_images]: 0 = decoded; otherwise the map is filled with 255 (ignore index) and the code says why
 *            (1 signature, 2 IHDR, 3 not 8-bit grey / palette or interlaced, 4 zlib header, 5 block, 6 code, 7 overrun,
 *            8 size mismatch, 9 filter).  Supported: colour type 0 or 3 (indices), bit depth 8, no interlace, any IDAT split,
 *            stored / fixed / dynamic DEFLATE blocks, all five PNG filters.  Exact (integer work): equals PIL's array.
 * ------------------------------------------------------------------------------------------ */
size_t oess_png_decode_scratch_bytes(long long total_file_bytes, int n_images, int H, int W);
int oess_png_decode_gray8_batch(const uint8_t* files, const int64_t* offsets, long long total_file_bytes, int n_images, int H, int W,
                                const uint8_t* flip, int64_t* out, void* scratch, size_t scratch_bytes, const int64_t* scratch_offsets,
                                int* status, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Linear probe: nn.Conv2d(K, K, 1) on the fp32 logits (models/style_networks.py:113-133,169-170; models/deeplabv3.py:162-170,
 * 186-187), K <= 32.  x, y, grad_*: dense NHWC fp32 [P x K]; w: [K x K] (Conv2d.weight viewed [out, in]); bias nullable in the
 * forward.  Backward: grad_x nullable (frozen producer); grad_w / grad_bias from per-workgroup double partial rows added in a fixed order (bit-repeatable);
 * partials: oess_linear_probe_partials_bytes(K) bytes of caller scratch.
 * ------------------------------------------------------------------------------------------ */
size_t oess_linear_probe_partials_bytes(int K);
int oess_linear_probe_fwd_f32(const float* x, const float* w, const float* bias, long long P, int K, float* y, oess_stream_t stream);
int oess_linear_probe_bwd_f32(const float* x, const float* grad_y, const float* w, long long P, int K, float* grad_x, float* grad_w,
                              float* grad_bias, void* partials, size_t partials_bytes, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backward of k = scatter_mean(F.normalize(nn.Upsample(bilinear)(x)), superpixels) as one node (models/image_model.py:121-143 +
 * training/pretrain_trainer.py:445-465): feat = the saved normalised full-resolution map [B x Ho x Wo x C] bf16, inv_norm its
 * 1 / max(|x|, eps) per pixel (both from oess_bilinear_l2norm_nhwc_bf16), ids raw superpixel ids [B*Ho*Wo], grad_k [S x C] fp32,
 * count [S] (oess_segment_mean_fwd).  grad_in: [B x H x W x C] bf16.  C in {64, 128, 256, 512}.  Deterministic (gather form).
 * ------------------------------------------------------------------------------------------ */
size_t oess_bilinear_l2norm_pool_bwd_workspace_bytes(int B, int W, int C, int Ho, int S);
int oess_bilinear_l2norm_pool_bwd_bf16(const void* feat, long long feat_pix_stride, const float* inv_norm, const int64_t* ids,
                                       const float* grad_k, const float* count, int superpixel_size, int S, int B, int H, int W, int C,
                                       int Ho, int Wo, int align_corners, float eps, void* workspace, size_t workspace_bytes,
                                       void* grad_in, long long gin_pix_stride, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Superpixel mean of a bilinearly upsampled map through its pooling matrix (models/deeplabv3.py:184 F.interpolate(feats, size=
 * input, bilinear, align_corners=False) followed by training/pretrain_trainer.py:445-465 on it): k[s] = (sum_q M[s][q] y[q]) /
 * (n[s] + 1e-6) with M[s][q] = the summed bilinear weights of superpixel row s's pixels on low-resolution pixel q.
 *   build: ids [B x Ho x Wo] raw ids (row = id + sample * superpixel_size, rows outside [0, S) dropped) -> matrix
 *          (oess_pool_matrix_bytes(S, B, h, w) bytes, 16-byte aligned; 2^-40 fixed-point sums + pixel counts; order-independent)
 *   fwd:   y [B x h x w x C] bf16 / fp32 -> k [S x C] fp32, count [S] fp32
 *   bwd:   grad_k [S x C] -> grad_y [B x h x w x C];  C <= 1024.  The full-resolution map is never formed.
 * ------------------------------------------------------------------------------------------ */
size_t oess_pool_matrix_bytes(int S, int B, int h, int w);
int oess_pool_matrix_build(const int64_t* ids, int B, int Ho, int Wo, int h, int w, int align_corners, int superpixel_size, int S,
                           void* matrix, size_t matrix_bytes, oess_stream_t stream);
int oess_pool_matrix_fwd(const void* matrix, const void* y, long long y_pix_stride, int is_bf16, int B, int h, int w, int C, int S,
                         float* k, float* count, oess_stream_t stream);
int oess_pool_matrix_bwd(const void* matrix, const float* grad_k, int B, int h, int w, int C, int S, void* grad_y, long long gy_pix_stride,
                         int is_bf16, oess_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Small ops of the DeepLabv3 training path (round 5; they replace the last ATen / MIOpen / hipBLASLt launches of that path).
 *
 * MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem (models/_resnet.py:124, 197 of the reference) on NHWC bf16:
 *   out [B x Ho x Wo x C], Ho = (H - 1) / 2 + 1; idx (nullable): [B x Ho x Wo x C] bytes, winning tap 0..8 (row-major, ATen's tie
 *   rule: first maximum, NaN wins); backward: grad_in [B x H x W x C] written once per element (gather form, deterministic).
 * Dropout (models/deeplabv3.py:343 nn.Dropout(0.1)): y = x * keep / (1 - p), keep from Philox-4x32-10 keyed by (seed, offset,
 *   element); the backward pass is the same call on the gradient with the same (seed, offset).  x == y allowed.
 * ASPP image-pooling branch (models/deeplabv3.py:305-316): in_scale * pooled [B x Cin] fp32 (AdaptiveAvgPool2d(1): per-sample
 *   channel sums and in_scale = 1 / (H W)) -> 1x1 conv w [Cout x Cin] -> BatchNorm2d in train mode over the B samples (running
 *   stats updated, unbiased variance) -> ReLU = z [B x Cout] (+ a bf16 copy, nullable); y_pre [B x Cout] and stat [2 x Cout] = {mean, rstd} are kept for
 *   the backward; 2 <= B <= 16.  Backward: grad_z [B x Cout] -> grad_w [Cout x Cin], grad_gamma / grad_beta [Cout], and the
 *   gradient of the per-sample sums as bf16 [B x Cin] (nullable: frozen producer); dy_scratch [B x Cout] floats.
 * ------------------------------------------------------------------------------------------ */
int oess_maxpool3x3s2_fwd_nhwc_bf16(const void* in, long long in_pix_stride, int B, int H, int W, int C, void* out, long long out_pix_stride,
                                    unsigned char* idx, oess_stream_t stream);
int oess_maxpool3x3s2_bwd_nhwc_bf16(const void* grad_out, long long go_pix_stride, const unsigned char* idx, int B, int H, int W, int C,
                                    void* grad_in, long long gi_pix_stride, oess_stream_t stream);
int oess_dropout_nhwc_bf16(const void* x, long long x_pix_stride, void* y, long long y_pix_stride, long long P, int C, float p,
                           unsigned long long seed, unsigned long long offset, oess_stream_t stream);
int oess_aspp_pool_fwd_f32(const float* pooled, float in_scale, const float* w, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float momentum, float eps, int B, int Cin, int Cout, float* y_pre, float* stat, float* z,
                           void* z_bf16, oess_stream_t stream);
int oess_aspp_pool_bwd_f32(const float* grad_z, const float* pooled, float in_scale, const float* w, const float* gamma, const float* y_pre,
                           const float* stat, const float* z, int B, int Cin, int Cout, float* dy_scratch, float* grad_w,
                           float* grad_gamma, float* grad_beta, void* grad_pooled_bf16, oess_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OESS_H */
