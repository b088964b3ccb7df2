/* flair_hip.h — C ABI of the MI355X-native FLAIR-1 segmentation hot path (libflair_hip.so).
 *
 * Drop-in boundary.  The reference is Python; the calls a maintainer would bind
 * (ctypes) are listed next to the reference interface each one replaces.
 * Conventions: plain pointers and sizes only, no torch types; every pointer is DEVICE memory unless
 * marked host; every function enqueues work on `stream` (a hipStream_t passed as void*) and never
 * synchronises, allocates or frees device memory; return value 0 = ok, >0 = hipError_t,
 * <0 = argument error (flair_strerror).  Tensors at the boundary use the reference's layouts:
 * images / logits / features NCHW fp32, labels (B,H,W), parameters PyTorch OIHW fp32.
 * Internally activations are NHWC in `dtype` (0 = fp32 parity mode, 1 = bf16 throughput mode).
 */
#ifndef FLAIR_HIP_H
#define FLAIR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLAIR_DT_F32 0
#define FLAIR_DT_BF16 1

const char* flair_strerror(int code);
int flair_version(void);

/* ------------------------------------------------------------------------------------------------
 * Model object.  Replaces smp.create_model(arch='unet', encoder_name='resnet34', classes, in_channels)
 * — /root/reference/src/flair/model.py:37-41 and src/zone_detect/model.py:34-39.
 * The handle holds only host-side metadata (layer table, arena plan); parameters stay in the
 * caller's flat fp32 buffers laid out as flair_unet_tensor_info describes (names = smp-0.3.3
 * state_dict keys). */
typedef struct flair_unet flair_unet_t;
int flair_unet_create(flair_unet_t** out, int in_channels, int classes, int dtype);
void flair_unet_destroy(flair_unet_t* h);
int64_t flair_unet_param_count(const flair_unet_t* h);   /* floats in the flat parameter / gradient buffer */
int64_t flair_unet_buffer_count(const flair_unet_t* h);  /* floats in the flat BN running-stat buffer */
int flair_unet_num_tensors(const flair_unet_t* h);
/* kind: 0 = parameter (offset into params/grads), 1 = running statistic (offset into buffers);
 * stage: 0 stem, 1-4 encoder layers, 5 decoder, 6 head — contiguous gradient buckets. */
int flair_unet_tensor_info(const flair_unet_t* h, int i, char* name, int name_cap, int64_t shape[4], int* ndim,
                           int64_t* offset, int* kind, int* stage);
int flair_unet_stage_range(const flair_unet_t* h, int stage, int64_t* begin, int64_t* end);
int64_t flair_unet_workspace_bytes(flair_unet_t* h, int B, int H, int W, int training);

/* seg_model(x)  — model.py:64 (and zone_detect/compare.py:31).  training != 0: BatchNorm batch
 * statistics + running-stat update in `buffers`, activations kept in `workspace` for backward. */
int flair_unet_forward(flair_unet_t* h, const float* params, float* buffers, const float* x_nchw, float* logits_nchw,
                       int B, int H, int W, int training, void* workspace, size_t workspace_bytes, void* stream);
/* autograd backward of the call above (Lightning's loss.backward(), task_module.py:82-86).
 * Exactly one of dlogits_nchw (fp32 (B,C,H,W)) / dlogits_nhwc (as written by flair_ce_head with
 * flair_unet_head_ld) is non-null.  Writes every parameter gradient into `grads` (flat, OIHW).
 * stage_events: optional array of 7 hipEvent_t (null = none); event k is recorded on `stream` as soon as
 * the gradients of bucket k (flair_unet_stage_range) are final — ready order 6,5,4,3,2,1,0 — so the
 * host can start the RCCL all-reduce of a bucket while the rest of backward still runs. */
int flair_unet_backward(flair_unet_t* h, const float* params, const float* dlogits_nchw, const void* dlogits_nhwc,
                        float* grads, void* workspace, size_t workspace_bytes, void* stream,
                        void* const* stage_events);
int flair_unet_head_ld(const flair_unet_t* h);
/* Inference with constant weights (zone_detect's window loop, predict): a one-shot promise that `params` and `buffers` are
 * bit-identical to those of the previous eval-mode flair_unet_forward on this handle.  If the next eval-mode forward also
 * uses the same workspace and shape, it skips re-packing the weights and re-deriving the 46 BatchNorm affine pairs (both
 * are still in the workspace); any other call clears the promise.  The reference has no counterpart: PyTorch modules keep
 * their weights in the layout they compute in. */
int flair_unet_reuse_constants(flair_unet_t* h, int on);
/* One-shot: the next flair_unet_forward with training = 0 and logits = NULL writes uint8 argmax predictions [B][H][W] to preds_u8
 * (device pointer) — predict_step's argmax(softmax(logits)), task_module.py:206-213 — from the head convolution's epilogue; the
 * logits are then not kept (flair_unet_logits_nhwc returns NULL until the next forward). */
int flair_unet_want_preds(flair_unet_t* h, uint8_t* preds_u8, float* maxprob_f32 /* optional: the winner's softmax probability */);

/* seg_model.encoder(x) / .decoder(*feats) / .segmentation_head(t) — the metadata path model.py:57-62.
 * feats[i] = feature i+1 of the encoder, NCHW fp32: (B,64,H/2,W/2) ... (B,512,H/32,W/32). */
int flair_unet_encoder_forward(flair_unet_t* h, const float* params, float* buffers, const float* x_nchw,
                               float* const feats_nchw[5], int B, int H, int W, int training, void* workspace,
                               size_t workspace_bytes, void* stream);
int flair_unet_decoder_forward(flair_unet_t* h, const float* params, float* buffers, const float* const feats_nchw[5],
                               float* out_nchw, int B, int H, int W, int training, void* workspace,
                               size_t workspace_bytes, void* stream);
int flair_unet_head_forward(flair_unet_t* h, const float* params, const float* x_nchw, float* logits_nchw, int B, int H,
                            int W, int training, void* workspace, size_t workspace_bytes, void* stream);
int flair_unet_head_backward(flair_unet_t* h, const float* params, const float* dlogits_nchw, float* dx_nchw,
                             float* grads, void* workspace, size_t workspace_bytes, void* stream);
int flair_unet_decoder_backward(flair_unet_t* h, const float* params, const float* dout_nchw,
                                float* const dfeats_nchw[5], float* grads, void* workspace, size_t workspace_bytes,
                                void* stream);
int flair_unet_encoder_backward(flair_unet_t* h, const float* params, const float* const dfeats_nchw[5], float* grads,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused per-pixel head.  Replaces, in one pass over the logits,
 *   targets = argmax(msk, 1); loss = criterion(logits, targets)            task_module.py:71-72
 *   nn.CrossEntropyLoss(weight) mean reduction                              tasks_utils.py:88-93
 *   preds = argmax(softmax(logits, 1), 1)                                   task_module.py:75-76
 *   MulticlassJaccardIndex.update: confmat += bincount(target*C + pred)     task_module.py:85,107-108
 * label_kind: 0 uint8 (B,H,W), 1 int32, 2 int64, 3 fp32 one-hot (B,C,H,W) (the reference's batch["msk"]).
 * Outputs are optional (null = skip).  workspace: flair_ce_workspace_bytes(). */
size_t flair_ce_workspace_bytes(int B, int H, int W);
int flair_ce_head(const float* logits_nchw, const void* labels, int label_kind, const float* class_weight, int B, int C,
                  int H, int W, float* loss, float* dlogits_nchw, void* dlogits_nhwc, int dlogits_dtype, int dlogits_ld,
                  uint8_t* preds_u8, int64_t* preds_i64, int32_t* targets_i32, int64_t* confmat, void* workspace,
                  void* stream);
/* The same head over logits left in the network's own layout: flair_unet_forward called with logits_nchw == NULL
 * (training) keeps the segmentation head's output as NHWC rows [B*H*W][flair_unet_head_ld()] of the model's compute
 * dtype inside the workspace; flair_unet_logits_nhwc returns that buffer (valid until the next forward on the arena).
 * Same arithmetic and outputs as flair_ce_head (in bf16 mode the fp32 NCHW logits are these values widened); dl_nhwc has
 * the layout flair_unet_backward takes.  Replaces task_module.py:71-79 + tasks_utils.py:88-93 for a loop that never
 * looks at the logits themselves (flair_amd.SegTrainer). */
const void* flair_unet_logits_nhwc(const flair_unet_t* h);
int flair_ce_head_nhwc(const void* logits_nhwc, int dtype, int ld, const void* labels, int label_kind,
                       const float* class_weight, int B, int C, int H, int W, float* loss, void* dlogits_nhwc,
                       uint8_t* preds_u8, int32_t* targets_i32, int64_t* confmat, void* workspace, void* stream);
/* predict_step over NHWC logits left in the workspace (flair_unet_forward with logits_nchw == NULL, flair_unet_logits_nhwc). */
int flair_softmax_argmax_nhwc(const void* logits_nhwc, int dtype, int ld, int B, int C, int H, int W, uint8_t* preds_u8,
                              int64_t* preds_i64, float* maxprob, void* stream);
/* predict_step: argmax(softmax(logits)) — task_module.py:211-212; with maxprob also
 * zone_detect inference + convert('argmax') — src/zone_detect/compare.py:35, dataset.py:23-30. */
int flair_softmax_argmax(const float* logits_nchw, int B, int C, int H, int W, uint8_t* preds_u8, int64_t* preds_i64,
                         float* maxprob, void* stream);
/* confmat[target][pred] += 1 — torchmetrics update / sklearn confusion_matrix at src/flair/metrics.py:67-71.
 * kinds: 0 uint8, 1 int32, 2 int64. */
int flair_confmat_update(const void* target, int target_kind, const void* pred, int pred_kind, int64_t n, int C,
                         int64_t* confmat, void* stream);
/* Offline evaluation: sum over tiles of sklearn.confusion_matrix((truth - 1).flatten(), pred.flatten(),
 * labels=range(C)) — src/flair/metrics.py:60-75.  truth_offset is added to the truth byte with uint8
 * wrap-around (-1: a stored 0 becomes 255); pairs with a member outside range(C) are dropped.  Both rasters
 * 16-byte aligned. */
int flair_confmat_masks(const uint8_t* truth_raw, const uint8_t* pred, int64_t n, int C, int truth_offset, int64_t* confmat,
                        void* stream);
/* MulticlassJaccardIndex.compute (average None / 'weighted' / 'macro') — task_module.py:36-51,90,113-114. */
int flair_jaccard(const int64_t* confmat, int C, float* per_class, float* weighted, float* macro, void* stream);

/* torch.optim.SGD(lr) step, no momentum / weight decay — tasks_utils.py:95. */
/* ------------------------------------------------------------------------------------------------
 * Input contract of the step on device: what fit_dataset.__getitem__ /
 * predict_dataset.__getitem__ (src/flair/data_loader.py:74-95,130-144) and the augmentation set of
 * src/flair/tasks_utils.py:37-41 do per tile on CPU workers, for a whole batch of stored uint8 rasters.
 *   img_u8 (B, bands, H, W): bands as stored; channels[n_channels] = the config's 1-based band list
 *   norm_type 0 'without' | 1 'scaling' (x * (1/255) in fp64) | 2 'custom' ((x - mean) / std in fp64), then fp32
 *     — data_loader.py:9-30
 *   msk_raw (B, H, W): stored label raster; labels_out = argmax over the one-hot of (raw - 1) (data_loader.py:65-69,
 *     task_module.py:71): raw 1..C -> 0..C-1, anything else -> class 0
 *   d4_flags (B) or null: bit0 VerticalFlip, bit1 HorizontalFlip, bits 2-3 RandomRotate90 factor, applied in that
 *     order to image and labels alike; needs H == W.
 * img_out (B, n_channels, H, W) fp32 NCHW and/or labels_out (B, H, W) uint8; either may be null. W % 4 == 0. */
int flair_feed_tiles(const uint8_t* img_u8, const uint8_t* msk_raw, const uint8_t* d4_flags, int B, int bands, int H, int W,
                     const int* channels, int n_channels, int norm_type, const double* means, const double* stds,
                     int num_classes, float* img_out, uint8_t* labels_out, void* stream);
/* zone_detect: softmax over classes (src/zone_detect/compare.py:35), margin crop (compare.py:71-75) and
 * convert (src/zone_detect/dataset.py:11-34) without the probability tensor ever leaving the device.
 * output_type 0 'argmax': out fp32 (B, 2, S-2m, S-2m) = [first argmax, max probability];
 * output_type 1 'class_prob': out uint8 (B, C, S-2m, S-2m) = trunc(p * 255);
 * output_type 2 (flair_detect_convert only): no convert, out fp32 (B, C, S-2m, S-2m) = the probabilities themselves — with
 *   m = 0 the return value of the reference's inference(), compare.py:35-39. */
int flair_detect_convert(const float* logits_nchw, int B, int C, int S, int margin, int output_type, void* out, void* stream);
/* Sliding-window detection over ONE raster resident in HBM (zone_detect default pipeline, src/zone_detect/main.py:386-428).
 * tiles (device, B x 6 int32): {x0, y0, wx0, wx1, wy0, wy1} per window — top-left pixel of the S x S window in raster
 * coordinates (may lie outside: Sliced_Dataset reads boundless, src/zone_detect/dataset.py:96-103, so missing pixels are 0
 * before normalisation) and the raster rectangle [wx0,wx1) x [wy0,wy1) this window owns in the output (its margin-cropped
 * centre minus what later windows of the slicing job overwrite; stitching 'exact-clipping', compare.py:69-82).
 * flair_gather_tiles: the dataset's read + normalization (dataset.py:66-88,90-113) -> img_out fp32 (B, n_channels, S, S).
 * flair_detect_stitch: softmax + crop + convert as flair_detect_convert, written straight into raster_out
 *   ((2, raster_h, raster_w) fp32 for 'argmax', (C, raster_h, raster_w) uint8 for 'class_prob'). */
int flair_gather_tiles(const uint8_t* raster_u8, int bands, int raster_h, int raster_w, const int32_t* tiles, int B, int S,
                       const int* channels, int n_channels, int norm_type, const double* means, const double* stds,
                       float* img_out, void* stream);
/* flair_detect_stitch_preds: convert('argmax') + stitch (dataset.py:23-30, main.py:404-421) from the per-window class / probability maps
 * flair_unet_want_preds leaves, instead of from logits: raster_out (2, raster_h, raster_w) fp32 = [class, its softmax probability]. */
int flair_detect_stitch_preds(const uint8_t* preds_u8, const float* maxprob_f32, int B, int S, int margin, const int32_t* tiles,
                              float* raster_out, int raster_h, int raster_w, void* stream);
int flair_detect_stitch(const float* logits_nchw, int B, int C, int S, int margin, int output_type, const int32_t* tiles,
                        void* raster_out, int raster_h, int raster_w, void* stream);

int flair_sgd_step(float* params, const float* grads, int64_t n, float lr, void* stream);
/* feats[-1] += x_enc.unsqueeze(1).unsqueeze(-1).repeat(1,512,1,16) — model.py:59-60: x (N,C,H,W) += v (N,H). */
int flair_add_rowvec_nchw(float* x, const float* v, int N, int C, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Operator-level entry points (unit parity tests; same kernels the model object launches).
 * Activations NHWC in `dtype`; weights PyTorch OIHW fp32. */
size_t flair_conv2d_workspace_bytes(int dtype, int N, int H, int W, int C0, int C1, int up0, int Cout, int R,
                                    int stride, int pad);
/* y = conv2d(cat([up2(x0)?, x1]), w) (+bias); optional fp32 NCHW copy; optional BN batch statistics
 * (sum, sum of squares per channel, reduced) into stats[2][Cout]. */
int flair_conv2d_forward(int dtype, const void* x0, const void* x1, int N, int H, int W, int C0, int C1, int up0,
                         const float* w_oihw, const float* bias, int Cout, int R, int stride, int pad, void* y_nhwc,
                         float* y_nchw, float* stats, void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv2d_backward_data(dy, w); dw = conv2d_backward_weight(x, dy) */
int flair_conv2d_backward(int dtype, const void* x0, int N, int H, int W, int Cin, const float* w_oihw, int Cout, int R,
                          int stride, int pad, const void* dy_nhwc, void* dx_nhwc, float* dw_oihw, void* workspace,
                          size_t workspace_bytes, void* stream);
int flair_bn_relu_forward(int dtype, const void* y, int64_t rows, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int training, const void* residual, int relu,
                          void* out, float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes,
                          void* stream);
int flair_bn_relu_backward(int dtype, const void* dout, const void* out, const void* y, int64_t rows, int C,
                           const float* gamma, const float* save_mean, const float* save_invstd, int relu, void* dy,
                           void* dres, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                           void* stream);
int flair_maxpool_forward(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int flair_maxpool_backward(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C,
                           void* stream);
int flair_nchw_to_nhwc(int dtype, const float* x_nchw, void* y_nhwc, int N, int C, int H, int W, int Cpad, void* stream);
int flair_nhwc_to_nchw(int dtype, const void* x_nhwc, float* y_nchw, int N, int C, int H, int W, int Cpad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MetadataMLP (replaces /root/reference/src/flair/model.py:74-96, called at model.py:58): Linear(45,64) -> Dropout(0.4) ->
 * ReLU -> Linear(64,32) -> Dropout -> ReLU -> Linear(32,16) -> Dropout -> ReLU.  x [B][45] fp32; w[i] / b[i] the three
 * nn.Linear weights ([out][in]) and biases; mask[i] optional [B][64|32|16] dropout masks ALREADY scaled by 1/(1-p)
 * (NULL array or NULL entries: eval mode); h1 [B][64], h2 [B][32] receive the hidden activations (needed by backward,
 * may be NULL); out [B][16].  backward: dout [B][16] -> dw[i] / db[i] (overwritten); scratch >= B*112 floats; B <= 256.
 * The metadata vector is a model input: no gradient w.r.t. x is produced (the reference never asks for one). */
int flair_metadata_mlp_forward(const float* x, const float* const w[3], const float* const b[3], const float* const mask[3],
                               float* h1, float* h2, float* out, int B, void* stream);
int flair_metadata_mlp_backward(const float* x, const float* h1, const float* h2, const float* out, const float* const w[3],
                                const float* const mask[3], const float* dout, int B, float* const dw[3], float* const db[3],
                                float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Measurement aid (bench.py "roofline"): while enabled every kernel launch of the library is bracketed by
 * HIP events on its launch stream.  stop() synchronises the device and returns the number of distinct
 * kernels; kernel(i) gives total time, launch count and the ALGORITHMIC flops / bytes of those launches. */
int flair_profile_start(int max_records);
int flair_profile_stop(void);
int flair_profile_kernel(int i, char* name, int name_cap, double* total_ms, int64_t* launches, double* flops,
                         double* bytes);

/* ------------------------------------------------------------------------------------------------
 * SegFormer (MiT encoder + all-MLP decode head), inference only: the HuggingFace provider of zone_detect —
 * /root/reference/src/zone_detect/model.py:42-50 (AutoModelForSemanticSegmentation.from_pretrained) and
 * compare.py:31-36 (`model(imgs).logits`); BASELINE config 5 names SegFormer-MiT-B2 with 5 input channels.
 * Replaces transformers' SegformerForSemanticSegmentation.forward in eval mode.  Tensor names / shapes
 * (flair_segformer_tensor_info) are that library's state_dict keys; every tensor, BatchNorm running
 * statistics included (kind 1), lives in ONE flat fp32 buffer.  Geometry: MiT-B1..B5 (heads of 64 channels,
 * reduction ratios 8/4/2/1); H, W multiples of 32 with (H/32)*(W/32) a multiple of 16 and <= 256. */
typedef struct flair_segformer flair_segformer_t;
int flair_segformer_create(flair_segformer_t** out, int in_channels, int num_labels, const int depths[4],
                           const int hidden_sizes[4], const int num_heads[4], const int sr_ratios[4],
                           int decoder_hidden_size, int dtype);
void flair_segformer_destroy(flair_segformer_t* h);
int64_t flair_segformer_param_count(const flair_segformer_t* h);
int flair_segformer_num_tensors(const flair_segformer_t* h);
int flair_segformer_tensor_info(const flair_segformer_t* h, int i, char* name, int name_cap, int64_t shape[4], int* ndim,
                                int64_t* offset, int* kind);
int64_t flair_segformer_workspace_bytes(flair_segformer_t* h, int B, int H, int W);
/* The forward keeps what depends on the weights alone (packed GEMM operands, folded BatchNorm, the decode head's
 * pre-multiplied matrices) at the front of the workspace and reuses it while `params` and `workspace` are the
 * same pointers as in the previous call.  Call this after changing the parameter buffer's CONTENTS in place
 * (transformers has no counterpart: its modules read their weights on every call). */
void flair_segformer_weights_changed(flair_segformer_t* h);
/* logits_quarter_nchw: fp32 (B, labels, H/4, W/4) = the library's `.logits`; logits_full_nchw: the same after
 * nn.functional.interpolate(size=(H, W), mode="bilinear", align_corners=False), what softmax / margin crop /
 * convert (compare.py:35, 69-82) need at tile resolution.  Either may be NULL, not both. */
int flair_segformer_forward(flair_segformer_t* h, const float* params, const float* x_nchw, float* logits_quarter_nchw,
                            float* logits_full_nchw, int B, int H, int W, void* workspace, size_t workspace_bytes,
                            void* stream);

/* Diagnostic tuning switch (kernel-variant A/B timing inside one process; keys are the FLAIR_* environment
 * variables DESIGN.md lists, the environment supplies the default).  Returns 0. */
int flair_tune_set(const char* key, int value);
/* Diagnostic: device buffer of [workgroups][8] uint64 that the next halo-GEMM launches fill with s_memtime stamps
 * (kernel start, main loop start, main loop end, accumulators staged, stored; + 100 MHz real-time start / end); NULL
 * switches it off.  Only the diagnostic build (FLAIR_STAMPS=1 python flair-1_amd/build.py) contains the stamps: the shipped
 * library returns -7 and its kernels execute none. */
int flair_debug_buffer(void* device_u64);

#ifdef __cplusplus
}
#endif
#endif /* FLAIR_HIP_H */
