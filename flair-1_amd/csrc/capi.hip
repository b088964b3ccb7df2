// extern "C" surface of libflair_hip.so — declarations and reference citations in include/flair_hip.h.
#include <new>
#include <string.h>

#include "../../include/flair_hip.h"
#include "unet.h"
#include "segformer.h"

using namespace flair;

namespace flair { int conv_weight_rows_pad(int cout); }

struct flair_unet { UNet net; flair_unet(int a, int b, int c) : net(a, b, c) {} };
struct flair_segformer {
  SegFormer net;
  flair_segformer(int a, int b, const int* d, const int* h, const int* hd, const int* sr, int dh, int dt) : net(a, b, d, h, hd, sr, dh, dt) {}
};

extern "C" {

const char* flair_strerror(int code) {
  if (code == 0) return "ok";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  switch (code) {
    case -1: return "null or invalid handle";
    case -2: return "channel count / pointer alignment not supported by the kernel";
    case -3: return "output rows too short for whole 16-byte chunks (channel count / row stride)";
    case -4: return "fused upsample needs even extents";
    case -5: return "unsupported stride";
    case -6: return "requested fused epilogue / input transform is not available in the kernel this shape dispatches to";
    case -7: return "diagnostic feature: rebuild with FLAIR_STAMPS=1";
    case -10: return "Wrong input shape: height and width must be divisible by 32";
    case -11: return "call order: backward / stage call without the matching forward on this workspace";
    case -12: return "gradient buffer already initialised";
    case -13: return "internal side stream: event record / wait failed";
    case -100: return "workspace too small";
    default: return "invalid argument";
  }
}

int flair_version(void) { return 1; }

int flair_unet_create(flair_unet_t** out, int in_channels, int classes, int dtype) {
  if (!out || in_channels < 1 || in_channels > 8 || classes < 1 || classes > 32 || (dtype != 0 && dtype != 1)) return -1;
  *out = new (std::nothrow) flair_unet(in_channels, classes, dtype);
  return *out ? 0 : -1;
}
void flair_unet_destroy(flair_unet_t* h) { delete h; }
int64_t flair_unet_param_count(const flair_unet_t* h) { return h ? h->net.n_params : -1; }
int64_t flair_unet_buffer_count(const flair_unet_t* h) { return h ? h->net.n_buffers : -1; }
int flair_unet_num_tensors(const flair_unet_t* h) { return h ? (int)h->net.tensors.size() : -1; }
int flair_unet_tensor_info(const flair_unet_t* h, int i, char* name, int name_cap, int64_t shape[4], int* ndim,
                           int64_t* offset, int* kind, int* stage) {
  if (!h || i < 0 || i >= (int)h->net.tensors.size()) return -1;
  const TensorInfo& t = h->net.tensors[i];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  for (int d = 0; d < 4; ++d) shape[d] = t.shape[d];
  *ndim = t.ndim; *offset = t.offset; *kind = t.kind; *stage = t.stage;
  return 0;
}
int flair_unet_stage_range(const flair_unet_t* h, int stage, int64_t* begin, int64_t* end) {
  if (!h || stage < 0 || stage > 6) return -1;
  *begin = h->net.stage_begin[stage]; *end = h->net.stage_begin[stage + 1];
  return 0;
}
int64_t flair_unet_workspace_bytes(flair_unet_t* h, int B, int H, int W, int training) {
  if (!h || B < 1 || (H % 32) || (W % 32)) return -1;
  return (int64_t)h->net.workspace_bytes(B, H, W, training);
}
int flair_unet_head_ld(const flair_unet_t* h) { return h ? h->net.convs.back().Cout_p : -1; }
int flair_unet_want_preds(flair_unet_t* h, uint8_t* preds_u8, float* maxprob_f32) {
  if (!h || (maxprob_f32 && !preds_u8)) return -1;
  h->net.want_preds(preds_u8, maxprob_f32);
  return 0;
}
int flair_unet_reuse_constants(flair_unet_t* h, int on) {
  if (!h) return -1;
  h->net.reuse_constants(on != 0);
  return 0;
}

int flair_unet_forward(flair_unet_t* h, const float* params, float* buffers, const float* x, float* logits, int B, int H,
                       int W, int training, void* ws, size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.forward(params, buffers, x, logits, B, H, W, training, ws, wsb, (hipStream_t)stream);
}
int flair_unet_backward(flair_unet_t* h, const float* params, const float* dl_nchw, const void* dl_nhwc, float* grads,
                        void* ws, size_t wsb, void* stream, void* const* stage_events) {
  if (!h || (!dl_nchw == !dl_nhwc)) return -1;
  return h->net.backward(params, dl_nchw, dl_nhwc, grads, ws, wsb, (hipStream_t)stream, stage_events);
}
int flair_unet_encoder_forward(flair_unet_t* h, const float* params, float* buffers, const float* x,
                               float* const feats[5], int B, int H, int W, int training, void* ws, size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.encoder_forward(params, buffers, x, feats, B, H, W, training, ws, wsb, (hipStream_t)stream);
}
int flair_unet_decoder_forward(flair_unet_t* h, const float* params, float* buffers, const float* const feats[5],
                               float* out, int B, int H, int W, int training, void* ws, size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.decoder_forward(params, buffers, feats, out, B, H, W, training, ws, wsb, (hipStream_t)stream);
}
int flair_unet_head_forward(flair_unet_t* h, const float* params, const float* x, float* logits, int B, int H, int W,
                            int training, void* ws, size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.head_forward(params, x, logits, B, H, W, training, ws, wsb, (hipStream_t)stream);
}
int flair_unet_head_backward(flair_unet_t* h, const float* params, const float* dl, float* dx, float* grads, void* ws,
                             size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.head_backward(params, dl, dx, grads, ws, wsb, (hipStream_t)stream);
}
int flair_unet_decoder_backward(flair_unet_t* h, const float* params, const float* dout, float* const dfeats[5],
                                float* grads, void* ws, size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.decoder_backward(params, dout, dfeats, grads, ws, wsb, (hipStream_t)stream);
}
int flair_unet_encoder_backward(flair_unet_t* h, const float* params, const float* const dfeats[5], float* grads,
                                void* ws, size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.encoder_backward(params, dfeats, grads, ws, wsb, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ head
size_t flair_ce_workspace_bytes(int B, int H, int W) { return ce_workspace_floats(B, H, W) * sizeof(float); }

int flair_ce_head(const float* logits, const void* labels, int label_kind, const float* weight, int B, int C, int H,
                  int W, float* loss, float* dl_nchw, void* dl_nhwc, int dl_dtype, int dl_ld, uint8_t* preds_u8,
                  int64_t* preds_i64, int32_t* targets_i32, int64_t* confmat, void* workspace, void* stream) {
  if (!logits || !labels || !loss || !workspace || label_kind < 0 || label_kind > 3) return -1;
  CeArgs a;
  a.logits = logits; a.labels = labels; a.label_kind = label_kind; a.weight = weight;
  a.B = B; a.C = C; a.H = H; a.W = W; a.loss = loss; a.dlogits_nchw = dl_nchw; a.dlogits_nhwc = dl_nhwc;
  a.dlogits_dtype = dl_dtype; a.dlogits_ld = dl_ld; a.preds_u8 = preds_u8; a.preds_i64 = (long long*)preds_i64;
  a.targets_i32 = targets_i32; a.confmat = (long long*)confmat; a.workspace = (float*)workspace;
  return ce_head(a, (hipStream_t)stream);
}
const void* flair_unet_logits_nhwc(const flair_unet_t* h) { return h ? h->net.logits_nhwc() : nullptr; }

int flair_ce_head_nhwc(const void* logits_nhwc, int dtype, int ld, const void* labels, int label_kind, const float* weight,
                       int B, int C, int H, int W, float* loss, void* dl_nhwc, uint8_t* preds_u8, int32_t* targets_i32,
                       int64_t* confmat, void* workspace, void* stream) {
  if (!logits_nhwc || !labels || !loss || !workspace || label_kind < 0 || label_kind > 3) return -1;
  if (dtype != DT_F32 && dtype != DT_BF16) return -2;
  CeArgs a;
  a.logits = nullptr; a.logits_nhwc = logits_nhwc; a.logits_dtype = dtype; a.logits_ld = ld;
  a.labels = labels; a.label_kind = label_kind; a.weight = weight;
  a.B = B; a.C = C; a.H = H; a.W = W; a.loss = loss; a.dlogits_nchw = nullptr; a.dlogits_nhwc = dl_nhwc;
  a.dlogits_dtype = dtype; a.dlogits_ld = ld; a.preds_u8 = preds_u8; a.preds_i64 = nullptr;
  a.targets_i32 = targets_i32; a.confmat = (long long*)confmat; a.workspace = (float*)workspace;
  return ce_head(a, (hipStream_t)stream);
}
int flair_softmax_argmax_nhwc(const void* logits_nhwc, int dtype, int ld, int B, int C, int H, int W, uint8_t* preds_u8,
                              int64_t* preds_i64, float* maxprob, void* stream) {
  if (!logits_nhwc) return -1;
  if (dtype != DT_F32 && dtype != DT_BF16) return -2;
  return softmax_argmax_nhwc(logits_nhwc, dtype, ld, (long)B * H * W, C, preds_u8, (long long*)preds_i64, maxprob, (hipStream_t)stream);
}
int flair_softmax_argmax(const float* logits, int B, int C, int H, int W, uint8_t* preds_u8, int64_t* preds_i64,
                         float* maxprob, void* stream) {
  if (!logits) return -1;
  return softmax_argmax(logits, B, C, H, W, preds_u8, (long long*)preds_i64, maxprob, (hipStream_t)stream);
}
int flair_confmat_update(const void* target, int tk, const void* pred, int pk, int64_t n, int C, int64_t* confmat, void* stream) {
  if (!target || !pred || !confmat) return -1;
  return confmat_update(target, tk, pred, pk, n, C, (long long*)confmat, (hipStream_t)stream);
}
int flair_jaccard(const int64_t* confmat, int C, float* per_class, float* weighted, float* macro, void* stream) {
  if (!confmat) return -1;
  return jaccard_from_confmat((const long long*)confmat, C, per_class, weighted, macro, (hipStream_t)stream);
}
int flair_feed_tiles(const uint8_t* img_u8, const uint8_t* msk_raw, const uint8_t* d4_flags, int B, int bands, int H, int W,
                     const int* channels, int n_channels, int norm_type, const double* means, const double* stds,
                     int num_classes, float* img_out, uint8_t* labels_out, void* stream) {
  if (!img_out && !labels_out) return -1;
  if (n_channels < 0 || n_channels > FeedArgs::MAXCH) return -2;
  if (img_out && (!img_u8 || !channels || n_channels < 1)) return -1;
  if (norm_type == 2 && (!means || !stds)) return -1;
  FeedArgs a{};
  a.img = img_u8; a.msk = msk_raw; a.d4 = d4_flags; a.out = img_out; a.labels = labels_out;
  a.B = B; a.Cb = bands; a.Cout = img_out ? n_channels : 0; a.H = H; a.W = W; a.mode = norm_type; a.num_classes = num_classes;
  for (int c = 0; c < a.Cout; ++c) {
    a.band[c] = channels[c] - 1;  // config "channels" start at 1 (rasterio band numbering)
    a.mean[c] = norm_type == 2 ? means[c] : 0.0;
    a.stdv[c] = norm_type == 2 ? stds[c] : 1.0;
  }
  return feed_tiles(a, (hipStream_t)stream);
}
int flair_detect_convert(const float* logits_nchw, int B, int C, int S, int margin, int output_type, void* out, void* stream) {
  if (!logits_nchw || !out) return -1;
  return detect_convert(logits_nchw, B, C, S, margin, output_type, out, nullptr, 0, 0, (hipStream_t)stream);
}
int flair_detect_stitch(const float* logits_nchw, int B, int C, int S, int margin, int output_type, const int32_t* tiles,
                        void* raster_out, int raster_h, int raster_w, void* stream) {
  if (!logits_nchw || !raster_out || !tiles) return -1;
  return detect_convert(logits_nchw, B, C, S, margin, output_type, raster_out, tiles, raster_h, raster_w, (hipStream_t)stream);
}
int flair_detect_stitch_preds(const uint8_t* preds_u8, const float* maxprob_f32, int B, int S, int margin, const int32_t* tiles,
                              float* raster_out, int raster_h, int raster_w, void* stream) {
  if (!preds_u8 || !maxprob_f32 || !raster_out || !tiles) return -1;
  return detect_stitch_preds(preds_u8, maxprob_f32, B, S, margin, tiles, raster_out, raster_h, raster_w, (hipStream_t)stream);
}
int flair_gather_tiles(const uint8_t* raster_u8, int bands, int raster_h, int raster_w, const int32_t* tiles, int B, int S,
                       const int* channels, int n_channels, int norm_type, const double* means, const double* stds,
                       float* img_out, void* stream) {
  if (!raster_u8 || !tiles || !channels || !img_out) return -1;
  if (n_channels < 1 || n_channels > FeedArgs::MAXCH) return -2;
  if (norm_type == 2 && (!means || !stds)) return -1;
  FeedArgs a{};
  a.img = raster_u8; a.out = img_out; a.B = B; a.Cb = bands; a.Cout = n_channels; a.H = S; a.W = S; a.mode = norm_type;
  for (int c = 0; c < n_channels; ++c) {
    a.band[c] = channels[c] - 1;
    a.mean[c] = norm_type == 2 ? means[c] : 0.0;
    a.stdv[c] = norm_type == 2 ? stds[c] : 1.0;
  }
  return gather_tiles(a, tiles, raster_h, raster_w, (hipStream_t)stream);
}
int flair_confmat_masks(const uint8_t* truth_raw, const uint8_t* pred, int64_t n, int C, int truth_offset, int64_t* confmat,
                        void* stream) {
  if (!truth_raw || !pred || !confmat) return -1;
  return confmat_masks(truth_raw, pred, n, C, truth_offset, (long long*)confmat, (hipStream_t)stream);
}
int flair_sgd_step(float* params, const float* grads, int64_t n, float lr, void* stream) {
  if (!params || !grads) return -1;
  return sgd_step(params, grads, n, lr, (hipStream_t)stream);
}
int flair_add_rowvec_nchw(float* x, const float* v, int N, int C, int H, int W, void* stream) {
  if (!x || !v) return -1;
  return add_rowvec_nchw(x, v, N, C, H, W, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ operators
struct OpArena {
  unsigned char* base; size_t cap, top; bool bad;
  OpArena(void* p, size_t c) : base((unsigned char*)p), cap(c), top(0), bad(false) {}
  void* get(size_t bytes) {
    size_t off = top;
    top = (size_t)round_up((long)(top + bytes), 256);
    if (top > cap) { bad = true; return base; }
    return base + off;
  }
};

static void conv_geom(int dtype, int C0, int C1, int Cout, int R, int& Cin, int& Kg, int& Kpad, int& rows_f) {
  Cin = C0 + C1;
  const int kstep = dtype == DT_F32 ? 32 : 64;
  Kg = R * R * Cin;
  Kpad = (int)round_up(Kg, kstep);
  rows_f = conv_weight_rows_pad(Cout);
}

size_t flair_conv2d_workspace_bytes(int dtype, int N, int H, int W, int C0, int C1, int up0, int Cout, int R, int stride,
                                    int pad) {
  int Cin, Kg, Kpad, rows_f;
  conv_geom(dtype, C0, C1, Cout, R, Cin, Kg, Kpad, rows_f);
  const int Hin = up0 ? 2 * H : H, Win = up0 ? 2 * W : W;
  const int Ho = (Hin + 2 * pad - R) / stride + 1, Wo = (Win + 2 * pad - R) / stride + 1;
  const size_t es = dtype_size(dtype);
  size_t b = (size_t)rows_f * Kpad * es + 4096;
  b += ((size_t)N * Ho * Wo / 128 + 2) * 2 * Cout * 4 + 4096;            // stats partial
  int CoutP = (int)round_up(Cout, 8);
  const int kstep = dtype == DT_F32 ? 32 : 64;
  b += (size_t)conv_weight_rows_pad(Cin) * round_up((long)R * R * CoutP, kstep) * es + 4096;  // dgrad pack
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.C0 = C0; w.C1 = C1; w.N = N; w.Hin = Hin; w.Win = Win; w.Hout = Ho; w.Wout = Wo; w.R = R; w.S = R; w.Cout = Cout;
  w.stride = stride; w.pad = pad; w.up0 = up0; w.dy_ld = CoutP;
  b += wgrad_workspace_bytes(dtype, w) + 4096;
  return b;
}

int flair_conv2d_forward(int dtype, const void* x0, const void* x1, int N, int H, int W, int C0, int C1, int up0,
                         const float* w_oihw, const float* bias, int Cout, int R, int stride, int pad, void* y_nhwc,
                         float* y_nchw, float* stats, void* workspace, size_t wsb, void* stream) {
  if (!x0 || !w_oihw || !workspace) return -1;
  hipStream_t s = (hipStream_t)stream;
  int Cin, Kg, Kpad, rows_f;
  conv_geom(dtype, C0, x1 ? C1 : 0, Cout, R, Cin, Kg, Kpad, rows_f);
  OpArena ar(workspace, wsb);
  void* wp = ar.get((size_t)rows_f * Kpad * dtype_size(dtype));
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.src0 = x0; a.src1 = x1; a.C0 = C0; a.C1 = x1 ? C1 : 0; a.up0 = up0; a.N = N;
  a.Hin = up0 ? 2 * H : H; a.Win = up0 ? 2 * W : W;
  a.Hout = (a.Hin + 2 * pad - R) / stride + 1; a.Wout = (a.Win + 2 * pad - R) / stride + 1;
  a.R = R; a.S = R; a.out_mul = stride; a.pad = pad; a.in_div = 1; a.Cout = Cout; a.Kg = Kg; a.Kpad = Kpad; a.w = wp;
  a.bias = bias; a.out = y_nhwc; a.out_ld = Cout; a.out_nchw = y_nchw;
  const int nblk = conv_grid_rows(dtype, a);
  float* partial = stats ? (float*)ar.get((size_t)nblk * 2 * Cout * 4) : nullptr;
  a.stats = partial;
  if (ar.bad) return -100;
  int rc = pack_weight(dtype, w_oihw, wp, Cout, Cin, R, R, Cin, rows_f, Kpad, 0, s);
  if (rc) return rc;
  rc = launch_conv(dtype, a, s);
  if (rc) return rc;
  if (stats) rc = partial_rows_sum(partial, nblk, 2 * Cout, stats, s);
  return rc;
}

int flair_conv2d_backward(int dtype, const void* x0, int N, int H, int W, int Cin, const float* w_oihw, int Cout, int R,
                          int stride, int pad, const void* dy, void* dx, float* dw, void* workspace, size_t wsb, void* stream) {
  if (!x0 || !w_oihw || !dy || !workspace) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - R) / stride + 1;
  OpArena ar(workspace, wsb);
  int rc = 0;
  if (dx) {
    const int kstep = dtype == DT_F32 ? 32 : 64;
    const int Kgd = R * R * Cout, Kpad_d = (int)round_up(Kgd, kstep), rows_d = conv_weight_rows_pad(Cin);
    void* wd = ar.get((size_t)rows_d * Kpad_d * dtype_size(dtype));
    if (ar.bad) return -100;
    rc = pack_weight(dtype, w_oihw, wd, Cout, Cin, R, R, Cout, rows_d, Kpad_d, 1, s);
    if (rc) return rc;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = dy; a.C0 = Cout; a.N = N; a.Hin = Ho; a.Win = Wo; a.Hout = H; a.Wout = W; a.R = R; a.S = R;
    a.out_mul = 1; a.pad = R - 1 - pad; a.in_div = stride; a.Cout = Cin; a.Kg = Kgd; a.Kpad = Kpad_d; a.w = wd;
    a.out = dx; a.out_ld = Cin;
    rc = launch_conv(dtype, a, s);
    if (rc) return rc;
  }
  if (dw) {
    WgradArgs w;
    memset(&w, 0, sizeof(w));
    w.x0 = x0; w.C0 = Cin; w.N = N; w.Hin = H; w.Win = W; w.Hout = Ho; w.Wout = Wo; w.R = R; w.S = R;
    w.stride = stride; w.pad = pad; w.dy = dy; w.dy_ld = Cout; w.Cout = Cout; w.dw = dw; w.Cin_real = Cin;
    w.partial = (float*)ar.get(wgrad_workspace_bytes(dtype, w));
    if (ar.bad) return -100;
    rc = launch_wgrad(dtype, w, s);
  }
  return rc;
}

int flair_bn_relu_forward(int dtype, const void* y, int64_t rows, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int training, const void* residual, int relu,
                          void* out, float* save_mean, float* save_invstd, void* workspace, size_t wsb, void* stream) {
  if (!y || !out || !workspace) return -1;
  hipStream_t s = (hipStream_t)stream;
  OpArena ar(workspace, wsb);
  float* scale = (float*)ar.get(C * 4);
  float* shift = (float*)ar.get(C * 4);
  int rc;
  if (training) {
    const int nblk = bn_bwd_blocks(rows);
    float* partial = (float*)ar.get((size_t)nblk * 2 * C * 4);
    float* zo = (float*)ar.get(2 * C * 4);
    if (ar.bad) return -100;
    rc = bn_stats_partial(dtype, y, rows, C, partial, zo, s);
    if (rc) return rc;
    rc = bn_finalize(partial, nblk, C, rows, gamma, beta, running_mean, running_var, 0.1f, 1e-5f, scale, shift,
                     save_mean, save_invstd, s);
  } else {
    if (ar.bad) return -100;
    rc = bn_eval_coeffs(C, gamma, beta, running_mean, running_var, 1e-5f, scale, shift, s);
  }
  if (rc) return rc;
  return bn_act(dtype, y, scale, shift, residual, nullptr, nullptr, out, rows, C, relu, s);
}

int flair_bn_relu_backward(int dtype, const void* dout, const void* out, const void* y, int64_t rows, int C,
                           const float* gamma, const float* save_mean, const float* save_invstd, int relu, void* dy,
                           void* dres, float* dgamma, float* dbeta, void* workspace, size_t wsb, void* stream) {
  if (!dout || !y || !dy || !workspace) return -1;
  OpArena ar(workspace, wsb);
  float* partial = (float*)ar.get((size_t)bn_bwd_blocks(rows) * 2 * C * 4);
  float* coef = (float*)ar.get(3 * C * 4);
  if (ar.bad) return -100;
  return bn_backward(dtype, dout, relu ? out : nullptr, y, save_mean, save_invstd, gamma, rows, C, partial, coef, dgamma,
                     dbeta, 0, dy, dres, 0, nullptr, nullptr, 0, 0, (hipStream_t)stream);
}

int flair_maxpool_forward(int dtype, const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
  return maxpool3x3s2_fwd(dtype, x, y, idx, N, H, W, C, (hipStream_t)stream);
}
int flair_maxpool_backward(int dtype, const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, void* stream) {
  return maxpool3x3s2_bwd(dtype, dy, idx, dx, 0, N, H, W, C, (hipStream_t)stream);
}
int flair_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, int Cpad, void* stream) {
  return nchw_f32_to_nhwc(dtype, x, y, N, C, H, W, Cpad, (hipStream_t)stream);
}
int flair_nhwc_to_nchw(int dtype, const void* x, float* y, int N, int C, int H, int W, int Cpad, void* stream) {
  return nhwc_to_nchw_f32(dtype, x, y, N, C, H, W, Cpad, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ SegFormer (zone_detect, HuggingFace provider)
int flair_segformer_create(flair_segformer_t** out, int in_channels, int num_labels, const int depths[4], const int hidden_sizes[4],
                           const int num_heads[4], const int sr_ratios[4], int decoder_hidden_size, int dtype) {
  if (!out || !depths || !hidden_sizes || !num_heads || !sr_ratios || in_channels < 1 || num_labels < 1) return -1;
  if (dtype != DT_F32 && dtype != DT_BF16) return -2;
  for (int i = 0; i < 4; ++i)
    if (depths[i] < 1 || hidden_sizes[i] < 64 || (hidden_sizes[i] % 64) || num_heads[i] < 1 || hidden_sizes[i] != 64 * num_heads[i] ||
        sr_ratios[i] != (8 >> i))
      return -2;   // MiT-B1 .. B5 geometry: heads of 64 channels, reduction ratios 8 / 4 / 2 / 1
  if (decoder_hidden_size < 64 || (decoder_hidden_size % 64)) return -2;
  flair_segformer* h = new (std::nothrow) flair_segformer(in_channels, num_labels, depths, hidden_sizes, num_heads, sr_ratios,
                                                          decoder_hidden_size, dtype);
  if (!h) return -100;
  *out = h;
  return 0;
}
void flair_segformer_destroy(flair_segformer_t* h) { delete h; }
int64_t flair_segformer_param_count(const flair_segformer_t* h) { return h ? h->net.n_params : -1; }
int flair_segformer_num_tensors(const flair_segformer_t* h) { return h ? (int)h->net.tensors.size() : -1; }
int flair_segformer_tensor_info(const flair_segformer_t* h, int i, char* name, int name_cap, int64_t shape[4], int* ndim, int64_t* offset,
                                int* kind) {
  if (!h || i < 0 || i >= (int)h->net.tensors.size() || !name || name_cap < 2) return -1;
  const SfTensor& t = h->net.tensors[i];
  strncpy(name, t.name.c_str(), name_cap - 1);
  name[name_cap - 1] = 0;
  for (int d = 0; d < 4; ++d) shape[d] = t.shape[d];
  *ndim = t.ndim; *offset = t.offset; *kind = t.kind;
  return 0;
}
int64_t flair_segformer_workspace_bytes(flair_segformer_t* h, int B, int H, int W) {
  if (!h || B < 1 || !h->net.shape_ok(H, W)) return -1;
  return (int64_t)h->net.workspace_bytes(B, H, W);
}
void flair_segformer_weights_changed(flair_segformer_t* h) {
  if (h) h->net.weights_changed();
}
int flair_segformer_forward(flair_segformer_t* h, const float* params, const float* x_nchw, float* logits_quarter_nchw,
                            float* logits_full_nchw, int B, int H, int W, void* ws, size_t wsb, void* stream) {
  if (!h) return -1;
  return h->net.forward(params, x_nchw, logits_quarter_nchw, logits_full_nchw, B, H, W, ws, wsb, (hipStream_t)stream);
}

}  // extern "C"
