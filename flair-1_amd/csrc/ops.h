// Host-callable launchers of the elementwise / reduction kernels (bn.hip, misc.hip, ce_head.hip).
#pragma once
#include "common.h"

namespace flair {

// ---- bn.hip
int bn_finalize(const float* partial, int nblk, int C, long count, const float* gamma, const float* beta,
                float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                float* mean_out, float* invstd_out, hipStream_t s);
int bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                   float* scale, float* shift, hipStream_t s);
int bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* res, const float* rscale,
           const float* rshift, void* out, long rows, int C, int relu, hipStream_t s);
int bn_bwd_blocks(long rows);
int bn_backward(int dtype, const void* dout, const void* out, const void* y, const float* mean, const float* invstd,
                const float* gamma, long rows, int C, float* partial, float* coef, float* dgamma, float* dbeta,
                int accumulate_param, void* dy, void* dres, int dres_accumulate, const float* mscale, const float* mshift,
                int pre_nblk /* > 0: partial already holds that many producer-side tile sums */,
                int premasked /* dout is dz already (ConvArgs::bnr_mask): no mask source needed; requires pre_nblk > 0 */, hipStream_t s);

int partial_rows_sum(const float* partial, int nblk, int ncols, float* out, hipStream_t s);
int bn_stats_partial(int dtype, const void* y, long rows, int C, float* partial, float* zeros_ones, hipStream_t s);
int colsum(int dtype, const void* x, long rows, int ld, int C, float* partial, float* out, hipStream_t s);

// ---- misc.hip
int maxpool3x3s2_fwd(int dtype, const void* in, void* out, unsigned char* idx, int N, int H, int W, int C, hipStream_t s);
// act = relu(y * scale + shift) and its 3x3 / stride-2 max pool (+ tap indices) in one pass (the stem in training mode)
int bn_act_maxpool3x3s2(int dtype, const void* y, const float* scale, const float* shift, void* act, void* out, unsigned char* idx, int N, int H,
                        int W, int C, hipStream_t s);
int maxpool3x3s2_bwd(int dtype, const void* dout, const unsigned char* idx, void* din, int accumulate, int N, int H,
                     int W, int C, hipStream_t s, const void* bnr_y = nullptr, const float* bnr_msc = nullptr,
                     const float* bnr_msh = nullptr, float* bnr_partial = nullptr);   // bnr_*: fused BN-backward reduction, [2][C][N*H] partials
int upcat_bwd(int dtype, const void* dcat, void* dx0, int dx0_accumulate, void* dskip, int dskip_accumulate, int N,
              int H, int W, int C0, int C1, hipStream_t s);
int nchw_f32_to_nhwc(int dtype, const float* in, void* out, int N, int C, int H, int W, int Cp, hipStream_t s);
int nhwc_to_nchw_f32(int dtype, const void* in, float* out, int N, int C, int H, int W, int Cp, float* accumulate_into,
                     hipStream_t s);
int pack_weight(int dtype, const float* w_oihw, void* dst, int Cout, int Cin, int R, int S, int Cin_p, int rows_pad,
                int Kpad, int transpose_flip, hipStream_t s);
struct PackDesc {
  long w_off;      // float offset of the OIHW master in the flat parameter buffer
  size_t dst_off;  // byte offset of the packed copy from the arena base
  int Cout, Cin, R, S, Cin_p, rows_pad, Kpad, tf;
  // tap subset of the packed copy: packed tap (ri, si), ri < Rc, si < Sc, is tap (r0 + ri*rstep, s0 + si*sstep) of the
  // full R x S pack (Rc = 0: all taps).  Used by the parity classes of a stride-2 data gradient.
  unsigned char r0, rstep, Rc, s0, sstep, Sc, pad_[2];
};
struct PackTable {
  static constexpr int MAX = 64;
  int n;
  PackDesc d[MAX];
};
int pack_weights_all(int dtype, const float* params, void* base, const PackTable& tb, hipStream_t s);
int sgd_step(float* params, const float* grads, long n, float lr, hipStream_t s);
int add_rowvec_nchw(float* x, const float* v, int N, int C, int H, int W, hipStream_t s);
int ew_add(int dtype, void* dst, const void* src, long n, hipStream_t s);
int fill_zero(void* p, size_t bytes, hipStream_t s);
int fill_f32(float* p, long n, float v, hipStream_t s);

// ---- ce_head.hip
struct CeArgs {
  const float* logits;        // NCHW fp32 [B][C][H][W]   (or null with logits_nhwc set)
  const void* logits_nhwc = nullptr;   // NHWC T [B*H*W][logits_ld]: the head convolution's own output layout
  int logits_dtype = 0, logits_ld = 0;
  const void* labels;         // label map, dtype per label_kind
  int label_kind;             // 0: uint8 [B][H][W]; 1: int32; 2: int64; 3: fp32 one-hot NCHW [B][C][H][W]
  const float* weight;        // optional class weights [C]
  int B, C, H, W;
  float* loss;                // out: scalar mean loss
  float* dlogits_nchw;        // optional out: fp32 NCHW gradient of the mean loss
  void* dlogits_nhwc;         // optional out: NHWC T gradient, row stride ld (>= C), padded columns zeroed
  int dlogits_dtype, dlogits_ld;
  unsigned char* preds_u8;    // optional out [B][H][W]
  long long* preds_i64;       // optional out [B][H][W]
  int* targets_i32;           // optional out [B][H][W]
  long long* confmat;         // optional in/out [C][C] += bincount(target*C + pred)
  float* workspace;           // >= ce_workspace_floats() floats
};
size_t ce_workspace_floats(int B, int H, int W);
int ce_head(const CeArgs& a, hipStream_t s);
int softmax_argmax(const float* logits, int B, int C, int H, int W, unsigned char* preds_u8, long long* preds_i64,
                   float* maxprob, hipStream_t s);
int softmax_argmax_nhwc(const void* logits, int dtype, int ld, long npix, int C, unsigned char* preds_u8, long long* preds_i64,
                        float* maxprob, hipStream_t s);
int confmat_update(const void* target, int target_kind, const void* pred, int pred_kind, long n, int C,
                   long long* confmat, hipStream_t s);
int jaccard_from_confmat(const long long* confmat, int C, float* per_class, float* weighted, float* macro, hipStream_t s);

// ---- feed.hip (rows either side of the hot path)
struct FeedArgs {
  static constexpr int MAXCH = 16;
  const unsigned char* img;     // [B][Cb][H][W] raster bands as stored
  const unsigned char* msk;     // optional [B][H][W] raw label raster (values 1..C as stored)
  const unsigned char* d4;      // optional [B] per-sample draw: bit0 V-flip, bit1 H-flip, bits 2-3 rot90 count
  float* out;                   // optional [B][Cout][H][W] fp32 NCHW (batch["img"])
  unsigned char* labels;        // optional [B][H][W] class index = argmax(batch["msk"], 1)
  int B, Cb, Cout, H, W;
  int mode;                     // 0 'without', 1 'scaling', 2 'custom'
  int num_classes;
  int band[MAXCH];              // 0-based band of output channel c (config "channels" minus one)
  double mean[MAXCH], stdv[MAXCH];
};
int feed_tiles(const FeedArgs& a, hipStream_t s);
int gather_tiles(const FeedArgs& a, const int* tiles, int Hr, int Wr, hipStream_t s);  // a.img: one (Cb, Hr, Wr) raster
int detect_stitch_preds(const unsigned char* preds, const float* maxprob, int B, int S, int margin, const int* tiles, float* out,
                        int Hr, int Wr, hipStream_t s);
int detect_convert(const float* logits, int B, int C, int S, int margin, int mode, void* out, const int* tiles, int Hr, int Wr,
                   hipStream_t s);
int confmat_masks(const unsigned char* truth, const unsigned char* pred, long n, int C, int truth_offset, long long* confmat,
                  hipStream_t s);

}  // namespace flair
