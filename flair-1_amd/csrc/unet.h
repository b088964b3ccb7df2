// Native executor for the smp-0.3.3 Unet(resnet34) graph the reference instantiates at
// /root/reference/src/flair/model.py:37-41 and runs at model.py:57-64.
// Owns the layer table, the flat parameter layout, the workspace arena and
// launches every kernel of forward / backward back-to-back on the caller's HIP stream.
#pragma once
#include <string>
#include <vector>

#include "ops.h"

namespace flair {

struct Act {
  void* p = nullptr;
  int N = 0, H = 0, W = 0, C = 0;  // C = stored (padded) channel count
  // "lazy" activation: p is the producing unit's PRE-BatchNorm tensor and consumers apply relu(x*lz_scale + lz_shift)
  // while they stage it (ConvArgs::in_scale) — the activation itself is never written
  const float* lz_scale = nullptr;
  const float* lz_shift = nullptr;
  long rows() const { return (long)N * H * W; }
  long elems() const { return rows() * C; }
};

struct ConvDesc {
  std::string name;
  int Cin, Cout, R, S, stride, pad;
  int Cin_p, Cout_p;      // stored channel counts of the NHWC input / output tensors
  bool bias;
  long w_off = -1, b_off = -1;  // float offsets in the flat parameter buffer
  // packed copies (byte offsets into the workspace weight arena)
  size_t wf = 0, wd = 0;
  int Kg, Kpad, rows_f;   // forward pack
  int Kgd, Kpad_d, rows_d;  // data-gradient pack (rows = Cin_p, K = R*S*Cout_p)
  // 3x3 stride-2 layers: the data gradient splits by output parity (py, px) into four stride-1 convolutions over dY
  // with 1, 2, 2 and 4 of the nine taps (class = 2*py + px) — a quarter of the gather-form kernel's MFMA work
  size_t wd_cls[4] = {0, 0, 0, 0};
  int Kg_cls[4] = {0, 0, 0, 0}, Kpad_cls[4] = {0, 0, 0, 0};
  bool parity_dgrad() const { return stride == 2 && R == 3 && S == 3 && pad == 1; }
};

struct BnDesc {
  std::string name;
  int C;
  long g_off, b_off;      // flat parameter buffer
  long rm_off, rv_off;    // flat running-stat buffer
};

struct TensorInfo {
  std::string name;
  int ndim;
  long shape[4];
  long offset;
  int kind;  // 0 parameter (flat param buffer), 1 running statistic (flat buffer)
  int stage; // 0 stem, 1..4 encoder layers, 5 decoder, 6 head (gradient bucket id)
};

struct Unit {  // conv -> BN -> (+ residual) -> ReLU
  int conv = -1, bn = -1;
  Act in0, in1;
  bool up0 = false;
  Act y, out;
  float *scale = nullptr, *shift = nullptr, *mean = nullptr, *invstd = nullptr;
  bool relu = true;
  int res_unit = -1;   // downsample unit whose BN output is the residual
  Act res;             // identity residual
  // backward: set when the one consumer of `out` already reduced sum(dz), sum(dz*y) per pixel tile in the epilogue of
  // its data-gradient kernel (ConvArgs::bnr_*); consumed (and cleared) by unit_backward
  float* bnr_partial = nullptr;
  int bnr_nblk = 0;
  // backward: that consumer also stored the gradient MASKED by this unit's ReLU (ConvArgs::bnr_mask): the gradient buffer of
  // `out` holds dz, which is what the identity branch of a BasicBlock receives and what the fused BatchNorm-backward apply of
  // the unit's own data gradient (ConvArgs::ap_*) expects.  Consumed (and cleared) by unit_backward
  bool bnr_masked = false;
};

class UNet {
 public:
  UNet(int in_channels, int classes, int dtype);
  int in_channels, classes, dtype;
  std::vector<ConvDesc> convs;
  std::vector<BnDesc> bns;
  std::vector<TensorInfo> tensors;
  long n_params = 0, n_buffers = 0;
  long stage_begin[8];  // flat-parameter offset where each stage starts (stage_begin[7] = n_params)

  size_t workspace_bytes(int B, int H, int W, int training) const;   // side-effect free

  // stage entry points; features cross the boundary as NCHW fp32 only in the split path
  int forward(const float* params, float* buffers, const float* x_nchw, float* logits_nchw, int B, int H, int W,
              int training, void* ws, size_t ws_bytes, hipStream_t s);
  // stage_events: optional 7 hipEvent_t recorded on `s` as soon as the gradients of stage k
  // (6 head, 5 decoder, 4..1 encoder layers, 0 stem) are complete — lets the host overlap the
  // bucketed RCCL all-reduce with the rest of the backward pass.
  int backward(const float* params, const float* dlogits_nchw, const void* dlogits_nhwc, float* grads, void* ws,
               size_t ws_bytes, hipStream_t s, void* const* stage_events = nullptr);

  int encoder_forward(const float* params, float* buffers, const float* x_nchw, float* const feats_nchw[5], int B,
                      int H, int W, int training, void* ws, size_t ws_bytes, hipStream_t s);
  int decoder_forward(const float* params, float* buffers, const float* const feats_nchw[5], float* out_nchw, int B,
                      int H, int W, int training, void* ws, size_t ws_bytes, hipStream_t s);
  int head_forward(const float* params, const float* x_nchw, float* logits_nchw, int B, int H, int W, int training,
                   void* ws, size_t ws_bytes, hipStream_t s);
  int head_backward(const float* params, const float* dlogits_nchw, float* dx_nchw, float* grads, void* ws,
                    size_t ws_bytes, hipStream_t s);
  int decoder_backward(const float* params, const float* dout_nchw, float* const dfeats_nchw[5], float* grads,
                       void* ws, size_t ws_bytes, hipStream_t s);
  int encoder_backward(const float* params, const float* const dfeats_nchw[5], float* grads, void* ws,
                       size_t ws_bytes, hipStream_t s);

  // Inference with constant weights: when the caller vouches that parameters and buffers are unchanged since the previous
  // eval-mode forward on this handle, the next eval-mode forward with the same arena and shape skips the weight pack and
  // the 46 BatchNorm-coefficient launches (both still sit in the arena at the same offsets).  One-shot: cleared by forward.
  void reuse_constants(bool on) { reuse_req_ = on; }
  // every entry point other than forward() lays the arena out differently (the split path re-imports five feature tensors
  // in front of the decoder units) or overwrites it: the constants of the last eval forward are gone, and so are pending
  // one-shot requests
  void invalidate_reuse() { last_valid_ = false; reuse_req_ = false; preds_req_ = nullptr; maxprob_req_ = nullptr; }
  // the next forward without fp32 logits writes argmax predictions [B][H][W] uint8 here (one-shot, like reuse_constants)
  void want_preds(unsigned char* p, float* maxprob) { preds_req_ = p; maxprob_req_ = maxprob; }
  void* last_dlogits_nhwc() const { return dl_nhwc_; }
  const void* logits_nhwc() const { return logits_nhwc_; }
  int head_ld() const { return convs.back().Cout_p; }

 private:
  // ---- arena
  unsigned char* base_ = nullptr;
  size_t cap_ = 0, top_ = 0;
  bool dry_ = false;
  int err_ = 0;
  hipStream_t s_ = nullptr;
  void* alloc(size_t bytes);
  Act alloc_act(int N, int H, int W, int C);
  float* alloc_f(long n) { return (float*)alloc((size_t)n * 4); }

  // ---- per-call state
  const float* params_ = nullptr;
  float* buffers_ = nullptr;
  float* grads_ = nullptr;
  int B_ = 0, H_ = 0, W_ = 0, training_ = 0;
  std::vector<Unit> units_;
  Act xin_, f_[6], dec_in_[6], pool_, dec_out_;
  unsigned char* pool_idx_ = nullptr;
  int enc_units_end_ = 0, dec_units_begin_ = 0;
  size_t fwd_top_ = 0;      // arena top after forward (backward scratch starts here)
  bool packed_d_ = false;
  bool reuse_req_ = false, reuse_ = false, last_valid_ = false;
  const void* last_ws_ = nullptr;
  int last_B_ = 0, last_H_ = 0, last_W_ = 0;
  bool lazy_ok_ = false;
  int lazy_max_c_ = 16;    // widest unit that hands out a lazy activation (FLAIR_LAZY_BN=2: 32)
  // weight-gradient kernels run on an internal side stream, forked from / joined to the caller's stream with events, so
  // that their tails overlap the BatchNorm / data-gradient chain of the following units (FLAIR_WGRAD_STREAM=0: off)
  hipStream_t side_ = nullptr;
  hipStream_t note_ = nullptr;   // carries the listened-for stage events: waits for s_ and side_, blocks neither
  std::vector<hipEvent_t> fork_ev_;
  hipEvent_t join_ev_ = nullptr;
  size_t fork_next_ = 0;
  bool side_pending_ = false;
  unsigned char* preds_req_ = nullptr;
  float* maxprob_req_ = nullptr;
  bool side_init();
  hipStream_t wgrad_stream();
  int side_cus(int unit) const;   // forks the side stream behind everything queued on s_ so far
  void side_join();
 public:
  ~UNet();
 private:   // whole-model training forward only: small-channel decoder units hand out lazy activations
  void* dl_nhwc_ = nullptr;
  void* logits_nhwc_ = nullptr;   // head output kept in NHWC T when forward() was called without an NCHW logits buffer
  void* const* stage_events_ = nullptr;
  void stage_done(int stage);

  struct GradBuf { void* act; void* g; bool init; };
  // unit whose ReLU output is `a`, if `a` has exactly one consumer and the unit's mask comes from y; else -1
  int sole_producer(const Act& a) const;
  int residual_producer(const Act& a) const;
  void attach_bn_reduce(ConvArgs& a, const Act& target);
  std::vector<GradBuf> gbufs_;
  // set by the BasicBlock walk just before the backward of a block's conv1: its data gradient adds this tensor (the masked
  // gradient of the block's output = the identity branch's share) instead of reading a copy of it from its own output buffer
  const void* acc_src_next_ = nullptr;
  bool bwd_fuse() const;   // FLAIR_BWD_FUSE (default on): masked gradients + BatchNorm-backward apply inside the halo-GEMM data gradients
  void* grad_of(const Act& a, bool* accumulate);
  void* grad_peek(const Act& a);

  size_t plan_bytes(int B, int H, int W, int training);   // the dry run itself (clobbers per-call state)
  void build_table();
  int add_conv(const std::string& name, int cin, int cout, int k, int stride, int pad, bool bias, int stage);
  int add_bn(const std::string& name, int c, int stage);
  void begin(void* ws, size_t ws_bytes, hipStream_t s, bool dry);
  void pack_forward_weights();
  void pack_dgrad_weights();
  int run_unit(int conv, int bn, const Act& in0, const Act& in1, bool up0, bool relu, int res_unit, const Act& res,
               bool materialize, int lazy_cons = -1, bool lazy_up0 = false, const Act* lazy_skip = nullptr);
  bool lazy_into_hg(const Act& y, int cons, bool up0, const Act& skip) const;
  void encoder_fwd_impl(const float* x_nchw);
  void decoder_fwd_impl();
  void head_fwd_impl(float* logits_nchw);
  void unit_backward(int ui, const void* dout, void* dres, bool dres_acc, bool need_dgrad, void* dx_override,
                     bool upcat = false);
  void head_bwd_impl(const void* dl);
  void decoder_bwd_impl();
  void encoder_bwd_impl();
  void fwd_common_begin(const float* params, float* buffers, int B, int H, int W, int training);
};

}  // namespace flair
