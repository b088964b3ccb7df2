// Native inference executor for transformers' SegformerForSemanticSegmentation (MiT encoder + all-MLP decode head), the model
// BASELINE config 5 names for zone_detect: /root/reference/src/zone_detect/model.py:42-50 builds it with
// AutoModelForSemanticSegmentation, compare.py:31-36 runs `model(imgs).logits`.  Eval mode only (zone_detect never trains).
// The algorithm is the library's published one (modeling_segformer.py of transformers 5.15: overlapping patch embeddings,
// efficient self-attention with sequence reduction, Mix-FFN with a depth-wise 3x3, LayerNorms; decode head: per-stage Linear to
// decoder_hidden_size, bilinear upsample to 1/4 resolution, concatenation in REVERSED stage order, 1x1 fuse conv + BatchNorm +
// ReLU, 1x1 classifier).  Parameter names and shapes are that library's state_dict keys.
// Every matrix product (Linear layers, patch-embedding / sequence-reduction / 1x1 convolutions) runs on the implicit-GEMM
// kernel of conv_igemm.hip with its fused epilogues (bias, residual add, folded BatchNorm + ReLU, fp32 NCHW logits).
#include "segformer.h"

#include <utility>

#include <string.h>

#include "segformer_ops.h"
#include "tune.h"

namespace flair {

int conv_weight_rows_pad(int cout);   // conv_igemm.hip

#define SF_RUN(expr)                     \
  do {                                   \
    if (!dry_ && !err_) {                \
      int rc__ = (expr);                 \
      if (rc__) err_ = rc__;             \
    }                                    \
  } while (0)

long SegFormer::add_tensor(const std::string& name, int ndim, long d0, long d1, long d2, long d3, int kind) {
  SfTensor t;
  t.name = name; t.ndim = ndim; t.shape[0] = d0; t.shape[1] = d1; t.shape[2] = d2; t.shape[3] = d3; t.kind = kind;
  long n = 1;
  for (int i = 0; i < ndim; ++i) n *= t.shape[i];
  t.offset = n_params;
  n_params = round_up(n_params + n, 4);   // every tensor 16-byte aligned in the flat buffer
  tensors.push_back(t);
  return t.offset;
}

int SegFormer::add_lin(const std::string& name, int cin, int cout, int k, int stride, int pad, bool bias) {
  SfLin L;
  L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.pad = pad;
  L.cin_p = (int)round_up(cin, 8);
  const bool conv = name.find("#conv") != std::string::npos;
  std::string base = name.substr(0, name.find('#'));
  L.w_off = conv ? add_tensor(base + ".weight", 4, cout, cin, k, k, 0) : add_tensor(base + ".weight", 2, cout, cin, 1, 1, 0);
  L.b_off = bias ? add_tensor(base + ".bias", 1, cout, 1, 1, 1, 0) : -1;
  const int kstep = dtype == DT_F32 ? 32 : 64;
  L.Kg = k * k * L.cin_p;
  L.Kpad = (int)round_up(L.Kg, kstep);
  L.rows = conv_weight_rows_pad(cout);
  lins.push_back(L);
  return (int)lins.size() - 1;
}

int SegFormer::add_ln(const std::string& name, int C) {
  SfNorm n;
  n.C = C;
  n.g_off = add_tensor(name + ".weight", 1, C, 1, 1, 1, 0);
  n.b_off = add_tensor(name + ".bias", 1, C, 1, 1, 1, 0);
  norms.push_back(n);
  return (int)norms.size() - 1;
}

SegFormer::SegFormer(int in_ch, int labels, const int* depths_, const int* hidden_, const int* heads_, const int* sr_, int dec_hidden_,
                     int dt)
    : in_channels(in_ch), num_labels(labels), dec_hidden(dec_hidden_), dtype(dt) {
  const int patch[4] = {7, 3, 3, 3}, stride[4] = {4, 2, 2, 2};
  for (int i = 0; i < 4; ++i) { depths[i] = depths_[i]; hidden[i] = hidden_[i]; heads[i] = heads_[i]; sr[i] = sr_[i]; }
  for (int i = 0; i < 4; ++i) {
    const std::string st = "segformer.stages." + std::to_string(i);
    const int h = hidden[i];
    SfStage S;
    S.patch = add_lin(st + ".patch_embeddings.proj#conv", i == 0 ? in_ch : hidden[i - 1], h, patch[i], stride[i], patch[i] / 2, true);
    S.patch_ln = add_ln(st + ".patch_embeddings.layer_norm", h);
    for (int b = 0; b < depths[i]; ++b) {
      const std::string bl = st + ".blocks." + std::to_string(b);
      SfBlock K;
      K.ln1 = add_ln(bl + ".layernorm_before", h);
      K.q = add_lin(bl + ".attention.q_proj", h, h, 1, 1, 0, true);
      K.k = add_lin(bl + ".attention.k_proj", h, h, 1, 1, 0, true);
      K.v = add_lin(bl + ".attention.v_proj", h, h, 1, 1, 0, true);
      K.o = add_lin(bl + ".attention.o_proj", h, h, 1, 1, 0, true);
      K.sr = K.sr_ln = -1;
      if (sr[i] > 1) {
        K.sr = add_lin(bl + ".attention.sequence_reduction.sequence_reduction#conv", h, h, sr[i], sr[i], 0, true);
        K.sr_ln = add_ln(bl + ".attention.sequence_reduction.layer_norm", h);
      }
      K.ln2 = add_ln(bl + ".layernorm_after", h);
      K.fc1 = add_lin(bl + ".mlp.fc1", h, 4 * h, 1, 1, 0, true);
      K.dw_w = add_tensor(bl + ".mlp.dwconv.dwconv.weight", 4, 4 * h, 1, 3, 3, 0);
      K.dw_b = add_tensor(bl + ".mlp.dwconv.dwconv.bias", 1, 4 * h, 1, 1, 1, 0);
      K.fc2 = add_lin(bl + ".mlp.fc2", 4 * h, h, 1, 1, 0, true);
      S.blocks.push_back(K);
    }
    S.out_ln = add_ln(st + ".layer_norm", h);
    stages.push_back(S);
  }
  for (int i = 0; i < 4; ++i)
    dec_proj[i] = add_lin("decode_head.linear_projections." + std::to_string(i) + ".proj", hidden[i], dec_hidden, 1, 1, 0, true);
  fuse = add_lin("decode_head.linear_fuse#conv", 4 * dec_hidden, dec_hidden, 1, 1, 0, false);
  bn_g = add_tensor("decode_head.batch_norm.weight", 1, dec_hidden, 1, 1, 1, 0);
  bn_b = add_tensor("decode_head.batch_norm.bias", 1, dec_hidden, 1, 1, 1, 0);
  bn_rm = add_tensor("decode_head.batch_norm.running_mean", 1, dec_hidden, 1, 1, 1, 1);
  bn_rv = add_tensor("decode_head.batch_norm.running_var", 1, dec_hidden, 1, 1, 1, 1);
  cls = add_lin("decode_head.classifier#conv", dec_hidden, labels, 1, 1, 0, true);
}

bool SegFormer::shape_ok(int H, int W) const {
  // 1/32 grids; the reduced key / value sequence of every stage, (H/32) x (W/32) tokens, fits the attention kernel's LDS
  if (H < 32 || W < 32 || (H % 32) || (W % 32)) return false;
  const int nk = (H / 32) * (W / 32);
  if (nk > 256 || (nk % 16)) return false;
  for (int i = 0; i < 4; ++i) {
    if (hidden[i] % 64 || hidden[i] / heads[i] != 64) return false;   // heads of 64 channels (MiT-B1 .. B5)
    if (sr[i] != (8 >> i)) return false;                              // the published reduction ratios: every stage ends at 1/32
  }
  return dec_hidden % 64 == 0;
}

void* SegFormer::alloc(size_t bytes) {
  const size_t off = top_;
  top_ = (size_t)round_up((long)(top_ + bytes), 256);
  if (top_ > peak_) peak_ = top_;   // (stage scratch is released with top_ = mark: the plan is the high-water mark)
  if (!dry_ && top_ > cap_) { if (!err_) err_ = -100; return base_; }
  return base_ + off;
}

void SegFormer::gemm(const SfLin& L, const void* in, int B, int Hin, int Win, void* out, int out_ld, const void* res, const float* oscale,
                     const float* oshift, int relu, float* out_nchw) {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.src0 = in; a.C0 = L.cin_p; a.N = B; a.Hin = Hin; a.Win = Win;
  a.Hout = (Hin + 2 * L.pad - L.k) / L.stride + 1; a.Wout = (Win + 2 * L.pad - L.k) / L.stride + 1;
  a.R = L.k; a.S = L.k; a.out_mul = L.stride; a.pad = L.pad; a.in_div = 1;
  a.Cout = L.cout; a.Kg = L.Kg; a.Kpad = L.Kpad; a.w = base_ + L.packed;
  a.bias = L.b_off >= 0 ? params_ + L.b_off : nullptr;
  a.out = out; a.out_ld = out_ld; a.out_nchw = out_nchw;
  a.ores = res; a.oscale = oscale; a.oshift = oshift; a.orelu = relu;
  SF_RUN(launch_conv(dtype, a, s_));
}

void SegFormer::layernorm(const SfNorm& n, const void* x, void* y, long rows) {
  SF_RUN(sf_layernorm(dtype, x, params_ + n.g_off, params_ + n.b_off, y, rows, n.C, 1e-6f, s_));
}

int SegFormer::run(const float* params, const float* x_nchw, float* logits_quarter, float* logits_full, int B, int H, int W, void* ws,
                   size_t ws_bytes, hipStream_t s, bool dry) {
  if (!shape_ok(H, W)) return -10;
  base_ = dry ? (unsigned char*)0x100000 : (unsigned char*)ws;
  cap_ = ws_bytes; top_ = 0; peak_ = 0; dry_ = dry; err_ = 0; s_ = s; params_ = params;
  const size_t es = dtype_size(dtype);
  // ---- everything that depends on the weights alone — packed weights, folded BatchNorm, the decode head's pre-multiplied
  // matrices — sits at the front of the arena at shape-independent offsets and is rebuilt only when the parameter buffer, the
  // workspace or the head mode changes (or after weights_changed()): 0.3 ms of small launches per forward otherwise
  // 0: the library's order; 1: restructured by linearity; 2 (default): + everything after the per-stage products in one kernel (bf16)
  int head_mode = tune("FLAIR_SF_HEAD", 2);
  if (head_mode == 2 && !sf_head_fused_ok(dtype, H / 4, W / 4, hidden[0], dec_hidden, num_labels)) head_mode = 1;
  const bool fresh = !dry && cache_ok_ && cache_params_ == params && cache_ws_ == ws && cache_head_ == head_mode;
  for (auto& L : lins) { L.packed = top_; alloc((size_t)L.rows * L.Kpad * es); }
  if (!fresh) {
    PackTable tb;
    tb.n = 0;
    for (size_t i = 0; i < lins.size(); ++i) {
      const SfLin& L = lins[i];
      PackDesc& d = tb.d[tb.n++];
      memset(&d, 0, sizeof(d));
      d.w_off = L.w_off; d.dst_off = L.packed; d.Cout = L.cout; d.Cin = L.cin; d.R = L.k; d.S = L.k;
      d.Cin_p = L.cin_p; d.rows_pad = L.rows; d.Kpad = L.Kpad; d.tf = 0;
      if (tb.n == PackTable::MAX || i + 1 == lins.size()) {
        SF_RUN(pack_weights_all(dtype, params_, base_, tb, s_));
        tb.n = 0;
      }
    }
  }
  // key and value projections of a block as ONE product over [k_proj.weight; v_proj.weight] (2 h rows) with the concatenated bias:
  // they read the same (reduced) sequence of 8 192 tokens at B = 32 — two 7-16 us launches of 64 workgroups each otherwise
  kv_w_.clear(); kv_b_.clear();
  {
    PackTable tb;
    tb.n = 0;
    for (size_t si = 0; si < stages.size(); ++si)
      for (const SfBlock& K : stages[si].blocks) {
        const SfLin& Lk = lins[K.k];
        const SfLin& Lv = lins[K.v];
        const int h = Lk.cout, rows = conv_weight_rows_pad(2 * h);
        const size_t off = top_;
        alloc((size_t)rows * Lk.Kpad * es);
        float* bias = (float*)alloc((size_t)2 * h * 4);
        kv_w_.push_back(off); kv_b_.push_back(bias);
        if (fresh || dry_) continue;
        for (int half = 0; half < 2; ++half) {
          const SfLin& L = half ? Lv : Lk;
          PackDesc& d = tb.d[tb.n++];
          memset(&d, 0, sizeof(d));
          d.w_off = L.w_off; d.dst_off = off + (size_t)half * h * Lk.Kpad * es; d.Cout = h; d.Cin = L.cin; d.R = 1; d.S = 1;
          d.Cin_p = L.cin_p; d.rows_pad = half ? rows - h : h; d.Kpad = Lk.Kpad; d.tf = 0;
          if (!err_ && hipMemcpyAsync(bias + half * h, params_ + L.b_off, (size_t)h * 4, hipMemcpyDeviceToDevice, s_) != hipSuccess) err_ = -101;
        }
        if (tb.n + 2 > PackTable::MAX) {
          SF_RUN(pack_weights_all(dtype, params_, base_, tb, s_));
          tb.n = 0;
        }
      }
    if (tb.n) SF_RUN(pack_weights_all(dtype, params_, base_, tb, s_));
  }
  // depth-wise weights regrouped for the fused Mix-FFN kernel (stages of 64 / 128 channels, bf16)
  ffn_dw_.clear();
  for (size_t si = 0; si < stages.size(); ++si)
    for (const SfBlock& K : stages[si].blocks) {
      const int h = hidden[si];
      float* p = nullptr;
      if (dtype == DT_BF16 && (h == 64 || h == 128)) {
        p = (float*)alloc((size_t)4 * h * 10 * 4);
        if (!fresh) SF_RUN(sf_ffn_dw_pack(params_ + K.dw_w, params_ + K.dw_b, p, 4 * h, s_));
      }
      ffn_dw_.push_back(p);
    }
  float* bn_scale = (float*)alloc((size_t)dec_hidden * 4);
  float* bn_shift = (float*)alloc((size_t)dec_hidden * 4);
  if (!fresh) SF_RUN(bn_eval_coeffs(dec_hidden, params_ + bn_g, params_ + bn_b, params_ + bn_rm, params_ + bn_rv, 1e-5f, bn_scale, bn_shift, s_));
  // decode head by linearity (see below): W_i = F_i P_i per stage, shift2 = folded BatchNorm shift + the projected biases
  const int D = dec_hidden;
  const SfLin& Lf = lins[fuse];
  const int kstep = dtype == DT_F32 ? 32 : 64;
  float* shift2 = nullptr;
  void* wi[4] = {nullptr, nullptr, nullptr, nullptr};
  if (head_mode) {
    shift2 = (float*)alloc((size_t)D * 4);
    if (!fresh)
      SF_RUN(sf_fuse_bias(params_ + Lf.w_off, D, params_ + lins[dec_proj[3]].b_off, params_ + lins[dec_proj[2]].b_off,
                          params_ + lins[dec_proj[1]].b_off, params_ + lins[dec_proj[0]].b_off, bn_scale, bn_shift, shift2, s_));
    for (int i = 0; i < 4; ++i) {
      const SfLin& Lp = lins[dec_proj[i]];
      const int Ci = Lp.cin;
      // P_i transposed, packed as the weight of a product over D: rows = the C_i input channels, K = D
      const size_t pt_off = top_;
      const int pt_rows = conv_weight_rows_pad(Ci), pt_kpad = (int)round_up(D, kstep);
      alloc((size_t)pt_rows * pt_kpad * es);
      // F_i = columns [(3 - i) D, (4 - i) D) of the fuse weight, as a [D pixels][D channels] tensor
      void* fi = alloc((size_t)D * D * es);
      // W_i = F_i P_i: [D][C_i], exactly the packed layout of a C_i -> D product (K = C_i is a whole number of K steps)
      wi[i] = alloc((size_t)conv_weight_rows_pad(D) * round_up(Ci, kstep) * es);
      if (fresh) continue;
      PackTable tb;
      memset(&tb, 0, sizeof(tb));
      tb.n = 1;
      PackDesc& d = tb.d[0];
      d.w_off = Lp.w_off; d.dst_off = pt_off; d.Cout = D; d.Cin = Ci; d.R = 1; d.S = 1; d.Cin_p = D; d.rows_pad = pt_rows; d.Kpad = pt_kpad; d.tf = 1;
      SF_RUN(pack_weights_all(dtype, params_, base_, tb, s_));
      SF_RUN(sf_slice_cols(dtype, params_ + Lf.w_off, 4 * D, (3 - i) * D, D, D, fi, s_));
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.src0 = fi; a.C0 = D; a.N = 1; a.Hin = D; a.Win = 1; a.Hout = D; a.Wout = 1; a.R = 1; a.S = 1; a.out_mul = 1; a.in_div = 1;
      a.Cout = Ci; a.Kg = D; a.Kpad = pt_kpad; a.w = base_ + pt_off; a.out = wi[i]; a.out_ld = (int)round_up(Ci, kstep);
      SF_RUN(launch_conv(dtype, a, s_));
    }
  }
  void* wint = nullptr;
  void* wc32 = nullptr;
  if (head_mode == 2) {
    wint = alloc((size_t)128 * 96 * es);
    wc32 = alloc((size_t)32 * D * es);
    if (!fresh) {
      SF_RUN(sf_head_wint(wint, s_));
      if (!dry_ && !err_ && hipMemsetAsync(wc32, 0, (size_t)32 * D * es, s_) != hipSuccess) err_ = -101;
      SF_RUN(sf_slice_cols(dtype, params_ + lins[cls].w_off, D, 0, D, num_labels, wc32, s_));
    }
  }
  if (!dry) { cache_ok_ = err_ == 0; cache_params_ = params; cache_ws_ = ws; cache_head_ = head_mode; }
  // ---- input and the tensors that live to the decode head
  const int Cin_p = lins[stages[0].patch].cin_p;
  void* xin = alloc((size_t)B * H * W * Cin_p * es);
  SF_RUN(nchw_f32_to_nhwc(dtype, x_nchw, xin, B, in_channels, H, W, Cin_p, s_));
  const int H4 = H / 4, W4 = W / 4;
  void* feat[4];
  int fh[4], fw[4];
  for (int i = 0; i < 4; ++i) {
    fh[i] = H >> (i + 2); fw[i] = W >> (i + 2);
    feat[i] = alloc((size_t)B * fh[i] * fw[i] * hidden[i] * es);
  }
  // ---- encoder
  size_t blk = 0;
  const void* sin = xin;
  int sH = H, sW = W;
  for (int i = 0; i < 4; ++i) {
    const SfStage& S = stages[i];
    const int h = hidden[i], Hs = fh[i], Ws = fw[i];
    const long tokens = (long)B * Hs * Ws;
    const size_t mark = top_;
    void* x = alloc((size_t)tokens * h * es);
    void* ln = alloc((size_t)tokens * h * es);
    void* qb = alloc((size_t)tokens * h * es);
    void* ctx = alloc((size_t)tokens * h * es);
    void* f1 = alloc((size_t)tokens * 4 * h * es);
    void* f2 = alloc((size_t)tokens * 4 * h * es);
    const int Hk = Hs / sr[i], Wk = Ws / sr[i];
    const long ktok = (long)B * Hk * Wk;
    void* red = alloc((size_t)ktok * h * es);
    void* redn = alloc((size_t)ktok * h * es);
    void* kvb = alloc((size_t)ktok * 2 * h * es);
    // overlapping patch embedding (strided convolution + bias), LayerNorm
    gemm(lins[S.patch], sin, B, sH, sW, ln, h, nullptr, nullptr, nullptr, 0, nullptr);
    layernorm(norms[S.patch_ln], ln, x, tokens);
    const void* ln1_done = nullptr;   // layernorm_before of the coming block, already written by the previous block's fused FFN
    bool stage_norm_done = false;
    for (size_t bi = 0; bi < S.blocks.size(); ++bi) {
      const SfBlock& K = S.blocks[bi];
      const void* lnin = ln1_done;
      if (!lnin) {
        layernorm(norms[K.ln1], x, ln, tokens);
        lnin = ln;
      }
      ln1_done = nullptr;
      gemm(lins[K.q], lnin, B, Hs, Ws, qb, h, nullptr, nullptr, nullptr, 0, nullptr);
      const void* kv_in = lnin;
      if (K.sr >= 0) {   // sequence reduction: sr x sr convolution of stride sr over the token grid, LayerNorm
        gemm(lins[K.sr], lnin, B, Hs, Ws, red, h, nullptr, nullptr, nullptr, 0, nullptr);
        layernorm(norms[K.sr_ln], red, redn, ktok);
        kv_in = redn;
      }
      {
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        const SfLin& Lk = lins[K.k];
        a.src0 = kv_in; a.C0 = Lk.cin_p; a.N = B; a.Hin = Hk; a.Win = Wk; a.Hout = Hk; a.Wout = Wk; a.R = 1; a.S = 1; a.out_mul = 1; a.in_div = 1;
        a.Cout = 2 * h; a.Kg = Lk.Kg; a.Kpad = Lk.Kpad; a.w = base_ + kv_w_[blk]; a.bias = kv_b_[blk]; a.out = kvb; a.out_ld = 2 * h;
        SF_RUN(launch_conv(dtype, a, s_));
      }
      ++blk;
      SF_RUN(sf_attention(dtype, qb, kvb, (const unsigned char*)kvb + (size_t)h * es, ctx, B, Hs * Ws, Hk * Wk, h, 2 * h, s_));
      gemm(lins[K.o], ctx, B, Hs, Ws, x, h, /*residual*/ x, nullptr, nullptr, 0, nullptr);   // x = o_proj(ctx) + x, element by element in place
      if (ffn_dw_[blk - 1] && sf_ffn_fused_ok(dtype, h, Hs, Ws) && tune("FLAIR_SF_FFN", 1)) {
        // LayerNorm, fc1, depth-wise 3x3 + GELU, fc2 and the residual in one kernel; the result lands in the other buffer
        // ... and the LayerNorm that FOLLOWS the block from the same registers: the next block's layernorm_before (into ctx: free
        // until that block's attention writes it, by when its readers — q and the sequence reduction — are done) or the stage's
        // final norm (into the feature map).  K.sr >= 0 in these stages, so nothing else reads layernorm_before's output.
        const SfNorm& n2 = norms[K.ln2];
        const bool last = bi + 1 == S.blocks.size();
        const bool emit = K.sr >= 0 && tune("FLAIR_SF_FFN_LN", 1);
        const SfNorm& nn = norms[last ? S.out_ln : S.blocks[bi + 1].ln1];
        void* lnout = emit ? (last ? feat[i] : ctx) : nullptr;
        SF_RUN(sf_ffn_fused(x, params_ + n2.g_off, params_ + n2.b_off, base_ + lins[K.fc1].packed, params_ + lins[K.fc1].b_off, ffn_dw_[blk - 1],
                            base_ + lins[K.fc2].packed, params_ + lins[K.fc2].b_off, ln, B, Hs, Ws, h, 1e-6f, params_ + nn.g_off,
                            params_ + nn.b_off, lnout, s_));
        std::swap(x, ln);
        if (emit && last) stage_norm_done = true;
        else if (emit) ln1_done = ctx;
      } else {
        layernorm(norms[K.ln2], x, ln, tokens);
        gemm(lins[K.fc1], ln, B, Hs, Ws, f1, 4 * h, nullptr, nullptr, nullptr, 0, nullptr);
        SF_RUN(sf_dwconv3x3_gelu(dtype, f1, params_ + K.dw_w, params_ + K.dw_b, f2, B, Hs, Ws, 4 * h, s_));
        gemm(lins[K.fc2], f2, B, Hs, Ws, x, h, /*residual*/ x, nullptr, nullptr, 0, nullptr);
      }
    }
    if (!stage_norm_done) layernorm(norms[S.out_ln], x, feat[i], tokens);
    top_ = mark;   // the stage's scratch is free again (one stream: later launches are ordered behind its readers)
    sin = feat[i]; sH = Hs; sW = Ws;
  }
  // ---- decode head.  The library's order (Linear to D per stage, upsample, concatenate stage 3 .. 0, 1x1 fuse over 4 D channels,
  // BatchNorm, ReLU) restructured by linearity (segformer_ops.hip, 'decode-head helpers'): per stage  g_i = (F_i P_i) f_i  at the
  // stage's own resolution, then  z = relu(bn(g_0 + up(g_1) + up(g_2) + up(g_3) + F [b_3 | b_2 | b_1 | b_0])).
  // FLAIR_SF_HEAD=0 runs the library's order (A/B and parity of the restructuring itself).
  float* lq = logits_quarter ? logits_quarter : (float*)alloc((size_t)B * num_labels * H4 * W4 * 4);
  void* z = head_mode == 2 ? nullptr : alloc((size_t)B * H4 * W4 * dec_hidden * es);
  if (head_mode) {
    void* g[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = head_mode == 2 ? 1 : 0; i < 4; ++i) {
      const int Ci = lins[dec_proj[i]].cin;
      g[i] = alloc((size_t)B * fh[i] * fw[i] * D * es);
      ConvArgs a;
      memset(&a, 0, sizeof(a));
      a.src0 = feat[i]; a.C0 = Ci; a.N = B; a.Hin = fh[i]; a.Win = fw[i]; a.Hout = fh[i]; a.Wout = fw[i]; a.R = 1; a.S = 1; a.out_mul = 1;
      a.in_div = 1; a.Cout = D; a.Kg = Ci; a.Kpad = (int)round_up(Ci, kstep); a.w = wi[i]; a.out = g[i]; a.out_ld = D;
      SF_RUN(launch_conv(dtype, a, s_));
    }
    if (head_mode == 2)
      SF_RUN(sf_head_fused(feat[0], wi[0], g[1], g[2], g[3], wint, bn_scale, shift2, wc32, params_ + lins[cls].b_off, lq, B, H4, W4, D,
                           num_labels, s_));
    else
      SF_RUN(sf_upsample_sum_bn_relu(dtype, g[0], g[1], g[2], g[3], bn_scale, shift2, z, B, H4, W4, D, s_));
  } else {
    const int cat_ld = 4 * dec_hidden;
    void* cat = alloc((size_t)B * H4 * W4 * cat_ld * es);
    for (int i = 0; i < 4; ++i) {
      unsigned char* slot = (unsigned char*)cat + (size_t)(3 - i) * dec_hidden * es;
      if (i == 0) {   // already at 1/4 resolution: straight into its channel slice
        gemm(lins[dec_proj[0]], feat[0], B, fh[0], fw[0], slot, cat_ld, nullptr, nullptr, nullptr, 0, nullptr);
      } else {
        const size_t mark = top_;
        void* p = alloc((size_t)B * fh[i] * fw[i] * dec_hidden * es);
        gemm(lins[dec_proj[i]], feat[i], B, fh[i], fw[i], p, dec_hidden, nullptr, nullptr, nullptr, 0, nullptr);
        SF_RUN(sf_bilinear_nhwc(dtype, p, slot, B, fh[i], fw[i], dec_hidden, H4, W4, cat_ld, s_));
        top_ = mark;
      }
    }
    gemm(lins[fuse], cat, B, H4, W4, z, dec_hidden, nullptr, bn_scale, bn_shift, 1, nullptr);   // 1x1 conv + folded BatchNorm + ReLU
  }
  if (head_mode != 2) gemm(lins[cls], z, B, H4, W4, nullptr, 0, nullptr, nullptr, nullptr, 0, lq);   // classifier: fp32 NCHW logits at 1/4 resolution
  if (logits_full) SF_RUN(sf_bilinear_nchw_f32(lq, logits_full, (long)B * num_labels, H4, W4, H, W, s_));
  need_ = peak_ + (1 << 20);
  if (err_) cache_ok_ = false;
  return err_;
}

size_t SegFormer::workspace_bytes(int B, int H, int W) {
  if (run(nullptr, nullptr, nullptr, reinterpret_cast<float*>(16), B, H, W, nullptr, 0, nullptr, true)) return 0;
  return need_;
}

int SegFormer::forward(const float* params, const float* x_nchw, float* logits_quarter, float* logits_full, int B, int H, int W,
                       void* ws, size_t ws_bytes, hipStream_t s) {
  if (!params || !x_nchw || (!logits_quarter && !logits_full) || !ws) return -1;
  return run(params, x_nchw, logits_quarter, logits_full, B, H, W, ws, ws_bytes, s, false);
}

}  // namespace flair
