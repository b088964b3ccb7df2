// Native inference executor of SegformerForSemanticSegmentation (see segformer.hip).
#pragma once
#include <string>
#include <vector>

#include "ops.h"

namespace flair {

struct SfTensor {
  std::string name;
  int ndim;
  long shape[4];
  long offset;   // floats into the flat buffer
  int kind;      // 0 parameter, 1 BatchNorm running statistic
};

struct SfLin {   // Linear (k = 1) or Conv2d(k, stride, pad) as an implicit GEMM
  int cin, cout, k, stride, pad, cin_p;
  long w_off, b_off;
  int Kg, Kpad, rows;
  size_t packed = 0;
};
struct SfNorm { int C; long g_off, b_off; };
struct SfBlock { int ln1, q, k, v, o, sr, sr_ln, ln2, fc1, fc2; long dw_w, dw_b; };
struct SfStage { int patch, patch_ln, out_ln; std::vector<SfBlock> blocks; };

class SegFormer {
 public:
  SegFormer(int in_channels, int num_labels, const int* depths, const int* hidden, const int* heads, const int* sr, int dec_hidden,
            int dtype);
  int in_channels, num_labels, dec_hidden, dtype;
  int depths[4], hidden[4], heads[4], sr[4];
  std::vector<SfTensor> tensors;
  long n_params = 0;
  bool shape_ok(int H, int W) const;
  size_t workspace_bytes(int B, int H, int W);
  // logits_quarter: fp32 NCHW (B, labels, H/4, W/4) = the library's `.logits`; logits_full: the same upsampled x4 (bilinear,
  // align_corners = False) to the tile size.  Either may be null, not both.
  int forward(const float* params, const float* x_nchw, float* logits_quarter, float* logits_full, int B, int H, int W, void* ws,
              size_t ws_bytes, hipStream_t s);
  // The packed weights at the front of the workspace are reused while `params` and `ws` stay the same pointers; call this after
  // changing the parameter buffer's contents in place.
  void weights_changed() { cache_ok_ = false; }

 private:
  std::vector<SfLin> lins;
  std::vector<SfNorm> norms;
  std::vector<SfStage> stages;
  int dec_proj[4], fuse, cls;
  long bn_g, bn_b, bn_rm, bn_rv;
  unsigned char* base_ = nullptr;
  size_t cap_ = 0, top_ = 0, peak_ = 0, need_ = 0;
  bool dry_ = false;
  int err_ = 0;
  hipStream_t s_ = nullptr;
  const float* params_ = nullptr;
  std::vector<size_t> kv_w_;   // per block, in stage order: arena offset of the fused [k; v] packed weight, its concatenated bias
  std::vector<float*> kv_b_;
  std::vector<float*> ffn_dw_;   // per block: regrouped depth-wise weights for the fused Mix-FFN kernel (null: not applicable)
  bool cache_ok_ = false;
  const float* cache_params_ = nullptr;
  const void* cache_ws_ = nullptr;
  int cache_head_ = -1;
  long add_tensor(const std::string& name, int ndim, long d0, long d1, long d2, long d3, int kind);
  int add_lin(const std::string& name, int cin, int cout, int k, int stride, int pad, bool bias);
  int add_ln(const std::string& name, int C);
  void* alloc(size_t bytes);
  void gemm(const SfLin& L, const void* in, int B, int Hin, int Win, void* out, int out_ld, const void* res, const float* oscale,
            const float* oshift, int relu, float* out_nchw);
  void layernorm(const SfNorm& n, const void* x, void* y, long rows);
  int run(const float* params, const float* x_nchw, float* logits_quarter, float* logits_full, int B, int H, int W, void* ws,
          size_t ws_bytes, hipStream_t s, bool dry);
};

}  // namespace flair
