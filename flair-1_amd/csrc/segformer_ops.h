// Launchers of segformer_ops.hip (see there).
#pragma once
#include "common.h"

namespace flair {
int sf_layernorm(int dtype, const void* x, const float* gamma, const float* beta, void* y, long rows, int C, float eps, hipStream_t s);
int sf_dwconv3x3_gelu(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, hipStream_t s);
// NHWC [B][h][w][C] -> channels [0, C) of rows of ld elements of an NHWC [B][H][W][ld] tensor, align_corners = False
int sf_bilinear_nhwc(int dtype, const void* x, void* y, int B, int h, int w, int C, int H, int W, int ld, hipStream_t s);
int sf_bilinear_nchw_f32(const float* x, float* y, long planes, int h, int w, int H, int W, hipStream_t s);
// decode-head restructuring (see segformer_ops.hip): fp32 column slice -> T, fused projection-bias term, upsample-sum + BN + ReLU
int sf_slice_cols(int dtype, const float* src, int ld, int col0, int ncols, long rows, void* dst, hipStream_t s);
int sf_fuse_bias(const float* wf, int D, const float* b3, const float* b2, const float* b1, const float* b0, const float* scale,
                 const float* shift, float* shift2, hipStream_t s);
int sf_upsample_sum_bn_relu(int dtype, const void* g0, const void* g1, const void* g2, const void* g3, const float* scale, const float* shift2,
                            void* z, int B, int H, int W, int D, hipStream_t s);
// Mix-FFN of one block in one kernel (bf16, C = 64 / 128; see segformer_ops.hip): out = x + fc2(gelu(dwconv3x3(fc1(LayerNorm(x))))),
// x / out [B][H][W][C] (different buffers), w1 [4C][C], w2 [C][4C] packed row-major, dwp from sf_ffn_dw_pack ([4C / 4][10][4] fp32)
bool sf_ffn_fused_ok(int dtype, int C, int H, int W);
int sf_ffn_dw_pack(const float* w, const float* b, float* out, int nch, hipStream_t s);
// out_ln (optional, not x): LayerNorm(out) with ln2_g / ln2_b — the norm that follows the block, from the epilogue's registers
int sf_ffn_fused(const void* x, const float* ln_g, const float* ln_b, const void* w1, const float* b1, const float* dwp, const void* w2,
                 const float* b2, void* out, int B, int H, int W, int C, float eps, const float* ln2_g, const float* ln2_b, void* out_ln,
                 hipStream_t s);
// the whole decode head after the per-stage products in one kernel (bf16; see segformer_ops.hip): f0 [B][H][W][64], w0 [D][64],
// g1..g3 [B][H >> i][W >> i][D], wint = the 128 x 96 interpolation matrix (sf_head_wint), wc [32][D] (rows >= labels zero), out fp32 NCHW
bool sf_head_fused_ok(int dtype, int H, int W, int C0, int D, int labels);
int sf_head_wint(void* wint, hipStream_t s);
int sf_head_fused(const void* f0, const void* w0, const void* g1, const void* g2, const void* g3, const void* wint, const float* scale,
                  const float* shift2, const void* wc, const float* bc, float* out, int B, int H, int W, int D, int labels, hipStream_t s);
// q [B][N][hidden], k / v [B][Nk] rows of kv_ld elements (hidden of them used: k and v may interleave as the halves of one fused
// projection), token-major, heads of 64 channels; out like q
int sf_attention(int dtype, const void* q, const void* k, const void* v, void* out, int B, int N, int Nk, int hidden, int kv_ld, hipStream_t s);
}  // namespace flair
