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
// q [B][N][hidden], k / v [B][Nk][hidden] token-major, heads of 64 channels; out like q
int sf_attention(int dtype, const void* q, const void* k, const void* v, void* out, int B, int N, int Nk, int hidden, hipStream_t s);
}  // namespace flair
