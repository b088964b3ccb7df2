// Shared device/host helpers for the FLAIR segmentation hot path on gfx950 (MI355X).
// All activations are NHWC in HBM; T is float (parity mode) or bf16 (throughput mode).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace flair {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

enum { DT_F32 = 0, DT_BF16 = 1 };

__host__ __device__ inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int CH = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int CH = 8;
  __device__ static __forceinline__ float to_f(bf16_t v) { return bf16_to_f32(v); }
  __device__ static __forceinline__ bf16_t from_f(float v) { return f32_to_bf16(v); }
};

// unpack / pack one 16-byte chunk to floats
template <typename T> __device__ __forceinline__ void chunk_to_f(const uint4& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f<float>(const uint4& c, float* f) {
  f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y); f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ __forceinline__ void chunk_to_f<bf16_t>(const uint4& c, float* f) {
  f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
  f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
  f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
  f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 f_to_chunk(const float* f);
template <> __device__ __forceinline__ uint4 f_to_chunk<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 f_to_chunk<bf16_t>(const float* f) {
  uint4 r;
  r.x = (unsigned)f32_to_bf16(f[0]) | ((unsigned)f32_to_bf16(f[1]) << 16);
  r.y = (unsigned)f32_to_bf16(f[2]) | ((unsigned)f32_to_bf16(f[3]) << 16);
  r.z = (unsigned)f32_to_bf16(f[4]) | ((unsigned)f32_to_bf16(f[5]) << 16);
  r.w = (unsigned)f32_to_bf16(f[6]) | ((unsigned)f32_to_bf16(f[7]) << 16);
  return r;
}

// relu(x*sc + sh) on one 16-byte chunk (the bn_act arithmetic: fmaf, fmaxf, round to T)
template <typename T, typename V>
__device__ __forceinline__ V chunk_bn_relu(const V& v, const float* sc, const float* sh) {
  constexpr int CH = Elem<T>::CH;
  float f[CH];
  chunk_to_f<T>(__builtin_bit_cast(uint4, v), f);
#pragma unroll
  for (int e = 0; e < CH; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
  return __builtin_bit_cast(V, f_to_chunk<T>(f));
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

#define FLAIR_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return (int)e__;                    \
  } while (0)

// ----------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (forward and data-gradient share one gather-form kernel).
struct ConvArgs {
  const void* src0;  // NHWC T, C0 channels (at half resolution when up0)
  const void* src1;  // NHWC T, C1 channels (skip), may be null
  int C0, C1;        // Cin = C0 + C1 (each a multiple of the 16-byte chunk)
  int up0;           // nearest x2 upsample of src0 fused into the gather
  int N, Hin, Win;   // logical input extent (after the upsample)
  int Hout, Wout;
  int R, S;
  int out_mul, pad, in_div;  // num = o*out_mul - pad + r ; tap valid iff num % in_div == 0 ; i = num / in_div
  int Cout;          // real output channels
  int Kg;            // R*S*Cin
  int Kpad;          // row length of the packed weight (multiple of the K step)
  const void* w;     // packed [Cout_pad][Kpad] T, kk = (r*S+s)*Cin + c
  const float* bias; // optional [Cout]
  void* out;         // NHWC T, row stride out_ld elements (null when only out_nchw is wanted)
  int out_ld;
  float* out_nchw;   // optional fp32 NCHW [N][Cout][Hout][Wout]
  float* stats;      // optional per-row-block partial sums, channel-major [2][Cout][gridDim.x] (sum, sum of squares)
  int accumulate;    // out += result
  // optional fused input transform (small-channel halo kernels only): the input tensor is a pre-BatchNorm conv
  // output and the kernel applies x' = relu(x*in_scale[c] + in_shift[c]) per concatenated input channel while it
  // stages the halo — the activation tensor of the producing unit is never written ("lazy" BN + ReLU).  Pixels
  // outside the image stay zero.  The rounding is that of bn_act, so results are bit-identical to the materialised path.
  const float* in_scale;
  const float* in_shift;
  // optional fused output epilogue (inference: BatchNorm folded to a per-channel affine, residual, ReLU):
  //   out = [relu]( acc * oscale[n] + oshift[n] (+ bias[n]) [+ ores] )
  const float* oscale;
  const float* oshift;
  const void* ores;  // NHWC T, same shape / row stride as out
  int orelu;
  // optional fused backward of "nearest x2 upsample + concat" (data gradient of a decoder conv1, halo-tile
  // kernels only): columns [0, pool_c0) are 2x2 sum-pooled into `out` ([N][H/2][W/2][out_ld]), the remaining
  // columns go to out_skip ([N][H][W][out_skip_ld]); pool_c0 is a multiple of the kernel's column tile
  int pool_c0;
  void* out_skip;
  int out_skip_ld;
  int skip_accumulate;
  // optional output sub-sampling (gather-form kernel only): output pixel (ho, wo) is pixel (2*ho + out_oy, 2*wo + out_ox)
  // of a [N][2*Hout][2*Wout][out_ld] tensor — one parity class of a stride-2 data gradient
  int out_sub, out_oy, out_ox;
  // out_sub with ncls == 4: all four parity classes of a 3x3 stride-2 data gradient in one launch, class = blockIdx.z
  // = 2*oy + ox, taps (oy ? 2 : 1) x (ox ? 2 : 1), packed weights cls_w[class] with row length cls_kpad[class]
  int ncls;
  const void* cls_w[4];
  int cls_kpad[4];
  int xcd_remap;   // set by the halo-GEMM launcher: XCD-contiguous workgroup -> work-item map
  // optional fused first pass of the BatchNorm backward of the unit that PRODUCED this conv's input (halo-tile
  // kernels only).  `out` (the pooled part when pool_c0 > 0) is then the complete gradient dz w.r.t. that unit's
  // ReLU output; the epilogue also reads the unit's pre-BN tensor bnr_y (same shape and row stride as `out`) and
  // leaves  sum(dz*m)  and  sum(dz*m*y),  m = [y*bnr_scale + bnr_shift > 0]  (or [bnr_out > 0]),  per pixel tile in
  // bnr_partial[2][bnr_C][gridDim.x] — what bn_bwd_reduce_kernel would compute from HBM.
  const void* bnr_y;
  const void* bnr_out;   // optional: the unit's post-ReLU output; the mask is then out > 0 (units with a residual branch)
  const float* bnr_scale;
  const float* bnr_shift;
  float* bnr_partial;
  int bnr_C;
  // diagnostic builds only (FLAIR_HG_STAMP): per-workgroup s_memtime stamps [gridDim.x][8], written by thread 0
  unsigned long long* dbg;
  // optional (persistent small-channel kernel with one column block, conv_halo_preds_ok()): argmax over the Cout output channels
  // of every pixel, first maximum wins, NaN wins over numbers — what argmax(softmax(logits)) of the reference's predict_step
  // returns away from exact ties of the rounded probabilities — as uint8 [N][H][W]; `out` may then be null
  unsigned char* preds_u8;
  float* maxprob_f32;   // with preds_u8: the winner's softmax probability, 1 / sum_c exp(x_c - max), fp32 [N][H][W] (convert('argmax') band 1)
  // optional fused BatchNorm-backward APPLY on the input (halo-GEMM data gradients only; ask conv_bnapply_fusable()).  src0 then
  // holds dz, the gradient w.r.t. the ReLU output of the unit this data gradient belongs to, ALREADY masked by the ReLU (the kernel
  // that completed it ran with bnr_mask), ap_y the unit's pre-BN tensor (same shape / row stride) and ap_coef = k1 | k2 | k3
  // (C0 floats each, bn_bwd_finalize_kernel).  The kernel stages  dy = k1*dz + k2*y + k3  with bn_bwd_apply_kernel's arithmetic
  // and rounding while it fills its halo — the apply pass is gone — and the workgroups of output-channel block 0 also leave the
  // interior of every staged chunk in ap_dy ([N][H][W][C0] T, optional) for the unit's weight-gradient kernel.
  const void* ap_y;
  const float* ap_coef;
  void* ap_dy;
  // accumulate == 1 reads its addend from acc_src instead of `out` when set (halo-GEMM epilogue, plain stores only): the
  // identity branch of a BasicBlock hands its (masked) gradient over without the copy bn_bwd_apply used to make
  const void* acc_src;
  // with bnr_partial: store dz * m (m = the ReLU mask the fused reduction computes anyway) instead of dz, so that the consumers
  // of this gradient need no mask source (halo-GEMM epilogue only)
  int bnr_mask;
};
bool conv_bnapply_fusable(int dtype, const ConvArgs& a);   // a.ap_y set: will launch_conv run the halo-GEMM kernel that applies it?
bool conv_acc_src_ok(int dtype, const ConvArgs& a);        // a.acc_src set: does the kernel launch_conv picks honour it?
bool conv_halo_preds_ok(int dtype, const ConvArgs& a);

int launch_conv(int dtype, const ConvArgs& a, hipStream_t s);
int conv_grid_rows(int dtype, const ConvArgs& a);  // number of row blocks (= partial-stat rows)
bool conv_mfma_bound(int dtype, const ConvArgs& a);  // true when launch_conv picks the 128-wide halo-GEMM kernel

// Weight gradient: dW[k][(r,s),c] = sum_p dY[p][k] * im2col(X)[p][(r,s),c]; split over pixels.
struct WgradArgs {
  const void* x0; const void* x1; int C0, C1, up0;  // same gather as the forward conv
  int N, Hin, Win, Hout, Wout, R, S, stride, pad;
  const void* dy; int dy_ld; int Cout;               // NHWC T [M][dy_ld], first Cout channels used
  float* partial;                                    // [splits][Cout_pad][Kpad] fp32 workspace
  float* dw;                                         // OIHW fp32 [Cout][Cin_real][R][S]
  int Cin_real;                                      // channels of dw (<= C0+C1, rest is padding)
  int accumulate;
  const float* in_scale;                             // optional lazy BN + ReLU on x (see ConvArgs::in_scale);
  const float* in_shift;                             // small-channel halo kernel only
  int cus;                                           // persistent workgroups (= CUs) the register-resident kernel may occupy;
                                                     // 0 = the whole chip.  The executor gives its side stream half of it.
  // optional bias gradient of the same layer, dbias[k] = sum_p dY[p][k], folded into the small-channel weight-gradient
  // kernel (it stages every dY tile anyway; the head's separate column-sum pass read 268 MB for 13 numbers):
  // dbias_partial = [dy_ld][WGRAD_DBIAS_ROWS] floats of scratch.  Ask wgrad_dbias_fusable() first.
  float* dbias;
  float* dbias_partial;
  // optional fused BatchNorm-backward apply on dy (the stem's weight-gradient kernel only; ask wgrad_bnapply_fusable()): `dy` is
  // then the gradient w.r.t. the unit's ReLU output, and the kernel stages  k1*dz + k2*y + k3,  dz = dy * [y*msc + msh > 0],
  // with the rounding of bn_bwd_apply_kernel — the tensor bn_bwd_apply would have written is never materialised
  const void* fuse_y;
  const float* fuse_coef;   // k1 | k2 | k3, each Cout floats (bn_bwd_finalize_kernel)
  const float* fuse_msc;
  const float* fuse_msh;
};
bool wgrad_bnapply_fusable(int dtype, const WgradArgs& a);
constexpr int WGRAD_DBIAS_ROWS = 1024;
bool wgrad_dbias_fusable(int dtype, const WgradArgs& a);
int launch_wgrad(int dtype, const WgradArgs& a, hipStream_t s);
size_t wgrad_workspace_bytes(int dtype, const WgradArgs& a);

}  // namespace flair
