// Kernels of the SegFormer (MiT) inference path that the U-Net path does not have: LayerNorm over channels, efficient
// self-attention with a spatially reduced key / value sequence, depth-wise 3x3 + GELU of the Mix-FFN, bilinear resampling.
// (BASELINE config 5: zone_detect with a HuggingFace model, /root/reference/src/zone_detect/model.py:42-50, compare.py:31-36.
// The model itself is third-party — transformers' SegformerForSemanticSegmentation — restated from its published algorithm;
// the matrix products run on the gather-form implicit-GEMM kernel of conv_igemm.hip, 1x1 and strided convolutions alike.)
// Activations are token-major [B * H * W][C] = NHWC, T = float (parity mode, exact-fp32 MFMA) or bf16.
#include "segformer_ops.h"

#include <type_traits>

#include "prof.h"
#include "tune.h"

namespace flair {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------ LayerNorm
// One group of G lanes (a power of two <= 64) per row, up to two 16-byte chunks per lane in registers; mean, then the
// biased variance of the centred values (two passes over registers, fp32), eps inside the square root — nn.LayerNorm.
template <typename T, int G>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, long rows, int C, float eps) {
  constexpr int CH = Elem<T>::CH;
  const int nch = C / CH;
  const int lane = threadIdx.x % G;
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) / G;
  const bool live = row < rows;
  float v[2][CH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + i * G;
    const bool ok = live && c < nch;
#pragma unroll
    for (int e = 0; e < CH; ++e) v[i][e] = 0.f;
    if (ok) chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + row * C + (long)c * CH), v[i]);
#pragma unroll
    for (int e = 0; e < CH; ++e) s += v[i][e];
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const bool ok = lane + i * G < nch;
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      const float d = ok ? v[i][e] - mean : 0.f;
      q = fmaf(d, d, q);
    }
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = lane + i * G;
    if (live && c < nch) {
      float o[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) o[e] = fmaf((v[i][e] - mean) * rstd, gamma[c * CH + e], beta[c * CH + e]);
      *reinterpret_cast<uint4*>(y + row * C + (long)c * CH) = f_to_chunk<T>(o);
    }
  }
}

// ------------------------------------------------------------------------------------------- depth-wise 3x3 + bias + GELU
// SegformerMixMLP: dwconv(fc1(x)) then the exact (erf) GELU.  A thread keeps ONE 16-byte channel chunk for the whole launch —
// its 9 x CH weights and CH biases live in registers — and walks a segment of L pixels along an image row with the 3 x 3 window
// of unpacked inputs in registers: three new chunks (the next column) per output instead of nine.  Neighbouring threads hold
// neighbouring chunks of the same pixel, so every load and store of a wave is a contiguous run.  (History: one (pixel, chunk)
// item per thread with its 72 weights re-read per item 0.24 TB/s; register-resident weights, nine loads and unpacks per output
// 1.9 TB/s — ~430 vector instructions per 16 bytes of output, half of them unpacking and the library's branchy erff.)
// The accumulation order (bias, then taps row-major) is the same in every version.
template <typename T>
__device__ __forceinline__ float gelu_erf(float v) {
  if constexpr (sizeof(T) == 4) {
    return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));   // parity mode: the library's erf
  } else {
    // bf16 outputs (2^-9 relative): Phi(-|v|) = erfc(|v| / sqrt 2) / 2 by Abramowitz-Stegun 7.1.26 (|error| < 1e-7), branch-free
    const float a = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float h = p * t * __builtin_amdgcn_exp2f(a * a * -1.4426950408889634f);
    return v * (v < 0.f ? h : 1.f - h);
  }
}

// E = 4 channels per thread in both modes (16 bytes of fp32, 8 bytes of bf16): 36 + 36 registers of weights and window, so
// four waves per SIMD are resident (E = 8 in bf16: 190 registers, two waves, 2.6 TB/s).  W % L == 0.
template <typename T, int L>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void dwconv3x3_gelu_kernel(const T* x, const float* __restrict__ w /* [C][3][3] */,
                                                             const float* __restrict__ bias, T* y, int B, int H, int W, int C,
                                                             int cg /* 4-channel groups per workgroup: a power of two <= 256 */) {
  constexpr int E = 4;
  typedef typename std::conditional<sizeof(T) == 4, uint4, uint2>::type V;
  const int c = blockIdx.y * cg + threadIdx.x % cg;     // this thread's channel group
  const int pl = threadIdx.x / cg, npl = 256 / cg;      // pixel lane of the workgroup
  float wr[9][E], bs[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    bs[e] = bias[c * E + e];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wr[tp][e] = w[(c * E + e) * 9 + tp];
  }
  auto unpack = [](const V& v, float* f) {
    if constexpr (sizeof(T) == 4) {
      f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    } else {
      f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
      f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    }
  };
  const int nseg = W / L;
  const int items = B * H * nseg;                       // (image row, segment)
  for (int it = blockIdx.x * npl + pl; it < items; it += gridDim.x * npl) {
    const int seg = it % nseg, row = it / nseg, py = row % H;
    const int xs = seg * L;
    const T* rp = x + ((long)row * W) * C + (long)c * E;
    T* yp = y + ((long)row * W) * C + (long)c * E;
    const bool up = py > 0, dn = py < H - 1;
    // rows outside the image: read the centre row instead (a valid address) and zero the values
    const T* rows[3] = {up ? rp - (long)W * C : rp, rp, dn ? rp + (long)W * C : rp};
    float win[3][3][E];                                 // [column slot][row][element]
    V raw[3];                                           // the next column, in flight during one step's arithmetic
    auto issue_col = [&](int xx, bool edge) {           // edge: this column may lie outside the image (compile-time per call site)
      const int xo = ((!edge || (unsigned)xx < (unsigned)W) ? xx : 0) * C;
#pragma unroll
      for (int r = 0; r < 3; ++r) raw[r] = *reinterpret_cast<const V*>(rows[r] + xo);
    };
    auto take_col = [&](int slot, int xx, bool edge) {
      const bool xok = !edge || (unsigned)xx < (unsigned)W;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool ok = xok && (r == 0 ? up : r == 2 ? dn : true);
        V v = raw[r];
        if (r != 1 || edge) {
          v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u;
          if constexpr (sizeof(T) == 4) { v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u; }
        }
        unpack(v, win[slot][r]);
      }
    };
    issue_col(xs - 1, true);
    take_col(0, xs - 1, true);
    issue_col(xs, false);
    take_col(1, xs, false);
    issue_col(xs + 1, L == 1);
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int xx = xs + i;
      take_col((i + 2) % 3, xx + 1, i == L - 1);
      if (i + 1 < L) issue_col(xx + 2, i + 1 == L - 1);
      float acc[E];
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = bs[e];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int e = 0; e < E; ++e) acc[e] = fmaf(win[(i + q) % 3][r][e], wr[r * 3 + q][e], acc[e]);
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = gelu_erf<T>(acc[e]);
      V o;
      if constexpr (sizeof(T) == 4) {
        o = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3]));
      } else {
        o.x = (unsigned)f32_to_bf16(acc[0]) | ((unsigned)f32_to_bf16(acc[1]) << 16);
        o.y = (unsigned)f32_to_bf16(acc[2]) | ((unsigned)f32_to_bf16(acc[3]) << 16);
      }
      *reinterpret_cast<V*>(yp + xx * C) = o;
      // keep the unrolled steps apart (x and y are deliberately NOT __restrict__: with no-alias loads the straight-line bf16 body
      // gets all 48 loads of a segment hoisted to its top — 250 registers, or 140 spilled at four waves per SIMD)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ------------------------------------------------------------------------------------------------------ bilinear resize
// nn.functional.interpolate(mode='bilinear', align_corners=False): src = max(0, (dst + 0.5) * in / out - 0.5)
__device__ __forceinline__ void bilinear_src(int dst, int in, int out, int& i0, int& i1, float& l1) {
  float s = ((float)dst + 0.5f) * ((float)in / (float)out) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// NHWC [B][h][w][C] -> a channel slice of NHWC [B][H][W][ld] (the decode head's concatenation buffer)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int h, int w, int C, int H,
                                                            int W, int ld) {
  constexpr int CH = Elem<T>::CH;
  const int nch = C / CH;
  const long total = (long)B * H * W * nch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % nch);
    const long p = i / nch;
    const int ox = (int)(p % W), oy = (int)((p / W) % H);
    const long b = p / ((long)W * H);
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(oy, h, H, y0, y1, ly);
    bilinear_src(ox, w, W, x0, x1, lx);
    float a[CH], bq[CH], cq[CH], d[CH], o[CH];
    const T* base = x + b * h * w * C + (long)c * CH;
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y0 * w + x0) * C), a);
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y0 * w + x1) * C), bq);
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y1 * w + x0) * C), cq);
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y1 * w + x1) * C), d);
#pragma unroll
    for (int e = 0; e < CH; ++e)
      o[e] = (1.f - ly) * ((1.f - lx) * a[e] + lx * bq[e]) + ly * ((1.f - lx) * cq[e] + lx * d[e]);
    *reinterpret_cast<uint4*>(y + p * ld + (long)c * CH) = f_to_chunk<T>(o);
  }
}

// fp32 NCHW [B][C][h][w] -> fp32 NCHW [B][C][H][W] (the logits, x4).  A workgroup owns RPB output rows of one plane and a thread
// four consecutive outputs (one 16-byte store); all index arithmetic is 32-bit.  (First version: one output per thread from a
// flat 64-bit index, two 64-bit divisions per element: 1.2 TB/s.)  W % 4 == 0, H % RPB == 0.
template <int RPB>
__global__ __launch_bounds__(256) void bilinear_nchw_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int H, int W) {
  const int W4 = W >> 2;
  const long row0 = (long)blockIdx.x * RPB;             // first output row of this workgroup, over all planes
  const int pl = (int)(row0 / H), oy0 = (int)(row0 - (long)pl * H);
  const float* p = x + (long)pl * h * w;
  float* yo = y + row0 * W;
  for (int it = threadIdx.x; it < RPB * W4; it += 256) {
    const int r = it / W4, k = it - r * W4;
    int y0, y1;
    float ly;
    bilinear_src(oy0 + r, h, H, y0, y1, ly);
    const float* r0 = p + y0 * w;
    const float* r1 = p + y1 * w;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float lx;
      bilinear_src(4 * k + j, w, W, x0, x1, lx);
      o[j] = (1.f - ly) * ((1.f - lx) * r0[x0] + lx * r0[x1]) + ly * ((1.f - lx) * r1[x0] + lx * r1[x1]);
    }
    *reinterpret_cast<float4*>(yo + (long)r * W + 4 * k) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// the same, any shape: one output per thread
__global__ __launch_bounds__(256) void bilinear_nchw_f32_any_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int h, int w,
                                                                    int H, int W) {
  const long total = planes * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % W), oy = (int)((i / W) % H);
    const long pl = i / ((long)W * H);
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(oy, h, H, y0, y1, ly);
    bilinear_src(ox, w, W, x0, x1, lx);
    const float* p = x + pl * h * w;
    y[i] = (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * ((1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
  }
}

// ------------------------------------------------------------------------------------------------- decode-head helpers
// The decode head as the library writes it — Linear to D channels per stage, bilinear upsample to 1/4 resolution, concatenation,
// 1x1 fuse convolution over 4 D channels, BatchNorm, ReLU — spends 3/4 of the model's FLOPs in the fuse product at full 1/4
// resolution.  Everything before the BatchNorm is linear, and a per-channel bilinear resize commutes with a per-pixel channel
// mix, so  fuse(cat_i up(P_i f_i + b_i)) = sum_i up((F_i P_i) f_i) + F [b_3 | b_2 | b_1 | b_0]  with F_i the i-th column block of
// the fuse weight: the products run at each stage's OWN resolution with pre-multiplied D x C_i weights (25x fewer FLOPs for
// MiT-B2 at 512x512), and one pass adds the upsampled terms, applies the folded BatchNorm and the ReLU.

// fp32 [rows][ld] columns [col0, col0 + ncols) -> T [rows][ncols]
template <typename T>
__global__ __launch_bounds__(256) void slice_cols_kernel(const float* __restrict__ src, int ld, int col0, int ncols, long rows, T* __restrict__ dst) {
  constexpr int CH = Elem<T>::CH;
  const int nch = ncols / CH;
  const long total = rows * nch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / nch;
    const int c = (int)(i % nch) * CH;
    float f[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) f[e] = src[r * ld + col0 + c + e];
    *reinterpret_cast<uint4*>(dst + r * ncols + c) = f_to_chunk<T>(f);
  }
}

// shift2[o] = shift[o] + scale[o] * sum_k Wf[o][k] * bcat[k],  bcat = the four projection biases in concatenation order
// (stage 3 first); one wave per output channel, fp32
__global__ __launch_bounds__(256) void fuse_bias_kernel(const float* __restrict__ wf, int D, const float* __restrict__ b3, const float* __restrict__ b2,
                                                        const float* __restrict__ b1, const float* __restrict__ b0, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ shift2) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= D) return;
  const float* bs[4] = {b3, b2, b1, b0};
  float s = 0.f;
  for (int k = lane; k < 4 * D; k += 64) s = fmaf(wf[(long)o * 4 * D + k], bs[k / D][k % D], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) shift2[o] = fmaf(scale[o], s, shift[o]);
}

// z = relu(scale * (g0 + up(g1) + up(g2) + up(g3)) + shift2); g0 at [B][H][W][D], g_i at [B][H >> i][W >> i][D]
template <typename T>
__global__ __launch_bounds__(256) void upsample_sum_bn_relu_kernel(const T* __restrict__ g0, const T* __restrict__ g1, const T* __restrict__ g2,
                                                                   const T* __restrict__ g3, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift2, T* __restrict__ z, int B, int H, int W, int D) {
  constexpr int CH = Elem<T>::CH;
  const int nch = D / CH;
  const long total = (long)B * H * W * nch;
  const T* gs[3] = {g1, g2, g3};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % nch);
    const long p = i / nch;
    const int ox = (int)(p % W), oy = (int)((p / W) % H);
    const long b = p / ((long)W * H);
    float acc[CH];
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(g0 + p * D + (long)c * CH), acc);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int h = H >> (k + 1), w = W >> (k + 1);
      int y0, y1, x0, x1;
      float ly, lx;
      bilinear_src(oy, h, H, y0, y1, ly);
      bilinear_src(ox, w, W, x0, x1, lx);
      const T* base = gs[k] + b * h * w * D + (long)c * CH;
      float a[CH], bq[CH], cq[CH], d[CH];
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y0 * w + x0) * D), a);
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y0 * w + x1) * D), bq);
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y1 * w + x0) * D), cq);
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(base + ((long)y1 * w + x1) * D), d);
#pragma unroll
      for (int e = 0; e < CH; ++e) acc[e] += (1.f - ly) * ((1.f - lx) * a[e] + lx * bq[e]) + ly * ((1.f - lx) * cq[e] + lx * d[e]);
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = fmaxf(fmaf(acc[e], scale[c * CH + e], shift2[c * CH + e]), 0.f);
    *reinterpret_cast<uint4*>(z + p * D + (long)c * CH) = f_to_chunk<T>(acc);
  }
}

// ----------------------------------------------------------------------------------------- decode head in ONE kernel (bf16)
// The restructured head still moved 3.5 GB per 32 windows: g_0 = W_0 f_0 (768 channels at 1/4 resolution) written and re-read,
// z written and re-read by the classifier.  Everything after the per-stage products is local to a pixel tile, so one workgroup
// takes a 16 x 8 tile of the 1/4-resolution grid through all of it, 64 of the D channels at a time, entirely on the matrix pipe:
//   acc[ch][px]  = sum_k   W_0[ch][k] f_0[px][k]                 (K = 64: the stage-0 product, never stored)
//                + sum_src G[src][ch] Wint[px][src]              (K = 96: the bilinear x2 / x4 / x8 upsamples of g_1, g_2, g_3 AS A
//                                                                 PRODUCT with a constant 128 x 96 interpolation matrix over the
//                                                                 60 + 24 + 12 source pixels under the tile; its entries a b / 4^s
//                                                                 with odd a, b < 2^(s+1) are exact in bf16, the accumulation is fp32)
//   z            = relu(scale[ch] acc + shift2[ch])  -> bf16, in the accumulator registers
//   logits[cls][px] += sum_ch Wc[cls][ch] z[ch][px]              (the accumulator tiles ARE the B operand of the classifier product:
//                                                                 cdna_hip_programming.md, 'An accumulator tile as the next MFMA's
//                                                                 operand' — the k order inside a step is permuted, Wc is read with
//                                                                 the same permutation)
// Source pixels outside the image are the clamped ones (replicate), which is exactly align_corners = False's edge rule: the
// two taps of a clamped coordinate coincide and their weights sum to one.  G chunks go global -> registers -> LDS one chunk
// ahead and come back as A fragments through the transposing read (rows = source pixels, columns = channels).
// HBM traffic per 32 windows: f_0 67 MB + g_1..3 268 MB + logits 40 MB, against 3.5 GB.
constexpr int HT_W = 16, HT_H = 8, HT_PX = HT_W * HT_H, HT_SRC = 96, HT_ROW = 144 /* bytes per source pixel in LDS: 64 bf16 + pad */;

// source pixel s of the patch -> stage (1..3), row, column inside that stage's patch (6 x 10, 4 x 6, 3 x 4)
__host__ __device__ inline void head_src(int s, int& st, int& sy, int& sx) {
  if (s < 60) { st = 1; sy = s / 10; sx = s % 10; }
  else if (s < 84) { st = 2; sy = (s - 60) / 6; sx = (s - 60) % 6; }
  else { st = 3; sy = (s - 84) / 4; sx = (s - 84) % 4; }
}

// Wint[px][src]: product of the two 1-D weights of an (unclamped) align_corners = False resize by 2^st at tile-local output
// coordinate o; the patch of a stage starts one source pixel before the tile's first
__global__ __launch_bounds__(256) void head_wint_kernel(bf16_t* __restrict__ wint) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HT_PX * HT_SRC) return;
  const int px = i / HT_SRC, s = i % HT_SRC, ox = px % HT_W, oy = px / HT_W;
  int st, sy, sx;
  head_src(s, st, sy, sx);
  const float inv = 1.f / (float)(1 << st);
  auto w1 = [&](int o, int idx) {
    const float sc = ((float)o + 0.5f) * inv - 0.5f, fl = floorf(sc), l1 = sc - fl;
    const int i0 = (int)fl + 1;
    return idx == i0 ? 1.f - l1 : idx == i0 + 1 ? l1 : 0.f;
  };
  wint[i] = f32_to_bf16(w1(oy, sy) * w1(ox, sx));
}

// LDS image of one channel chunk: the source patch [96][64 + pad], W_0 rows [64 ch][64 k + pad], classifier rows [32 cls][64 ch + pad],
// scale / shift2 [2][64] fp32 — everything the chunk's MFMAs read, staged one chunk ahead (global -> registers -> LDS).  (First
// version: the weight fragments straight from global memory inside the chunk loop, 481 us: every chunk waited for L2 round trips.)
constexpr int HT_W0 = HT_SRC * HT_ROW, HT_WC = HT_W0 + 64 * HT_ROW, HT_SS = HT_WC + 32 * HT_ROW, HT_BUF = HT_SS + 512;

__global__ __launch_bounds__(256) void head_fused_kernel(const bf16_t* __restrict__ f0, const bf16_t* __restrict__ w0, const bf16_t* __restrict__ g1,
                                                         const bf16_t* __restrict__ g2, const bf16_t* __restrict__ g3,
                                                         const bf16_t* __restrict__ wint, const float* __restrict__ scale,
                                                         const float* __restrict__ shift2, const bf16_t* __restrict__ wc /* [32][D] */,
                                                         const float* __restrict__ bc, float* __restrict__ out, int B, int H, int W, int D,
                                                         int labels) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][HT_BUF];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
  const int tiles_x = W / HT_W, tiles_y = H / HT_H;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const int x0 = tx * HT_W, y0 = ty * HT_H;
  // ---- staging items of a chunk: patch 96 x 8 (3 per thread), W_0 64 x 8 (2), classifier 32 x 8 (1), scale / shift 32 (threads < 32)
  const bf16_t* sp[6];
  int so[6], sstep[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = t + 256 * k, src = i >> 3, part = i & 7;
    int st, sy, sx;
    head_src(src, st, sy, sx);
    const int hs = H >> st, ws = W >> st;
    int cy = (y0 >> st) - 1 + sy, cx = (x0 >> st) - 1 + sx;
    cy = cy < 0 ? 0 : cy > hs - 1 ? hs - 1 : cy;
    cx = cx < 0 ? 0 : cx > ws - 1 ? ws - 1 : cx;
    const bf16_t* gs = st == 1 ? g1 : st == 2 ? g2 : g3;
    sp[k] = gs + (((long)b * hs + cy) * ws + cx) * D + part * 8;
    so[k] = src * HT_ROW + part * 16;
    sstep[k] = 64;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = t + 256 * k, row = i >> 3, part = i & 7;
    sp[3 + k] = w0 + (long)row * 64 + part * 8;
    so[3 + k] = HT_W0 + row * HT_ROW + part * 16;
    sstep[3 + k] = 64 * 64;
  }
  {
    const int row = t >> 3, part = t & 7;
    sp[5] = wc + (long)row * D + part * 8;
    so[5] = HT_WC + row * HT_ROW + part * 16;
    sstep[5] = 64;
  }
  const float* ssp = (t < 16 ? scale + 4 * t : shift2 + 4 * (t & 15));   // threads 0..15 scale, 16..31 shift2
  u32x4 stg[6];
  float4 sst = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage_load = [&](int c) {
#pragma unroll
    for (int k = 0; k < 6; ++k) stg[k] = *reinterpret_cast<const u32x4*>(sp[k] + (long)c * sstep[k]);
    if (t < 32) sst = *reinterpret_cast<const float4*>(ssp + c * 64);
  };
  auto stage_store = [&](unsigned char* buf) {
#pragma unroll
    for (int k = 0; k < 6; ++k) *reinterpret_cast<u32x4*>(buf + so[k]) = stg[k];
    if (t < 32) *reinterpret_cast<float4*>(buf + HT_SS + t * 16) = sst;
  };
  stage_load(0);
  // ---- constant B fragments of this wave's two pixel rows (N tiles): f_0 (K = 64) and the interpolation matrix (K = 96)
  u32x4 bf0[2][2], bw[2][3];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int oy = 2 * wave + n;
    const bf16_t* fp = f0 + (((long)b * H + y0 + oy) * W + x0 + lr) * 64 + 8 * g;
    bf0[n][0] = *reinterpret_cast<const u32x4*>(fp);
    bf0[n][1] = *reinterpret_cast<const u32x4*>(fp + 32);
    const bf16_t* wp = wint + (oy * HT_W + lr) * HT_SRC + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) bw[n][ks] = *reinterpret_cast<const u32x4*>(wp + ks * 32);
  }
  stage_store(smem[0]);
  __syncthreads();
  f32x4_t la[2][2];
#pragma unroll
  for (int cm = 0; cm < 2; ++cm)
#pragma unroll
    for (int n = 0; n < 2; ++n) la[cm][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const int q = lr >> 2, p = lr & 3;
  const int nchunks = D / 64;
  for (int c = 0; c < nchunks; ++c) {
    const unsigned char* buf = smem[c & 1];
    if (c + 1 < nchunks) stage_load(c + 1);
    unsigned zp[2][4][2];   // [pixel row][16-channel tile][packed pairs]: z in bf16
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const unsigned char* wr = buf + HT_W0 + (mt * 16 + lr) * HT_ROW + 16 * g;
      const u32x4 aw0 = *reinterpret_cast<const u32x4*>(wr), aw1 = *reinterpret_cast<const u32x4*>(wr + 64);
      u32x4 ag[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const unsigned char* a = buf + (ks * 32 + 8 * g + q) * HT_ROW + mt * 32 + 8 * p;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a + 4 * HT_ROW));
        ag[ks].x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
        ag[ks].y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
        ag[ks].z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
        ag[ks].w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
      }
      const float4 sc4 = *reinterpret_cast<const float4*>(buf + HT_SS + (mt * 16 + 4 * g) * 4);
      const float4 sh4 = *reinterpret_cast<const float4*>(buf + HT_SS + 256 + (mt * 16 + 4 * g) * 4);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aw0), __builtin_bit_cast(bf16x8_t, bf0[n][0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aw1), __builtin_bit_cast(bf16x8_t, bf0[n][1]), acc, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ag[ks]), __builtin_bit_cast(bf16x8_t, bw[n][ks]), acc, 0, 0, 0);
        const float z0 = fmaxf(fmaf(acc[0], sc4.x, sh4.x), 0.f), z1 = fmaxf(fmaf(acc[1], sc4.y, sh4.y), 0.f);
        const float z2 = fmaxf(fmaf(acc[2], sc4.z, sh4.z), 0.f), z3 = fmaxf(fmaf(acc[3], sc4.w, sh4.w), 0.f);
        zp[n][mt][0] = (unsigned)f32_to_bf16(z0) | ((unsigned)f32_to_bf16(z1) << 16);
        zp[n][mt][1] = (unsigned)f32_to_bf16(z2) | ((unsigned)f32_to_bf16(z3) << 16);
      }
    }
    // ---- classifier: K step s = channels 32 s .. 32 s + 31 of the chunk; slot (g, j < 4) <- tile 2 s row 4 g + j, (g, j >= 4) <- tile 2 s + 1
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int cm = 0; cm < 2; ++cm) {
        const unsigned char* cp = buf + HT_WC + (cm * 16 + lr) * HT_ROW + (32 * s2 + 4 * g) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(cp), hi = *reinterpret_cast<const uint2*>(cp + 32);
        const u32x4 af = u32x4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const u32x4 zf = u32x4{zp[n][2 * s2][0], zp[n][2 * s2][1], zp[n][2 * s2 + 1][0], zp[n][2 * s2 + 1][1]};
          la[cm][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, zf), la[cm][n], 0, 0, 0);
        }
      }
    }
    if (c + 1 < nchunks) stage_store(smem[(c + 1) & 1]);
    __syncthreads();
  }
  // ---- logits^T[cls = 16 cm + 4 g + i][pixel (row 2 wave + n, column lr)] -> fp32 NCHW
#pragma unroll
  for (int cm = 0; cm < 2; ++cm)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cls = cm * 16 + 4 * g + i;
      if (cls < labels) {
        const float bias = bc ? bc[cls] : 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
          out[(((long)b * labels + cls) * H + y0 + 2 * wave + n) * W + x0 + lr] = la[cm][n][i] + bias;
      }
    }
}

// ---------------------------------------------------------------------------------- Mix-FFN of a block in ONE kernel (bf16)
// LayerNorm -> fc1 (C -> 4C) -> depth-wise 3x3 + GELU -> fc2 (4C -> C) -> + residual moves, as separate passes, the 4C-wide
// intermediate through HBM four times (stage 0 at B = 32: 4 x 268 MB of the block's 2 GB).  All of it is local to a pixel tile
// plus a one-pixel halo, so a workgroup takes an 8 x 8 tile of the token grid through the whole chain, 64 hidden channels at a
// time, and the intermediate never leaves the CU:
//   X halo tile (10 x 10 pixels x C) -> LDS, LayerNorm in place (two threads per pixel)
//   per chunk of 64 hidden channels:  H1[ch][px] = W1 XN^T (+ bias, zero outside the image: the conv pads ITS input) -> LDS [px][64]
//                                     A2 = gelu(dwconv3x3(H1)) on the vector pipe, sliding 3 x 6 windows                -> LDS [px][64]
//                                     OUT[c][px] += W2[:, chunk] A2^T  (accumulators live across the chunks)
//   x_out = OUT + bias + x
// fc1 is recomputed on the halo (100 / 64 pixels); HBM traffic is x once (x 1.56) + x_out.  The chunk's weights (W1 rows, W2
// columns, depth-wise taps, fc1 bias) go global -> registers -> LDS ONE CHUNK AHEAD, once per workgroup.  (First version: every
// wave loaded its weight fragments straight from global memory inside the chunk loop — four copies of the same 30 KB per chunk
// through the texture path and an L2 round trip per phase: 355 us per stage-0 block, 0.83 ms of a forward's 2.3 ms of fused FFN.)
// The output is a DIFFERENT buffer than x: neighbouring tiles read each other's halo pixels.
template <int C>
struct FfnCfg {
  static constexpr int XROW = C * 2 + 16, HPX = 100, HROWS = 112, ROW = 144;
  static constexpr int OFF_H1 = HROWS * XROW, OFF_A2 = OFF_H1 + HROWS * ROW, OFF_W1 = OFF_A2 + 64 * ROW, OFF_B1 = OFF_W1 + 64 * XROW,
                       OFF_W2 = OFF_B1 + 256, OFF_DW = OFF_W2 + C * ROW, SMEM = OFF_DW + 160 * 16;
};

// dw weights [4C][3][3] + bias [4C] -> [4C / 4][10][4]: per group of four channels nine tap vectors and the bias vector
__global__ __launch_bounds__(256) void ffn_dw_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out, int nch) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nch * 10) return;
  const int e = i & 3, tp = (i >> 2) % 10, grp = i / 40;
  const int ch = grp * 4 + e;
  out[i] = tp < 9 ? w[ch * 9 + tp] : b[ch];
}

// 512 threads: a tile's serial chain (halo load, LayerNorm, then per chunk fc1 -> barrier -> depth-wise -> barrier -> fc2) is what
// bounds the kernel at two workgroups per CU, so eight waves split every phase: fc1 by halo pixel tile (7 of the 8 waves), the
// depth-wise phase two pixels per thread, fc2 by (pixel tile, K half) with the halves added through LDS once per tile.
// (Four waves per tile: 261 us per stage-0 block.)
constexpr int FFN_NT = 512;

template <int C>
__global__ __launch_bounds__(FFN_NT) __attribute__((amdgpu_waves_per_eu(C == 64 ? 4 : 2, 8))) void ffn_fused_kernel(const bf16_t* __restrict__ x, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                           const bf16_t* __restrict__ w1 /* [4C][C] */, const float* __restrict__ b1,
                                                           const float* __restrict__ dwp /* [C][10][4] */, const bf16_t* __restrict__ w2 /* [C][4C] */,
                                                           const float* __restrict__ b2, bf16_t* __restrict__ out, int B, int H, int W, float eps,
                                                           const float* __restrict__ ln2_g, const float* __restrict__ ln2_b,
                                                           bf16_t* __restrict__ out_ln /* optional: LayerNorm(out) with ln2_*, the next consumer's input */) {
  using Cfg = FfnCfg<C>;
  constexpr int NT = FFN_NT;
  constexpr int XROW = Cfg::XROW, ROW = Cfg::ROW, CPR = C / 8, KS1 = C / 32, MT2 = C / 16, NCHUNK = 4 * C / 64, HID = 4 * C;
  constexpr int NW = (64 * CPR + NT - 1) / NT;   // 16-byte items per thread of a W1 / W2 chunk (64 rows x C, C rows x 64)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;
  unsigned char* h1 = smem + Cfg::OFF_H1;
  unsigned char* a2 = smem + Cfg::OFF_A2;
  unsigned char* w1s = smem + Cfg::OFF_W1;
  unsigned char* b1s = smem + Cfg::OFF_B1;
  unsigned char* w2s = smem + Cfg::OFF_W2;
  unsigned char* dws = smem + Cfg::OFF_DW;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 15, g = lane >> 4;
  const int tiles_x = W / 8, tiles_y = H / 8;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const int x0 = tx * 8, y0 = ty * 8;
  const bf16_t* xb = x + (long)b * H * W * C;
  // ---- weight staging: registers of this thread for the next chunk (64 * CPR = 512 or 1024 items of 16 bytes per matrix)
  u32x4 rw1[NW], rw2[NW], rdw = u32x4{0u, 0u, 0u, 0u};
  auto ld_w1 = [&](int hc) {   // + the chunk's fc1 bias (threads 160 .. 175) and depth-wise taps (threads < 160) ride along
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int i = t + NT * k, row = i / CPR, part = i - row * CPR;
      rw1[k] = *reinterpret_cast<const u32x4*>(w1 + (long)(hc * 64 + row) * C + part * 8);
    }
    if (t < 160) rdw = *reinterpret_cast<const u32x4*>(dwp + ((long)hc * 160 + t) * 4);
    else if (t < 176) rdw = *reinterpret_cast<const u32x4*>(b1 + hc * 64 + (t - 160) * 4);
  };
  auto st_w1 = [&]() {         // W1 rows + bias (read by fc1)
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int i = t + NT * k, row = i / CPR, part = i - row * CPR;
      *reinterpret_cast<u32x4*>(w1s + row * XROW + part * 16) = rw1[k];
    }
    if (t >= 160 && t < 176) *reinterpret_cast<u32x4*>(b1s + (t - 160) * 16) = rdw;
  };
  auto st_dw = [&]() {         // depth-wise taps (read by the depth-wise phase)
    if (t < 160) *reinterpret_cast<u32x4*>(dws + t * 16) = rdw;
  };
  auto ld_w2 = [&](int hc) {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int i = t + NT * k, row = i >> 3, part = i & 7;
      rw2[k] = *reinterpret_cast<const u32x4*>(w2 + (long)row * HID + hc * 64 + part * 8);
    }
  };
  auto st_w2 = [&]() {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int i = t + NT * k, row = i >> 3, part = i & 7;
      *reinterpret_cast<u32x4*>(w2s + row * ROW + part * 16) = rw2[k];
    }
  };
  ld_w1(0);
  ld_w2(0);
  // ---- X halo tile -> LDS (zeros outside the image; rows 100 .. 111 are never consumed but must be finite); all loads of a
  // thread are issued before the first LDS write
  {
    constexpr int NIT = (Cfg::HROWS * CPR + NT - 1) / NT;
    u32x4 v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = t + NT * k, hp = i / CPR, part = i - hp * CPR;
      const int hy = hp / 10, hx = hp - hy * 10;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = hp < Cfg::HPX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      const u32x4 ld = *reinterpret_cast<const u32x4*>(xb + (ok ? ((long)gy * W + gx) * C + part * 8 : 0));
      v[k] = ok ? ld : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = t + NT * k, hp = i / CPR, part = i - hp * CPR;
      if (i < Cfg::HROWS * CPR) *reinterpret_cast<u32x4*>(xs + hp * XROW + part * 16) = v[k];
    }
  }
  st_w1();
  st_dw();
  __syncthreads();
  // ---- LayerNorm in place, FOUR threads per halo pixel, each a quarter of the channels (mean, biased variance of the centred
  // values, eps inside the root); the quarters meet through lane shuffles
  {
    const int hp = t >> 2, qtr = t & 3;
    const bool live = hp < Cfg::HPX;
    unsigned char* row = xs + (live ? hp : 0) * XROW + qtr * (C / 4) * 2;
    constexpr int HC = CPR / 4;   // 16-byte chunks per quarter row
    float f[HC][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      chunk_to_f<bf16_t>(*reinterpret_cast<const uint4*>(row + c * 16), f[c]);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += f[c][e];
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float mean = sum / (float)C;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < HC; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = f[c][e] - mean; q = fmaf(d, d, q); }
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    const float rstd = 1.f / sqrtf(q / (float)C + eps);
    if (live) {
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        const int c0 = qtr * (C / 4) + c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[c][e] = fmaf((f[c][e] - mean) * rstd, ln_g[c0 + e], ln_b[c0 + e]);
        *reinterpret_cast<uint4*>(row + c * 16) = f_to_chunk<bf16_t>(f[c]);
      }
    }
  }
  // ---- per-lane constant: does this wave's fc1 pixel tile (halo rows 16 wave .. + 15, waves 0 .. 6) hold an in-image pixel here
  bool inimg;
  {
    const int hp = wave * 16 + lr, hy = hp / 10, hx = hp - hy * 10;
    inimg = hp < Cfg::HPX && (unsigned)(y0 - 1 + hy) < (unsigned)H && (unsigned)(x0 - 1 + hx) < (unsigned)W;
  }
  // depth-wise item of this thread: channel group gq (4 channels), row py, quarter xq (2 pixels)
  const int gq = t & 15, seg = t >> 4, py = seg >> 2, xq = seg & 3;
  // fc2 item of this wave: pixel tile pt2 (16 pixels), K half kh of the chunk's 64 hidden channels
  const int pt2 = wave & 3, kh = wave >> 2;
  f32x4_t acc2[MT2];
#pragma unroll
  for (int m = 0; m < MT2; ++m) acc2[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int hc = 0; hc < NCHUNK; ++hc) {
    // rw2 holds W2's chunk hc (loaded a chunk ago); the next chunk's W1 / bias / taps start their trip now
    if (hc + 1 < NCHUNK) ld_w1(hc + 1);
    // ---- fc1 on the halo: H1[ch][px], this wave's pixel tile x the chunk's four 16-channel tiles
    if (wave < 7) {   // wave-uniform
      const int pt = wave;
      u32x4 bx[KS1];
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) bx[ks] = *reinterpret_cast<const u32x4*>(xs + (pt * 16 + lr) * XROW + (ks * 32 + 8 * g) * 2);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const u32x4 wf = *reinterpret_cast<const u32x4*>(w1s + (ct * 16 + lr) * XROW + (ks * 32 + 8 * g) * 2);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, bx[ks]), a, 0, 0, 0);
        }
        const float4 bb = *reinterpret_cast<const float4*>(b1s + (ct * 16 + 4 * g) * 4);
        const float v0 = inimg ? a[0] + bb.x : 0.f, v1 = inimg ? a[1] + bb.y : 0.f;
        const float v2 = inimg ? a[2] + bb.z : 0.f, v3 = inimg ? a[3] + bb.w : 0.f;
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(v0) | ((unsigned)f32_to_bf16(v1) << 16);
        pk.y = (unsigned)f32_to_bf16(v2) | ((unsigned)f32_to_bf16(v3) << 16);
        *reinterpret_cast<uint2*>(h1 + (pt * 16 + lr) * ROW + (ct * 16 + 4 * g) * 2) = pk;
      }
    }
    __syncthreads();                                 // H1 complete; every wave is past fc1 (W1 / bias image free), fc2 of hc - 1 (W2 free)
    st_w2();                                         // W2 chunk hc for this chunk's fc2
    if (hc + 1 < NCHUNK) ld_w2(hc + 1);
    // ---- depth-wise 3 x 3 + GELU: two outputs along x from a 3 x 4 window of H1 (halo rows py .. py + 2)
    {
      float4 dw[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) dw[k] = *reinterpret_cast<const float4*>(dws + (gq * 10 + k) * 16);
      float o[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i) { o[i][0] = dw[9].x; o[i][1] = dw[9].y; o[i][2] = dw[9].z; o[i][3] = dw[9].w; }
#pragma unroll
      for (int r = 0; r < 3; ++r) {   // one window row at a time (the whole 3 x 4 window in registers spills at four waves per SIMD);
                                      // per output the taps still accumulate row-major: r, then q
        float win[4][4];
#pragma unroll
        for (int col = 0; col < 4; ++col) {
          const uint2 v = *reinterpret_cast<const uint2*>(h1 + ((py + r) * 10 + xq * 2 + col) * ROW + gq * 8);
          win[col][0] = __uint_as_float(v.x << 16); win[col][1] = __uint_as_float(v.x & 0xffff0000u);
          win[col][2] = __uint_as_float(v.y << 16); win[col][3] = __uint_as_float(v.y & 0xffff0000u);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int qq = 0; qq < 3; ++qq) {
            const float4 wv = dw[r * 3 + qq];
            const float* xv = win[i + qq];
            o[i][0] = fmaf(xv[0], wv.x, o[i][0]); o[i][1] = fmaf(xv[1], wv.y, o[i][1]);
            o[i][2] = fmaf(xv[2], wv.z, o[i][2]); o[i][3] = fmaf(xv[3], wv.w, o[i][3]);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[i][e] = gelu_erf<bf16_t>(o[i][e]);
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(o[i][0]) | ((unsigned)f32_to_bf16(o[i][1]) << 16);
        pk.y = (unsigned)f32_to_bf16(o[i][2]) | ((unsigned)f32_to_bf16(o[i][3]) << 16);
        *reinterpret_cast<uint2*>(a2 + (py * 8 + xq * 2 + i) * ROW + gq * 8) = pk;
      }
    }
    if (hc + 1 < NCHUNK) st_w1();                    // next chunk's W1 rows + bias (their readers start behind the barrier below)
    __syncthreads();                                 // A2 and W2 complete; every wave is past the depth-wise phase (tap image free)
    if (hc + 1 < NCHUNK) st_dw();
    // ---- fc2: OUT[c][px] += W2[:, chunk K half] A2^T, this wave's 16 pixels and 32 of the chunk's hidden channels
    {
      const u32x4 ba = *reinterpret_cast<const u32x4*>(a2 + (pt2 * 16 + lr) * ROW + (kh * 32 + 8 * g) * 2);
#pragma unroll
      for (int m = 0; m < MT2; ++m) {
        const u32x4 wf = *reinterpret_cast<const u32x4*>(w2s + (m * 16 + lr) * ROW + (kh * 32 + 8 * g) * 2);
        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), __builtin_bit_cast(bf16x8_t, ba), acc2[m], 0, 0, 0);
      }
    }
    // (no barrier here: the next writers of H1 / A2 / W2 sit behind the next chunk's first barrier)
  }
  // ---- the two K halves of a pixel tile meet: waves 4 .. 7 hand their sums to waves 0 .. 3 through LDS (the X image is free)
  __syncthreads();
  float* xch = reinterpret_cast<float*>(smem);   // [4 pixel tiles][MT2][64 lanes] float4
  if (kh == 1) {
#pragma unroll
    for (int m = 0; m < MT2; ++m) *reinterpret_cast<f32x4_t*>(xch + ((pt2 * MT2 + m) * 64 + lane) * 4) = acc2[m];
  }
  __syncthreads();
  // ---- x_out = OUT + b2 + x: lane holds channels 16 m + 4 g .. + 3 of pixel (pt2 * 16 + lr)
  if (kh == 0) {   // wave-uniform
    const int p = pt2 * 16 + lr, gy = y0 + (p >> 3), gx = x0 + (p & 7);
    const long base = (((long)b * H + gy) * W + gx) * C;
    float xo[MT2][4];
#pragma unroll
    for (int m = 0; m < MT2; ++m) {
      const f32x4_t other = *reinterpret_cast<const f32x4_t*>(xch + ((pt2 * MT2 + m) * 64 + lane) * 4);
      const int c = m * 16 + 4 * g;
      const uint2 rv = *reinterpret_cast<const uint2*>(x + base + c);
      const float4 bb = *reinterpret_cast<const float4*>(b2 + c);
      const float r0 = __uint_as_float(rv.x << 16), r1 = __uint_as_float(rv.x & 0xffff0000u);
      const float r2 = __uint_as_float(rv.y << 16), r3 = __uint_as_float(rv.y & 0xffff0000u);
      uint2 pk;
      pk.x = (unsigned)f32_to_bf16(acc2[m][0] + other[0] + bb.x + r0) | ((unsigned)f32_to_bf16(acc2[m][1] + other[1] + bb.y + r1) << 16);
      pk.y = (unsigned)f32_to_bf16(acc2[m][2] + other[2] + bb.z + r2) | ((unsigned)f32_to_bf16(acc2[m][3] + other[3] + bb.w + r3) << 16);
      *reinterpret_cast<uint2*>(out + base + c) = pk;
      // the stored (rounded) values: what a separate LayerNorm pass would read back
      xo[m][0] = __uint_as_float(pk.x << 16); xo[m][1] = __uint_as_float(pk.x & 0xffff0000u);
      xo[m][2] = __uint_as_float(pk.y << 16); xo[m][3] = __uint_as_float(pk.y & 0xffff0000u);
    }
    // ---- the LayerNorm that follows the block (next block's layernorm_before, or the stage's final norm): the C channels of a
    // pixel sit in this lane's registers and in the three lanes lr + 16, + 32, + 48 — two shuffles per reduction
    if (out_ln) {   // workgroup-uniform
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < MT2; ++m) sum += (xo[m][0] + xo[m][1]) + (xo[m][2] + xo[m][3]);
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float mean = sum / (float)C;
      float q2 = 0.f;
#pragma unroll
      for (int m = 0; m < MT2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xo[m][e] - mean; q2 = fmaf(d, d, q2); }
      q2 += __shfl_xor(q2, 16);
      q2 += __shfl_xor(q2, 32);
      const float rstd = 1.f / sqrtf(q2 / (float)C + eps);
#pragma unroll
      for (int m = 0; m < MT2; ++m) {
        const int c = m * 16 + 4 * g;
        const float4 gg = *reinterpret_cast<const float4*>(ln2_g + c), be = *reinterpret_cast<const float4*>(ln2_b + c);
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(fmaf((xo[m][0] - mean) * rstd, gg.x, be.x)) | ((unsigned)f32_to_bf16(fmaf((xo[m][1] - mean) * rstd, gg.y, be.y)) << 16);
        pk.y = (unsigned)f32_to_bf16(fmaf((xo[m][2] - mean) * rstd, gg.z, be.z)) | ((unsigned)f32_to_bf16(fmaf((xo[m][3] - mean) * rstd, gg.w, be.w)) << 16);
        *reinterpret_cast<uint2*>(out_ln + base + c) = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ attention
// softmax(Q K^T / 8) V for one head of 64 channels, keys / values = the spatially reduced sequence (Nk <= 256 tokens), whole
// in LDS: K as [key][64] rows, V transposed to [64][key].  The scores are computed TRANSPOSED, S^T = K Q^T, so that a lane
// holds one query's scores for 4 keys per 16-key tile: the softmax reduces in registers plus two shuffles, and the
// exponentiated scores ARE the B operand of the second product O^T = V^T P^T (cdna_hip_programming.md §3, 'An accumulator
// tile as the next MFMA's operand': the k order inside a step is permuted, so V^T is read with the same permutation).
// A workgroup of 4 waves walks QB query blocks of 64 queries (16 per wave).
template <typename T> struct AttCfg;
template <> struct AttCfg<bf16_t> {
  static constexpr int KROW = 128 + 16;                     // bytes per key row of K (64 bf16 + pad)
  static __host__ __device__ int vrow(int nk) { return nk * 2 + 16; }   // bytes per channel row of V^T
};
template <> struct AttCfg<float> {
  static constexpr int KROW = 256 + 16;
  static __host__ __device__ int vrow(int nk) { return nk * 4 + 16; }
};

template <typename T, int NKT /* key tiles of 16 */>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        T* __restrict__ out, int N, int Nk, int hidden, int qblocks, int kv_ld) {
  constexpr int CH = Elem<T>::CH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int KROW = AttCfg<T>::KROW, VROW = AttCfg<T>::vrow(Nk);
  unsigned char* ks = smem;
  unsigned char* vt = smem + (size_t)Nk * KROW;
  // ---- K rows and V^T into LDS
  const T* kb = k + ((long)b * Nk) * kv_ld + head * 64;   // (kv_ld: k and v may be the two halves of one fused projection's rows)
  const T* vb = v + ((long)b * Nk) * kv_ld + head * 64;
  constexpr int CPR = 64 / CH;   // 16-byte chunks per 64-channel row
  for (int i = t; i < Nk * CPR; i += 256) {
    const int key = i / CPR, c = i % CPR;
    *reinterpret_cast<uint4*>(ks + key * KROW + c * 16) = *reinterpret_cast<const uint4*>(kb + (long)key * kv_ld + c * CH);
    const uint4 vv = *reinterpret_cast<const uint4*>(vb + (long)key * kv_ld + c * CH);
    const T* ve = reinterpret_cast<const T*>(&vv);
#pragma unroll
    for (int e = 0; e < CH; ++e) *reinterpret_cast<T*>(vt + (c * CH + e) * VROW + key * (int)sizeof(T)) = ve[e];
  }
  __syncthreads();
  const float scale = 0.125f;   // 64 ** -0.5
  // bf16: the query fragments of block qb + 1 are loaded while block qb is computed (a block is ~1.5 us of arithmetic behind
  // a ~2 us HBM round trip, and only two workgroups fit a CU)
  auto q_ptr = [&](int qb) {
    const int q0 = (blockIdx.x * qblocks + qb) * 64 + wave * 16;
    const int qi = q0 + lr < N ? q0 + lr : N - 1;   // ragged tail: clamp, masked at the store
    return q + ((long)b * N + qi) * hidden + head * 64;
  };
  u32x4 qn0 = u32x4{0u, 0u, 0u, 0u}, qn1 = qn0;
  if constexpr (sizeof(T) == 2) {
    const T* qp0 = q_ptr(0);
    qn0 = *reinterpret_cast<const u32x4*>(qp0 + 8 * g);
    qn1 = *reinterpret_cast<const u32x4*>(qp0 + 32 + 8 * g);
  }
  for (int qb = 0; qb < qblocks; ++qb) {
    const int q0 = (blockIdx.x * qblocks + qb) * 64 + wave * 16;
    if (q0 >= N) break;   // wave-uniform
    const T* qp = q_ptr(qb);
    f32x4_t acc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) acc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
      // B = Q^T: lane holds Q[query lr][dims 8g .. 8g+7] (+32 for the second K step)
      const u32x4 qf0 = qn0, qf1 = qn1;
      if (qb + 1 < qblocks) {   // (rows past N clamp to the last query)
        const T* qpn = q_ptr(qb + 1);
        qn0 = *reinterpret_cast<const u32x4*>(qpn + 8 * g);
        qn1 = *reinterpret_cast<const u32x4*>(qpn + 32 + 8 * g);
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt * 16 < Nk) {
          const unsigned char* kr = ks + (kt * 16 + lr) * KROW + 16 * g;
          const u32x4 a0 = *reinterpret_cast<const u32x4*>(kr), a1 = *reinterpret_cast<const u32x4*>(kr + 64);
          acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, qf0), acc[kt], 0, 0, 0);
          acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, qf1), acc[kt], 0, 0, 0);
        }
      }
    } else {
      // fp32: K step s, lane group g <-> dim 16 g + s on both operands (any bijection sums the same 64 products)
      float qf[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 v4 = *reinterpret_cast<const float4*>(qp + 16 * g + 4 * c);
        qf[4 * c] = v4.x; qf[4 * c + 1] = v4.y; qf[4 * c + 2] = v4.z; qf[4 * c + 3] = v4.w;
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        if (kt * 16 < Nk) {
          const unsigned char* kr = ks + (kt * 16 + lr) * KROW + 64 * g;
          float kf[16];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float4 v4 = *reinterpret_cast<const float4*>(kr + 16 * c);
            kf[4 * c] = v4.x; kf[4 * c + 1] = v4.y; kf[4 * c + 2] = v4.z; kf[4 * c + 3] = v4.w;
          }
#pragma unroll
          for (int s = 0; s < 16; ++s) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], acc[kt], 0, 0, 0);
        }
      }
    }
    // ---- softmax over the keys of query lr: rows 4 g + reg of every tile, lane groups g = 0 .. 3
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt * 16 < Nk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[kt][r] *= scale; m = fmaxf(m, acc[kt][r]); }
      }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt * 16 < Nk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[kt][r] = __expf(acc[kt][r] - m); sum += acc[kt][r]; }
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    // ---- O^T = V^T P^T: four 16-channel tiles
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int s = 0; s < NKT / 2; ++s) {   // 32 keys per step: slots j < 4 <- tile 2s row 4g + j, j >= 4 <- tile 2s+1 row 4g + j - 4
        if (s * 32 < Nk) {
          const bool two = s * 32 + 16 < Nk;
          u32x4 pf;
          pf.x = (unsigned)f32_to_bf16(acc[2 * s][0]) | ((unsigned)f32_to_bf16(acc[2 * s][1]) << 16);
          pf.y = (unsigned)f32_to_bf16(acc[2 * s][2]) | ((unsigned)f32_to_bf16(acc[2 * s][3]) << 16);
          pf.z = two ? ((unsigned)f32_to_bf16(acc[2 * s + 1][0]) | ((unsigned)f32_to_bf16(acc[2 * s + 1][1]) << 16)) : 0u;
          pf.w = two ? ((unsigned)f32_to_bf16(acc[2 * s + 1][2]) | ((unsigned)f32_to_bf16(acc[2 * s + 1][3]) << 16)) : 0u;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const unsigned char* vr = vt + (dt * 16 + lr) * VROW + (s * 32 + 4 * g) * 2;
            const uint2 lo = *reinterpret_cast<const uint2*>(vr);
            const uint2 hi = two ? *reinterpret_cast<const uint2*>(vr + 32) : make_uint2(0u, 0u);
            const u32x4 vf = u32x4{lo.x, lo.y, hi.x, hi.y};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf), o[dt], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {   // step r of tile kt: slot g <-> key 16 kt + 4 g + r = register r of the scores
        if (kt * 16 < Nk) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const float4 v4 = *reinterpret_cast<const float4*>(vt + (dt * 16 + lr) * VROW + (kt * 16 + 4 * g) * 4);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.x, acc[kt][0], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.y, acc[kt][1], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.z, acc[kt][2], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.w, acc[kt][3], o[dt], 0, 0, 0);
          }
        }
      }
    }
    // ---- O^T tile dt: column = query lr, rows = channels dt * 16 + 4 g + reg -> 4 consecutive channels of the token row
    if (q0 + lr < N) {
      T* op = out + ((long)b * N + q0 + lr) * hidden + head * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if constexpr (sizeof(T) == 2) {
          uint2 pk;
          pk.x = (unsigned)f32_to_bf16(o[dt][0] * inv) | ((unsigned)f32_to_bf16(o[dt][1] * inv) << 16);
          pk.y = (unsigned)f32_to_bf16(o[dt][2] * inv) | ((unsigned)f32_to_bf16(o[dt][3] * inv) << 16);
          *reinterpret_cast<uint2*>(op + dt * 16) = pk;
        } else {
          *reinterpret_cast<float4*>(op + dt * 16) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
        }
      }
    }
  }
}

// bf16 form with TWO query tiles per wave and the keys in halves of 128 (online softmax across the halves): every K / V^T
// fragment read from LDS feeds two MFMAs.  The one-tile form above re-read all of K and V^T (69 KB) per 16 queries — 96 LDS
// reads per 64 MFMAs, eight waves of a CU queueing on one LDS: ~6k LDS cycles per round against 2k of matrix issue.  Scores stay
// in the log2 domain (one fused multiply-add + v_exp per score).
__global__ __launch_bounds__(256) void attention2_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                         bf16_t* __restrict__ out, int N, int Nk, int hidden, int qblocks, int kv_ld) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int KROW = AttCfg<bf16_t>::KROW, VROW = AttCfg<bf16_t>::vrow(Nk);
  unsigned char* ks = smem;
  unsigned char* vt = smem + (size_t)Nk * KROW;
  const bf16_t* kb = k + ((long)b * Nk) * kv_ld + head * 64;
  const bf16_t* vb = v + ((long)b * Nk) * kv_ld + head * 64;
  for (int i = t; i < Nk * 8; i += 256) {
    const int key = i >> 3, c = i & 7;
    *reinterpret_cast<uint4*>(ks + key * KROW + c * 16) = *reinterpret_cast<const uint4*>(kb + (long)key * kv_ld + c * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(vb + (long)key * kv_ld + c * 8);
    const bf16_t* ve = reinterpret_cast<const bf16_t*>(&vv);
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16_t*>(vt + (c * 8 + e) * VROW + key * 2) = ve[e];
  }
  __syncthreads();
  const float cs = 0.125f * 1.4426950408889634f;   // 64 ** -0.5 * log2(e)
  const int nhalf = (Nk + 127) >> 7;
  auto q_ptr = [&](int qb, int qt) {
    const int q0 = (blockIdx.x * qblocks + qb) * 128 + wave * 32 + qt * 16;
    const int qi = q0 + lr < N ? q0 + lr : N - 1;   // ragged tail: clamp, masked at the store
    return q + ((long)b * N + qi) * hidden + head * 64;
  };
  u32x4 qn[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const bf16_t* qp = q_ptr(0, qt);
    qn[qt][0] = *reinterpret_cast<const u32x4*>(qp + 8 * g);
    qn[qt][1] = *reinterpret_cast<const u32x4*>(qp + 32 + 8 * g);
  }
  for (int qb = 0; qb < qblocks; ++qb) {
    const int q0 = (blockIdx.x * qblocks + qb) * 128 + wave * 32;
    if (q0 >= N) break;   // wave-uniform
    u32x4 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) { qf[qt][0] = qn[qt][0]; qf[qt][1] = qn[qt][1]; }
    if (qb + 1 < qblocks) {
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const bf16_t* qp = q_ptr(qb + 1, qt);
        qn[qt][0] = *reinterpret_cast<const u32x4*>(qp + 8 * g);
        qn[qt][1] = *reinterpret_cast<const u32x4*>(qp + 32 + 8 * g);
      }
    }
    float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};
    f32x4_t o[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < nhalf; ++h) {
      const int key0 = h * 128;
      f32x4_t acc[8][2];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        acc[kt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        acc[kt][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (key0 + kt * 16 < Nk) {
          const unsigned char* kr = ks + (key0 + kt * 16 + lr) * KROW + 16 * g;
          const u32x4 a0 = *reinterpret_cast<const u32x4*>(kr), a1 = *reinterpret_cast<const u32x4*>(kr + 64);
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            acc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, qf[qt][0]), acc[kt][qt], 0, 0, 0);
            acc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, qf[qt][1]), acc[kt][qt], 0, 0, 0);
          }
        }
      }
      // ---- online softmax of this half, per query tile (rows 4 g + r of every key tile; lane groups g share a query)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
          if (key0 + kt * 16 < Nk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[kt][qt][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrun[qt], mx * cs);
        const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);   // first half: exp2(-inf) = 0 on zeros
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
          if (key0 + kt * 16 < Nk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[kt][qt][r] = __builtin_amdgcn_exp2f(fmaf(acc[kt][qt][r], cs, -mnew)); ps += acc[kt][qt][r]; }
          }
        lrun[qt] = fmaf(lrun[qt], alpha, ps);
        mrun[qt] = mnew;
        if (h > 0) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
        }
      }
      // ---- O^T += V^T P^T: 32 keys per step; slots j < 4 <- tile 2 s row 4 g + j, j >= 4 <- tile 2 s + 1 row 4 g + j - 4
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        if (key0 + s2 * 32 < Nk) {
          const bool two = key0 + s2 * 32 + 16 < Nk;
          u32x4 pf[2];
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            pf[qt].x = (unsigned)f32_to_bf16(acc[2 * s2][qt][0]) | ((unsigned)f32_to_bf16(acc[2 * s2][qt][1]) << 16);
            pf[qt].y = (unsigned)f32_to_bf16(acc[2 * s2][qt][2]) | ((unsigned)f32_to_bf16(acc[2 * s2][qt][3]) << 16);
            pf[qt].z = two ? ((unsigned)f32_to_bf16(acc[2 * s2 + 1][qt][0]) | ((unsigned)f32_to_bf16(acc[2 * s2 + 1][qt][1]) << 16)) : 0u;
            pf[qt].w = two ? ((unsigned)f32_to_bf16(acc[2 * s2 + 1][qt][2]) | ((unsigned)f32_to_bf16(acc[2 * s2 + 1][qt][3]) << 16)) : 0u;
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const unsigned char* vr = vt + (dt * 16 + lr) * VROW + (key0 + s2 * 32 + 4 * g) * 2;
            const uint2 lo = *reinterpret_cast<const uint2*>(vr);
            const uint2 hi = two ? *reinterpret_cast<const uint2*>(vr + 32) : make_uint2(0u, 0u);
            const u32x4 vf = u32x4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
              o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[qt]), o[qt][dt], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float l = lrun[qt];
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv = 1.f / l;
      const int qi = q0 + qt * 16 + lr;
      if (qi < N) {
        bf16_t* op = out + ((long)b * N + qi) * hidden + head * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 pk;
          pk.x = (unsigned)f32_to_bf16(o[qt][dt][0] * inv) | ((unsigned)f32_to_bf16(o[qt][dt][1] * inv) << 16);
          pk.y = (unsigned)f32_to_bf16(o[qt][dt][2] * inv) | ((unsigned)f32_to_bf16(o[qt][dt][3] * inv) << 16);
          *reinterpret_cast<uint2*>(op + dt * 16) = pk;
        }
      }
    }
  }
}

static int attention2_launch(const void* q, const void* k, const void* v, void* out, int B, int N, int Nk, int hidden, int kv_ld, hipStream_t s) {
  const int smem = Nk * AttCfg<bf16_t>::KROW + 64 * AttCfg<bf16_t>::vrow(Nk);
  static bool attr_set = false;
  if (!attr_set) {
    const int most = 256 * AttCfg<bf16_t>::KROW + 64 * AttCfg<bf16_t>::vrow(256);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int heads = hidden / 64;
  int qblocks = 1;
  while (qblocks < 4 && (long)((N + 128 * qblocks * 2 - 1) / (128 * qblocks * 2)) * heads * B >= 512) qblocks *= 2;
  dim3 grid((N + 128 * qblocks - 1) / (128 * qblocks), heads, B);
  ProfScope ps("sf_attention_bf16", 4.0 * B * (double)N * Nk * hidden, ((double)B * N * hidden * 2 + (double)B * Nk * hidden * 2) * 2, s);
  hipLaunchKernelGGL(attention2_kernel, grid, dim3(256), smem, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, N, Nk, hidden,
                     qblocks, kv_ld);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  return (int)(b < 1 ? 1 : b);
}

template <typename T>
int layernorm_t(const void* x, const float* gamma, const float* beta, void* y, long rows, int C, float eps, hipStream_t s) {
  constexpr int CH = Elem<T>::CH;
  const int nch = C / CH;
  int G = 1;
  while (G < nch && G < 64) G <<= 1;
  if (nch > 2 * G) return -2;
  const long blocks = (rows * G + 255) / 256;
#define FLAIR_LN_CASE(g) \
  case g: hipLaunchKernelGGL((layernorm_kernel<T, g>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)x, gamma, beta, (T*)y, rows, C, eps); break;
  switch (G) {
    FLAIR_LN_CASE(1) FLAIR_LN_CASE(2) FLAIR_LN_CASE(4) FLAIR_LN_CASE(8) FLAIR_LN_CASE(16) FLAIR_LN_CASE(32) FLAIR_LN_CASE(64)
  }
#undef FLAIR_LN_CASE
  FLAIR_CHECK_LAUNCH();
  return 0;
}

template <typename T, int NKT>
int attention_launch(const void* q, const void* k, const void* v, void* out, int B, int N, int Nk, int hidden, int kv_ld, hipStream_t s) {
  auto kern = attention_kernel<T, NKT>;
  const int smem = Nk * AttCfg<T>::KROW + 64 * AttCfg<T>::vrow(Nk);
  static bool attr_set = false;
  if (!attr_set) {
    const int most = NKT * 16 * AttCfg<T>::KROW + 64 * AttCfg<T>::vrow(NKT * 16);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, most);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int heads = hidden / 64;
  // enough workgroups to fill the chip, each amortising its K / V load over several query blocks
  int qblocks = 1;
  while (qblocks < 8 && (long)((N + 64 * qblocks * 2 - 1) / (64 * qblocks * 2)) * heads * B >= 512) qblocks *= 2;
  dim3 grid((N + 64 * qblocks - 1) / (64 * qblocks), heads, B);
  ProfScope ps(sizeof(T) == 2 ? "sf_attention_bf16" : "sf_attention_f32", 4.0 * B * (double)N * Nk * hidden,
               ((double)B * N * hidden * 2 + (double)B * Nk * hidden * 2) * sizeof(T), s);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const T*)q, (const T*)k, (const T*)v, (T*)out, N, Nk, hidden, qblocks, kv_ld);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

}  // namespace

int sf_layernorm(int dtype, const void* x, const float* gamma, const float* beta, void* y, long rows, int C, float eps, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch) return -2;
  ProfScope ps("sf_layernorm", 0.0, 2.0 * rows * C * dtype_size(dtype), s);
  return dtype == DT_F32 ? layernorm_t<float>(x, gamma, beta, y, rows, C, eps, s) : layernorm_t<bf16_t>(x, gamma, beta, y, rows, C, eps, s);
}

template <int L>
static void dwconv_launch(int dtype, dim3 grid, const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int cg,
                          hipStream_t s) {
  if (dtype == DT_F32) hipLaunchKernelGGL((dwconv3x3_gelu_kernel<float, L>), grid, dim3(256), 0, s, (const float*)x, w, bias, (float*)y, B, H, W, C, cg);
  else hipLaunchKernelGGL((dwconv3x3_gelu_kernel<bf16_t, L>), grid, dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, B, H, W, C, cg);
}

int sf_dwconv3x3_gelu(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, hipStream_t s) {
  if (C % 4) return -2;
  const int ng = C / 4;                                   // 4-channel groups, one per thread
  int cg = 1;
  while (cg * 2 <= 256 && ng % (cg * 2) == 0) cg *= 2;    // groups per workgroup: the largest power of two dividing both 256 and
                                                          // the group count (4 x 320 channels = 320 groups -> 64 per workgroup, 5 in y)
  int L = W % 16 == 0 ? 16 : W % 8 == 0 ? 8 : W % 4 == 0 ? 4 : W % 2 == 0 ? 2 : 1;   // pixels per segment
  const int lmax = tune("FLAIR_SF_DW_L", dtype == DT_F32 ? 16 : 4);   // (bf16: 16 and 8 spill at four waves per SIMD)
  while (L > lmax && L > 1) L >>= 1;
  const long items = (long)B * H * (W / L);
  if (items > (1L << 30)) return -2;
  const int npl = 256 / cg;
  long gx = (items + npl - 1) / npl;
  if (gx > 8192) gx = 8192;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, ng / cg);
  ProfScope ps("sf_dwconv_gelu", 18.0 * B * H * W * C, 2.0 * B * H * W * C * dtype_size(dtype), s);
  switch (L) {
    case 16: dwconv_launch<16>(dtype, grid, x, w, bias, y, B, H, W, C, cg, s); break;
    case 8: dwconv_launch<8>(dtype, grid, x, w, bias, y, B, H, W, C, cg, s); break;
    case 4: dwconv_launch<4>(dtype, grid, x, w, bias, y, B, H, W, C, cg, s); break;
    case 2: dwconv_launch<2>(dtype, grid, x, w, bias, y, B, H, W, C, cg, s); break;
    default: dwconv_launch<1>(dtype, grid, x, w, bias, y, B, H, W, C, cg, s); break;
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_bilinear_nhwc(int dtype, const void* x, void* y, int B, int h, int w, int C, int H, int W, int ld, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if ((C % ch) || (ld % ch)) return -2;
  const long total = (long)B * H * W * (C / ch);
  ProfScope ps("sf_bilinear", 0.0, ((double)B * H * W * C + (double)B * h * w * C) * dtype_size(dtype), s);
  if (dtype == DT_F32) hipLaunchKernelGGL(bilinear_nhwc_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)x, (float*)y, B, h, w, C, H, W, ld);
  else hipLaunchKernelGGL(bilinear_nhwc_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, h, w, C, H, W, ld);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_bilinear_nchw_f32(const float* x, float* y, long planes, int h, int w, int H, int W, hipStream_t s) {
  const long total = planes * H * W;
  ProfScope ps("sf_bilinear_logits", 0.0, 4.0 * (planes * H * W + planes * h * w), s);
  constexpr int RPB = 8;
  if (W % 4 == 0 && H % RPB == 0 && planes * H / RPB < (1L << 31)) {
    hipLaunchKernelGGL(bilinear_nchw_f32_kernel<RPB>, dim3((unsigned)(planes * H / RPB)), dim3(256), 0, s, x, y, h, w, H, W);
  } else {
    hipLaunchKernelGGL(bilinear_nchw_f32_any_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, planes, h, w, H, W);
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_slice_cols(int dtype, const float* src, int ld, int col0, int ncols, long rows, void* dst, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (ncols % ch) return -2;
  const long total = rows * (ncols / ch);
  if (dtype == DT_F32) hipLaunchKernelGGL(slice_cols_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, src, ld, col0, ncols, rows, (float*)dst);
  else hipLaunchKernelGGL(slice_cols_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, src, ld, col0, ncols, rows, (bf16_t*)dst);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_fuse_bias(const float* wf, int D, const float* b3, const float* b2, const float* b1, const float* b0, const float* scale,
                 const float* shift, float* shift2, hipStream_t s) {
  hipLaunchKernelGGL(fuse_bias_kernel, dim3((D + 3) / 4), dim3(256), 0, s, wf, D, b3, b2, b1, b0, scale, shift, shift2);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_upsample_sum_bn_relu(int dtype, const void* g0, const void* g1, const void* g2, const void* g3, const float* scale, const float* shift2,
                            void* z, int B, int H, int W, int D, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if ((D % ch) || (H % 8) || (W % 8)) return -2;
  const long total = (long)B * H * W * (D / ch);
  ProfScope ps("sf_upsample_sum", 0.0, 2.3 * B * H * W * D * dtype_size(dtype), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(upsample_sum_bn_relu_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)g0, (const float*)g1, (const float*)g2,
                       (const float*)g3, scale, shift2, (float*)z, B, H, W, D);
  else
    hipLaunchKernelGGL(upsample_sum_bn_relu_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)g0, (const bf16_t*)g1,
                       (const bf16_t*)g2, (const bf16_t*)g3, scale, shift2, (bf16_t*)z, B, H, W, D);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_ffn_dw_pack(const float* w, const float* b, float* out, int nch, hipStream_t s) {
  if (nch % 4) return -2;
  hipLaunchKernelGGL(ffn_dw_pack_kernel, dim3((nch * 10 + 255) / 256), dim3(256), 0, s, w, b, out, nch);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

bool sf_ffn_fused_ok(int dtype, int C, int H, int W) {
  return dtype == DT_BF16 && (C == 64 || C == 128) && H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8;
}

template <int C>
static int ffn_fused_launch(const void* x, const float* ln_g, const float* ln_b, const void* w1, const float* b1, const float* dwp, const void* w2,
                            const float* b2, void* out, int B, int H, int W, float eps, const float* ln2_g, const float* ln2_b, void* out_ln,
                            hipStream_t s) {
  auto kern = ffn_fused_kernel<C>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FfnCfg<C>::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long tiles = (long)B * (H / 8) * (W / 8);
  if (tiles > (1L << 30)) return -2;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(FFN_NT), FfnCfg<C>::SMEM, s, (const bf16_t*)x, ln_g, ln_b, (const bf16_t*)w1, b1, dwp,
                     (const bf16_t*)w2, b2, (bf16_t*)out, B, H, W, eps, ln2_g, ln2_b, (bf16_t*)out_ln);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_ffn_fused(const void* x, const float* ln_g, const float* ln_b, const void* w1, const float* b1, const float* dwp, const void* w2,
                 const float* b2, void* out, int B, int H, int W, int C, float eps, const float* ln2_g, const float* ln2_b, void* out_ln,
                 hipStream_t s) {
  if (!sf_ffn_fused_ok(DT_BF16, C, H, W) || x == out || out_ln == x || (out_ln && (!ln2_g || !ln2_b))) return -2;
  ProfScope ps("sf_ffn_fused", 2.0 * B * H * W * C * 4.0 * C * 2.0 + 18.0 * B * H * W * 4.0 * C, 2.0 * B * H * W * C * 2.0, s);
  return C == 64 ? ffn_fused_launch<64>(x, ln_g, ln_b, w1, b1, dwp, w2, b2, out, B, H, W, eps, ln2_g, ln2_b, out_ln, s)
                 : ffn_fused_launch<128>(x, ln_g, ln_b, w1, b1, dwp, w2, b2, out, B, H, W, eps, ln2_g, ln2_b, out_ln, s);
}

int sf_head_wint(void* wint, hipStream_t s) {
  hipLaunchKernelGGL(head_wint_kernel, dim3((HT_PX * HT_SRC + 255) / 256), dim3(256), 0, s, (bf16_t*)wint);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

bool sf_head_fused_ok(int dtype, int H, int W, int C0, int D, int labels) {
  return dtype == DT_BF16 && C0 == 64 && D % 64 == 0 && labels <= 32 && H % HT_H == 0 && W % HT_W == 0 && H >= 8 && W >= 16;
}

int sf_head_fused(const void* f0, const void* w0, const void* g1, const void* g2, const void* g3, const void* wint, const float* scale,
                  const float* shift2, const void* wc, const float* bc, float* out, int B, int H, int W, int D, int labels, hipStream_t s) {
  if (!sf_head_fused_ok(DT_BF16, H, W, 64, D, labels)) return -2;
  const long tiles = (long)B * (H / HT_H) * (W / HT_W);
  if (tiles > (1L << 30)) return -2;
  ProfScope ps("sf_head_fused", 2.0 * B * H * W * D * (64.0 + HT_SRC + 32.0),
               2.0 * B * H * W * 64 + 2.0 * B * H * W * D * (1.0 / 4 + 1.0 / 16 + 1.0 / 64) + 4.0 * B * H * W * labels, s);
  hipLaunchKernelGGL(head_fused_kernel, dim3((unsigned)tiles), dim3(256), 0, s, (const bf16_t*)f0, (const bf16_t*)w0, (const bf16_t*)g1,
                     (const bf16_t*)g2, (const bf16_t*)g3, (const bf16_t*)wint, scale, shift2, (const bf16_t*)wc, bc, out, B, H, W, D, labels);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sf_attention(int dtype, const void* q, const void* k, const void* v, void* out, int B, int N, int Nk, int hidden, int kv_ld, hipStream_t s) {
  if (hidden % 64 || Nk < 1 || Nk > 256 || (Nk % 16)) return -2;   // heads of 64 channels; keys in whole 16-key tiles, all in LDS
  if (kv_ld < hidden || (kv_ld % (dtype == DT_F32 ? 4 : 8))) return -2;
  if (dtype == DT_F32) return attention_launch<float, 16>(q, k, v, out, B, N, Nk, hidden, kv_ld, s);
  if (tune("FLAIR_SF_ATT2", 1)) return attention2_launch(q, k, v, out, B, N, Nk, hidden, kv_ld, s);
  return attention_launch<bf16_t, 16>(q, k, v, out, B, N, Nk, hidden, kv_ld, s);
}

}  // namespace flair
