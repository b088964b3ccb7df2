// Kernels of the SegFormer (MiT) inference path that the U-Net path does not have: LayerNorm over channels, efficient
// self-attention with a spatially reduced key / value sequence, depth-wise 3x3 + GELU of the Mix-FFN, bilinear resampling.
// (BASELINE config 5: zone_detect with a HuggingFace model, /root/reference/src/zone_detect/model.py:42-50, compare.py:31-36.
// The model itself is third-party — transformers' SegformerForSemanticSegmentation — restated from its published algorithm;
// the matrix products run on the gather-form implicit-GEMM kernel of conv_igemm.hip, 1x1 and strided convolutions alike.)
// Activations are token-major [B * H * W][C] = NHWC, T = float (parity mode, exact-fp32 MFMA) or bf16.
#include "segformer_ops.h"

#include "prof.h"

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
// its 9 x CH weights and CH biases live in registers — and walks pixels; neighbouring threads hold neighbouring chunks of the
// same pixel, so every load and store of a wave is one contiguous run.  (First version: one (pixel, chunk) item per thread
// with its 72 weights re-read at a 36-byte stride per item: 0.24 TB/s, half of the SegFormer forward.)
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_gelu_kernel(const T* __restrict__ x, const float* __restrict__ w /* [C][3][3] */,
                                                             const float* __restrict__ bias, T* __restrict__ y, int B, int H, int W, int C,
                                                             int cg /* chunks per workgroup: min(C / CH, 256) */) {
  constexpr int CH = Elem<T>::CH;
  const int c = blockIdx.y * cg + threadIdx.x % cg;     // this thread's chunk
  const int pl = threadIdx.x / cg, npl = 256 / cg;      // pixel lane of the workgroup
  float wr[9][CH], bs[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    bs[e] = bias[c * CH + e];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wr[tp][e] = w[(c * CH + e) * 9 + tp];
  }
  const long P = (long)B * H * W;
  for (long p = (long)blockIdx.x * npl + pl; p < P; p += (long)gridDim.x * npl) {
    const int px = (int)(p % W), py = (int)((p / W) % H);
    const T* base = x + p * C + (long)c * CH;
    float acc[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = bs[e];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const bool yok = (unsigned)(py + r - 1) < (unsigned)H;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const bool ok = yok && (unsigned)(px + q - 1) < (unsigned)W;
        float f[CH];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(ok ? base + ((long)(r - 1) * W + (q - 1)) * C : base), f);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] = fmaf(ok ? f[e] : 0.f, wr[r * 3 + q][e], acc[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = 0.5f * acc[e] * (1.f + erff(acc[e] * 0.70710678118654752f));
    *reinterpret_cast<uint4*>(y + p * C + (long)c * CH) = f_to_chunk<T>(acc);
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

// fp32 NCHW [B][C][h][w] -> fp32 NCHW [B][C][H][W] (the logits, x4)
__global__ __launch_bounds__(256) void bilinear_nchw_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int h, int w,
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
                                                        T* __restrict__ out, int N, int Nk, int hidden, int qblocks) {
  constexpr int CH = Elem<T>::CH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int KROW = AttCfg<T>::KROW, VROW = AttCfg<T>::vrow(Nk);
  unsigned char* ks = smem;
  unsigned char* vt = smem + (size_t)Nk * KROW;
  // ---- K rows and V^T into LDS
  const T* kb = k + ((long)b * Nk) * hidden + head * 64;
  const T* vb = v + ((long)b * Nk) * hidden + head * 64;
  constexpr int CPR = 64 / CH;   // 16-byte chunks per 64-channel row
  for (int i = t; i < Nk * CPR; i += 256) {
    const int key = i / CPR, c = i % CPR;
    *reinterpret_cast<uint4*>(ks + key * KROW + c * 16) = *reinterpret_cast<const uint4*>(kb + (long)key * hidden + c * CH);
    const uint4 vv = *reinterpret_cast<const uint4*>(vb + (long)key * hidden + c * CH);
    const T* ve = reinterpret_cast<const T*>(&vv);
#pragma unroll
    for (int e = 0; e < CH; ++e) *reinterpret_cast<T*>(vt + (c * CH + e) * VROW + key * (int)sizeof(T)) = ve[e];
  }
  __syncthreads();
  const float scale = 0.125f;   // 64 ** -0.5
  for (int qb = 0; qb < qblocks; ++qb) {
    const int q0 = (blockIdx.x * qblocks + qb) * 64 + wave * 16;
    if (q0 >= N) break;   // wave-uniform
    const int qi = q0 + lr < N ? q0 + lr : N - 1;   // ragged tail: clamp, masked at the store
    const T* qp = q + ((long)b * N + qi) * hidden + head * 64;
    f32x4_t acc[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) acc[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
      // B = Q^T: lane holds Q[query lr][dims 8g .. 8g+7] (+32 for the second K step)
      const u32x4 qf0 = *reinterpret_cast<const u32x4*>(qp + 8 * g), qf1 = *reinterpret_cast<const u32x4*>(qp + 32 + 8 * g);
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
int attention_launch(const void* q, const void* k, const void* v, void* out, int B, int N, int Nk, int hidden, hipStream_t s) {
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
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const T*)q, (const T*)k, (const T*)v, (T*)out, N, Nk, hidden, qblocks);
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

int sf_dwconv3x3_gelu(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch) return -2;
  const int nch = C / ch;
  int cg = 1;
  while (cg * 2 <= 256 && nch % (cg * 2) == 0) cg *= 2;   // chunks per workgroup: the largest power of two dividing both 256 and the
                                                          // chunk count (4 x 320 channels = 160 bf16 chunks -> 32 per workgroup, 5 groups)
  const long P = (long)B * H * W;
  const int npl = 256 / cg;
  long gx = (P + (long)npl * 16 - 1) / ((long)npl * 16);   // ~16 pixels per thread amortise its 9 x CH weight loads
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, nch / cg);
  ProfScope ps("sf_dwconv_gelu", 18.0 * B * H * W * C, 2.0 * B * H * W * C * dtype_size(dtype), s);
  if (dtype == DT_F32) hipLaunchKernelGGL(dwconv3x3_gelu_kernel<float>, grid, dim3(256), 0, s, (const float*)x, w, bias, (float*)y, B, H, W, C, cg);
  else hipLaunchKernelGGL(dwconv3x3_gelu_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, B, H, W, C, cg);
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
  hipLaunchKernelGGL(bilinear_nchw_f32_kernel, dim3(ew_blocks(total)), dim3(256), 0, s, x, y, planes, h, w, H, W);
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

int sf_attention(int dtype, const void* q, const void* k, const void* v, void* out, int B, int N, int Nk, int hidden, hipStream_t s) {
  if (hidden % 64 || Nk < 1 || Nk > 256 || (Nk % 16)) return -2;   // heads of 64 channels; keys in whole 16-key tiles, all in LDS
  if (dtype == DT_F32) return attention_launch<float, 16>(q, k, v, out, B, N, Nk, hidden, s);
  return attention_launch<bf16_t, 16>(q, k, v, out, B, N, Nk, hidden, s);
}

}  // namespace flair
