// Fused per-pixel head of the reference's step(): one-hot -> label argmax, (weighted) softmax
// cross-entropy mean, its gradient, argmax(softmax) predictions and the confusion-matrix bincount
// in ONE pass over the fp32 NCHW logits.  Replaces the chain at
//   /root/reference/src/flair/task_module.py:71-79  (argmax, criterion, softmax, argmax, flatten)
//   /root/reference/src/flair/tasks_utils.py:88-93  (nn.CrossEntropyLoss(weight), mean reduction)
//   torchmetrics MulticlassJaccardIndex.update      (bincount(target*C + pred)), task_module.py:85,107-108
// HBM-bound: reads C floats per pixel once (coalesced across pixels), integer histogram in LDS,
// one 64-bit integer atomic per (block, cell); loss reduced through deterministic partials.
#include "ops.h"
#include "prof.h"

namespace flair {

constexpr int MAXC = 32;
constexpr int CE_MAX_BLOCKS = 2048;

static inline int ce_blocks(long npix) {
  long b = (npix + 255) / 256;
  if (b > CE_MAX_BLOCKS) b = CE_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

size_t ce_workspace_floats(int B, int H, int W) {
  const long npix = (long)B * H * W;
  return (size_t)((npix + 3) / 4 + 2 * CE_MAX_BLOCKS + 16);
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
  }
  __syncthreads();
  return r;  // valid in thread 0
}

// labels -> uint8 (255 = ignored), partial sums of w[y]
__global__ __launch_bounds__(256) void ce_labels_kernel(const void* __restrict__ labels, int kind, const float* __restrict__ weight,
                                 int C, long HW, long npix, unsigned char* __restrict__ lab8,
                                 int* __restrict__ targets_i32, float* __restrict__ den_partial) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    int y;
    if (kind == 0) y = ((const unsigned char*)labels)[i];
    else if (kind == 1) y = ((const int*)labels)[i];
    else if (kind == 2) y = (int)((const long long*)labels)[i];
    else {
      const long n = i / HW, pix = i - n * HW;
      const float* p = (const float*)labels + n * C * HW + pix;
      float best = p[0];
      y = 0;
      for (int c = 1; c < C; ++c) {
        const float v = p[(long)c * HW];
        if (v > best) { best = v; y = c; }
      }
    }
    const bool ok = (unsigned)y < (unsigned)C;
    lab8[i] = ok ? (unsigned char)y : (unsigned char)255;
    if (targets_i32) targets_i32[i] = y;
    if (ok) acc += weight ? weight[y] : 1.f;
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) den_partial[blockIdx.x] = r;
}

__global__ void ce_sum_kernel(const float* __restrict__ partial, int n, float* __restrict__ out, const float* __restrict__ den) {
  __shared__ double sh[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)partial[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = den ? (float)(sh[0] / (double)den[0]) : (float)sh[0];
}

template <typename TO>
__device__ __forceinline__ void store_row(TO* dst, const float* g, int C, int ld);
template <>
__device__ __forceinline__ void store_row<float>(float* dst, const float* g, int C, int ld) {
#pragma unroll
  for (int c0 = 0; c0 < MAXC; c0 += 4)
    if (c0 < ld) {
      float f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = (c0 + e) < C ? g[c0 + e] : 0.f;
      *reinterpret_cast<uint4*>(dst + c0) = f_to_chunk<float>(f);
    }
}
template <>
__device__ __forceinline__ void store_row<bf16_t>(bf16_t* dst, const float* g, int C, int ld) {
#pragma unroll
  for (int c0 = 0; c0 < MAXC; c0 += 8)
    if (c0 < ld) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (c0 + e) < C ? g[c0 + e] : 0.f;
      *reinterpret_cast<uint4*>(dst + c0) = f_to_chunk<bf16_t>(f);
    }
}

__global__ __launch_bounds__(256) void ce_main_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ lab8,
                                                      const float* __restrict__ weight, const float* __restrict__ den,
                                                      int C, long HW, long npix, float* __restrict__ loss_partial,
                                                      float* __restrict__ dl_nchw, void* __restrict__ dl_nhwc, int dl_dtype,
                                                      int dl_ld, unsigned char* __restrict__ preds_u8,
                                                      long long* __restrict__ preds_i64, long long* __restrict__ confmat) {
  __shared__ float sh[4];
  __shared__ unsigned int hist[MAXC * MAXC];
  if (confmat) {
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
  }
  const float inv_den = 1.f / den[0];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long n = i / HW, pix = i - n * HW;
    const float* p = logits + n * C * HW + pix;
    float x[MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { x[c] = p[(long)c * HW]; m = fmaxf(m, x[c]); }
    float ssum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { x[c] = __expf(x[c] - m); ssum += x[c]; }
    // probabilities, argmax(softmax) with first-index tie break (task_module.py:75-76)
    const float rs = 1.f / ssum;   // one division per pixel (was one per class; with __expf: 665 -> ~300 vector instructions per pixel)
    int pred = 0;
    float pbest = -1.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        x[c] = x[c] * rs;
        if (x[c] > pbest) { pbest = x[c]; pred = c; }
      }
    const int y = lab8[i];
    const bool ok = y < C;
    const float w = ok ? (weight ? weight[y] : 1.f) : 0.f;
    if (ok) {
      const float xy = p[(long)y * HW];
      acc += w * (logf(ssum) - (xy - m));
    }
    if (preds_u8) preds_u8[i] = (unsigned char)pred;
    if (preds_i64) preds_i64[i] = pred;
    if (confmat && ok) atomicAdd(&hist[y * C + pred], 1u);
    if (dl_nchw || dl_nhwc) {
      const float sc = w * inv_den;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) x[c] = (x[c] - ((ok && c == y) ? 1.f : 0.f)) * sc;
      if (dl_nchw) {
        float* d = dl_nchw + n * C * HW + pix;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) d[(long)c * HW] = x[c];
      }
      if (dl_nhwc) {
        if (dl_dtype == DT_F32) store_row<float>((float*)dl_nhwc + i * dl_ld, x, C, dl_ld);
        else store_row<bf16_t>((bf16_t*)dl_nhwc + i * dl_ld, x, C, dl_ld);
      }
    }
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) loss_partial[blockIdx.x] = r;
  if (confmat) {
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) {
      const unsigned int v = hist[i];
      if (v) atomicAdd(reinterpret_cast<unsigned long long*>(confmat + i), (unsigned long long)v);
    }
  }
}

// Same arithmetic, four consecutive pixels per thread: every class plane is read with 16-byte loads (a wave covers
// 1 KB per plane instead of 256 B) and predictions leave as one 32-bit word.  Needs HW % 4 == 0.
template <int CM>
__global__ __launch_bounds__(256) void ce_main4_kernel(const float* __restrict__ logits, const unsigned char* __restrict__ lab8,
                                                       const float* __restrict__ weight, const float* __restrict__ den,
                                                       int C, long HW, long npix, float* __restrict__ loss_partial,
                                                       float* __restrict__ dl_nchw, void* __restrict__ dl_nhwc, int dl_dtype,
                                                       int dl_ld, unsigned char* __restrict__ preds_u8,
                                                       long long* __restrict__ preds_i64, long long* __restrict__ confmat) {
  __shared__ float sh[4];
  __shared__ unsigned int hist[MAXC * MAXC];
  if (confmat) {
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
  }
  const float inv_den = 1.f / den[0];
  float acc = 0.f;
  const long ngroups = npix >> 2;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (long)gridDim.x * blockDim.x) {
    const long i = gi << 2;
    const long n = i / HW, pix = i - n * HW;
    const float* p = logits + n * C * HW + pix;
    float4 xv[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) xv[c] = *reinterpret_cast<const float4*>(p + (long)c * HW);
    const unsigned int lab4 = *reinterpret_cast<const unsigned int*>(lab8 + i);
    unsigned int pred4 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x[CM];
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) { x[c] = reinterpret_cast<const float*>(&xv[c])[q]; m = fmaxf(m, x[c]); }
      const int y = (lab4 >> (8 * q)) & 0xff;
      const bool ok = y < C;
      float xy = 0.f;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C && c == y) xy = x[c];
      float ssum = 0.f;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) { x[c] = __expf(x[c] - m); ssum += x[c]; }
      const float rs = 1.f / ssum;
      int pred = 0;
      float pbest = -1.f;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) {
          x[c] = x[c] * rs;
          if (x[c] > pbest) { pbest = x[c]; pred = c; }
        }
      const float w = ok ? (weight ? weight[y] : 1.f) : 0.f;
      if (ok) acc += w * (logf(ssum) - (xy - m));
      pred4 |= (unsigned int)pred << (8 * q);
      if (confmat && ok) atomicAdd(&hist[y * C + pred], 1u);
      if (dl_nchw || dl_nhwc) {
        const float sc = w * inv_den;
#pragma unroll
        for (int c = 0; c < CM; ++c)
          if (c < C) x[c] = (x[c] - ((ok && c == y) ? 1.f : 0.f)) * sc;
        if (dl_nhwc) {
          float g[MAXC];
#pragma unroll
          for (int c = 0; c < MAXC; ++c) g[c] = c < CM ? x[c < CM ? c : 0] : 0.f;
          if (dl_dtype == DT_F32) store_row<float>((float*)dl_nhwc + (i + q) * dl_ld, g, C, dl_ld);
          else store_row<bf16_t>((bf16_t*)dl_nhwc + (i + q) * dl_ld, g, C, dl_ld);
        }
        if (dl_nchw) {
#pragma unroll
          for (int c = 0; c < CM; ++c)
            if (c < C) reinterpret_cast<float*>(&xv[c])[q] = x[c];
        }
      }
    }
    if (preds_u8) *reinterpret_cast<unsigned int*>(preds_u8 + i) = pred4;
    if (preds_i64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) preds_i64[i + q] = (pred4 >> (8 * q)) & 0xff;
    }
    if (dl_nchw) {
      float* d = dl_nchw + n * C * HW + pix;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) *reinterpret_cast<float4*>(d + (long)c * HW) = xv[c];
    }
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) loss_partial[blockIdx.x] = r;
  if (confmat) {
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) {
      const unsigned int v = hist[i];
      if (v) atomicAdd(reinterpret_cast<unsigned long long*>(confmat + i), (unsigned long long)v);
    }
  }
}

// The same head over logits kept in the network's own layout (NHWC T, row stride ld >= C, what the head convolution
// writes when nobody asks for fp32 NCHW logits): one pixel per thread, the whole row in ld*sizeof(T) contiguous bytes.
// In bf16 mode the NCHW fp32 logits are the same bf16 values widened, so both layouts give bit-identical results.
template <typename T, int LD>
__global__ __launch_bounds__(256) void ce_main_nhwc_kernel(const T* __restrict__ logits, const unsigned char* __restrict__ lab8,
                                                           const float* __restrict__ weight, const float* __restrict__ den,
                                                           int C, long npix, float* __restrict__ loss_partial,
                                                           void* __restrict__ dl_nhwc, int dl_ld,
                                                           unsigned char* __restrict__ preds_u8, long long* __restrict__ confmat) {
  constexpr int CH = Elem<T>::CH;
  __shared__ float sh[4];
  __shared__ unsigned int hist[MAXC * MAXC];
  if (confmat) {
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
  }
  const float inv_den = 1.f / den[0];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float x[MAXC];
#pragma unroll
    for (int c0 = 0; c0 < LD; c0 += CH) chunk_to_f<T>(*reinterpret_cast<const uint4*>(logits + i * LD + c0), x + c0);
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < LD; ++c)
      if (c < C) m = fmaxf(m, x[c]);
    const int y = lab8[i];
    const bool ok = y < C;
    float xy = 0.f, ssum = 0.f;
#pragma unroll
    for (int c = 0; c < LD; ++c)
      if (c < C) {
        if (c == y) xy = x[c];
        x[c] = __expf(x[c] - m);
        ssum += x[c];
      }
    const float rs = 1.f / ssum;
    int pred = 0;
    float pbest = -1.f;
#pragma unroll
    for (int c = 0; c < LD; ++c)
      if (c < C) {
        x[c] = x[c] * rs;
        if (x[c] > pbest) { pbest = x[c]; pred = c; }
      }
    const float w = ok ? (weight ? weight[y] : 1.f) : 0.f;
    if (ok) acc += w * (logf(ssum) - (xy - m));
    if (preds_u8) preds_u8[i] = (unsigned char)pred;
    if (confmat && ok) atomicAdd(&hist[y * C + pred], 1u);
    if (dl_nhwc) {
      const float sc = w * inv_den;
      float gq[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) gq[c] = (c < LD && c < C) ? (x[c < LD ? c : 0] - ((ok && c == y) ? 1.f : 0.f)) * sc : 0.f;
      store_row<T>((T*)dl_nhwc + i * dl_ld, gq, C, dl_ld);
    }
  }
  const float r = block_sum(acc, sh);
  if (threadIdx.x == 0) loss_partial[blockIdx.x] = r;
  if (confmat) {
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) {
      const unsigned int v = hist[i];
      if (v) atomicAdd(reinterpret_cast<unsigned long long*>(confmat + i), (unsigned long long)v);
    }
  }
}

int ce_head(const CeArgs& a, hipStream_t s) {
  if (a.C > MAXC || a.C < 1) return -2;
  if (a.dlogits_nhwc && (a.dlogits_ld < a.C || a.dlogits_ld > MAXC || a.dlogits_ld % (a.dlogits_dtype == DT_F32 ? 4 : 8))) return -3;
  const long HW = (long)a.H * a.W, npix = HW * a.B;
  const int nb = ce_blocks(npix);
  unsigned char* lab8 = reinterpret_cast<unsigned char*>(a.workspace);
  float* den_partial = a.workspace + (npix + 3) / 4;
  float* loss_partial = den_partial + CE_MAX_BLOCKS;
  float* den = loss_partial + CE_MAX_BLOCKS;
  ProfScope* p1 = new ProfScope("ce_labels", 0.0, (double)npix * (a.label_kind == 3 ? 4.0 * a.C : 1.0), s);
  hipLaunchKernelGGL(ce_labels_kernel, dim3(nb), dim3(256), 0, s, a.labels, a.label_kind, a.weight, a.C, HW, npix, lab8,
                     a.targets_i32, den_partial);
  FLAIR_CHECK_LAUNCH();
  delete p1;
  hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(256), 0, s, den_partial, nb, den, (const float*)nullptr);
  FLAIR_CHECK_LAUNCH();
  if (a.logits_nhwc) {
    const int ld = a.logits_ld;
    if ((ld != 16 && ld != 24 && ld != 32) || ld < a.C || a.dlogits_nchw || a.preds_i64 || (a.dlogits_nhwc && (a.dlogits_ld != ld || a.dlogits_dtype != a.logits_dtype)))
      return -3;
    ProfScope ps("ce_main", 0.0, (double)npix * (2.0 * ld * dtype_size(a.logits_dtype) + 2), s);
    if (a.logits_dtype == DT_F32) {
      auto kern = ld == 16 ? ce_main_nhwc_kernel<float, 16> : ld == 24 ? ce_main_nhwc_kernel<float, 24> : ce_main_nhwc_kernel<float, 32>;
      hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, s, (const float*)a.logits_nhwc, lab8, a.weight, den, a.C, npix, loss_partial,
                         a.dlogits_nhwc, a.dlogits_ld, a.preds_u8, a.confmat);
    } else {
      auto kern = ld == 16 ? ce_main_nhwc_kernel<bf16_t, 16> : ld == 24 ? ce_main_nhwc_kernel<bf16_t, 24> : ce_main_nhwc_kernel<bf16_t, 32>;
      hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, s, (const bf16_t*)a.logits_nhwc, lab8, a.weight, den, a.C, npix, loss_partial,
                         a.dlogits_nhwc, a.dlogits_ld, a.preds_u8, a.confmat);
    }
    FLAIR_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(256), 0, s, loss_partial, nb, a.loss, den);
    FLAIR_CHECK_LAUNCH();
    return 0;
  }
  ProfScope* p2 = new ProfScope("ce_main", 0.0, (double)npix * (4.0 * a.C * (1 + (a.dlogits_nchw ? 1 : 0)) + 2 + (a.dlogits_nhwc ? a.dlogits_ld * dtype_size(a.dlogits_dtype) : 0)), s);
  int nb_main = nb;
  const bool vec4 = (HW % 4) == 0 && (!a.dlogits_nhwc || a.dlogits_ld <= (a.C <= 16 ? 16 : 32)) &&
                    !(((uintptr_t)a.logits | (uintptr_t)a.dlogits_nchw) & 15) && !((uintptr_t)a.preds_u8 & 3);
  if (vec4) {
    nb_main = ce_blocks(npix / 4);
    auto kern = a.C <= 16 ? ce_main4_kernel<16> : ce_main4_kernel<32>;
    hipLaunchKernelGGL(kern, dim3(nb_main), dim3(256), 0, s, a.logits, lab8, a.weight, den, a.C, HW, npix, loss_partial,
                       a.dlogits_nchw, a.dlogits_nhwc, a.dlogits_dtype, a.dlogits_ld, a.preds_u8, a.preds_i64, a.confmat);
  } else {
    hipLaunchKernelGGL(ce_main_kernel, dim3(nb), dim3(256), 0, s, a.logits, lab8, a.weight, den, a.C, HW, npix,
                       loss_partial, a.dlogits_nchw, a.dlogits_nhwc, a.dlogits_dtype, a.dlogits_ld, a.preds_u8,
                       a.preds_i64, a.confmat);
  }
  FLAIR_CHECK_LAUNCH();
  delete p2;
  hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(256), 0, s, loss_partial, nb_main, a.loss, den);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

// predict_step (task_module.py:211-212) and zone_detect inference+convert('argmax')
// (src/zone_detect/compare.py:35, dataset.py:23-30): argmax(softmax) [+ max probability]
__global__ void softmax_argmax_kernel(const float* __restrict__ logits, int C, long HW, long npix,
                                      unsigned char* __restrict__ preds_u8, long long* __restrict__ preds_i64,
                                      float* __restrict__ maxprob) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long n = i / HW, pix = i - n * HW;
    const float* p = logits + n * C * HW + pix;
    float x[MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { x[c] = p[(long)c * HW]; m = fmaxf(m, x[c]); }
    float ssum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { x[c] = expf(x[c] - m); ssum += x[c]; }
    int pred = 0;
    float pbest = -1.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        const float q = x[c] / ssum;
        if (q > pbest) { pbest = q; pred = c; }
      }
    if (preds_u8) preds_u8[i] = (unsigned char)pred;
    if (preds_i64) preds_i64[i] = pred;
    if (maxprob) maxprob[i] = pbest;
  }
}

// four consecutive pixels per thread, 16-byte plane loads (HW % 4 == 0); same arithmetic as above
template <int CM>
__global__ __launch_bounds__(256) void softmax_argmax4_kernel(const float* __restrict__ logits, int C, long HW, long npix,
                                                              unsigned char* __restrict__ preds_u8,
                                                              long long* __restrict__ preds_i64, float* __restrict__ maxprob) {
  const long ngroups = npix >> 2;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (long)gridDim.x * blockDim.x) {
    const long i = gi << 2;
    const long n = i / HW, pix = i - n * HW;
    const float* p = logits + n * C * HW + pix;
    float4 xv[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c)
      if (c < C) xv[c] = *reinterpret_cast<const float4*>(p + (long)c * HW);
    unsigned int pred4 = 0;
    float pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x[CM];
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) { x[c] = reinterpret_cast<const float*>(&xv[c])[q]; m = fmaxf(m, x[c]); }
      float ssum = 0.f;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) { x[c] = expf(x[c] - m); ssum += x[c]; }
      int pred = 0;
      float pbest = -1.f;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < C) {
          const float qv = x[c] / ssum;
          if (qv > pbest) { pbest = qv; pred = c; }
        }
      pred4 |= (unsigned int)pred << (8 * q);
      pb[q] = pbest;
    }
    if (preds_u8) *reinterpret_cast<unsigned int*>(preds_u8 + i) = pred4;
    if (preds_i64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) preds_i64[i + q] = (pred4 >> (8 * q)) & 0xff;
    }
    if (maxprob) *reinterpret_cast<float4*>(maxprob + i) = float4{pb[0], pb[1], pb[2], pb[3]};
  }
}

// argmax(softmax) over logits kept NHWC in the network's dtype (see ce_main_nhwc_kernel)
template <typename T, int LD>
__global__ __launch_bounds__(256) void softmax_argmax_nhwc_kernel(const T* __restrict__ logits, int C, long npix,
                                                                  unsigned char* __restrict__ preds_u8,
                                                                  long long* __restrict__ preds_i64, float* __restrict__ maxprob) {
  constexpr int CH = Elem<T>::CH;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float x[MAXC];
#pragma unroll
    for (int c0 = 0; c0 < LD; c0 += CH) chunk_to_f<T>(*reinterpret_cast<const uint4*>(logits + i * LD + c0), x + c0);
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < LD; ++c)
      if (c < C) m = fmaxf(m, x[c]);
    float ssum = 0.f;
#pragma unroll
    for (int c = 0; c < LD; ++c)
      if (c < C) { x[c] = expf(x[c] - m); ssum += x[c]; }
    int pred = 0;
    float pbest = -1.f;
#pragma unroll
    for (int c = 0; c < LD; ++c)
      if (c < C) {
        const float q = x[c] / ssum;
        if (q > pbest) { pbest = q; pred = c; }
      }
    if (preds_u8) preds_u8[i] = (unsigned char)pred;
    if (preds_i64) preds_i64[i] = pred;
    if (maxprob) maxprob[i] = pbest;
  }
}

int softmax_argmax_nhwc(const void* logits, int dtype, int ld, long npix, int C, unsigned char* preds_u8, long long* preds_i64,
                        float* maxprob, hipStream_t s) {
  if (C > MAXC || C < 1 || (ld != 16 && ld != 32) || ld < C) return -2;
  ProfScope ps("softmax_argmax", 0.0, (double)npix * (ld * dtype_size(dtype) + 1), s);
  const dim3 grid(ce_blocks(npix));
  if (dtype == DT_F32) {
    auto kern = ld == 16 ? softmax_argmax_nhwc_kernel<float, 16> : softmax_argmax_nhwc_kernel<float, 32>;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, (const float*)logits, C, npix, preds_u8, preds_i64, maxprob);
  } else {
    auto kern = ld == 16 ? softmax_argmax_nhwc_kernel<bf16_t, 16> : softmax_argmax_nhwc_kernel<bf16_t, 32>;
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, (const bf16_t*)logits, C, npix, preds_u8, preds_i64, maxprob);
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int softmax_argmax(const float* logits, int B, int C, int H, int W, unsigned char* preds_u8, long long* preds_i64,
                   float* maxprob, hipStream_t s) {
  if (C > MAXC || C < 1) return -2;
  const long HW = (long)H * W, npix = HW * B;
  ProfScope ps("softmax_argmax", 0.0, (double)npix * (4.0 * C + 1), s);
  const bool aligned = !(((uintptr_t)logits | (uintptr_t)maxprob) & 15) && !((uintptr_t)preds_u8 & 3);
  if ((HW % 4) == 0 && aligned) {
    auto kern = C <= 16 ? softmax_argmax4_kernel<16> : softmax_argmax4_kernel<32>;
    hipLaunchKernelGGL(kern, dim3(ce_blocks(npix / 4)), dim3(256), 0, s, logits, C, HW, npix, preds_u8, preds_i64, maxprob);
    FLAIR_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(softmax_argmax_kernel, dim3(ce_blocks(npix)), dim3(256), 0, s, logits, C, HW, npix, preds_u8,
                     preds_i64, maxprob);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

__device__ __forceinline__ int load_label(const void* p, int kind, long i) {
  if (kind == 0) return ((const unsigned char*)p)[i];
  if (kind == 1) return ((const int*)p)[i];
  return (int)((const long long*)p)[i];
}

// confmat[target][pred] += 1  (torchmetrics update; sklearn confusion_matrix(labels=range(C)) of
// src/flair/metrics.py:67-71: pairs with an out-of-range member are dropped)
__global__ __launch_bounds__(256) void confmat_kernel(const void* __restrict__ target, int tk, const void* __restrict__ pred, int pk,
                               long n, int C, long long* __restrict__ confmat) {
  __shared__ unsigned int hist[MAXC * MAXC];
  for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = load_label(target, tk, i), p = load_label(pred, pk, i);
    if ((unsigned)t < (unsigned)C && (unsigned)p < (unsigned)C) atomicAdd(&hist[t * C + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256) {
    const unsigned int v = hist[i];
    if (v) atomicAdd(reinterpret_cast<unsigned long long*>(confmat + i), (unsigned long long)v);
  }
}

int confmat_update(const void* target, int target_kind, const void* pred, int pred_kind, long n, int C,
                   long long* confmat, hipStream_t s) {
  if (C > MAXC || C < 1 || target_kind > 2 || pred_kind > 2) return -2;
  hipLaunchKernelGGL(confmat_kernel, dim3(ce_blocks(n)), dim3(256), 0, s, target, target_kind, pred, pred_kind, n, C, confmat);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

// torchmetrics _jaccard_index_reduce: per-class (0 where denom == 0), support-weighted, macro
__global__ void jaccard_kernel(const long long* __restrict__ cm, int C, float* __restrict__ per_class,
                               float* __restrict__ weighted, float* __restrict__ macro) {
  __shared__ double jac[MAXC], sup[MAXC], den[MAXC];
  const int c = threadIdx.x;
  if (c < C) {
    double row = 0, col = 0;
    for (int j = 0; j < C; ++j) { row += (double)cm[c * C + j]; col += (double)cm[j * C + c]; }
    const double num = (double)cm[c * C + c], d = row + col - num;
    jac[c] = d != 0 ? num / d : 0.0;
    sup[c] = row;
    den[c] = d;
    if (per_class) per_class[c] = (float)jac[c];
  }
  __syncthreads();
  if (c == 0) {
    double tot = 0, ws = 0, ms = 0;
    int present = 0;
    for (int j = 0; j < C; ++j) { tot += sup[j]; ws += jac[j] * sup[j]; if (den[j] != 0) { ms += jac[j]; ++present; } }
    if (weighted) weighted[0] = tot > 0 ? (float)(ws / tot) : 0.f;
    if (macro) macro[0] = present ? (float)(ms / present) : 0.f;
  }
}

int jaccard_from_confmat(const long long* confmat, int C, float* per_class, float* weighted, float* macro, hipStream_t s) {
  if (C > MAXC || C < 1) return -2;
  hipLaunchKernelGGL(jaccard_kernel, dim3(1), dim3(64), 0, s, confmat, C, per_class, weighted, macro);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

}  // namespace flair
