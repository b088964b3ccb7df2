// BatchNorm2d (training-mode batch statistics, eval-mode running statistics) + ReLU + residual add,
// forward and backward, NHWC.  HBM-bound elementwise / reduction kernels: 16-byte accesses per lane,
// two-stage deterministic reductions, no float atomics.
// Partial sums are stored CHANNEL-MAJOR, partial[2][C][nblk], so the finalize kernels (one workgroup per
// channel, fp64 accumulation) read them fully coalesced instead of walking a strided column serially.
// Semantics follow torch.nn.BatchNorm2d(eps=1e-5, momentum=0.1) as used by smp Unet(resnet34)
// (SURVEY.md §2.3 K5/K6): biased variance for normalisation, unbiased for running_var.
#include "ops.h"
#include "prof.h"

namespace flair {

// sum of v over the 256 threads of the block (valid in every thread)
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// this thread's share of two rows of tile partials (256 threads per row): 16-byte loads, 8 loads in flight,
// fp64 accumulation in a fixed order
__device__ __forceinline__ void column_sums_f64(const float* __restrict__ p1, const float* __restrict__ p2, int nblk,
                                                double& s1, double& s2) {
  if ((nblk & 3) == 0) {
    const float4* q1 = reinterpret_cast<const float4*>(p1);
    const float4* q2 = reinterpret_cast<const float4*>(p2);
    const int n4 = nblk >> 2;
    int b = threadIdx.x;
    for (; b + 768 < n4; b += 1024) {
      float4 u[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { u[k] = q1[b + 256 * k]; v[k] = q2[b + 256 * k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1 += ((double)u[k].x + (double)u[k].y) + ((double)u[k].z + (double)u[k].w);
        s2 += ((double)v[k].x + (double)v[k].y) + ((double)v[k].z + (double)v[k].w);
      }
    }
    for (; b < n4; b += 256) {
      const float4 u = q1[b], v = q2[b];
      s1 += ((double)u.x + (double)u.y) + ((double)u.z + (double)u.w);
      s2 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
  } else {
    for (int b = threadIdx.x; b < nblk; b += 256) { s1 += (double)p1[b]; s2 += (double)p2[b]; }
  }
}

// partial [2][C][nblk] -> scale/shift (+ saved mean / invstd, running-stat update); grid = C
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float momentum, float eps, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ mean_out,
                                                          float* __restrict__ invstd_out) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  const float* p1 = partial + (long)c * nblk;
  const float* p2 = partial + ((long)C + c) * nblk;
  double s1 = 0.0, s2 = 0.0;
  column_sums_f64(p1, p2, nblk, s1, s2);
  const double a1 = block_sum_f64(s1, sh);
  const double a2 = block_sum_f64(s2, sh);
  if (threadIdx.x == 0) {
    const double mean = a1 / count;
    double var = a2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (mean_out) mean_out[c] = (float)mean;
    if (invstd_out) invstd_out[c] = invstd;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

// out = [relu]( y*scale + shift [+ res | + res*rscale + rshift] )
// The grid stride (gridDim.x * 256) is a multiple of the chunks per row (a power of two <= 128), so a thread
// always works on the same channel chunk; the coefficients are still re-read (L1 hits) every iteration
// rather than kept in 30-40 VGPRs: measured, the lost occupancy costs a streaming kernel 1.5x.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const T* __restrict__ res,
                                                     const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                     T* __restrict__ out, long rows, int C, int relu) {
  constexpr int CH = Elem<T>::CH;
  const int cpr = C / CH;
  const long total = rows * cpr;
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(first % cpr) * CH;
  for (long idx = first; idx < total; idx += (long)gridDim.x * blockDim.x) {
    float f[CH];
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(y + idx * CH), f);
#pragma unroll
    for (int e = 0; e < CH; ++e) f[e] = fmaf(f[e], scale[c + e], shift[c + e]);
    if (res) {
      float r[CH];
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(res + idx * CH), r);
      if (rscale) {
#pragma unroll
        for (int e = 0; e < CH; ++e) r[e] = fmaf(r[e], rscale[c + e], rshift[c + e]);
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) f[e] += r[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < CH; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    *reinterpret_cast<uint4*>(out + idx * CH) = f_to_chunk<T>(f);
  }
}

// partial[0][c][blk] = sum dz ; partial[1][c][blk] = sum dz * y ; dz = dout * (out > 0)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            float* __restrict__ partial, long rows, int C,
                                                            long rows_per_block, const float* __restrict__ mscale,
                                                            const float* __restrict__ mshift) {
  constexpr int CH = Elem<T>::CH;
  __shared__ float sh[256 * 2 * CH];
  const int cpr = C / CH;           // chunk columns (<= 128)
  const int rl_n = 256 / cpr;       // row lanes
  const int t = threadIdx.x;
  const int cx = t % cpr, rl = t / cpr;
  const int c = cx * CH;
  float s1[CH], s2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  // accumulates sum(dz) and sum(dz*y); the finalize kernel turns the latter into sum(dz*xhat) =
  // invstd*(sum(dz*y) - mean*sum(dz)), so no per-channel constants are held in registers here
  // (occupancy is what a streaming reduction lives on)
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  // four rows per iteration, all 8-12 loads issued before the first use (branch-free: rows past the end are
  // clamped to a valid row and weighted by zero)
  for (long r = r0 + rl; r < r1; r += 4L * rl_n) {
    uint4 dv[4], yv[4], ov[4];
    float wgt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long ru = r + (long)u * rl_n;
      const bool ok = ru < r1;
      const long off = (ok ? ru : r) * C + c;
      dv[u] = *reinterpret_cast<const uint4*>(dout + off);
      yv[u] = *reinterpret_cast<const uint4*>(y + off);
      if (out) ov[u] = *reinterpret_cast<const uint4*>(out + off);
      wgt[u] = ok ? 1.f : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float d[CH], yy[CH];
      chunk_to_f<T>(dv[u], d);
      chunk_to_f<T>(yv[u], yy);
      if (out) {
        float o[CH];
        chunk_to_f<T>(ov[u], o);
#pragma unroll
        for (int e = 0; e < CH; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
      } else if (mscale) {  // ReLU mask recomputed from the pre-BN tensor: out > 0  <=>  y*scale + shift > 0
#pragma unroll
        for (int e = 0; e < CH; ++e) d[e] = fmaf(yy[e], mscale[c + e], mshift[c + e]) > 0.f ? d[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        d[e] *= wgt[u];
        s1[e] += d[e];
        s2[e] = fmaf(d[e], yy[e], s2[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) { sh[(t * 2 + 0) * CH + e] = s1[e]; sh[(t * 2 + 1) * CH + e] = s2[e]; }
  __syncthreads();
  for (int stride = rl_n >> 1; stride > 0; stride >>= 1) {  // rl_n is a power of two
    if (rl < stride) {
      const int o = (rl + stride) * cpr + cx;
#pragma unroll
      for (int e = 0; e < 2 * CH; ++e) sh[t * 2 * CH + e] += sh[o * 2 * CH + e];
    }
    __syncthreads();
  }
  if (rl == 0) {
    const long nblk = gridDim.x;
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      partial[(long)(c + e) * nblk + blockIdx.x] = sh[(t * 2 + 0) * CH + e];
      partial[((long)C + c + e) * nblk + blockIdx.x] = sh[(t * 2 + 1) * CH + e];
    }
  }
}

// grid = C
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              const float* __restrict__ mean,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ k3) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  const float* p1 = partial + (long)c * nblk;
  const float* p2 = partial + ((long)C + c) * nblk;
  double s1 = 0.0, s2 = 0.0;
  column_sums_f64(p1, p2, nblk, s1, s2);
  const double a1 = block_sum_f64(s1, sh);
  const double a2 = (block_sum_f64(s2, sh) - (double)mean[c] * a1) * (double)invstd[c];  // sum dz*xhat
  if (threadIdx.x == 0) {
    if (dgamma) {
      dgamma[c] = accumulate ? dgamma[c] + (float)a2 : (float)a2;
      dbeta[c] = accumulate ? dbeta[c] + (float)a1 : (float)a1;
    }
    // dy = g*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)), xhat = (y - mu)*invstd  ->  A*dz + B*y + Cc
    const float gi = (gamma ? gamma[c] : 1.f) * invstd[c];
    const float m1 = (float)(a1 / count), m2 = (float)(a2 / count);
    k1[c] = gi;
    k2[c] = -gi * m2 * invstd[c];
    k3[c] = gi * (m2 * invstd[c] * mean[c] - m1);
  }
}

// dy = A*dz + B*y + Cc (coefficients from bn_bwd_finalize_kernel) ; optional dres (+)= dz ; dz = dout * relu-mask
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                           const T* __restrict__ y, const float* __restrict__ k1,
                                                           const float* __restrict__ k2, const float* __restrict__ k3,
                                                           T* __restrict__ dy, T* __restrict__ dres, int dres_accumulate,
                                                           long rows, int C, const float* __restrict__ mscale,
                                                           const float* __restrict__ mshift) {
  constexpr int CH = Elem<T>::CH;
  const int cpr = C / CH;
  const long total = rows * cpr;
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(first % cpr) * CH;  // loop-invariant: the grid stride is a multiple of cpr
  for (long idx = first; idx < total; idx += (long)gridDim.x * blockDim.x) {
    float d[CH], yy[CH];
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(dout + idx * CH), d);
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(y + idx * CH), yy);
    if (out) {
      float o[CH];
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(out + idx * CH), o);
#pragma unroll
      for (int e = 0; e < CH; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
    } else if (mscale) {
#pragma unroll
      for (int e = 0; e < CH; ++e) d[e] = fmaf(yy[e], mscale[c + e], mshift[c + e]) > 0.f ? d[e] : 0.f;
    }
    if (dres) {
      float r[CH];
      if (dres_accumulate) {
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(dres + idx * CH), r);
#pragma unroll
        for (int e = 0; e < CH; ++e) r[e] += d[e];
      } else {
#pragma unroll
        for (int e = 0; e < CH; ++e) r[e] = d[e];
      }
      *reinterpret_cast<uint4*>(dres + idx * CH) = f_to_chunk<T>(r);
    }
    float g[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) g[e] = fmaf(k1[c + e], d[e], fmaf(k2[c + e], yy[e], k3[c + e]));
    *reinterpret_cast<uint4*>(dy + idx * CH) = f_to_chunk<T>(g);
  }
}

// column sums of x[rows][ld] -> partial[ld][nblk]; two-stage, deterministic (head bias gradient)
template <typename T>
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const T* __restrict__ x, long rows, int ld,
                                                            float* __restrict__ partial, long rows_per_block) {
  constexpr int CH = Elem<T>::CH;
  __shared__ float sh[256 * CH];
  const int cpr = ld / CH, rl_n = 256 / cpr, t = threadIdx.x;
  const int cx = t % cpr, rl = t / cpr;
  float s1[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) s1[e] = 0.f;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  for (long r = r0 + rl; r < r1; r += rl_n) {
    float d[CH];
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(x + r * ld + cx * CH), d);
#pragma unroll
    for (int e = 0; e < CH; ++e) s1[e] += d[e];
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) sh[t * CH + e] = s1[e];
  __syncthreads();
  for (int stride = rl_n >> 1; stride > 0; stride >>= 1) {
    if (rl < stride) {
      const int o = (rl + stride) * cpr + cx;
#pragma unroll
      for (int e = 0; e < CH; ++e) sh[t * CH + e] += sh[o * CH + e];
    }
    __syncthreads();
  }
  if (rl == 0) {
#pragma unroll
    for (int e = 0; e < CH; ++e) partial[(long)(cx * CH + e) * gridDim.x + blockIdx.x] = sh[t * CH + e];
  }
}

// out[c] = sum_b partial[c][b]; grid = number of columns
__global__ __launch_bounds__(256) void partial_rows_sum_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
  __shared__ double sh[4];
  const float* p = partial + (long)blockIdx.x * nblk;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)p[b];
  const double a = block_sum_f64(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = (float)a;
}

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

int bn_finalize(const float* partial, int nblk, int C, long count, const float* gamma, const float* beta,
                float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                float* mean_out, float* invstd_out, hipStream_t s) {
  ProfScope ps("bn_finalize", 0.0, (double)nblk * 2 * C * 4.0, s);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, partial, nblk, C, (double)count, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                   float* scale, float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, C, gamma, beta, rm, rv, eps, scale, shift);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int bn_act(int dtype, const void* y, const float* scale, const float* shift, const void* res, const float* rscale,
           const float* rshift, void* out, long rows, int C, int relu, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch) return -2;
  const long total = rows * (C / ch);
  ProfScope ps("bn_act", 0.0, (double)rows * C * dtype_size(dtype) * (res ? 3 : 2), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(bn_act_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)y, scale, shift,
                       (const float*)res, rscale, rshift, (float*)out, rows, C, relu);
  else
    hipLaunchKernelGGL(bn_act_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)y, scale, shift,
                       (const bf16_t*)res, rscale, rshift, (bf16_t*)out, rows, C, relu);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int bn_bwd_blocks(long rows) {
  long b = (rows + 255) / 256;  // >= 256 rows per block
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

int partial_rows_sum(const float* partial, int nblk, int ncols, float* out, hipStream_t s) {
  hipLaunchKernelGGL(partial_rows_sum_kernel, dim3(ncols), dim3(256), 0, s, partial, nblk, out);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

// per-block partial (sum y, sum y^2) of a standalone tensor: [2][C][bn_bwd_blocks(rows)]
// zeros_ones: scratch of 2*C floats (filled here with mean = 0, invstd = 1)
int bn_stats_partial(int dtype, const void* y, long rows, int C, float* partial, float* zeros_ones, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch || C / ch > 128) return -2;
  int rc = fill_f32(zeros_ones, C, 0.f, s);
  if (rc) return rc;
  rc = fill_f32(zeros_ones + C, C, 1.f, s);
  if (rc) return rc;
  const int nblk = bn_bwd_blocks(rows);
  const long rpb = (rows + nblk - 1) / nblk;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), 0, s, (const float*)y, (const float*)nullptr,
                       (const float*)y, zeros_ones, zeros_ones + C, partial, rows, C, rpb, (const float*)nullptr, (const float*)nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), 0, s, (const bf16_t*)y,
                       (const bf16_t*)nullptr, (const bf16_t*)y, zeros_ones, zeros_ones + C, partial, rows, C, rpb, (const float*)nullptr, (const float*)nullptr);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int colsum(int dtype, const void* x, long rows, int ld, int C, float* partial, float* out, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (ld % ch || ld / ch > 256 || C > ld) return -2;
  const int nblk = bn_bwd_blocks(rows);
  const long rpb = (rows + nblk - 1) / nblk;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(colsum_reduce_kernel<float>, dim3(nblk), dim3(256), 0, s, (const float*)x, rows, ld, partial, rpb);
  else
    hipLaunchKernelGGL(colsum_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, rows, ld, partial, rpb);
  FLAIR_CHECK_LAUNCH();
  return partial_rows_sum(partial, nblk, C, out, s);
}

int bn_backward(int dtype, const void* dout, const void* out, const void* y, const float* mean, const float* invstd,
                const float* gamma, long rows, int C, float* partial, float* coef /*3*C*/, float* dgamma, float* dbeta,
                int accumulate_param, void* dy, void* dres, int dres_accumulate, const float* mscale, const float* mshift,
                int pre_nblk, int premasked, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch || C / ch > 128) return -2;
  if (premasked && pre_nblk <= 0) return -2;
  if (pre_nblk > 0 && !out && !mscale && !premasked) return -2;   // a producer-side reduction needs a mask source (y or out)
  const int nblk = pre_nblk > 0 ? pre_nblk : bn_bwd_blocks(rows);
  const long rpb = (rows + nblk - 1) / nblk;
  if (pre_nblk <= 0) {
    if (out) { mscale = nullptr; mshift = nullptr; }
    ProfScope ps1("bn_bwd_reduce", 0.0, (double)rows * C * dtype_size(dtype) * (out ? 3 : 2), s);
    if (dtype == DT_F32)
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), 0, s, (const float*)dout, (const float*)out,
                         (const float*)y, mean, invstd, partial, rows, C, rpb, mscale, mshift);
    else
      hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), 0, s, (const bf16_t*)dout,
                         (const bf16_t*)out, (const bf16_t*)y, mean, invstd, partial, rows, C, rpb, mscale, mshift);
    FLAIR_CHECK_LAUNCH();
  }
  float *k1 = coef, *k2 = coef + C, *k3 = coef + 2 * C;
  {
    ProfScope ps3("bn_bwd_finalize", 0.0, (double)nblk * 2 * C * 4.0, s);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, partial, nblk, C, (double)rows, gamma, invstd,
                       mean, dgamma, dbeta, accumulate_param, k1, k2, k3);
  }
  FLAIR_CHECK_LAUNCH();
  if (!dy && !dres) return 0;   // the consumer applies dy = k1*dz + k2*y + k3 itself (stem: inside its weight-gradient kernel)
  const long total = rows * (C / ch);
  ProfScope ps2("bn_bwd_apply", 0.0, (double)rows * C * dtype_size(dtype) * ((out ? 3 : 2) + 1 + (dres ? 1 : 0)), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)dout,
                       (const float*)out, (const float*)y, k1, k2, k3, (float*)dy, (float*)dres,
                       dres_accumulate, rows, C, mscale, mshift);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)dout,
                       (const bf16_t*)out, (const bf16_t*)y, k1, k2, k3, (bf16_t*)dy, (bf16_t*)dres,
                       dres_accumulate, rows, C, mscale, mshift);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

}  // namespace flair
