// HBM-bound helpers of the U-Net hot path (NHWC, 16 bytes per lane): max-pool 3x3/s2 fwd+bwd,
// backward of the fused nearest-x2-upsample + concat, NCHW fp32 <-> NHWC T boundary converters,
// weight packing (PyTorch OIHW fp32 master -> [Cout][(r,s),c] T, optionally flipped/transposed for
// the data gradient), plain SGD (src/flair/tasks_utils.py:95) and small utilities.
#include "ops.h"
#include "prof.h"

namespace flair {

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ---------------------------------------------------------------- maxpool 3x3 stride 2 pad 1
// First maximum in (kh, kw) scan order wins, like torch.nn.MaxPool2d on CPU; idx keeps the tap.
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, unsigned char* __restrict__ idx,
                                   int N, int H, int W, int C) {
  constexpr int CH = Elem<T>::CH;
  // one workgroup per output row (n, ho): 32-bit index arithmetic (the flat 64-bit div / mod chain of round 1 was a third of
  // this kernel's ~750 vector instructions per item — it ran at 3.2 TB/s, issue-bound)
  const int Ho = H / 2, Wo = W / 2, cpr = C / CH;
  const int ho = blockIdx.x % Ho, n = blockIdx.x / Ho;
  const unsigned row_items = (unsigned)Wo * cpr;
  for (unsigned it = threadIdx.x; it < row_items; it += blockDim.x) {
    const int cx = (int)(it % (unsigned)cpr), wo = (int)(it / (unsigned)cpr);
    const long i = ((long)blockIdx.x * Wo + wo) * cpr + cx;
    float best[CH];
    unsigned char bi[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int r = 0; r < 3; ++r) {
      const int h = 2 * ho - 1 + r;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int q = 0; q < 3; ++q) {
        const int w = 2 * wo - 1 + q;
        if ((unsigned)w >= (unsigned)W) continue;
        float f[CH];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(in + (((long)n * H + h) * W + w) * C + cx * CH), f);
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          if (first || f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; bi[e] = (unsigned char)(r * 3 + q); }
        }
        first = false;
      }
    }
    *reinterpret_cast<uint4*>(out + i * CH) = f_to_chunk<T>(best);
    if (idx) {
      if constexpr (CH == 8) {
        uint2 iv;
        iv.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
        iv.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
        *reinterpret_cast<uint2*>(idx + i * CH) = iv;
      } else {
        *reinterpret_cast<unsigned*>(idx + i * CH) =
            (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
      }
    }
  }
}

// The stem's BatchNorm + ReLU and the 3x3 / stride-2 max pool in one pass over the pre-BN tensor: a thread normalises the nine
// taps of its window (the rounding sequence of bn_act: fmaf, max, round to T), pools the rounded values exactly like
// maxpool_fwd_kernel does on the stored activation, and writes the activation of the 2 x 2 pixels it owns (rows 2 ho, 2 ho + 1,
// columns 2 wo, 2 wo + 1 = taps r, q in {1, 2}).  Saves the pool's re-read of the 268 MB activation.
template <typename T>
__global__ void bn_act_maxpool_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                      T* __restrict__ act, T* __restrict__ out, unsigned char* __restrict__ idx, int N, int H, int W, int C) {
  constexpr int CH = Elem<T>::CH;
  const int Ho = H / 2, Wo = W / 2, cpr = C / CH;
  const int ho = blockIdx.x % Ho, n = blockIdx.x / Ho;
  const unsigned row_items = (unsigned)Wo * cpr;
  for (unsigned it = threadIdx.x; it < row_items; it += blockDim.x) {
    const int cx = (int)(it % (unsigned)cpr), wo = (int)(it / (unsigned)cpr);
    const long i = ((long)blockIdx.x * Wo + wo) * cpr + cx;
    float sc[CH], sh[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { sc[e] = scale[cx * CH + e]; sh[e] = shift[cx * CH + e]; }
    float best[CH];
    unsigned char bi[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int r = 0; r < 3; ++r) {
      const int h = 2 * ho - 1 + r;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int q = 0; q < 3; ++q) {
        const int w = 2 * wo - 1 + q;
        if ((unsigned)w >= (unsigned)W) continue;
        const long off = (((long)n * H + h) * W + w) * C + cx * CH;
        float f[CH];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(y + off), f);
#pragma unroll
        for (int e = 0; e < CH; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
        const uint4 a = f_to_chunk<T>(f);
        if (r >= 1 && q >= 1) *reinterpret_cast<uint4*>(act + off) = a;
        chunk_to_f<T>(a, f);   // the stored (rounded) activation is what is pooled
#pragma unroll
        for (int e = 0; e < CH; ++e) {
          if (first || f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; bi[e] = (unsigned char)(r * 3 + q); }
        }
        first = false;
      }
    }
    *reinterpret_cast<uint4*>(out + i * CH) = f_to_chunk<T>(best);
    if (idx) {
      if constexpr (CH == 8) {
        uint2 iv;
        iv.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
        iv.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
        *reinterpret_cast<uint2*>(idx + i * CH) = iv;
      } else {
        *reinterpret_cast<unsigned*>(idx + i * CH) =
            (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
      }
    }
  }
}

// bnr_*: optional first pass of the BatchNorm backward of the unit whose ReLU output this gradient belongs to (the stem): the
// kernel completes that gradient, so it also leaves sum(dz*m), sum(dz*m*y), m = [y*msc + msh > 0], per workgroup in
// bnr_partial[2][C][gridDim.x] — what bn_bwd_reduce_kernel would re-read the 268 MB tensor for.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ idx,
                                   T* __restrict__ din, int accumulate, int N, int H, int W, int C,
                                   const T* __restrict__ bnr_y, const float* __restrict__ bnr_msc, const float* __restrict__ bnr_msh,
                                   float* __restrict__ bnr_partial) {
  constexpr int CH = Elem<T>::CH;
  __shared__ float bsh[256 * 2 * CH];
  float r1[CH], r2[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { r1[e] = 0.f; r2[e] = 0.f; }
  const int Ho = H / 2, Wo = W / 2, cpr = C / CH;
  const int h = blockIdx.x % H, n = blockIdx.x / H;   // one workgroup per input row, 32-bit index arithmetic
  const unsigned row_items = (unsigned)W * cpr;
  for (unsigned it = threadIdx.x; it < row_items; it += blockDim.x) {
    const int cx = (int)(it % (unsigned)cpr), w = (int)(it / (unsigned)cpr);
    const long i = ((long)blockIdx.x * W + w) * cpr + cx;
    float g[CH];
    if (accumulate) chunk_to_f<T>(*reinterpret_cast<const uint4*>(din + i * CH), g);
    else {
#pragma unroll
      for (int e = 0; e < CH; ++e) g[e] = 0.f;
    }
    for (int ho = h / 2; ho <= (h + 1) / 2; ++ho) {
      if (ho >= Ho) continue;
      const int r = h - (2 * ho - 1);
      for (int wo = w / 2; wo <= (w + 1) / 2; ++wo) {
        if (wo >= Wo) continue;
        const int q = w - (2 * wo - 1);
        const unsigned char tap = (unsigned char)(r * 3 + q);
        const long o = ((((long)n * Ho + ho) * Wo + wo) * cpr + cx) * CH;
        float d[CH];
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(dout + o), d);
        // the CH tap indices of the chunk in ONE load (round 1 read them byte by byte: 9 loads per contributor, 1.7 TB/s)
        unsigned char ti[CH];
        if constexpr (CH == 8) {
          const uint2 iv = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) { ti[e] = (unsigned char)(iv.x >> (8 * e)); ti[4 + e] = (unsigned char)(iv.y >> (8 * e)); }
        } else {
          const unsigned iv = *reinterpret_cast<const unsigned*>(idx + o);
#pragma unroll
          for (int e = 0; e < CH; ++e) ti[e] = (unsigned char)(iv >> (8 * e));
        }
#pragma unroll
        for (int e = 0; e < CH; ++e)
          if (ti[e] == tap) g[e] += d[e];
      }
    }
    const uint4 gv = f_to_chunk<T>(g);
    *reinterpret_cast<uint4*>(din + i * CH) = gv;
    if (bnr_partial) {   // a thread keeps its channel chunk cx (blockDim.x % cpr == 0)
      float d[CH], yy[CH];
      chunk_to_f<T>(gv, d);
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(bnr_y + i * CH), yy);
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        const float dm = fmaf(yy[e], bnr_msc[cx * CH + e], bnr_msh[cx * CH + e]) > 0.f ? d[e] : 0.f;
        r1[e] += dm;
        r2[e] = fmaf(dm, yy[e], r2[e]);
      }
    }
  }
  if (bnr_partial) {   // fixed-order sum over the threads of a chunk column
    const int t = threadIdx.x;
#pragma unroll
    for (int e = 0; e < CH; ++e) { bsh[(t * 2) * CH + e] = r1[e]; bsh[(t * 2 + 1) * CH + e] = r2[e]; }
    __syncthreads();
    if (t < C) {
      const int cx = t / CH, e = t - cx * CH;
      float x1 = 0.f, x2 = 0.f;
      for (int th = cx; th < (int)blockDim.x; th += cpr) { x1 += bsh[(th * 2) * CH + e]; x2 += bsh[(th * 2 + 1) * CH + e]; }
      bnr_partial[(long)t * gridDim.x + blockIdx.x] = x1;
      bnr_partial[((long)C + t) * gridDim.x + blockIdx.x] = x2;
    }
  }
}

// ---------------------------------------------------------------- backward of cat([up2(x0), skip])
template <typename T>
__global__ void upcat_bwd_kernel(const T* __restrict__ dcat, T* __restrict__ dx0, int dx0_acc, T* __restrict__ dskip,
                                 int dskip_acc, int N, int H, int W, int C0, int C1) {
  constexpr int CH = Elem<T>::CH;
  const int Ct = C0 + C1;
  const int c0r = C0 / CH, c1r = C1 / CH;
  const int Hh = H / 2, Wh = W / 2;
  const long n0 = (long)N * Hh * Wh * c0r;
  const long n1 = (long)N * H * W * c1r;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += (long)gridDim.x * blockDim.x) {
    if (i < n0) {
      const int cx = (int)(i % c0r);
      long p = i / c0r;
      const int w = (int)(p % Wh); p /= Wh;
      const int h = (int)(p % Hh);
      const int n = (int)(p / Hh);
      float g[CH];
      if (dx0_acc) chunk_to_f<T>(*reinterpret_cast<const uint4*>(dx0 + i * CH), g);
      else {
#pragma unroll
        for (int e = 0; e < CH; ++e) g[e] = 0.f;
      }
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          float d[CH];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>(dcat + (((long)n * H + 2 * h + dh) * W + 2 * w + dw) * Ct + cx * CH), d);
#pragma unroll
          for (int e = 0; e < CH; ++e) g[e] += d[e];
        }
      *reinterpret_cast<uint4*>(dx0 + i * CH) = f_to_chunk<T>(g);
    } else {
      const long j = i - n0;
      const int cx = (int)(j % c1r);
      const long p = j / c1r;
      uint4 v = *reinterpret_cast<const uint4*>(dcat + p * Ct + C0 + cx * CH);
      if (dskip_acc) {
        float a[CH], b[CH];
        chunk_to_f<T>(v, a);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(dskip + j * CH), b);
#pragma unroll
        for (int e = 0; e < CH; ++e) a[e] += b[e];
        v = f_to_chunk<T>(a);
      }
      *reinterpret_cast<uint4*>(dskip + j * CH) = v;
    }
  }
}

// ---------------------------------------------------------------- layout converters at the NCHW fp32 boundary
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int C, long HW, int Cp) {
  constexpr int CH = Elem<T>::CH;
  const long total = (long)N * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / HW, pix = i - n * HW;
    const float* src = in + n * C * HW + pix;
    for (int c0 = 0; c0 < Cp; c0 += CH) {
      float f[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) f[e] = (c0 + e) < C ? src[(long)(c0 + e) * HW] : 0.f;
      *reinterpret_cast<uint4*>(out + i * Cp + c0) = f_to_chunk<T>(f);
    }
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int C, long HW, int Cp,
                                    int accumulate) {
  constexpr int CH = Elem<T>::CH;
  const long total = (long)N * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / HW, pix = i - n * HW;
    float* dst = out + n * C * HW + pix;
    for (int c0 = 0; c0 < Cp; c0 += CH) {
      float f[CH];
      chunk_to_f<T>(*reinterpret_cast<const uint4*>(in + i * Cp + c0), f);
#pragma unroll
      for (int e = 0; e < CH; ++e)
        if (c0 + e < C) {
          float* d = dst + (long)(c0 + e) * HW;
          *d = accumulate ? (*d + f[e]) : f[e];
        }
    }
  }
}

// ---------------------------------------------------------------- weight packing
// forward:  dst[k][(r*S+s)*Cin_p + c] = w[k][c][r][s]
// dgrad  :  dst[c][(r*S+s)*Cin_p + k] = w[k][c][R-1-r][S-1-s]   (rows = forward Cin, Cin_p = padded forward Cout)
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int R, int S,
                                   int Cin_p, int rows_pad, int Kpad, int tf) {
  const long total = (long)rows_pad * Kpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i / Kpad), kk = (int)(i - (long)row * Kpad);
    const int tap = kk / Cin_p, c = kk - tap * Cin_p;
    float v = 0.f;
    if (tap < R * S) {
      const int r = tap / S, q = tap - r * S;
      if (!tf) {
        if (row < Cout && c < Cin) v = w[(((long)row * Cin + c) * R + r) * S + q];
      } else {
        if (row < Cin && c < Cout) v = w[(((long)c * Cin + row) * R + (R - 1 - r)) * S + (S - 1 - q)];
      }
    }
    dst[i] = Elem<T>::from_f(v);
  }
}

// all layers of the network in ONE launch (blockIdx.y = layer): the per-layer table travels as a kernel argument.
// The fp32 master is OIHW, i.e. for one (k, c) pair the R*S taps are contiguous.  Reads therefore go (k, c)-major with c
// fastest — a wave covers 64 x R*S contiguous floats — and
//   forward packs    write dst[k][tap][c] straight from registers (64 consecutive c per tap: 128-byte runs);
//   data-grad packs  (dst[c][tap'][k], flipped taps) go through a 32(k) x 32(c) x taps LDS tile so that the writes
//                    run along k (64-byte runs) instead of scattering 2-byte elements over 64 rows.
// Padding (rows >= real rows, channels >= real channels, row tails) is zero-filled by a second sweep without loads.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_all_kernel(const float* __restrict__ params, unsigned char* __restrict__ base,
                                                              const PackTable tb) {
  __shared__ float tile[32 * 32 * 9 + 32];
  const PackDesc d = tb.d[blockIdx.y];
  const float* __restrict__ w = params + d.w_off;
  T* __restrict__ dst = reinterpret_cast<T*>(base + d.dst_off);
  const int RS = d.R * d.S;
  const int Rc = d.Rc > 0 ? d.Rc : d.R, Sc = d.Rc > 0 ? d.Sc : d.S;
  const int r0 = d.Rc > 0 ? d.r0 : 0, rstep = d.Rc > 0 ? d.rstep : 1, s0 = d.Rc > 0 ? d.s0 : 0, sstep = d.Rc > 0 ? d.sstep : 1;
  const int RSc = Rc * Sc;
  const int t = threadIdx.x;
  if (!d.tf && RS == 9 && d.Rc == 0 && (d.Cin & 3) == 0 && sizeof(T) == 2) {
    // four input channels per thread: 36 contiguous floats in (nine 16-byte loads), one 8-byte store per tap out (the
    // one-element version below issued nine 2-byte stores and a 64-bit division per element: 2 TB/s)
    const int c4n = d.Cin >> 2, nitems = d.Cout * c4n;
    for (int item = blockIdx.x * 256 + t; item < nitems; item += gridDim.x * 256) {
      const int k = item / c4n, c = (item - k * c4n) << 2;
      const float4* src = reinterpret_cast<const float4*>(w + ((long)k * d.Cin + c) * 9);
      float f[36];
#pragma unroll
      for (int j = 0; j < 9; ++j) { const float4 v = src[j]; f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w; }
      T* drow = dst + (long)k * d.Kpad + c;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(f[tp]) | ((unsigned)f32_to_bf16(f[9 + tp]) << 16);
        pk.y = (unsigned)f32_to_bf16(f[18 + tp]) | ((unsigned)f32_to_bf16(f[27 + tp]) << 16);
        *reinterpret_cast<uint2*>(drow + tp * d.Cin_p) = pk;
      }
    }
  } else if (!d.tf) {
    const long nitems = (long)d.Cout * d.Cin;
    for (long item = (long)blockIdx.x * 256 + t; item < nitems; item += (long)gridDim.x * 256) {
      const int k = (int)(item / d.Cin), c = (int)(item - (long)k * d.Cin);
      const float* src = w + item * RS;
      T* drow = dst + (long)k * d.Kpad + c;
      for (int tp = 0; tp < RSc; ++tp) {
        const int r = r0 + (tp / Sc) * rstep, q = s0 + (tp % Sc) * sstep;
        drow[tp * d.Cin_p] = Elem<T>::from_f(src[r * d.S + q]);
      }
    }
  } else if (RS <= 9) {
    // rows of the pack = forward input channels c (d.Cin of them), columns = (tap', k) with k < d.Cout
    const int ct = (d.Cin + 31) / 32, kt = (d.Cout + 31) / 32;
    const bool fast = RS == 9 && sizeof(T) == 2 && (d.Cin & 31) == 0 && (d.Cout & 31) == 0;
    for (int tl = blockIdx.x; tl < ct * kt; tl += gridDim.x) {
      const int c0 = (tl % ct) * 32, k0 = (tl / ct) * 32;
      __syncthreads();
      if (fast) {
        // 32 output channels x (32 input channels x 9 taps = 288 contiguous floats): 16-byte loads, LDS rows of 289 floats (reads
        // along k hit 32 different banks), two output channels per 4-byte store (round 2a: scalar loads, 8-way bank conflicts and
        // 2-byte stores: 110 us for the data-gradient packs of a step)
        for (int e = t; e < 32 * 72; e += 256) {
          const int kk = e / 72, j = e - kk * 72;
          const float4 v = *reinterpret_cast<const float4*>(w + ((long)(k0 + kk) * d.Cin + c0) * 9 + 4 * j);
          float* tr = tile + kk * 289 + 4 * j;
          tr[0] = v.x; tr[1] = v.y; tr[2] = v.z; tr[3] = v.w;
        }
        __syncthreads();
        for (int e = t; e < 16 * 32 * RSc; e += 256) {   // store: k pairs fastest
          const int k2 = e & 15, rest = e >> 4;
          const int tp = rest % RSc, cc = rest / RSc;
          const int r = r0 + (tp / Sc) * rstep, q = s0 + (tp % Sc) * sstep;
          const int src = cc * 9 + (d.R - 1 - r) * d.S + (d.S - 1 - q);
          const unsigned pk = (unsigned)f32_to_bf16(tile[(2 * k2) * 289 + src]) | ((unsigned)f32_to_bf16(tile[(2 * k2 + 1) * 289 + src]) << 16);
          *reinterpret_cast<unsigned*>(dst + (long)(c0 + cc) * d.Kpad + tp * d.Cin_p + k0 + 2 * k2) = pk;
        }
        continue;
      }
      for (int e = t; e < 32 * 32; e += 256) {       // load: c fastest
        const int kk = e >> 5, cc = e & 31;
        if (k0 + kk < d.Cout && c0 + cc < d.Cin) {
          const float* src = w + ((long)(k0 + kk) * d.Cin + c0 + cc) * RS;
          for (int tp = 0; tp < RS; ++tp) tile[(kk * 32 + cc) * RS + tp + (kk >> 3)] = src[tp];
        }
      }
      __syncthreads();
      for (int e = t; e < 32 * 32 * RSc; e += 256) {   // store: k fastest
        const int kk = e & 31, rest = e >> 5;
        const int tp = rest % RSc, cc = rest / RSc;
        if (k0 + kk < d.Cout && c0 + cc < d.Cin) {
          const int r = r0 + (tp / Sc) * rstep, q = s0 + (tp % Sc) * sstep;
          const float v = tile[(kk * 32 + cc) * RS + (d.R - 1 - r) * d.S + (d.S - 1 - q) + (kk >> 3)];
          dst[(long)(c0 + cc) * d.Kpad + tp * d.Cin_p + k0 + kk] = Elem<T>::from_f(v);
        }
      }
    }
  } else {   // large filters never need a data-gradient pack in this network; keep a plain path for the op-level API
    const long total = (long)d.Cin * d.Cout * RSc;
    for (long i = (long)blockIdx.x * 256 + t; i < total; i += (long)gridDim.x * 256) {
      const int k = (int)(i % d.Cout);
      const long rest = i / d.Cout;
      const int tp = (int)(rest % RSc), c = (int)(rest / RSc);
      const int r = r0 + (tp / Sc) * rstep, q = s0 + (tp % Sc) * sstep;
      dst[(long)c * d.Kpad + tp * d.Cin_p + k] =
          Elem<T>::from_f(w[(((long)k * d.Cin + c) * d.R + (d.R - 1 - r)) * d.S + (d.S - 1 - q)]);
    }
  }
  // zero fill of the padding: real region = rows < nrows, per tap columns < ncols
  const int nrows = d.tf ? d.Cin : d.Cout, ncols = d.tf ? d.Cout : d.Cin;
  const long total = (long)d.rows_pad * d.Kpad;
  for (long i = (long)blockIdx.x * 256 + t; i < total; i += (long)gridDim.x * 256) {
    const int row = (int)(i / d.Kpad), kk = (int)(i - (long)row * d.Kpad);
    const int tp = kk / d.Cin_p, c = kk - tp * d.Cin_p;
    if (row >= nrows || tp >= RSc || c >= ncols) dst[i] = Elem<T>::from_f(0.f);
  }
}

int pack_weights_all(int dtype, const float* params, void* base, const PackTable& tb, hipStream_t s) {
  if (tb.n < 1 || tb.n > PackTable::MAX) return -2;
  double bytes = 0;
  for (int i = 0; i < tb.n; ++i) bytes += (double)tb.d[i].Cout * tb.d[i].Cin * tb.d[i].R * tb.d[i].S * 4.0 + (double)tb.d[i].rows_pad * tb.d[i].Kpad * dtype_size(dtype);
  ProfScope ps("pack_weights_all", 0.0, bytes, s);
  dim3 grid(256, tb.n);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(pack_weights_all_kernel<float>, grid, dim3(256), 0, s, params, (unsigned char*)base, tb);
  else
    hipLaunchKernelGGL(pack_weights_all_kernel<bf16_t>, grid, dim3(256), 0, s, params, (unsigned char*)base, tb);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, long n, float lr) {
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(p)[i];
    const float4 b = reinterpret_cast<const float4*>(g)[i];
    a.x -= lr * b.x; a.y -= lr * b.y; a.z -= lr * b.z; a.w -= lr * b.w;
    reinterpret_cast<float4*>(p)[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = n4 * 4 + threadIdx.x;
    p[i] -= lr * g[i];
  }
}

// x[n][c][h][w] += v[n][h]   (metadata fusion layout quirk, src/flair/model.py:59-60: the 16-d
// encoding varies along H only, constant over channels and W)
__global__ void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ v, int N, int C, int H, int W) {
  const long total = (long)N * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int h = (int)((i / W) % H);
    const int n = (int)(i / ((long)C * H * W));
    x[i] += v[n * H + h];
  }
}

template <typename T>
__global__ void ew_add_kernel(T* __restrict__ dst, const T* __restrict__ src, long nchunks) {
  constexpr int CH = Elem<T>::CH;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float a[CH], b[CH];
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(dst + i * CH), a);
    chunk_to_f<T>(*reinterpret_cast<const uint4*>(src + i * CH), b);
#pragma unroll
    for (int e = 0; e < CH; ++e) a[e] += b[e];
    *reinterpret_cast<uint4*>(dst + i * CH) = f_to_chunk<T>(a);
  }
}

int maxpool3x3s2_fwd(int dtype, const void* in, void* out, unsigned char* idx, int N, int H, int W, int C, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch || (H & 1) || (W & 1)) return -2;
  ProfScope ps("maxpool_fwd", 0.0, (double)N * H * W * C * dtype_size(dtype) * 1.25 + (double)N * H * W * C / 4, s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(N * (H / 2)), dim3(256), 0, s, (const float*)in, (float*)out, idx, N, H, W, C);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(N * (H / 2)), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, idx, N, H, W, C);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int bn_act_maxpool3x3s2(int dtype, const void* y, const float* scale, const float* shift, void* act, void* out, unsigned char* idx, int N, int H,
                        int W, int C, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch || (H & 1) || (W & 1) || (256 % (C / ch))) return -2;
  ProfScope ps("bn_act_maxpool", 0.0, (double)N * H * W * C * dtype_size(dtype) * 2.25 + (double)N * H * W * C / 4, s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(bn_act_maxpool_kernel<float>, dim3(N * (H / 2)), dim3(256), 0, s, (const float*)y, scale, shift, (float*)act, (float*)out, idx,
                       N, H, W, C);
  else
    hipLaunchKernelGGL(bn_act_maxpool_kernel<bf16_t>, dim3(N * (H / 2)), dim3(256), 0, s, (const bf16_t*)y, scale, shift, (bf16_t*)act, (bf16_t*)out,
                       idx, N, H, W, C);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int maxpool3x3s2_bwd(int dtype, const void* dout, const unsigned char* idx, void* din, int accumulate, int N, int H,
                     int W, int C, hipStream_t s, const void* bnr_y, const float* bnr_msc, const float* bnr_msh,
                     float* bnr_partial) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C % ch || (H & 1) || (W & 1)) return -2;
  if (bnr_partial && (C > 256 || 256 % (C / ch) || !bnr_y || !bnr_msc || !bnr_msh)) return -2;
  ProfScope ps("maxpool_bwd", 0.0, (double)N * H * W * C * (dtype_size(dtype) * (1.25 + accumulate) + 0.25), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(N * H), dim3(256), 0, s, (const float*)dout, idx, (float*)din, accumulate, N, H, W, C,
                       (const float*)bnr_y, bnr_msc, bnr_msh, bnr_partial);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(N * H), dim3(256), 0, s, (const bf16_t*)dout, idx, (bf16_t*)din, accumulate, N, H, W, C,
                       (const bf16_t*)bnr_y, bnr_msc, bnr_msh, bnr_partial);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int upcat_bwd(int dtype, const void* dcat, void* dx0, int dx0_accumulate, void* dskip, int dskip_accumulate, int N,
              int H, int W, int C0, int C1, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (C0 % ch || C1 % ch || (H & 1) || (W & 1)) return -2;
  const long total = (long)N * (H / 2) * (W / 2) * (C0 / ch) + (long)N * H * W * (C1 / ch);
  ProfScope ps("upcat_bwd", 0.0, ((double)N * H * W * (C0 + C1) + (double)N * H * W * (C0 / 4.0 * (1 + dx0_accumulate) + C1 * (1 + dskip_accumulate))) * dtype_size(dtype), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(upcat_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)dcat, (float*)dx0, dx0_accumulate, (float*)dskip, dskip_accumulate, N, H, W, C0, C1);
  else
    hipLaunchKernelGGL(upcat_bwd_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)dcat, (bf16_t*)dx0, dx0_accumulate, (bf16_t*)dskip, dskip_accumulate, N, H, W, C0, C1);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int nchw_f32_to_nhwc(int dtype, const float* in, void* out, int N, int C, int H, int W, int Cp, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (Cp % ch || Cp < C) return -2;
  const long total = (long)N * H * W;
  ProfScope ps("nchw_to_nhwc", 0.0, (double)N * H * W * (C * 4.0 + Cp * dtype_size(dtype)), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, in, (float*)out, N, C, (long)H * W, Cp);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, in, (bf16_t*)out, N, C, (long)H * W, Cp);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int nhwc_to_nchw_f32(int dtype, const void* in, float* out, int N, int C, int H, int W, int Cp, float* accumulate_into,
                     hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (Cp % ch || Cp < C) return -2;
  const long total = (long)N * H * W;
  float* dst = accumulate_into ? accumulate_into : out;
  const int acc = accumulate_into ? 1 : 0;
  ProfScope ps("nhwc_to_nchw", 0.0, (double)N * H * W * (C * 4.0 + Cp * dtype_size(dtype)), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, (const float*)in, dst, N, C, (long)H * W, Cp, acc);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, (const bf16_t*)in, dst, N, C, (long)H * W, Cp, acc);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int pack_weight(int dtype, const float* w_oihw, void* dst, int Cout, int Cin, int R, int S, int Cin_p, int rows_pad,
                int Kpad, int transpose_flip, hipStream_t s) {
  const long total = (long)rows_pad * Kpad;
  ProfScope ps("pack_weight", 0.0, (double)Cout * Cin * R * S * 4.0 + (double)total * dtype_size(dtype), s);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, s, w_oihw, (float*)dst, Cout, Cin, R, S, Cin_p, rows_pad, Kpad, transpose_flip);
  else
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, s, w_oihw, (bf16_t*)dst, Cout, Cin, R, S, Cin_p, rows_pad, Kpad, transpose_flip);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int sgd_step(float* params, const float* grads, long n, float lr, hipStream_t s) {
  if (((uintptr_t)params | (uintptr_t)grads) & 15) return -2;
  ProfScope ps("sgd", 0.0, (double)n * 12.0, s);
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, s, params, grads, n, lr);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int add_rowvec_nchw(float* x, const float* v, int N, int C, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(ew_blocks((long)N * C * H * W)), dim3(256), 0, s, x, v, N, C, H, W);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int ew_add(int dtype, void* dst, const void* src, long n, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if (n % ch) return -2;
  if (dtype == DT_F32)
    hipLaunchKernelGGL(ew_add_kernel<float>, dim3(ew_blocks(n / ch)), dim3(256), 0, s, (float*)dst, (const float*)src, n / ch);
  else
    hipLaunchKernelGGL(ew_add_kernel<bf16_t>, dim3(ew_blocks(n / ch)), dim3(256), 0, s, (bf16_t*)dst, (const bf16_t*)src, n / ch);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

__global__ void fill_f32_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
int fill_f32(float* p, long n, float v, hipStream_t s) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, p, n, v);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int fill_zero(void* p, size_t bytes, hipStream_t s) { return (int)hipMemsetAsync(p, 0, bytes, s); }

}  // namespace flair
