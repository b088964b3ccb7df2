// The rows either side of the U-Net hot path, as HBM-bound byte kernels:
//   feed_tiles      uint8 raster bands (+ raw uint8 label raster) -> the batch the step consumes: band
//                   selection, normalisation ('custom' in fp64 like numpy, 'scaling', 'without'), the D4
//                   augmentation (V-flip, H-flip, rot90^k) and label decoding, in one pass.
//                   src/flair/data_loader.py:9-30,65-95; src/flair/tasks_utils.py:37-41.
//   detect_convert  softmax -> margin crop -> convert('argmax' | 'class_prob') of zone_detect.
//                   src/zone_detect/compare.py:35,71-76; src/zone_detect/dataset.py:11-34.
//   confmat_masks   offline evaluation: confusion matrix of (truth raster - 1) against a prediction raster.
//                   src/flair/metrics.py:60-75.
#include "ops.h"
#include "prof.h"

namespace flair {

namespace {

constexpr int MAXC = 32;

// where output pixel (i, j) of rot90^k(hflip(vflip(a))) comes from (numpy: rot90 is counter-clockwise,
// r[i][j] = m[j][N-1-i]); flags: bit0 V-flip, bit1 H-flip, bits 2-3 k
__device__ __forceinline__ void d4_source(int flags, int i, int j, int H, int W, int& si, int& sj) {
  const int k = (flags >> 2) & 3;
  if (k == 0) { si = i; sj = j; }
  else if (k == 1) { si = j; sj = W - 1 - i; }
  else if (k == 2) { si = H - 1 - i; sj = W - 1 - j; }
  else { si = H - 1 - j; sj = i; }
  if (flags & 2) sj = W - 1 - sj;
  if (flags & 1) si = H - 1 - si;
}

__device__ __forceinline__ float norm_byte(int mode, unsigned char v, double mean, double stdv) {
  if (mode == 2) return (float)(((double)v - mean) / stdv);  // numpy: float64 subtract, float64 divide, then float32
  if (mode == 1) return (float)((double)v * (1.0 / 255.0));  // skimage img_as_float(uint8), then float32
  return (float)v;
}

__global__ __launch_bounds__(256) void feed_tiles_kernel(FeedArgs a) {
  const int W4 = a.W / 4;
  const long per_img = (long)a.H * W4;
  const long total = per_img * a.B;
  const long HW = (long)a.H * a.W;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int b = (int)(t / per_img);
    const long r = t - (long)b * per_img;
    const int i = (int)(r / W4), j0 = (int)(r - (long)i * W4) * 4;
    const int flags = a.d4 ? a.d4[b] : 0;
    long src[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int si, sj;
      d4_source(flags, i, j0 + e, a.H, a.W, si, sj);
      src[e] = (long)si * a.W + sj;
    }
    if (a.out) {
      for (int c = 0; c < a.Cout; ++c) {
        const unsigned char* p = a.img + ((long)b * a.Cb + a.band[c]) * HW;
        const double m = a.mean[c], sd = a.stdv[c];
        float4 o;
        o.x = norm_byte(a.mode, p[src[0]], m, sd);
        o.y = norm_byte(a.mode, p[src[1]], m, sd);
        o.z = norm_byte(a.mode, p[src[2]], m, sd);
        o.w = norm_byte(a.mode, p[src[3]], m, sd);
        *reinterpret_cast<float4*>(a.out + ((long)b * a.Cout + c) * HW + (long)i * a.W + j0) = o;
      }
    }
    if (a.labels) {
      const unsigned char* p = a.msk + (long)b * HW;
      unsigned int packed = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // read_msk: (raw - 1) in uint8, one-hot over range(C); argmax of an all-zero one-hot is class 0
        const unsigned char v = (unsigned char)(p[src[e]] - 1);
        packed |= (unsigned int)(v < a.num_classes ? v : 0) << (8 * e);
      }
      *reinterpret_cast<unsigned int*>(a.labels + (long)b * HW + (long)i * a.W + j0) = packed;
    }
  }
}

// zone_detect Sliced_Dataset.__getitem__ (src/zone_detect/dataset.py:90-113) for a batch of windows of ONE raster held
// in HBM: boundless read (pixels outside the raster are 0 BEFORE normalisation), band selection, normalisation.
__global__ __launch_bounds__(256) void gather_tiles_kernel(FeedArgs a, const int* __restrict__ tiles, int Hr, int Wr) {
  const int S = a.W, S4 = S / 4;
  const long per_img = (long)S * S4, total = per_img * a.B;
  const long plane = (long)Hr * Wr;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int b = (int)(t / per_img);
    const long r = t - (long)b * per_img;
    const int i = (int)(r / S4), j0 = (int)(r - (long)i * S4) * 4;
    const int x0 = tiles[b * 6 + 0], y0 = tiles[b * 6 + 1];
    const int y = y0 + i;
    const bool yin = (unsigned)y < (unsigned)Hr;
    for (int c = 0; c < a.Cout; ++c) {
      const unsigned char* p = a.img + (long)a.band[c] * plane + (long)y * Wr;
      const double m = a.mean[c], sd = a.stdv[c];
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = x0 + j0 + e;
        const unsigned char v = (yin && (unsigned)x < (unsigned)Wr) ? p[x] : (unsigned char)0;
        o[e] = norm_byte(a.mode, v, m, sd);
      }
      *reinterpret_cast<float4*>(a.out + ((long)b * a.Cout + c) * S * S + (long)i * S + j0) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// one thread per kept pixel; logits NCHW fp32, so a wave reads 256 contiguous bytes per class plane
// tiles == null: out is the per-tile (B, 2 | C, K, K) result.  Otherwise out is the whole (2 | C, Hr, Wr) output raster
// and tile b = {x0, y0, wx0, wx1, wy0, wy1}: its pixel (i, j) lands at (y0 + i, x0 + j) if inside the write window
// (the part of its margin-cropped centre that no later tile of the slicing job overwrites).
__global__ __launch_bounds__(256) void detect_convert_kernel(const float* __restrict__ logits, int B, int C, int S, int margin,
                                                             int mode, void* __restrict__ out, const int* __restrict__ tiles,
                                                             int Hr, int Wr) {
  const int K = S - 2 * margin;
  const long KK = (long)K * K, SS = (long)S * S, total = KK * B;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int b = (int)(t / KK);
    const long r = t - (long)b * KK;
    const int i = (int)(r / K), j = (int)(r - (long)i * K);
    const float* p = logits + (long)b * C * SS + (long)(i + margin) * S + (j + margin);
    long obase = (long)b * (mode == 0 ? 2 : C) * KK + r, ostride = KK;   // mode 2 ('probs'): C fp32 planes
    if (tiles) {
      const int* tb = tiles + b * 6;
      const int gx = tb[0] + margin + j, gy = tb[1] + margin + i;
      if (gx < tb[2] || gx >= tb[3] || gy < tb[4] || gy >= tb[5]) continue;
      if ((unsigned)gx >= (unsigned)Wr || (unsigned)gy >= (unsigned)Hr) continue;  // never write outside the raster
      obase = (long)gy * Wr + gx;
      ostride = (long)Hr * Wr;
    }
    float x[MAXC];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { x[c] = p[(long)c * SS]; m = fmaxf(m, x[c]); }
    float ssum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { x[c] = expf(x[c] - m); ssum += x[c]; }
    if (mode == 0) {  // convert('argmax'): [first argmax as float32, max probability]
      int best = 0;
      float pbest = -1.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) {
          const float q = x[c] / ssum;
          if (q > pbest) { pbest = q; best = c; }
        }
      float* o = reinterpret_cast<float*>(out) + obase;
      o[0] = (float)best;
      o[ostride] = pbest;
    } else if (mode == 2) {  // no convert: the fp32 probabilities torch.softmax(logits, dim=1) of compare.py:35
      float* o = reinterpret_cast<float*>(out) + obase;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) o[(long)c * ostride] = x[c] / ssum;
    } else {  // convert('class_prob'): (p * 255).astype(uint8) — truncation
      unsigned char* o = reinterpret_cast<unsigned char*>(out) + obase;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) o[(long)c * ostride] = (unsigned char)((x[c] / ssum) * 255.f);
    }
  }
}

// convert('argmax') from per-window (class, probability) maps that the head convolution left (flair_unet_want_preds): the same
// ownership windows as detect_convert_kernel, band 0 = class as float32, band 1 = its softmax probability
__global__ __launch_bounds__(256) void detect_stitch_preds_kernel(const unsigned char* __restrict__ preds, const float* __restrict__ maxprob,
                                                                  int B, int S, int margin, const int* __restrict__ tiles,
                                                                  float* __restrict__ out, int Hr, int Wr) {
  const int K = S - 2 * margin;
  const long KK = (long)K * K, SS = (long)S * S, total = KK * B;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int b = (int)(t / KK);
    const long r = t - (long)b * KK;
    const int i = (int)(r / K), j = (int)(r - (long)i * K);
    const int* tb = tiles + b * 6;
    const int gx = tb[0] + margin + j, gy = tb[1] + margin + i;
    if (gx < tb[2] || gx >= tb[3] || gy < tb[4] || gy >= tb[5]) continue;
    if ((unsigned)gx >= (unsigned)Wr || (unsigned)gy >= (unsigned)Hr) continue;
    const long src = (long)b * SS + (long)(i + margin) * S + (j + margin);
    const long dst = (long)gy * Wr + gx;
    out[dst] = (float)preds[src];
    out[(long)Hr * Wr + dst] = maxprob[src];
  }
}

__global__ __launch_bounds__(256) void confmat_masks_kernel(const unsigned char* __restrict__ truth, const unsigned char* __restrict__ pred,
                                                            long n, int C, int truth_offset, long long* __restrict__ confmat) {
  __shared__ unsigned int hist[MAXC * MAXC];
  for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
  __syncthreads();
  const long n16 = n / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    const uint4 tv = reinterpret_cast<const uint4*>(truth)[i], pv = reinterpret_cast<const uint4*>(pred)[i];
    const unsigned int tw[4] = {tv.x, tv.y, tv.z, tv.w}, pw[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      // numpy: uint8 raster - 1 wraps 0 -> 255; sklearn drops pairs with a member outside range(C)
      const unsigned int t = ((tw[e >> 2] >> (8 * (e & 3))) + (unsigned)truth_offset) & 0xffu;
      const unsigned int p = (pw[e >> 2] >> (8 * (e & 3))) & 0xffu;
      if (t < (unsigned)C && p < (unsigned)C) atomicAdd(&hist[t * C + p], 1u);
    }
  }
  if (blockIdx.x == 0)
    for (long i = n16 * 16 + threadIdx.x; i < n; i += 256) {
      const unsigned int t = ((unsigned)truth[i] + (unsigned)truth_offset) & 0xffu, p = pred[i];
      if (t < (unsigned)C && p < (unsigned)C) atomicAdd(&hist[t * C + p], 1u);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256) {
    const unsigned int v = hist[i];
    if (v) atomicAdd(reinterpret_cast<unsigned long long*>(confmat + i), (unsigned long long)v);
  }
}

inline int stream_blocks(long items) {
  long b = (items + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

int feed_tiles(const FeedArgs& a, hipStream_t s) {
  if (a.B < 1 || a.H < 1 || a.W < 4 || (a.W & 3) || a.Cout < 0 || a.Cout > FeedArgs::MAXCH || a.mode < 0 || a.mode > 2) return -2;
  if (a.out && (!a.img || a.Cout < 1)) return -2;
  if (a.labels && (!a.msk || a.num_classes < 1 || a.num_classes > 255)) return -2;
  if (a.d4 && a.H != a.W) return -2;  // rot90 of a non-square tile changes its shape
  for (int c = 0; c < a.Cout; ++c)
    if (a.band[c] < 0 || a.band[c] >= a.Cb) return -2;
  const long px = (long)a.B * a.H * a.W;
  ProfScope ps("feed_tiles", 0.0, (double)px * ((a.out ? 5.0 * a.Cout : 0.0) + (a.labels ? 2.0 : 0.0)), s);
  hipLaunchKernelGGL(feed_tiles_kernel, dim3(stream_blocks(px / 4)), dim3(256), 0, s, a);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int detect_convert(const float* logits, int B, int C, int S, int margin, int mode, void* out, const int* tiles, int Hr, int Wr,
                   hipStream_t s) {
  if (C < 1 || C > MAXC || B < 1 || margin < 0 || S - 2 * margin < 1 || mode < 0 || mode > 2 || (tiles && mode == 2)) return -2;
  if (tiles && (Hr < 1 || Wr < 1)) return -2;
  const long K = S - 2 * margin;
  ProfScope ps(tiles ? "detect_stitch" : "detect_convert", 0.0, (double)B * K * K * (4.0 * C + (mode == 2 ? 4.0 * C : mode ? C : 8.0)), s);
  hipLaunchKernelGGL(detect_convert_kernel, dim3(stream_blocks((long)B * K * K)), dim3(256), 0, s, logits, B, C, S, margin, mode, out,
                     tiles, Hr, Wr);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int detect_stitch_preds(const unsigned char* preds, const float* maxprob, int B, int S, int margin, const int* tiles, float* out,
                        int Hr, int Wr, hipStream_t s) {
  if (B < 1 || margin < 0 || S - 2 * margin < 1 || Hr < 1 || Wr < 1) return -2;
  const long K = S - 2 * margin;
  ProfScope ps("detect_stitch", 0.0, (double)B * K * K * (5.0 + 8.0), s);
  hipLaunchKernelGGL(detect_stitch_preds_kernel, dim3(stream_blocks((long)B * K * K)), dim3(256), 0, s, preds, maxprob, B, S, margin, tiles,
                     out, Hr, Wr);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int gather_tiles(const FeedArgs& a, const int* tiles, int Hr, int Wr, hipStream_t s) {
  if (a.B < 1 || a.H != a.W || a.W < 4 || (a.W & 3) || a.Cout < 1 || a.Cout > FeedArgs::MAXCH || a.mode < 0 || a.mode > 2) return -2;
  if (!a.img || !a.out || !tiles || Hr < 1 || Wr < 1) return -1;
  for (int c = 0; c < a.Cout; ++c)
    if (a.band[c] < 0 || a.band[c] >= a.Cb) return -2;
  const long px = (long)a.B * a.H * a.W;
  ProfScope ps("gather_tiles", 0.0, (double)px * 5.0 * a.Cout, s);
  hipLaunchKernelGGL(gather_tiles_kernel, dim3(stream_blocks(px / 4)), dim3(256), 0, s, a, tiles, Hr, Wr);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int confmat_masks(const unsigned char* truth, const unsigned char* pred, long n, int C, int truth_offset, long long* confmat,
                  hipStream_t s) {
  if (C < 1 || C > MAXC || n < 0) return -2;
  if (((uintptr_t)truth | (uintptr_t)pred) & 15) return -2;
  if (n == 0) return 0;
  ProfScope ps("confmat_masks", 0.0, 2.0 * (double)n, s);
  hipLaunchKernelGGL(confmat_masks_kernel, dim3(stream_blocks(n / 16)), dim3(256), 0, s, truth, pred, n, C, truth_offset, confmat);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

}  // namespace flair
