// Shared epilogue of the halo-tile convolution kernels: C tile in LDS -> HBM with 16-byte stores.
// Besides the plain NHWC store (+ fused residual / ReLU for inference, + read-modify-write accumulate) it
// implements the backward of the decoder's fused "nearest x2 upsample + concat" directly in the data-gradient
// epilogue: output columns [0, pool_c0) belong to the upsampled source and are 2x2 sum-pooled into its
// half-resolution gradient, columns >= pool_c0 go to the skip tensor's gradient — the concatenated gradient
// is never written to HBM (replaces the separate upcat_bwd pass for tile-able shapes).
#pragma once
#include "common.h"

namespace flair {

template <typename T, int TW, int TPIX, int BN, int NT, int CLD>
__device__ __forceinline__ void store_tile(const ConvArgs& a, const unsigned char* ct, int n, int y0, int x0, int n0, int t) {
  constexpr int CH = Elem<T>::CH;
  constexpr int CPR = BN / CH;
  const int H = a.Hout, W = a.Wout;
  if (a.pool_c0 > 0 && n0 < a.pool_c0) {
    // ---- 2x2 sum-pool into the half-resolution gradient of the upsampled source
    T* __restrict__ out = (T*)a.out;
    const int Hh = H >> 1, Wh = W >> 1;
    constexpr int PW = TW / 2;
    for (int idx = t; idx < (TPIX / 4) * CPR; idx += NT) {
      const int prow = idx / CPR, ch = idx - prow * CPR;
      const int py2 = prow / PW, px2 = prow - py2 * PW;
      const int nn = n0 + ch * CH;
      if (nn < a.pool_c0) {
        float fa[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) fa[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (2 * py2 + (q >> 1)) * TW + 2 * px2 + (q & 1);
          float fb[CH];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>(ct + row * CLD + ch * 16), fb);
#pragma unroll
          for (int e = 0; e < CH; ++e) fa[e] += fb[e];
        }
        T* dst = out + ((long)(n * Hh + (y0 >> 1) + py2) * Wh + (x0 >> 1) + px2) * a.out_ld + nn;
        if (a.accumulate) {
          float fb[CH];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>(dst), fb);
#pragma unroll
          for (int e = 0; e < CH; ++e) fa[e] += fb[e];
        }
        *reinterpret_cast<uint4*>(dst) = f_to_chunk<T>(fa);
      }
    }
    return;
  }
  T* __restrict__ out = (T*)(a.pool_c0 > 0 ? a.out_skip : a.out);
  if (!out) return;
  const int ld = a.pool_c0 > 0 ? a.out_skip_ld : a.out_ld;
  const int cbase = a.pool_c0 > 0 ? a.pool_c0 : 0;
  const int acc = a.pool_c0 > 0 ? a.skip_accumulate : a.accumulate;
  for (int idx = t; idx < TPIX * CPR; idx += NT) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const int py = row / TW, px = row - py * TW;
    const int nn = n0 + ch * CH;
    if (nn < a.Cout) {
      uint4 v = *reinterpret_cast<const uint4*>(ct + row * CLD + ch * 16);
      const long goff = ((long)(n * H + y0 + py) * W + x0 + px) * ld + (nn - cbase);
      T* dst = out + goff;
      if (a.ores || a.orelu) {
        float fa[CH];
        chunk_to_f<T>(v, fa);
        if (a.ores) {
          float fb[CH];
          chunk_to_f<T>(*reinterpret_cast<const uint4*>((const T*)a.ores + goff), fb);
#pragma unroll
          for (int e = 0; e < CH; ++e) fa[e] += fb[e];
        }
        if (a.orelu) {
#pragma unroll
          for (int e = 0; e < CH; ++e) fa[e] = fmaxf(fa[e], 0.f);
        }
        v = f_to_chunk<T>(fa);
      }
      if (acc) {
        float fa[CH], fb[CH];
        chunk_to_f<T>(v, fa);
        chunk_to_f<T>(*reinterpret_cast<const uint4*>(dst), fb);
#pragma unroll
        for (int e = 0; e < CH; ++e) fa[e] += fb[e];
        v = f_to_chunk<T>(fa);
      }
      *reinterpret_cast<uint4*>(dst) = v;
    }
  }
}

}  // namespace flair
