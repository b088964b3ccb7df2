// Shared epilogue of the halo-tile convolution kernels: C tile in LDS -> HBM with 16-byte stores.
// Besides the plain NHWC store (+ fused residual / ReLU for inference, + read-modify-write accumulate) it
// implements the backward of the decoder's fused "nearest x2 upsample + concat" directly in the data-gradient
// epilogue: output columns [0, pool_c0) belong to the upsampled source and are 2x2 sum-pooled into its
// half-resolution gradient, columns >= pool_c0 go to the skip tensor's gradient — the concatenated gradient
// is never written to HBM (replaces the separate upcat_bwd pass for tile-able shapes).
#pragma once
#include "common.h"

namespace flair {

// Items of the plain store loop: thread t handles (row, chunk) = divmod(t + k*NT, BN/CH), k < ITEMS.
template <typename T, int TPIX, int BN, int NT>
struct TileItems {
  static constexpr int CH = Elem<T>::CH, CPR = BN / CH, ITEMS = (TPIX * CPR + NT - 1) / NT;
};

// All reads of the upstream unit's pre-BN tensor for the fused BatchNorm-backward reduction, issued together before the
// first store.  (Issuing them earlier still, ahead of the LDS staging of the accumulators, costs more in registers
// than it hides in latency: measured 3.3 -> 3.6 ms on the 128-wide halo-GEMM.)
template <typename T, int TW, int TPIX, int BN, int NT>
__device__ __forceinline__ void tile_bnr_prefetch(const ConvArgs& a, int n, int y0, int x0, int n0, int t,
                                                  uint4 (&yv)[TileItems<T, TPIX, BN, NT>::ITEMS],
                                                  uint4 (&ov)[TileItems<T, TPIX, BN, NT>::ITEMS]) {
  using TI = TileItems<T, TPIX, BN, NT>;
  constexpr int CH = TI::CH, CPR = TI::CPR;
  const int H = a.Hout, W = a.Wout;
#pragma unroll
  for (int k = 0; k < TI::ITEMS; ++k) {
    const int idx = t + k * NT;
    const int row = idx / CPR, ch = idx - row * CPR;
    const int py = row / TW, px = row - py * TW;
    const int nn = n0 + ch * CH;
    const bool ok = idx < TPIX * CPR && nn < a.Cout;
    const long goff = ok ? ((long)(n * H + y0 + py) * W + x0 + px) * a.out_ld + nn : 0;
    yv[k] = *reinterpret_cast<const uint4*>((const T*)a.bnr_y + goff);
    if (a.bnr_out) ov[k] = *reinterpret_cast<const uint4*>((const T*)a.bnr_out + goff);
  }
}

// tile_id / ntiles index the per-tile partial sums of the fused BN-backward reduction (default: the launch grid's x)
template <typename T, int TW, int TPIX, int BN, int NT, int CLD, bool BNR = false>
__device__ __forceinline__ void store_tile(const ConvArgs& a, const unsigned char* ct, int n, int y0, int x0, int n0, int t,
                                           int tile_id = -1, int ntiles = 0) {
  if (tile_id < 0) { tile_id = blockIdx.x; ntiles = gridDim.x; }
  constexpr int CH = Elem<T>::CH;
  constexpr int CPR = BN / CH;
  static_assert(NT % CPR == 0 && CPR <= 64, "a thread must keep one chunk column across its items");
  const int H = a.Hout, W = a.Wout;
  // fused BatchNorm-backward reduction (see ConvArgs::bnr_*): thread t always handles chunk column t % CPR
  const bool bnr = BNR && a.bnr_partial != nullptr && (a.pool_c0 == 0 || n0 < a.pool_c0);
  float r1[CH], r2[CH], msc[CH], msh[CH];
  if (bnr) {
    const int nn = n0 + (t % CPR) * CH;
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      r1[e] = 0.f; r2[e] = 0.f;
      msc[e] = nn + e < a.bnr_C ? a.bnr_scale[nn + e] : 0.f;
      msh[e] = nn + e < a.bnr_C ? a.bnr_shift[nn + e] : 0.f;
    }
  }
  // v: the stored (rounded) gradient chunk; y (and, for units with a residual branch, the ReLU output) at the same place
  // returns the chunk to store: v itself, or (bnr_mask) v with the elements the ReLU switched off set to zero
  auto bnr_item = [&](const uint4& v, const uint4& ychunk, const uint4& ochunk) -> uint4 {
    float d[CH], yy[CH], oo[CH];
    chunk_to_f<T>(v, d);
    chunk_to_f<T>(ychunk, yy);
    if (a.bnr_out) chunk_to_f<T>(ochunk, oo);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      const bool on = a.bnr_out ? (oo[e] > 0.f) : (fmaf(yy[e], msc[e], msh[e]) > 0.f);
      const float dm = on ? d[e] : 0.f;
      d[e] = dm;
      r1[e] += dm;
      r2[e] = fmaf(dm, yy[e], r2[e]);
    }
    return a.bnr_mask ? f_to_chunk<T>(d) : v;   // (values of T or zeros: the repack is exact)
  };
  auto bnr_finish = [&]() {   // block-uniform: every thread of the workgroup calls it
    __syncthreads();          // the C tile has been consumed; reuse its LDS
    float* red = reinterpret_cast<float*>(const_cast<unsigned char*>(ct));
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int off = CPR; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < CH; ++e) { r1[e] += __shfl_xor(r1[e], off); r2[e] += __shfl_xor(r2[e], off); }
    }
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        red[((wave * BN) + lane * CH + e) * 2 + 0] = r1[e];
        red[((wave * BN) + lane * CH + e) * 2 + 1] = r2[e];
      }
    }
    __syncthreads();
    if (t < BN && n0 + t < a.bnr_C) {
      float x1 = 0.f, x2 = 0.f;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) { x1 += red[(w * BN + t) * 2]; x2 += red[(w * BN + t) * 2 + 1]; }
      a.bnr_partial[(long)(n0 + t) * ntiles + tile_id] = x1;
      a.bnr_partial[((long)a.bnr_C + n0 + t) * ntiles + tile_id] = x2;
    }
  };
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
        uint4 v = f_to_chunk<T>(fa);
        if (bnr) v = bnr_item(v, *reinterpret_cast<const uint4*>((const T*)a.bnr_y + (dst - out)),
                              a.bnr_out ? *reinterpret_cast<const uint4*>((const T*)a.bnr_out + (dst - out)) : v);
        *reinterpret_cast<uint4*>(dst) = v;
      }
    }
    if (bnr) bnr_finish();
    return;
  }
  T* __restrict__ out = (T*)(a.pool_c0 > 0 ? a.out_skip : a.out);
  if (!out) return;
  const int ld = a.pool_c0 > 0 ? a.out_skip_ld : a.out_ld;
  const int cbase = a.pool_c0 > 0 ? a.pool_c0 : 0;
  const int acc = a.pool_c0 > 0 ? a.skip_accumulate : a.accumulate;
  constexpr int ITEMS = TileItems<T, TPIX, BN, NT>::ITEMS;
  uint4 yv[ITEMS], ov[ITEMS], av[ITEMS];
  if (bnr) tile_bnr_prefetch<T, TW, TPIX, BN, NT>(a, n, y0, x0, n0, t, yv, ov);
  const T* asrc = (a.acc_src && a.pool_c0 == 0) ? (const T*)a.acc_src : out;
  if (acc) {
    // all read-modify-write operands in flight together (inside the store loop the compiler cannot move a load above the
    // previous item's store: eight dependent HBM round trips per thread in the epilogue of every accumulating data gradient)
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int idx = t + k * NT;
      const int row = idx / CPR, ch = idx - row * CPR;
      const int py = row / TW, px = row - py * TW;
      const int nn = n0 + ch * CH;
      const bool ok = idx < TPIX * CPR && nn < a.Cout;
      const long goff = ok ? ((long)(n * H + y0 + py) * W + x0 + px) * ld + (nn - cbase) : 0;
      av[k] = *reinterpret_cast<const uint4*>(asrc + goff);
    }
  }
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int idx = t + k * NT;
    if (idx >= TPIX * CPR) break;
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
        chunk_to_f<T>(av[k], fb);
#pragma unroll
        for (int e = 0; e < CH; ++e) fa[e] += fb[e];
        v = f_to_chunk<T>(fa);
      }
      if (bnr) v = bnr_item(v, yv[k], ov[k]);
      *reinterpret_cast<uint4*>(dst) = v;
    }
  }
  if (bnr) bnr_finish();
}

}  // namespace flair
