// Register epilogue of the persistent small-channel convolution kernels (conv_halo.hip).
//
// Those layers (16-32 channels at 256^2-512^2) are priced in HBM bytes but were bound by instruction issue: the epilogue
// through an LDS C tile cost 16-32 ds_write_b16 + as many conversions per lane and tile, two barriers and a second pass
// (store_tile) — 350-700 VALU instructions per wave and tile against 20-70 MFMAs.  With the weights as the MFMA A operand
// the accumulators come out as [cout][pixel]: lane (lq, lr) holds output channels 4*lq .. 4*lq+3 of pixel lr of a 16-pixel
// row segment, i.e. 8 (bf16) / 16 (fp32) contiguous bytes of the NHWC row, and 16 lanes cover 16 consecutive pixels — the
// tile goes from registers to HBM with one store per lane and 16x16 block, no LDS, no barrier.  The fused variants of
// store_tile are reproduced on the registers with the same rounding sequence:
//   * accumulate / skip_accumulate: read-modify-write of the same 8 / 16 bytes
//   * pool_c0 (backward of "nearest x2 upsample + concat"): the 2x2 window is the lane pair (lr, lr ^ 1) x the accumulator
//     pair (tile rows 2w, 2w + 1 of wave w) — one DPP exchange per value, summed in store_tile's order
//   * oscale / oshift / bias / ores / orelu (inference), fp32 NCHW logits (the head), BatchNorm partial statistics per lane
#pragma once
#include "common.h"

namespace flair {

// NCH = 4 or 8 consecutive channels of one pixel <-> floats
template <typename T, int NCH> struct LaneVec;
template <> struct LaneVec<bf16_t, 4> {
  __device__ static __forceinline__ void store(bf16_t* p, const float* f) {
    uint2 r;
    r.x = (unsigned)f32_to_bf16(f[0]) | ((unsigned)f32_to_bf16(f[1]) << 16);
    r.y = (unsigned)f32_to_bf16(f[2]) | ((unsigned)f32_to_bf16(f[3]) << 16);
    *reinterpret_cast<uint2*>(p) = r;
  }
  __device__ static __forceinline__ void load(const bf16_t* p, float* f) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  }
};
template <> struct LaneVec<bf16_t, 8> {
  __device__ static __forceinline__ void store(bf16_t* p, const float* f) { *reinterpret_cast<uint4*>(p) = f_to_chunk<bf16_t>(f); }
  __device__ static __forceinline__ void load(const bf16_t* p, float* f) { chunk_to_f<bf16_t>(*reinterpret_cast<const uint4*>(p), f); }
};
template <> struct LaneVec<float, 4> {
  __device__ static __forceinline__ void store(float* p, const float* f) { *reinterpret_cast<uint4*>(p) = f_to_chunk<float>(f); }
  __device__ static __forceinline__ void load(const float* p, float* f) { chunk_to_f<float>(*reinterpret_cast<const uint4*>(p), f); }
};
template <> struct LaneVec<float, 8> {
  __device__ static __forceinline__ void store(float* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = f_to_chunk<float>(f);
    *reinterpret_cast<uint4*>(p + 4) = f_to_chunk<float>(f + 4);
  }
  __device__ static __forceinline__ void load(const float* p, float* f) {
    chunk_to_f<float>(*reinterpret_cast<const uint4*>(p), f);
    chunk_to_f<float>(*reinterpret_cast<const uint4*>(p + 4), f + 4);
  }
};

// value of lane ^ 1 (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float lane_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}

// Which output channel MFMA A-row m (= weight row) of 16-row block q stands for, relative to the column block: the rows are
// permuted so that a lane's 4*TN accumulator elements are 4*TN CONSECUTIVE channels (lane (lq, .) holds rows 4*lq .. 4*lq+3 of
// every block): one 16-byte store per lane and pixel, every pixel's row of the block written whole by four lanes.  (With the
// natural order a 32-wide block needed two 8-byte stores 32 bytes apart: half-written sectors, 76 -> 111 us on 32 -> 32.)
template <int TN>
__device__ __forceinline__ constexpr int direct_row_channel(int q, int m) { return (m >> 2) * (4 * TN) + q * 4 + (m & 3); }

// per-lane epilogue constants of one column block: element e = q*4 + rr is channel n0 + 4*TN*lq + e
template <int TN>
struct DirectCoef {
  float osc[4 * TN], obi[4 * TN];
  __device__ __forceinline__ void load(const ConvArgs& a, int n0, int lq) {
#pragma unroll
    for (int e = 0; e < 4 * TN; ++e) {
      const int col = n0 + 4 * TN * lq + e;
      const bool ok = col < a.Cout;
      osc[e] = (a.oscale && ok) ? a.oscale[col] : 1.f;
      obi[e] = ((a.bias && ok) ? a.bias[col] : 0.f) + ((a.oshift && ok) ? a.oshift[col] : 0.f);
    }
  }
};

// One 8x32 tile of one column block (16*TN channels from n0): acc[i][q] = block (tile row 2*wave + (i >> 1), pixels
// (i & 1)*16 .. +15) x (weight rows q*16 .. +15), element [rr] = row 4*lq + rr of pixel lr = channel n0 + 4*TN*lq + q*4 + rr.
// s1 / s2: running per-lane BatchNorm sums of the ROUNDED outputs (statistics of what the next layer reads).
// Fused first pass of the upstream unit's BatchNorm backward (ConvArgs::bnr_*, mask from y): per-lane state of one column block
template <int TN>
struct DirectBnr {
  float msc[4 * TN], msh[4 * TN], r1[4 * TN], r2[4 * TN];
  __device__ __forceinline__ void init(const ConvArgs& a, int n0, int lq) {
#pragma unroll
    for (int e = 0; e < 4 * TN; ++e) {
      const int col = n0 + 4 * TN * lq + e;
      r1[e] = 0.f; r2[e] = 0.f;
      msc[e] = col < a.bnr_C ? a.bnr_scale[col] : 0.f;
      msh[e] = col < a.bnr_C ? a.bnr_shift[col] : 0.f;
    }
  }
  // f: the values just stored (rounded here as store_tile's bnr_item sees them), y: the unit's pre-BN tensor at the same place
  template <typename T>
  __device__ __forceinline__ void item(const float* f, const float* y) {
#pragma unroll
    for (int e = 0; e < 4 * TN; ++e) {
      const float d = Elem<T>::to_f(Elem<T>::from_f(f[e]));
      const float dm = fmaf(y[e], msc[e], msh[e]) > 0.f ? d : 0.f;
      r1[e] += dm;
      r2[e] = fmaf(dm, y[e], r2[e]);
    }
  }
};

// AFF = false: no per-channel affine and no statistics (plain data gradients): cf / s1 / s2 are not touched
// BNR: `bn` accumulates sum(dz*m), sum(dz*m*y) of the stored gradient (the pooled half when pool_c0 > 0)
template <typename T, int TW_, int TN, bool AFF = true, bool BNR = false>
__device__ __forceinline__ void direct_store(const ConvArgs& a, const f32x4_t (&acc)[4][TN], int n, int y0, int x0, int n0, int wave,
                                             int lane, const DirectCoef<TN>& cf, float (&s1)[4 * TN], float (&s2)[4 * TN],
                                             DirectBnr<TN>* bn = nullptr) {
  constexpr int CH = Elem<T>::CH, NCH = 4 * TN;
  typedef LaneVec<T, NCH> LV;
  const int lr = lane & 15, lq = lane >> 4;
  const int H = a.Hout, W = a.Wout;
  const int cout_pad = (a.Cout + CH - 1) / CH * CH;   // store_tile writes whole 16-byte chunks: same column coverage
  const int nn = n0 + NCH * lq;
  float v[4][NCH];   // rounded results as floats
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < TN; ++q)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int e = q * 4 + rr;
        if constexpr (AFF) {
          v[i][e] = Elem<T>::to_f(Elem<T>::from_f(fmaf(acc[i][q][rr], cf.osc[e], cf.obi[e])));
          s1[e] += v[i][e];
          s2[e] = fmaf(v[i][e], v[i][e], s2[e]);
        } else {
          v[i][e] = Elem<T>::to_f(Elem<T>::from_f(acc[i][q][rr]));
        }
      }
  if (a.preds_u8) {
    // argmax over the column block's channels: a lane holds 4*TN consecutive ones of pixel lr, lanes lr + 16*lq the others
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float bv = 0.f; int bi = 0x7fffffff;
#pragma unroll
      for (int e = 0; e < NCH; ++e) {
        const float x = v[i][e];
        const bool take = nn + e < a.Cout && (bi == 0x7fffffff || x > bv || (x != x && bv == bv));
        if (take) { bv = x; bi = nn + e; }
      }
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        // the other lane's candidate wins if it is larger (NaN counts as larger than any number), or equal with a lower index
        const bool o_nan = ov != ov, m_nan = bv != bv;
        const bool take = oi != 0x7fffffff && (bi == 0x7fffffff || (o_nan && !m_nan) || (!m_nan && ov > bv) ||
                                               ((o_nan == m_nan) && (o_nan || ov == bv) && oi < bi));
        if (take) { bv = ov; bi = oi; }
      }
      const long pix = ((long)n * H + y0 + 2 * wave + (i >> 1)) * W + x0 + (i & 1) * 16 + lr;
      if (lq == 0) a.preds_u8[pix] = (unsigned char)bi;
      if (a.maxprob_f32) {   // softmax probability of the winner: exp(0) / sum_c exp(x_c - max)
        float ssum = 0.f;
#pragma unroll
        for (int e = 0; e < NCH; ++e)
          if (nn + e < a.Cout) ssum += __expf(v[i][e] - bv);   // (2e-7 relative per term; the reference tolerance on this band is 1e-6)
        ssum += __shfl_xor(ssum, 16);
        ssum += __shfl_xor(ssum, 32);
        if (lq == 0) a.maxprob_f32[pix] = 1.f / ssum;
      }
    }
  }
  if (a.out_nchw) {
    const long HWp = (long)H * W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long pix = (long)(y0 + 2 * wave + (i >> 1)) * W + x0 + (i & 1) * 16 + lr;
#pragma unroll
      for (int e = 0; e < NCH; ++e)
        if (nn + e < a.Cout) a.out_nchw[((long)n * a.Cout + nn + e) * HWp + pix] = v[i][e];
    }
  }
  if (a.pool_c0 > 0 && n0 < a.pool_c0) {
    // 2x2 sum-pool into the half-resolution gradient of the upsampled source (window order of store_tile:
    // (y, x), (y, x+1), (y+1, x), (y+1, x+1))
    T* __restrict__ out = (T*)a.out;
    const int Hh = H >> 1, Wh = W >> 1;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      float f[NCH];
#pragma unroll
      for (int e = 0; e < NCH; ++e) {
        const float t01 = lane_xor1(v[ii][e]), t11 = lane_xor1(v[ii + 2][e]);
        f[e] = ((v[ii][e] + t01) + v[ii + 2][e]) + t11;
      }
      if (!(lr & 1) && nn < a.pool_c0) {
        T* dst = out + ((long)(n * Hh + (y0 >> 1) + wave) * Wh + (x0 >> 1) + ii * 8 + (lr >> 1)) * a.out_ld + nn;
        if (a.accumulate) {
          float g[NCH];
          LV::load(dst, g);
#pragma unroll
          for (int e = 0; e < NCH; ++e) f[e] += g[e];
        }
        LV::store(dst, f);
        if constexpr (BNR) {
          float yv[NCH];
          LV::load((const T*)a.bnr_y + (dst - out), yv);
          bn->template item<T>(f, yv);
        }
      }
    }
    return;
  }
  T* __restrict__ out = (T*)(a.pool_c0 > 0 ? a.out_skip : a.out);
  if (!out || nn >= cout_pad) return;
  const int ld = a.pool_c0 > 0 ? a.out_skip_ld : a.out_ld;
  const int cbase = a.pool_c0 > 0 ? a.pool_c0 : 0;
  const int accm = a.pool_c0 > 0 ? a.skip_accumulate : a.accumulate;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long goff = ((long)(n * H + y0 + 2 * wave + (i >> 1)) * W + x0 + (i & 1) * 16 + lr) * ld + (nn - cbase);
    T* dst = out + goff;
    float f[NCH];
#pragma unroll
    for (int e = 0; e < NCH; ++e) f[e] = v[i][e];
    if (a.ores || a.orelu) {
      if (a.ores) {
        float g[NCH];
        LV::load((const T*)a.ores + goff, g);
#pragma unroll
        for (int e = 0; e < NCH; ++e) f[e] += g[e];
      }
      if (a.orelu) {
#pragma unroll
        for (int e = 0; e < NCH; ++e) f[e] = fmaxf(f[e], 0.f);
      }
      if (accm) {   // store_tile rounds before it accumulates
#pragma unroll
        for (int e = 0; e < NCH; ++e) f[e] = Elem<T>::to_f(Elem<T>::from_f(f[e]));
      }
    }
    if (accm) {
      float g[NCH];
      LV::load(dst, g);
#pragma unroll
      for (int e = 0; e < NCH; ++e) f[e] += g[e];
    }
    LV::store(dst, f);
    if constexpr (BNR) {
      if (a.pool_c0 == 0) {   // (with pool_c0 > 0 only the pooled half is the upstream unit's gradient)
        float yv[NCH];
        LV::load((const T*)a.bnr_y + goff, yv);
        bn->template item<T>(f, yv);
      }
    }
  }
}

// Per-lane statistics -> st[wave][ncols][2] in LDS (ncols = columns of the workgroup, col0 = first column of this block)
template <int TN>
__device__ __forceinline__ void direct_stats_wave(float (&s1)[4 * TN], float (&s2)[4 * TN], float* st, int ncols, int col0, int wave, int lane) {
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int e = 0; e < 4 * TN; ++e) {
    float x1 = s1[e], x2 = s2[e];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) { x1 += __shfl_xor(x1, off); x2 += __shfl_xor(x2, off); }
    if (lr == 0) {
      st[(wave * ncols + col0 + 4 * TN * lq + e) * 2 + 0] = x1;
      st[(wave * ncols + col0 + 4 * TN * lq + e) * 2 + 1] = x2;
    }
  }
}

}  // namespace flair
