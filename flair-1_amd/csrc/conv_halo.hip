// Direct 3x3 / stride-1 / pad-1 convolution for the HBM-bound, small-channel, high-resolution layers of
// the U-Net tail (decoder blocks 3-4, the segmentation head and their data gradients: 16-32 channels at
// 256^2-512^2; SURVEY.md §7 "decoder 32->16 = 96, 16->16 = 72, head 16->13 = 65 FLOP/B").
//
// The implicit-GEMM kernel re-gathers every input pixel 9x through L1/L2, which is what bounds these
// layers.  Here a workgroup owns an 8x32 output tile, stages its (8+2)x(32+2) input halo ONCE into LDS
// (16-byte coalesced loads, ~1.3x the algorithmic read), and feeds all 9 taps to the MFMAs from LDS by
// shifted fragment reads; K is walked tap-major in chunks of CK input channels.  Same operands, epilogue
// (LDS transpose -> 16-byte NHWC stores / fp32 NCHW logits, BN partial sums) and numerics as conv_igemm.
#include "common.h"
#include "prof.h"
#include "tile_store.h"
#include "tile_direct.h"
#include "tune.h"

namespace flair {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct HMma;
template <> struct HMma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct HMma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

constexpr int TH = 8, TW = 32, HH = TH + 2, HW_ = TW + 2, HPIX = HH * HW_;  // 340 halo pixels

template <typename T, int CK, int BN>
struct HaloCfg {
  static constexpr int CH = Elem<T>::CH;
  static constexpr int KF = 4 * CH;                       // K elements consumed per fragment read
  static constexpr int CPP = CK / CH;                     // 16-byte chunks per halo pixel
  // bytes per halo pixel.  64-byte pixels get 96: a ds_read_b128 serves 16 lanes = 16 consecutive pixels, eight of them with the
  // k-chunk of lane group lq and eight with lq ^ 1, and 24-dword strides put the first eight on the multiples of 8 and the others on
  // 8k + 4 — no two on a bank.  (With 80 bytes three lane pairs of every read collided: SQ_LDS_BANK_CONFLICT was half of the LDS
  // cycles of the 32-channel kernels; 32 -> 16 at 512^2 190 -> 180 us in same-box pairs.)
  static constexpr int PSTRIDE = CK * (int)sizeof(T) == 64 ? 96 : CK * (int)sizeof(T) + ((CK * (int)sizeof(T)) >= 64 ? 16 : 0);
  static constexpr int HALO = HPIX * PSTRIDE;
  static constexpr int KW = (9 * CK + KF - 1) / KF * KF;  // weight row length in LDS (zero padded)
  static constexpr int WROW = KW * (int)sizeof(T) + 16;   // bytes, padded against bank conflicts
  static constexpr int WBYTES = BN * WROW;
  static constexpr int CLD = BN * (int)sizeof(T) + 16;
  static constexpr int CT = TH * TW * CLD;
  static constexpr int MAIN = (HALO + WBYTES > CT) ? HALO + WBYTES : CT;
  static constexpr int STATS = 4 * BN * 2 * 4;
  static constexpr int SMEM = MAIN + STATS;
  static constexpr int HITEMS = (HPIX * CPP + 255) / 256;
  static constexpr int WITEMS = (BN * 9 * CPP + 255) / 256;
};

// LZ: instantiate the lazy BN + ReLU input transform (ConvArgs::in_scale); kept out of the plain variant
template <typename T, int CK, int BN, bool LZ>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvArgs a) {
  using Cfg = HaloCfg<T, CK, BN>;
  constexpr int CH = Cfg::CH, KF = Cfg::KF, CPP = Cfg::CPP, TN = BN / 16, TM = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* halo = smem;
  unsigned char* wl = smem + Cfg::HALO;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int H = a.Hout, W = a.Wout;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int tile = blockIdx.x;
  const int n = tile / (tiles_x * tiles_y);
  const int trem = tile - n * tiles_x * tiles_y;
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C0 + a.C1;
  const T* __restrict__ src0 = (const T*)a.src0;
  const T* __restrict__ src1 = (const T*)a.src1;
  const T* __restrict__ wp = (const T*)a.w;
  const int Hs0 = a.up0 ? (H >> 1) : H, Ws0 = a.up0 ? (W >> 1) : W;

  // zero the padded tail of every weight row once (K beyond 9*CK contributes nothing)
  if constexpr (Cfg::KW > 9 * CK) {
    constexpr int PADC = (Cfg::KW - 9 * CK) / CH;
    for (int it = t; it < BN * PADC; it += 256) {
      const int row = it / PADC, pc = it - row * PADC;
      *reinterpret_cast<u32x4*>(wl + row * Cfg::WROW + (9 * CK + pc * CH) * (int)sizeof(T)) = u32x4{0u, 0u, 0u, 0u};
    }
  }

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // The next chunk's halo and weights ride in registers while the current chunk is multiplied (multi-chunk layers:
  // decoder block 3's 128 -> 32 convolution has four; round 1 loaded, waited, staged and only then multiplied).
  u32x4 hreg[Cfg::HITEMS], wreg[Cfg::WITEMS];
  unsigned hbits = 0;
  float lsc[CH], lsh[CH];   // lazy BN + ReLU of the producing unit: a thread always stages the same chunk column (256 % CPP == 0)
  auto load_chunk = [&](int cbase) {
    const bool use0 = cbase < a.C0;
    const T* __restrict__ base = use0 ? src0 : src1;
    const int Hs = use0 ? Hs0 : H, Ws = use0 ? Ws0 : W, Cs = use0 ? a.C0 : a.C1;
    const int sh = (use0 && a.up0) ? 1 : 0;
    const int coff = use0 ? cbase : cbase - a.C0;
    if (LZ && a.in_scale) {
#pragma unroll
      for (int e = 0; e < CH; ++e) {
        lsc[e] = a.in_scale[cbase + (t % CPP) * CH + e];
        lsh[e] = a.in_shift[cbase + (t % CPP) * CH + e];
      }
    }
    unsigned hb2 = 0;
#pragma unroll
    for (int k = 0; k < Cfg::HITEMS; ++k) {
      const int it = t + 256 * k;
      const int hp = it / CPP, ch = it - hp * CPP;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = (it < HPIX * CPP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((n * Hs + (iy >> sh)) * Ws + (ix >> sh)) * Cs + coff + ch * CH) : 0u;
      hreg[k] = *reinterpret_cast<const u32x4*>(base + off);
      hb2 |= (ok ? 1u : 0u) << k;
    }
    hbits = hb2;
#pragma unroll
    for (int k = 0; k < Cfg::WITEMS; ++k) {
      const int it = t + 256 * k;
      const int itc = it < BN * 9 * CPP ? it : 0;
      const int row = itc / (9 * CPP), rem = itc - row * (9 * CPP);
      const int tap = rem / CPP, ch = rem - tap * CPP;
      wreg[k] = *reinterpret_cast<const u32x4*>(wp + (long)(n0 + row) * a.Kpad + tap * Cin + cbase + ch * CH);
    }
  };
  load_chunk(0);
  for (int cbase = 0; cbase < Cin; cbase += CK) {
    if (cbase) __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int k = 0; k < Cfg::HITEMS; ++k) {
      const int it = t + 256 * k;
      if (it < HPIX * CPP) {
        const int hp = it / CPP, ch = it - hp * CPP;
        const u32x4 hv = (LZ && a.in_scale) ? chunk_bn_relu<T>(hreg[k], lsc, lsh) : hreg[k];
        *reinterpret_cast<u32x4*>(halo + hp * Cfg::PSTRIDE + ch * 16) = hv & (0u - ((hbits >> k) & 1u));
      }
    }
#pragma unroll
    for (int k = 0; k < Cfg::WITEMS; ++k) {
      const int it = t + 256 * k;
      if (it < BN * 9 * CPP) {
        const int row = it / (9 * CPP), rem = it - row * (9 * CPP);
        *reinterpret_cast<u32x4*>(wl + row * Cfg::WROW + rem * 16) = wreg[k];
      }
    }
    if (cbase + CK < Cin) {
      load_chunk(cbase + CK);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMA phase
    }
    __syncthreads();
    // ---- 9 taps x CK channels from LDS
#pragma unroll
    for (int j = 0; j < Cfg::KW / KF; ++j) {
      const int k0 = j * KF + lq * CH;
      int tap = k0 / CK;
      const int c = k0 - tap * CK;
      tap = tap > 8 ? 8 : tap;  // padded K: the weights there are zero, any valid address will do
      const int r = tap / 3, s = tap - 3 * r;
      u32x4 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int py = 2 * wave + (i >> 1), px = (i & 1) * 16 + lr;
        af[i] = *reinterpret_cast<const u32x4*>(halo + ((py + r) * HW_ + px + s) * Cfg::PSTRIDE + c * (int)sizeof(T));
      }
#pragma unroll
      for (int q = 0; q < TN; ++q)
        bfr[q] = *reinterpret_cast<const u32x4*>(wl + (q * 16 + lr) * Cfg::WROW + k0 * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < TN; ++q) HMma<T>::run(af[i], bfr[q], acc[i][q]);
    }
  }

  // ------------------------------------------------------------------ epilogue (as conv_igemm)
  __syncthreads();
  unsigned char* ct = smem;
  float* st = reinterpret_cast<float*>(smem + Cfg::MAIN);
  float s1[TN], s2[TN];
#pragma unroll
  for (int q = 0; q < TN; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
#pragma unroll
  for (int q = 0; q < TN; ++q) {
    const int col = q * 16 + lr;
    const bool cin_ok = (n0 + col) < a.Cout;
    const float osc = (a.oscale && cin_ok) ? a.oscale[n0 + col] : 1.f;
    const float bias = ((a.bias && cin_ok) ? a.bias[n0 + col] : 0.f) + ((a.oshift && cin_ok) ? a.oshift[n0 + col] : 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = (2 * wave + (i >> 1)) * TW + (i & 1) * 16 + lq * 4 + rr;  // tile-local pixel
        T v = Elem<T>::from_f(fmaf(acc[i][q][rr], osc, bias));
        const float vf = Elem<T>::to_f(v);
        s1[q] += vf;
        s2[q] += vf * vf;
        *reinterpret_cast<T*>(ct + row * Cfg::CLD + col * (int)sizeof(T)) = v;
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int q = 0; q < TN; ++q) {
      s1[q] += __shfl_xor(s1[q], 16); s1[q] += __shfl_xor(s1[q], 32);
      s2[q] += __shfl_xor(s2[q], 16); s2[q] += __shfl_xor(s2[q], 32);
      if (lq == 0) {
        st[(wave * BN + q * 16 + lr) * 2 + 0] = s1[q];
        st[(wave * BN + q * 16 + lr) * 2 + 1] = s2[q];
      }
    }
  }
  __syncthreads();
  if (a.stats && t < BN && (n0 + t) < a.Cout) {
    float x1 = 0.f, x2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { x1 += st[(w * BN + t) * 2]; x2 += st[(w * BN + t) * 2 + 1]; }
    a.stats[(long)(n0 + t) * gridDim.x + blockIdx.x] = x1;
    a.stats[((long)a.Cout + n0 + t) * gridDim.x + blockIdx.x] = x2;
  }
  store_tile<T, TW, TH * TW, BN, 256, Cfg::CLD>(a, ct, n, y0, x0, n0, t);
  if (a.out_nchw) {
    const long HWp = (long)H * W;
    for (int idx = t; idx < TH * TW * BN; idx += 256) {
      const int nl = idx / (TH * TW), ml = idx - nl * (TH * TW);
      const int py = ml / TW, px = ml - py * TW;
      const int nn = n0 + nl;
      if (nn < a.Cout)
        a.out_nchw[((long)n * a.Cout + nn) * HWp + (long)(y0 + py) * W + x0 + px] =
            Elem<T>::to_f(*reinterpret_cast<const T*>(ct + ml * Cfg::CLD + nl * (int)sizeof(T)));
    }
  }
}

template <typename T, int CK, int BN, bool LZ>
int launch_halo_cfg_l(const ConvArgs& a, hipStream_t s) {
  using Cfg = HaloCfg<T, CK, BN>;
  auto kern = conv3x3_halo_kernel<T, CK, BN, LZ>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long M = (long)a.N * a.Hout * a.Wout;
  dim3 grid((unsigned)(M / (TH * TW)), cdiv(a.Cout, BN));
  {
    const double flops = 2.0 * (double)M * a.Cout * a.Kg;
    const double bytes = ((double)M / (a.up0 ? 4 : 1) * a.C0 + (double)M * a.C1 + (double)M * a.Cout * (a.accumulate ? 2 : 1)) * sizeof(T) +
                         (double)a.Cout * a.Kg * sizeof(T) + (a.out_nchw ? (double)M * a.Cout * 4.0 : 0.0);
    static const char* names[2][2] = {{"conv3x3_halo_f32_ck16", "conv3x3_halo_f32_ck32"}, {"conv3x3_halo_bf16_ck16", "conv3x3_halo_bf16_ck32"}};
    ProfScope ps(names[sizeof(T) == 2][CK == 32], flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::SMEM, s, a);
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Persistent variant for the single-chunk layers (16 or 32 input channels from one source, one column block): the
// kernel above spends its life in phases — load halo + weights, barrier, MFMAs, barrier, store — and the HBM phase of one
// workgroup overlaps the LDS/MFMA phase of its neighbours only by luck.  Here a workgroup keeps the weights in LDS for
// the whole launch, walks many tiles, prefetches the next tile's halo into registers while the current one is
// multiplied, and keeps its BatchNorm partial statistics in registers ([2][Cout][gridDim.x] partials).
template <typename T, int CK, int BN>
struct HaloPCfg {
  using B = HaloCfg<T, CK, BN>;
  static constexpr int WOFF = 0, HOFF = (B::WBYTES + 255) / 256 * 256, SOFF = HOFF + (B::HALO + 255) / 256 * 256;
  static constexpr int SMEM = SOFF + B::STATS;   // no C tile: the epilogue goes from registers to HBM (tile_direct.h)
};

template <typename T, int CK, int BN, bool LZ, bool BNR = false>
__global__ __launch_bounds__(256) void conv3x3_halo_p_kernel(const ConvArgs a, int ntiles) {
  using Cfg = HaloCfg<T, CK, BN>;
  using PC = HaloPCfg<T, CK, BN>;
  constexpr int CH = Cfg::CH, KF = Cfg::KF, CPP = Cfg::CPP, TN = BN / 16, TM = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem + PC::WOFF;
  unsigned char* halo = smem + PC::HOFF;
  float* st = reinterpret_cast<float*>(smem + PC::SOFF);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int H = a.Hout, W = a.Wout;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const T* __restrict__ src0 = (const T*)a.src0;
  const T* __restrict__ wp = (const T*)a.w;
  const int Hs = a.up0 ? (H >> 1) : H, Ws = a.up0 ? (W >> 1) : W;
  const int sh = a.up0 ? 1 : 0;

  // weights: all nine taps of the one chunk, once (+ the zero tail of every row)
  if constexpr (Cfg::KW > 9 * CK) {
    constexpr int PADC = (Cfg::KW - 9 * CK) / CH;
    for (int it = t; it < BN * PADC; it += 256) {
      const int row = it / PADC, pc = it - row * PADC;
      *reinterpret_cast<u32x4*>(wl + row * Cfg::WROW + (9 * CK + pc * CH) * (int)sizeof(T)) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  for (int it = t; it < BN * 9 * CPP; it += 256) {
    const int row = it / (9 * CPP), rem = it - row * (9 * CPP);
    const int tap = rem / CPP, ch = rem - tap * CPP;
    *reinterpret_cast<u32x4*>(wl + row * Cfg::WROW + rem * 16) =
        *reinterpret_cast<const u32x4*>(wp + (long)row * a.Kpad + tap * CK + ch * CH);
  }
  float lsc[CH], lsh[CH];   // lazy BN + ReLU of the producing unit: a thread always stages the same chunk column
  if (LZ && a.in_scale) {
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      lsc[e] = a.in_scale[(t % CPP) * CH + e];
      lsh[e] = a.in_shift[(t % CPP) * CH + e];
    }
  }
  DirectCoef<TN> cf;
  cf.load(a, 0, lq);
  float s1[4 * TN], s2[4 * TN];
#pragma unroll
  for (int e = 0; e < 4 * TN; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  DirectBnr<TN> bn;
  if constexpr (BNR) bn.init(a, 0, lq);

  // Two register slots: the halo of tile i + 2 is requested while tile i is multiplied (the MFMA phase of a 16-32-channel tile
  // is ~0.3 us, an HBM round trip under load ~2 us: with one slot every iteration waited for its loads)
  u32x4 hregA[Cfg::HITEMS], hregB[Cfg::HITEMS];
  unsigned hbitsA = 0, hbitsB = 0;
  auto halo_load = [&](int tile, u32x4 (&hreg)[Cfg::HITEMS], unsigned& hbits) {
    const bool tok = tile < ntiles;
    const int tl = tok ? tile : 0;
    const int n = tl / (tiles_x * tiles_y);
    const int trem = tl - n * tiles_x * tiles_y;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    unsigned hb = 0;
#pragma unroll
    for (int k = 0; k < Cfg::HITEMS; ++k) {
      const int it = t + 256 * k;
      const int hp = it / CPP, ch = it - hp * CPP;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = tok && (it < HPIX * CPP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((n * Hs + (iy >> sh)) * Ws + (ix >> sh)) * CK + ch * CH) : 0u;
      hreg[k] = *reinterpret_cast<const u32x4*>(src0 + off);
      hb |= (ok ? 1u : 0u) << k;
    }
    hbits = hb;
  };
  auto halo_store = [&](const u32x4 (&hreg)[Cfg::HITEMS], unsigned hbits) {
#pragma unroll
    for (int k = 0; k < Cfg::HITEMS; ++k) {
      const int it = t + 256 * k;
      if (it < HPIX * CPP) {
        const int hp = it / CPP, ch = it - hp * CPP;
        const u32x4 hv = (LZ && a.in_scale) ? chunk_bn_relu<T>(hreg[k], lsc, lsh) : hreg[k];
        *reinterpret_cast<u32x4*>(halo + hp * Cfg::PSTRIDE + ch * 16) = hv & (0u - ((hbits >> k) & 1u));
      }
    }
  };

  halo_load(blockIdx.x, hregA, hbitsA);
  halo_load(blockIdx.x + gridDim.x, hregB, hbitsB);
  halo_store(hregA, hbitsA);
  __syncthreads();
  // one tile: request tile + 2 strides into the slot that is free (`ld`), multiply, store from registers, stage the next
  // tile from the other slot (`stg`)
  auto do_tile = [&](int tile, u32x4 (&ld)[Cfg::HITEMS], unsigned& ldbits, const u32x4 (&stg)[Cfg::HITEMS], const unsigned& stbits) {
    halo_load(tile + 2 * gridDim.x, ld, ldbits);
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto rdk = [&](u32x4 (&af)[TM], u32x4 (&bfr)[TN], int j) {
      const int k0 = j * KF + lq * CH;
      int tap = k0 / CK;
      const int c = k0 - tap * CK;
      tap = tap > 8 ? 8 : tap;  // padded K: the weights there are zero, any valid address will do
      const int r = tap / 3, s = tap - 3 * r;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int py = 2 * wave + (i >> 1), px = (i & 1) * 16 + lr;
        af[i] = *reinterpret_cast<const u32x4*>(halo + ((py + r) * HW_ + px + s) * Cfg::PSTRIDE + c * (int)sizeof(T));
      }
#pragma unroll
      for (int q = 0; q < TN; ++q)
        bfr[q] = *reinterpret_cast<const u32x4*>(wl + direct_row_channel<TN>(q, lr) * Cfg::WROW + k0 * (int)sizeof(T));
    };
    auto mmk = [&](const u32x4 (&af)[TM], const u32x4 (&bfr)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < TN; ++q) HMma<T>::run(bfr[q], af[i], acc[i][q]);   // weights as A: D = [cout][pixel]
    };
    constexpr int NK = Cfg::KW / KF;
    if constexpr (CK == 16) {
      // 16-channel layers: fragment reads one K step ahead of the MFMAs in two register sets pinned by scheduling fences (hipcc's
      // own order read a step's five fragments, drained lgkmcnt, then issued its four MFMAs — five exposed LDS round trips per
      // tile): 16 -> 16 at 512^2 144 -> 122 us, step -0.07 ms in same-box pairs of two builds; the 32-channel variants (nine
      // steps, more registers) lost 3 % and keep the plain loop
      u32x4 afA[TM], bfA[TN], afB[TM], bfB[TN];
      rdk(afA, bfA, 0);
#pragma unroll
      for (int j = 0; j < NK; j += 2) {
        if (j + 1 < NK) rdk(afB, bfB, j + 1);
        __builtin_amdgcn_sched_barrier(0);
        mmk(afA, bfA);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < NK) {
          if (j + 2 < NK) rdk(afA, bfA, j + 2);
          __builtin_amdgcn_sched_barrier(0);
          mmk(afB, bfB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        u32x4 af[TM], bfr[TN];
        rdk(af, bfr, j);
        mmk(af, bfr);
      }
    }
    {
      const int n = tile / (tiles_x * tiles_y);
      const int trem = tile - n * tiles_x * tiles_y;
      const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
      direct_store<T, TW, TN, !BNR, BNR>(a, acc, n, y0, x0, 0, wave, lane, cf, s1, s2, &bn);   // (a data gradient has no affine / statistics)
    }
    __syncthreads();   // every wave is done with the halo
    halo_store(stg, stbits);
    __syncthreads();   // next halo visible
  };
  for (int tile = blockIdx.x; tile < ntiles; tile += 2 * gridDim.x) {
    do_tile(tile, hregA, hbitsA, hregB, hbitsB);   // slot A went to LDS before this tile; B holds the next one
    if (tile + gridDim.x < ntiles) do_tile(tile + gridDim.x, hregB, hbitsB, hregA, hbitsA);
  }
  if constexpr (BNR) {   // one partial per workgroup and channel, like the statistics: [2][bnr_C][gridDim.x]
    direct_stats_wave<TN>(bn.r1, bn.r2, st, BN, 0, wave, lane);
    __syncthreads();
    if (t < BN && t < a.bnr_C) {
      float x1 = 0.f, x2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { x1 += st[(w * BN + t) * 2]; x2 += st[(w * BN + t) * 2 + 1]; }
      a.bnr_partial[(long)t * gridDim.x + blockIdx.x] = x1;
      a.bnr_partial[((long)a.bnr_C + t) * gridDim.x + blockIdx.x] = x2;
    }
    return;
  }
  if (a.stats) {
    direct_stats_wave<TN>(s1, s2, st, BN, 0, wave, lane);
    __syncthreads();
    if (t < BN && t < a.Cout) {
      float x1 = 0.f, x2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { x1 += st[(w * BN + t) * 2]; x2 += st[(w * BN + t) * 2 + 1]; }
      a.stats[(long)t * gridDim.x + blockIdx.x] = x1;
      a.stats[((long)a.Cout + t) * gridDim.x + blockIdx.x] = x2;
    }
  }
}

bool halo_persistent(const ConvArgs& a);
// the fused BatchNorm-backward reduction of the persistent single-block kernel: a plain data gradient (no affine, statistics,
// residual, NCHW copy), mask from y, and an output row that IS the unit's channel row
static bool conv_halo_bnr_ok(const ConvArgs& a) {
  return tune("FLAIR_BNR_HALO", 1) && halo_persistent(a) && !a.in_scale && !a.stats && !a.oscale && !a.oshift && !a.bias && !a.ores && !a.orelu &&
         !a.out_nchw && !a.bnr_out && a.bnr_y && a.bnr_C == a.out_ld && a.bnr_C <= 32 && a.bnr_C <= (a.Cout <= 16 ? 16 : 32) &&
         (a.pool_c0 == 0 || a.pool_c0 == a.Cout);
}

// persistent grid = the workgroups that are resident at once (registers and LDS: asked from the runtime once per variant; a
// grid above that runs its surplus as a second, unbalanced round — 16 -> 16 at 512^2: 141 us with 4 per CU against 125 with the
// 3 that fit).  Without a device (planning on a CPU-only host) the LDS bound stands in.
template <typename T, int CK, int BN, bool LZ, bool BNR = false>
int halo_p_per_cu() {
  static const int per_cu = [] {
    int lds = (160 * 1024) / HaloPCfg<T, CK, BN>::SMEM;
    lds = lds > 4 ? 4 : (lds < 1 ? 1 : lds);
    int nb = 0;
    auto kern = conv3x3_halo_p_kernel<T, CK, BN, LZ, BNR>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, HaloPCfg<T, CK, BN>::SMEM) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, HaloPCfg<T, CK, BN>::SMEM) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      return lds;
    }
    return nb > 8 ? 8 : nb;
  }();
  const int cap = tune("FLAIR_HALO_P_WGS", 0);
  return cap > 0 && cap < per_cu ? cap : per_cu;
}

template <typename T, int CK, int BN>
int halo_p_blocks(const ConvArgs& a) {
  const long ntiles = (long)a.N * a.Hout * a.Wout / (TH * TW);
  const long cap = 256L * (a.bnr_partial ? halo_p_per_cu<T, CK, BN, false, true>()
                                         : a.in_scale ? halo_p_per_cu<T, CK, BN, true>() : halo_p_per_cu<T, CK, BN, false>());
  return (int)(ntiles < cap ? ntiles : cap);
}

template <typename T, int CK, int BN, bool LZ, bool BNR = false>
int launch_halo_p_l(const ConvArgs& a, hipStream_t s) {
  using PC = HaloPCfg<T, CK, BN>;
  auto kern = conv3x3_halo_p_kernel<T, CK, BN, LZ, BNR>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PC::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long M = (long)a.N * a.Hout * a.Wout;
  {
    const double flops = 2.0 * (double)M * a.Cout * a.Kg;
    const double bytes = ((double)M / (a.up0 ? 4 : 1) * a.C0 + (double)M * a.Cout * (a.accumulate ? 2 : 1)) * sizeof(T) +
                         (double)a.Cout * a.Kg * sizeof(T) + (a.out_nchw ? (double)M * a.Cout * 4.0 : 0.0);
    static const char* names[2][2] = {{"conv3x3_halo_f32_ck16", "conv3x3_halo_f32_ck32"}, {"conv3x3_halo_bf16_ck16", "conv3x3_halo_bf16_ck32"}};
    ProfScope ps(names[sizeof(T) == 2][CK == 32], flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(halo_p_blocks<T, CK, BN>(a)), dim3(256), PC::SMEM, s, a, (int)(M / (TH * TW)));
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

template <typename T, int CK, int BN>
int launch_halo_p(const ConvArgs& a, hipStream_t s) {
  if (a.bnr_partial) return conv_halo_bnr_ok(a) ? launch_halo_p_l<T, CK, BN, false, true>(a, s) : -6;
  return a.in_scale ? launch_halo_p_l<T, CK, BN, true>(a, s) : launch_halo_p_l<T, CK, BN, false>(a, s);
}


// ------------------------------------------------------------------------------------------------------------------
// Persistent variant with SEVERAL column blocks per workgroup (32 -> 128 at 256^2: the data gradient of decoder block 3's
// first convolution, the slowest launch of the step when four 32-wide column blocks each staged the same halo: 373 us for
// 671 MB).  The weights of all NB blocks stay in LDS (NB x 19 KB), a tile's halo is staged once and multiplied NB times,
// every block's result goes from registers to HBM (tile_direct.h): two barriers per TILE; one workgroup per CU (LDS),
// prefetch distance 2.
template <typename T, int CK, int BN, int NB>
struct HaloPMCfg {
  using B = HaloCfg<T, CK, BN>;
  static constexpr int WOFF = 0, HOFF = (NB * B::WBYTES + 255) / 256 * 256, SOFF = HOFF + (B::HALO + 255) / 256 * 256;
  static constexpr int SMEM = SOFF;
};

// NT = 512: two wave groups share the staged halo, each takes half of the column blocks (two waves per SIMD: with one, every
// LDS read and conversion of the epilogue sat in the MFMA's shadow-less critical path: 349 us)
template <typename T, int CK, int BN, int NB, bool LZ, int NT>
__global__ __launch_bounds__(NT) void conv3x3_halo_pm_kernel(const ConvArgs a, int ntiles) {
  using Cfg = HaloCfg<T, CK, BN>;
  using PC = HaloPMCfg<T, CK, BN, NB>;
  constexpr int CH = Cfg::CH, KF = Cfg::KF, CPP = Cfg::CPP, TN = BN / 16, TM = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem + PC::WOFF;
  unsigned char* halo = smem + PC::HOFF;
  const int t = threadIdx.x, lane = t & 63, wave = (t >> 6) & 3, wgrp = t >> 8;
  constexpr int NG = NT / 256, NBG = NB / NG, HIT = (HPIX * Cfg::CPP + NT - 1) / NT;
  static_assert(NB % NG == 0, "column blocks per wave group");
  const int lr = lane & 15, lq = lane >> 4;
  const int H = a.Hout, W = a.Wout;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const T* __restrict__ src0 = (const T*)a.src0;
  const T* __restrict__ wp = (const T*)a.w;
  const int Hs = a.up0 ? (H >> 1) : H, Ws = a.up0 ? (W >> 1) : W;
  const int sh = a.up0 ? 1 : 0;

  if constexpr (Cfg::KW > 9 * CK) {
    constexpr int PADC = (Cfg::KW - 9 * CK) / CH;
    for (int it = t; it < NB * BN * PADC; it += NT) {
      const int row = it / PADC, pc = it - row * PADC;
      *reinterpret_cast<u32x4*>(wl + row * Cfg::WROW + (9 * CK + pc * CH) * (int)sizeof(T)) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  for (int it = t; it < NB * BN * 9 * CPP; it += NT) {
    const int row = it / (9 * CPP), rem = it - row * (9 * CPP);
    const int tap = rem / CPP, ch = rem - tap * CPP;
    *reinterpret_cast<u32x4*>(wl + row * Cfg::WROW + rem * 16) =
        *reinterpret_cast<const u32x4*>(wp + (long)row * a.Kpad + tap * CK + ch * CH);
  }
  float lsc[CH], lsh[CH];
  if (LZ && a.in_scale) {
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      lsc[e] = a.in_scale[(t % CPP) * CH + e];
      lsh[e] = a.in_shift[(t % CPP) * CH + e];
    }
  }
  DirectCoef<TN> cf;              // unused: plain results only (no per-channel affine, no statistics: halo_persistent_multi)
  float s1[4 * TN], s2[4 * TN];

  u32x4 hregA[HIT], hregB[HIT];
  unsigned hbitsA = 0, hbitsB = 0;
  auto halo_load = [&](int tile, u32x4 (&hreg)[HIT], unsigned& hbits) {
    const bool tok = tile < ntiles;
    const int tl = tok ? tile : 0;
    const int n = tl / (tiles_x * tiles_y);
    const int trem = tl - n * tiles_x * tiles_y;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    unsigned hb = 0;
#pragma unroll
    for (int k = 0; k < HIT; ++k) {
      const int it = t + NT * k;
      const int hp = it / CPP, ch = it - hp * CPP;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = tok && (it < HPIX * CPP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((n * Hs + (iy >> sh)) * Ws + (ix >> sh)) * CK + ch * CH) : 0u;
      hreg[k] = *reinterpret_cast<const u32x4*>(src0 + off);
      hb |= (ok ? 1u : 0u) << k;
    }
    hbits = hb;
  };
  auto halo_store = [&](const u32x4 (&hreg)[HIT], unsigned hbits) {
#pragma unroll
    for (int k = 0; k < HIT; ++k) {
      const int it = t + NT * k;
      if (it < HPIX * CPP) {
        const int hp = it / CPP, ch = it - hp * CPP;
        const u32x4 hv = (LZ && a.in_scale) ? chunk_bn_relu<T>(hreg[k], lsc, lsh) : hreg[k];
        *reinterpret_cast<u32x4*>(halo + hp * Cfg::PSTRIDE + ch * 16) = hv & (0u - ((hbits >> k) & 1u));
      }
    }
  };

  halo_load(blockIdx.x, hregA, hbitsA);
  halo_load(blockIdx.x + gridDim.x, hregB, hbitsB);
  halo_store(hregA, hbitsA);
  __syncthreads();
  auto do_tile = [&](int tile, u32x4 (&ld)[HIT], unsigned& ldbits, const u32x4 (&stg)[HIT], const unsigned& stbits) {
    halo_load(tile + 2 * gridDim.x, ld, ldbits);
    __builtin_amdgcn_sched_barrier(0);
    const int n = tile / (tiles_x * tiles_y);
    const int trem = tile - n * tiles_x * tiles_y;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
#pragma unroll
    for (int jb = 0; jb < NBG; ++jb) {
      const int nb = wgrp * NBG + jb;
      f32x4_t acc[TM][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < Cfg::KW / KF; ++j) {
        const int k0 = j * KF + lq * CH;
        int tap = k0 / CK;
        const int c = k0 - tap * CK;
        tap = tap > 8 ? 8 : tap;
        const int r = tap / 3, s = tap - 3 * r;
        u32x4 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int py = 2 * wave + (i >> 1), px = (i & 1) * 16 + lr;
          af[i] = *reinterpret_cast<const u32x4*>(halo + ((py + r) * HW_ + px + s) * Cfg::PSTRIDE + c * (int)sizeof(T));
        }
#pragma unroll
        for (int q = 0; q < TN; ++q)
          bfr[q] = *reinterpret_cast<const u32x4*>(wl + (nb * BN + direct_row_channel<TN>(q, lr)) * Cfg::WROW + k0 * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < TN; ++q) HMma<T>::run(bfr[q], af[i], acc[i][q]);
      }
      direct_store<T, TW, TN, false>(a, acc, n, y0, x0, nb * BN, wave, lane, cf, s1, s2);   // registers -> HBM, no barrier
    }
    __syncthreads();   // every wave is done with the halo
    halo_store(stg, stbits);
    __syncthreads();   // next halo visible
  };
  for (int tile = blockIdx.x; tile < ntiles; tile += 2 * gridDim.x) {
    do_tile(tile, hregA, hbitsA, hregB, hbitsB);
    if (tile + gridDim.x < ntiles) do_tile(tile + gridDim.x, hregB, hbitsB, hregA, hbitsA);
  }
}

constexpr int PM_NT = 512;

template <typename T, int CK, int BN, int NB, bool LZ>
int halo_pm_per_cu() {
  static const int per_cu = [] {
    int nb = 0;
    auto kern = conv3x3_halo_pm_kernel<T, CK, BN, NB, LZ, PM_NT>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, HaloPMCfg<T, CK, BN, NB>::SMEM) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, PM_NT, HaloPMCfg<T, CK, BN, NB>::SMEM) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      return 1;
    }
    return nb > 4 ? 4 : nb;
  }();
  return per_cu;
}

template <typename T, int CK, int BN, int NB>
int halo_pm_blocks(const ConvArgs& a) {
  const long ntiles = (long)a.N * a.Hout * a.Wout / (TH * TW);
  const long cap = 256L * (a.in_scale ? halo_pm_per_cu<T, CK, BN, NB, true>() : halo_pm_per_cu<T, CK, BN, NB, false>());
  return (int)(ntiles < cap ? ntiles : cap);
}

template <typename T, int CK, int BN, int NB, bool LZ>
int launch_halo_pm_l(const ConvArgs& a, hipStream_t s) {
  using PC = HaloPMCfg<T, CK, BN, NB>;
  auto kern = conv3x3_halo_pm_kernel<T, CK, BN, NB, LZ, PM_NT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, PC::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long M = (long)a.N * a.Hout * a.Wout;
  {
    const double flops = 2.0 * (double)M * a.Cout * a.Kg;
    const double bytes = ((double)M / (a.up0 ? 4 : 1) * a.C0 + (double)M * a.Cout * (a.accumulate ? 2 : 1)) * sizeof(T) + (double)a.Cout * a.Kg * sizeof(T);
    ProfScope ps(CK == 32 ? "conv3x3_halo_bf16_ck32" : "conv3x3_halo_bf16_ck16", flops, bytes, s);
    hipLaunchKernelGGL(kern, dim3(halo_pm_blocks<T, CK, BN, NB>(a)), dim3(PM_NT), PC::SMEM, s, a, (int)(M / (TH * TW)));
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

// bf16, 32 input channels from one source, 128 output channels (four 32-wide blocks), no fp32 NCHW copy
bool halo_persistent_multi(int dtype, const ConvArgs& a) {
  return dtype != DT_F32 && tune("FLAIR_HALO_PM", 1) && a.C1 == 0 && a.C0 == 32 && a.Cout == 128 && !a.out_nchw && (a.pool_c0 % 32) == 0 &&
         !a.stats && !a.oscale && !a.oshift && !a.bias;
}

// single chunk from one source, one column block
bool halo_persistent(const ConvArgs& a) {
  const int on = tune("FLAIR_HALO_PERSIST", 1);   // 2: the 32-channel layers only (round 1 .. mid round 2, when the 16-channel
  const int Cin = a.C0 + a.C1;                    // layers did better with many short-lived workgroups: prefetch distance 1)
  return on && a.C1 == 0 && (Cin == 32 || (Cin == 16 && on != 2)) && a.Cout <= 32;
}

template <typename T, int CK, int BN>
int launch_halo_cfg(const ConvArgs& a, hipStream_t s) {
  return a.in_scale ? launch_halo_cfg_l<T, CK, BN, true>(a, s) : launch_halo_cfg_l<T, CK, BN, false>(a, s);
}

}  // namespace

// The halo kernel serves 3x3 / stride 1 / pad 1 convolutions with <= 32 output channels per workgroup and an
// input whose channel count is 16 or a multiple of 32 (sources not straddled by a chunk), on 8x32-tileable images.
bool conv_halo_applicable(const ConvArgs& a) {
  const int Cin = a.C0 + a.C1;
  if (a.R != 3 || a.S != 3 || a.out_mul != 1 || a.in_div != 1 || a.pad != 1) return false;
  if (a.Hout != a.Hin || a.Wout != a.Win || (a.Hout % TH) || (a.Wout % TW)) return false;
  if (Cin == 16) return a.C1 == 0 && a.Cout <= 32;
  if ((Cin % 32) || (a.C0 % 32)) return false;
  return Cin <= 128 && a.Cout <= 128 && (a.Cout <= 32 || Cin <= 32);
}

template <typename T>
static int halo_rows_t(const ConvArgs& a) {
  const int Cin = a.C0 + a.C1;
  const bool n16 = a.Cout <= 16;
  if (Cin == 16) return n16 ? halo_p_blocks<T, 16, 16>(a) : halo_p_blocks<T, 16, 32>(a);
  return n16 ? halo_p_blocks<T, 32, 16>(a) : halo_p_blocks<T, 32, 32>(a);
}

bool conv_halo_bnr_applicable(const ConvArgs& a) { return conv_halo_applicable(a) && conv_halo_bnr_ok(a); }

// fused argmax (ConvArgs::preds_u8): the persistent single-block kernel whose column block holds every class
bool conv_halo_preds_ok(int dtype, const ConvArgs& a) {
  (void)dtype;
  return tune("FLAIR_HEAD_ARGMAX", 1) && conv_halo_applicable(a) && halo_persistent(a) && !a.pool_c0 && !a.bnr_partial && a.Cout <= 32;
}

int conv_halo_grid_rows(int dtype, const ConvArgs& a) {
  if (halo_persistent_multi(dtype, a)) return halo_pm_blocks<bf16_t, 32, 32, 4>(a);
  if (halo_persistent(a)) return dtype == DT_F32 ? halo_rows_t<float>(a) : halo_rows_t<bf16_t>(a);
  return (int)((long)a.N * a.Hout * a.Wout / (TH * TW));
}

int launch_conv_halo(int dtype, const ConvArgs& a, hipStream_t s) {
  const int Cin = a.C0 + a.C1;
  const bool n16 = a.Cout <= 16;
  if (a.preds_u8 && !conv_halo_preds_ok(dtype, a)) return -6;
  if (halo_persistent_multi(dtype, a))
    return a.in_scale ? launch_halo_pm_l<bf16_t, 32, 32, 4, true>(a, s) : launch_halo_pm_l<bf16_t, 32, 32, 4, false>(a, s);
  if (halo_persistent(a)) {
    if (dtype == DT_F32) {
      if (Cin == 16) return n16 ? launch_halo_p<float, 16, 16>(a, s) : launch_halo_p<float, 16, 32>(a, s);
      return n16 ? launch_halo_p<float, 32, 16>(a, s) : launch_halo_p<float, 32, 32>(a, s);
    }
    if (Cin == 16) return n16 ? launch_halo_p<bf16_t, 16, 16>(a, s) : launch_halo_p<bf16_t, 16, 32>(a, s);
    return n16 ? launch_halo_p<bf16_t, 32, 16>(a, s) : launch_halo_p<bf16_t, 32, 32>(a, s);
  }
  if (dtype == DT_F32) {
    if (Cin == 16) return n16 ? launch_halo_cfg<float, 16, 16>(a, s) : launch_halo_cfg<float, 16, 32>(a, s);
    return n16 ? launch_halo_cfg<float, 32, 16>(a, s) : launch_halo_cfg<float, 32, 32>(a, s);
  }
  if (Cin == 16) return n16 ? launch_halo_cfg<bf16_t, 16, 16>(a, s) : launch_halo_cfg<bf16_t, 16, 32>(a, s);
  return n16 ? launch_halo_cfg<bf16_t, 32, 16>(a, s) : launch_halo_cfg<bf16_t, 32, 32>(a, s);
}

}  // namespace flair
