// Weight gradient of the 3x3 / stride-1 / pad-1 layers with many channels (ResNet34 layer2-4, decoder
// blocks 0-1):  dW[k][(r,s)][c] = sum_p dY[p][k] * X[p + (r,s)][c].
//
// The GEMM formulation (wgrad.hip) re-gathers the im2col operand once per tap and re-reads dY once per
// 128-column tile, ~64 FLOP per L1 byte.  Here a persistent workgroup of 8 waves keeps a whole
// [CO output channels] x [9 taps x CK input channels] block of the gradient in registers (144 fp32
// accumulator VGPRs per lane in bf16 mode) and walks 16x8-pixel tiles: per tile it stages the X halo
// (18x10 pixels x 128 B) and the dY tile (128 pixels x 256 B) ONCE in double-buffered LDS; the nine taps
// are shifted transposed reads (ds_read_b64_tr_b16) of the same halo.  Waves are arranged 4 (output-channel
// groups) x 2 (input-channel halves) so that one A fragment feeds 18 MFMAs.  One fp32 slab per workgroup,
// summed in a fixed order by wgrad_reduce_kernel.
#include "common.h"
#include "prof.h"
#include "tune.h"

namespace flair {

void launch_wgrad_reduce(const float* partial, float* dw, int splits, int Cout, int Cout_pad, int Kpad, int Cin,
                         int Cin_real, int R, int S, int accumulate, hipStream_t s);  // wgrad.hip

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TW = 16, TH = 8, HW_ = TW + 2, HH = TH + 2, HPIX = HW_ * HH, TPIX = TW * TH;  // 180 halo / 128 tile pixels
constexpr int NT = 512;   // 8 waves: 4 (output-channel groups) x 2 (input-channel halves) at 128 output channels per workgroup;
                          // 2 x 2 x 2 (tile-row halves, each with its own slab) at 64

template <typename T> struct BFrag;
template <> struct BFrag<bf16_t> {
  static constexpr int KSTEP = 32;
  template <int STRIDE>
  __device__ static __forceinline__ u32x4 load(const unsigned char* img, int row0, int cb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) s16x4_t* lds_p;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + (row0 + 8 * g + q) * STRIDE + cb + 8 * p));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + (row0 + 8 * g + 4 + q) * STRIDE + cb + 8 * p));
    u32x4 r;
    r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
    r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
    r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
    r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
    return r;
  }
  __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct BFrag<float> {
  static constexpr int KSTEP = 16;
  template <int STRIDE>
  __device__ static __forceinline__ u32x4 load(const unsigned char* img, int row0, int cb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    u32x4 r;
    r.x = *reinterpret_cast<const unsigned*>(img + (row0 + g) * STRIDE + cb + 4 * i);
    r.y = *reinterpret_cast<const unsigned*>(img + (row0 + 4 + g) * STRIDE + cb + 4 * i);
    r.z = *reinterpret_cast<const unsigned*>(img + (row0 + 8 + g) * STRIDE + cb + 4 * i);
    r.w = *reinterpret_cast<const unsigned*>(img + (row0 + 12 + g) * STRIDE + cb + 4 * i);
    return r;
  }
  __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

struct WgBigArgs {
  const void* x0; const void* x1; int C0, C1, up0;
  int N, H, W;
  const void* dy; int dy_ld; int Cout;
  float* partial;   // [gridDim.x][Cout][9*Cin] fp32
  int Kg, ntiles;
  const float* in_scale; const float* in_shift;   // lazy BN + ReLU on x0 (WgradArgs::in_scale)
};

// KG = 1: 16*CH output channels per workgroup (256 B of dY per pixel), waves 4 x 2.
// KG = 2:  8*CH output channels per workgroup (128 B per pixel), waves 2 x 2 x 2: the two wave groups take the upper /
//          lower half of every tile's rows and fold their sums through LDS at the end (the 64-output-channel layers).
template <typename T, int KG>
struct WgBigCfg {
  static constexpr int CH = Elem<T>::CH;
  static constexpr int CK = 8 * CH;        // input channels per workgroup: 128 B per pixel
  static constexpr int CO = 16 * CH / KG;  // output channels per workgroup
  static constexpr int DCH = 16 / KG;      // 16-byte chunks of dY per pixel
  static constexpr int XSTRIDE = 128 + 16, DSTRIDE = 256 / KG + 16;
  static constexpr int XBYTES = HPIX * XSTRIDE, DBYTES = TPIX * DSTRIDE;
  static constexpr int STAGE = XBYTES + DBYTES;
  static constexpr int SMEM = 2 * STAGE;
  static constexpr int WK = 4 / KG;                               // output-channel wave groups
  static constexpr int COT = CO / WK / 16, CIT = CK / 2 / 16;     // per-wave 16x16 blocks: bf16 2 x 2, f32 1 x 1
  static constexpr int XITEMS = (HPIX * 8 + NT - 1) / NT;         // 3
  static constexpr int DITEMS = (TPIX * DCH) / NT;                // 4 | 2
};

// LZ: x0 is the PRE-BatchNorm output of the producing unit; relu(x * in_scale[c] + in_shift[c]) is applied while the halo
// is staged (a thread always stages the same 16-byte channel chunk, so its coefficients are loaded once)
template <typename T, int KG, bool LZ>
__global__ __launch_bounds__(512) void wgrad3x3_big_kernel(const WgBigArgs a) {
  using Cfg = WgBigCfg<T, KG>;
  constexpr int CH = Cfg::CH, CK = Cfg::CK, CO = Cfg::CO, COT = Cfg::COT, CIT = Cfg::CIT, DCH = Cfg::DCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // output-channel group, input-channel half (of 2), tile-row half
  const int wk = wave % Cfg::WK, wc = (wave / Cfg::WK) & 1, kg = wave / (2 * Cfg::WK);
  constexpr int YROWS = TH / KG;
  const int H = a.H, W = a.W;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int cbase = blockIdx.y * CK, kbase = blockIdx.z * CO;
  const T* __restrict__ x0 = (const T*)a.x0;
  const T* __restrict__ x1 = (const T*)a.x1;
  const T* __restrict__ dy = (const T*)a.dy;
  const bool use0 = cbase < a.C0;
  const T* __restrict__ xb = use0 ? x0 : x1;
  const int Hs = (use0 && a.up0) ? (H >> 1) : H, Ws = (use0 && a.up0) ? (W >> 1) : W, Cs = use0 ? a.C0 : a.C1;
  const int sh = (use0 && a.up0) ? 1 : 0;
  const int coff = use0 ? cbase : cbase - a.C0;

  f32x4_t acc[COT][CIT][9];
#pragma unroll
  for (int co = 0; co < COT; ++co)
#pragma unroll
    for (int ci = 0; ci < CIT; ++ci)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) acc[co][ci][tp] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4 xr[Cfg::XITEMS], dr[Cfg::DITEMS];
  unsigned xm[Cfg::XITEMS];
  unsigned dmask = 0u;
  const bool lz = LZ && use0 && a.in_scale != nullptr;
  // (the lazy transform's coefficients are re-read from L1 when a halo is staged: held in registers for the whole kernel they
  // were the 16 registers the pipelined multiply phase then spilled)
  const float* __restrict__ lzs = lz ? a.in_scale + coff + (t & 7) * CH : nullptr;
  const float* __restrict__ lzh = lz ? a.in_shift + coff + (t & 7) * CH : nullptr;
  auto load_tile = [&](int tile) {
    const bool tok = tile < a.ntiles;
    const int tl = tok ? tile : 0;
    const int n = tl / (tiles_x * tiles_y);
    const int trem = tl - n * tiles_x * tiles_y;
    const int y0 = (trem / tiles_x) * TH, x0p = (trem % tiles_x) * TW;
#pragma unroll
    for (int k = 0; k < Cfg::XITEMS; ++k) {
      const int it = t + NT * k;
      const int hp = it >> 3, ch = it & 7;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0p - 1 + hx;
      const bool ok = tok && (it < HPIX * 8) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((n * Hs + (iy >> sh)) * Ws + (ix >> sh)) * Cs + coff + ch * CH) : 0u;
      xr[k] = *reinterpret_cast<const u32x4*>(xb + off);
      xm[k] = ok ? 0xffffffffu : 0u;
    }
#pragma unroll
    for (int k = 0; k < Cfg::DITEMS; ++k) {
      const int it = t + NT * k;
      const int px = it / DCH, ch = it % DCH;
      const int py = px / TW, pxx = px - py * TW;
      const unsigned off = (unsigned)(((n * H + y0 + py) * W + x0p + pxx) * a.dy_ld + kbase + ch * CH);
      dr[k] = *reinterpret_cast<const u32x4*>(dy + (tok ? off : 0u));
    }
    dmask = tok ? 0xffffffffu : 0u;   // applied when the registers go to LDS: a select HERE would wait for the loads it masks
  };
  auto store_x = [&](int buf) {
    unsigned char* xh = smem + buf * Cfg::STAGE;
    float lsc[CH], lsh[CH];
    if constexpr (LZ) {
      if (lz) {
#pragma unroll
        for (int e = 0; e < CH; ++e) { lsc[e] = lzs[e]; lsh[e] = lzh[e]; }
      }
    }
#pragma unroll
    for (int k = 0; k < Cfg::XITEMS; ++k) {
      const int it = t + NT * k;
      if (it < HPIX * 8) {
        u32x4 xv = xr[k];
        if constexpr (LZ) {
          if (lz) xv = chunk_bn_relu<T, u32x4>(xv, lsc, lsh);
        }
        *reinterpret_cast<u32x4*>(xh + (it >> 3) * Cfg::XSTRIDE + (it & 7) * 16) = xv & xm[k];
      }
    }
  };
  auto store_d = [&](int buf) {
    unsigned char* dyt = smem + buf * Cfg::STAGE + Cfg::XBYTES;
#pragma unroll
    for (int k = 0; k < Cfg::DITEMS; ++k) {
      const int it = t + NT * k;
      *reinterpret_cast<u32x4*>(dyt + (it / DCH) * Cfg::DSTRIDE + (it % DCH) * 16) = dr[k] & dmask;
    }
  };

  // Staging is folded INTO the multiply phase (round 2b): while tile i is multiplied from LDS buffer i & 1, the registers hold
  // tile i + 1; its halo goes to the other buffer after the first K step, its dY tile after the second, and the loads of tile
  // i + 2 are issued right behind — they have the rest of this tile and the first K step of the next to land.  Before, the
  // ds_writes, the lazy transform and the address arithmetic of the next loads sat between the last MFMA and the barrier:
  // 3.9 us per tile against 2.4 us of MFMA issue (`FLAIR_WG_DBG`-style timing).
  load_tile(blockIdx.x);
  store_x(0); store_d(0);
  load_tile(blockIdx.x + gridDim.x);
  __syncthreads();
  int it = 0;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, ++it) {
    const int nbuf = (it + 1) & 1;
    const int tile2 = tile + 2 * gridDim.x;
    const unsigned char* xh = smem + (it & 1) * Cfg::STAGE;
    const unsigned char* dyt = xh + Cfg::XBYTES;
    if constexpr (sizeof(T) == 2) {
      // bf16: one MFMA K step = 32 pixels = tile rows y, y+1 (16 pixels each).  Lane (g,i) supplies the
      // transposed-read addresses of pixels 8g+q and 8g+4+q of the step; in the halo image the second row
      // starts HW_ pixels later, in the dense dY image 16 pixels later.
      const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
#ifndef FLAIR_WG_KMAP
#define FLAIR_WG_KMAP 1
#endif
      constexpr int pstep = FLAIR_WG_KMAP ? 2 : 1, hi_px = FLAIR_WG_KMAP ? 1 : 4;
      const int dpix = 8 * g + pstep * q;
      const int xpix = (g >> 1) * HW_ + 8 * (g & 1) + pstep * q;
      typedef __attribute__((address_space(3))) s16x4_t* lds_p;
      auto tr2 = [&](const unsigned char* lo_addr, int hi_delta) {
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo_addr));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo_addr + hi_delta));
        u32x4 r;
        r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
        r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
        r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
        r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
        return r;
      };
      // Fragment reads one tap ahead of the MFMAs, in two register sets pinned by scheduling fences (hipcc's own order read a
      // tap's fragments and drained lgkmcnt right in front of its four MFMAs: the LDS latency of every tap was exposed, 3.9 us
      // per tile against 2.4 us of MFMA issue)
      auto rdb = [&](u32x4 (&bf)[CIT], int y, int tp) {
        const int r = tp / 3, s = tp - 3 * r;
#pragma unroll
        for (int ci = 0; ci < CIT; ++ci)
          bf[ci] = tr2(xh + (xpix + (y + r) * HW_ + s) * Cfg::XSTRIDE + (wc * CIT + ci) * 32 + 8 * p, hi_px * Cfg::XSTRIDE);
      };
      auto kstep = [&](int y) {
        u32x4 af[COT], bfA[CIT], bfB[CIT];
#pragma unroll
        for (int co = 0; co < COT; ++co)
          af[co] = tr2(dyt + (y * TW + dpix) * Cfg::DSTRIDE + (wk * COT + co) * 32 + 8 * p, hi_px * Cfg::DSTRIDE);
        rdb(bfA, y, 0);
#pragma unroll
        for (int tp = 0; tp < 9; tp += 2) {
          if (tp + 1 < 9) rdb(bfB, y, tp + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ci = 0; ci < CIT; ++ci)
#pragma unroll
            for (int co = 0; co < COT; ++co) BFrag<T>::mma(af[co], bfA[ci], acc[co][ci][tp]);
          __builtin_amdgcn_sched_barrier(0);
          if (tp + 1 < 9) {
            if (tp + 2 < 9) rdb(bfA, y, tp + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ci = 0; ci < CIT; ++ci)
#pragma unroll
              for (int co = 0; co < COT; ++co) BFrag<T>::mma(af[co], bfB[ci], acc[co][ci][tp + 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      const int yb = kg * YROWS;
      kstep(yb);
      __builtin_amdgcn_sched_barrier(0);
      store_x(nbuf);
      if constexpr (KG == 2) { store_d(nbuf); load_tile(tile2); }
      __builtin_amdgcn_sched_barrier(0);
      kstep(yb + 2);
      if constexpr (KG == 1) {
        __builtin_amdgcn_sched_barrier(0);
        store_d(nbuf);
        load_tile(tile2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int y = yb + 4; y < yb + YROWS; y += 2) kstep(y);
      }
    } else {
#pragma unroll 1
      for (int y = kg * YROWS; y < (kg + 1) * YROWS; ++y) {   // f32: K step = 16 pixels = one tile row
        u32x4 af[COT];
#pragma unroll
        for (int co = 0; co < COT; ++co)
          af[co] = BFrag<T>::template load<Cfg::DSTRIDE>(dyt, y * TW, (wk * COT + co) * 16 * (int)sizeof(T), lane);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int r = tp / 3, s = tp - 3 * r;
#pragma unroll
          for (int ci = 0; ci < CIT; ++ci) {
            const u32x4 bf = BFrag<T>::template load<Cfg::XSTRIDE>(xh, (y + r) * HW_ + s, (wc * CIT + ci) * 16 * (int)sizeof(T), lane);
#pragma unroll
            for (int co = 0; co < COT; ++co) BFrag<T>::mma(af[co], bf, acc[co][ci][tp]);
          }
        }
      }
      store_x(nbuf); store_d(nbuf);
      load_tile(tile2);
    }
    __syncthreads();
  }

  if constexpr (KG == 2) {
    // fold the lower tile-row half's sums into the upper half's through LDS (the stages are free now), one
    // output-channel block at a time: [register][lane] words, conflict-free
    static_assert(COT * CIT * 9 * 4 * 64 * 4 * 4 / COT <= Cfg::SMEM, "reduction scratch exceeds the staging LDS");
    float* red = reinterpret_cast<float*>(smem) + (wave & 3) * (CIT * 9 * 4) * 64;
#pragma unroll
    for (int co = 0; co < COT; ++co) {
      __syncthreads();
      if (kg == 1) {
#pragma unroll
        for (int ci = 0; ci < CIT; ++ci)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[((ci * 9 + tp) * 4 + e) * 64 + lane] = acc[co][ci][tp][e];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int ci = 0; ci < CIT; ++ci)
#pragma unroll
          for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[co][ci][tp][e] += red[((ci * 9 + tp) * 4 + e) * 64 + lane];
      }
    }
    if (kg == 1) return;
  }
  // Every (remaining) wave owns a disjoint block of the gradient.  The slab is stored in the order the accumulators sit
  // in registers — one 16-byte store per lane and accumulator tile, 1 KiB contiguous per wave-instruction (the
  // [cout][9*Cin] image of round 1 cost four 4-byte stores per tile in 64-byte segments: the tail was store-issue bound,
  // ~20 us of a ~100 us kernel).  wgrad_big_reduce_kernel undoes the permutation while it sums the slabs.
  //   slab of workgroup (x, y, z) at ((x * gridDim.y + y) * gridDim.z + z) * (CO * 9 * CK) floats,
  //   element [wave-slot][co][ci][tap][lane][e]
  constexpr int WSLOTS = Cfg::WK * 2;   // waves that hold a block (KG == 2: the upper tile-row half only)
  float* __restrict__ part = a.partial + (((long)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * (long)(CO * 9 * CK);
  const int wslot = wc * Cfg::WK + wk;
  float4* __restrict__ dst = reinterpret_cast<float4*>(part) + (long)wslot * (COT * CIT * 9 * 64) + lane;
  (void)WSLOTS;
#pragma unroll
  for (int co = 0; co < COT; ++co)
#pragma unroll
    for (int ci = 0; ci < CIT; ++ci)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
        dst[((co * CIT + ci) * 9 + tp) * 64] = make_float4(acc[co][ci][tp][0], acc[co][ci][tp][1], acc[co][ci][tp][2], acc[co][ci][tp][3]);
}

// Ordered sum of the nsplit slabs of every (input-channel block y, output-channel block z) + the permutation back to
// PyTorch OIHW.  A workgroup handles 256 / G slab elements (16 bytes = four consecutive output channels of one
// (tap, cin)); its G thread groups each sum the slabs x = g, g + G, ... with two partial sums in flight, then group 0 adds
// the G partials in a fixed order: bit-reproducible.  G = 16 for the many-slab layers (64 -> 64 at 128^2: 256 slabs of a
// 147 KB gradient — with one thread per element that launch had 36 workgroups and ran at 0.5 TB/s), G = 4 otherwise.
template <int CH, int KG, int G>
__global__ __launch_bounds__(256) void wgrad_big_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit,
                                                               int ny, int nz, int Cin_real, int accumulate) {
  constexpr int CK = 8 * CH, CO = 16 * CH / KG, WK = 4 / KG, COT = CO / WK / 16, CIT = CK / 2 / 16;
  constexpr int BLK4 = CO * 9 * CK / 4;   // float4 elements per slab
  constexpr int E = 256 / G;
  __shared__ float4 sh[G][E];
  const long total = (long)ny * nz * BLK4;
  const int e = threadIdx.x % E, g = threadIdx.x / E;
  const long xs = (long)ny * nz * BLK4;   // stride between the slabs of consecutive pixel splits
  for (long base = (long)blockIdx.x * E; base < total; base += (long)gridDim.x * E) {
    const long idx = base + e;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (idx < total) {
      const float4* p = reinterpret_cast<const float4*>(part) + idx;
      int x = g;
      for (; x + G < nsplit; x += 2 * G) {
        const float4 v0 = p[(long)x * xs], v1 = p[(long)(x + G) * xs];
        s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
        s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
      }
      if (x < nsplit) {
        const float4 v0 = p[(long)x * xs];
        s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
      }
    }
    sh[g][e] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (g == 0 && idx < total) {
      float4 r4 = sh[0][e];
#pragma unroll
      for (int q = 1; q < G; ++q) { const float4 v = sh[q][e]; r4.x += v.x; r4.y += v.y; r4.z += v.z; r4.w += v.w; }
      const float r[4] = {r4.x, r4.y, r4.z, r4.w};
      const int el = (int)(idx % BLK4);
      const int yz = (int)(idx / BLK4);
      const int z = yz % nz, y = yz / nz;
      // el = ((wslot * COT + co) * CIT + ci) * 9 * 64 + tap * 64 + lane
      const int lane = el & 63;
      int q = el >> 6;
      const int tp = q % 9; q /= 9;
      const int ci = q % CIT; q /= CIT;
      const int co = q % COT; q /= COT;
      const int wk = q % WK, wc = q / WK;
      const int lr = lane & 15, lq = lane >> 4;
      const int c = y * CK + (wc * CIT + ci) * 16 + lr;
      if (c < Cin_real) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const int k = z * CO + (wk * COT + co) * 16 + lq * 4 + k4;
          float* d = dw + ((long)k * Cin_real + c) * 9 + tp;
          *d = accumulate ? (*d + r[k4]) : r[k4];
        }
      }
    }
    __syncthreads();
  }
}

template <int CH, int KG>
static void launch_big_reduce(const float* part, float* dw, int nsplit, int ny, int nz, long total4, int Cin_real, int accumulate,
                              hipStream_t s) {
  if (nsplit >= 64) {
    int blocks = cdiv(total4, 16);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((wgrad_big_reduce_kernel<CH, KG, 16>), dim3(blocks), dim3(256), 0, s, part, dw, nsplit, ny, nz, Cin_real, accumulate);
  } else {
    int blocks = cdiv(total4, 64);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((wgrad_big_reduce_kernel<CH, KG, 4>), dim3(blocks), dim3(256), 0, s, part, dw, nsplit, ny, nz, Cin_real, accumulate);
  }
}

}  // namespace

// kg = 1: 128 (f32: 64) output channels per workgroup; kg = 2: 64 (32), two slabs per workgroup
static bool wg_big_geom(int dtype, const WgradArgs& a, int& nsplit, int& kg) {
  const int ck = dtype == DT_F32 ? 32 : 64;
  int co = dtype == DT_F32 ? 64 : 128;
  const int Cin = a.C0 + a.C1;
  kg = 1;
  if (a.Cout % co) { co /= 2; kg = 2; }
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1) return false;
  if (a.Hout != a.Hin || a.Wout != a.Win || (a.Hin % TH) || (a.Win % TW)) return false;
  if ((Cin % ck) || (a.C0 % ck) || (a.Cout % co) || a.dy_ld != a.Cout) return false;
  const long ntiles = (long)a.N * a.Hin * a.Win / TPIX;
  const int per = (Cin / ck) * (a.Cout / co);
  long ns = (a.cus > 0 ? a.cus : tune("FLAIR_WG_CUS_OP", 256)) / per;   // one 8-wave workgroup per CU (the standalone operator: all of them)
  if (ns < 1) ns = 1;
  if (ns > ntiles) ns = ntiles;
  nsplit = (int)ns;
  return true;
}

bool wgrad_big_applicable(int dtype, const WgradArgs& a) {
  int ns, kg;
  return wg_big_geom(dtype, a, ns, kg);
}

size_t wgrad_big_workspace_bytes(int dtype, const WgradArgs& a) {
  int ns, kg;
  if (!wg_big_geom(dtype, a, ns, kg)) return 0;
  return (size_t)ns * a.Cout * 9 * (a.C0 + a.C1) * sizeof(float);
}

template <typename T, int KG, bool LZ>
static int launch_big_l(const WgBigArgs& h, int nsplit, int Cin, hipStream_t s) {
  using Cfg = WgBigCfg<T, KG>;
  auto kern = wgrad3x3_big_kernel<T, KG, LZ>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(nsplit, Cin / Cfg::CK, h.Cout / Cfg::CO);
  const double M = (double)h.N * h.H * h.W;
  ProfScope ps(sizeof(T) == 2 ? (KG == 1 ? "wgrad3x3_big_bf16" : "wgrad3x3_big_bf16_co64") : "wgrad3x3_big_f32", 2.0 * M * h.Cout * 9.0 * Cin,
               (M * h.dy_ld + M * (h.C0 / (h.up0 ? 4.0 : 1.0) + h.C1)) * sizeof(T), s);
  hipLaunchKernelGGL(kern, grid, dim3(NT), Cfg::SMEM, s, h);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

template <typename T, int KG>
static int launch_big_t(const WgBigArgs& h, int nsplit, int Cin, hipStream_t s) {
  return h.in_scale ? launch_big_l<T, KG, true>(h, nsplit, Cin, s) : launch_big_l<T, KG, false>(h, nsplit, Cin, s);
}

int launch_wgrad_big(int dtype, const WgradArgs& a, hipStream_t s) {
  int nsplit, kg;
  if (!wg_big_geom(dtype, a, nsplit, kg)) return -2;
  const int Cin = a.C0 + a.C1;
  WgBigArgs h;
  h.x0 = a.x0; h.x1 = a.x1; h.C0 = a.C0; h.C1 = a.C1; h.up0 = a.up0; h.N = a.N; h.H = a.Hin; h.W = a.Win;
  h.dy = a.dy; h.dy_ld = a.dy_ld; h.Cout = a.Cout; h.partial = a.partial; h.Kg = 9 * Cin;
  h.ntiles = (int)((long)a.N * a.Hin * a.Win / TPIX);
  h.in_scale = a.in_scale; h.in_shift = a.in_shift;
  int rc;
  if (dtype == DT_F32) rc = kg == 1 ? launch_big_t<float, 1>(h, nsplit, Cin, s) : launch_big_t<float, 2>(h, nsplit, Cin, s);
  else rc = kg == 1 ? launch_big_t<bf16_t, 1>(h, nsplit, Cin, s) : launch_big_t<bf16_t, 2>(h, nsplit, Cin, s);
  if (rc) return rc;
  {
    const int ch = dtype == DT_F32 ? 4 : 8;
    const int ck = 8 * ch, co = 16 * ch / kg;
    const int ny = Cin / ck, nz = a.Cout / co;
    const long total4 = (long)a.Cout * 9 * Cin / 4;
    ProfScope ps("wgrad_reduce", 0.0, ((double)nsplit + 1.0) * a.Cout * 9.0 * Cin * 4.0, s);
    if (dtype == DT_F32) {
      if (kg == 1) launch_big_reduce<4, 1>(a.partial, a.dw, nsplit, ny, nz, total4, a.Cin_real, a.accumulate, s);
      else launch_big_reduce<4, 2>(a.partial, a.dw, nsplit, ny, nz, total4, a.Cin_real, a.accumulate, s);
    } else {
      if (kg == 1) launch_big_reduce<8, 1>(a.partial, a.dw, nsplit, ny, nz, total4, a.Cin_real, a.accumulate, s);
      else launch_big_reduce<8, 2>(a.partial, a.dw, nsplit, ny, nz, total4, a.Cin_real, a.accumulate, s);
    }
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace flair
