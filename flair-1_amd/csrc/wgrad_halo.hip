// Weight gradient of the 3x3 / stride-1 / pad-1 layers with few channels (decoder tail, head, layer1):
//   dW[k][(r,s)][c] = sum_p dY[p][k] * X[p + (r,s)][c]
// These layers are HBM-bound (hundreds of MB of activations against a few KB of gradient), so the kernel is
// organised around reading X and dY exactly once: a persistent workgroup walks 8x32-pixel tiles, stages the
// X halo (10x34 pixels x CK channels) and the dY tile (256 pixels x CO channels) in LDS, and accumulates ALL
// nine taps for its (CO x CK) block of the gradient in registers; the reduction index (pixels) is the MFMA K
// dimension, fetched from the pixel-major LDS images with ds_read_b64_tr_b16 (f32: dword reads).
// Each wave owns two tile rows (K split), partial accumulators are merged through LDS once per workgroup and
// written as one fp32 slab per workgroup; wgrad_reduce_kernel (wgrad.hip) sums the slabs in a fixed order.
#include "common.h"
#include "prof.h"
#include "tune.h"
#include "ops.h"

namespace flair {

void launch_wgrad_reduce(const float* partial, float* dw, int splits, int Cout, int Cout_pad, int Kpad, int Cin,
                         int Cin_real, int R, int S, int accumulate, hipStream_t s);  // wgrad.hip

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TH = 8, TW = 32, HH = TH + 2, HW_ = TW + 2, HPIX = HH * HW_;

template <typename T> struct HFrag;
template <> struct HFrag<bf16_t> {
  static constexpr int KSTEP = 32;  // pixels per MFMA
  // 16 columns starting at byte cb, rows row0 + 0..31 (row stride STRIDE bytes)
  template <int STRIDE>
  __device__ static __forceinline__ u32x4 load(const unsigned char* img, int row0, int cb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) s16x4_t* lds_p;
    // pixel -> K-slot map: the `lo` read takes the even pixels of the lane group's eight, the `hi` read the odd ones.  A
    // 32-lane half then reads eight rows 2 * STRIDE apart: with STRIDE = 48 or 80 bytes they fall on disjoint bank octets
    // (rows q, q + 4 put two rows on half of the banks: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE, and the LDS was
    // active 45-62 % of these kernels' time).  Both operands use the same map, the products are unchanged.
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + (row0 + 8 * g + 2 * q) * STRIDE + cb + 8 * p));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + (row0 + 8 * g + 2 * q + 1) * STRIDE + cb + 8 * p));
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
template <> struct HFrag<float> {
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

struct WgHaloArgs {
  const void* x0; const void* x1; int C0, C1, up0;
  int N, H, W;
  const void* dy; int dy_ld; int Cout;
  float* partial;   // [gridDim.x][Cout_pad][Kg] fp32
  int Cout_pad, Kg;
  int ntiles;
  const float* in_scale; const float* in_shift;   // lazy BN + ReLU on x (WgradArgs::in_scale)
  float* dbias_partial;   // [CO][gridDim.x] column sums of dY per workgroup (DB variants: one input- and one output-channel block)
};

template <typename T, int CK, int CO>
struct WgHaloCfg {
  static constexpr int CH = Elem<T>::CH;
  static constexpr int CPP = CK / CH;                 // chunks per halo pixel
  static constexpr int DPP = CO / CH;                 // chunks per dY pixel
  static constexpr int XSTRIDE = CK * (int)sizeof(T) + 16;
  static constexpr int DSTRIDE = CO * (int)sizeof(T) + 16;
  static constexpr int XBYTES = HPIX * XSTRIDE;
  static constexpr int DBYTES = TH * TW * DSTRIDE;
  static constexpr int COT = CO / 16, CIT = CK / 16, NT = COT * CIT * 9;
  static constexpr int RED = NT * 4 * 64 * 4;         // cross-wave reduction buffer
  static constexpr int STAGE = XBYTES + DBYTES;
  static constexpr int SMEM = STAGE > RED ? STAGE : RED;
  static constexpr int XITEMS = (HPIX * CPP + 255) / 256;
  static constexpr int DITEMS = (TH * TW * DPP + 255) / 256;
};

// DB: also the bias gradient (column sums of dY), summed from the staged dY chunks: a thread always stages the same chunk column
template <typename T, int CK, int CO, bool LZ, bool DB = false>
__global__ __launch_bounds__(256) void wgrad3x3_halo_kernel(const WgHaloArgs a) {
  using Cfg = WgHaloCfg<T, CK, CO>;
  constexpr int CH = Cfg::CH, CPP = Cfg::CPP, DPP = Cfg::DPP, COT = Cfg::COT, CIT = Cfg::CIT;
  constexpr int KSTEP = HFrag<T>::KSTEP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xh = smem;
  unsigned char* dyt = smem + Cfg::XBYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int cbase = blockIdx.y * CK;     // input-channel chunk of this workgroup
  const int kbase = blockIdx.z * CO;     // output-channel chunk
  const T* __restrict__ x0 = (const T*)a.x0;
  const T* __restrict__ x1 = (const T*)a.x1;
  const T* __restrict__ dy = (const T*)a.dy;
  const bool use0 = cbase < a.C0;
  const T* __restrict__ xb = use0 ? x0 : x1;
  const int Hs = (use0 && a.up0) ? (H >> 1) : H, Ws = (use0 && a.up0) ? (W >> 1) : W, Cs = use0 ? a.C0 : a.C1;
  const int sh = (use0 && a.up0) ? 1 : 0;
  const int coff = use0 ? cbase : cbase - a.C0;

  // wave -> (16x16 gradient block, K part): the 4 waves cover the COT*CIT blocks first and split the tile
  // rows (the MFMA K dimension) with what is left, so every wave keeps just 9 accumulator tiles (one per tap)
  constexpr int WC = COT * CIT, WK = 4 / WC, ROWS = TH / WK;
  static_assert(WC == 1 || WC == 2 || WC == 4, "wave mapping");
  const int combo = wave % WC, kp = wave / WC;
  const int co = combo / CIT, ci = combo % CIT;
  f32x4_t acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) acc[tp] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float lsc[CH], lsh[CH];   // this thread's chunk column of the lazy transform (256 % CPP == 0)
  if (LZ && a.in_scale) {
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      lsc[e] = a.in_scale[cbase + (t % CPP) * CH + e];
      lsh[e] = a.in_shift[cbase + (t % CPP) * CH + e];
    }
  }

  // The next tile's X halo and dY tile ride in registers while the current tile is multiplied (round 1 issued the loads
  // of a tile, waited for them and only then computed: with ~3 workgroups per CU the HBM latency of every tile was
  // exposed — 330 us for 300 MB, a fifth of the HBM rate).
  // (round 2b: two register slots, the loads of tile i + 2 are in flight while tile i is multiplied — the MFMA phase of a tile
  // is a fraction of an HBM round trip, so distance 1 still waited for every tile)
  u32x4 xrA[Cfg::XITEMS], drA[Cfg::DITEMS], xrB[Cfg::XITEMS], drB[Cfg::DITEMS];
  unsigned xbitsA = 0, dbitsA = 0, xbitsB = 0, dbitsB = 0;
  auto load_tile = [&](int tile, u32x4 (&xr)[Cfg::XITEMS], u32x4 (&dr)[Cfg::DITEMS], unsigned& xbits, unsigned& dbits) {
    const bool tok = tile < a.ntiles;
    const int tl = tok ? tile : 0;
    const int n = tl / (tiles_x * tiles_y);
    const int trem = tl - n * tiles_x * tiles_y;
    const int y0 = (trem / tiles_x) * TH, x0p = (trem % tiles_x) * TW;
    unsigned xb2 = 0, db2 = 0;
#pragma unroll
    for (int k = 0; k < Cfg::XITEMS; ++k) {
      const int it = t + 256 * k;
      const int hp = it / CPP, ch = it - hp * CPP;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0p - 1 + hx;
      const bool ok = tok && (it < HPIX * CPP) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((n * Hs + (iy >> sh)) * Ws + (ix >> sh)) * Cs + coff + ch * CH) : 0u;
      xr[k] = *reinterpret_cast<const u32x4*>(xb + off);
      xb2 |= (ok ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < Cfg::DITEMS; ++k) {
      const int it = t + 256 * k;
      const int px = it / DPP, ch = it - px * DPP;
      const int py = px / TW, pxx = px - py * TW;
      const int col = kbase + ch * CH;
      const bool ok = tok && (it < TH * TW * DPP) && col < a.dy_ld;
      const unsigned off = ok ? (unsigned)(((n * H + y0 + py) * W + x0p + pxx) * a.dy_ld + col) : 0u;
      dr[k] = *reinterpret_cast<const u32x4*>(dy + off);
      db2 |= (ok ? 1u : 0u) << k;
    }
    xbits = xb2; dbits = db2;
  };
  load_tile(blockIdx.x, xrA, drA, xbitsA, dbitsA);
  load_tile(blockIdx.x + gridDim.x, xrB, drB, xbitsB, dbitsB);
  float bsum[CH];
  if constexpr (DB) {
    static_assert(256 % DPP == 0, "a thread keeps one dY chunk column");
#pragma unroll
    for (int e = 0; e < CH; ++e) bsum[e] = 0.f;
  }
  auto do_tile = [&](int tile, u32x4 (&xr)[Cfg::XITEMS], u32x4 (&dr)[Cfg::DITEMS], unsigned& xbits, unsigned& dbits) {
    __syncthreads();  // the previous tile's fragment reads are done
#pragma unroll
    for (int k = 0; k < Cfg::XITEMS; ++k) {
      const int it = t + 256 * k;
      if (it < HPIX * CPP) {
        const int hp = it / CPP, ch = it - hp * CPP;
        const u32x4 xv = (LZ && a.in_scale) ? chunk_bn_relu<T>(xr[k], lsc, lsh) : xr[k];
        *reinterpret_cast<u32x4*>(xh + hp * Cfg::XSTRIDE + ch * 16) = xv & (0u - ((xbits >> k) & 1u));
      }
    }
#pragma unroll
    for (int k = 0; k < Cfg::DITEMS; ++k) {
      const int it = t + 256 * k;
      if (it < TH * TW * DPP) {
        const int px = it / DPP, ch = it - px * DPP;
        const u32x4 dv = dr[k] & (0u - ((dbits >> k) & 1u));
        *reinterpret_cast<u32x4*>(dyt + px * Cfg::DSTRIDE + ch * 16) = dv;
        if constexpr (DB) {
          float f[CH];
          chunk_to_f<T>(__builtin_bit_cast(uint4, dv), f);
#pragma unroll
          for (int e = 0; e < CH; ++e) bsum[e] += f[e];
        }
      }
    }
    load_tile(tile + 2 * gridDim.x, xr, dr, xbits, dbits);   // the slot just staged is free
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMA phase (hipcc would sink it to its use)
    __syncthreads();
    // ---- this wave's tile rows (K = ROWS x 32 pixels) of its 16x16 gradient block, all nine taps
#pragma unroll
    for (int yy = 0; yy < ROWS; ++yy) {
      const int y = kp * ROWS + yy;
#pragma unroll
      for (int xs = 0; xs < TW; xs += KSTEP) {
        const u32x4 af = HFrag<T>::template load<Cfg::DSTRIDE>(dyt, y * TW + xs, co * 16 * (int)sizeof(T), lane);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int r = tp / 3, s = tp - 3 * r;
          const u32x4 bf = HFrag<T>::template load<Cfg::XSTRIDE>(xh, (y + r) * HW_ + xs + s, ci * 16 * (int)sizeof(T), lane);
          HFrag<T>::mma(af, bf, acc[tp]);
        }
      }
    }
  };
  for (int tile = blockIdx.x; tile < a.ntiles; tile += 2 * gridDim.x) {
    do_tile(tile, xrA, drA, xbitsA, dbitsA);
    if (tile + gridDim.x < a.ntiles) do_tile(tile + gridDim.x, xrB, drB, xbitsB, dbitsB);
  }

  if constexpr (DB) {   // workgroup sum of the staged dY columns, fixed order
    __syncthreads();
    float* bs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < CH; ++e) bs[t * CH + e] = bsum[e];
    __syncthreads();
    if (t < CO) {
      const int chunk = t / CH, e = t - chunk * CH;
      float x = 0.f;
      for (int th = chunk; th < 256; th += DPP) x += bs[th * CH + e];
      a.dbias_partial[(long)(kbase + t) * gridDim.x + blockIdx.x] = x;
    }
  }
  // ---- merge the K parts (fixed order kp 0 <- 1, 2, 3), then one slab per workgroup
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  for (int w = 1; w < WK; ++w) {
    if (kp == w) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[((combo * 9 + tp) * 4 + e) * 64 + lane] = acc[tp][e];
    }
    __syncthreads();
    if (kp == 0) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[tp][e] += red[((combo * 9 + tp) * 4 + e) * 64 + lane];
    }
    __syncthreads();
  }
  if (kp == 0) {
    const int Cin = a.C0 + a.C1;
    float* __restrict__ part = a.partial + (long)blockIdx.x * a.Cout_pad * a.Kg;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kbase + co * 16 + lq * 4 + e;
        const int c = cbase + ci * 16 + lr;
        if (k < a.Cout_pad) part[(long)k * a.Kg + tp * Cin + c] = acc[tp][e];
      }
  }
}

template <typename T, int CK, int CO, bool LZ, bool DB = false>
int launch_wg_halo_l(const WgHaloArgs& a, int nsplit, hipStream_t s) {
  using Cfg = WgHaloCfg<T, CK, CO>;
  auto kern = wgrad3x3_halo_kernel<T, CK, CO, LZ, DB>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int Cin = a.C0 + a.C1;
  dim3 grid(nsplit, Cin / CK, a.Cout_pad / CO);
  const double M = (double)a.N * a.H * a.W;
  ProfScope ps(sizeof(T) == 2 ? (CK == 32 ? "wgrad3x3_halo_bf16_ck32" : "wgrad3x3_halo_bf16_ck16") : "wgrad3x3_halo_f32",
               2.0 * M * a.Cout * 9.0 * Cin, (M * a.dy_ld + M * (a.C0 / (a.up0 ? 4.0 : 1.0) + a.C1)) * sizeof(T), s);
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::SMEM, s, a);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

// resident workgroups per CU of a variant (registers and LDS), asked from the runtime once; 3 without a device
template <typename T, int CK, int CO, bool LZ>
int wg_halo_per_cu() {
  static const int per_cu = [] {
    int nb = 0;
    auto kern = wgrad3x3_halo_kernel<T, CK, CO, LZ>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WgHaloCfg<T, CK, CO>::SMEM) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, WgHaloCfg<T, CK, CO>::SMEM) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      return 3;
    }
    return nb > 4 ? 4 : nb;
  }();
  return per_cu;
}

template <typename T>
int wg_halo_per_cu_t(int CK, int CO, bool lz) {
  if (CK == 16) {
    if (CO == 16) return lz ? wg_halo_per_cu<T, 16, 16, true>() : wg_halo_per_cu<T, 16, 16, false>();
    return lz ? wg_halo_per_cu<T, 16, 32, true>() : wg_halo_per_cu<T, 16, 32, false>();
  }
  if (CO == 16) return lz ? wg_halo_per_cu<T, 32, 16, true>() : wg_halo_per_cu<T, 32, 16, false>();
  return lz ? wg_halo_per_cu<T, 32, 32, true>() : wg_halo_per_cu<T, 32, 32, false>();
}

template <typename T, int CK, int CO>
int launch_wg_halo(const WgHaloArgs& a, int nsplit, hipStream_t s) {
  if constexpr (CK == 16 && CO == 16) {
    if (a.dbias_partial) return a.in_scale ? launch_wg_halo_l<T, CK, CO, true, true>(a, nsplit, s) : launch_wg_halo_l<T, CK, CO, false, true>(a, nsplit, s);
  }
  return a.in_scale ? launch_wg_halo_l<T, CK, CO, true>(a, nsplit, s) : launch_wg_halo_l<T, CK, CO, false>(a, nsplit, s);
}

}  // namespace

// dtype < 0: applicability only (nsplit is not computed)
static bool wg_halo_geom(int dtype, const WgradArgs& a, int& CK, int& CO, int& nsplit, int& Cout_pad) {
  const int Cin = a.C0 + a.C1;
  if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1) return false;
  if (a.Hout != a.Hin || a.Wout != a.Win || (a.Hin % TH) || (a.Win % TW)) return false;
  if (Cin == 16 && a.C1 == 0) CK = 16;
  else if (Cin % 32 == 0 && a.C0 % 32 == 0 && Cin <= 256) CK = 32;
  else return false;
  if (a.Cout > 64 || (a.dy_ld % 16)) return false;
  Cout_pad = (int)round_up(a.Cout, 16);
  CO = Cout_pad % 32 == 0 ? 32 : 16;
  const long ntiles = (long)a.N * a.Hin * a.Win / (TH * TW);
  const int per = (Cin / CK) * (Cout_pad / CO);
  nsplit = 1;
  if (dtype < 0) return true;
  // persistent grid = the workgroups resident at once (a surplus would run as a second, unbalanced round)
  int wgs = tune("FLAIR_WGH_WGS", 0);
  if (wgs <= 0) wgs = 256 * (dtype == DT_F32 ? wg_halo_per_cu_t<float>(CK, CO, a.in_scale != nullptr) : wg_halo_per_cu_t<bf16_t>(CK, CO, a.in_scale != nullptr));
  long ns = wgs / per;
  if (ns > ntiles) ns = ntiles;
  if (ns < 1) ns = 1;
  nsplit = (int)ns;
  return true;
}

bool wgrad_halo_applicable(const WgradArgs& a) {
  int CK, CO, ns, cp;
  return wg_halo_geom(-1, a, CK, CO, ns, cp);
}

size_t wgrad_halo_workspace_bytes(int dtype, const WgradArgs& a) {
  int CK, CO, ns, cp;
  if (!wg_halo_geom(dtype, a, CK, CO, ns, cp)) return 0;
  return (size_t)ns * cp * 9 * (a.C0 + a.C1) * sizeof(float);
}

int launch_wgrad_halo(int dtype, const WgradArgs& a, hipStream_t s) {
  int CK, CO, nsplit, Cout_pad;
  if (!wg_halo_geom(dtype, a, CK, CO, nsplit, Cout_pad)) return -2;
  const int Cin = a.C0 + a.C1;
  WgHaloArgs h;
  h.x0 = a.x0; h.x1 = a.x1; h.C0 = a.C0; h.C1 = a.C1; h.up0 = a.up0; h.N = a.N; h.H = a.Hin; h.W = a.Win;
  h.dy = a.dy; h.dy_ld = a.dy_ld; h.Cout = a.Cout; h.partial = a.partial; h.Cout_pad = Cout_pad; h.Kg = 9 * Cin;
  h.ntiles = (int)((long)a.N * a.Hin * a.Win / (TH * TW));
  h.in_scale = a.in_scale; h.in_shift = a.in_shift;
  h.dbias_partial = nullptr;
  if (a.dbias) {
    if (!wgrad_dbias_fusable(dtype, a) || !a.dbias_partial || nsplit > WGRAD_DBIAS_ROWS) return -6;
    h.dbias_partial = a.dbias_partial;
  }
  int rc;
  if (dtype == DT_F32) {
    if (CK == 16) rc = CO == 16 ? launch_wg_halo<float, 16, 16>(h, nsplit, s) : launch_wg_halo<float, 16, 32>(h, nsplit, s);
    else rc = CO == 16 ? launch_wg_halo<float, 32, 16>(h, nsplit, s) : launch_wg_halo<float, 32, 32>(h, nsplit, s);
  } else {
    if (CK == 16) rc = CO == 16 ? launch_wg_halo<bf16_t, 16, 16>(h, nsplit, s) : launch_wg_halo<bf16_t, 16, 32>(h, nsplit, s);
    else rc = CO == 16 ? launch_wg_halo<bf16_t, 32, 16>(h, nsplit, s) : launch_wg_halo<bf16_t, 32, 32>(h, nsplit, s);
  }
  if (rc) return rc;
  launch_wgrad_reduce(a.partial, a.dw, nsplit, a.Cout, Cout_pad, 9 * Cin, Cin, a.Cin_real, 3, 3, a.accumulate, s);
  if (h.dbias_partial) {
    rc = partial_rows_sum(h.dbias_partial, nsplit, a.Cout, a.dbias, s);
    if (rc) return rc;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// the 16 -> <= 16 channel layers (one input- and one output-channel block per workgroup: every dY tile is staged exactly once)
bool wgrad_dbias_fusable(int dtype, const WgradArgs& a) {
  int CK, CO, ns, cp;
  if (!tune("FLAIR_DBIAS_FUSE", 1) || !wg_halo_geom(-1, a, CK, CO, ns, cp)) return false;
  (void)dtype;
  return CK == 16 && CO == 16 && cp == 16 && a.C0 + a.C1 == 16 && a.dy_ld == 16;
}

}  // namespace flair
