// Implicit-GEMM convolution for gfx950: D[pixel][cout] = sum_k im2col(X)[pixel][k] * W[cout][k].
//
// One gather-form kernel serves
//   * every forward conv of smp Unet(resnet34) (7x7 s2, 3x3 s1/s2, 1x1 s2),
//   * the decoder's nearest-x2 upsample + skip concat, fused into the gather (two sources),
//   * the data gradient (flipped/transposed packed weights, in_div = forward stride).
// K is walked in 128-byte steps (64 bf16 / 32 f32) through a double-buffered, XOR-swizzled LDS
// image with 128-byte rows (conflict-free ds_read_b128, cdna_hip_programming.md §5.5 T2);
// MFMA 16x16x32 bf16 (throughput) or 16x16x4 f32 (parity: exact fp32 FMA chain).
// Epilogue: accumulators -> LDS -> 16-byte coalesced NHWC stores (or fp32 NCHW for the head),
// plus per-row-block partial BatchNorm sums (deterministic: no atomics).
#include "common.h"
#include "prof.h"
#include "tune.h"

namespace flair {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // first-class 16-byte vector (SSA-friendly, unlike HIP's uint4 struct)

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
struct ConvCfg {
  static constexpr int CH = Elem<T>::CH;
  static constexpr int BKE = 8 * CH;
  static constexpr int STAGE = (BM + BN) * 128;
  static constexpr int CLD = BN * (int)sizeof(T) + 16;
  static constexpr int CT = BM * CLD;
  static constexpr int MAIN = (2 * STAGE > CT) ? 2 * STAGE : CT;
  static constexpr int STATS = WAVES_M * BN * 2 * 4;
  static constexpr int SMEM = MAIN + STATS;
};

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvArgs a) {
  using Cfg = ConvCfg<T, BM, BN, WAVES_M, WAVES_N>;
  constexpr int CH = Cfg::CH, BKE = Cfg::BKE;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int AROWS = BM / 32;
  constexpr int BROWS = (BN + 31) / 32;
  static_assert(WAVES_M * WAVES_N == 4, "256 threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const long M = (long)a.N * a.Hout * a.Wout;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C0 + a.C1;
  const int cc = t & 7, rb = t >> 3;
  const T* __restrict__ src0 = (const T*)a.src0;
  const T* __restrict__ src1 = (const T*)a.src1;
  const T* __restrict__ wp = (const T*)a.w;
  int cR = a.R, cS = a.S, cKpad = a.Kpad, c_oy = a.out_oy, c_ox = a.out_ox;
  if (a.ncls) {   // one parity class of a stride-2 data gradient per blockIdx.z
    const int cls = blockIdx.z;
    c_oy = cls >> 1; c_ox = cls & 1;
    cR = c_oy ? 2 : 1; cS = c_ox ? 2 : 1;
    cKpad = a.cls_kpad[cls];
    wp = (const T*)a.cls_w[cls];
  }

  int ih0[AROWS], iw0[AROWS], nb[AROWS];
  {
    const int HW = a.Hout * a.Wout;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      long m = m0 + rb + 32 * i;
      if (m < M) {
        int n = (int)(m / HW);
        int rem = (int)(m - (long)n * HW);
        int ho = rem / a.Wout, wo = rem - ho * a.Wout;
        ih0[i] = ho * a.out_mul - a.pad;
        iw0[i] = wo * a.out_mul - a.pad;
        nb[i] = n;
      } else {
        ih0[i] = -(1 << 28); iw0[i] = -(1 << 28); nb[i] = 0;  // fails every bounds check -> zero rows
      }
    }
  }
  // K decode for this thread's chunk column
  int kc, kr, ks;
  {
    int kk = cc * CH;
    int tap = kk / Cin;
    kc = kk - tap * Cin;
    kr = tap / cS;
    ks = tap - kr * cS;
  }
  const int Hs0 = a.up0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.up0 ? (a.Win >> 1) : a.Win;
  const int nsteps = cKpad / BKE;

  // Branch-free gather: every lane always issues its 16-byte load (out-of-image taps read the tensor base
  // and are zeroed by a select when the registers are written to LDS), so all loads of a K step are in
  // flight together behind one s_waitcnt instead of one memory round trip per row.
  // Two register sets: the loads of K step s+2 are issued while step s is multiplied and are written to LDS
  // one iteration later, so a load has a whole iteration (MFMA phase + barrier) to land instead of having
  // to beat the MFMA phase of its own iteration.
  struct Regs { u32x4 a[AROWS]; u32x4 b[BROWS]; unsigned m[AROWS]; };
  auto load_regs = [&](int step, Regs& R) {
    const bool use0 = kc < a.C0;
    const T* __restrict__ base = use0 ? src0 : src1;
    const int Hs = use0 ? Hs0 : a.Hin, Ws = use0 ? Ws0 : a.Win, Cs = use0 ? a.C0 : a.C1;
    const int sh = (use0 && a.up0) ? 1 : 0;
    const int coff = use0 ? kc : kc - a.C0;
    const bool tap_ok = kr < cR;
    const int bstep = step < nsteps ? step : nsteps - 1;  // past the end: harmless re-read, never consumed
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int hn = ih0[i] + kr, wq = iw0[i] + ks;
      bool ok = tap_ok;
      if (a.in_div == 2) {
        ok = ok && (((hn | wq) & 1) == 0);
        hn >>= 1; wq >>= 1;
      }
      ok = ok && ((unsigned)hn < (unsigned)a.Hin) && ((unsigned)wq < (unsigned)a.Win);
      const unsigned off = ok ? (unsigned)(((nb[i] * Hs + (hn >> sh)) * Ws + (wq >> sh)) * Cs + coff) : 0u;
      R.a[i] = *reinterpret_cast<const u32x4*>(base + off);
      R.m[i] = ok ? 0xffffffffu : 0u;
    }
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      const int row = (BN >= 32) ? rb + 32 * j : (rb & (BN - 1));
      R.b[j] = *reinterpret_cast<const u32x4*>(wp + (long)(n0 + row) * cKpad + (long)bstep * BKE + cc * CH);
    }
    // advance the K decode by one step
    kc += BKE;
    while (kc >= Cin) {
      kc -= Cin;
      if (++ks == cS) { ks = 0; ++kr; }
    }
  };
  auto write_lds = [&](int stage, const Regs& R) {
    unsigned char* sA = smem + stage * Cfg::STAGE;
    unsigned char* sB = sA + BM * 128;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) *reinterpret_cast<u32x4*>(sA + lds_off(rb + 32 * i, cc)) = R.a[i] & R.m[i];
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      if constexpr (BN >= 32) {
        *reinterpret_cast<u32x4*>(sB + lds_off(rb + 32 * j, cc)) = R.b[j];
      } else {
        if (rb < BN) *reinterpret_cast<u32x4*>(sB + lds_off(rb, cc)) = R.b[j];
      }
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lq = lane >> 4;
  auto compute = [&](int cur) {
    const unsigned char* sA = smem + cur * Cfg::STAGE;
    const unsigned char* sB = sA + BM * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const u32x4*>(sA + lds_off(wm * WM + i * 16 + lr, lq + 4 * h));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const u32x4*>(sB + lds_off(wn * WN + j * 16 + lr, lq + 4 * h));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
    }
  };

  // One register set (prefetch distance 1): a second set (distance 2) measured +4 % on the 128x128 tile but
  // pushes it past 256 VGPRs once the fused epilogue is in, and this kernel now only serves the strided /
  // 1x1 / 7x7 layers.  The prefetch is pinned at the top of the step (hipcc would sink it to the ds_write).
  Regs r0;
  load_regs(0, r0);
  write_lds(0, r0);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    load_regs(step + 1, r0);
    __builtin_amdgcn_sched_barrier(0);
    compute(step & 1);
    write_lds((step + 1) & 1, r0);
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  unsigned char* ct = smem;  // C tile [BM][CLD bytes]; all stage reads are behind the barrier above
  float* st = reinterpret_cast<float*>(smem + Cfg::MAIN);
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * WN + j * 16 + lr;
    const bool cin_ok = (n0 + col) < a.Cout;
    const float osc = (a.oscale && cin_ok) ? a.oscale[n0 + col] : 1.f;
    const float bias = ((a.bias && cin_ok) ? a.bias[n0 + col] : 0.f) + ((a.oshift && cin_ok) ? a.oshift[n0 + col] : 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * WM + i * 16 + lq * 4 + r;
        T v = Elem<T>::from_f(fmaf(acc[i][j][r], osc, bias));
        float vf = Elem<T>::to_f(v);
        s1[j] += vf;
        s2[j] += vf * vf;
        *reinterpret_cast<T*>(ct + row * Cfg::CLD + col * (int)sizeof(T)) = v;
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
      if (lq == 0) {
        st[(wm * BN + wn * WN + j * 16 + lr) * 2 + 0] = s1[j];
        st[(wm * BN + wn * WN + j * 16 + lr) * 2 + 1] = s2[j];
      }
    }
  }
  __syncthreads();
  if (a.stats && t < BN && (n0 + t) < a.Cout) {
    float x1 = 0.f, x2 = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES_M; ++w) { x1 += st[(w * BN + t) * 2]; x2 += st[(w * BN + t) * 2 + 1]; }
    // channel-major partials [2][Cout][gridDim.x]: the finalize kernel reads them coalesced
    a.stats[(long)(n0 + t) * gridDim.x + blockIdx.x] = x1;
    a.stats[((long)a.Cout + n0 + t) * gridDim.x + blockIdx.x] = x2;
  }
  if (a.out) {
    constexpr int CPR = BN / CH;
    T* __restrict__ out = (T*)a.out;
    for (int idx = t; idx < BM * CPR; idx += 256) {
      const int row = idx / CPR, ch = idx - row * CPR;
      const long m = m0 + row;
      const int n = n0 + ch * CH;
      if (m < M && n < a.Cout) {
        uint4 v = *reinterpret_cast<const uint4*>(ct + row * Cfg::CLD + ch * 16);
        long mo = m;
        if (a.out_sub) {
          const int HWo = a.Hout * a.Wout;
          const int img = (int)(m / HWo), rem = (int)(m - (long)img * HWo);
          const int ho = rem / a.Wout, wo = rem - ho * a.Wout;
          mo = ((long)img * 2 * a.Hout + 2 * ho + c_oy) * (2 * a.Wout) + 2 * wo + c_ox;
        }
        T* dst = out + mo * a.out_ld + n;
        if (a.ores || a.orelu) {
          float fa[CH];
          chunk_to_f<T>(v, fa);
          if (a.ores) {
            float fb[CH];
            chunk_to_f<T>(*reinterpret_cast<const uint4*>((const T*)a.ores + mo * a.out_ld + n), fb);
#pragma unroll
            for (int e = 0; e < CH; ++e) fa[e] += fb[e];
          }
          if (a.orelu) {
#pragma unroll
            for (int e = 0; e < CH; ++e) fa[e] = fmaxf(fa[e], 0.f);
          }
          v = f_to_chunk<T>(fa);
        }
        if (a.accumulate) {
          uint4 o = *reinterpret_cast<const uint4*>(dst);
          float fa[CH], fb[CH];
          chunk_to_f<T>(v, fa);
          chunk_to_f<T>(o, fb);
#pragma unroll
          for (int e = 0; e < CH; ++e) fa[e] += fb[e];
          v = f_to_chunk<T>(fa);
        }
        *reinterpret_cast<uint4*>(dst) = v;
      }
    }
  }
  if (a.out_nchw) {
    const int HW = a.Hout * a.Wout;
    for (int idx = t; idx < BM * BN; idx += 256) {
      const int nl = idx / BM, ml = idx - nl * BM;
      const long m = m0 + ml;
      const int n = n0 + nl;
      if (m < M && n < a.Cout) {
        const long img = m / HW, pix = m - img * HW;
        a.out_nchw[(img * a.Cout + n) * HW + pix] =
            Elem<T>::to_f(*reinterpret_cast<const T*>(ct + ml * Cfg::CLD + nl * (int)sizeof(T)));
      }
    }
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_cfg(const ConvArgs& a, hipStream_t s) {
  using Cfg = ConvCfg<T, BM, BN, WAVES_M, WAVES_N>;
  const long M = (long)a.N * a.Hout * a.Wout;
  dim3 grid(cdiv(M, BM), cdiv(a.Cout, BN), a.ncls ? a.ncls : 1);
  auto kern = conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  {
    // algorithmic work: 2*M*Cout*K real MACs (a data gradient with in_div = 2 only touches 1/4 of the taps)
    const double flops = 2.0 * (double)M * a.Cout * a.Kg / (a.in_div * a.in_div);
    const double bytes = ((double)a.N * a.Hin * a.Win / (a.in_div * a.in_div) / (a.up0 ? 4 : 1) * a.C0 +
                          (double)a.N * a.Hin * a.Win * a.C1 + (double)M * a.Cout * (a.accumulate ? 2 : 1)) * sizeof(T) +
                         (double)a.Cout * a.Kg * sizeof(T);
    static const char* names[2][4] = {{"conv_igemm_f32_128x128", "conv_igemm_f32_128x64", "conv_igemm_f32_256x32", "conv_igemm_f32_256x16"},
                                      {"conv_igemm_bf16_128x128", "conv_igemm_bf16_128x64", "conv_igemm_bf16_256x32", "conv_igemm_bf16_256x16"}};
    const int ci = BN == 128 ? 0 : BN == 64 ? 1 : BN == 32 ? 2 : 3;
    ProfScope ps(names[sizeof(T) == 2][ci], flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::SMEM, s, a);
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

static inline int pick_bn(int cout) { return cout >= 128 ? 128 : cout >= 64 ? 64 : cout >= 32 ? 32 : 16; }
static inline int pick_bm(int cout) { return cout >= 64 ? 128 : 256; }

bool conv_halo_applicable(const ConvArgs& a);           // conv_halo.hip
int conv_halo_grid_rows(int dtype, const ConvArgs& a);
int launch_conv_halo(int dtype, const ConvArgs& a, hipStream_t s);
bool conv_halo_bnr_applicable(const ConvArgs& a);
bool conv_stem_applicable(int dtype, const ConvArgs& a);  // stem.hip
int conv_stem_grid_rows(const ConvArgs& a);
int launch_conv_stem(const ConvArgs& a, hipStream_t s);
bool conv_hg_applicable(int dtype, const ConvArgs& a);  // conv_hg.hip
int conv_hg_grid_rows(int dtype, const ConvArgs& a);
int launch_conv_hg(int dtype, const ConvArgs& a, hipStream_t s);

// true when launch_conv will pick a halo-tile kernel whose epilogue implements pool_c0 / out_skip
bool conv_tile_epilogue_ok(int dtype, const ConvArgs& a) {
  if ((a.Hout & 1) || (a.Wout & 1)) return false;
  if (conv_hg_applicable(dtype, a)) return a.pool_c0 % ((a.Cout % 128) == 0 ? 128 : 64) == 0;
  if (conv_halo_applicable(a)) return a.pool_c0 % (a.Cout <= 16 ? 16 : 32) == 0 && !a.out_nchw;
  return false;
}

bool conv_mfma_bound(int dtype, const ConvArgs& a) {
  static int mode = -1;  // tuning override FLAIR_BNR: 0 = never fuse, 1 = 128-wide halo-GEMM only, 2 = every halo-GEMM
  if (mode < 0) {
    const char* e = getenv("FLAIR_BNR");
    mode = e ? atoi(e) : 2;
  }
  if (mode == 0) return false;
  if (!conv_hg_applicable(dtype, a)) return conv_halo_bnr_applicable(a);   // (not MFMA-bound: the persistent small-channel kernel
                                                                           // saves the reduce pass's read of the gradient)
  return mode == 2 || (a.Cout % 128) == 0;
}

int conv_grid_rows(int dtype, const ConvArgs& a) {
  if (a.in_scale && conv_hg_applicable(dtype, a)) return conv_hg_grid_rows(dtype, a);
  if (a.in_scale && conv_halo_applicable(a)) return conv_halo_grid_rows(dtype, a);
  if (conv_stem_applicable(dtype, a)) return conv_stem_grid_rows(a);
  if (conv_hg_applicable(dtype, a)) return conv_hg_grid_rows(dtype, a);
  if (conv_halo_applicable(a)) return conv_halo_grid_rows(dtype, a);
  return cdiv((long)a.N * a.Hout * a.Wout, pick_bm(a.Cout));
}

int conv_weight_rows_pad(int cout) { return (int)round_up(cout, pick_bn(cout)); }

template <typename T>
static int launch_t(const ConvArgs& a, hipStream_t s) {
  constexpr int CH = Elem<T>::CH;
  if ((a.C0 % CH) || (a.C1 % CH) || (a.Kpad % (8 * CH))) return -2;
  if (a.ncls && (a.ncls != 4 || !a.out_sub)) return -2;
  for (int c = 0; c < a.ncls; ++c)
    if (!a.cls_w[c] || (a.cls_kpad[c] % (8 * CH))) return -2;
  if (a.out && (a.Cout % CH)) return -3;
  if (a.up0 && ((a.Hin | a.Win) & 1)) return -4;
  if (a.in_div != 1 && a.in_div != 2) return -5;
  // few pixel rows, long K (the SegFormer sequence-reduction convolutions: 8 192 rows x K = 1 280 .. 4 096 at B = 32, 64 workgroups
  // of 128 rows on 256 CUs): 32-row tiles, four times the workgroups
  const long M = (long)a.N * a.Hout * a.Wout;
  const int bn = pick_bn(a.Cout);
  const bool small = !a.stats && !a.ncls && bn >= 64 && a.Kpad >= 8 * 8 * Elem<T>::CH && (long)cdiv(M, 128) * cdiv(a.Cout, bn) < 256 &&
                     tune("FLAIR_IGEMM_BM32", 2);
  if (small) {
    // (32 x 32 tiles when even the 32-row tiles leave CUs without a second workgroup: K = 4 096 at 64 output channels)
    const int mode = tune("FLAIR_IGEMM_BM32", 2);
    if (mode >= 2 && (long)cdiv(M, 32) * cdiv(a.Cout, bn) < 512) return launch_cfg<T, 32, 32, 2, 2>(a, s);
    return bn == 128 ? launch_cfg<T, 32, 128, 2, 2>(a, s) : launch_cfg<T, 32, 64, 2, 2>(a, s);
  }
  switch (bn) {
    case 128: return launch_cfg<T, 128, 128, 2, 2>(a, s);
    case 64: return launch_cfg<T, 128, 64, 4, 1>(a, s);
    case 32: return launch_cfg<T, 256, 32, 4, 1>(a, s);
    default: return launch_cfg<T, 256, 16, 4, 1>(a, s);
  }
}

int launch_conv(int dtype, const ConvArgs& a, hipStream_t s) {
  if (a.preds_u8 && !conv_halo_preds_ok(dtype, a)) return -6;   // fused argmax: persistent small-channel kernel only
  if (a.out_sub && (a.out_nchw || a.stats || a.pool_c0 > 0 || a.bnr_partial || conv_hg_applicable(dtype, a) || conv_halo_applicable(a)))
    return -6;  // sub-sampled stores exist in the gather-form epilogue only
  // input / epilogue options only the halo-GEMM kernels implement (fused BN-backward apply, addend from another tensor, masked store)
  if ((a.ap_y || a.acc_src || a.bnr_mask) && !conv_hg_applicable(dtype, a)) return -6;
  if (a.acc_src && !conv_acc_src_ok(dtype, a)) return -6;
  if (a.bnr_mask && !a.bnr_partial) return -6;
  if (a.in_scale) {   // lazy BN + ReLU on the input: the halo-GEMM (>= 64 channels) and the small-channel halo kernel apply it
    if (conv_hg_applicable(dtype, a)) return (a.pool_c0 > 0 || a.bnr_partial) ? -6 : launch_conv_hg(dtype, a, s);
    if (!conv_halo_applicable(a) || (a.pool_c0 > 0 && !conv_tile_epilogue_ok(dtype, a))) return -6;
    const int ch = dtype == DT_F32 ? 4 : 8;
    // the tile epilogue stores whole 16-byte chunks: a ragged channel count needs the row padded to the next chunk
    // (the padded columns receive the zero-weight rows' results)
    if (a.out && (a.Cout % ch) && a.out_ld < (int)round_up(a.Cout, ch)) return -3;
    return launch_conv_halo(dtype, a, s);
  }
  if (a.pool_c0 > 0 && !conv_tile_epilogue_ok(dtype, a)) return -6;
  if (a.bnr_partial && !(conv_tile_epilogue_ok(dtype, a) && (conv_hg_applicable(dtype, a) || conv_halo_bnr_applicable(a)))) return -6;
  if (conv_stem_applicable(dtype, a)) return launch_conv_stem(a, s);     // 7x7 stride-2 stem
  if (conv_hg_applicable(dtype, a)) return launch_conv_hg(dtype, a, s);  // MFMA-bound 3x3 s1 layers: halo GEMM
  if (conv_halo_applicable(a)) {  // HBM-bound small-channel 3x3 layers: halo-tile direct kernel
    const int ch = dtype == DT_F32 ? 4 : 8;
    // the tile epilogue stores whole 16-byte chunks: a ragged channel count needs the row padded to the next chunk
    // (the padded columns receive the zero-weight rows' results)
    if (a.out && (a.Cout % ch) && a.out_ld < (int)round_up(a.Cout, ch)) return -3;
    return launch_conv_halo(dtype, a, s);
  }
  return dtype == DT_F32 ? launch_t<float>(a, s) : launch_t<bf16_t>(a, s);
}

}  // namespace flair
