// Weight gradient for gfx950: dW[k][(r,s),c] = sum_p dY[p][k] * im2col(X)[p][(r,s),c].
//
// Both operands sit in HBM pixel-major (NHWC), i.e. with the REDUCTION index as the slow axis,
// so the MFMA fragments (8 consecutive k per lane) are fetched from row-major [pixel][channel]
// LDS panels with the gfx950 transposed read ds_read_b64_tr_b16 (cdna_hip_programming.md T10);
// the f32 parity path reads single dwords.  The pixel axis is split over gridDim.z; fp32 partial
// slabs are summed in a fixed order by wgrad_reduce_kernel (deterministic, no float atomics),
// which also permutes [k][(r,s),c] -> PyTorch OIHW.
#include "common.h"
#include "prof.h"

namespace flair {

// LDS panel = [rows = pixels][128 bytes of channels]; XOR on byte bits 5,6 keeps the 8 rows a
// half-wave touches in one transposed read on distinct bank groups.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // first-class 16-byte vector (keeps staging arrays in registers)

__device__ __forceinline__ int wg_off(int row, int byte_in_row) {
  const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
  return row * 128 + (byte_in_row ^ (f << 5));
}

template <typename T, int BMK, int BNN>
struct WgCfg {
  static constexpr int CH = Elem<T>::CH;
  static constexpr int PC = 8 * CH;           // channels per panel (64 bf16 / 32 f32)
  static constexpr int BP = 64;               // pixels per LDS stage
  static constexpr int PA = BMK / PC;         // dY panels
  static constexpr int PB = BNN / PC;         // im2col panels
  static constexpr int PANEL = BP * 128;
  static constexpr int STAGE = (PA + PB) * PANEL;
  static constexpr int SMEM = 2 * STAGE;
};

template <typename T> struct WgFrag;
template <> struct WgFrag<bf16_t> {
  static constexpr int KSTEP = 32;  // pixels per MFMA
  // fragment for a 16-column group starting at byte cb of the panel, pixels kb..kb+31
  __device__ static __forceinline__ u32x4 load(const unsigned char* panel, int kb, int cb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) s16x4_t* lds_p;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(panel + wg_off(kb + 8 * g + q, cb + 8 * p)));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(panel + wg_off(kb + 8 * g + 4 + q, cb + 8 * p)));
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
template <> struct WgFrag<float> {
  static constexpr int KSTEP = 16;  // 4 MFMAs of k=4
  __device__ static __forceinline__ u32x4 load(const unsigned char* panel, int kb, int cb, int lane) {
    const int g = lane >> 4, i = lane & 15;
    u32x4 r;
    r.x = *reinterpret_cast<const unsigned*>(panel + wg_off(kb + g, cb + 4 * i));
    r.y = *reinterpret_cast<const unsigned*>(panel + wg_off(kb + 4 + g, cb + 4 * i));
    r.z = *reinterpret_cast<const unsigned*>(panel + wg_off(kb + 8 + g, cb + 4 * i));
    r.w = *reinterpret_cast<const unsigned*>(panel + wg_off(kb + 12 + g, cb + 4 * i));
    return r;
  }
  __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

struct WgKArgs {
  WgradArgs a;
  int Kg, Kpad, Cout_pad;
  long pix_per_split;
};

template <typename T, int BMK, int BNN>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgKArgs ka) {
  using Cfg = WgCfg<T, BMK, BNN>;
  constexpr int CH = Cfg::CH, PC = Cfg::PC, BP = Cfg::BP, PA = Cfg::PA, PB = Cfg::PB;
  constexpr int WM = BMK / 2, WN = BNN / 2, TM = WM / 16, TN = WN / 16;
  constexpr int KSTEP = WgFrag<T>::KSTEP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WgradArgs& a = ka.a;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int k0 = blockIdx.x * BMK;   // cout tile base
  const int n0 = blockIdx.y * BNN;   // (tap,c) tile base
  const long M = (long)a.N * a.Hout * a.Wout;
  const long pbeg = (long)blockIdx.z * ka.pix_per_split;
  long pend = pbeg + ka.pix_per_split;
  if (pend > M) pend = M;
  const int Cin = a.C0 + a.C1;
  const int cc = t & 7, rb = t >> 3;
  const T* __restrict__ x0 = (const T*)a.x0;
  const T* __restrict__ x1 = (const T*)a.x1;
  const T* __restrict__ dy = (const T*)a.dy;
  const int HW = a.Hout * a.Wout;
  const int Hs0 = a.up0 ? (a.Hin >> 1) : a.Hin, Ws0 = a.up0 ? (a.Win >> 1) : a.Win;

  // im2col column decode: fixed per thread and panel
  int cr[PB], cs[PB], ccn[PB];
  bool cok[PB];
#pragma unroll
  for (int pn = 0; pn < PB; ++pn) {
    int kk = n0 + pn * PC + cc * CH;
    cok[pn] = kk < ka.Kg;
    int tap = kk / Cin;
    ccn[pn] = kk - tap * Cin;
    cr[pn] = tap / a.S;
    cs[pn] = tap - cr[pn] * a.S;
  }

  // Branch-free staging loads (invalid lanes read the tensor base and are masked to zero when written to
  // LDS) so that every load of a stage is in flight behind a single wait.
  u32x4 ra[PA][2], rbv[PB][2];
  unsigned ma[PA][2], mb[PB][2];
  auto load_regs = [&](long p0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long p = p0 + rb + 32 * i;
      const bool pok = p < pend;
      const long pc = pok ? p : pbeg;
      const int n = (int)(pc / HW);
      const int rem = (int)(pc - (long)n * HW);
      const int ho = rem / a.Wout;
      const int wo = rem - ho * a.Wout;
#pragma unroll
      for (int pn = 0; pn < PA; ++pn) {
        const int col = k0 + pn * PC + cc * CH;
        const bool ok = pok && col < a.dy_ld;
        ra[pn][i] = *reinterpret_cast<const u32x4*>(dy + (ok ? pc * a.dy_ld + col : 0));
        ma[pn][i] = ok ? 0xffffffffu : 0u;
      }
#pragma unroll
      for (int pn = 0; pn < PB; ++pn) {
        const int hn = ho * a.stride - a.pad + cr[pn], wq = wo * a.stride - a.pad + cs[pn];
        const bool ok = pok && cok[pn] && (unsigned)hn < (unsigned)a.Hin && (unsigned)wq < (unsigned)a.Win;
        const bool use0 = ccn[pn] < a.C0;
        const T* __restrict__ base = use0 ? x0 : x1;
        const int sh = (use0 && a.up0) ? 1 : 0;
        const long off = use0 ? (((long)n * Hs0 + (hn >> sh)) * Ws0 + (wq >> sh)) * a.C0 + ccn[pn]
                              : (((long)n * a.Hin + hn) * a.Win + wq) * a.C1 + (ccn[pn] - a.C0);
        rbv[pn][i] = *reinterpret_cast<const u32x4*>(base + (ok ? off : 0));
        mb[pn][i] = ok ? 0xffffffffu : 0u;
      }
    }
  };
  auto write_lds = [&](int stage) {
    unsigned char* base = smem + stage * Cfg::STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int pn = 0; pn < PA; ++pn)
        *reinterpret_cast<u32x4*>(base + pn * Cfg::PANEL + wg_off(rb + 32 * i, cc * 16)) = ra[pn][i] & ma[pn][i];
#pragma unroll
      for (int pn = 0; pn < PB; ++pn)
        *reinterpret_cast<u32x4*>(base + (PA + pn) * Cfg::PANEL + wg_off(rb + 32 * i, cc * 16)) = rbv[pn][i] & mb[pn][i];
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nsteps = (int)((pend - pbeg + BP - 1) / BP);
  if (nsteps > 0) {
    load_regs(pbeg);
    write_lds(0);
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    load_regs(pbeg + (long)(step + 1) * BP);  // past the end every lane is masked: straight-line loop body
    __builtin_amdgcn_sched_barrier(0);        // pin the prefetch ahead of the MFMA phase (hipcc would sink it)
    const unsigned char* base = smem + cur * Cfg::STAGE;
#pragma unroll
    for (int kb = 0; kb < BP; kb += KSTEP) {
      u32x4 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int col = wm * WM + i * 16;  // channel within the dY tile
        af[i] = WgFrag<T>::load(base + (col / PC) * Cfg::PANEL, kb, (col % PC) * (int)sizeof(T), lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * WN + j * 16;
        bfr[j] = WgFrag<T>::load(base + (PA + col / PC) * Cfg::PANEL, kb, (col % PC) * (int)sizeof(T), lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) WgFrag<T>::mma(af[i], bfr[j], acc[i][j]);
    }
    write_lds(cur ^ 1);
    __syncthreads();
  }
  // partial slab [z][Cout_pad][Kpad]
  float* __restrict__ part = a.partial + (long)blockIdx.z * ka.Cout_pad * ka.Kpad;
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = k0 + wm * WM + i * 16 + lq * 4 + r;
        const int col = n0 + wn * WN + j * 16 + lr;
        if (row < ka.Cout_pad && col < ka.Kpad) part[(long)row * ka.Kpad + col] = acc[i][j][r];
      }
}

// dw[k][c][r][s] (+)= sum_z partial[z][k][(r*S+s)*Cin + c]
// EL consecutive elements x G = 256 / EL slab groups per workgroup, four independent partial sums per thread in flight, fixed
// summation order (group-major) -> bit-reproducible.  EL = 64 / G = 4: coalesced 256-byte slab reads for the large gradients;
// EL = 16 / G = 16 for the small-channel layers' many slabs of a tiny gradient (the head: 1024 slabs of 13 x 144 — with G = 4 that
// was 30 workgroups and 64 dependent round trips per thread, 43 us for 9 MB).
template <int EL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits,
                                                           int Cout, int Cout_pad, int Kpad, int Cin, int Cin_real, int R,
                                                           int S, int accumulate) {
  constexpr int G = 256 / EL;
  __shared__ float sh[G][EL];
  const long total = (long)Cout * R * S * Cin;
  const int e = threadIdx.x % EL, zg = threadIdx.x / EL;
  for (long base = (long)blockIdx.x * EL; base < total; base += (long)gridDim.x * EL) {
    const long idx = base + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0, kk = 0;
    if (idx < total) {
      k = (int)(idx / (R * S * Cin));
      kk = (int)(idx - (long)k * R * S * Cin);
      const float* p = part + (long)k * Kpad + kk;
      const long zs = (long)Cout_pad * Kpad;
      int z = zg;
      for (; z + 3 * G < splits; z += 4 * G) {
        s0 += p[(long)z * zs]; s1 += p[(long)(z + G) * zs]; s2 += p[(long)(z + 2 * G) * zs]; s3 += p[(long)(z + 3 * G) * zs];
      }
      for (; z < splits; z += G) s0 += p[(long)z * zs];
    }
    sh[zg][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zg == 0 && idx < total) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < G; g += 4) s += (sh[g][e] + sh[g + 1][e]) + (sh[g + 2][e] + sh[g + 3][e]);
      const int tap = kk / Cin, c = kk - tap * Cin;
      if (c < Cin_real) {
        const int r = tap / S, q = tap - r * S;
        float* d = dw + (((long)k * Cin_real + c) * R + r) * S + q;
        *d = accumulate ? (*d + s) : s;
      }
    }
    __syncthreads();
  }
}

static void wg_plan(int dtype, const WgradArgs& a, int& bmk, int& bnn, int& splits, long& pps, int& Kg, int& Kpad, int& Cout_pad) {
  const int Cin = a.C0 + a.C1;
  Kg = a.R * a.S * Cin;
  const int pc = dtype == DT_F32 ? 32 : 64;
  const bool big = (a.Cout >= 128 && Kg >= 128);
  bmk = big ? 128 : 64;
  bnn = big ? 128 : 64;
  if (dtype == DT_F32) { bmk = 64; bnn = 64; }  // f32 panels are 32 channels wide; keep LDS modest
  (void)pc;
  Kpad = (int)round_up(Kg, bnn);
  Cout_pad = (int)round_up(a.Cout, bmk);
  const long M = (long)a.N * a.Hout * a.Wout;
  const int tiles = (Cout_pad / bmk) * (Kpad / bnn);
  long want = (768 + tiles - 1) / tiles;  // ~3 blocks per CU in flight; every split costs a slab write + read
  if (want < 1) want = 1;
  long max_splits = (M + 64 * 8 - 1) / (64 * 8);  // at least 8 LDS stages per split
  if (max_splits < 1) max_splits = 1;
  if (want > max_splits) want = max_splits;
  pps = round_up((M + want - 1) / want, 64);
  splits = (int)((M + pps - 1) / pps);
}

bool wgrad_halo_applicable(const WgradArgs& a);          // wgrad_halo.hip
size_t wgrad_halo_workspace_bytes(int dtype, const WgradArgs& a);
int launch_wgrad_halo(int dtype, const WgradArgs& a, hipStream_t s);
bool wgrad_stem_applicable(int dtype, const WgradArgs& a);  // stem.hip
size_t wgrad_stem_workspace_bytes(const WgradArgs& a);
int launch_wgrad_stem(const WgradArgs& a, hipStream_t s);
bool wgrad_big_applicable(int dtype, const WgradArgs& a);  // wgrad_hg.hip
size_t wgrad_big_workspace_bytes(int dtype, const WgradArgs& a);
int launch_wgrad_big(int dtype, const WgradArgs& a, hipStream_t s);

void launch_wgrad_reduce(const float* partial, float* dw, int splits, int Cout, int Cout_pad, int Kpad, int Cin,
                         int Cin_real, int R, int S, int accumulate, hipStream_t s) {
  const long total = (long)Cout * R * S * Cin;
  ProfScope ps("wgrad_reduce", 0.0, ((double)splits + 1.0) * total * 4.0, s);
  if (splits >= 64 && total <= 65536) {   // many slabs of a small gradient
    int blocks = cdiv(total, 16);
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(blocks), dim3(256), 0, s, partial, dw, splits, Cout, Cout_pad, Kpad, Cin,
                       Cin_real, R, S, accumulate);
    return;
  }
  int blocks = cdiv(total, 64);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3(blocks), dim3(256), 0, s, partial, dw, splits, Cout, Cout_pad, Kpad, Cin,
                     Cin_real, R, S, accumulate);
}

size_t wgrad_workspace_bytes(int dtype, const WgradArgs& a) {
  if (a.in_scale && wgrad_big_applicable(dtype, a)) return wgrad_big_workspace_bytes(dtype, a);
  if (a.in_scale && wgrad_halo_applicable(a)) return wgrad_halo_workspace_bytes(dtype, a);
  if (wgrad_stem_applicable(dtype, a)) return wgrad_stem_workspace_bytes(a);
  if (wgrad_big_applicable(dtype, a)) return wgrad_big_workspace_bytes(dtype, a);
  if (wgrad_halo_applicable(a)) return wgrad_halo_workspace_bytes(dtype, a);
  int bmk, bnn, splits, Kg, Kpad, Cout_pad;
  long pps;
  wg_plan(dtype, a, bmk, bnn, splits, pps, Kg, Kpad, Cout_pad);
  return (size_t)splits * Cout_pad * Kpad * sizeof(float);
}

template <typename T, int BMK, int BNN>
static int wg_launch(const WgKArgs& ka, int splits, hipStream_t s) {
  using Cfg = WgCfg<T, BMK, BNN>;
  auto kern = wgrad_kernel<T, BMK, BNN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(ka.Cout_pad / BMK, ka.Kpad / BNN, splits);
  const double M = (double)ka.a.N * ka.a.Hout * ka.a.Wout;
  const double flops = 2.0 * M * ka.a.Cout * ka.Kg;
  const double bytes = (M * ka.a.dy_ld + (double)ka.a.N * ka.a.Hin * ka.a.Win * (ka.a.C0 / (ka.a.up0 ? 4.0 : 1.0) + ka.a.C1)) * sizeof(T);
  ProfScope ps(sizeof(T) == 2 ? (BMK == 128 ? "wgrad_bf16_128x128" : "wgrad_bf16_64x64") : "wgrad_f32_64x64", flops, bytes, s);
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::SMEM, s, ka);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int launch_wgrad(int dtype, const WgradArgs& a, hipStream_t s) {
  const int ch = dtype == DT_F32 ? 4 : 8;
  if ((a.C0 % ch) || (a.C1 % ch) || (a.dy_ld % ch)) return -2;
  if (a.dbias) return wgrad_dbias_fusable(dtype, a) ? launch_wgrad_halo(dtype, a, s) : -6;
  if (a.fuse_y && !wgrad_stem_applicable(dtype, a)) return -6;
  if (a.in_scale) {   // lazy input: the two halo-staging kernels apply it
    if (wgrad_big_applicable(dtype, a)) return launch_wgrad_big(dtype, a, s);
    return wgrad_halo_applicable(a) ? launch_wgrad_halo(dtype, a, s) : -6;
  }
  if (wgrad_stem_applicable(dtype, a)) return launch_wgrad_stem(a, s);        // 7x7 stride-2 stem
  if (wgrad_big_applicable(dtype, a)) return launch_wgrad_big(dtype, a, s);  // MFMA-bound 3x3 s1 layers, >= 64 channels
  if (wgrad_halo_applicable(a)) return launch_wgrad_halo(dtype, a, s);  // HBM-bound small-channel 3x3 layers
  WgKArgs ka;
  ka.a = a;
  int bmk, bnn, splits;
  wg_plan(dtype, a, bmk, bnn, splits, ka.pix_per_split, ka.Kg, ka.Kpad, ka.Cout_pad);
  int rc;
  if (dtype == DT_F32) rc = wg_launch<float, 64, 64>(ka, splits, s);
  else if (bmk == 128) rc = wg_launch<bf16_t, 128, 128>(ka, splits, s);
  else rc = wg_launch<bf16_t, 64, 64>(ka, splits, s);
  if (rc) return rc;
  launch_wgrad_reduce(a.partial, a.dw, splits, a.Cout, ka.Cout_pad, ka.Kpad, a.C0 + a.C1, a.Cin_real, a.R, a.S, a.accumulate, s);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

}  // namespace flair
