// "Halo GEMM": 3x3 / stride-1 / pad-1 convolution (and its data gradient) for the MFMA-bound layers
// (>= 64 channels: ResNet34 layer1-4 and decoder blocks 0-2; SURVEY.md §8a-3 = 57 % + 22 % of the FLOPs).
//
// The plain implicit GEMM re-gathers the im2col operand once per tap, i.e. 9x through the vector L1
// (64 B/clk/CU), which — not the MFMA pipe — bounds it at ~64 FLOP per L1 byte.  Here a workgroup owns a
// TW x TH output tile and, per 128-byte channel chunk, stages the (TH+2)x(TW+2) input halo ONCE in LDS
// (prefetched in registers across the 9 tap steps of the previous chunk); all nine taps read their A
// fragments from that halo with shifted addresses.  Only the weight tile [BN][128 B] per (tap, chunk) is
// streamed (register-staged, double-buffered LDS).  Every wave owns a 64x64 accumulator tile.
// Numerics, operand packing and epilogue (LDS transpose -> 16-byte NHWC stores, BN partial sums) are those
// of conv_igemm.  Tile / buffering variants are template parameters; launch_conv_hg picks per layer.
#include <stdlib.h>

#include "common.h"
#include "prof.h"
#include "tile_store.h"
#include "tune.h"

namespace flair {
namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct GMma;
template <> struct GMma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct GMma<float> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

__device__ __forceinline__ int sw_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// TW x TH output pixels per workgroup (128 or 256), BN output channels (64 or 128), HB halo buffers (1 or 2)
// TMv: 16-pixel row segments per wave.  4 = a 64 x 64 accumulator tile per wave (what every launch uses).  8 (128 x 64, a
// quarter fewer LDS fragment bytes per MFMA) compiles, but at two workgroups per CU its 128 accumulator registers push
// the kernel past the 256-VGPR cap: 500 bytes of scratch per lane, 15.3 vs 13.9 ms per step — not instantiated.
template <typename T, int TW, int TH, int BN, int HB, int TMv = 4>
struct HgCfg {
  static constexpr int CH = Elem<T>::CH;
  static constexpr int CK = 8 * CH;                  // channels per 128-byte chunk (64 bf16 / 32 f32)
  static constexpr int TPIX = TW * TH;
  static constexpr int HW_ = TW + 2, HH = TH + 2, HPIX = HW_ * HH;
  static constexpr int WMN = TPIX / (16 * TMv), WNN = BN / 64;
  static constexpr int NWAVE = WMN * WNN;
  static constexpr int NT = 64 * NWAVE;
  static constexpr int HALO = HPIX * 128;
  static constexpr int BTILE = BN * 128;
  static constexpr int CLD = BN * (int)sizeof(T) + 16;
  static constexpr int CT = TPIX * CLD;
  static constexpr int STAGES = HB * HALO + 2 * BTILE;
  static constexpr int MAIN = STAGES > CT ? STAGES : CT;
  static constexpr int STATS = WMN * BN * 2 * 4;
  static constexpr int SMEM = MAIN + STATS;
  static constexpr int HITEMS = (HPIX * 8 + NT - 1) / NT;
  static constexpr int BITEMS = (BN * 8 + NT - 1) / NT;
};

// BNR: instantiate the fused BatchNorm-backward reduction of the epilogue (ConvArgs::bnr_*); kept out of the plain
// variant so that its register allocation does not pay for it
template <typename T, int TW, int TH, int BN, int HB, bool BNR, int TMv = 4>
__global__ __launch_bounds__(TW * TH * BN / (16 * TMv), 2) void conv3x3_hg_kernel(const ConvArgs a) {  // 64 * (TPIX/(16 TM)) * (BN/64) threads
  using Cfg = HgCfg<T, TW, TH, BN, HB, TMv>;
  constexpr int CH = Cfg::CH, CK = Cfg::CK, HW_ = Cfg::HW_, HPIX = Cfg::HPIX, NT = Cfg::NT, TPIX = Cfg::TPIX;
  constexpr int TM = TMv, TN = 4, MTX = TW / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* halo0 = smem;
  unsigned char* bt0 = smem + HB * Cfg::HALO;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int wm = wave % Cfg::WMN, wn = wave / Cfg::WMN;
  const int H = a.Hout, W = a.Wout;
  const int tiles_x = W / TW, tiles_y = H / TH;
  // Work item w = tile * NB + (output-channel block): the NB column blocks of a tile are neighbours in w, and each of the
  // 8 XCDs (which take consecutive workgroup ids round-robin) gets one contiguous range of w — so the workgroups that
  // share an input tile, and the tiles that share halo rows, run on the same L2 close in time.
  const int NB = a.Cout / BN;
  const int nwork = gridDim.x;
  int w = blockIdx.x;
  if (a.xcd_remap) {
    const int q = nwork >> 3, r = nwork & 7, xcd = w & 7;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
  }
  const int tile = w / NB;
  const int ntiles = nwork / NB;
  const int n = tile / (tiles_x * tiles_y);
  const int trem = tile - n * tiles_x * tiles_y;
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
  const int n0 = (w - tile * NB) * BN;
  const int Cin = a.C0 + a.C1;
  const T* __restrict__ src0 = (const T*)a.src0;
  const T* __restrict__ src1 = (const T*)a.src1;
  const T* __restrict__ wp = (const T*)a.w;
  const int Hs0 = a.up0 ? (H >> 1) : H, Ws0 = a.up0 ? (W >> 1) : W;
  const int nchunks = Cin / CK;

  // ---- halo chunk: global -> registers (issued early), registers -> LDS (one chunk later)
  u32x4 hreg[Cfg::HITEMS];
  unsigned hbits = 0;   // bit k: item k is inside the image (else it is stored as zeros)
  auto halo_load = [&](int chunk) {
    const int cbase = chunk * CK;
    const bool use0 = cbase < a.C0 || chunk >= nchunks;  // past-the-end prefetch: masked, but must read a valid base
    const T* __restrict__ base = use0 ? src0 : src1;
    const int Hs = use0 ? Hs0 : H, Ws = use0 ? Ws0 : W, Cs = use0 ? a.C0 : a.C1;
    const int sh = (use0 && a.up0) ? 1 : 0;
    const int coff = use0 ? cbase : cbase - a.C0;
    unsigned hb2 = 0;
#pragma unroll
    for (int k = 0; k < Cfg::HITEMS; ++k) {
      const int it = t + NT * k;
      const int hp = it >> 3, ch = it & 7;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = (it < HPIX * 8) && (chunk < nchunks) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((n * Hs + (iy >> sh)) * Ws + (ix >> sh)) * Cs + coff + ch * CH) : 0u;
      hreg[k] = *reinterpret_cast<const u32x4*>(base + off);
      hb2 |= (ok ? 1u : 0u) << k;
    }
    hbits = hb2;
  };
  auto halo_store = [&](int buf) {
    unsigned char* hb = halo0 + buf * Cfg::HALO;
#pragma unroll
    for (int k = 0; k < Cfg::HITEMS; ++k) {
      const int it = t + NT * k;
      if (it < HPIX * 8) {
        const int hp = it >> 3, hx = hp % HW_;
        *reinterpret_cast<u32x4*>(hb + hp * 128 + (((it & 7) ^ (hx & 7)) << 4)) = hreg[k] & (0u - ((hbits >> k) & 1u));
      }
    }
  };
  // ---- weight tile of (chunk, tap): [BN][128 B]
  u32x4 breg[Cfg::BITEMS];
  unsigned bgo[Cfg::BITEMS];   // per-thread element offset of its weight chunk(s) at (tap 0, chunk 0)
#pragma unroll
  for (int k = 0; k < Cfg::BITEMS; ++k) {
    const int it = t + NT * k;
    const int itc = it < BN * 8 ? it : 0;
    bgo[k] = (unsigned)((n0 + (itc >> 3)) * a.Kpad + (itc & 7) * CH);
  }
  auto b_load = [&](int chunk, int tap) {   // wave-uniform (chunk, tap): one scalar offset for all lanes
    const int cl = chunk < nchunks ? chunk : nchunks - 1;
    const unsigned so = (unsigned)(tap * Cin + cl * CK);
#pragma unroll
    for (int k = 0; k < Cfg::BITEMS; ++k) breg[k] = *reinterpret_cast<const u32x4*>(wp + (bgo[k] + so));
  };
  auto b_store = [&](int buf) {
    unsigned char* bb = bt0 + buf * Cfg::BTILE;
#pragma unroll
    for (int k = 0; k < Cfg::BITEMS; ++k) {
      const int it = t + NT * k;
      if ((BN * 8) % NT == 0 || it < BN * 8) *reinterpret_cast<u32x4*>(bb + sw_off(it >> 3, it & 7)) = breg[k];
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Fragment addresses: the halo swizzle depends on the column only, and (px & 7) does not see the multiple-of-16 column
  // offset of an M tile, so three lane-dependent bases (one per column shift sx) serve every M tile, tap row and chunk
  // half through compile-time offsets: tap (r, sx), M tile g -> abase[sx] + ((g / MTX + r) * HW_ + (g % MTX) * 16) * 128,
  // the h = 1 half is the same address with bit 6 flipped (chunk + 4).
  unsigned abase[3], bfo[TN];
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) abase[sx] = (unsigned)((lr + sx) * 128 + ((lq ^ ((lr + sx) & 7)) << 4));
#pragma unroll
  for (int j = 0; j < TN; ++j) bfo[j] = (unsigned)sw_off(wn * 64 + j * 16 + lr, lq);

  static_assert(TM % MTX == 0, "a wave's M tiles start at a tile-row boundary");
  const unsigned wbase = (unsigned)((wm * TM / MTX) * HW_ * 128);   // first tile row of this wave

  halo_load(0);
  b_load(0, 0);
  halo_store(0);
  b_store(0);
  halo_load(1);   // chunk 1 (masked to nothing when there is a single chunk) rides in registers through chunk 0
  __syncthreads();

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const unsigned char* hb = halo0 + (HB == 2 ? (chunk & 1) : 0) * Cfg::HALO;
    const int par = chunk & 1;   // 9 taps per chunk: the weight-stage parity flips from chunk to chunk
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      const int tap = r * 3 + sx;
      const int cur = (par + tap) & 1;
      if (tap < 8) b_load(chunk, tap + 1); else b_load(chunk + 1, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch at the top of the step: hipcc otherwise sinks the
                                          // loads next to their ds_write and exposes the full L2 latency
      const unsigned char* bb = bt0 + cur * Cfg::BTILE;
      const unsigned char* hbr = hb + r * (HW_ * 128);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 bfr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(bb + (bfo[j] ^ (h << 6)));
#pragma unroll
        for (int i0 = 0; i0 < TM; i0 += 4) {   // four M tiles at a time: 16 A-fragment registers live, not 4 * TM
          u32x4 af[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int g = wm * TM + i0 + i;   // wm enters as a run-time multiple of the per-wave row stride below
            (void)g;
            af[i] = *reinterpret_cast<const u32x4*>(hbr + wbase + (((i0 + i) / MTX) * HW_ + ((i0 + i) % MTX) * 16) * 128 +
                                                    (abase[sx] ^ (h << 6)));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) GMma<T>::run(af[i], bfr[j], acc[i0 + i][j]);
        }
      }
      b_store(cur ^ 1);
      if (tap == 8) {
        if constexpr (HB == 2) {   // next chunk's halo: landed long ago, becomes visible with this barrier
          halo_store((chunk + 1) & 1);
        } else {                   // single halo buffer: every wave must be done reading it first
          __syncthreads();
          halo_store(0);
        }
        halo_load(chunk + 2);
      }
      __syncthreads();
    }
    }
  }

  // ------------------------------------------------------------------ epilogue (as conv_igemm)
  unsigned char* ct = smem;
  float* st = reinterpret_cast<float*>(smem + Cfg::MAIN);
  float s1[TN], s2[TN];
#pragma unroll
  for (int q = 0; q < TN; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
#pragma unroll
  for (int q = 0; q < TN; ++q) {
    const int col = wn * 64 + q * 16 + lr;
    const bool cin_ok = (n0 + col) < a.Cout;
    const float osc = (a.oscale && cin_ok) ? a.oscale[n0 + col] : 1.f;
    const float bias = ((a.bias && cin_ok) ? a.bias[n0 + col] : 0.f) + ((a.oshift && cin_ok) ? a.oshift[n0 + col] : 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int g = wm * TM + i;
        const int row = (g / MTX) * TW + (g % MTX) * 16 + lq * 4 + rr;  // tile-local pixel
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
        st[(wm * BN + wn * 64 + q * 16 + lr) * 2 + 0] = s1[q];
        st[(wm * BN + wn * 64 + q * 16 + lr) * 2 + 1] = s2[q];
      }
    }
  }
  __syncthreads();
  if (a.stats && t < BN && (n0 + t) < a.Cout) {
    float x1 = 0.f, x2 = 0.f;
#pragma unroll
    for (int w = 0; w < Cfg::WMN; ++w) { x1 += st[(w * BN + t) * 2]; x2 += st[(w * BN + t) * 2 + 1]; }
    a.stats[(long)(n0 + t) * ntiles + tile] = x1;
    a.stats[((long)a.Cout + n0 + t) * ntiles + tile] = x2;
  }
  store_tile<T, TW, TPIX, BN, NT, Cfg::CLD, BNR>(a, ct, n, y0, x0, n0, t, tile, ntiles);
}


// ------------------------------------------------------------------------------------------------------------------
// Round-2 structure: the weight tile never passes through registers.  Each wave streams its share of the [BN][128 B]
// tile of (chunk, tap) with LDS-DMA (global_load_lds_dwordx4: 1 KiB = 8 rows per wave-instruction, the XOR swizzle applied
// on the per-lane SOURCE address, the LDS image stays lane-linear — cdna_hip_programming.md §5.4 rule 21) into a ring of
// NS slots, NS - 1 taps ahead; a counted s_waitcnt vmcnt(N) + raw s_barrier per tap leaves the younger tiles in
// flight across the barrier.  The 32 KB per tap and CU of ds_write_b128 (half of the LDS cycles of the register-staged
// kernel above) and the breg[] registers are gone.  The halo still goes through registers once per chunk: that is where
// the lazy BatchNorm + ReLU of the producing unit is applied (LAZY), so the normalised activation is never written.
// Accumulation order per output element is unchanged (chunk, tap, half), so results equal the kernel above bit for bit.
void* g_debug_buffer = nullptr;   // flair_debug_buffer(): stamps of the next halo-GEMM launches (diagnostics)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T, int TW, int TH, int BN, int HB, int NS>
struct HgdCfg {
  static constexpr int CH = Elem<T>::CH;
  static constexpr int CK = 8 * CH;
  static constexpr int TPIX = TW * TH;
  static constexpr int HW_ = TW + 2, HH = TH + 2, HPIX = HW_ * HH;
  static constexpr int WMN = TPIX / 64, WNN = BN >= 64 ? BN / 64 : 1;   // BN = 32: one wave column of two 16-wide blocks
  static constexpr int TN = BN / 16 / WNN;
  static constexpr int NWAVE = WMN * WNN;
  static constexpr int NT = 64 * NWAVE;
  static constexpr int HALO = HPIX * 128;
  static constexpr int BTILE = BN * 128;
  static constexpr int CLD = BN * (int)sizeof(T) + 16;
  static constexpr int CT = TPIX * CLD;
  static constexpr int STAGES = HB * HALO + NS * BTILE;
  static constexpr int MAIN = STAGES > CT ? STAGES : CT;
  static constexpr int STATS = WMN * BN * 2 * 4;
  static constexpr int SMEM = MAIN + STATS;
  static constexpr int HITEMS = (HPIX * 8 + NT - 1) / NT;
  static constexpr int DPW = BTILE / (1024 * NWAVE);   // LDS-DMA instructions per wave and weight tile
  static constexpr int D = NS - 1;                      // prefetch distance in taps
  static constexpr int APC = 512;                       // XF_APPLY: most input channels (k1 | k2 | k3 table in LDS behind SMEM)
  static constexpr int APTAB = 3 * APC * 4;
  static_assert(BTILE % (1024 * NWAVE) == 0, "every wave streams whole 1 KiB pieces");
  static_assert(D >= 1 && D <= 3, "ring depth");
};

// s_waitcnt vmcnt(n) lgkmcnt(0), n a run-time (wave-uniform) value out of a small set
__device__ __forceinline__ void wait_vm_lgkm0(int n) {
#define FLAIR_VM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ") lgkmcnt(0)" ::: "memory"); break;
  switch (n) {
    FLAIR_VM_CASE(1) FLAIR_VM_CASE(2) FLAIR_VM_CASE(3) FLAIR_VM_CASE(4) FLAIR_VM_CASE(5) FLAIR_VM_CASE(6) FLAIR_VM_CASE(7)
    FLAIR_VM_CASE(8) FLAIR_VM_CASE(9) FLAIR_VM_CASE(10) FLAIR_VM_CASE(11) FLAIR_VM_CASE(12) FLAIR_VM_CASE(13)
    FLAIR_VM_CASE(14) FLAIR_VM_CASE(15) FLAIR_VM_CASE(16) FLAIR_VM_CASE(17) FLAIR_VM_CASE(18) FLAIR_VM_CASE(19)
    FLAIR_VM_CASE(20) FLAIR_VM_CASE(21) FLAIR_VM_CASE(22) FLAIR_VM_CASE(23) FLAIR_VM_CASE(24) FLAIR_VM_CASE(25)
    FLAIR_VM_CASE(26) FLAIR_VM_CASE(27) FLAIR_VM_CASE(28) FLAIR_VM_CASE(29) FLAIR_VM_CASE(30) FLAIR_VM_CASE(31)
    FLAIR_VM_CASE(32) FLAIR_VM_CASE(33) FLAIR_VM_CASE(34) FLAIR_VM_CASE(35) FLAIR_VM_CASE(36) FLAIR_VM_CASE(37)
    FLAIR_VM_CASE(38) FLAIR_VM_CASE(39) FLAIR_VM_CASE(40) FLAIR_VM_CASE(41) FLAIR_VM_CASE(42) FLAIR_VM_CASE(43)
    FLAIR_VM_CASE(44) FLAIR_VM_CASE(45) FLAIR_VM_CASE(46) FLAIR_VM_CASE(47) FLAIR_VM_CASE(48)
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
#undef FLAIR_VM_CASE
}

// XF: transform applied to the input while the halo is staged.  XF_LAZY: BatchNorm + ReLU of the producing unit
// (ConvArgs::in_scale).  XF_APPLY: the BatchNorm-backward apply dy = k1*dz + k2*y + k3 of the unit this data gradient belongs to
// (ConvArgs::ap_*): two tensors are read per chunk, and the workgroups of output-channel block 0 write the interior of the
// transformed chunk back to HBM for the weight-gradient kernel — bn_bwd_apply (45 launches, 155 MB each) is gone for these units.
enum { XF_NONE = 0, XF_LAZY = 1, XF_APPLY = 2 };

template <typename T, int TW, int TH, int BN, int HB, int NS, bool BNR, int XF>
__global__ __launch_bounds__(TW * TH * (BN >= 64 ? BN : 64) / 64, 2) void conv3x3_hgd_kernel(const ConvArgs a) {
  constexpr bool LAZY = XF == XF_LAZY, APPLY = XF == XF_APPLY;
  static_assert(!APPLY || HB == 1, "the apply flavour is written for the single halo buffer");
  using Cfg = HgdCfg<T, TW, TH, BN, HB, NS>;
  constexpr int CH = Cfg::CH, CK = Cfg::CK, HW_ = Cfg::HW_, HPIX = Cfg::HPIX, NT = Cfg::NT, TPIX = Cfg::TPIX;
  constexpr int TM = 4, TN = Cfg::TN, MTX = TW / 16, DPW = Cfg::DPW, D = Cfg::D, HITEMS = Cfg::HITEMS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* halo0 = smem;
  unsigned char* bt0 = smem + HB * Cfg::HALO;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int wm = wave % Cfg::WMN, wn = wave / Cfg::WMN;
  const int H = a.Hout, W = a.Wout;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const int NB = a.Cout / BN;
  const int nwork = gridDim.x;
  int w = blockIdx.x;
  if (a.xcd_remap) {   // XCD-contiguous work order (see the kernel above)
    const int q = nwork >> 3, r = nwork & 7, xcd = w & 7;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
  }
  const int tile = w / NB;
  const int ntiles = nwork / NB;
  const int n = tile / (tiles_x * tiles_y);
  const int trem = tile - n * tiles_x * tiles_y;
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
  const int n0 = (w - tile * NB) * BN;
  const int Cin = a.C0 + a.C1;
  const T* __restrict__ src0 = (const T*)a.src0;
  const T* __restrict__ src1 = (const T*)a.src1;
  const unsigned char* __restrict__ wp = (const unsigned char*)a.w;
  const int Hs0 = a.up0 ? (H >> 1) : H, Ws0 = a.up0 ? (W >> 1) : W;
  const int nchunks = Cin / CK;

  // ---- halo chunk: global -> registers (one chunk ahead), registers -> [lazy BN + ReLU] -> LDS
  u32x4 hreg[HITEMS];
  u32x4 yreg[APPLY ? HITEMS : 1];   // XF_APPLY: the pre-BN tensor at the same places
  unsigned hbits = 0;
  float lsc[CH], lsh[CH];
  bool hlazy = false;
  // XF_APPLY: does this workgroup write the transformed chunks back (output-channel block 0, and somebody wants them)?
  const bool ap_writer = APPLY && a.ap_dy != nullptr && n0 == 0;
  int ap_coff = 0;                  // channel offset of the chunk held in hreg / yreg
  auto halo_load = [&](int chunk) {
    const int cbase = chunk * CK;
    const bool use0 = cbase < a.C0;
    const T* __restrict__ base = use0 ? src0 : src1;
    const int Hs = use0 ? Hs0 : H, Ws = use0 ? Ws0 : W, Cs = use0 ? a.C0 : a.C1;
    const int sh = (use0 && a.up0) ? 1 : 0;
    const int coff = use0 ? cbase : cbase - a.C0;
    unsigned hb2 = 0;
#pragma unroll
    for (int k = 0; k < HITEMS; ++k) {
      const int it = t + NT * k;
      const int hp = it >> 3, ch = it & 7;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = (it < HPIX * 8) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
      const unsigned off = ok ? (unsigned)(((n * Hs + (iy >> sh)) * Ws + (ix >> sh)) * Cs + coff + ch * CH) : 0u;
      hreg[k] = *reinterpret_cast<const u32x4*>(base + off);
      if constexpr (APPLY) yreg[k] = *reinterpret_cast<const u32x4*>((const T*)a.ap_y + off);
      hb2 |= (ok ? 1u : 0u) << k;
    }
    hbits = hb2;
    if constexpr (APPLY) ap_coff = coff;
    if constexpr (LAZY) {
      hlazy = use0 && a.in_scale != nullptr;
      const int c0 = hlazy ? coff + (t & 7) * CH : 0;
      const float* __restrict__ ps = hlazy ? a.in_scale : (const float*)a.w;   // any valid address when unused
      const float* __restrict__ ph = hlazy ? a.in_shift : (const float*)a.w;
#pragma unroll
      for (int e = 0; e < CH; e += 4) {
        const float4 v1 = *reinterpret_cast<const float4*>(ps + c0 + e), v2 = *reinterpret_cast<const float4*>(ph + c0 + e);
        lsc[e] = v1.x; lsc[e + 1] = v1.y; lsc[e + 2] = v1.z; lsc[e + 3] = v1.w;
        lsh[e] = v2.x; lsh[e + 1] = v2.y; lsh[e + 2] = v2.z; lsh[e + 3] = v2.w;
      }
    }
  };
  constexpr int HLOADS = (APPLY ? 2 : 1) * HITEMS + (LAZY ? 2 * (CH / 4) : 0);   // vector-memory instructions of one halo_load
  // XF_APPLY: stores a writer workgroup issues per halo_store (one per item, masked lanes go out of range: every wave issues
  // every instruction, which is what the counted s_waitcnt vmcnt(N) below relies on)
  const int nst = ap_writer ? HITEMS : 0;
  // XF_APPLY: the transform runs IN PLACE on the halo registers, one item per tap step of the chunk before (xf_step), so that
  // its ~50 vector instructions per item sit under that step's LDS fragment latency instead of between the two barriers that
  // fence the halo buffer; the coefficients come from a table in LDS (filled once per workgroup), held in registers from
  // tap 2 to tap 8 only.
  float k1[APPLY ? CH : 1], k2[APPLY ? CH : 1], k3[APPLY ? CH : 1];
  const float* aptab = reinterpret_cast<const float*>(smem + Cfg::SMEM);
  auto xf_k_load = [&]() {
    if constexpr (APPLY) {
      const float* kc = aptab + ap_coff + (t & 7) * CH;
#pragma unroll
      for (int e = 0; e < CH; e += 4) {
        const float4 v1 = *reinterpret_cast<const float4*>(kc + e), v2 = *reinterpret_cast<const float4*>(kc + Cfg::APC + e),
                     v3 = *reinterpret_cast<const float4*>(kc + 2 * Cfg::APC + e);
        k1[e] = v1.x; k1[e + 1] = v1.y; k1[e + 2] = v1.z; k1[e + 3] = v1.w;
        k2[e] = v2.x; k2[e + 1] = v2.y; k2[e + 2] = v2.z; k2[e + 3] = v2.w;
        k3[e] = v3.x; k3[e + 1] = v3.y; k3[e + 2] = v3.z; k3[e + 3] = v3.w;
      }
    }
  };
  auto xf_item = [&](int k) {
    if constexpr (APPLY) {
      float d[CH], yy[CH];
      chunk_to_f<T>(__builtin_bit_cast(uint4, hreg[k]), d);
      chunk_to_f<T>(__builtin_bit_cast(uint4, yreg[k]), yy);
#pragma unroll
      for (int e = 0; e < CH; ++e) d[e] = fmaf(k1[e], d[e], fmaf(k2[e], yy[e], k3[e]));   // bn_bwd_apply_kernel's expression
      hreg[k] = __builtin_bit_cast(u32x4, f_to_chunk<T>(d)) & (0u - ((hbits >> k) & 1u));    // outside the image: zero padding
    }
  };
  // tap step `tap` of the chunk in front of the one the registers hold: coefficients at tap 2, the items spread over taps 3 .. 8
  auto xf_step = [&](int tap) {
    if constexpr (APPLY) {
      if (tap == 2) xf_k_load();
#pragma unroll
      for (int k = 0; k < HITEMS; ++k)
        if (3 + (k * 6) / HITEMS == tap) xf_item(k);
    }
  };
  auto halo_store = [&](int buf) {
    unsigned char* hb = halo0 + buf * Cfg::HALO;
    if constexpr (APPLY) {   // hreg holds the transformed chunk
      const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
          a.ap_dy, 0, ap_writer ? (int)((long)a.N * H * W * a.C0 * (long)sizeof(T)) : 0, 0x00020000);
#pragma unroll
      for (int k = 0; k < HITEMS; ++k) {
        const int it = t + NT * k;
        const int hp = it >> 3, hy = hp / HW_, hx = hp - hy * HW_;
        if (it < HPIX * 8) *reinterpret_cast<u32x4*>(hb + hp * 128 + (((it & 7) ^ (hx & 7)) << 4)) = hreg[k];
        if (ap_writer) {   // workgroup-uniform
          const bool inner = it < HPIX * 8 && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
          const unsigned off = (unsigned)((((n * H + y0 - 1 + hy) * W + x0 - 1 + hx) * a.C0 + ap_coff + (it & 7) * CH) * (int)sizeof(T));
          __builtin_amdgcn_raw_buffer_store_b128(hreg[k], drs, inner ? off : 0x80000000u, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < HITEMS; ++k) {
      const int it = t + NT * k;
      if (it < HPIX * 8) {
        const int hp = it >> 3, hx = hp % HW_;
        u32x4 v = hreg[k];
        if constexpr (LAZY) {
          if (hlazy) v = chunk_bn_relu<T, u32x4>(v, lsc, lsh);
        }
        *reinterpret_cast<u32x4*>(hb + hp * 128 + (((it & 7) ^ (hx & 7)) << 4)) = v & (0u - ((hbits >> k) & 1u));
        if constexpr (LAZY) __builtin_amdgcn_sched_barrier(0);   // one item at a time: interleaved, the unpacked floats of
                                                                  // several items push the 64-wide kernels into scratch
      }
    }
  };
  // ---- weight tile of (chunk, tap): LDS-DMA, row r of the tile = cout n0 + r, physical 16-byte slot p of the row holds
  // logical chunk p ^ (r & 7); one wave-instruction = rows 8q .. 8q+7, lane l -> row 8q + (l >> 3), slot l & 7
  const unsigned bvo = (unsigned)(((n0 + wave * (DPW * 8) + (lane >> 3)) * a.Kpad + (((lane & 7) ^ (lane >> 3)) * CH)) * (int)sizeof(T));
  const unsigned brow = (unsigned)(8 * a.Kpad * (int)sizeof(T));
  auto b_dma = [&](int chunk, int tap, int slot) {
    const unsigned char* gb = wp + (size_t)(unsigned)((tap * Cin + chunk * CK) * (int)sizeof(T));   // wave-uniform
    unsigned char* lb = bt0 + slot * Cfg::BTILE + wave * (DPW * 1024);
#pragma unroll
    for (int j = 0; j < DPW; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(gb + (bvo + j * brow)), (lptr_t)(lb + j * 1024), 16, 0, 0);
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  unsigned abase[3], bfo[TN];
#pragma unroll
  for (int sx = 0; sx < 3; ++sx) abase[sx] = (unsigned)((lr + sx) * 128 + ((lq ^ ((lr + sx) & 7)) << 4));
#pragma unroll
  for (int j = 0; j < TN; ++j) bfo[j] = (unsigned)sw_off(wn * 64 + j * 16 + lr, lq);
  static_assert(TM % MTX == 0, "a wave's M tiles start at a tile-row boundary");
  const unsigned wbase = (unsigned)((wm * TM / MTX) * HW_ * 128);

  const int nsteps = 9 * nchunks;
  // Phase stamps (scripts/stamp_hg.py) exist in the diagnostic build only (FLAIR_STAMPS=1 python flair-1_amd/build.py):
  // in the shipped kernel no stamp executes (cdna_hip_programming.md §7, In-kernel stamps).
  auto stamp = [&](int i) {
#ifdef FLAIR_HG_STAMPS
    if (a.dbg && t == 0) a.dbg[(size_t)blockIdx.x * 8 + i] = i == 5 || i == 6 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
#else
    (void)i;
#endif
  };
  stamp(0);
  stamp(5);
  // Fragment reads and MFMAs are software-pipelined by hand in two register sets: set A holds half 0 of a tap (read
  // right after the barrier that publishes its weight tile), set B half 1 (read while half 0 multiplies).
  u32x4 afA[TM], bfA[TN], afB[TM], bfB[TN];
  auto rd = [&](u32x4 (&af)[TM], u32x4 (&bf)[TN], const unsigned char* hbuf, int slot, int tap, int h) {
    const unsigned char* bb = bt0 + slot * Cfg::BTILE;
    const unsigned char* hbr = hbuf + (tap / 3) * (HW_ * 128);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(bb + (bfo[j] ^ (h << 6)));
#pragma unroll
    for (int i = 0; i < TM; ++i)
      af[i] = *reinterpret_cast<const u32x4*>(hbr + wbase + ((i / MTX) * HW_ + (i % MTX) * 16) * 128 + (abase[tap % 3] ^ (h << 6)));
  };
  // The WEIGHT fragment is the MFMA's A operand: the accumulator tile is [cout][pixel], a lane holds FOUR CONSECUTIVE
  // OUTPUT CHANNELS of one pixel (cout 4 * lq + reg, pixel lr) — 8 contiguous bytes of the NHWC row, so the epilogue
  // stages a 16x16 block with one ds_write_b64 per lane instead of four ds_write_b16.  Same products, same k order.
  auto mm = [&](const u32x4 (&af)[TM], const u32x4 (&bf)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) GMma<T>::run(bf[j], af[i], acc[i][j]);
  };
  // ---- prologue: halo 0, weight tiles of steps 0 .. D-1, halo 1 into registers
  halo_load(0);
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nsteps) b_dma(d / 9, d % 9, d % NS);
  if constexpr (APPLY) {   // coefficient table -> LDS, then the first chunk's transform (nothing to hide it behind yet)
    float* tab = reinterpret_cast<float*>(smem + Cfg::SMEM);
    for (int i = t; i < 3 * (a.C0 >> 2); i += NT) {
      const int j = i / (a.C0 >> 2), c4 = i - j * (a.C0 >> 2);
      *reinterpret_cast<float4*>(tab + j * Cfg::APC + c4 * 4) = *reinterpret_cast<const float4*>(a.ap_coef + j * a.C0 + c4 * 4);
    }
    __syncthreads();
    xf_k_load();
#pragma unroll
    for (int k = 0; k < HITEMS; ++k) xf_item(k);
  }
  halo_store(0);
  const bool pre1 = nchunks > 1;
  if (pre1) halo_load(1);
  // younger than the tile of step 0: the D-1 other tiles and (when issued) the loads of halo 1
  wait_vm_lgkm0((D - 1) * DPW + nst + (pre1 ? HLOADS : 0));
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  stamp(1);
  // fp32 runs the fragment reads and the (16x slower, MFMA-bound) multiplies back to back in ONE register set: the
  // second set only costs registers there
  // (so does the lazy-input flavour of the 64-wide tiles: 11 halo registers per lane plus the BN coefficients)
  // (and the apply flavour: two tensors' halo registers; with the second fragment set the allocator spills 200 registers)
  constexpr bool PIPE = sizeof(T) == 2 && !(LAZY && BN == 64) && !APPLY;
  if constexpr (PIPE) rd(afA, bfA, halo0, 0, 0, 0);

  int sb = 0;   // (9 * chunk) % NS
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const unsigned char* hb = halo0 + (HB == 2 ? (chunk & 1) : 0) * Cfg::HALO;
    const unsigned char* hbn = halo0 + (HB == 2 ? ((chunk + 1) & 1) : 0) * Cfg::HALO;
    const bool last = chunk == nchunks - 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = (9 % NS == 0) ? (tap % NS) : ((sb + tap) % NS);
      const int nxt = (9 % NS == 0) ? ((tap + 1) % NS) : ((sb + tap + 1) % NS);
      // stream the tile of step s + D into the slot the step before this one has just released
      {
        const int tp = tap + D;
        const int pc = tp >= 9 ? chunk + 1 : chunk, pt = tp >= 9 ? tp - 9 : tp;
        const int pslot = (9 % NS == 0) ? (tp % NS) : ((sb + tp) % NS);
        if (pc < nchunks) b_dma(pc, pt, pslot);
      }
      if constexpr (PIPE) {
        rd(afB, bfB, hb, cur, tap, 1);
        mm(afA, bfA);
      } else {
        rd(afA, bfA, hb, cur, tap, 0);
        if constexpr (APPLY) {
          if (!last) xf_step(tap);   // the registers hold chunk + 1: its transform rides under this step's LDS latency
        }
        mm(afA, bfA);
        rd(afA, bfA, hb, cur, tap, 1);
        mm(afA, bfA);
      }
      // ---- end of step: wait for the tile of step s + 1 (the younger ones stay in flight) and meet the other waves
      bool halo_younger = false;   // loads of a halo chunk issued after the tile of step s + 1
      int st_younger = 0;          // XF_APPLY: write-back stores of a halo_store issued after the tile of step s + 1
      if (tap == 8) {
        if constexpr (HB == 2) {
          if (!last) {   // the other halo buffer: published by this step's barrier
            halo_store((chunk + 1) & 1);
            if (chunk + 2 < nchunks) { halo_load(chunk + 2); halo_younger = true; }
          }
        }
      } else if (tap <= D - 2) {
        halo_younger = chunk + 1 < nchunks;   // issued at tap 8 of the previous chunk (or in the prologue)
        st_younger = nst;                     // the halo_store of THIS chunk, issued there too, in front of those loads
      }
      const int left = last ? 8 - tap : 9;            // steps after this one (capped)
      const int groups = left < 1 ? 0 : ((left < D ? left : D) - 1);
      wait_vm_lgkm0(left < 1 ? 0 : groups * DPW + st_younger + (halo_younger ? HLOADS : 0));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (tap < 8) {
        if constexpr (PIPE) {
          rd(afA, bfA, hb, nxt, tap + 1, 0);
          mm(afB, bfB);
        }
      } else if (!last) {
        if constexpr (HB == 1) {   // single halo buffer: every wave has passed the barrier, nobody reads it any more
          halo_store(0);
          if (chunk + 2 < nchunks) halo_load(chunk + 2);
          if constexpr (PIPE) mm(afB, bfB);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (PIPE) rd(afA, bfA, hbn, nxt, 0, 0);
        } else if constexpr (PIPE) {
          rd(afA, bfA, hbn, nxt, 0, 0);
          mm(afB, bfB);
        }
      } else if constexpr (PIPE) {
        mm(afB, bfB);
      }
    }
    if (9 % NS != 0) sb = (sb + 9) % NS;
  }
  stamp(2);

  // ------------------------------------------------------------------ epilogue
  // Accumulators -> T -> LDS C tile ([pixel][cout]); every wave stages its own 64 x 64 block.  The BatchNorm batch
  // statistics of the ROUNDED outputs (sum and sum of squares per output channel over the wave's 64 pixels) come from the
  // matrix pipe: the wave re-reads its block as MFMA B operands X ([pixel k][cout n]: transposed LDS reads) and
  // ones * X has the column sums in every row, X^T * X the sums of squares on its diagonal (products of two bf16 values
  // are exact in fp32; fp32 mode: the exact k-ordered FMA chain of v_mfma_f32_16x16x4_f32).  ~100 instructions per lane
  // instead of ~500 (the epilogue is issue-bound).
  unsigned char* ct = smem;
  float* st = reinterpret_cast<float*>(smem + Cfg::MAIN);
  const bool want_stats = a.stats != nullptr;
  const bool affine = a.oscale != nullptr || a.bias != nullptr || a.oshift != nullptr;
#pragma unroll
  for (int q = 0; q < TN; ++q) {
    const int colb = wn * 64 + q * 16 + lq * 4;   // first of this lane's four output channels
    float osc[4], bia[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { osc[rr] = 1.f; bia[rr] = 0.f; }
    if (affine) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int c = n0 + colb + rr;
        const bool ok = c < a.Cout;
        if (a.oscale && ok) osc[rr] = a.oscale[c];
        bia[rr] = ((a.bias && ok) ? a.bias[c] : 0.f) + ((a.oshift && ok) ? a.oshift[c] : 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int g = wm * TM + i;
      const int row = (g / MTX) * TW + (g % MTX) * 16 + lr;   // tile-local pixel
      float v[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) v[rr] = affine ? fmaf(acc[i][q][rr], osc[rr], bia[rr]) : acc[i][q][rr];
      if constexpr (sizeof(T) == 2) {
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(ct + row * Cfg::CLD + colb * 2) = pk;
      } else {
        *reinterpret_cast<float4*>(ct + row * Cfg::CLD + colb * 4) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  stamp(7);
  if (want_stats) {   // wave-uniform: EXEC is all ones for the transposed reads
    const unsigned char* cw = ct + (wm * 64) * Cfg::CLD + (wn * 64) * (int)sizeof(T);   // this wave's own block: no barrier needed
#pragma unroll
    for (int q = 0; q < TN; ++q) {
      f32x4_t m1 = f32x4_t{0.f, 0.f, 0.f, 0.f}, m2 = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if constexpr (sizeof(T) == 2) {
        typedef __attribute__((address_space(3))) s16x4_t* lds_p;
        const int p4 = lr & 3, q4 = lr >> 2;   // lane 4 * q4 + p4 of its 16-lane group: row q4, columns 4 * p4 .. 4 * p4 + 3
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // 32 pixels per MFMA K step
          const unsigned char* lo = cw + (ks * 32 + 8 * lq + q4) * Cfg::CLD + q * 32 + 8 * p4;
          const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo));
          const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo + 4 * Cfg::CLD));
          u32x4 xf;
          xf.x = (unsigned)(unsigned short)x0[0] | ((unsigned)(unsigned short)x0[1] << 16);
          xf.y = (unsigned)(unsigned short)x0[2] | ((unsigned)(unsigned short)x0[3] << 16);
          xf.z = (unsigned)(unsigned short)x1[0] | ((unsigned)(unsigned short)x1[1] << 16);
          xf.w = (unsigned)(unsigned short)x1[2] | ((unsigned)(unsigned short)x1[3] << 16);
          const u32x4 ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
          GMma<T>::run(ones, xf, m1);
          GMma<T>::run(xf, xf, m2);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {   // four pixels per MFMA K step, four steps per call
          u32x4 xf;
          xf.x = *reinterpret_cast<const unsigned*>(cw + ((kk + 0) * 4 + lq) * Cfg::CLD + (q * 16 + lr) * 4);
          xf.y = *reinterpret_cast<const unsigned*>(cw + ((kk + 1) * 4 + lq) * Cfg::CLD + (q * 16 + lr) * 4);
          xf.z = *reinterpret_cast<const unsigned*>(cw + ((kk + 2) * 4 + lq) * Cfg::CLD + (q * 16 + lr) * 4);
          xf.w = *reinterpret_cast<const unsigned*>(cw + ((kk + 3) * 4 + lq) * Cfg::CLD + (q * 16 + lr) * 4);
          const u32x4 ones = u32x4{0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u};
          GMma<T>::run(ones, xf, m1);
          GMma<T>::run(xf, xf, m2);
        }
      }
      // m1: every row holds the column sums -> row 0 = lanes with lq == 0, register 0.  m2: the diagonal element of
      // column lr sits in row lr = 4 * (lr >> 2) + (lr & 3), i.e. lane group lq == lr >> 2, register lr & 3.
      const int dr = lr & 3;
      const float d2 = dr == 0 ? m2[0] : dr == 1 ? m2[1] : dr == 2 ? m2[2] : m2[3];
      if (lq == 0) st[(wm * BN + wn * 64 + q * 16 + lr) * 2 + 0] = m1[0];
      if (lq == (lr >> 2)) st[(wm * BN + wn * 64 + q * 16 + lr) * 2 + 1] = d2;
    }
  }
  __syncthreads();
  if (a.stats && t < BN && (n0 + t) < a.Cout) {
    float x1 = 0.f, x2 = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < Cfg::WMN; ++w2) { x1 += st[(w2 * BN + t) * 2]; x2 += st[(w2 * BN + t) * 2 + 1]; }
    a.stats[(long)(n0 + t) * ntiles + tile] = x1;
    a.stats[((long)a.Cout + n0 + t) * ntiles + tile] = x2;
  }
  stamp(3);
  store_tile<T, TW, TPIX, BN, NT, Cfg::CLD, BNR>(a, ct, n, y0, x0, n0, t, tile, ntiles);
  stamp(4);
  stamp(6);
}

template <typename T, int TW, int TH, int BN, int HB, int NS, bool BNR, int XF>
int launch_hgd_cfg_b(const ConvArgs& a, hipStream_t s) {
  using Cfg = HgdCfg<T, TW, TH, BN, HB, NS>;
  static_assert(HgdCfg<T, TW, TH, BN, HB, NS>::DPW + 3 * HgdCfg<T, TW, TH, BN, HB, NS>::HITEMS + 4 <= 48, "wait_vm_lgkm0 covers the largest count");
  auto kern = conv3x3_hgd_kernel<T, TW, TH, BN, HB, NS, BNR, XF>;
  constexpr int SMEM = Cfg::SMEM + (XF == XF_APPLY ? Cfg::APTAB : 0);
  static_assert(2 * Cfg::SMEM > 160 * 1024 || 2 * SMEM <= 160 * 1024, "the coefficient table must not cost the second workgroup of a CU");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long M = (long)a.N * a.Hout * a.Wout;
  dim3 grid((unsigned)(M / Cfg::TPIX) * (a.Cout / BN));
  ConvArgs b = a;
  b.xcd_remap = tune("FLAIR_XCD_REMAP", 1);
  b.dbg = (unsigned long long*)g_debug_buffer;
  {
    const double flops = 2.0 * (double)M * a.Cout * a.Kg;
    // (the fused BN-backward apply reads a second input tensor and writes the transformed one once)
    const double bytes = ((double)M / (a.up0 ? 4 : 1) * a.C0 * (XF == XF_APPLY ? (a.ap_dy ? 3 : 2) : 1) + (double)M * a.C1 +
                          (double)M * a.Cout * (a.accumulate ? 2 : 1)) * sizeof(T) + (double)a.Cout * a.Kg * sizeof(T);
    static const char* names[2][3] = {{"conv3x3_hg_f32_n32", "conv3x3_hg_f32_n64", "conv3x3_hg_f32_n128"},
                                      {"conv3x3_hg_bf16_n32", "conv3x3_hg_bf16_n64", "conv3x3_hg_bf16_n128"}};
    ProfScope ps(names[sizeof(T) == 2][BN == 128 ? 2 : BN == 64 ? 1 : 0], flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), SMEM, s, b);
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

template <typename T, int TW, int TH, int BN, int HB, int NS>
int launch_hgd_cfg(const ConvArgs& a, hipStream_t s) {
  if constexpr (BN == 32) {   // forward of decoder block 3 conv1 only: no lazy input, no fused data-gradient epilogue
    return (a.in_scale || a.bnr_partial || a.ap_y) ? -6 : launch_hgd_cfg_b<T, TW, TH, BN, HB, NS, false, XF_NONE>(a, s);
  } else {
    if (a.in_scale) return (a.bnr_partial || a.ap_y) ? -6 : launch_hgd_cfg_b<T, TW, TH, BN, HB, NS, false, XF_LAZY>(a, s);
    if (a.ap_y) {   // 128-wide column blocks only (conv_hg_applicable): the 64-wide layers have a single input chunk in bf16
      if constexpr (BN == 128)
        return a.bnr_partial ? launch_hgd_cfg_b<T, TW, TH, BN, HB, NS, true, XF_APPLY>(a, s)
                             : launch_hgd_cfg_b<T, TW, TH, BN, HB, NS, false, XF_APPLY>(a, s);
      else
        return -6;
    }
    return a.bnr_partial ? launch_hgd_cfg_b<T, TW, TH, BN, HB, NS, true, XF_NONE>(a, s)
                         : launch_hgd_cfg_b<T, TW, TH, BN, HB, NS, false, XF_NONE>(a, s);
  }
}

int g_hg_variant = -1;  // tuning override (FLAIR_HG_VARIANT): 0 = 256 px x 8 waves, 1 = 128 px x 4 waves x 2 WG/CU

template <typename T, int TW, int TH, int BN, int HB, bool BNR, int TMv = 4>
int launch_hg_cfg_b(const ConvArgs& a, hipStream_t s) {
  using Cfg = HgCfg<T, TW, TH, BN, HB, TMv>;
  auto kern = conv3x3_hg_kernel<T, TW, TH, BN, HB, BNR, TMv>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long M = (long)a.N * a.Hout * a.Wout;
  dim3 grid((unsigned)(M / Cfg::TPIX) * (a.Cout / BN));
  ConvArgs b = a;
  {
    b.xcd_remap = tune("FLAIR_XCD_REMAP", 1);
  }
  {
    const double flops = 2.0 * (double)M * a.Cout * a.Kg;
    const double bytes = ((double)M / (a.up0 ? 4 : 1) * a.C0 + (double)M * a.C1 + (double)M * a.Cout * (a.accumulate ? 2 : 1)) * sizeof(T) +
                         (double)a.Cout * a.Kg * sizeof(T);
    static const char* names[2][2] = {{"conv3x3_hg_f32_n64", "conv3x3_hg_f32_n128"}, {"conv3x3_hg_bf16_n64", "conv3x3_hg_bf16_n128"}};
    ProfScope ps(names[sizeof(T) == 2][BN == 128], flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), Cfg::SMEM, s, b);
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

template <typename T, int TW, int TH, int BN, int HB, int TMv = 4>
int launch_hg_cfg(const ConvArgs& a, hipStream_t s) {
  return a.bnr_partial ? launch_hg_cfg_b<T, TW, TH, BN, HB, true, TMv>(a, s) : launch_hg_cfg_b<T, TW, TH, BN, HB, false, TMv>(a, s);
}

}  // namespace

// FLAIR_HG_DMA: 0 = register-staged weight tiles (round-1 kernel), 1 = LDS-DMA ring of 3 slots + single halo buffer.
// Measured and dropped (profiles/r2_hg_variants.txt): ring of 2 slots + double halo buffer (prefetch distance 1: l4 676 vs
// 839 TFLOP/s), 16x16-pixel tiles with 8 waves and a ring of 4 (one workgroup per CU: 609-902 vs 839-933 TFLOP/s).
static int hg_dma_mode() { return tune("FLAIR_HG_DMA", 1); }

// tile pixel count for this layer: 128-pixel tiles (2 workgroups per CU, de-phased barriers) unless the image
// only tiles by 256; small layers also need the finer tiling to fill 256 CUs
static int hg_tile_pixels(int dtype, const ConvArgs& a) {
  (void)dtype;
  if (g_hg_variant < 0) g_hg_variant = tune("FLAIR_HG_VARIANT", 1);
  const bool can128 = (a.Wout % 16 == 0) && (a.Hout % 8 == 0);
  const bool can256 = (a.Wout % 32 == 0 && a.Hout % 8 == 0) || (a.Wout % 16 == 0 && a.Hout % 16 == 0);
  if ((a.Cout % 128) != 0) return can256 ? 256 : (can128 ? 128 : 0);  // 64-wide: 4 waves x 256 px, single halo buffer
  if (hg_dma_mode() == 0 && g_hg_variant == 0 && can256) return 256;
  if (can128) return 128;
  return can256 ? 256 : 0;
}

// 3x3 / stride 1 / pad 1, >= 64 output channels (multiple of 64), input channels in whole 128-byte chunks
// per source, NHWC output (no fp32-NCHW head here).  A lazy BatchNorm + ReLU on the input (in_scale) needs the
// LDS-DMA kernel and applies to src0 only.
bool conv_hg_applicable(int dtype, const ConvArgs& a) {
  const int ck = dtype == DT_F32 ? 32 : 64;
  const int Cin = a.C0 + a.C1;
  if (a.R != 3 || a.S != 3 || a.out_mul != 1 || a.in_div != 1 || a.pad != 1) return false;
  if (a.Hout != a.Hin || a.Wout != a.Win || a.out_nchw || !a.out) return false;
  if (a.in_scale && (hg_dma_mode() == 0 || a.bnr_partial || a.ap_y)) return false;
  // fused BN-backward apply: one source, no upsample, the LDS-DMA kernel; the write-back has 32-bit byte offsets
  // (at least two input chunks: the transform of chunk c + 1 hides under the tap steps of chunk c; with one chunk it is all
  // prologue — the 64-channel layers, which are HBM-bound anyway, lost 80 us per launch that way — and 128-wide column blocks)
  if (a.ap_y && (hg_dma_mode() == 0 || a.C1 || a.up0 || !a.ap_coef || (a.Cout % 128) != 0 || a.C0 > 512 || a.C0 < 2 * ck ||
                 (long)a.N * a.Hout * a.Wout * a.C0 * (long)dtype_size(dtype) >= (1L << 31)))
    return false;
  if ((Cin % ck) || (a.C0 % ck)) return false;
  // 32 output channels (decoder block 3 conv1, 128 -> 32 at 256^2: 155 GFLOP on the ridge): the LDS-DMA kernel with one
  // 32-wide column block on 256-pixel tiles; no fused data-gradient epilogues there
  if (a.Cout == 32) return hg_dma_mode() != 0 && tune("FLAIR_HG_N32", 1) != 0 && !a.bnr_partial && !a.in_scale && a.pool_c0 == 0 &&
                           ((a.Wout % 32 == 0 && a.Hout % 8 == 0) || (a.Wout % 16 == 0 && a.Hout % 16 == 0));
  if (a.Cout < 64 || (a.Cout % 64)) return false;
  return hg_tile_pixels(dtype, a) != 0;
}

int conv_hg_grid_rows(int dtype, const ConvArgs& a) { return (int)((long)a.N * a.Hout * a.Wout / hg_tile_pixels(dtype, a)); }

template <typename T>
static int launch_hg_t(int tp, const ConvArgs& a, hipStream_t s) {
  const bool n128 = (a.Cout % 128) == 0;
  const int dma = hg_dma_mode();
  if (a.Cout == 32) {
    if (a.Wout % 32 == 0) return launch_hgd_cfg<T, 32, 8, 32, 1, 3>(a, s);
    return launch_hgd_cfg<T, 16, 16, 32, 1, 3>(a, s);
  }
  // fp32 (parity mode) is MFMA-bound at 1/16 of the bf16 rate: the register-staged kernel is as fast there; the LDS-DMA
  // kernel serves it only for a lazy BatchNorm + ReLU input
  if (dma != 0 && (sizeof(T) == 2 || a.in_scale || a.ap_y)) {
    if (tp == 128) return n128 ? launch_hgd_cfg<T, 16, 8, 128, 1, 3>(a, s) : launch_hgd_cfg<T, 16, 8, 64, 1, 3>(a, s);
    if (a.Wout % 32 == 0) return n128 ? launch_hgd_cfg<T, 32, 8, 128, 1, 3>(a, s) : launch_hgd_cfg<T, 32, 8, 64, 1, 3>(a, s);
    return n128 ? launch_hgd_cfg<T, 16, 16, 128, 1, 3>(a, s) : launch_hgd_cfg<T, 16, 16, 64, 1, 3>(a, s);
  }
  if (tp == 128) return n128 ? launch_hg_cfg<T, 16, 8, 128, 2>(a, s) : launch_hg_cfg<T, 16, 8, 64, 2>(a, s);
  if (a.Wout % 32 == 0) return n128 ? launch_hg_cfg<T, 32, 8, 128, 2>(a, s) : launch_hg_cfg<T, 32, 8, 64, 1>(a, s);
  return n128 ? launch_hg_cfg<T, 16, 16, 128, 2>(a, s) : launch_hg_cfg<T, 16, 16, 64, 1>(a, s);
}

int set_debug_buffer(void* p) {
#ifdef FLAIR_HG_STAMPS
  g_debug_buffer = p;
  return 0;
#else
  (void)p;
  return -7;   // not a diagnostic build
#endif
}

bool conv_bnapply_fusable(int dtype, const ConvArgs& a) { return a.ap_y && conv_hg_applicable(dtype, a); }
bool conv_acc_src_ok(int dtype, const ConvArgs& a) { return a.acc_src && a.accumulate && a.pool_c0 == 0 && conv_hg_applicable(dtype, a); }

int launch_conv_hg(int dtype, const ConvArgs& a, hipStream_t s) {
  const int tp = hg_tile_pixels(dtype, a);
  return dtype == DT_F32 ? launch_hg_t<float>(tp, a, s) : launch_hg_t<bf16_t>(tp, a, s);
}

}  // namespace flair
