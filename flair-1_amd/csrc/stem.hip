// Weight gradient of the ResNet stem (encoder.conv1: 7x7, stride 2, pad 3, 5 -> 64 channels; input stored with 8
// channels = one 16-byte chunk per pixel):   dW[k][(r,s)][c] = sum_p dY[p][k] * X[2p - 3 + (r,s)][c].
//
// The GEMM formulation (wgrad.hip) re-reads dY once per 64-column tile of the 392-wide gradient (7x) and gathers the
// im2col operand 16 bytes at a time: 0.49 ms for 66 GFLOP.  Here a persistent workgroup of 4 waves keeps the WHOLE
// 64 x (49 taps x 8 channels) gradient in registers (each wave 14 taps: 4 x 7 accumulator tiles) and walks 8x16-pixel
// output tiles: the 21x37-pixel input halo and the 128-pixel dY tile are staged once per tile (double-buffered LDS,
// next tile prefetched in registers) and every tap is a strided transposed read (ds_read_b64_tr_b16) of that halo —
// X and dY cross HBM exactly once.  One fp32 slab per workgroup, summed in a fixed order by wgrad_reduce_kernel.
// bf16 only; fp32 (the parity mode) stays on the generic kernel.
#include "common.h"
#include "prof.h"
#include "tune.h"

namespace flair {

void launch_wgrad_reduce(const float* partial, float* dw, int splits, int Cout, int Cout_pad, int Kpad, int Cin,
                         int Cin_real, int R, int S, int accumulate, hipStream_t s);  // wgrad.hip

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TH = 8, TW = 16, TPIX = TH * TW;            // output tile
constexpr int HH = 2 * TH + 5, HWD = 2 * TW + 5, HPIX = HH * HWD;  // 21 x 37 input halo
constexpr int XSTRIDE = 16, DSTRIDE = 128 + 16;           // bytes per halo pixel / per dY pixel (64 bf16 + pad)
constexpr int XBYTES = (HPIX * XSTRIDE + 255) / 256 * 256, DBYTES = TPIX * DSTRIDE;
constexpr int STAGE = XBYTES + DBYTES, SMEM = 2 * STAGE;
constexpr int NT = 256;
constexpr int XITEMS = (HPIX + NT - 1) / NT;              // 4
constexpr int DITEMS = TPIX * 8 / NT;                     // 4
constexpr int TAPS = 49, NTW = 7;                         // accumulator column tiles (2 taps each) per wave: 4 x 7 x 2 = 56 >= 49

struct StemWgArgs {
  const bf16_t* x;    // [N][2H][2W][8]
  const bf16_t* dy;   // [N][H][W][64]
  float* partial;     // [gridDim.x][64][392]
  int N, H, W, ntiles;
  const bf16_t* y; const float* coef; const float* msc; const float* msh;   // FUSE: WgradArgs::fuse_*
};

// FUSE: dy holds the gradient w.r.t. the stem's ReLU output; the BatchNorm-backward apply runs on the dY chunks as they are staged
template <bool FUSE>
__global__ __launch_bounds__(NT) void wgrad_stem_kernel(const StemWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int H = a.H, W = a.W, Hi = 2 * H, Wi = 2 * W;
  const int tiles_x = W / TW, tiles_y = H / TH;

  f32x4_t acc[4][NTW];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transposed-read geometry of this lane (see wgrad_hg.hip): it supplies the addresses of pixels 8g+q and 8g+4+q of
  // a 32-pixel K step (two tile rows of 16) and receives, for column lane&15 of the fragment, 8 pixels
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int kpx = 8 * g + q;                         // pixel within the step; + 4 for the second read
  const int krow = kpx >> 4, kcol = kpx & 15;
  // B operand (X): column tile n of this wave = taps 2*(7*wave + n) + {0, 1}; this lane reads tap (p >> 1), channel
  // half (p & 1).  Byte offset of that tap's halo pixel relative to the tile's (2*row, 2*col) origin:
  int xoff[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    int tap = 2 * (NTW * wave + n) + (p >> 1);
    tap = tap < TAPS ? tap : TAPS - 1;               // padding columns: any valid address, never stored
    const int r = tap / 7, s = tap - 7 * r;
    xoff[n] = ((2 * krow + r) * HWD + 2 * kcol + s) * XSTRIDE + 8 * (p & 1);
  }
  const int doff = kpx * DSTRIDE + 8 * p;            // A operand (dY): + 32 bytes per 16-channel row tile

  u32x4 xr[XITEMS], dr[DITEMS], yr[FUSE ? DITEMS : 1];
  unsigned dr_tok = 0u;
  auto load_tile = [&](int tile) {
    const bool tok = tile < a.ntiles;
    dr_tok = tok ? 1u : 0u;
    const int tl = tok ? tile : 0;
    const int n = tl / (tiles_x * tiles_y);
    const int trem = tl - n * tiles_x * tiles_y;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
#pragma unroll
    for (int k = 0; k < XITEMS; ++k) {
      const int it = t + NT * k;
      const int hy = it / HWD, hx = it - hy * HWD;
      const int iy = 2 * y0 - 3 + hy, ix = 2 * x0 - 3 + hx;
      const bool ok = tok && it < HPIX && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
      const unsigned off = ok ? (unsigned)(((n * Hi + iy) * Wi + ix) * 8) : 0u;
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.x + off);
      xr[k] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int k = 0; k < DITEMS; ++k) {
      const int it = t + NT * k;
      const int px = it >> 3, ch = it & 7;
      const int py = px / TW, pxx = px - py * TW;
      const unsigned off = (unsigned)(((n * H + y0 + py) * W + x0 + pxx) * 64 + ch * 8);
      const u32x4 v = *reinterpret_cast<const u32x4*>(a.dy + (tok ? off : 0u));
      dr[k] = tok ? v : u32x4{0u, 0u, 0u, 0u};
      if constexpr (FUSE) yr[k] = *reinterpret_cast<const u32x4*>(a.y + (tok ? off : 0u));
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* xh = smem + buf * STAGE;
    unsigned char* dyt = xh + XBYTES;
#pragma unroll
    for (int k = 0; k < XITEMS; ++k) {
      const int it = t + NT * k;
      if (it < HPIX) *reinterpret_cast<u32x4*>(xh + it * XSTRIDE) = xr[k];
    }
    float k1[8], k2[8], k3[8], sc[8], sh[8];
    if constexpr (FUSE) {   // this thread's channel chunk (t & 7) never changes; re-read per tile (L1) rather than held
      const int c0 = (t & 7) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        k1[e] = a.coef[c0 + e]; k2[e] = a.coef[64 + c0 + e]; k3[e] = a.coef[128 + c0 + e];
        sc[e] = a.msc[c0 + e]; sh[e] = a.msh[c0 + e];
      }
    }
#pragma unroll
    for (int k = 0; k < DITEMS; ++k) {
      const int it = t + NT * k;
      u32x4 v = dr[k];
      if constexpr (FUSE) {
        float d[8], yy[8], g[8];
        chunk_to_f<bf16_t>(__builtin_bit_cast(uint4, dr[k]), d);
        chunk_to_f<bf16_t>(__builtin_bit_cast(uint4, yr[k]), yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dm = fmaf(yy[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
          g[e] = fmaf(k1[e], dm, fmaf(k2[e], yy[e], k3[e]));
        }
        v = __builtin_bit_cast(u32x4, f_to_chunk<bf16_t>(g));
        // (tiles past the end: dz = 0 but k2*y + k3 is not — mask the whole chunk)
        if (dr_tok == 0u) v = u32x4{0u, 0u, 0u, 0u};
      }
      *reinterpret_cast<u32x4*>(dyt + (it >> 3) * DSTRIDE + (it & 7) * 16) = v;
    }
  };
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

  load_tile(blockIdx.x);
  store_tile(0);
  __syncthreads();
  int it = 0;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, ++it) {
    load_tile(tile + gridDim.x);        // next tile rides in registers while this one is multiplied
    __builtin_amdgcn_sched_barrier(0);  // pin the prefetch ahead of the MFMA phase
    const unsigned char* xh = smem + (it & 1) * STAGE;
    const unsigned char* dyt = xh + XBYTES;
#pragma unroll 1
    for (int y = 0; y < TH; y += 2) {   // one K step = tile rows y, y+1
      u32x4 af[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = tr2(dyt + y * TW * DSTRIDE + doff + m * 32, 4 * DSTRIDE);
      const unsigned char* xrow = xh + 2 * y * HWD * XSTRIDE;
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const u32x4 bf = tr2(xrow + xoff[n], 8 * XSTRIDE);   // pixel + 4 of the step = halo column + 8
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[m]), __builtin_bit_cast(bf16x8_t, bf),
                                                              acc[m][n], 0, 0, 0);
      }
    }
    store_tile((it + 1) & 1);
    __syncthreads();
  }

  // every wave owns 14 taps of all 64 output channels: write the slab directly
  float* __restrict__ part = a.partial + (long)blockIdx.x * 64 * (TAPS * 8);
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int tap = 2 * (NTW * wave + n) + (lr >> 3);
    if (tap < TAPS) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(long)(m * 16 + lq * 4 + e) * (TAPS * 8) + tap * 8 + (lr & 7)] = acc[m][n][e];
    }
  }
}

}  // namespace

bool wgrad_stem_applicable(int dtype, const WgradArgs& a) {
  return dtype == DT_BF16 && a.R == 7 && a.S == 7 && a.stride == 2 && a.pad == 3 && a.C0 == 8 && a.C1 == 0 && !a.up0 &&
         a.Cout == 64 && a.dy_ld == 64 && a.Hin == 2 * a.Hout && a.Win == 2 * a.Wout && (a.Hout % TH) == 0 && (a.Wout % TW) == 0 &&
         (long)a.N * a.Hin * a.Win * 8 < (1L << 31);
}

static int stem_splits(const WgradArgs& a) {
  const long ntiles = (long)a.N * a.Hout * a.Wout / TPIX;
  return (int)(ntiles < 512 ? ntiles : 512);   // two persistent workgroups per CU
}

size_t wgrad_stem_workspace_bytes(const WgradArgs& a) { return (size_t)stem_splits(a) * 64 * TAPS * 8 * sizeof(float); }

bool wgrad_bnapply_fusable(int dtype, const WgradArgs& a) { return tune("FLAIR_STEM_FUSE", 1) && wgrad_stem_applicable(dtype, a); }

int launch_wgrad_stem(const WgradArgs& a, hipStream_t s) {
  StemWgArgs h;
  h.x = (const bf16_t*)a.x0; h.dy = (const bf16_t*)a.dy; h.partial = a.partial;
  h.N = a.N; h.H = a.Hout; h.W = a.Wout; h.ntiles = (int)((long)a.N * a.Hout * a.Wout / TPIX);
  h.y = (const bf16_t*)a.fuse_y; h.coef = a.fuse_coef; h.msc = a.fuse_msc; h.msh = a.fuse_msh;
  if (h.y && (!h.coef || !h.msc || !h.msh)) return -2;
  const int nsplit = stem_splits(a);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stem_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stem_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  {
    const double M = (double)a.N * a.Hout * a.Wout;
    ProfScope ps("wgrad_stem_bf16", 2.0 * M * 64 * TAPS * 8, (M * 64 * (h.y ? 2 : 1) + 4.0 * M * 8) * 2.0, s);
    if (h.y) hipLaunchKernelGGL(wgrad_stem_kernel<true>, dim3(nsplit), dim3(NT), SMEM, s, h);
    else hipLaunchKernelGGL(wgrad_stem_kernel<false>, dim3(nsplit), dim3(NT), SMEM, s, h);
  }
  FLAIR_CHECK_LAUNCH();
  launch_wgrad_reduce(a.partial, a.dw, nsplit, 64, 64, TAPS * 8, 8, a.Cin_real, 7, 7, a.accumulate, s);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ------------------------------------------------------------------------------------------------------------------
// Forward of the same layer (and nothing else): y[p][k] = sum_{(r,s),c} X[2p - 3 + (r,s)][c] * W[k][(r,s),c].
// The gather-form kernel spends 0.32 ms on it (16-byte gathers, 64-wide column tile).  Here the 49 x 8 = 392-deep
// reduction runs as 13 MFMA K steps of 4 taps each; a wave keeps its 32 output channels' weights in registers for the
// whole launch (104 VGPRs), the 21x37 input halo of an 8x16 output tile is staged once in LDS with even and odd
// columns de-interleaved (so the stride-2 window of a tap is 16 consecutive 16-byte slots: conflict-free ds_read_b128),
// persistent workgroups prefetch the next halo in registers, BatchNorm partial statistics accumulate in registers over
// all tiles of a workgroup ([2][64][gridDim.x] partials), and the bf16 tile leaves through LDS as 16-byte stores.
namespace {

constexpr int FH2 = (HWD + 1) / 2;                              // 19 slots per column parity
constexpr int FXBYTES = (HH * 2 * FH2 * 16 + 255) / 256 * 256;  // de-interleaved halo image
constexpr int FCLD = 128 + 16, FCBYTES = TPIX * FCLD;           // output tile [128 px][64 ch] bf16, padded rows
constexpr int FSMEM = 2 * FXBYTES + 2 * FCBYTES;
constexpr int KSTEPS = 13;

__global__ __launch_bounds__(NT) void conv_stem_kernel(const ConvArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nh = wave & 1, ph = wave >> 1;          // output-channel half, tile-row half
  const int H = a.Hout, W = a.Wout, Hi = a.Hin, Wi = a.Win;
  const int tiles_x = W / TW, tiles_y = H / TH;
  const bf16_t* __restrict__ x = (const bf16_t*)a.src0;
  const bf16_t* __restrict__ wp = (const bf16_t*)a.w;
  bf16_t* __restrict__ out = (bf16_t*)a.out;
  const int li = lane & 15, lg = lane >> 4;

  // stationary B fragments: column tile j -> output channels (2*nh + j)*16 + li, K step ks -> taps 4*ks + lg
  u32x4 bw[2][KSTEPS];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      bw[j][ks] = *reinterpret_cast<const u32x4*>(wp + (long)((2 * nh + j) * 16 + li) * a.Kpad + (4 * ks + lg) * 8);
  // A fragment addresses: pixel column li of a tile row, tap 4*ks + lg (taps past the 49th have zero weights: any
  // in-range address will do)
  int aoff[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    int tap = 4 * ks + lg;
    tap = tap < TAPS ? tap : TAPS - 1;
    const int r = tap / 7, s = tap - 7 * r;
    aoff[ks] = ((r * 2 + (s & 1)) * FH2 + li + (s >> 1)) * 16;
  }
  float osc[2], obi[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = (2 * nh + j) * 16 + li;
    osc[j] = a.oscale ? a.oscale[col] : 1.f;
    obi[j] = (a.bias ? a.bias[col] : 0.f) + (a.oshift ? a.oshift[col] : 0.f);
  }
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};

  u32x4 xr[XITEMS];
  auto load_tile = [&](int tile) {
    const bool tok = tile < ntiles;
    const int tl = tok ? tile : 0;
    const int n = tl / (tiles_x * tiles_y);
    const int trem = tl - n * tiles_x * tiles_y;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
#pragma unroll
    for (int k = 0; k < XITEMS; ++k) {
      const int it = t + NT * k;
      const int hy = it / HWD, hx = it - hy * HWD;
      const int iy = 2 * y0 - 3 + hy, ix = 2 * x0 - 3 + hx;
      const bool ok = tok && it < HPIX && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
      const unsigned off = ok ? (unsigned)(((n * Hi + iy) * Wi + ix) * 8) : 0u;
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + off);
      xr[k] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto store_halo = [&](int buf) {
    unsigned char* xh = smem + buf * FXBYTES;
#pragma unroll
    for (int k = 0; k < XITEMS; ++k) {
      const int it = t + NT * k;
      const int hy = it / HWD, hx = it - hy * HWD;
      if (it < HPIX) *reinterpret_cast<u32x4*>(xh + ((hy * 2 + (hx & 1)) * FH2 + (hx >> 1)) * 16) = xr[k];
    }
  };

  load_tile(blockIdx.x);
  store_halo(0);
  __syncthreads();
  int it = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
    load_tile(tile + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* xh = smem + (it & 1) * FXBYTES;
    unsigned char* ct = smem + 2 * FXBYTES + (it & 1) * FCBYTES;
    f32x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      u32x4 af[4];
#pragma unroll
      for (int m = 0; m < 4; ++m)   // tile row 4*ph + m starts 2 halo rows (of 2 parities x FH2 slots) further down each
        af[m] = *reinterpret_cast<const u32x4*>(xh + (4 * ph + m) * (4 * FH2 * 16) + aoff[ks]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[m]), __builtin_bit_cast(bf16x8_t, bw[j][ks]),
                                                              acc[m][j], 0, 0, 0);
    }
    // accumulators -> bf16 tile in LDS (+ running BatchNorm statistics of the ROUNDED values, like the other kernels)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (2 * nh + j) * 16 + li;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = (4 * ph + m) * TW + lg * 4 + e;   // MFMA D: row = 4*(lane >> 4) + e = pixel column, col = lane & 15
          float f = fmaf(acc[m][j][e], osc[j], obi[j]);
          if (a.orelu) f = fmaxf(f, 0.f);
          const bf16_t v = Elem<bf16_t>::from_f(f);
          const float vf = Elem<bf16_t>::to_f(v);
          s1[j] += vf;
          s2[j] = fmaf(vf, vf, s2[j]);
          *reinterpret_cast<bf16_t*>(ct + row * FCLD + col * 2) = v;
        }
    }
    store_halo((it + 1) & 1);
    __syncthreads();
    {
      const int n = tile / (tiles_x * tiles_y);
      const int trem = tile - n * tiles_x * tiles_y;
      const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
#pragma unroll
      for (int k = 0; k < TPIX * 8 / NT; ++k) {
        const int idx = t + NT * k;
        const int px = idx >> 3, ch = idx & 7;
        const int py = px / TW, pxx = px - py * TW;
        *reinterpret_cast<uint4*>(out + ((long)(n * H + y0 + py) * W + x0 + pxx) * a.out_ld + ch * 8) =
            *reinterpret_cast<const uint4*>(ct + px * FCLD + ch * 16);
      }
    }
  }
  if (a.stats) {
    // lanes lg = 0..3 and the two tile-row halves hold disjoint pixels of the same columns
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      s1[j] += __shfl_xor(s1[j], 16); s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 16); s2[j] += __shfl_xor(s2[j], 32);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [2 halves][64 cols][2]
    if (lg == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        red[(ph * 64 + (2 * nh + j) * 16 + li) * 2 + 0] = s1[j];
        red[(ph * 64 + (2 * nh + j) * 16 + li) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    if (t < 64) {
      a.stats[(long)t * gridDim.x + blockIdx.x] = red[t * 2] + red[(64 + t) * 2];
      a.stats[((long)64 + t) * gridDim.x + blockIdx.x] = red[t * 2 + 1] + red[(64 + t) * 2 + 1];
    }
  }
}

int stem_fwd_blocks(const ConvArgs& a) {
  const long ntiles = (long)a.N * a.Hout * a.Wout / TPIX;
  return (int)(ntiles < 512 ? ntiles : 512);
}

}  // namespace

bool conv_stem_applicable(int dtype, const ConvArgs& a) {
  return dtype == DT_BF16 && a.R == 7 && a.S == 7 && a.out_mul == 2 && a.in_div == 1 && a.pad == 3 && a.C0 == 8 && a.C1 == 0 &&
         !a.up0 && a.Cout == 64 && a.out && a.out_ld == 64 && !a.out_nchw && !a.accumulate && !a.ores && !a.in_scale &&
         a.pool_c0 == 0 && !a.bnr_partial && !a.out_sub && a.Hin == 2 * a.Hout && a.Win == 2 * a.Wout && (a.Hout % TH) == 0 &&
         (a.Wout % TW) == 0 && a.Kpad >= 8 * 4 * KSTEPS && (long)a.N * a.Hin * a.Win * 8 < (1L << 31);
}

int conv_stem_grid_rows(const ConvArgs& a) { return stem_fwd_blocks(a); }

int launch_conv_stem(const ConvArgs& a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FSMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const long M = (long)a.N * a.Hout * a.Wout;
  {
    ProfScope ps("conv_stem_bf16", 2.0 * (double)M * 64 * TAPS * 8, ((double)M * 64 + 4.0 * M * 8) * 2.0, s);
    hipLaunchKernelGGL(conv_stem_kernel, dim3(stem_fwd_blocks(a)), dim3(NT), FSMEM, s, a, (int)(M / TPIX));
  }
  FLAIR_CHECK_LAUNCH();
  return 0;
}

}  // namespace flair
