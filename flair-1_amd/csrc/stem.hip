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
};

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

  u32x4 xr[XITEMS], dr[DITEMS];
  auto load_tile = [&](int tile) {
    const bool tok = tile < a.ntiles;
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
#pragma unroll
    for (int k = 0; k < DITEMS; ++k) {
      const int it = t + NT * k;
      *reinterpret_cast<u32x4*>(dyt + (it >> 3) * DSTRIDE + (it & 7) * 16) = dr[k];
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

int launch_wgrad_stem(const WgradArgs& a, hipStream_t s) {
  StemWgArgs h;
  h.x = (const bf16_t*)a.x0; h.dy = (const bf16_t*)a.dy; h.partial = a.partial;
  h.N = a.N; h.H = a.Hout; h.W = a.Wout; h.ntiles = (int)((long)a.N * a.Hout * a.Wout / TPIX);
  const int nsplit = stem_splits(a);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  {
    const double M = (double)a.N * a.Hout * a.Wout;
    ProfScope ps("wgrad_stem_bf16", 2.0 * M * 64 * TAPS * 8, (M * 64 + 4.0 * M * 8) * 2.0, s);
    hipLaunchKernelGGL(wgrad_stem_kernel, dim3(nsplit), dim3(NT), SMEM, s, h);
  }
  FLAIR_CHECK_LAUNCH();
  launch_wgrad_reduce(a.partial, a.dw, nsplit, 64, 64, TAPS * 8, 8, a.Cin_real, 7, 7, a.accumulate, s);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace flair
