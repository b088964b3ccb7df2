// MetadataMLP of the reference (src/flair/model.py:74-96): Linear(45,64) -> Dropout(0.4) -> ReLU -> Linear(64,32) ->
// Dropout -> ReLU -> Linear(32,16) -> Dropout -> ReLU on one 45-float metadata vector per tile, forward and backward.
// 5.5 kFLOP per sample: one workgroup per sample forward, one workgroup for the whole (small) batch backward; fp32 FMA
// chains in k order.  Dropout is an input: optional masks (already scaled by 1/(1-p), zero = dropped) per layer, null in
// eval mode — the host mirror draws them with torch's generator, so the draw stream stays the caller's.
#include "common.h"

namespace flair {
namespace {

constexpr int D0 = 45, D1 = 64, D2 = 32, D3 = 16;

__global__ __launch_bounds__(64) void mlp_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, const float* __restrict__ w3,
                                                     const float* __restrict__ b3, const float* __restrict__ m1,
                                                     const float* __restrict__ m2, const float* __restrict__ m3,
                                                     float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ out) {
  __shared__ float s0[D0], s1[D1], s2[D2];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < D0) s0[t] = x[(long)b * D0 + t];
  __syncthreads();
  {   // Linear -> Dropout -> ReLU (the reference's order: the mask multiplies the pre-activation)
    float a = b1[t];
    for (int k = 0; k < D0; ++k) a = fmaf(w1[t * D0 + k], s0[k], a);
    if (m1) a *= m1[(long)b * D1 + t];
    a = fmaxf(a, 0.f);
    s1[t] = a;
    if (h1) h1[(long)b * D1 + t] = a;
  }
  __syncthreads();
  if (t < D2) {
    float a = b2[t];
    for (int k = 0; k < D1; ++k) a = fmaf(w2[t * D1 + k], s1[k], a);
    if (m2) a *= m2[(long)b * D2 + t];
    a = fmaxf(a, 0.f);
    s2[t] = a;
    if (h2) h2[(long)b * D2 + t] = a;
  }
  __syncthreads();
  if (t < D3) {
    float a = b3[t];
    for (int k = 0; k < D2; ++k) a = fmaf(w3[t * D2 + k], s2[k], a);
    if (m3) a *= m3[(long)b * D3 + t];
    out[(long)b * D3 + t] = fmaxf(a, 0.f);
  }
}

// one workgroup, B <= 256 samples: dz of a layer for the whole batch lives in LDS, then every thread owns weight-gradient
// elements (fixed summation order over the batch: deterministic)
__global__ __launch_bounds__(256) void mlp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ h1,
                                                      const float* __restrict__ h2, const float* __restrict__ out,
                                                      const float* __restrict__ w2, const float* __restrict__ w3,
                                                      const float* __restrict__ m1, const float* __restrict__ m2,
                                                      const float* __restrict__ m3, const float* __restrict__ dout, int B,
                                                      float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                      float* __restrict__ db2, float* __restrict__ dw3, float* __restrict__ db3,
                                                      float* __restrict__ dz /* scratch [B][64+32+16] */) {
  const int t = threadIdx.x;
  float* dz3 = dz;
  float* dz2 = dz + (long)B * D3;
  float* dz1 = dz2 + (long)B * D2;
  for (int i = t; i < B * D3; i += 256) {
    float g = out[i] > 0.f ? dout[i] : 0.f;
    if (m3) g *= m3[i];
    dz3[i] = g;
  }
  __syncthreads();
  for (int i = t; i < B * D2; i += 256) {
    const int b = i / D2, j = i - b * D2;
    float g = 0.f;
    for (int o = 0; o < D3; ++o) g = fmaf(w3[o * D2 + j], dz3[b * D3 + o], g);
    g = h2[i] > 0.f ? g : 0.f;
    if (m2) g *= m2[i];
    dz2[i] = g;
  }
  __syncthreads();
  for (int i = t; i < B * D1; i += 256) {
    const int b = i / D1, j = i - b * D1;
    float g = 0.f;
    for (int o = 0; o < D2; ++o) g = fmaf(w2[o * D1 + j], dz2[b * D2 + o], g);
    g = h1[i] > 0.f ? g : 0.f;
    if (m1) g *= m1[i];
    dz1[i] = g;
  }
  __syncthreads();
  for (int i = t; i < D3 * D2; i += 256) {
    const int o = i / D2, j = i - o * D2;
    float g = 0.f;
    for (int b = 0; b < B; ++b) g = fmaf(dz3[b * D3 + o], h2[b * D2 + j], g);
    dw3[i] = g;
  }
  for (int i = t; i < D2 * D1; i += 256) {
    const int o = i / D1, j = i - o * D1;
    float g = 0.f;
    for (int b = 0; b < B; ++b) g = fmaf(dz2[b * D2 + o], h1[b * D1 + j], g);
    dw2[i] = g;
  }
  for (int i = t; i < D1 * D0; i += 256) {
    const int o = i / D0, j = i - o * D0;
    float g = 0.f;
    for (int b = 0; b < B; ++b) g = fmaf(dz1[b * D1 + o], x[b * D0 + j], g);
    dw1[i] = g;
  }
  if (t < D3) { float g = 0.f; for (int b = 0; b < B; ++b) g += dz3[b * D3 + t]; db3[t] = g; }
  if (t < D2) { float g = 0.f; for (int b = 0; b < B; ++b) g += dz2[b * D2 + t]; db2[t] = g; }
  if (t < D1) { float g = 0.f; for (int b = 0; b < B; ++b) g += dz1[b * D1 + t]; db1[t] = g; }
}

}  // namespace
}  // namespace flair

using namespace flair;

extern "C" {
int flair_metadata_mlp_forward(const float* x, const float* const w[3], const float* const b[3], const float* const mask[3],
                               float* h1, float* h2, float* out, int B, void* stream) {
  if (!x || !w || !b || !out || B < 1) return -1;
  const float* m1 = mask ? mask[0] : nullptr;
  const float* m2 = mask ? mask[1] : nullptr;
  const float* m3 = mask ? mask[2] : nullptr;
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, w[0], b[0], w[1], b[1], w[2], b[2], m1, m2, m3,
                     h1, h2, out);
  FLAIR_CHECK_LAUNCH();
  return 0;
}

int flair_metadata_mlp_backward(const float* x, const float* h1, const float* h2, const float* out, const float* const w[3],
                                const float* const mask[3], const float* dout, int B, float* const dw[3], float* const db[3],
                                float* scratch, void* stream) {
  if (!x || !h1 || !h2 || !out || !w || !dout || !dw || !db || !scratch || B < 1 || B > 256) return -1;
  const float* m1 = mask ? mask[0] : nullptr;
  const float* m2 = mask ? mask[1] : nullptr;
  const float* m3 = mask ? mask[2] : nullptr;
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, h1, h2, out, w[1], w[2], m1, m2, m3, dout, B,
                     dw[0], db[0], dw[1], db[1], dw[2], db[2], scratch);
  FLAIR_CHECK_LAUNCH();
  return 0;
}
}
