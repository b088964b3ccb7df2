// Executor for smp-0.3.3 Unet(resnet34) on gfx950 — see unet.h.
// Reference call sites replaced: /root/reference/src/flair/model.py:57-64 (encoder / decoder /
// segmentation_head / whole-model forward) and their autograd backward.
#include <stdlib.h>

#include "unet.h"
#include "tune.h"

#include <string.h>

namespace flair {

int conv_weight_rows_pad(int cout);  // conv_igemm.hip
bool conv_tile_epilogue_ok(int dtype, const ConvArgs& a);

#define RUN(expr)                         \
  do {                                    \
    if (!dry_ && !err_) {                 \
      int rc__ = (expr);                  \
      if (rc__) err_ = rc__;              \
    }                                     \
  } while (0)

// ------------------------------------------------------------------------------------------ table
int UNet::add_conv(const std::string& name, int cin, int cout, int k, int stride, int pad, bool bias, int stage) {
  ConvDesc c;
  c.name = name; c.Cin = cin; c.Cout = cout; c.R = k; c.S = k; c.stride = stride; c.pad = pad; c.bias = bias;
  c.Cin_p = (int)round_up(cin, 8);
  // stored output channels: whole 32-byte (bf16) / 64-byte (fp32) groups.  Every layer of the graph is a multiple of 16 already;
  // the head's 13 classes take 16 columns, 19 classes 32 — with 24 (whole 16-byte chunks only, rounds 1-2) the head's backward had
  // no tile kernel (its data gradient has Cout_p INPUT channels, its weight gradient dy rows of Cout_p) and the lazy-BN training
  // path of the fused trainer refused 19 classes
  c.Cout_p = (int)round_up(cout < 16 ? 16 : cout, 16);
  c.w_off = n_params;
  TensorInfo t;
  t.name = name + ".weight"; t.ndim = 4; t.shape[0] = cout; t.shape[1] = cin; t.shape[2] = k; t.shape[3] = k;
  t.offset = n_params; t.kind = 0; t.stage = stage;
  tensors.push_back(t);
  n_params += (long)cout * cin * k * k;
  if (bias) {
    c.b_off = n_params;
    TensorInfo b;
    b.name = name + ".bias"; b.ndim = 1; b.shape[0] = cout; b.shape[1] = b.shape[2] = b.shape[3] = 1;
    b.offset = n_params; b.kind = 0; b.stage = stage;
    tensors.push_back(b);
    n_params += cout;
  }
  const int kstep = dtype == DT_F32 ? 32 : 64;
  c.Kg = k * k * c.Cin_p;
  c.Kpad = (int)round_up(c.Kg, kstep);
  c.rows_f = conv_weight_rows_pad(cout);
  c.Kgd = k * k * c.Cout_p;
  c.Kpad_d = (int)round_up(c.Kgd, kstep);
  c.rows_d = conv_weight_rows_pad(c.Cin_p);
  convs.push_back(c);
  return (int)convs.size() - 1;
}

int UNet::add_bn(const std::string& name, int C, int stage) {
  BnDesc b;
  b.name = name; b.C = C;
  const char* pn[2] = {".weight", ".bias"};
  for (int i = 0; i < 2; ++i) {
    TensorInfo t;
    t.name = name + pn[i]; t.ndim = 1; t.shape[0] = C; t.shape[1] = t.shape[2] = t.shape[3] = 1;
    t.offset = n_params; t.kind = 0; t.stage = stage;
    tensors.push_back(t);
    (i == 0 ? b.g_off : b.b_off) = n_params;
    n_params += C;
  }
  const char* bn_[2] = {".running_mean", ".running_var"};
  for (int i = 0; i < 2; ++i) {
    TensorInfo t;
    t.name = name + bn_[i]; t.ndim = 1; t.shape[0] = C; t.shape[1] = t.shape[2] = t.shape[3] = 1;
    t.offset = n_buffers; t.kind = 1; t.stage = stage;
    tensors.push_back(t);
    (i == 0 ? b.rm_off : b.rv_off) = n_buffers;
    n_buffers += C;
  }
  bns.push_back(b);
  return (int)bns.size() - 1;
}

// Layer order == forward order == flat parameter order, so the gradient of stage k occupies
// [stage_begin[k], stage_begin[k+1]) and becomes ready in reverse order during backward
// (bucketed RCCL all-reduce).  Every stage start is padded to 4 floats.
void UNet::build_table() {
  auto pad4 = [&]() { n_params = round_up(n_params, 4); };
  stage_begin[0] = 0;
  add_conv("encoder.conv1", in_channels, 64, 7, 2, 3, false, 0);
  add_bn("encoder.bn1", 64, 0);
  const int planes[4] = {64, 128, 256, 512}, nblk[4] = {3, 4, 6, 3};
  int inpl = 64;
  for (int L = 0; L < 4; ++L) {
    pad4();
    stage_begin[L + 1] = n_params;
    for (int b = 0; b < nblk[L]; ++b) {
      const std::string p = "encoder.layer" + std::to_string(L + 1) + "." + std::to_string(b);
      const int stride = (b == 0 && L > 0) ? 2 : 1;
      add_conv(p + ".conv1", inpl, planes[L], 3, stride, 1, false, L + 1);
      add_bn(p + ".bn1", planes[L], L + 1);
      add_conv(p + ".conv2", planes[L], planes[L], 3, 1, 1, false, L + 1);
      add_bn(p + ".bn2", planes[L], L + 1);
      if (b == 0 && (stride != 1 || inpl != planes[L])) {
        add_conv(p + ".downsample.0", inpl, planes[L], 1, stride, 0, false, L + 1);
        add_bn(p + ".downsample.1", planes[L], L + 1);
      }
      inpl = planes[L];
    }
  }
  pad4();
  stage_begin[5] = n_params;
  const int din[5] = {512, 256, 128, 64, 32}, dskip[5] = {256, 128, 64, 64, 0}, dout[5] = {256, 128, 64, 32, 16};
  for (int i = 0; i < 5; ++i) {
    const std::string p = "decoder.blocks." + std::to_string(i);
    add_conv(p + ".conv1.0", din[i] + dskip[i], dout[i], 3, 1, 1, false, 5);
    add_bn(p + ".conv1.1", dout[i], 5);
    add_conv(p + ".conv2.0", dout[i], dout[i], 3, 1, 1, false, 5);
    add_bn(p + ".conv2.1", dout[i], 5);
  }
  pad4();
  stage_begin[6] = n_params;
  add_conv("segmentation_head.0", 16, classes, 3, 1, 1, true, 6);
  pad4();
  stage_begin[7] = n_params;
}

UNet::UNet(int in_ch, int cls, int dt) : in_channels(in_ch), classes(cls), dtype(dt) { build_table(); }

UNet::~UNet() {
  for (hipEvent_t e : fork_ev_) (void)hipEventDestroy(e);
  if (join_ev_) (void)hipEventDestroy(join_ev_);
  if (side_) (void)hipStreamDestroy(side_);
  if (note_) (void)hipStreamDestroy(note_);
}

bool UNet::side_init() {
  if (!tune("FLAIR_WGRAD_STREAM", 1)) return false;
  if (side_) return true;
  // LOW priority: (a) the weight gradients are off the critical path — when both streams have workgroups to dispatch,
  // the BatchNorm / data-gradient chain goes first; (b) a priority level has hardware queues of its own, so the side
  // stream can never be mapped onto the caller's queue.  (HIP multiplexes streams over a few hardware queues; with RCCL's
  // and torch's extra streams alive the round-robin put this stream on the caller's queue and the overlap was gone:
  // +1.4 ms per step in the single-rank rehearsal of the exchange, gpurun_out/prof_exch Queue_Id column.)
  int least = 0, greatest = 0;
  const bool low = tune("FLAIR_SIDE_PRIO", 1) != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
  const hipError_t ce = low ? hipStreamCreateWithPriority(&side_, hipStreamNonBlocking, least)
                            : hipStreamCreateWithFlags(&side_, hipStreamNonBlocking);
  if (ce != hipSuccess) { side_ = nullptr; return false; }
  // The fork / join events order streams of ONE device: no system-scope fence (the default makes every record write back and
  // invalidate the caches — 50 times per step, in front of the kernels that re-read what was just written)
  const unsigned evflags = hipEventDisableTiming | (tune("FLAIR_EVENT_SYSFENCE", 0) ? 0u : hipEventDisableSystemFence);
  fork_ev_.resize(128);
  for (auto& e : fork_ev_)
    if (hipEventCreateWithFlags(&e, evflags) != hipSuccess) return false;
  return hipEventCreateWithFlags(&join_ev_, evflags) == hipSuccess;
}

// Space sharing instead of time sharing: on the side stream the persistent weight-gradient kernel takes only HALF of the
// CUs (FLAIR_WG_CUS, 128), for twice as long.  The caller's stream always finds free CUs — its BatchNorm kernels, HBM-bound,
// lose little on the other half and no longer queue behind 256 persistent workgroups (the "35 us bn_bwd_finalize" of the
// round-2 trace) — and half the workgroups means half the fp32 slabs (75 -> 37 MB written and re-read per launch).
// Step 13.2 -> 12.4 ms (192 CUs: 12.55, 144: 12.48, 112: 12.56, 96: 12.87; giving the last units of backward, when the
// caller's stream runs dry, the whole chip again: +0.05 ... +0.3 ms).
int UNet::side_cus(int) const {
  if (!tune("FLAIR_WGRAD_STREAM", 1)) return 0;
  // fp32 mode: everything is matrix-pipe-bound there (fp32 MFMA is 16x slower), the BatchNorm kernels are a small share and
  // halving the weight-gradient kernels' CUs costs more than it frees: 57.4 ms with the whole chip, 58.4 with 192, 62.2 with 128
  const int v = tune("FLAIR_WG_CUS", 0);   // (tune() caches its default per key: the dtype-dependent default lives here)
  return v > 0 ? v : (dtype == DT_F32 ? 256 : 128);
}

hipStream_t UNet::wgrad_stream() {
  if (dry_ || err_ || !side_init()) return s_;
  hipEvent_t e = fork_ev_[fork_next_++ % fork_ev_.size()];
  if (hipEventRecord(e, s_) != hipSuccess || hipStreamWaitEvent(side_, e, 0) != hipSuccess) return s_;
  side_pending_ = true;
  return side_;
}

void UNet::side_join() {
  if (!side_pending_ || dry_) return;
  side_pending_ = false;
  if (hipEventRecord(join_ev_, side_) != hipSuccess || hipStreamWaitEvent(s_, join_ev_, 0) != hipSuccess) {
    if (!err_) err_ = -13;
  }
}

// ------------------------------------------------------------------------------------------ arena
void* UNet::alloc(size_t bytes) {
  const size_t off = top_;
  top_ = (size_t)round_up((long)(top_ + bytes), 256);
  if (!dry_ && top_ > cap_) { if (!err_) err_ = -100; return base_; }
  return base_ + off;  // dry runs use a fake, never dereferenced base so tensors keep distinct identities
}

Act UNet::alloc_act(int N, int H, int W, int C) {
  Act a;
  a.N = N; a.H = H; a.W = W; a.C = C;
  a.p = alloc((size_t)a.elems() * dtype_size(dtype));
  return a;
}

void UNet::begin(void* ws, size_t ws_bytes, hipStream_t s, bool dry) {
  base_ = dry ? (unsigned char*)0x100000 : (unsigned char*)ws;
  cap_ = ws_bytes; top_ = 0; dry_ = dry; err_ = 0; s_ = s;
  units_.clear();
  gbufs_.clear();
  packed_d_ = false;
  dl_nhwc_ = nullptr;
  logits_nhwc_ = nullptr;
  // weight arena
  for (auto& c : convs) {
    c.wf = top_; alloc((size_t)c.rows_f * c.Kpad * dtype_size(dtype));
  }
  for (auto& c : convs) {
    c.wd = top_; alloc((size_t)c.rows_d * c.Kpad_d * dtype_size(dtype));
  }
  const int kstep = dtype == DT_F32 ? 32 : 64;
  for (auto& c : convs) {
    if (!c.parity_dgrad()) continue;
    for (int cls = 0; cls < 4; ++cls) {
      const int taps = ((cls >> 1) ? 2 : 1) * ((cls & 1) ? 2 : 1);
      c.Kg_cls[cls] = taps * c.Cout_p;
      c.Kpad_cls[cls] = (int)round_up(c.Kg_cls[cls], kstep);
      c.wd_cls[cls] = top_; alloc((size_t)c.rows_d * c.Kpad_cls[cls] * dtype_size(dtype));
    }
  }
}

// fp32 OIHW masters -> packed T copies, all 47 convolutions in one launch
void UNet::pack_forward_weights() {
  PackTable tb;
  tb.n = 0;
  for (auto& c : convs) {
    PackDesc& d = tb.d[tb.n++];
    d.w_off = c.w_off; d.dst_off = c.wf; d.Cout = c.Cout; d.Cin = c.Cin; d.R = c.R; d.S = c.S;
    d.Cin_p = c.Cin_p; d.rows_pad = c.rows_f; d.Kpad = c.Kpad; d.tf = 0;
    d.Rc = 0; d.r0 = d.rstep = d.s0 = d.sstep = d.Sc = 0;
  }
  RUN(pack_weights_all(dtype, params_, base_, tb, s_));
}

void UNet::pack_dgrad_weights() {
  if (packed_d_) return;
  packed_d_ = true;
  PackTable tb;
  tb.n = 0;
  for (size_t i = 1; i < convs.size(); ++i) {  // the stem needs no data gradient
    auto& c = convs[i];
    PackDesc& d = tb.d[tb.n++];
    d.w_off = c.w_off; d.dst_off = c.wd; d.Cout = c.Cout; d.Cin = c.Cin; d.R = c.R; d.S = c.S;
    d.Cin_p = c.Cout_p; d.rows_pad = c.rows_d; d.Kpad = c.Kpad_d; d.tf = 1;
    d.Rc = 0; d.r0 = d.rstep = d.s0 = d.sstep = d.Sc = 0;
    if (c.parity_dgrad()) {
      for (int cls = 0; cls < 4; ++cls) {
        // gather-form tap kr reads dY row (ho - 1 + kr) / 2: even output rows use kr = 1, odd rows kr = 0 and 2
        PackDesc& e = tb.d[tb.n++];
        e = d;
        e.dst_off = c.wd_cls[cls]; e.Kpad = c.Kpad_cls[cls];
        const int py = cls >> 1, px = cls & 1;
        e.Rc = py ? 2 : 1; e.r0 = py ? 0 : 1; e.rstep = 2;
        e.Sc = px ? 2 : 1; e.s0 = px ? 0 : 1; e.sstep = 2;
      }
    }
  }
  RUN(pack_weights_all(dtype, params_, base_, tb, s_));
}

void* UNet::grad_of(const Act& a, bool* accumulate) {
  for (auto& g : gbufs_)
    if (g.act == a.p) { *accumulate = g.init; g.init = true; return g.g; }
  GradBuf g;
  g.act = a.p;
  g.g = alloc((size_t)a.elems() * dtype_size(dtype));
  g.init = true;
  *accumulate = false;
  gbufs_.push_back(g);
  return g.g;
}

void* UNet::grad_peek(const Act& a) {
  for (auto& g : gbufs_)
    if (g.act == a.p) return g.g;
  return nullptr;
}

// ------------------------------------------------------------------------------------------ forward
static void fill_conv_args(ConvArgs& a, const ConvDesc& c, const Act& in0, const Act& in1, bool up0, const Act& y,
                           const void* wpacked) {
  memset(&a, 0, sizeof(a));
  a.src0 = in0.p; a.src1 = in1.p;
  a.C0 = in0.C; a.C1 = in1.p ? in1.C : 0;
  a.up0 = up0 ? 1 : 0;
  a.N = in0.N;
  a.Hin = up0 ? in0.H * 2 : in0.H;
  a.Win = up0 ? in0.W * 2 : in0.W;
  a.Hout = y.H; a.Wout = y.W;
  a.R = c.R; a.S = c.S;
  a.out_mul = c.stride; a.pad = c.pad; a.in_div = 1;
  a.Cout = c.Cout;
  a.Kg = c.Kg; a.Kpad = c.Kpad;
  a.w = wpacked;
  a.out = y.p; a.out_ld = y.C;
  a.in_scale = in0.lz_scale; a.in_shift = in0.lz_shift;   // lazy producers never feed a concat partner (in1)
}

bool wgrad_big_applicable(int dtype, const WgradArgs& a);   // wgrad_hg.hip
bool conv_hg_applicable(int dtype, const ConvArgs& a);      // conv_hg.hip

// Can the unit producing `y` (conv + BN + ReLU, no residual) hand its PRE-BatchNorm tensor to its single consumer, the
// 3x3 convolution `cons` (input = [up2(y) if up0] ++ skip), which then applies BN + ReLU while it stages its halo — in
// the forward pass (halo-GEMM kernel) and again in its weight gradient (wgrad3x3_big)?
bool UNet::lazy_into_hg(const Act& y, int cons, bool up0, const Act& skip) const {
  if (!training_ || !tune("FLAIR_LAZY_HG", 1)) return false;
  const ConvDesc& c = convs[cons];
  if (c.R != 3 || c.stride != 1) return false;
  Act in0 = y, out;
  in0.p = reinterpret_cast<void*>(16);
  out.N = y.N; out.H = up0 ? 2 * y.H : y.H; out.W = up0 ? 2 * y.W : y.W; out.C = c.Cout_p; out.p = reinterpret_cast<void*>(16);
  ConvArgs a;
  fill_conv_args(a, c, in0, skip, up0, out, reinterpret_cast<void*>(16));
  a.in_scale = a.in_shift = reinterpret_cast<const float*>(16);
  if (!conv_hg_applicable(dtype, a)) return false;
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.x0 = in0.p; w.x1 = skip.p; w.C0 = in0.C; w.C1 = skip.p ? skip.C : 0; w.up0 = up0 ? 1 : 0;
  w.N = y.N; w.Hin = out.H; w.Win = out.W; w.Hout = out.H; w.Wout = out.W; w.R = 3; w.S = 3; w.stride = 1; w.pad = 1;
  w.dy = in0.p; w.dy_ld = c.Cout_p; w.Cout = c.Cout; w.Cin_real = c.Cin;
  w.in_scale = w.in_shift = reinterpret_cast<const float*>(16);
  return wgrad_big_applicable(dtype, w);
}

int UNet::run_unit(int ci, int bi, const Act& in0, const Act& in1, bool up0, bool relu, int res_unit, const Act& res,
                   bool materialize, int lazy_cons, bool lazy_up0, const Act* lazy_skip) {
  const ConvDesc& c = convs[ci];
  const BnDesc& b = bns[bi];
  Unit u;
  u.conv = ci; u.bn = bi; u.in0 = in0; u.in1 = in1; u.up0 = up0; u.relu = relu; u.res_unit = res_unit; u.res = res;
  const int Hin = up0 ? in0.H * 2 : in0.H, Win = up0 ? in0.W * 2 : in0.W;
  const int Ho = (Hin + 2 * c.pad - c.R) / c.stride + 1, Wo = (Win + 2 * c.pad - c.S) / c.stride + 1;
  u.y = alloc_act(in0.N, Ho, Wo, c.Cout_p);
  u.scale = alloc_f(b.C); u.shift = alloc_f(b.C); u.mean = alloc_f(b.C); u.invstd = alloc_f(b.C);
  ConvArgs a;
  fill_conv_args(a, c, in0, in1, up0, u.y, base_ + c.wf);
  if (!training_) {
    // inference: BatchNorm (running statistics) is a per-channel affine -> folded, together with the residual
    // add and the ReLU, into the conv epilogue; the pre-BN tensor is never written
    if (!reuse_)   // constant weights: the coefficients of the previous eval forward are still at u.scale / u.shift
      RUN(bn_eval_coeffs(b.C, params_ + b.g_off, params_ + b.b_off, buffers_ + b.rm_off, buffers_ + b.rv_off, 1e-5f,
                         u.scale, u.shift, s_));
    u.out = u.y;
    a.oscale = u.scale; a.oshift = u.shift; a.orelu = relu ? 1 : 0;
    a.ores = res_unit >= 0 ? units_[res_unit].out.p : res.p;
    RUN(launch_conv(dtype, a, s_));
    units_.push_back(u);
    return (int)units_.size() - 1;
  }
  const int nblk = conv_grid_rows(dtype, a);
  float* partial = training_ ? alloc_f((long)nblk * 2 * b.C) : nullptr;
  a.stats = partial;
  RUN(launch_conv(dtype, a, s_));
  if (training_)
    RUN(bn_finalize(partial, nblk, b.C, u.y.rows(), params_ + b.g_off, params_ + b.b_off, buffers_ + b.rm_off,
                    buffers_ + b.rv_off, 0.1f, 1e-5f, u.scale, u.shift, u.mean, u.invstd, s_));
  else
    RUN(bn_eval_coeffs(b.C, params_ + b.g_off, params_ + b.b_off, buffers_ + b.rm_off, buffers_ + b.rv_off, 1e-5f,
                       u.scale, u.shift, s_));
  // The 16-channel decoder units (block 4: 45 % of all BN-apply bytes) feed only small-channel halo kernels,
  // which can apply BN + ReLU while staging their input: skip the activation pass and hand out the pre-BN tensor.
  // The >= 64-channel units whose single consumer is a halo-GEMM convolution do the same (conv1 of every BasicBlock,
  // decoder conv1 / conv2 of the wide blocks): bn_act and the normalised activation tensor vanish for them.
  const bool plain = lazy_ok_ && training_ && materialize && relu && res_unit < 0 && !res.p;
  bool lazy = plain && c.R == 3 && c.stride == 1 && c.Cout_p <= lazy_max_c_ && (Ho % 8) == 0 && (Wo % 32) == 0;
  if (!lazy && plain && lazy_cons >= 0) {
    Act none_;
    lazy = lazy_into_hg(u.y, lazy_cons, lazy_up0, lazy_skip ? *lazy_skip : none_);
  }
  if (lazy) {
    u.out = u.y;
    u.out.lz_scale = u.scale; u.out.lz_shift = u.shift;
  } else if (materialize) {
    u.out = alloc_act(in0.N, Ho, Wo, c.Cout_p);
    const void* rp = nullptr;
    const float *rs = nullptr, *rh = nullptr;
    if (res_unit >= 0) { rp = units_[res_unit].y.p; rs = units_[res_unit].scale; rh = units_[res_unit].shift; }
    else if (res.p) rp = res.p;
    RUN(bn_act(dtype, u.y.p, u.scale, u.shift, rp, rs, rh, u.out.p, u.y.rows(), b.C, relu ? 1 : 0, s_));
  }
  units_.push_back(u);
  return (int)units_.size() - 1;
}

void UNet::fwd_common_begin(const float* params, float* buffers, int B, int H, int W, int training) {
  params_ = params; buffers_ = buffers; B_ = B; H_ = H; W_ = W; training_ = training;
}

void UNet::encoder_fwd_impl(const float* x_nchw) {
  xin_ = alloc_act(B_, H_, W_, convs[0].Cin_p);
  RUN(nchw_f32_to_nhwc(dtype, x_nchw, xin_.p, B_, in_channels, H_, W_, xin_.C, s_));
  Act none;
  int ci = 0, bi = 0;
  // training: the stem's BatchNorm + ReLU pass and the max pool read the pre-BN tensor together (bn_act_maxpool_kernel)
  const bool fuse_pool = training_ && tune("FLAIR_STEM_POOL", 1);
  int u = run_unit(ci++, bi++, xin_, none, false, true, -1, none, !fuse_pool);
  if (fuse_pool) units_[u].out = alloc_act(units_[u].y.N, units_[u].y.H, units_[u].y.W, units_[u].y.C);
  f_[1] = units_[u].out;
  pool_ = alloc_act(B_, f_[1].H / 2, f_[1].W / 2, 64);
  pool_idx_ = training_ ? (unsigned char*)alloc((size_t)pool_.elems()) : nullptr;
  if (fuse_pool)
    RUN(bn_act_maxpool3x3s2(dtype, units_[u].y.p, units_[u].scale, units_[u].shift, f_[1].p, pool_.p, pool_idx_, B_, f_[1].H, f_[1].W, 64, s_));
  else
    RUN(maxpool3x3s2_fwd(dtype, f_[1].p, pool_.p, pool_idx_, B_, f_[1].H, f_[1].W, 64, s_));
  Act x = pool_;
  const int nblk[4] = {3, 4, 6, 3};
  for (int L = 0; L < 4; ++L) {
    for (int b = 0; b < nblk[L]; ++b) {
      const bool ds = (b == 0 && L > 0);
      const int c1 = ci++, b1 = bi++, c2 = ci++, b2 = bi++;
      int u1 = run_unit(c1, b1, x, none, false, true, -1, none, true, /*single consumer: conv2*/ c2);
      int ud = -1;
      if (ds) { const int cd = ci++, bd = bi++; ud = run_unit(cd, bd, x, none, false, false, -1, none, false); }
      int u2 = run_unit(c2, b2, units_[u1].out, none, false, true, ud, ds ? none : x, true);
      x = units_[u2].out;
    }
    f_[L + 2] = x;
  }
  enc_units_end_ = (int)units_.size();
}

void UNet::decoder_fwd_impl() {
  Act none;
  dec_units_begin_ = (int)units_.size();
  int ci = 36, bi = 36;  // decoder convs / bns follow the 36 encoder ones
  Act x = f_[5];
  for (int i = 0; i < 5; ++i) {
    const Act skip = i < 4 ? f_[4 - i] : none;
    const Act nskip = i < 3 ? f_[3 - i] : none;   // the next block's skip partner
    const int c1 = ci++, b1 = bi++, c2 = ci++, b2 = bi++;
    int u1 = run_unit(c1, b1, x, skip, true, true, -1, none, true, /*single consumer: conv2*/ c2);
    int u2 = run_unit(c2, b2, units_[u1].out, none, false, true, -1, none, true, /*next block's conv1*/ i < 4 ? c2 + 1 : -1, true, &nskip);
    x = units_[u2].out;
  }
  dec_out_ = x;
}

void UNet::head_fwd_impl(float* logits_nchw) {
  const ConvDesc& c = convs.back();
  Act none, y;
  y.N = dec_out_.N; y.H = dec_out_.H; y.W = dec_out_.W; y.C = c.Cout_p; y.p = nullptr;
  ConvArgs a;
  fill_conv_args(a, c, dec_out_, none, false, y, base_ + c.wf);
  if (logits_nchw || dry_) {
    a.out = nullptr;
    a.out_nchw = logits_nchw;
    logits_nhwc_ = nullptr;
  }
  if (!logits_nchw) {
    // nobody asked for fp32 NCHW logits (the fused trainer): keep them in the network's layout, [B*H*W][Cout_p] T, for
    // flair_ce_head_nhwc — 40 % fewer bytes than the NCHW fp32 tensor and contiguous rows on both sides
    logits_nhwc_ = alloc((size_t)dec_out_.rows() * c.Cout_p * dtype_size(dtype));
    if (!dry_) { a.out = logits_nhwc_; a.out_ld = c.Cout_p; a.out_nchw = nullptr; }
  }
  a.bias = params_ + c.b_off;
  unsigned char* preds = (!logits_nchw && !training_) ? preds_req_ : nullptr;
  float* maxprob = preds ? maxprob_req_ : nullptr;
  preds_req_ = nullptr; maxprob_req_ = nullptr;
  if (preds && !dry_) {
    // predict: argmax in the head convolution's register epilogue, the logits never reach HBM (536 MB of traffic and the
    // softmax_argmax launch saved); otherwise the separate kernel over the NHWC logits
    ConvArgs b = a;
    b.out = nullptr; b.preds_u8 = preds; b.maxprob_f32 = maxprob;
    if (conv_halo_preds_ok(dtype, b)) {
      logits_nhwc_ = nullptr;
      RUN(launch_conv(dtype, b, s_));
      return;
    }
  }
  RUN(launch_conv(dtype, a, s_));
  if (preds && !dry_)
    RUN(softmax_argmax_nhwc(logits_nhwc_, dtype, c.Cout_p, dec_out_.rows(), c.Cout, preds, nullptr, maxprob, s_));
}

int UNet::forward(const float* params, float* buffers, const float* x_nchw, float* logits_nchw, int B, int H, int W,
                  int training, void* ws, size_t ws_bytes, hipStream_t s) {
  // one-shot requests are consumed on EVERY path out of this call (a refused shape must not leave pointers to the caller's
  // already freed tensors behind for the next forward without logits)
  unsigned char* const preds_now = preds_req_;
  float* const maxprob_now = maxprob_req_;
  const bool reuse_now = reuse_req_;
  preds_req_ = nullptr; maxprob_req_ = nullptr; reuse_req_ = false;
  if ((H % 32) || (W % 32)) return -10;
  reuse_ = reuse_now && !training && last_valid_ && ws == last_ws_ && B == last_B_ && H == last_H_ && W == last_W_;
  preds_req_ = preds_now; maxprob_req_ = maxprob_now;   // consumed by head_fwd_impl below
  begin(ws, ws_bytes, s, false);
  fwd_common_begin(params, buffers, B, H, W, training);
  const int lazy_env = tune("FLAIR_LAZY_BN", 1);
  lazy_ok_ = lazy_env != 0;
  lazy_max_c_ = lazy_env >= 2 ? 32 : 16;
  if (!reuse_) pack_forward_weights();
  encoder_fwd_impl(x_nchw);
  decoder_fwd_impl();
  head_fwd_impl(logits_nchw);
  fwd_top_ = top_;
  reuse_ = false;
  last_valid_ = !training && !err_;   // (a training forward lays the arena out differently)
  last_ws_ = ws; last_B_ = B; last_H_ = H; last_W_ = W;
  return err_;
}

int UNet::encoder_forward(const float* params, float* buffers, const float* x_nchw, float* const feats[5], int B, int H,
                          int W, int training, void* ws, size_t ws_bytes, hipStream_t s) {
  invalidate_reuse();
  if ((H % 32) || (W % 32)) return -10;
  begin(ws, ws_bytes, s, false);
  fwd_common_begin(params, buffers, B, H, W, training);
  lazy_ok_ = false;   // split path: the decoder output crosses the boundary, every activation is materialised
  pack_forward_weights();
  encoder_fwd_impl(x_nchw);
  for (int i = 0; i < 5; ++i) {
    const Act& f = f_[i + 1];
    RUN(nhwc_to_nchw_f32(dtype, f.p, feats[i], f.N, f.C, f.H, f.W, f.C, nullptr, s_));
  }
  fwd_top_ = top_;
  return err_;
}

int UNet::decoder_forward(const float* params, float* buffers, const float* const feats[5], float* out_nchw, int B, int H,
                          int W, int training, void* ws, size_t ws_bytes, hipStream_t s) {
  invalidate_reuse();
  if (ws != base_ || B != B_ || H != H_ || W != W_) return -11;  // must follow encoder_forward on the same arena
  s_ = s; training_ = training; params_ = params; buffers_ = buffers;
  for (int i = 0; i < 5; ++i) {  // re-import: the caller may have modified feats[-1] (model.py:60)
    Act f = alloc_act(f_[i + 1].N, f_[i + 1].H, f_[i + 1].W, f_[i + 1].C);
    RUN(nchw_f32_to_nhwc(dtype, feats[i], f.p, f.N, f.C, f.H, f.W, f.C, s_));
    dec_in_[i + 1] = f;
  }
  Act saved[6];
  for (int i = 1; i <= 5; ++i) { saved[i] = f_[i]; f_[i] = dec_in_[i]; }
  decoder_fwd_impl();
  for (int i = 1; i <= 5; ++i) f_[i] = saved[i];
  RUN(nhwc_to_nchw_f32(dtype, dec_out_.p, out_nchw, dec_out_.N, 16, dec_out_.H, dec_out_.W, dec_out_.C, nullptr, s_));
  fwd_top_ = top_;
  return err_;
}

int UNet::head_forward(const float* params, const float* x_nchw, float* logits_nchw, int B, int H, int W, int training,
                       void* ws, size_t ws_bytes, hipStream_t s) {
  invalidate_reuse();
  if (ws != base_ || B != B_ || H != H_ || W != W_) return -11;
  s_ = s; params_ = params; training_ = training;
  Act x = alloc_act(B, H, W, 16);
  RUN(nchw_f32_to_nhwc(dtype, x_nchw, x.p, B, 16, H, W, 16, s_));
  dec_out_ = x;
  head_fwd_impl(logits_nchw);
  fwd_top_ = top_;
  return err_;
}

// ------------------------------------------------------------------------------------------ backward
// conv1 of every BasicBlock feeds conv2 only, decoder conv1 feeds conv2 only, decoder conv2 feeds the next block's
// upsample (or the head) only: for these units the data-gradient kernel of the single consumer writes the COMPLETE dz,
// so it can also do the first pass of their BatchNorm backward in its epilogue.
int UNet::sole_producer(const Act& a) const {
  if (!a.p) return -1;
  for (int i = 0; i < (int)units_.size(); ++i) {
    const Unit& u = units_[i];
    if (u.out.p != a.p) continue;
    const bool mask_from_y = u.relu && u.res_unit < 0 && !u.res.p;
    if (!mask_from_y) return -1;
    int consumers = 0;
    for (const Unit& v : units_) consumers += (v.in0.p == a.p) + (v.in1.p == a.p) + (v.res.p == a.p);
    for (int f = 1; f <= 5; ++f) consumers += (f_[f].p == a.p);  // encoder features also feed the decoder
    if (a.p == dec_out_.p) consumers += 1;                        // the head
    return consumers == 1 ? i : -1;
  }
  return -1;
}

// Residual units (conv2 of a BasicBlock: out = relu(bn(y) + x)) have two consumers inside the encoder, the next block's
// conv1 and its identity branch.  Backward visits the identity branch first (the BN-backward apply of that block's conv2
// writes dres) and conv1's data gradient last: an ACCUMULATING epilogue that completes the gradient.  Returns the
// producing unit when `a` is such an activation (mask from the ReLU output), else -1.
int UNet::residual_producer(const Act& a) const {
  if (!a.p) return -1;
  for (int f = 1; f <= 5; ++f)
    if (f_[f].p == a.p) return -1;   // encoder features also feed the decoder / the next stage
  if (a.p == dec_out_.p) return -1;
  for (int i = 0; i < (int)units_.size(); ++i) {
    const Unit& u = units_[i];
    if (u.out.p != a.p) continue;
    if (!u.relu || (u.res_unit < 0 && !u.res.p)) return -1;
    int convs_in = 0, res_in = 0;
    for (const Unit& v : units_) { convs_in += (v.in0.p == a.p) + (v.in1.p == a.p); res_in += (v.res.p == a.p); }
    return (convs_in == 1 && res_in == 1) ? i : -1;
  }
  return -1;
}

void UNet::attach_bn_reduce(ConvArgs& a, const Act& target) {
  static const int mode = tune("FLAIR_BNR_RES", 1);   // 0: never fuse the reduction of residual units
  int p = sole_producer(target);
  bool from_out = false;
  if (p < 0 && a.accumulate && mode) { p = residual_producer(target); from_out = p >= 0; }
  if (p < 0) return;
  Unit& u = units_[p];
  a.bnr_y = u.y.p; a.bnr_scale = u.scale; a.bnr_shift = u.shift; a.bnr_C = u.y.C;
  a.bnr_out = from_out ? u.out.p : nullptr;
  a.bnr_partial = reinterpret_cast<float*>(1);  // provisional, so that the geometry check sees the request
  // only where the data-gradient kernel is MFMA-bound (the extra read of y hides under it); the small-channel
  // halo kernels are HBM-bound and gain nothing over the separate streaming reduction
  if ((a.accumulate && !from_out) || u.y.C != a.out_ld || !conv_tile_epilogue_ok(dtype, a) || !conv_mfma_bound(dtype, a)) {
    a.bnr_y = nullptr; a.bnr_out = nullptr; a.bnr_partial = nullptr; a.bnr_scale = a.bnr_shift = nullptr; a.bnr_C = 0;
    return;
  }
  const int nblk = conv_grid_rows(dtype, a);
  a.bnr_partial = alloc_f((long)nblk * 2 * u.y.C);
  u.bnr_partial = a.bnr_partial;
  u.bnr_nblk = nblk;
  // the halo-GEMM epilogue has the ReLU mask in hand: it stores dz = gradient * mask, and the unit's own backward needs no mask
  // source, no copy for the identity branch and no separate BatchNorm-backward apply pass (unit_backward)
  if (bwd_fuse() && conv_hg_applicable(dtype, a)) { a.bnr_mask = 1; u.bnr_masked = true; }
}

bool UNet::bwd_fuse() const { return tune("FLAIR_BWD_FUSE", 1) != 0; }

// Backward of one conv->BN->(+res)->ReLU unit.  dout = gradient w.r.t. u.out (or w.r.t. the BN
// output when the unit was not materialised).  Produces parameter gradients, optionally the
// residual-branch gradient dz (dres) and the input gradient (into grad_of(in0) or dx_override).
//
// Round 3: when the kernel that completed dout stored it masked by this unit's ReLU (Unit::bnr_masked) and the unit's data
// gradient runs on the halo-GEMM, that kernel applies  dy = k1*dz + k2*y + k3  itself while it stages its halo and writes dy
// once for the weight-gradient kernel, which is then forked BEHIND it: bn_bwd_apply (read dz, read y, write dy; for residual
// units also read out, write dres) does not run for the unit.
void UNet::unit_backward(int ui, const void* dout, void* dres, bool dres_acc, bool need_dgrad, void* dx_override,
                         bool upcat) {
  Unit& u = units_[ui];
  const ConvDesc& c = convs[u.conv];
  const BnDesc& b = bns[u.bn];
  const long rows = u.y.rows();
  void* dy = alloc((size_t)u.y.elems() * dtype_size(dtype));
  const int pre_nblk = u.bnr_nblk;
  const bool masked = u.bnr_masked && pre_nblk > 0;
  float* partial = pre_nblk > 0 ? u.bnr_partial : alloc_f((long)bn_bwd_blocks(rows) * 2 * b.C);
  u.bnr_partial = nullptr; u.bnr_nblk = 0; u.bnr_masked = false;
  float* coef = alloc_f(3 * b.C);
  const void* acc_src = acc_src_next_;
  acc_src_next_ = nullptr;
  // ReLU mask: units without a residual recompute it from y (out > 0 <=> y*scale + shift > 0) and skip
  // reading the activation tensor in both backward passes
  const bool mask_from_y = u.relu && u.res_unit < 0 && !u.res.p;
  // weight gradient (described first: a unit without a data gradient may fold the BatchNorm-backward apply into it)
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.x0 = u.in0.p; w.x1 = u.in1.p; w.C0 = u.in0.C; w.C1 = u.in1.p ? u.in1.C : 0; w.up0 = u.up0 ? 1 : 0;
  w.N = u.in0.N; w.Hin = u.up0 ? u.in0.H * 2 : u.in0.H; w.Win = u.up0 ? u.in0.W * 2 : u.in0.W;
  w.Hout = u.y.H; w.Wout = u.y.W; w.R = c.R; w.S = c.S; w.stride = c.stride; w.pad = c.pad;
  w.dy = dy; w.dy_ld = u.y.C; w.Cout = c.Cout;
  // the stem: dy feeds nothing but the weight gradient, whose kernel stages it chunk by chunk — it applies the affine itself and
  // the tensor is never written (bn_bwd_apply: 804 MB of traffic for a 268 MB operand read once)
  const bool fuse_apply = !need_dgrad && !dres && mask_from_y && wgrad_bnapply_fusable(dtype, w);

  // ---- data gradient, described before anything is launched: gather-form conv over dy with the flipped / transposed pack
  enum { DG_NONE, DG_PLAIN, DG_UPCAT_TILE, DG_UPCAT_SPLIT, DG_PARITY } dg = DG_NONE;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  void *dx0 = nullptr, *dsk = nullptr, *dcat = nullptr;
  bool acc0 = false, acc1 = false;
  if (need_dgrad) {
    dg = DG_PLAIN;
    a.src0 = dy; a.C0 = u.y.C; a.C1 = 0; a.up0 = 0;
    a.N = u.y.N; a.Hin = u.y.H; a.Win = u.y.W;
    a.Hout = w.Hin; a.Wout = w.Win;
    a.R = c.R; a.S = c.S; a.out_mul = 1; a.pad = c.R - 1 - c.pad; a.in_div = c.stride;
    a.Cout = c.Cin_p;
    a.Kg = c.Kgd; a.Kpad = c.Kpad_d;
    a.w = base_ + c.wd;
    a.out_ld = c.Cin_p;
    if (upcat) {
      // decoder conv1: the input was cat([up2(in0), in1]).  Halo-tile kernels pool / split the gradient in their
      // epilogue; other shapes write the concatenated gradient and run the separate upcat_bwd pass.
      const int C0 = u.in0.C, C1 = u.in1.p ? u.in1.C : 0;
      dx0 = grad_of(u.in0, &acc0);
      dsk = C1 ? grad_of(u.in1, &acc1) : nullptr;
      a.pool_c0 = C0; a.out = dx0; a.out_ld = C0; a.accumulate = acc0 ? 1 : 0;
      a.out_skip = dsk; a.out_skip_ld = C1; a.skip_accumulate = acc1 ? 1 : 0;
      if (conv_tile_epilogue_ok(dtype, a)) {
        dg = DG_UPCAT_TILE;
        attach_bn_reduce(a, u.in0);   // the pooled half is the whole gradient of the previous block's output
      } else {
        dg = DG_UPCAT_SPLIT;
        dcat = alloc((size_t)u.y.rows() * (C0 + C1) * dtype_size(dtype));
        a.pool_c0 = 0; a.out_skip = nullptr; a.out = dcat; a.out_ld = C0 + C1; a.accumulate = 0;
      }
    } else {
      if (dx_override) {
        a.out = dx_override; a.accumulate = 0;
      } else {
        bool acc = false;
        a.out = grad_of(u.in0, &acc);
        a.accumulate = acc ? 1 : 0;
        if (acc_src) {
          // the identity branch's share arrives as a tensor of its own (the masked gradient of the block's output)
          ConvArgs t = a;
          t.accumulate = 1; t.acc_src = acc_src;
          if (!acc && conv_acc_src_ok(dtype, t)) {
            a = t;
          } else {   // some other kernel takes this layer: put the share into the buffer first
            if (acc) RUN(ew_add(dtype, a.out, acc_src, u.in0.elems(), s_));
            else if (!dry_ && !err_ && hipMemcpyAsync(a.out, acc_src, (size_t)u.in0.elems() * dtype_size(dtype), hipMemcpyDeviceToDevice, s_) != hipSuccess) err_ = -14;
            a.accumulate = 1;
          }
        }
        attach_bn_reduce(a, u.in0);
      }
      const bool doubled = c.stride == 2 && a.Hout == 2 * u.y.H && a.Wout == 2 * u.y.W;
      if (doubled && (c.parity_dgrad() || (c.R == 1 && c.S == 1 && c.pad == 0 && a.accumulate))) {
        // stride-2 data gradient by output parity class: stride-1 convolutions over dY, stores interleaved into dX.
        // A 1x1 stride-2 layer only reaches the (even, even) pixels; when accumulating, the other classes add nothing.
        dg = DG_PARITY;
        a.Hout = u.y.H; a.Wout = u.y.W; a.out_mul = 1; a.pad = 0; a.in_div = 1; a.out_sub = 1;
        if (c.R == 3) {   // the four classes (1, 2, 2 and 4 taps) ride in one launch, class = blockIdx.z
          a.ncls = 4;
          a.R = 2; a.S = 2;
          a.Kg = (c.Kg_cls[0] + c.Kg_cls[1] + c.Kg_cls[2] + c.Kg_cls[3]) / 4;  // mean over the classes (work accounting only)
          a.Kpad = c.Kpad_cls[3]; a.w = base_ + c.wd_cls[3];
          for (int cls = 0; cls < 4; ++cls) { a.cls_w[cls] = base_ + c.wd_cls[cls]; a.cls_kpad[cls] = c.Kpad_cls[cls]; }
        }
      }
    }
  }
  // fused BatchNorm-backward apply inside the data gradient: dout is dz already (masked), nobody else wants dz (dres)
  bool fuse_dy = false;
  if (masked && !dres && tune("FLAIR_BWD_FUSE", 1) >= 2 && (dg == DG_PLAIN || dg == DG_UPCAT_TILE)) {
    ConvArgs t = a;
    t.src0 = dout; t.ap_y = u.y.p; t.ap_coef = coef; t.ap_dy = dy;
    if (conv_bnapply_fusable(dtype, t)) { a = t; fuse_dy = true; }
  }

  // a masked gradient needs no mask source in the apply pass: the residual units' read of `out` and the others' recomputation
  // of relu'(y) go (bn_backward still needs one on paper when it runs the reduction itself, which it does not here: pre_nblk > 0)
  const bool nomask = masked && !dres && bwd_fuse();
  RUN(bn_backward(dtype, dout, (u.relu && !mask_from_y && !nomask) ? u.out.p : nullptr, u.y.p, u.mean, u.invstd, params_ + b.g_off,
                  rows, b.C, partial, coef, grads_ + b.g_off, grads_ + b.b_off, 0, (fuse_apply || fuse_dy) ? nullptr : dy, dres,
                  dres_acc ? 1 : 0, (mask_from_y && !nomask) ? u.scale : nullptr, (mask_from_y && !nomask) ? u.shift : nullptr,
                  pre_nblk, nomask ? 1 : 0, s_));
  if (fuse_apply) { w.dy = dout; w.fuse_y = u.y.p; w.fuse_coef = coef; w.fuse_msc = u.scale; w.fuse_msh = u.shift; }
  w.dw = grads_ + c.w_off; w.Cin_real = c.Cin; w.accumulate = 0;
  w.in_scale = u.in0.lz_scale; w.in_shift = u.in0.lz_shift;
  w.cus = side_cus(ui);
  w.partial = (float*)alloc(wgrad_workspace_bytes(dtype, w));
  auto launch_wg = [&]() {
    hipStream_t ws = wgrad_stream();   // dy is complete on s_; nothing later on s_ writes what this kernel reads
    RUN(launch_wgrad(dtype, w, ws));
  };
  if (!fuse_dy) launch_wg();
  switch (dg) {
    case DG_NONE: break;
    case DG_UPCAT_SPLIT:
      RUN(launch_conv(dtype, a, s_));
      RUN(upcat_bwd(dtype, dcat, dx0, acc0 ? 1 : 0, dsk, acc1 ? 1 : 0, u.y.N, u.y.H, u.y.W, u.in0.C, u.in1.p ? u.in1.C : 0, s_));
      break;
    default:
      RUN(launch_conv(dtype, a, s_));
  }
  if (fuse_dy) launch_wg();   // dy was written by the data gradient's staging pass
}

void UNet::head_bwd_impl(const void* dl) {
  const ConvDesc& c = convs.back();
  WgradArgs w;
  memset(&w, 0, sizeof(w));
  w.x0 = dec_out_.p; w.C0 = dec_out_.C; w.N = dec_out_.N; w.Hin = dec_out_.H; w.Win = dec_out_.W;
  w.Hout = dec_out_.H; w.Wout = dec_out_.W; w.R = 3; w.S = 3; w.stride = 1; w.pad = 1;
  w.dy = dl; w.dy_ld = c.Cout_p; w.Cout = c.Cout; w.dw = grads_ + c.w_off; w.Cin_real = c.Cin;
  w.in_scale = dec_out_.lz_scale; w.in_shift = dec_out_.lz_shift;
  w.partial = (float*)alloc(wgrad_workspace_bytes(dtype, w));
  // bias gradient = column sums of dl: inside the weight-gradient kernel where it stages dl anyway, else a pass of its own
  const bool fuse_db = wgrad_dbias_fusable(dtype, w);
  if (fuse_db) { w.dbias = grads_ + c.b_off; w.dbias_partial = alloc_f((long)WGRAD_DBIAS_ROWS * c.Cout_p); }
  {
    hipStream_t ws = wgrad_stream();
    RUN(launch_wgrad(dtype, w, ws));
  }
  if (!fuse_db) {
    const long rows = dec_out_.rows();
    float* partial = alloc_f((long)bn_bwd_blocks(rows) * c.Cout_p);
    RUN(colsum(dtype, dl, rows, c.Cout_p, c.Cout, partial, grads_ + c.b_off, s_));
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.src0 = dl; a.C0 = c.Cout_p; a.N = dec_out_.N; a.Hin = dec_out_.H; a.Win = dec_out_.W;
  a.Hout = dec_out_.H; a.Wout = dec_out_.W; a.R = 3; a.S = 3; a.out_mul = 1; a.pad = 1; a.in_div = 1;
  a.Cout = c.Cin_p; a.Kg = c.Kgd; a.Kpad = c.Kpad_d; a.w = base_ + c.wd;
  bool acc = false;
  a.out = grad_of(dec_out_, &acc);
  a.out_ld = c.Cin_p; a.accumulate = acc ? 1 : 0;
  attach_bn_reduce(a, dec_out_);
  RUN(launch_conv(dtype, a, s_));
}

void UNet::decoder_bwd_impl() {
  for (int i = 4; i >= 0; --i) {
    const int u1 = dec_units_begin_ + 2 * i, u2 = u1 + 1;
    unit_backward(u2, grad_peek(units_[u2].out), nullptr, false, true, nullptr);
    // conv1: gradient w.r.t. the virtual concat: 2x2 sum-pool -> x, rest -> skip (fused in the epilogue)
    unit_backward(u1, grad_peek(units_[u1].out), nullptr, false, true, nullptr, true);
  }
}

void UNet::encoder_bwd_impl() {
  // walk the encoder units backwards; layout per block: u1, [ud], u2
  int ui = enc_units_end_ - 1;
  const int nblk[4] = {3, 4, 6, 3};
  for (int L = 3; L >= 0; --L) {
    for (int b = nblk[L] - 1; b >= 0; --b) {
      const bool ds = (b == 0 && L > 0);
      const int u2 = ui, ud = ds ? ui - 1 : -1, u1 = ds ? ui - 2 : ui - 1;
      ui = u1 - 1;
      const Act& X = units_[u1].in0;
      const void* dO = grad_peek(units_[u2].out);
      // dO stored masked by the block's last ReLU (the data gradient that completed it, Unit::bnr_masked): it IS dz2, the
      // identity / downsample branch takes it as it stands and bn_bwd_apply need not write a copy
      const bool dz_ready = units_[u2].bnr_masked && units_[u2].bnr_nblk > 0 && bwd_fuse();
      if (!ds) {
        if (dz_ready) {
          unit_backward(u2, dO, nullptr, false, true, nullptr);
          acc_src_next_ = dO;   // conv1's data gradient (below): dX = dz2 + its own result
        } else {
          bool acc = false;
          void* dX = grad_of(X, &acc);
          unit_backward(u2, dO, dX, acc, true, nullptr);   // identity branch: dX (+)= dz2
        }
      } else {
        void* dz = dz_ready ? const_cast<void*>(dO) : alloc((size_t)units_[u2].y.elems() * dtype_size(dtype));
        unit_backward(u2, dO, dz_ready ? nullptr : dz, false, true, nullptr);
        // conv1 first: its stride-2 data gradient writes every pixel of dX, so the 1x1 stride-2 downsample branch
        // (no ReLU) then only accumulates into the (even, even) ones
        unit_backward(u1, grad_peek(units_[u1].out), nullptr, false, true, nullptr);
        unit_backward(ud, dz, nullptr, false, true, nullptr);
        continue;
      }
      unit_backward(u1, grad_peek(units_[u1].out), nullptr, false, true, nullptr);
    }
    stage_done(L + 1);
  }
  // maxpool: d f1 (+)= scatter(d pool)
  bool acc = false;
  void* df1 = grad_of(f_[1], &acc);
  {
    // the max-pool backward completes the gradient of the stem's ReLU output: it also runs the first pass of the stem's
    // BatchNorm backward on it (mask from y; the stem has no residual)
    Unit& u0 = units_[0];
    const bool fuse = tune("FLAIR_POOL_BNR", 1) && u0.relu && u0.res_unit < 0 && !u0.res.p && u0.y.C == 64;
    float* part = nullptr;
    const int nblk = B_ * f_[1].H;
    if (fuse) { part = alloc_f((long)nblk * 2 * u0.y.C); u0.bnr_partial = part; u0.bnr_nblk = nblk; }
    RUN(maxpool3x3s2_bwd(dtype, grad_peek(pool_), pool_idx_, df1, acc ? 1 : 0, B_, f_[1].H, f_[1].W, 64, s_,
                         fuse ? u0.y.p : nullptr, fuse ? u0.scale : nullptr, fuse ? u0.shift : nullptr, part));
  }
  unit_backward(0, df1, nullptr, false, false, nullptr);  // stem: no data gradient
  stage_done(0);
}

void UNet::stage_done(int stage) {
  const bool listened = stage_events_ && stage_events_[stage] && !dry_ && !err_;
  if (stage == 0) side_join();   // end of backward: everything is back on the caller's stream
  if (!listened) return;
  hipEvent_t ev = (hipEvent_t)stage_events_[stage];
  if (stage != 0 && side_pending_) {
    // The stage's weight gradients live on the side stream, the rest on s_.  Neither stream may wait for the other
    // here (that would serialise the weight-gradient tails behind the data-gradient chain six times per step): a third
    // stream waits for the current position of both and carries the stage event, which fires when both are done.
    if (!note_) {
      // LOW priority, like the side stream: a stream of the caller's priority level may share the caller's hardware queue, and its
      // wait for the side stream's position would then hold back every kernel the caller queues behind it
      int least = 0, greatest = 0;
      const int np = tune("FLAIR_NOTE_PRIO", 1);
      hipError_t ce;
      if (np && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        ce = hipStreamCreateWithPriority(&note_, hipStreamNonBlocking, np == 2 ? greatest : least);
      else
        ce = hipStreamCreateWithFlags(&note_, hipStreamNonBlocking);
      if (ce != hipSuccess) note_ = nullptr;
    }
    if (note_) {
      hipEvent_t e1 = fork_ev_[fork_next_++ % fork_ev_.size()];
      hipEvent_t e2 = fork_ev_[fork_next_++ % fork_ev_.size()];
      hipError_t e = hipEventRecord(e1, s_);
      if (e == hipSuccess) e = hipEventRecord(e2, side_);
      if (e == hipSuccess) e = hipStreamWaitEvent(note_, e1, 0);
      if (e == hipSuccess) e = hipStreamWaitEvent(note_, e2, 0);
      if (e == hipSuccess) e = hipEventRecord(ev, note_);
      if (e != hipSuccess) err_ = (int)e;
      return;
    }
    side_join();   // no third stream: fold the side stream back in and record on s_
  }
  const hipError_t e = hipEventRecord(ev, s_);
  if (e != hipSuccess) err_ = (int)e;
}

int UNet::backward(const float* params, const float* dlogits_nchw, const void* dlogits_nhwc, float* grads, void* ws,
                   size_t ws_bytes, hipStream_t s, void* const* stage_events) {
  invalidate_reuse();
  if (ws != base_ || !training_) return -11;
  s_ = s; params_ = params; grads_ = grads; top_ = fwd_top_;
  stage_events_ = stage_events;
  gbufs_.clear();
  pack_dgrad_weights();
  const void* dl = dlogits_nhwc;
  if (!dl) {
    const int ld = convs.back().Cout_p;
    void* t = alloc((size_t)dec_out_.rows() * ld * dtype_size(dtype));
    RUN(nchw_f32_to_nhwc(dtype, dlogits_nchw, t, B_, classes, H_, W_, ld, s_));
    dl = t;
  }
  head_bwd_impl(dl);
  stage_done(6);
  decoder_bwd_impl();
  stage_done(5);
  encoder_bwd_impl();
  stage_events_ = nullptr;
  return err_;
}

int UNet::head_backward(const float* params, const float* dlogits_nchw, float* dx_nchw, float* grads, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  invalidate_reuse();
  if (ws != base_ || !training_) return -11;
  s_ = s; params_ = params; grads_ = grads;
  pack_dgrad_weights();
  const int ld = convs.back().Cout_p;
  void* t = alloc((size_t)dec_out_.rows() * ld * dtype_size(dtype));
  RUN(nchw_f32_to_nhwc(dtype, dlogits_nchw, t, B_, classes, H_, W_, ld, s_));
  head_bwd_impl(t);
  side_join();
  RUN(nhwc_to_nchw_f32(dtype, grad_peek(dec_out_), dx_nchw, B_, 16, H_, W_, 16, nullptr, s_));
  return err_;
}

int UNet::decoder_backward(const float* params, const float* dout_nchw, float* const dfeats[5], float* grads, void* ws,
                           size_t ws_bytes, hipStream_t s) {
  invalidate_reuse();
  if (ws != base_ || !training_) return -11;
  s_ = s; params_ = params; grads_ = grads;
  pack_dgrad_weights();
  const Act& out = units_.back().out;  // decoder output (last unit)
  bool acc = false;
  void* g = grad_of(out, &acc);
  RUN(nchw_f32_to_nhwc(dtype, dout_nchw, g, out.N, 16, out.H, out.W, out.C, s_));
  decoder_bwd_impl();
  side_join();
  for (int i = 0; i < 5; ++i) {
    const Act& f = dec_in_[i + 1];
    RUN(nhwc_to_nchw_f32(dtype, grad_peek(f), dfeats[i], f.N, f.C, f.H, f.W, f.C, nullptr, s_));
  }
  return err_;
}

int UNet::encoder_backward(const float* params, const float* const dfeats[5], float* grads, void* ws, size_t ws_bytes,
                           hipStream_t s) {
  invalidate_reuse();
  if (ws != base_ || !training_) return -11;
  s_ = s; params_ = params; grads_ = grads;
  pack_dgrad_weights();
  for (int i = 0; i < 5; ++i) {
    const Act& f = f_[i + 1];
    bool acc = false;
    void* g = grad_of(f, &acc);
    if (acc) { if (!err_) err_ = -12; }
    RUN(nchw_f32_to_nhwc(dtype, dfeats[i], g, f.N, f.C, f.H, f.W, f.C, s_));
  }
  encoder_bwd_impl();
  return err_;
}

// Upper bound of the arena: dry run of the split sequence (superset of the fused one) on a SCRATCH executor, so that
// sizing a workspace between a training forward and its backward (an eval forward does that) leaves the recorded graph
// of the live handle alone.
size_t UNet::workspace_bytes(int B, int H, int W, int training) const {
  UNet scratch(in_channels, classes, dtype);
  return scratch.plan_bytes(B, H, W, training);
}

size_t UNet::plan_bytes(int B, int H, int W, int training) {
  begin(nullptr, 0, nullptr, true);
  fwd_common_begin(nullptr, nullptr, B, H, W, training);
  lazy_ok_ = false;   // the materialised (split) sequence is the upper bound
  encoder_fwd_impl(nullptr);
  Act saved[6];
  for (int i = 1; i <= 5; ++i) {
    dec_in_[i] = alloc_act(f_[i].N, f_[i].H, f_[i].W, f_[i].C);
    saved[i] = f_[i];
    f_[i] = dec_in_[i];
  }
  decoder_fwd_impl();
  for (int i = 1; i <= 5; ++i) f_[i] = saved[i];
  dec_out_ = alloc_act(B, H, W, 16);
  head_fwd_impl(nullptr);
  if (training) {
    alloc((size_t)dec_out_.rows() * convs.back().Cout_p * dtype_size(dtype));
    grads_ = nullptr;
    head_bwd_impl(nullptr);
    bool acc;
    grad_of(units_.back().out, &acc);
    decoder_bwd_impl();
    for (int i = 1; i <= 5; ++i) grad_of(f_[i], &acc);
    encoder_bwd_impl();
  }
  const size_t need = top_ + (1 << 20);
  dry_ = false;
  base_ = nullptr;
  units_.clear();
  gbufs_.clear();
  return need;
}

}  // namespace flair
