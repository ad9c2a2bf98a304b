#include "vae_net.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "conv_ops.hpp"

namespace mmvae {

#define MM_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ < 0) return rc__;  \
  } while (0)

static inline long align_up(long v, long a) { return (v + a - 1) / a * a; }
static inline int down_size(int H, int k, int s, int p) { return conv_down_size(H, k, s, p); }

// ------------------------------------------------------------------------------------------------ construction
void Net::add_entry(const std::string& name, std::initializer_list<int> shape, int kind, long off) {
  Entry e; e.name = name; e.ndim = (int)shape.size(); e.kind = kind; e.offset = off;
  int i = 0; for (int d : shape) e.shape[i++] = d;
  for (; i < 4; ++i) e.shape[i] = 1;
  entries.push_back(e);
}

ConvW Net::add_conv(const std::string& name, int D0, int D1, int k, int s, int p, bool pack, bool transposed) {
  ConvW w; w.off = n_params; w.D0 = D0; w.D1 = D1; w.k = k; w.s = s; w.p = p; w.packD = w.packU = -1; w.tr = transposed;
  add_entry(name, {D0, D1, k, k}, EK_PARAM, n_params);
  const long numel = (long)D0 * D1 * k * k;
  n_params += numel;
  if (numel > max_w) max_w = numel;
  if (pack) { w.packD = n_packed; n_packed += align_up(numel, 8); w.packU = n_packed; n_packed += align_up(numel, 8); }
  if (cfg.fp8 && pack && D0 >= 64 && D1 >= 64 && D0 % 64 == 0 && D1 % 64 == 0) {
    // candidate; settle_fp8() confirms it once the layer's place in the net (Hl) is known
    // static scale: PyTorch's default init is U(+-1/sqrt(D1*k*k)) for both weight layouts; bring that bound to ~16..32 (e4m3 normal range)
    w.fp8 = true;
    int e = 0; while ((float)(1 << e) < 16.f * std::sqrt((float)D1 * k * k)) ++e;
    w.wscale = (float)(1 << e);
  }
  return w;
}

// An fp8 layer's forward weights are e4m3 bytes that only the fp8 form of deep2_conv_kernel reads: keep the flag only where that kernel
// takes the layer's FORWARD launch (e.g. z = 192 makes decoder.conv1 a 3-chunk row, which it does not) -- the others stay bf16 layers.
// So do the layers on 2x2 / 4x4 maps whose forward the bf16 position-major kernel takes: it is faster there than the e4m3 deep2 form (with them
// in fp8 the mode ran at 0.96x of bf16 in round 4) -- fp8 arithmetic stays where it pays.
void Net::settle_fp8(ConvW& w) const {
  if (!w.fp8) return;
  const bool ok = w.Hl > 0 && !op_pos_fwd_takes(ConvGeom{w.D0, w.D1, w.k, w.s, w.p}, w.Hl, w.tr) && (w.tr ? op_deep2_up_ok(DT_BF16, ConvGeom{w.D0, w.D1, w.k, w.s, w.p}, w.Hl, w.Hl, 1, 1)
                                    : op_deep2_down_ok(DT_BF16, ConvGeom{w.D0, w.D1, w.k, w.s, w.p}, w.Hl, w.Hl, 1));
  if (!ok) { w.fp8 = false; w.wscale = 1.f; }
}

Bn Net::add_bn(const std::string& prefix, int C) {
  Bn b; b.C = C;
  b.g_off = n_params; add_entry(prefix + ".weight", {C}, EK_PARAM, n_params); n_params += C;
  b.b_off = n_params; add_entry(prefix + ".bias", {C}, EK_PARAM, n_params); n_params += C;
  b.rm_off = n_bnbuf; add_entry(prefix + ".running_mean", {C}, EK_BN_F32, n_bnbuf); n_bnbuf += C;
  b.rv_off = n_bnbuf; add_entry(prefix + ".running_var", {C}, EK_BN_F32, n_bnbuf); n_bnbuf += C;
  b.nbt_idx = n_nbt; { Entry e; e.name = prefix + ".num_batches_tracked"; e.ndim = 0; e.kind = EK_BN_I64; e.offset = n_nbt;
                       for (int i = 0; i < 4; ++i) e.shape[i] = 1; entries.push_back(e); } n_nbt += 1;
  b.ws = n_bnws; n_bnws += 7L * align_up(C, 4);
  return b;
}

Net::Net(const NetCfg& c) : cfg(c) {
  const int S = c.S;
  // ---- encoder (model.py:89-112)
  H1 = W1 = down_size(S, 5, 2, 2);
  stem = add_conv("encoder.conv1.weight", 32, c.in_ch, 5, 2, 2, false);
  bn0 = add_bn("encoder.bn1", 32);
  int inpl = 32, H = H1;
  const int planes[4] = {32, 64, 128, 256};
  const int nb = c.blocks < 1 ? 1 : c.blocks;
  for (int i = 0; i < 4; ++i) {
    for (int b = 0; b < nb; ++b) {
      // _make_layer (model.py:132-146): block 0 strides and carries the 1x1 downsample shortcut, the others keep the shape
      Block B;
      const std::string p = "encoder.layer" + std::to_string(i + 1) + "." + std::to_string(b) + ".";
      const int st = b == 0 ? 2 : 1;
      B.identity = b > 0;
      B.Cin = inpl; B.C = planes[i]; B.Hin = B.Win = H; B.Hout = B.Wout = down_size(H, 3, st, 1); B.Hmid = B.Wmid = B.Hout;
      B.c1 = add_conv(p + "conv1.weight", planes[i], inpl, 3, st, 1, true);       // conv3x3 (:29,:98-101)
      B.b1 = add_bn(p + "bn1", planes[i]);
      B.c2 = add_conv(p + "conv2.weight", planes[i], planes[i], 3, 1, 1, true);  // conv3x3 (:33)
      B.b2 = add_bn(p + "bn2", planes[i]);
      if (!B.identity) {
        B.cs = add_conv(p + "downsample.0.weight", planes[i], inpl, 1, 2, 0, true);  // conv1x1 stride 2 (:135-138)
        B.bs = add_bn(p + "downsample.1", planes[i]);
      }
      B.c1.Hl = B.Hin; B.c2.Hl = B.Hout; B.cs.Hl = B.Hin;
      inpl = planes[i]; H = B.Hout;
      enc.push_back(B);
    }
  }
  Hf = Wf = H;
  head_mu = add_conv("encoder.conv_mu.weight", c.z, 256, 1, 1, 0, false);
  if (c.need_logvar) head_lv = add_conv("encoder.conv_logvar.weight", c.z, 256, 1, 1, 0, false);
  const long hp = align_up((long)c.z * Hf * Wf * 256, 8);
  head_pack_mu = n_packed; n_packed += hp;
  head_pack_lv = n_packed; n_packed += hp;
  head_pack_dg = n_packed; n_packed += align_up(2L * c.z * 256, 8);
  stem_pack = n_packed; n_packed += 32L * 25 * 8;
  l1c2_flip = n_packed; n_packed += 32L * 9 * 32;          // encoder.layer1.conv2 as [cin][flipped tap][cout]: its data gradient as a forward conv
  tail_pack_f = n_packed; n_packed += 16L * 9 * 16;
  tail_pack_d = n_packed; n_packed += 16L * 9 * 8;
  // ---- decoder (model.py:154-179)
  dec_param_off = n_params;
  dstem = add_conv("decoder.conv1.weight", c.z, 128, 2, 2, 0, true, true);   // k2 s1 p0 on a 1x1 input == k2 s2 p0
  dstem.Hl = 2;
  dbn0 = add_bn("decoder.bn1", 128);
  nup = S > 32 ? 5 : 4;                                               // :169
  const int ups[5] = {128, 64, 32, 16, 16};
  int cin = 128; H = 2;
  for (int i = 0; i < nup; ++i) {
    for (int b = 0; b < nb; ++b) {
      // _make_up_block (model.py:196-209): the extra blocks come FIRST and the upsampling block last.  The reference builds the extra
      // ones as block(in, out) with the 2x ConvTranspose2d main path and an identity shortcut, which cannot run (shape mismatch,
      // :205-206); here they keep the stage's input shape: conv1x1 -> BN -> ReLU -> conv3x3 -> BN, + identity, ReLU.
      Block B;
      const std::string p = "decoder.uplayer" + std::to_string(i + 1) + "." + std::to_string(b) + ".";
      B.identity = b < nb - 1;
      if (B.identity) {
        B.Cin = cin; B.C = cin; B.Hin = B.Win = H; B.Hmid = B.Wmid = H; B.Hout = B.Wout = H;
        B.c1 = add_conv(p + "conv1.weight", cin, cin, 1, 1, 0, true);
        B.b1 = add_bn(p + "bn1", cin);
        B.c2 = add_conv(p + "conv2.weight", cin, cin, 3, 1, 1, true);           // Conv2d (out,in,3,3): the shape-preserving main path
        B.b2 = add_bn(p + "bn2", cin);
        B.c1.Hl = H; B.c2.Hl = H;
      } else {
        B.Cin = cin; B.C = ups[i]; B.Hin = B.Win = H; B.Hmid = B.Wmid = H; B.Hout = B.Wout = 2 * H;
        B.c1 = add_conv(p + "conv1.weight", ups[i], cin, 1, 1, 0, true);        // 1x1 (:60)
        B.b1 = add_bn(p + "bn1", ups[i]);
        B.c2 = add_conv(p + "conv2.weight", ups[i], ups[i], 4, 2, 1, true, true);      // ConvT k4 s2 p1 (:62-65)
        B.b2 = add_bn(p + "bn2", ups[i]);
        B.cs = add_conv(p + "upsample.0.weight", cin, ups[i], 4, 2, 1, true, true);    // ConvT k4 s2 p1 (:198-201)
        B.bs = add_bn(p + "upsample.1", ups[i]);
        B.c1.Hl = H; B.c2.Hl = 2 * H; B.cs.Hl = 2 * H;
        cin = ups[i]; H = 2 * H;
      }
      dec.push_back(B);
    }
  }
  Sd = H;
  // fp8 mode: the two largest activations of the step (the last up-block's branch outputs at 64x64: 671 MB each in bf16 at 5120 frames, written
  // once and read three times) are STORED as e4m3 bytes where the four stream kernels that touch them run (convT4_stream -> tail_fwd_stream,
  // tail_reduce_mfma, join_bwd_stream).  A static weight scale of 16 puts their values mid-range; the BatchNorms behind absorb it exactly
  // (the same mechanism as an fp8 layer's weight scale).
  if (cfg.fp8 && !dec.empty()) {
    Block& L = dec.back();
    constexpr bool env = true;
    if (env && !L.identity && L.C == 16 && L.Cin == 16 && L.Hin == 32 && Sd == 64 && c.out_ch == 1 && convT4_stream_ok(DT_BF16, 16, 16, 4, 2, 1, 32, 32) &&
        tail_fwd_stream_ok(DT_BF16, 1, 64, 64) && join_bwd_stream_ok(DT_BF16, 1, 16, 32, 64)) {
      store8 = true;
      L.c2.wscale = 16.f; L.cs.wscale = 16.f;
    }
  }
  settle_fp8(dstem);
  for (Block& B : enc) { settle_fp8(B.c1); settle_fp8(B.c2); settle_fp8(B.cs); }
  for (Block& B : dec) { settle_fp8(B.c1); settle_fp8(B.c2); settle_fp8(B.cs); }
  tail = add_conv("decoder.conv2.weight", c.out_ch, 16, 3, 1, 1, false);
  tail_bias = n_params; add_entry("decoder.conv2.bias", {c.out_ch}, EK_PARAM, n_params); n_params += c.out_ch;
  bn_out = add_bn("decoder.bn2", c.out_ch);
}

Net::~Net() {
  if (side_state_ != 1) return;
  // work still queued on the side streams belongs to buffers the caller may free next: drain first
  if (side_) { (void)hipStreamSynchronize(side_); (void)hipStreamDestroy(side_); }
  for (hipEvent_t e : ev_) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : blk_ev_) if (e) (void)hipEventDestroy(e);
  if (gram_ev_) (void)hipEventDestroy(gram_ev_);
  (void)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ planning
size_t Net::workspace_bytes(int N) { return plan(N).bytes; }

const Plan& Net::plan(int N) {
  if (plan_.N == N) return plan_;
  Plan P; P.N = N;
  long cur = 0, maxact = 0;
  const long e = (long)esz();
  auto take = [&](long bytes) { long o = cur; cur = align_up(cur + std::max(bytes, 16L), 256); return o; };
  auto act = [&](long elems) { maxact = std::max(maxact, elems * e); return take(elems * e); };
  P.x_t = act((long)N * cfg.in_ch * cfg.S * cfg.S);
  P.y0 = act((long)N * H1 * W1 * 32);
  for (Block& B : enc) {
    const long n = (long)N * B.Hout * B.Wout * B.C;
    B.y1 = act(n); B.y2 = act(n); B.ys = B.identity ? 0 : act(n); B.out = act(n);
  }
  P.enc_t = act((long)N * cfg.z);
  P.y0d = act((long)N * 4 * 128);
  P.act0d = cfg.blocks > 1 ? act((long)N * 4 * 128) : 0;
  for (Block& B : dec) {
    const long n = (long)N * B.Hout * B.Wout * B.C;
    B.y1 = act((long)N * B.Hin * B.Win * B.C); B.y2 = act(n); B.ys = B.identity ? 0 : act(n); B.out = act(n);
  }
  P.cvec = take(512 * 4);
  P.r_raw = take((long)N * cfg.out_ch * Sd * Sd * 4);
  P.d_raw = take((long)N * cfg.out_ch * Sd * Sd * 4);
  // im2col of the 1-channel image; in_channels > 1: the image as NHWC with 16 zero-padded channels (the generic weight gradient's G operand)
  P.col = act(cfg.in_ch == 1 ? (long)N * H1 * W1 * 32 : (long)N * cfg.S * cfg.S * 16);
  P.packed = take(n_packed * e);
  P.bnws = take(n_bnws * 4);
  P.partials = take(2 * kPartialFloats * 4);       // second half: the shortcut branch running on the side stream
  P.syncbuf = take(2 * 1024 * 4);
  for (int i = 0; i < 2; ++i) P.g[i] = take(maxact);
  for (int i = 0; i < 2; ++i) { P.dy1[i] = take(maxact); P.dy2[i] = take(maxact); P.dys[i] = take(maxact); }
  for (int i = 0; i < 2; ++i) {        // private dy sets of the two encoder blocks the backward pass visits first
    const Block& B = enc[enc.size() - 2 + i];
    const long nb = (long)N * B.Hout * B.Wout * B.C * e;
    P.edy1[i] = take(nb); P.edy2[i] = take(nb); P.edys[i] = take(nb);
  }
  P.da1 = take(maxact);
  P.dh = take((long)N * 2 * cfg.z * e);
  P.wscratch = take((long)kWgradScratchBytes);
  P.wscratch2 = take(2L * 1024 * 4096 * 4);           // partial images of the fused passes on the caller's stream: 512 blocks x ([16][32][16] + [16][32]) f32 (wgrad_stream), 2 x 1024 x [16][16][16] (join_bwd_stream)
  P.stem_R = take(1024 * 8);
  P.stem_gram = take(1024L * stem_bwd_part_floats() * 4);
  P.bytes = (size_t)cur;
  plan_ = P;
  return plan_;
}

int Net::fill_consts(char* base, hipStream_t s) {
  float* c = reinterpret_cast<float*>(base + plan_.cvec);
  MM_TRY(launch_fill_f32(c, 1.f, 256, s));
  return launch_fill_f32(c + 256, 0.f, 256, s);
}

float* Net::bnf(const Bn& bn, char* base, int which) const {
  return reinterpret_cast<float*>(base + plan_.bnws) + bn.ws + (long)which * align_up(bn.C, 4);
}

// ------------------------------------------------------------------------------------------------ conv helpers
static inline ConvGeom geom(const ConvW& w) { return ConvGeom{w.D0, w.D1, w.k, w.s, w.p}; }

// fwd = 1: this pack feeds the conv's FORWARD direction (an fp8 layer then gets e4m3 bytes); every pack of an fp8 layer is scaled
// the pack that feeds a conv's FORWARD direction (down for Conv2d, up for ConvTranspose2d) is e4m3 bytes on an fp8 layer; every pack
// of an fp8 layer carries its weight scale
// The up form of a 1x1 stride-2 shortcut has stride phases without a tap: in this net it only ever accumulates into (or rides along
// with) the main path's data gradient, so those phases are skipped.
// (an fp8 layer's FORWARD pack is e4m3: fragment-major when the fp8 form of the kernel takes the shape)
int Net::frag_down(const ConvW& w) const { return w.Hl > 0 ? op_frag_down(dt(), geom(w), w.Hl, w.Hl, (w.fp8 && !w.tr) ? 1 : 0) : 0; }
int Net::frag_up(const ConvW& w) const { return w.Hl > 0 ? op_frag_up(dt(), geom(w), w.Hl, w.Hl, 1, (w.fp8 && w.tr) ? 1 : 0) : 0; }
int Net::pack_down(const ConvW& w, const float* params, char* base, hipStream_t s) {
  return op_pack_down(dt(), geom(w), params + w.off, base + plan_.packed + w.packD * (long)esz(), s, w.wscale, (w.fp8 && !w.tr) ? 1 : 0,
                      frag_down(w));
}
int Net::pack_up(const ConvW& w, const float* params, char* base, hipStream_t s) {
  return op_pack_up(dt(), geom(w), params + w.off, base + plan_.packed + w.packU * (long)esz(), s, w.wscale, (w.fp8 && w.tr) ? 1 : 0,
                    frag_up(w));
}
int Net::run_down(const ConvW& w, char* base, int N, const void* L, int Hl, int Wl, void* S, int Hs, int Ws,
                  const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, int out_dt, hipStream_t s,
                  const ConvW* w2, const void* x2) {
  SecondSrc q;
  if (w2) { q.x2 = x2; q.w2 = base + plan_.packed + w2->packU * (long)esz(); q.Cin2 = w2->D0; }
  q.fp8 = (w.fp8 && !w.tr) ? 1 : 0;               // a Conv2d's forward
  q.wfrag = frag_down(w); q.wfrag2 = w2 ? frag_up(*w2) : 0;
  return op_run_down(dt(), out_dt, geom(w), base + plan_.packed + w.packD * (long)esz(), N, L, Hl, Wl, S, Hs, Ws, pro_s, pro_b, relu,
                     stats, accumulate, s, q);
}
int Net::run_up(const ConvW& w, char* base, int N, const void* S, int Hs, int Ws, void* L, int Hl, int Wl,
                const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, hipStream_t s,
                const ConvW* w2, const void* x2) {
  SecondSrc q;
  if (w2) { q.x2 = x2; q.w2 = base + plan_.packed + w2->packU * (long)esz(); q.Cin2 = w2->D0; }
  q.fp8 = (w.fp8 && w.tr) ? 1 : 0;                // a ConvTranspose2d's forward
  q.wfrag = frag_up(w); q.wfrag2 = w2 ? frag_up(*w2) : 0;
  return op_run_up(dt(), geom(w), base + plan_.packed + w.packU * (long)esz(), N, S, Hs, Ws, L, Hl, Wl, pro_s, pro_b, relu, stats,
                   accumulate, s, q);
}
hipStream_t Net::wgrad_stream(hipStream_t s) {
  if (side_state_ == 0) {
    side_state_ = 1;
    if (side_state_ == 1) {
      bool ok = hipStreamCreateWithFlags(&side_, hipStreamNonBlocking) == hipSuccess;
      for (int i = 0; i < kForkEvents && ok; ++i) ok = hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming) == hipSuccess;
      for (int i = 0; i < 16 && ok; ++i) ok = hipEventCreateWithFlags(&blk_ev_[i], hipEventDisableTiming) == hipSuccess;
      if (ok) ok = hipEventCreateWithFlags(&gram_ev_, hipEventDisableTiming) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); side_state_ = -1; }
    }
  }
  return side_state_ == 1 ? side_ : s;
}
// kForkEvents > 2 x (forks + joins of one backward pass): no event is re-recorded while a wait on it may be pending
int Net::side_fork(hipStream_t s) {
  if (wgrad_stream(s) == s) return MMVAE_OK;
  hipEvent_t e = ev_[evi_++ % kForkEvents];
  bool ok = hipEventRecord(e, s) == hipSuccess && hipStreamWaitEvent(side_, e, 0) == hipSuccess;
  if (!ok) { set_error("side stream fork failed"); return MMVAE_ERR_HIP; }
  return MMVAE_OK;
}
int Net::side_join(hipStream_t s) {
  if (wgrad_stream(s) == s) return MMVAE_OK;
  hipEvent_t e = ev_[evi_++ % kForkEvents];
  bool ok = hipEventRecord(e, side_) == hipSuccess && hipStreamWaitEvent(s, e, 0) == hipSuccess;
  if (!ok) { set_error("side stream join failed"); return MMVAE_ERR_HIP; }
  return MMVAE_OK;
}

// measured: im2col + 1x1 weight gradient 0.39 ms, planar-G patch-tile path 0.42 ms (MMVAE_STEM_PLANAR=1)
// MMVAE_STEM_FUSED=0 restores reduce -> apply -> im2col -> wgrad
bool Net::stem_bwd_fused() const {
  constexpr bool env = true;
  return env && cfg.in_ch == 1 && stem_bwd_fusable(cfg.S);
}

static bool stem_im2col_path() {
  constexpr bool v = true;
  return v;
}

int Net::side_mark(int slot) {
  if (side_state_ != 1) return MMVAE_OK;
  bool ok = hipEventRecord(blk_ev_[slot & 15], side_) == hipSuccess;
  if (!ok) { set_error("side stream mark failed"); return MMVAE_ERR_HIP; }
  return MMVAE_OK;
}
int Net::side_wait_mark(int slot, hipStream_t s) {
  if (side_state_ != 1) return MMVAE_OK;
  bool ok = hipStreamWaitEvent(s, blk_ev_[slot & 15], 0) == hipSuccess;
  if (!ok) { set_error("side stream wait failed"); return MMVAE_ERR_HIP; }
  return MMVAE_OK;
}

int Net::run_wgrad(const ConvW& w, int N, const void* P, int Hs, int Ws, const float* proP_s, const float* proP_b,
                   const void* G, int Hl, int Wl, const float* proG_s, const float* proG_b, float* grads, hipStream_t s) {
  // (measured in round 3: a SECOND side stream with its own scratch, the weight gradients alternating between the two, is slower --
  // 7.74 vs 7.63 ms per step: the phases where the side stream lags are bandwidth-bound, two wgrad kernels at once only thrash)
  return op_run_wgrad(dt(), geom(w), N, P, Hs, Ws, proP_s, proP_b, 1, G, Hl, Wl, proG_s, proG_b, 1, grads + w.off, s, wscratch_, w.wscale);
}

// conv1 (3x3 s2) + 1x1 s2 shortcut weight gradients in one stream pass: encoder.layer1's shape (bf16, 32 -> 32 channels, 32x32 -> 16x16)
static bool wgrad_pair_ok(int dt, const ConvGeom& g, const ConvGeom& gs, int Hout, int Hin) {
  constexpr bool env = true;
  return env && dt == DT_BF16 && g.k == 3 && g.s == 2 && g.p == 1 && gs.k == 1 && gs.s == 2 && gs.p == 0 && g.D0 == 32 && g.D1 == 32 && gs.D0 == 32 &&
         gs.D1 == 32 && Hout == 16 && Hin == 32;
}

// encoder.layer1.conv2's data gradient + bn1's backward sums in one stream pass (bf16, 32 -> 32 channels, 16x16 maps, one block per stage)
bool Net::l1_dgrad_stream() const {
  constexpr bool env = true;
  if (!env || enc.empty()) return false;
  const Block& B = enc[0];
  return !B.identity && !B.c2.fp8 && conv3_stream_ok(dt(), B.C, B.C, B.c2.k, B.c2.s, B.c2.p, B.Hout, B.Wout);
}

// The stem's backward recomputes its incoming gradient (encoder.layer1's conv1 3x3 s2 + 1x1 s2 shortcut data gradients) instead of reading it:
// bf16, 64x64 images, the reference's first stage (32 -> 32 channels, strided block with a 1x1 shortcut), row-major packed weights
bool Net::stem_dg_fused() const {
  if (enc.empty() || !stem_bwd_fused() || !stem_bwd_dg_ok(dt(), cfg.S)) return false;
  const Block& B = enc[0];
  return !B.identity && B.Cin == 32 && B.C == 32 && B.c1.k == 3 && B.c1.s == 2 && B.c1.p == 1 && B.cs.k == 1 && B.cs.s == 2 && B.cs.p == 0 &&
         !B.c1.fp8 && !B.cs.fp8 && B.c1.wscale == 1.f && B.cs.wscale == 1.f && frag_up(B.c1) == 0 && frag_up(B.cs) == 0 && H1 == 32 && B.Hout == 16;
}

bool Net::tail_fwd_fused() const {
  constexpr bool env = true;
  constexpr bool bwd_env = true;
  // N does not enter the geometry checks beyond the tile count limit, which the plan's maximum batch already satisfies
  return env && bwd_env && !dec.empty() && dec.back().C == 16 && tail_fwd_fusable(dt(), cfg.out_ch, 1, Sd, Sd) &&
         tail_join_fusable(dt(), cfg.out_ch, 1, Sd, Sd);
}

// SyncBN: sum the partial rows locally, let the host's collective sum the row over the ranks (stream-ordered), and hand the
// finalize kernels that one row.  Two operand slots: the caller's stream and the side stream may both have one in flight.
int Net::sync_rows(char* base, const float* partials, int nparts, int width, hipStream_t s, float** out, int row_stride) {
  if (width > 1024) { set_error("sync_bn: %d statistics per BatchNorm > 1024", width); return MMVAE_ERR_UNSUPPORTED; }
  float* buf = reinterpret_cast<float*>(base + plan_.syncbuf) + (s == side_ ? 1024 : 0);
  MM_TRY(launch_partial_rowsum(partials, nparts, width, buf, s, row_stride));
  if (comm_) MM_TRY(comm_allreduce_sum((s == side_ && comm_side_) ? comm_side_ : comm_, buf, width, s));
  else if (ar_fn_(buf, width, s, ar_user_) != 0) { set_error("sync_bn: the all-reduce callback failed"); return MMVAE_ERR_ARG; }
  *out = buf;
  return MMVAE_OK;
}

int Net::bn_train(const Bn& bn, const float* params, float* bnbuf, long long* nbt, char* base, int nparts, double count, hipStream_t s,
                  long part_off, float in_scale) {
  BnFinalizeArgs a;
  a.in_scale = in_scale;
  a.partials = reinterpret_cast<const float*>(base + plan_.partials) + part_off; a.nparts = nparts; a.C = bn.C; a.count = count;
  if (sync_bn_on()) {      // statistics over the global batch (equal shards per rank)
    float* row = nullptr;
    MM_TRY(sync_rows(base, a.partials, nparts, 2 * bn.C, s, &row));
    a.partials = row; a.nparts = 1; a.count = count * ar_world_;
  }
  a.gamma = params + bn.g_off; a.beta = params + bn.b_off;
  a.running_mean = bnbuf ? bnbuf + bn.rm_off : nullptr; a.running_var = bnbuf ? bnbuf + bn.rv_off : nullptr;
  a.nbt = nbt ? nbt + bn.nbt_idx : nullptr;
  a.mean = bnf(bn, base, 0); a.istd = bnf(bn, base, 1); a.scale = bnf(bn, base, 2); a.shift = bnf(bn, base, 3);
  a.momentum = 0.1f; a.eps = 1e-5f;
  return launch_bn_finalize(a, s);
}

int Net::fold_bn_eval(int which, const float* params, const float* bnbuf, char* base, hipStream_t s) {
  std::vector<BnFoldEntry> tab;
  auto add = [&](const Bn& bn, float in_scale) {
    BnFoldEntry e;
    e.g_off = (int)bn.g_off; e.b_off = (int)bn.b_off; e.rm_off = (int)bn.rm_off; e.rv_off = (int)bn.rv_off;
    e.scale_off = (int)(bn.ws + 2L * align_up(bn.C, 4)); e.shift_off = (int)(bn.ws + 3L * align_up(bn.C, 4)); e.C = bn.C; e.in_scale = in_scale;
    tab.push_back(e);
  };
  if (which == 0) add(bn0, 1.f); else add(dbn0, dstem.wscale);
  for (const Block& B : which == 0 ? enc : dec) {
    add(B.b1, B.c1.wscale); add(B.b2, B.c2.wscale);
    if (!B.identity) add(B.bs, B.cs.wscale);
  }
  if (which == 1) add(bn_out, 1.f);
  MM_TRY(launch_bn_fold_eval(tab.data(), (int)tab.size(), params, bnbuf, reinterpret_cast<float*>(base + plan_.bnws), 1e-5f, s));
  eval_folded_ = true;
  return MMVAE_OK;
}

// eval mode: fold_bn_eval() has written every (scale, shift) pair of the entry point already
int Net::bn_eval(const Bn&, const float*, const float*, char*, hipStream_t, float) {
  if (eval_folded_) return MMVAE_OK;
  set_error("bn_eval: the entry point did not fold its BatchNorms"); return MMVAE_ERR_ARG;
}

BnBwdFinalizeArgs Net::bwd_finalize_args(const Bn& bn, const float* params, float* grads, char* base, const float* partials, int nparts, int ny,
                                         int which, double count) const {
  const Net& net = *this;
  BnBwdFinalizeArgs a;
  a.partials = partials; a.nparts = nparts; a.C = bn.C; a.which = which; a.ny = ny;
  a.count = count; a.gamma = params + bn.g_off; a.mean = net.bnf(bn, base, 0); a.istd = net.bnf(bn, base, 1);
  a.dgamma = grads + bn.g_off; a.dbeta = grads + bn.b_off;
  a.coefA = net.bnf(bn, base, 4); a.coefB = net.bnf(bn, base, 5); a.coefC = net.bnf(bn, base, 6);
  return a;
}
// SyncBN backward: dgamma / dbeta stay the rank's own sums (the gradient all-reduce adds the ranks up), the coefficients of
// dx = A*g + B*y + C use the sums over the global batch -- so the local finalize runs first and a second one, fed with the
// all-reduced row, overwrites the coefficients.
int Net::bn_backward_coefs(const Bn& bn, const float* params, float* grads, char* base, int nparts, int ny, int which, double count,
                           hipStream_t s, float* dbias_conv) {
  const float* part = reinterpret_cast<const float*>(base + plan_.partials);
  BnBwdFinalizeArgs l = bwd_finalize_args(bn, params, grads, base, part, nparts, ny, which, count);
  if (!sync_bn_on()) l.dbias_conv = dbias_conv;
  MM_TRY(launch_bn_bwd_finalize(l, s));
  if (!sync_bn_on()) return MMVAE_OK;
  float* row = nullptr;
  MM_TRY(sync_rows(base, part, nparts, (1 + ny) * bn.C, s, &row));
  BnBwdFinalizeArgs g = bwd_finalize_args(bn, params, grads, base, row, 1, ny, which, count * ar_world_);
  g.dgamma = nullptr; g.dbeta = nullptr;
  // the gradient all-reduce sums the ranks: every rank contributes 1/world of the global closed form
  g.dbias_conv = dbias_conv; g.dbias_scale = 1.f / (float)ar_world_;
  return launch_bn_bwd_finalize(g, s);
}
// the two BatchNorms of a residual join (partials rows: sum g, sum g*y2, sum g*ys) in one launch
int Net::bn_backward_coefs_join(const Bn& b2, const Bn& bs, const float* params, float* grads, char* base, int nparts, double count,
                                hipStream_t s) {
  const float* part = reinterpret_cast<const float*>(base + plan_.partials);
  MM_TRY(launch_bn_bwd_finalize2(bwd_finalize_args(b2, params, grads, base, part, nparts, 2, 0, count),
                                 bwd_finalize_args(bs, params, grads, base, part, nparts, 2, 1, count), s));
  if (!sync_bn_on()) return MMVAE_OK;
  float* row = nullptr;
  MM_TRY(sync_rows(base, part, nparts, 3 * b2.C, s, &row));
  BnBwdFinalizeArgs g2 = bwd_finalize_args(b2, params, grads, base, row, 1, 2, 0, count * ar_world_);
  BnBwdFinalizeArgs gs = bwd_finalize_args(bs, params, grads, base, row, 1, 2, 1, count * ar_world_);
  g2.dgamma = g2.dbeta = gs.dgamma = gs.dbeta = nullptr;
  return launch_bn_bwd_finalize2(g2, gs, s);
}

// ------------------------------------------------------------------------------------------------ weight re-packs per entry point
// (recorded by launch_pack() while a batch is open; see pack_batch_begin/flush)
int Net::packs_enc_fwd(const float* params, char* base, hipStream_t s) {
  const Plan& P = plan_;
  {
    const int cpad = dt() == DT_F32 ? 4 : 8;
    PackArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.src = params + stem.off; pa.dst = base + P.packed + stem_pack * (long)esz();
    pa.cols = 32; pa.K = cpad; pa.K_valid = cfg.in_ch; pa.ntaps = 25; pa.s_col = 25 * cfg.in_ch; pa.s_k = 25; pa.scale = 1.f;
    for (int t = 0; t < 25; ++t) pa.tap_off[t] = t;
    MM_TRY(launch_pack(dt(), pa, s));
  }
  for (const Block& B : enc) {
    MM_TRY(pack_down(B.c1, params, base, s));
    MM_TRY(pack_down(B.c2, params, base, s));
    if (!B.identity) MM_TRY(pack_down(B.cs, params, base, s));
  }
  const int nt = Hf * Wf;
  for (int h = 0; h < (cfg.need_logvar ? 2 : 1); ++h) {
    const ConvW& hw = h == 0 ? head_mu : head_lv;
    const long poff = h == 0 ? head_pack_mu : head_pack_lv;
    PackArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.src = params + hw.off; pa.dst = base + P.packed + poff * (long)esz();
    pa.cols = cfg.z; pa.K = 256; pa.ntaps = nt; pa.s_col = 256; pa.s_k = 1; pa.scale = 1.0f / nt;
    MM_TRY(launch_pack(dt(), pa, s));
  }
  return MMVAE_OK;
}

int Net::packs_enc_bwd(const float* params, char* base, hipStream_t s) {
  const Plan& P = plan_;
  const int Ch = cfg.need_logvar ? 2 * cfg.z : cfg.z;
  PackArgs pa; std::memset(&pa, 0, sizeof(pa));
  pa.src = params + head_mu.off; pa.dst = base + P.packed + head_pack_dg * (long)esz();
  pa.cols = 256; pa.K = Ch; pa.ntaps = 1; pa.s_col = 1; pa.s_k = 256; pa.scale = 1.0f / (Hf * Wf);
  MM_TRY(launch_pack(dt(), pa, s));
  for (const Block& B : enc) {
    MM_TRY(pack_up(B.c2, params, base, s));
    MM_TRY(pack_up(B.c1, params, base, s));
    if (!B.identity) MM_TRY(pack_up(B.cs, params, base, s));
  }
  if (l1_dgrad_stream()) {
    // dx[n,h,w,ci] = sum_{kh,kw,co} dy[n, h+1-kh, w+1-kw, co] W[co][ci][kh][kw]: a forward 3x3 conv over dy with weights [ci][tap' = 8 - tap][co]
    const ConvW& w = enc[0].c2;
    PackArgs pf; std::memset(&pf, 0, sizeof(pf));
    pf.src = params + w.off; pf.dst = base + P.packed + l1c2_flip * (long)esz();
    pf.cols = 32; pf.K = 32; pf.ntaps = 9; pf.s_col = 9; pf.s_k = 32 * 9; pf.scale = w.wscale;
    for (int t = 0; t < 9; ++t) pf.tap_off[t] = 8 - t;
    MM_TRY(launch_pack(dt(), pf, s));
  }
  return MMVAE_OK;
}

int Net::packs_dec_fwd(const float* params, char* base, hipStream_t s) {
  const Plan& P = plan_;
  MM_TRY(pack_up(dstem, params, base, s));
  for (const Block& B : dec) {
    MM_TRY(pack_down(B.c1, params, base, s));
    if (B.identity) MM_TRY(pack_down(B.c2, params, base, s));       // 3x3 Conv2d
    else { MM_TRY(pack_up(B.c2, params, base, s)); MM_TRY(pack_up(B.cs, params, base, s)); }
  }
  PackArgs pa; std::memset(&pa, 0, sizeof(pa));
  pa.src = params + tail.off; pa.dst = base + P.packed + tail_pack_f * (long)esz();
  pa.cols = 16; pa.cols_valid = cfg.out_ch; pa.K = 16; pa.ntaps = 9; pa.s_col = 144; pa.s_k = 9; pa.scale = 1.f;
  for (int t = 0; t < 9; ++t) pa.tap_off[t] = t;
  MM_TRY(launch_pack(dt(), pa, s));
  return MMVAE_OK;
}

int Net::packs_dec_bwd(const float* params, char* base, bool need_denc, hipStream_t s) {
  const Plan& P = plan_;
  PackArgs pa; std::memset(&pa, 0, sizeof(pa));
  pa.src = params + tail.off; pa.dst = base + P.packed + tail_pack_d * (long)esz();
  pa.cols = 16; pa.K = 8; pa.K_valid = cfg.out_ch; pa.ntaps = 9; pa.s_col = 9; pa.s_k = 144; pa.scale = 1.f;
  for (int t = 0; t < 9; ++t) pa.tap_off[t] = t;
  MM_TRY(launch_pack(dt(), pa, s));
  for (const Block& B : dec) {
    MM_TRY(pack_up(B.c1, params, base, s));
    if (B.identity) MM_TRY(pack_up(B.c2, params, base, s));
    else { MM_TRY(pack_down(B.c2, params, base, s)); MM_TRY(pack_down(B.cs, params, base, s)); }
  }
  if (need_denc) MM_TRY(pack_down(dstem, params, base, s));
  return MMVAE_OK;
}

// ------------------------------------------------------------------------------------------------ encoder
int Net::stage_labels(int N, const void* labels, int label_bytes, float mean, float stdv, float* image, void* ws, size_t ws_bytes, hipStream_t s) {
  const Plan& P = plan(N);
  if (ws_bytes < P.bytes) { set_error("workspace too small: %zu < %zu", ws_bytes, P.bytes); return MMVAE_ERR_WORKSPACE; }
  return launch_normalise(dt(), labels, label_bytes, (long)N * cfg.in_ch * cfg.S * cfg.S, mean, stdv, static_cast<char*>(ws) + P.x_t, image, s);
}

int Net::encoder_fwd(int N, const float* x, const float* params, float* bnbuf, long long* nbt, void* ws, size_t ws_bytes,
                     float* mu, float* logvar, int training, hipStream_t s, bool staged) {
  if (Hf > 2) { set_error("encoder: image size %d unsupported (final map %dx%d)", cfg.S, Hf, Wf); return MMVAE_ERR_UNSUPPORTED; }
  const Plan& P = plan(N);
  if (ws_bytes < P.bytes) { set_error("workspace too small: %zu < %zu", ws_bytes, P.bytes); return MMVAE_ERR_WORKSPACE; }
  char* base = static_cast<char*>(ws);
  float* part = reinterpret_cast<float*>(base + P.partials);
  float* stats = training ? part : nullptr;
  const int S = cfg.S;
  pack_batch_begin();                       // every weight re-pack of this entry point in ONE launch
  MM_TRY(packs_enc_fwd(params, base, s));
  MM_TRY(pack_batch_flush(dt(), s));
  eval_folded_ = false;
  gram_ready_ = false; gram_fwd_N_ = training ? N : 0; gram_fwd_ws_ = ws;
  if (!training) MM_TRY(fold_bn_eval(0, params, bnbuf, base, s));
  if (!staged) MM_TRY(launch_convert(DT_F32, dt(), x, base + P.x_t, (long)N * cfg.in_ch * S * S, s));
  if (cfg.in_ch == 1 && stem_fwd_stream_ok(dt(), S)) {
    const int np = launch_stem_fwd_stream(dt(), base + P.x_t, params + stem.off, base + P.y0, stats, N, S, s);
    MM_TRY(np);
    if (training) MM_TRY(bn_train(bn0, params, bnbuf, nbt, base, np, (double)N * H1 * W1, s));
  } else {
    // stem Conv2d(in_channels -> 32, k5 s2 p2) (model.py:94): the planar image (in_channels <= 4 planes) is staged as a zero-padded
    // VE-channel NHWC patch in LDS and runs through the MFMA patch-tile kernel; BatchNorm statistics come out of its epilogue.
    const int cpad = dt() == DT_F32 ? 4 : 8;
    GatherArgs a; std::memset(&a, 0, sizeof(a));
    a.x = base + P.x_t; a.w = base + P.packed + stem_pack * (long)esz(); a.y = base + P.y0; a.stats = stats;
    a.x_planar = 2; a.x_planes = cfg.in_ch;
    a.N = N; a.Hi = S; a.Wi = S; a.Cin = cpad; a.Ho = H1; a.Wo = W1; a.Cout = 32; a.SI = 2; a.SO = 1;
    a.nphase = 1; a.phases[0] = Phase{0, 0, H1, W1, 25, 0, 0};
    for (int kh = 0; kh < 5; ++kh) for (int kw = 0; kw < 5; ++kw) a.taps[kh * 5 + kw] = Tap{kh - 2, kw - 2};
    const int np = launch_gather_gemm(dt(), dt(), a, s);
    MM_TRY(np);
    if (training) MM_TRY(bn_train(bn0, params, bnbuf, nbt, base, np, (double)N * H1 * W1, s));
  }
  if (training) {
  } else {
    MM_TRY(bn_eval(bn0, params, bnbuf, base, s));
  }
  const void* xin = base + P.y0;
  const float* xs = bnf(bn0, base, 2);
  const float* xb = bnf(bn0, base, 3);
  if (cfg.blocks > 1) MM_TRY(fill_consts(base, s));
  for (size_t i = 0; i < enc.size(); ++i) {
    Block& B = enc[i];
    const double cnt = (double)N * B.Hout * B.Wout;
    // shortcut branch (conv + its BatchNorm) on the side stream, concurrently with conv1 -> bn1 -> conv2 -> bn2
    constexpr bool side_fwd = true;
    // encoder.layer1 (32 -> 32 channels, bf16): conv1 AND the 1x1 shortcut from ONE read of the block input, conv2 likewise a per-wave
    // stream (conv_fstream.hip); both BatchNorms finalise on the caller's stream
    const bool stream1 = !B.identity && !B.c1.fp8 && !B.cs.fp8 && B.cs.k == 1 && B.cs.s == 2 &&
                         conv3_stream_ok(dt(), B.Cin, B.C, B.c1.k, B.c1.s, B.c1.p, B.Hin, B.Win);
    const bool stream2 = !B.c2.fp8 && conv3_stream_ok(dt(), B.C, B.C, B.c2.k, B.c2.s, B.c2.p, B.Hout, B.Wout);
    const bool fork = side_fwd && !B.identity && !stream1;
    int np;
    if (stream1) {
      np = launch_conv3_stream(dt(), 2, xin, base + plan_.packed + B.c1.packD * (long)esz(), base + plan_.packed + B.cs.packD * (long)esz(), base + B.y1,
                               base + B.ys, xs, xb, 1, stats, stats ? stats + kPartialFloats : nullptr, N, B.Hout, s);
      MM_TRY(np);
      MM_TRY(training ? bn_train(B.bs, params, bnbuf, nbt, base, np, cnt, s, kPartialFloats, B.cs.wscale) : bn_eval(B.bs, params, bnbuf, base, s, B.cs.wscale));
      MM_TRY(training ? bn_train(B.b1, params, bnbuf, nbt, base, np, cnt, s, 0, B.c1.wscale) : bn_eval(B.b1, params, bnbuf, base, s, B.c1.wscale));
    } else if (!B.identity) {
      if (fork) MM_TRY(side_fork(s));
      hipStream_t ss = fork ? wgrad_stream(s) : s;
      np = run_down(B.cs, base, N, xin, B.Hin, B.Win, base + B.ys, B.Hout, B.Wout, xs, xb, 1, stats ? stats + kPartialFloats : nullptr, 0, dt(), ss);
      MM_TRY(np);
      MM_TRY(training ? bn_train(B.bs, params, bnbuf, nbt, base, np, cnt, ss, kPartialFloats, B.cs.wscale) : bn_eval(B.bs, params, bnbuf, base, ss, B.cs.wscale));
    }
    if (!stream1) {
      np = run_down(B.c1, base, N, xin, B.Hin, B.Win, base + B.y1, B.Hout, B.Wout, xs, xb, 1, stats, 0, dt(), s);
      MM_TRY(np);
      MM_TRY(training ? bn_train(B.b1, params, bnbuf, nbt, base, np, cnt, s, 0, B.c1.wscale) : bn_eval(B.b1, params, bnbuf, base, s, B.c1.wscale));
    }
    if (stream2)
      np = launch_conv3_stream(dt(), 1, base + B.y1, base + plan_.packed + B.c2.packD * (long)esz(), nullptr, base + B.y2, nullptr, bnf(B.b1, base, 2),
                               bnf(B.b1, base, 3), 1, stats, nullptr, N, B.Hout, s);
    else
      np = run_down(B.c2, base, N, base + B.y1, B.Hout, B.Wout, base + B.y2, B.Hout, B.Wout, bnf(B.b1, base, 2), bnf(B.b1, base, 3), 1,
                    stats, 0, dt(), s);
    MM_TRY(np);
    MM_TRY(training ? bn_train(B.b2, params, bnbuf, nbt, base, np, cnt, s, 0, B.c2.wscale) : bn_eval(B.b2, params, bnbuf, base, s, B.c2.wscale));
    if (fork) MM_TRY(side_join(s));
    // identity shortcut (model.py:40,52): the block input itself, i.e. a unit "BatchNorm" (scale 1, shift 0) of it
    MM_TRY(launch_join_fwd(dt(), base + B.y2, bnf(B.b2, base, 2), bnf(B.b2, base, 3), B.identity ? xin : base + B.ys,
                           B.identity ? ones(base) : bnf(B.bs, base, 2), B.identity ? zeros(base) : bnf(B.bs, base, 3), base + B.out,
                           (long)N * B.Hout * B.Wout, B.C, s));
    xin = base + B.out; xs = xb = nullptr;
  }
  // global average pool + the two 1x1 heads (model.py:123-128) as ONE k=Hf,s=Hf conv whose taps share W/(Hf*Wf)
  const int nt = Hf * Wf;
  for (int h = 0; h < (cfg.need_logvar ? 2 : 1); ++h) {
    const ConvW& hw = h == 0 ? head_mu : head_lv;
    const long poff = h == 0 ? head_pack_mu : head_pack_lv;
    PackArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.src = params + hw.off; pa.dst = base + P.packed + poff * (long)esz();
    pa.cols = cfg.z; pa.K = 256; pa.ntaps = nt; pa.s_col = 256; pa.s_k = 1; pa.scale = 1.0f / nt;
    GatherArgs a; std::memset(&a, 0, sizeof(a));
    a.x = xin; a.w = base + P.packed + poff * (long)esz(); a.y = h == 0 ? mu : logvar;
    a.N = N; a.Hi = Hf; a.Wi = Wf; a.Cin = 256; a.Ho = 1; a.Wo = 1; a.Cout = cfg.z; a.SI = Hf; a.SO = 1;
    a.nphase = 1; a.phases[0] = Phase{0, 0, 1, 1, nt, 0, 0};
    for (int kh = 0; kh < Hf; ++kh) for (int kw = 0; kw < Wf; ++kw) a.taps[kh * Wf + kw] = Tap{kh, kw};
    MM_TRY(launch_gather_gemm(dt(), DT_F32, a, s));
  }
  return MMVAE_OK;
}

int Net::encoder_bwd(int N, const float* d_mu, const float* d_logvar, const float* params, float* grads, void* ws, size_t ws_bytes,
                     hipStream_t s) {
  const Plan& P = plan(N);
  if (ws_bytes < P.bytes) { set_error("workspace too small"); return MMVAE_ERR_WORKSPACE; }
  char* base = static_cast<char*>(ws);
  wscratch_ = reinterpret_cast<float*>(base + P.wscratch);
  float* part = reinterpret_cast<float*>(base + P.partials);
  const int Ch = cfg.need_logvar ? 2 * cfg.z : cfg.z;
  const int nt = Hf * Wf;
  pack_batch_begin();
  MM_TRY(packs_enc_bwd(params, base, s));
  MM_TRY(pack_batch_flush(dt(), s));
  // ---- heads: dh = [d_mu | d_logvar] in T
  MM_TRY(launch_concat2_to_t(dt(), d_mu, cfg.need_logvar ? d_logvar : nullptr, N, cfg.z, cfg.need_logvar ? cfg.z : 0, base + P.dh, s));
  {
    WgradArgs a; std::memset(&a, 0, sizeof(a));
    a.P = base + P.dh; a.G = base + enc.back().out; a.dW = grads + head_mu.off; a.proP_relu = a.proG_relu = 0;
    a.N = N; a.Hp = 1; a.Wp = 1; a.Ca = Ch; a.Hg = Hf; a.Wg = Wf; a.Cb = 256; a.Cb_valid = 256;
    a.stride = Hf; a.pad = 0; a.ksz = Hf; a.sA = 256; a.sB = 1; a.ntaps = nt; a.scale = 1.0f / nt;
    a.scratch = wscratch_;
    MM_TRY(side_fork(s));
    MM_TRY(launch_wgrad(dt(), a, wgrad_stream(s)));
    // the stem's im2col depends on the input image only: early, off the tail of the critical path
    // the stem's input-only work (patch gram matrix / im2col) early, off the tail of the critical path
    if (stem_bwd_fused() && !(gram_ready_ && gram_fwd_N_ == N && gram_fwd_ws_ == ws)) {
      MM_TRY(launch_stem_gram(dt(), base + P.x_t, reinterpret_cast<float*>(base + P.stem_gram), 1024L * stem_bwd_part_floats(),
                              reinterpret_cast<double*>(base + P.stem_R), N, cfg.S, H1, W1, wgrad_stream(s)));
      // the stem's finalize at the very end waits for THIS, not for the whole side stream (whose last weight gradients may still run)
      if (side_state_ == 1 && hipEventRecord(gram_ev_, side_) != hipSuccess) { set_error("side stream mark failed"); return MMVAE_ERR_HIP; }
    }
    gram_ready_ = false;            // (consumed: the next backward pass belongs to another forward pass)
    if (!stem_bwd_fused() && cfg.in_ch == 1 && stem_im2col_path() && wgrad_stream(s) != s) MM_TRY(launch_stem_im2col(dt(), base + P.x_t, base + P.col, N, cfg.S, cfg.S, H1, W1, wgrad_stream(s)));
    PackArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.src = params + head_mu.off; pa.dst = base + P.packed + head_pack_dg * (long)esz();
    pa.cols = 256; pa.K = Ch; pa.ntaps = 1; pa.s_col = 1; pa.s_k = 256; pa.scale = 1.0f / nt;
    GatherArgs g; std::memset(&g, 0, sizeof(g));
    g.x = base + P.dh; g.w = pa.dst; g.y = base + P.g[0];
    g.N = N; g.Hi = 1; g.Wi = 1; g.Cin = Ch; g.Ho = Hf; g.Wo = Wf; g.Cout = 256; g.SI = 1; g.SO = Hf;
    g.nphase = nt;
    for (int i = 0; i < nt; ++i) { g.phases[i] = Phase{i / Wf, i % Wf, 1, 1, 1, i, 0}; g.taps[i] = Tap{0, 0}; }
    MM_TRY(launch_gather_gemm(dt(), dt(), g, s));
  }
  int cur = 0;   // d_out lives in g[cur]
  const int ne = (int)enc.size();
  for (int i = ne - 1; i >= 0; --i) {
    Block& B = enc[i];
    const long npix = (long)N * B.Hout * B.Wout;
    const double cnt = (double)npix;
    const void* xin = i == 0 ? base + P.y0 : base + enc[i - 1].out;
    const float* xs = i == 0 ? bnf(bn0, base, 2) : nullptr;
    const float* xb = i == 0 ? bnf(bn0, base, 3) : nullptr;
    // dy1 / dy2 / dys alternate between two sets, so this block only has to wait for the weight gradients of the block
    // before the previous one (the side stream may lag one block behind)
    // The two blocks visited first (small tensors) own private sets: with a deferred decoder join the side stream may still be
    // reading the shared sets for the decoder's weight gradients when they start.  The third block waits for the first one's
    // mark, which -- the side stream being in order -- also covers everything the decoder left there.
    const int ds = i & 1;
    const bool priv = i >= ne - 2;
    const long dy1o = priv ? P.edy1[i - (ne - 2)] : P.dy1[ds], dy2o = priv ? P.edy2[i - (ne - 2)] : P.dy2[ds];
    // an identity shortcut's gradient is the masked incoming gradient itself: it goes straight into the block-input gradient
    const long dyso = B.identity ? P.g[cur ^ 1] : (priv ? P.edys[i - (ne - 2)] : P.dys[ds]);
    if (i + 2 <= ne - 1) MM_TRY(side_wait_mark(i + 2, s));
    const void* ysp = B.identity ? xin : base + B.ys;
    const float* ssc = B.identity ? ones(base) : bnf(B.bs, base, 2);
    const float* ssh = B.identity ? zeros(base) : bnf(B.bs, base, 3);
    // join backward: g = d_out * [out > 0] feeds bn2 (y2) and the shortcut BN (ys)
    int np = launch_bn_bwd_reduce(dt(), base + P.g[cur], nullptr, bnf(B.b2, base, 2), bnf(B.b2, base, 3), base + B.y2, ysp, npix, B.C, part, s, ssc, ssh);
    MM_TRY(np);
    if (B.identity) MM_TRY(bn_backward_coefs(B.b2, params, grads, base, np, 2, 0, cnt, s));
    else MM_TRY(bn_backward_coefs_join(B.b2, B.bs, params, grads, base, np, cnt, s));
    MM_TRY(launch_bn_bwd_apply(dt(), base + P.g[cur], nullptr, bnf(B.b2, base, 2), bnf(B.b2, base, 3), base + B.y2, bnf(B.b2, base, 4), bnf(B.b2, base, 5),
                               bnf(B.b2, base, 6), base + dy2o, ysp, B.identity ? ones(base) : bnf(B.bs, base, 4),
                               B.identity ? zeros(base) : bnf(B.bs, base, 5), B.identity ? zeros(base) : bnf(B.bs, base, 6), base + dyso, npix, B.C, s,
                               ssc, ssh));
    // conv2 (3x3 s1): wgrad with a1 = relu(bn1(y1)) recomputed in the load prologue; dgrad -> d_a1
    hipStream_t wsm = wgrad_stream(s);
    MM_TRY(side_fork(s));
    // (measured and not kept: a THIRD stream for this block's two weight-gradient kernels, so that they need not wait for the side stream's
    // backlog of deeper layers -- 6.93-6.96 against 6.96 ms per step, within the run-to-run spread)
    // encoder.layer1 (the block the step ends on): conv2's partial images go to the second scratch (idle in this pass) and are reduced
    // AFTER the block's other weight gradient has been enqueued -- under the stem backward's load a 19 MB reduce takes 120 us instead of
    // 10, and it sat in front of the last big kernel of the side stream
    WgradReduceArgs late; late.nparts = 0;
    if (i == 0 && wsm != s && !B.c2.fp8 && op_wgrad_is_stream(dt(), geom(B.c2), N, B.Hout, B.Wout, B.Hout, B.Wout, false, true))
      MM_TRY(op_run_wgrad(dt(), geom(B.c2), N, base + dy2o, B.Hout, B.Wout, nullptr, nullptr, 1, base + B.y1, B.Hout, B.Wout, bnf(B.b1, base, 2),
                          bnf(B.b1, base, 3), 1, grads + B.c2.off, wsm, reinterpret_cast<float*>(base + P.wscratch2), B.c2.wscale, &late));
    else
      MM_TRY(run_wgrad(B.c2, N, base + dy2o, B.Hout, B.Wout, nullptr, nullptr, base + B.y1, B.Hout, B.Wout, bnf(B.b1, base, 2),
                       bnf(B.b1, base, 3), grads, wsm));
    // (encoder.layer1: the shortcut's weight gradient rides on conv1's pass over the block input, below)
    const bool pair = !B.identity && !B.c1.fp8 && !B.cs.fp8 && wgrad_pair_ok(dt(), geom(B.c1), geom(B.cs), B.Hout, B.Hin);
    if (!B.identity && !pair) MM_TRY(run_wgrad(B.cs, N, base + dyso, B.Hout, B.Wout, nullptr, nullptr, xin, B.Hin, B.Win, xs, xb, grads, wsm));
    if (i == 0 && l1_dgrad_stream()) {
      // encoder.layer1.conv2: the data gradient as a per-wave stream over dy2 (conv3_stream_kernel) with bn1's backward sums from the same pass
      np = launch_conv3_stream_bwd(dt(), base + dy2o, base + plan_.packed + l1c2_flip * (long)esz(), base + P.da1, base + B.y1, bnf(B.b1, base, 2),
                                   bnf(B.b1, base, 3), part, N, B.Hout, s);
    } else {
      MM_TRY(run_up(B.c2, base, N, base + dy2o, B.Hout, B.Wout, base + P.da1, B.Hout, B.Wout, nullptr, nullptr, 0, nullptr, 0, s));
      // bn1 + relu backward
      np = launch_bn_bwd_reduce(dt(), base + P.da1, nullptr, bnf(B.b1, base, 2), bnf(B.b1, base, 3), base + B.y1, nullptr, npix, B.C, part, s);
    }
    MM_TRY(np);
    MM_TRY(bn_backward_coefs(B.b1, params, grads, base, np, 1, 0, cnt, s));
    MM_TRY(launch_bn_bwd_apply(dt(), base + P.da1, nullptr, bnf(B.b1, base, 2), bnf(B.b1, base, 3), base + B.y1, bnf(B.b1, base, 4),
                               bnf(B.b1, base, 5), bnf(B.b1, base, 6), base + dy1o, nullptr, nullptr, nullptr, nullptr, nullptr, npix,
                               B.C, s));
    // conv1 (3x3) and the 1x1 s2 shortcut: weight gradients, then d_xin = dgrad(conv1) + dgrad(shortcut)
    MM_TRY(side_fork(s));
    int taken = 0;
    if (pair) {
      taken = op_run_wgrad_pair(dt(), geom(B.c1), geom(B.cs), N, base + dy1o, base + dyso, B.Hout, B.Wout, xin, B.Hin, B.Win, xs, xb, 1,
                                grads + B.c1.off, grads + B.cs.off, wsm, wscratch_, B.c1.wscale, B.cs.wscale);
      MM_TRY(taken);
      if (!taken) MM_TRY(run_wgrad(B.cs, N, base + dyso, B.Hout, B.Wout, nullptr, nullptr, xin, B.Hin, B.Win, xs, xb, grads, wsm));
    }
    if (!taken) MM_TRY(run_wgrad(B.c1, N, base + dy1o, B.Hout, B.Wout, nullptr, nullptr, xin, B.Hin, B.Win, xs, xb, grads, wsm));
    if (late.nparts > 0) MM_TRY(launch_wgrad_reduce(late, wsm));
    MM_TRY(side_mark(i));
    if (i == 0 && stem_dg_fused()) {
      // the gradient of the stem's output is never a tensor: stem_bwd_kernel<DG> recomputes it row by row from dy1 / dys (below)
      stem_dy1_ = dy1o; stem_dys_ = dyso;
    } else if (B.identity)     // d_xin already holds the shortcut's share: the main path's data gradient is added to it
      MM_TRY(run_up(B.c1, base, N, base + dy1o, B.Hout, B.Wout, base + P.g[cur ^ 1], B.Hin, B.Win, nullptr, nullptr, 0, nullptr, 1, s));
    else                // one kernel: the 1x1 stride-2 shortcut's data gradient is a second source of the 3x3 conv's (phase (0,0))
      MM_TRY(run_up(B.c1, base, N, base + dy1o, B.Hout, B.Wout, base + P.g[cur ^ 1], B.Hin, B.Win, nullptr, nullptr, 0, nullptr, 0, s, &B.cs,
                    base + dyso));
    cur ^= 1;
  }
  // ---- stem: bn0 + relu backward and the 5x5 weight gradient
  if (stem_bwd_fused()) {
    // one pass over g and y0 (stem_bwd.hip): BatchNorm sums and the three pixel reductions dW is an affine function of; no dy
    // tensor, no im2col, nothing waits on a grid-wide reduction except the 32-block finalize
    const long npix = (long)N * H1 * W1;
    const Block& B0 = enc[0];
    const int np = stem_dg_fused()
        ? launch_stem_bwd_dg(base + stem_dy1_, base + stem_dys_, base + P.packed + B0.c1.packU * (long)esz(), base + P.packed + B0.cs.packU * (long)esz(),
                             base + P.y0, base + P.x_t, bnf(bn0, base, 2), bnf(bn0, base, 3), part, kPartialFloats, N, cfg.S, H1, W1, s)
        : launch_stem_bwd(dt(), base + P.g[cur], base + P.y0, base + P.x_t, bnf(bn0, base, 2), bnf(bn0, base, 3), part, kPartialFloats,
                          N, cfg.S, H1, W1, s);
    MM_TRY(np);
    const float* gsum = nullptr; double cnt = (double)npix;
    if (sync_bn_on()) {
      float* row = nullptr;
      MM_TRY(sync_rows(base, part, np, 64, s, &row, stem_bwd_part_floats()));
      gsum = row; cnt *= ar_world_;
    }
    if (side_state_ == 1 && hipStreamWaitEvent(s, gram_ev_, 0) != hipSuccess) { set_error("side stream wait failed"); return MMVAE_ERR_HIP; }
    MM_TRY(launch_stem_bwd_finalize(part, np, reinterpret_cast<const double*>(base + P.stem_R), params + stem.off, gsum, cnt,
                                    params + bn0.g_off, bnf(bn0, base, 0), bnf(bn0, base, 1), grads + bn0.g_off, grads + bn0.b_off,
                                    grads + stem.off, s));
    return side_join(s);             // every weight gradient of this pass
  }
  // ---- stem: bn0 + relu backward, then the 5x5 weight gradient
  MM_TRY(side_wait_mark(1, s));    // the stem uses dy set 1 (block index -1): block 1's weight gradients read it last
  const int ds = 1;
  {
    const long npix = (long)N * H1 * W1;
    int np = launch_bn_bwd_reduce(dt(), base + P.g[cur], nullptr, bnf(bn0, base, 2), bnf(bn0, base, 3), base + P.y0, nullptr, npix, 32, part, s);
    MM_TRY(np);
    MM_TRY(bn_backward_coefs(bn0, params, grads, base, np, 1, 0, (double)npix, s));
    MM_TRY(launch_bn_bwd_apply(dt(), base + P.g[cur], nullptr, bnf(bn0, base, 2), bnf(bn0, base, 3), base + P.y0, bnf(bn0, base, 4),
                               bnf(bn0, base, 5), bnf(bn0, base, 6), base + P.dy1[ds], nullptr, nullptr, nullptr, nullptr, nullptr, npix, 32, s));
    const bool stem_im2col = stem_im2col_path();
    MM_TRY(side_fork(s));
    hipStream_t wsm = wgrad_stream(s);
    WgradArgs a; std::memset(&a, 0, sizeof(a));
    a.P = base + P.dy1[ds]; a.dW = grads + stem.off; a.scratch = wscratch_;
    a.N = N; a.Hp = H1; a.Wp = W1; a.Ca = 32; a.scale = 1.f;
    if (cfg.in_ch > 1) {
      // in_channels > 1 (main.py:555): G = the image as NHWC with 16 zero-padded channels, the generic 25-tap weight gradient
      MM_TRY(launch_planar_to_nhwc16(dt(), static_cast<const void*>(base + P.x_t), dt(), base + P.col, N, cfg.in_ch, cfg.S * cfg.S, wsm));
      a.G = base + P.col; a.Hg = cfg.S; a.Wg = cfg.S; a.Cb = 16; a.Cb_valid = cfg.in_ch;
      a.stride = 2; a.pad = 2; a.ksz = 5; a.sA = 25 * cfg.in_ch; a.sB = 25; a.ntaps = 25;
      for (int t = 0; t < 25; ++t) a.tap_off[t] = t;
    } else if (stem_im2col) {
      // im2col of the 1-channel image (25 taps padded to 32 columns) + the MFMA weight-gradient kernel as a 1x1 conv
      if (wsm == s) MM_TRY(launch_stem_im2col(dt(), base + P.x_t, base + P.col, N, cfg.S, cfg.S, H1, W1, wsm));   // else: done early
      a.G = base + P.col; a.Hg = H1; a.Wg = W1; a.Cb = 32; a.Cb_valid = 25;
      a.stride = 1; a.pad = 0; a.ksz = 1; a.sA = 25; a.sB = 1; a.ntaps = 1;
    } else {
      // dW[co][0][kh][kw]: P = dy (32 channels), G = the planar 1-channel image staged as 16 zero-padded channels in LDS
      a.G = base + P.x_t; a.G_planar = 1; a.Hg = cfg.S; a.Wg = cfg.S; a.Cb = 16; a.Cb_valid = 1;
      a.stride = 2; a.pad = 2; a.ksz = 5; a.sA = 25; a.sB = 25; a.ntaps = 25;
      for (int t = 0; t < 25; ++t) a.tap_off[t] = t;
    }
    MM_TRY(launch_wgrad(dt(), a, wsm));
    MM_TRY(side_join(s));
  }
  return MMVAE_OK;
}

// ------------------------------------------------------------------------------------------------ decoder
int Net::decoder_fwd(int N, const float* encv, const float* params, float* bnbuf, long long* nbt, void* ws, size_t ws_bytes,
                     float* recon, int training, hipStream_t s) {
  const Plan& P = plan(N);
  if (ws_bytes < P.bytes) { set_error("workspace too small: %zu < %zu", ws_bytes, P.bytes); return MMVAE_ERR_WORKSPACE; }
  char* base = static_cast<char*>(ws);
  float* part = reinterpret_cast<float*>(base + P.partials);
  float* stats = training ? part : nullptr;
  pack_batch_begin();
  MM_TRY(packs_dec_fwd(params, base, s));
  MM_TRY(pack_batch_flush(dt(), s));
  eval_folded_ = false;
  if (!training) MM_TRY(fold_bn_eval(1, params, bnbuf, base, s));
  MM_TRY(launch_convert(DT_F32, dt(), encv, base + P.enc_t, (long)N * cfg.z, s));
  // stem ConvTranspose2d(z -> 128, k2) on the 1x1 latent (model.py:159-161,182)
  int np = run_up(dstem, base, N, base + P.enc_t, 1, 1, base + P.y0d, 2, 2, nullptr, nullptr, 0, stats, 0, s);
  MM_TRY(np);
  MM_TRY(training ? bn_train(dbn0, params, bnbuf, nbt, base, np, (double)N * 4, s, 0, dstem.wscale) : bn_eval(dbn0, params, bnbuf, base, s, dstem.wscale));
  const void* xin = base + P.y0d;
  const float* xs = bnf(dbn0, base, 2);
  const float* xb = bnf(dbn0, base, 3);
  const int nd = (int)dec.size();
  if (cfg.blocks > 1) {
    // the first block of a deeper decoder has an identity shortcut: it needs the stem's activation as a tensor
    MM_TRY(fill_consts(base, s));
    MM_TRY(launch_affine_act(dt(), base + P.y0d, xs, xb, 1, base + P.act0d, (long)N * 4, 128, s));
    xin = base + P.act0d; xs = xb = nullptr;
  }
  int c1_done = 0;                 // rows of bn1 statistics the previous block's join kernel left for this block's conv1 (0: conv1 not done)
  for (int i = 0; i < nd; ++i) {
    Block& B = dec[i];
    const double cnt = (double)N * B.Hout * B.Wout;
    // upsample (shortcut) branch on the side stream, concurrently with conv1 -> bn1 -> conv2 -> bn2
    constexpr bool side_fwd = true;
    const bool fork = side_fwd && !B.identity;
    if (!B.identity) {
      if (fork) MM_TRY(side_fork(s));
      hipStream_t ss = fork ? wgrad_stream(s) : s;
      if (!B.cs.fp8 && convT4_stream_ok(dt(), B.cs.D0, B.cs.D1, B.cs.k, B.cs.s, B.cs.p, B.Hin, B.Win))
        np = launch_convT4_stream(dt(), xin, base + plan_.packed + B.cs.packU * (long)esz(), base + B.ys, xs, xb, 1, stats ? stats + kPartialFloats : nullptr,
                                  N, B.Hin, ss, (store8 && i == nd - 1) ? 1 : 0);
      else
        np = run_up(B.cs, base, N, xin, B.Hin, B.Win, base + B.ys, B.Hout, B.Wout, xs, xb, 1, stats ? stats + kPartialFloats : nullptr, 0, ss);
      MM_TRY(np);
      MM_TRY(training ? bn_train(B.bs, params, bnbuf, nbt, base, np, cnt, ss, kPartialFloats, B.cs.wscale) : bn_eval(B.bs, params, bnbuf, base, ss, B.cs.wscale));
    }
    // (conv1 already done: the previous block's join kernel computed it from the joined row it had in LDS -- join_conv1_fwd below)
    np = c1_done > 0 ? c1_done : run_down(B.c1, base, N, xin, B.Hin, B.Win, base + B.y1, B.Hin, B.Win, xs, xb, 1, stats, 0, dt(), s);
    c1_done = 0;
    MM_TRY(np);
    MM_TRY(training ? bn_train(B.b1, params, bnbuf, nbt, base, np, (double)N * B.Hin * B.Win, s, 0, B.c1.wscale) : bn_eval(B.b1, params, bnbuf, base, s, B.c1.wscale));
    if (B.identity)    // 3x3 Conv2d, shape preserving
      np = run_down(B.c2, base, N, base + B.y1, B.Hin, B.Win, base + B.y2, B.Hout, B.Wout, bnf(B.b1, base, 2), bnf(B.b1, base, 3), 1, stats, 0, dt(), s);
    else
      if (!B.c2.fp8 && convT4_stream_ok(dt(), B.c2.D0, B.c2.D1, B.c2.k, B.c2.s, B.c2.p, B.Hin, B.Win))   // per-wave stream (conv_fstream.hip)
        np = launch_convT4_stream(dt(), base + B.y1, base + plan_.packed + B.c2.packU * (long)esz(), base + B.y2, bnf(B.b1, base, 2), bnf(B.b1, base, 3), 1,
                                  stats, N, B.Hin, s, (store8 && i == nd - 1) ? 1 : 0);
      else
        np = run_up(B.c2, base, N, base + B.y1, B.Hin, B.Win, base + B.y2, B.Hout, B.Wout, bnf(B.b1, base, 2), bnf(B.b1, base, 3), 1, stats, 0, s);
    MM_TRY(np);
    MM_TRY(training ? bn_train(B.b2, params, bnbuf, nbt, base, np, cnt, s, 0, B.c2.wscale) : bn_eval(B.b2, params, bnbuf, base, s, B.c2.wscale));
    if (fork) MM_TRY(side_join(s));
    if (i == nd - 1 && tail_fwd_fused()) break;     // the join of the last block happens inside the tail conv kernel
    if (i == nd - 1 && store8) { set_error("fp8 storage of the last up-block needs the fused tail kernels (MMVAE_TAIL_FWD_FUSED / MMVAE_TAIL_FUSED)"); return MMVAE_ERR_UNSUPPORTED; }
    // the join, fused with the next block's 1x1 conv1 and bn1's statistics where that block's conv1 is 16-wide (uplayer3 -> 4, uplayer4 -> 5):
    // the joined row is the conv's operand while it is in LDS -- one launch and one pass over `out` fewer than join -> conv
    const bool fuse_c1 = i + 1 < nd && !B.identity && !dec[i + 1].identity && !dec[i + 1].c1.fp8 && dec[i + 1].c1.k == 1 && dec[i + 1].c1.s == 1 &&
                         dec[i + 1].c1.D1 == B.C && frag_down(dec[i + 1].c1) == 0 && dec[i + 1].c1.wscale == 1.f &&
                         join_conv1_fwd_ok(dt(), B.C, dec[i + 1].c1.D0, (long)N * B.Hout * B.Wout);
    if (fuse_c1) {
      c1_done = launch_join_conv1_fwd(B.C, base + B.y2, bnf(B.b2, base, 2), bnf(B.b2, base, 3), base + B.ys, bnf(B.bs, base, 2), bnf(B.bs, base, 3),
                                      base + plan_.packed + dec[i + 1].c1.packD * (long)esz(), base + B.out, base + dec[i + 1].y1, stats,
                                      (long)N * B.Hout * B.Wout, s);
      MM_TRY(c1_done);
    } else
    MM_TRY(launch_join_fwd(dt(), base + B.y2, bnf(B.b2, base, 2), bnf(B.b2, base, 3), B.identity ? xin : base + B.ys,
                           B.identity ? ones(base) : bnf(B.bs, base, 2), B.identity ? zeros(base) : bnf(B.bs, base, 3), base + B.out,
                           (long)N * B.Hout * B.Wout, B.C, s));
    xin = base + B.out; xs = xb = nullptr;
  }
  // tail conv (+bias) and the output BatchNorm (model.py:193)
  float* r_raw = reinterpret_cast<float*>(base + P.r_raw);
  if (tail_fwd_fused()) {
    const Block& B = dec.back();
    if (!store8 && !B.identity && !B.c2.fp8 && !B.cs.fp8 && tail_fwd_stream_ok(dt(), cfg.out_ch, Sd, Sd) &&
        up5_tail_fwd_ok(dt(), cfg.out_ch, B.C, B.Cin, B.Hin, B.Hout) && convT4_stream_ok(dt(), B.c2.D0, B.c2.D1, B.c2.k, B.c2.s, B.c2.p, B.Hin, B.Win))
      // the join + tail conv with both branch outputs recomputed from the ConvTranspose2d inputs (0.34 GB) instead of read back (1.34 GB)
      np = launch_up5_tail_fwd(base + B.y1, bnf(B.b1, base, 2), bnf(B.b1, base, 3), base + plan_.packed + B.c2.packU * (long)esz(), xin, xs, xb,
                               base + plan_.packed + B.cs.packU * (long)esz(), bnf(B.b2, base, 2), bnf(B.b2, base, 3), bnf(B.bs, base, 2),
                               bnf(B.bs, base, 3), params + tail.off, params + tail_bias, r_raw, stats, N, s);
    else if (tail_fwd_stream_ok(dt(), cfg.out_ch, Sd, Sd))
      np = launch_tail_fwd_stream(dt(), base + B.y2, bnf(B.b2, base, 2), bnf(B.b2, base, 3), base + B.ys, bnf(B.bs, base, 2), bnf(B.bs, base, 3),
                                  params + tail.off, params + tail_bias, r_raw, stats, N, Sd, Sd, s, store8 ? 1 : 0);
    else if (store8) { set_error("fp8 storage of the last up-block needs tail_fwd_stream"); return MMVAE_ERR_UNSUPPORTED; }
    else
      np = launch_tail_join_fwd(dt(), base + B.y2, bnf(B.b2, base, 2), bnf(B.b2, base, 3), base + B.ys, bnf(B.bs, base, 2), bnf(B.bs, base, 3),
                                params + tail.off, params + tail_bias, r_raw, stats, N, Sd, Sd, s);
    MM_TRY(np);
    if (training) MM_TRY(bn_train(bn_out, params, bnbuf, nbt, base, np, (double)N * Sd * Sd, s));
  } else {
    // Conv2d(16 -> out_ch, k3 p1, bias): GEMM rows padded to 16 in LDS, epilogue stores the out_ch real rows as NCHW f32
    PackArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.src = params + tail.off; pa.dst = base + P.packed + tail_pack_f * (long)esz();
    pa.cols = 16; pa.cols_valid = cfg.out_ch; pa.K = 16; pa.ntaps = 9; pa.s_col = 144; pa.s_k = 9; pa.scale = 1.f;
    for (int t = 0; t < 9; ++t) pa.tap_off[t] = t;
    GatherArgs a; std::memset(&a, 0, sizeof(a));
    a.x = xin; a.w = pa.dst; a.y = r_raw; a.bias = params + tail_bias; a.stats = stats; a.y_planes = cfg.out_ch;
    a.N = N; a.Hi = Sd; a.Wi = Sd; a.Cin = 16; a.Ho = Sd; a.Wo = Sd; a.Cout = 16; a.SI = 1; a.SO = 1;
    a.nphase = 1; a.phases[0] = Phase{0, 0, Sd, Sd, 9, 0, 0};
    for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) a.taps[kh * 3 + kw] = Tap{kh - 1, kw - 1};
    np = launch_gather_gemm(dt(), DT_F32, a, s);
    MM_TRY(np);
    if (training) MM_TRY(bn_train(bn_out, params, bnbuf, nbt, base, np, (double)N * Sd * Sd, s));
  }
  if (training) {
  } else {
    MM_TRY(bn_eval(bn_out, params, bnbuf, base, s));
  }
  MM_TRY(launch_affine_nchw(r_raw, bnf(bn_out, base, 2), bnf(bn_out, base, 3), recon, N, cfg.out_ch, Sd * Sd, s));
  return MMVAE_OK;
}

int Net::decoder_bwd(int N, const float* d_recon, const float* params, float* grads, void* ws, size_t ws_bytes, float* d_enc,
                     hipStream_t s, const GaussTail* gauss) {
  const Plan& P = plan(N);
  if (ws_bytes < P.bytes) { set_error("workspace too small"); return MMVAE_ERR_WORKSPACE; }
  char* base = static_cast<char*>(ws);
  wscratch_ = reinterpret_cast<float*>(base + P.wscratch);
  float* part = reinterpret_cast<float*>(base + P.partials);
  float* r_raw = reinterpret_cast<float*>(base + P.r_raw);
  float* d_raw = reinterpret_cast<float*>(base + P.d_raw);
  const int HW = Sd * Sd;
  pack_batch_begin();
  MM_TRY(packs_dec_bwd(params, base, d_enc != nullptr, s));
  MM_TRY(pack_batch_flush(dt(), s));
  // the encoder stem's patch gram matrix (input only; stem_bwd.hip): here the side stream is idle for ~1 ms beside HBM-bound kernels, at the
  // start of encoder_bwd it ran beside the latency-bound kernels around the latent code and held their small launches up (lesson 55)
  if (stem_bwd_fused() && !gram_ready_ && gram_fwd_N_ == N && gram_fwd_ws_ == ws && wgrad_stream(s) != s) {
    MM_TRY(side_fork(s));
    MM_TRY(launch_stem_gram(dt(), base + P.x_t, reinterpret_cast<float*>(base + P.stem_gram), 1024L * stem_bwd_part_floats(),
                            reinterpret_cast<double*>(base + P.stem_R), N, cfg.S, H1, W1, wgrad_stream(s)));
    if (hipEventRecord(gram_ev_, side_) != hipSuccess) { set_error("side stream mark failed"); return MMVAE_ERR_HIP; }
    gram_ready_ = true;
    // (the encoder backward's weight packs enqueued here as well -- 26 us off the caller's stream at the latent boundary -- change nothing:
    // 6.018 vs 6.018 ms over three A/B pairs; not kept)
  }
  // ---- output BN backward, tail conv backward
  int np = gauss ? launch_gauss_tail_reduce(r_raw, gauss->target, bnf(bn_out, base, 2), bnf(bn_out, base, 3), gauss->sigma, gauss->coef, gauss->gscale, N,
                                            cfg.out_ch, HW, part, s)
                 : launch_bn_bwd_reduce_nchw(d_recon, r_raw, N, cfg.out_ch, HW, part, s);
  MM_TRY(np);
  // (the tail conv's bias gradient, sum of d_raw per output channel, in closed form from the same sums: BnBwdFinalizeArgs::dbias_conv)
  MM_TRY(bn_backward_coefs(bn_out, params, grads, base, np, 1, 0, (double)N * HW, s, grads + tail_bias));
  if (gauss)
    MM_TRY(launch_gauss_tail_apply(r_raw, gauss->target, bnf(bn_out, base, 2), bnf(bn_out, base, 3), gauss->sigma, gauss->coef, gauss->gscale,
                                   bnf(bn_out, base, 4), bnf(bn_out, base, 5), bnf(bn_out, base, 6), d_raw, N, cfg.out_ch, HW, s));
  else
    MM_TRY(launch_bn_bwd_apply_nchw(d_recon, r_raw, bnf(bn_out, base, 4), bnf(bn_out, base, 5), bnf(bn_out, base, 6), d_raw, N, cfg.out_ch, HW, s));
  constexpr bool tail_fused_env = true;
  const bool tail_fused = tail_fused_env && !dec.empty() && dec.back().C == 16 && tail_join_fusable(dt(), cfg.out_ch, N, Sd, Sd);
  // forward did not store the joined activation: the weight gradient recomputes it inside the join-backward reduce pass (below)
  const bool tail_wg_in_reduce = tail_fwd_fused() && tail_fused;
  if (store8 && !tail_wg_in_reduce) { set_error("fp8 storage of the last up-block needs the fused tail backward"); return MMVAE_ERR_UNSUPPORTED; }
  if (tail_wg_in_reduce) {
  } else if (tail_fwd_fused()) {
    set_error("decoder_bwd: the fused tail forward (no stored join) needs the fused tail backward (MMVAE_TAIL_FUSED)"); return MMVAE_ERR_UNSUPPORTED;
  } else {
    // dW[oc][ci][kh][kw]: P = d_raw (planar f32, out_ch planes staged as 16 zero-padded channels), G = the last up-block's output
    WgradArgs a; std::memset(&a, 0, sizeof(a));
    a.P = d_raw; a.P_planar = 1; a.P_planes = cfg.out_ch; a.G = base + dec.back().out; a.dW = grads + tail.off; a.scratch = wscratch_;
    a.N = N; a.Hp = Sd; a.Wp = Sd; a.Ca = 16; a.Ca_valid = cfg.out_ch; a.Hg = Sd; a.Wg = Sd; a.Cb = 16; a.Cb_valid = 16;
    a.stride = 1; a.pad = 1; a.ksz = 3; a.sA = 16 * 9; a.sB = 9; a.ntaps = 9; a.scale = 1.f;
    for (int t = 0; t < 9; ++t) a.tap_off[t] = t;
    MM_TRY(side_fork(s));
    MM_TRY(launch_wgrad(dt(), a, wgrad_stream(s)));
  }
  int cur = 0;
  // The tail conv's input gradient is not materialised: the last up-block's join backward recomputes it from d_raw
  // (launch_tail_join_bwd_*; MMVAE_TAIL_FUSED=0 restores the separate dgrad kernel).
  if (tail_fused) {
  } else {
    // dx[n,h,w,ci] = sum dy[n,oc,h+1-kh,w+1-kw] * w[oc][ci][kh][kw]: planar f32 source padded to 8 channels in LDS
    PackArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.src = params + tail.off; pa.dst = base + P.packed + tail_pack_d * (long)esz();
    pa.cols = 16; pa.K = 8; pa.K_valid = cfg.out_ch; pa.ntaps = 9; pa.s_col = 9; pa.s_k = 144; pa.scale = 1.f;
    for (int t = 0; t < 9; ++t) pa.tap_off[t] = t;
    GatherArgs a; std::memset(&a, 0, sizeof(a));
    a.x = d_raw; a.w = pa.dst; a.y = base + P.g[cur]; a.x_planar = 1; a.x_planes = cfg.out_ch;
    a.N = N; a.Hi = Sd; a.Wi = Sd; a.Cin = 8; a.Ho = Sd; a.Wo = Sd; a.Cout = 16; a.SI = 1; a.SO = 1;
    a.nphase = 1; a.phases[0] = Phase{0, 0, Sd, Sd, 9, 0, 0};
    for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) a.taps[kh * 3 + kw] = Tap{1 - kh, 1 - kw};
    MM_TRY(launch_gather_gemm(dt(), dt(), a, s));
  }
  const int nd = (int)dec.size();
  // Join-gradient form of a block's backward (blocks on 16-wide maps, uplayer4): the block above hands down its input gradient already masked
  // by this block's join ReLU (conv1_bwd_stream_kernel, MASK), the BatchNorm-backward reduce runs unmasked, and dy2 / dys are evaluated by the
  // loaders of the two fused ConvTranspose2d backward passes -- the bn_bwd_apply launch (840 MB at N = 5120) and both dy tensors disappear.
  auto jg_block = [&](int j) {
    if (j < 0 || j >= nd || dt() != DT_BF16 || !join_grad_) return false;
    const Block& Bj = dec[j];
    if (Bj.identity || Bj.C != 16 || Bj.c2.fp8 || Bj.cs.fp8 || Bj.c1.fp8 || (j == 0 && cfg.blocks <= 1)) return false;
    return op_bwd_fusable_jg(dt(), geom(Bj.c2), N, Bj.Hin, Bj.Win, Bj.Hout, Bj.Wout, true, false, true) &&
           op_bwd_fusable_jg(dt(), geom(Bj.cs), N, Bj.Hin, Bj.Win, Bj.Hout, Bj.Wout, false, true, false);
  };
  bool g_masked = false;             // g[cur] is masked by the ReLU of the block about to run its backward
  for (int i = nd - 1; i >= 0; --i) {
    Block& B = dec[i];
    const bool jg = g_masked;
    g_masked = false;
    const long npo = (long)N * B.Hout * B.Wout, npi = (long)N * B.Hin * B.Win;
    // (blocks > 1: block 0 has an identity shortcut and reads the stem's materialised activation)
    const void* xin = i == 0 ? (cfg.blocks > 1 ? base + P.act0d : base + P.y0d) : base + dec[i - 1].out;
    const float* xs = (i == 0 && cfg.blocks <= 1) ? bnf(dbn0, base, 2) : nullptr;
    const float* xb = (i == 0 && cfg.blocks <= 1) ? bnf(dbn0, base, 3) : nullptr;
    const int ds = i & 1;          // dy set of this block (see encoder_bwd)
    if (i + 2 <= nd - 1) MM_TRY(side_wait_mark(i + 2, s));
    const bool from_tail = tail_fused && i == nd - 1;
    const void* ysp = B.identity ? xin : base + B.ys;
    const float* ssc = B.identity ? ones(base) : bnf(B.bs, base, 2);
    const float* ssh = B.identity ? zeros(base) : bnf(B.bs, base, 3);
    // an identity shortcut's gradient is the masked incoming gradient itself: it goes straight into the block-input gradient
    const long dyso = B.identity ? P.g[cur ^ 1] : P.dys[ds];
    if (from_tail) {
      np = launch_tail_join_bwd_reduce(dt(), d_raw, params + tail.off, cfg.out_ch, N, Sd, Sd, bnf(B.b2, base, 2), bnf(B.b2, base, 3), bnf(B.bs, base, 2),
                                       bnf(B.bs, base, 3), base + B.y2, base + B.ys, part, s, tail_wg_in_reduce ? wscratch_ : nullptr, store8 ? 1 : 0);
      if (tail_wg_in_reduce && np > 0) {
        MM_TRY(side_fork(s));
        MM_TRY(launch_tail_wgrad_finalize(wscratch_, np, grads + tail.off, wgrad_stream(s)));
      }
    } else if (jg)
      np = launch_bn_bwd_reduce(dt(), base + P.g[cur], nullptr, nullptr, nullptr, base + B.y2, ysp, npo, B.C, part, s, nullptr, nullptr);
    else
      np = launch_bn_bwd_reduce(dt(), base + P.g[cur], nullptr, bnf(B.b2, base, 2), bnf(B.b2, base, 3), base + B.y2, ysp, npo, B.C, part, s, ssc, ssh);
    MM_TRY(np);
    if (B.identity) MM_TRY(bn_backward_coefs(B.b2, params, grads, base, np, 2, 0, (double)npo, s));
    else MM_TRY(bn_backward_coefs_join(B.b2, B.bs, params, grads, base, np, (double)npo, s));
    // The last up-block in one pass (conv_joinbwd.hip): dy2 / dys are produced row by row inside the kernel that consumes them -- both
    // ConvTranspose2d weight gradients, both data gradients, bn1's backward sums -- and never stored; conv1 (1x1) follows once bn1's
    // sums are final.
    if (from_tail && !B.identity && !B.c2.fp8 && !B.cs.fp8 && !B.c1.fp8 && B.Cin == 16 && join_bwd_stream_ok(dt(), cfg.out_ch, B.C, B.Hin, B.Hout)) {
      float* ws2 = reinterpret_cast<float*>(base + P.wscratch2);
      JoinBwdLaunch L;
      L.d_raw = d_raw; L.w_tail = params + tail.off; L.y2 = base + B.y2; L.ys = base + B.ys;
      L.ms2 = bnf(B.b2, base, 2); L.mb2 = bnf(B.b2, base, 3); L.mss = bnf(B.bs, base, 2); L.mbs = bnf(B.bs, base, 3);
      L.A2 = bnf(B.b2, base, 4); L.B2 = bnf(B.b2, base, 5); L.C2 = bnf(B.b2, base, 6);
      L.As = bnf(B.bs, base, 4); L.Bs = bnf(B.bs, base, 5); L.Cs = bnf(B.bs, base, 6);
      L.y1 = base + B.y1; L.p1s = bnf(B.b1, base, 2); L.p1b = bnf(B.b1, base, 3);
      L.wd2 = base + plan_.packed + B.c2.packD * (long)esz(); L.da1 = base + P.da1; L.part2 = ws2; L.bn_part = part;
      L.xin = xin; L.pxs = xs; L.pxb = xb; L.wds = base + plan_.packed + B.cs.packD * (long)esz(); L.gin = base + P.g[cur ^ 1];
      L.parts = ws2 + 1024L * 4096;                        // (at most 1024 blocks, one [16][16][16] partial image per conv each)
      L.N = N; L.f8in = store8 ? 1 : 0;
      const int nb = launch_join_bwd_stream(L, s);
      MM_TRY(nb);
      auto reduce16 = [&](const float* parts, int nparts, const ConvW& w, int ntaps) {
        WgradReduceArgs u; std::memset(&u, 0, sizeof(u));
        u.part = parts; u.dW = grads + w.off; u.Ca = w.D0; u.Cb = w.D1; u.ntaps = ntaps; u.nparts = nparts;
        u.Ca_valid = w.D0; u.Cb_valid = w.D1; u.sA = w.D1 * ntaps; u.sB = ntaps; u.scale = w.wscale;
        for (int t = 0; t < ntaps; ++t) u.tap_off[t] = t;
        return launch_wgrad_reduce(u, s);
      };
      MM_TRY(reduce16(L.part2, nb, B.c2, 16));
      MM_TRY(reduce16(L.parts, nb, B.cs, 16));
      MM_TRY(bn_backward_coefs(B.b1, params, grads, base, nb, 1, 0, (double)npi, s));
      Conv1BwdLaunch C1;
      C1.da1 = base + P.da1; C1.y1 = base + B.y1; C1.ms = bnf(B.b1, base, 2); C1.mb = bnf(B.b1, base, 3);
      C1.A = bnf(B.b1, base, 4); C1.B = bnf(B.b1, base, 5); C1.C = bnf(B.b1, base, 6);
      C1.xin = xin; C1.pxs = xs; C1.pxb = xb; C1.w1u = base + plan_.packed + B.c1.packU * (long)esz();
      C1.gin = base + P.g[cur ^ 1]; C1.part = ws2; C1.nrows = (long)N * B.Hin;
      C1.mask_out = jg_block(i - 1) ? 1 : 0;
      g_masked = C1.mask_out != 0;
      const int nb1 = launch_conv1_bwd_stream(C1, s);
      MM_TRY(nb1);
      {   // conv1: Conv2d weight (out = 16, in = Cin): partial images [out][in]
        WgradReduceArgs u; std::memset(&u, 0, sizeof(u));
        u.part = ws2; u.dW = grads + B.c1.off; u.Ca = B.c1.D0; u.Cb = B.c1.D1; u.ntaps = 1; u.nparts = nb1;
        u.Ca_valid = B.c1.D0; u.Cb_valid = B.c1.D1; u.sA = B.c1.D1; u.sB = 1; u.scale = B.c1.wscale;
        MM_TRY(launch_wgrad_reduce(u, s));
      }
      MM_TRY(side_mark(i));
      cur ^= 1;
      continue;
    }
    if (store8 && i == nd - 1) { set_error("fp8 storage of the last up-block needs join_bwd_stream"); return MMVAE_ERR_UNSUPPORTED; }
    if (jg) {
      // (no apply pass: the two fused passes below evaluate dy2 / dys from g[cur], y2 / ys and the coefficients)
    } else if (from_tail)
      MM_TRY(launch_tail_join_bwd_apply(dt(), d_raw, params + tail.off, cfg.out_ch, N, Sd, Sd, bnf(B.b2, base, 2), bnf(B.b2, base, 3), bnf(B.bs, base, 2),
                                        bnf(B.bs, base, 3), base + B.y2, bnf(B.b2, base, 4), bnf(B.b2, base, 5), bnf(B.b2, base, 6), base + P.dy2[ds],
                                        base + B.ys, bnf(B.bs, base, 4), bnf(B.bs, base, 5), bnf(B.bs, base, 6), base + P.dys[ds], s));
    else
      MM_TRY(launch_bn_bwd_apply(dt(), base + P.g[cur], nullptr, bnf(B.b2, base, 2), bnf(B.b2, base, 3), base + B.y2, bnf(B.b2, base, 4), bnf(B.b2, base, 5),
                                 bnf(B.b2, base, 6), base + P.dy2[ds], ysp, B.identity ? ones(base) : bnf(B.bs, base, 4),
                                 B.identity ? zeros(base) : bnf(B.bs, base, 5), B.identity ? zeros(base) : bnf(B.bs, base, 6), base + dyso, npo, B.C, s,
                                 ssc, ssh));
    hipStream_t wsm = wgrad_stream(s);
    MM_TRY(side_fork(s));
    bool fuse_c2 = false, fuse_cs = false;
    int np_b1 = 0;                       // rows of bn1's backward sums when the fused pass has left them in `part`
    if (B.identity) {
      // conv2 (3x3 Conv2d): wgrad(P = dy2, G = a1 with BN+ReLU prologue); dgrad -> d_a1
      MM_TRY(run_wgrad(B.c2, N, base + P.dy2[ds], B.Hout, B.Wout, nullptr, nullptr, base + B.y1, B.Hin, B.Win, bnf(B.b1, base, 2), bnf(B.b1, base, 3),
                       grads, wsm));
      MM_TRY(run_up(B.c2, base, N, base + P.dy2[ds], B.Hout, B.Wout, base + P.da1, B.Hin, B.Win, nullptr, nullptr, 0, nullptr, 0, s));
    } else {
      // conv2 (ConvT k4 s2): wgrad(P = a1 small side with BN+ReLU prologue, G = dy2 large side); dgrad = strided conv -> d_a1
      // Where the shape allows (uplayer4 / uplayer5: 16 output channels, 16x16 -> 32x32 or 32x32 -> 64x64, bf16) dgrad and wgrad of a ConvT
      // are ONE pass over its dy tensor on the caller's stream (wgrad_stream_kernel with DG): dy2 / dys (671 MB each at N = 5120 in
      // uplayer5) are read once instead of twice.
      fuse_c2 = !B.c2.fp8 && op_bwd_fusable(dt(), geom(B.c2), N, B.Hin, B.Win, B.Hout, B.Wout);
      fuse_cs = !B.cs.fp8 && op_bwd_fusable(dt(), geom(B.cs), N, B.Hin, B.Win, B.Hout, B.Wout) && B.C == 16;
      float* wsc2 = reinterpret_cast<float*>(base + P.wscratch2);
      if (!fuse_c2)
        MM_TRY(run_wgrad(B.c2, N, base + B.y1, B.Hin, B.Win, bnf(B.b1, base, 2), bnf(B.b1, base, 3), base + P.dy2[ds], B.Hout, B.Wout, nullptr,
                         nullptr, grads, wsm));
      if (!fuse_cs) MM_TRY(run_wgrad(B.cs, N, xin, B.Hin, B.Win, xs, xb, base + P.dys[ds], B.Hout, B.Wout, nullptr, nullptr, grads, wsm));
      if (jg && !(fuse_c2 && fuse_cs)) { set_error("decoder_bwd: the join-gradient form needs both fused passes"); return MMVAE_ERR_UNSUPPORTED; }
      if (fuse_c2) {
        // ... and bn1's backward sums come out of the same pass (the data gradient is in registers, y1 is the pass's P operand)
        const JoinGrad j2{base + B.y2, bnf(B.b2, base, 4), bnf(B.b2, base, 5), bnf(B.b2, base, 6)};
        const int rcf = op_run_bwd_fused(dt(), geom(B.c2), N, base + B.y1, B.Hin, B.Win, bnf(B.b1, base, 2), bnf(B.b1, base, 3), 1,
                                         jg ? base + P.g[cur] : base + P.dy2[ds],
                                         B.Hout, B.Wout, base + plan_.packed + B.c2.packD * (long)esz(), base + P.da1, nullptr, nullptr,
                                         grads + B.c2.off, s, wsc2, B.c2.wscale, B.C == 16 ? part : nullptr, nullptr, 1.f, jg ? &j2 : nullptr);
        if (rcf <= 0) { if (rcf == 0) set_error("decoder_bwd: fused backward of %s not taken", "conv2"); return rcf < 0 ? rcf : MMVAE_ERR_UNSUPPORTED; }
        if (B.C == 16) np_b1 = rcf;
      } else
        MM_TRY(run_down(B.c2, base, N, base + P.dy2[ds], B.Hout, B.Wout, base + P.da1, B.Hin, B.Win, nullptr, nullptr, 0, nullptr, 0, dt(), s));
    }
    np = np_b1 > 0 ? np_b1 : launch_bn_bwd_reduce(dt(), base + P.da1, nullptr, bnf(B.b1, base, 2), bnf(B.b1, base, 3), base + B.y1, nullptr, npi, B.C, part, s);
    MM_TRY(np);
    MM_TRY(bn_backward_coefs(B.b1, params, grads, base, np, 1, 0, (double)npi, s));
    MM_TRY(launch_bn_bwd_apply(dt(), base + P.da1, nullptr, bnf(B.b1, base, 2), bnf(B.b1, base, 3), base + B.y1, bnf(B.b1, base, 4),
                               bnf(B.b1, base, 5), bnf(B.b1, base, 6), base + P.dy1[ds], nullptr, nullptr, nullptr, nullptr, nullptr, npi,
                               B.C, s));
    // conv1 (1x1): wgrad(P = dy1, G = xin); upsample (ConvT): wgrad(P = xin, G = dys)
    MM_TRY(side_fork(s));
    // (with the fused shortcut pass below the 1x1 conv's weight gradient comes out of that pass: dy1 and the block input are its rows)
    if (!fuse_cs) MM_TRY(run_wgrad(B.c1, N, base + P.dy1[ds], B.Hin, B.Win, nullptr, nullptr, xin, B.Hin, B.Win, xs, xb, grads, wsm));
    MM_TRY(side_mark(i));
    if (B.identity)     // d_xin already holds the shortcut's share: the 1x1 conv's data gradient is added to it
      MM_TRY(run_up(B.c1, base, N, base + P.dy1[ds], B.Hin, B.Win, base + P.g[cur ^ 1], B.Hin, B.Win, nullptr, nullptr, 0, nullptr, 1, s));
    else if (fuse_cs) {  // shortcut ConvT: weight gradient + data gradient + the 1x1 conv's share (dy1 (x) w1) and ITS weight gradient, one pass over dys
      const JoinGrad js{base + B.ys, bnf(B.bs, base, 4), bnf(B.bs, base, 5), bnf(B.bs, base, 6)};
      const int rcf = op_run_bwd_fused(dt(), geom(B.cs), N, xin, B.Hin, B.Win, xs, xb, 1, jg ? base + P.g[cur] : base + P.dys[ds], B.Hout, B.Wout,
                                       base + plan_.packed + B.cs.packD * (long)esz(), base + P.g[cur ^ 1], base + P.dy1[ds],
                                       base + plan_.packed + B.c1.packU * (long)esz(), grads + B.cs.off, s,
                                       reinterpret_cast<float*>(base + P.wscratch2), B.cs.wscale, nullptr, grads + B.c1.off, B.c1.wscale, jg ? &js : nullptr);
      if (rcf <= 0) { if (rcf == 0) set_error("decoder_bwd: fused backward of %s not taken", "upsample"); return rcf < 0 ? rcf : MMVAE_ERR_UNSUPPORTED; }
    } else              // one kernel: the 1x1 conv's data gradient (dy1, already on this block's input grid) is a second source of the shortcut's
      MM_TRY(run_down(B.cs, base, N, base + P.dys[ds], B.Hout, B.Wout, base + P.g[cur ^ 1], B.Hin, B.Win, nullptr, nullptr, 0, nullptr, 0, dt(), s,
                      &B.c1, base + P.dy1[ds]));
    cur ^= 1;
  }
  // ---- decoder stem
  if (dec.size() >= 2) MM_TRY(side_wait_mark(1, s));   // the stem uses dy set 1 (block index -1)
  const int ds = 1;
  {
    const long npix = (long)N * 4;
    np = launch_bn_bwd_reduce(dt(), base + P.g[cur], nullptr, bnf(dbn0, base, 2), bnf(dbn0, base, 3), base + P.y0d, nullptr, npix, 128, part, s);
    MM_TRY(np);
    MM_TRY(bn_backward_coefs(dbn0, params, grads, base, np, 1, 0, (double)npix, s));
    MM_TRY(launch_bn_bwd_apply(dt(), base + P.g[cur], nullptr, bnf(dbn0, base, 2), bnf(dbn0, base, 3), base + P.y0d, bnf(dbn0, base, 4),
                               bnf(dbn0, base, 5), bnf(dbn0, base, 6), base + P.dy1[ds], nullptr, nullptr, nullptr, nullptr, nullptr, npix, 128, s));
    MM_TRY(side_fork(s));
    MM_TRY(run_wgrad(dstem, N, base + P.enc_t, 1, 1, nullptr, nullptr, base + P.dy1[ds], 2, 2, nullptr, nullptr, grads, wgrad_stream(s)));
    if (d_enc) {
      MM_TRY(run_down(dstem, base, N, base + P.dy1[ds], 2, 2, base + P.dh, 1, 1, nullptr, nullptr, 0, nullptr, 0, dt(), s));
      MM_TRY(launch_convert(dt(), DT_F32, base + P.dh, d_enc, (long)N * cfg.z, s));
    }
    if (!defer_join_) MM_TRY(side_join(s));
  }
  return MMVAE_OK;
}

}  // namespace mmvae
