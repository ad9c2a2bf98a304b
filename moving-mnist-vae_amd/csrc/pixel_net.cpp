// PixelCNN (reference model.py:227-255: InstanceNorm -> [MaskedConv2d 7x7 -> InstanceNorm -> ReLU] x (layers - 1) -> MaskedConv2d 7x7)
// forward and backward on the generic convolution kernels: a type-A / type-B mask keeps the first 24 / 25 taps of the 7x7 kernel in
// row-major order (model.py:216-220), so every layer is a stride-1 convolution with that tap list -- the masked taps are never packed,
// multiplied or differentiated.  Activations: NHWC storage type, channel counts padded to multiples of 16 at the two boundaries
// (pixelcnn.hip converts from / to the reference's NCHW f32 tensors).  Owns no tensors: parameters, gradients and the workspace are the
// caller's.  Not a benchmarked path: correctness first, no fusion beyond InstanceNorm + ReLU.
#include "pixel_net.hpp"

#include <cstring>

namespace mmvae {

#define MM_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ < 0) return rc__;  \
  } while (0)

static inline long align_up(long v, long a) { return (v + a - 1) / a * a; }
static inline int pad16(int c) { return (c + 15) / 16 * 16; }

PixelNet::PixelNet(int in_ch_, int mid_, int out_ch_, int layers_, int dtype_) : in_ch(in_ch_), mid(mid_), out_ch(out_ch_), layers(layers_), dtype(dtype_) {
  long off = 0, poff = 0;
  for (int i = 0; i < layers; ++i) {
    Layer L;
    L.cin = i == 0 ? in_ch : mid; L.cout = i == layers - 1 ? out_ch : mid;
    L.cin_p = i == 0 ? pad16(in_ch) : mid; L.cout_p = i == layers - 1 ? pad16(out_ch) : mid;
    L.ntaps = i == 0 ? 24 : 25;                                   // type A: rows 0..2 and 3 taps of row 3; type B: + the centre
    L.w_off = off; off += (long)L.cout * L.cin * 49;
    L.b_off = off; off += L.cout;
    L.packF = poff; poff += align_up((long)L.cout_p * L.ntaps * L.cin_p, 8);
    L.packB = poff; poff += align_up((long)L.cin_p * L.ntaps * L.cout_p, 8);
    lay.push_back(L);
  }
  n_params = off; n_packed = poff;
}

const PixelPlan& PixelNet::plan(int N, int S) {
  if (plan_.N == N && plan_.S == S) return plan_;
  PixelPlan P; P.N = N; P.S = S;
  const long e = (long)dtype_size(dtype), HW = (long)S * S;
  long cur = 0;
  auto take = [&](long bytes) { long o = cur; cur = align_up(cur + (bytes > 16 ? bytes : 16), 256); return o; };
  P.x0 = take((long)N * HW * lay[0].cin_p * e);
  P.st0 = take((long)N * in_ch * 2 * 4);
  for (int i = 0; i < layers; ++i) {
    P.h[i] = take((long)N * HW * lay[i].cout_p * e);
    if (i < layers - 1) { P.a[i] = take((long)N * HW * mid * e); P.st[i] = take((long)N * mid * 2 * 4); }
  }
  P.g[0] = take((long)N * HW * (mid > 16 ? mid : 16) * e);
  P.g[1] = take((long)N * HW * (mid > 16 ? mid : 16) * e);
  P.packed = take(n_packed * e);
  P.bias_pad = take(16 * 4);
  P.partials = take(1024L * 2 * 256 * 4);
  P.wscratch = take((long)kWgradScratchBytes);
  P.bytes = (size_t)cur;
  plan_ = P;
  return plan_;
}

// taps of layer L: t = 7 kh + kw for the first ntaps positions; forward offset (kh - 3, kw - 3), data gradient the mirrored one
static void fill_taps(GatherArgs& a, int ntaps, bool mirrored) {
  for (int t = 0; t < ntaps; ++t) {
    const int kh = t / 7, kw = t % 7;
    a.taps[t] = mirrored ? Tap{3 - kh, 3 - kw} : Tap{kh - 3, kw - 3};
  }
}

int PixelNet::pack(const float* params, char* base, hipStream_t s) {
  const PixelPlan& P = plan_;
  const long e = (long)dtype_size(dtype);
  pack_batch_begin();
  for (const Layer& L : lay) {
    PackArgs f; std::memset(&f, 0, sizeof(f));     // forward: [cout_p][tap][cin_p]
    f.src = params + L.w_off; f.dst = base + P.packed + L.packF * e;
    f.cols = L.cout_p; f.cols_valid = L.cout; f.K = L.cin_p; f.K_valid = L.cin; f.ntaps = L.ntaps; f.s_col = L.cin * 49; f.s_k = 49; f.scale = 1.f;
    for (int t = 0; t < L.ntaps; ++t) f.tap_off[t] = t;
    MM_TRY(launch_pack(dtype, f, s));
    PackArgs b; std::memset(&b, 0, sizeof(b));     // data gradient: [cin_p][tap][cout_p]
    b.src = params + L.w_off; b.dst = base + P.packed + L.packB * e;
    b.cols = L.cin_p; b.cols_valid = L.cin; b.K = L.cout_p; b.K_valid = L.cout; b.ntaps = L.ntaps; b.s_col = 49; b.s_k = L.cin * 49; b.scale = 1.f;
    for (int t = 0; t < L.ntaps; ++t) b.tap_off[t] = t;
    MM_TRY(launch_pack(dtype, b, s));
  }
  return pack_batch_flush(dtype, s);
}

int PixelNet::forward(int N, int S, const float* x, const float* params, void* ws, size_t ws_bytes, float* out, hipStream_t s) {
  const PixelPlan& P = plan(N, S);
  if (ws_bytes < P.bytes) { set_error("pixelcnn: workspace too small: %zu < %zu", ws_bytes, P.bytes); return MMVAE_ERR_WORKSPACE; }
  char* base = static_cast<char*>(ws);
  const int HW = S * S;
  const long e = (long)dtype_size(dtype);
  MM_TRY(pack(params, base, s));
  MM_TRY(launch_inorm_planar_fwd(dtype, x, base + P.x0, reinterpret_cast<float*>(base + P.st0), N, in_ch, HW, s));      // model.py:249
  const void* cur = base + P.x0;
  for (int i = 0; i < layers; ++i) {
    const Layer& L = lay[i];
    float* bias = const_cast<float*>(params) + L.b_off;
    if (L.cout_p != L.cout) {     // last layer: the bias vector padded with zeros to the GEMM's 16 rows
      bias = reinterpret_cast<float*>(base + P.bias_pad);
      MM_TRY(launch_fill_f32(bias, 0.f, 16, s));
      if (hipMemcpyAsync(bias, params + L.b_off, (size_t)L.cout * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) { set_error("pixelcnn: bias copy failed"); return MMVAE_ERR_HIP; }
    }
    GatherArgs a; std::memset(&a, 0, sizeof(a));
    a.x = cur; a.w = base + P.packed + L.packF * e; a.y = base + P.h[i]; a.bias = bias;
    a.N = N; a.Hi = S; a.Wi = S; a.Cin = L.cin_p; a.Ho = S; a.Wo = S; a.Cout = L.cout_p; a.SI = 1; a.SO = 1;
    a.nphase = 1; a.phases[0] = Phase{0, 0, S, S, L.ntaps, 0, 0};
    fill_taps(a, L.ntaps, false);
    MM_TRY(launch_gather_gemm(dtype, dtype, a, s));                                                                       // model.py:223
    if (i < layers - 1) {
      MM_TRY(launch_inorm_nhwc_fwd(dtype, base + P.h[i], base + P.a[i], reinterpret_cast<float*>(base + P.st[i]), N, mid, HW, 1, s));   // :252-253
      cur = base + P.a[i];
    }
  }
  return launch_nhwc16_to_planar(dtype, base + P.h[layers - 1], out, N, out_ch, HW, s);
}

int PixelNet::backward(int N, int S, const float* x, const float* d_out, const float* params, float* grads, void* ws, size_t ws_bytes, float* d_x,
                       hipStream_t s) {
  const PixelPlan& P = plan(N, S);
  if (ws_bytes < P.bytes) { set_error("pixelcnn: workspace too small"); return MMVAE_ERR_WORKSPACE; }
  char* base = static_cast<char*>(ws);
  const int HW = S * S;
  const long e = (long)dtype_size(dtype);
  float* part = reinterpret_cast<float*>(base + P.partials);
  float* wsc = reinterpret_cast<float*>(base + P.wscratch);
  // gradient w.r.t. the last conv's output, padded to 16 channels
  MM_TRY(launch_planar_to_nhwc16(dtype, d_out, base + P.g[0], N, out_ch, HW, s));
  int cur = 0;
  for (int i = layers - 1; i >= 0; --i) {
    const Layer& L = lay[i];
    const void* dy = base + P.g[cur];                         // [N][S][S][cout_p]
    const void* xin = i == 0 ? base + P.x0 : base + P.a[i - 1];
    // bias gradient: per-channel sums of dy
    {
      const int np = launch_chan_stats_nhwc(dtype, dy, (long)N * HW, L.cout_p, part, s);
      MM_TRY(np);
      MM_TRY(launch_partials_add(part, np, 2 * L.cout_p, L.cout, grads + L.b_off, s));
    }
    // weight gradient over the layer's taps: dW[co][ci][t] += sum_pix dy[pix][co] * x[pix + off_t][ci]
    {
      WgradArgs w; std::memset(&w, 0, sizeof(w));
      w.P = dy; w.G = xin; w.dW = grads + L.w_off; w.scratch = wsc;
      w.N = N; w.Hp = S; w.Wp = S; w.Ca = L.cout_p; w.Ca_valid = L.cout; w.Hg = S; w.Wg = S; w.Cb = L.cin_p; w.Cb_valid = L.cin;
      w.stride = 1; w.pad = 3; w.ksz = 7; w.sA = L.cin * 49; w.sB = 49; w.ntaps = L.ntaps; w.scale = 1.f;
      for (int t = 0; t < L.ntaps; ++t) w.tap_off[t] = t;
      MM_TRY(launch_wgrad(dtype, w, s));
    }
    if (i == 0 && !d_x) break;
    // data gradient: dx[pix][ci] = sum_t sum_co dy[pix - off_t][co] * W[co][ci][t]
    {
      GatherArgs a; std::memset(&a, 0, sizeof(a));
      a.x = dy; a.w = base + P.packed + L.packB * e; a.y = base + P.g[cur ^ 1];
      a.N = N; a.Hi = S; a.Wi = S; a.Cin = L.cout_p; a.Ho = S; a.Wo = S; a.Cout = L.cin_p; a.SI = 1; a.SO = 1;
      a.nphase = 1; a.phases[0] = Phase{0, 0, S, S, L.ntaps, 0, 0};
      fill_taps(a, L.ntaps, true);
      MM_TRY(launch_gather_gemm(dtype, dtype, a, s));
    }
    cur ^= 1;
    if (i > 0) {
      // InstanceNorm + ReLU backward of layer i - 1's output (in place on the gradient buffer is not possible: g and dh differ per element only)
      MM_TRY(launch_inorm_nhwc_bwd(dtype, base + P.g[cur], base + P.h[i - 1], reinterpret_cast<const float*>(base + P.st[i - 1]), base + P.g[cur ^ 1], N, mid, HW,
                                   1, s));
      cur ^= 1;
    } else {
      MM_TRY(launch_inorm_planar_bwd(dtype, base + P.g[cur], x, reinterpret_cast<const float*>(base + P.st0), d_x, N, in_ch, HW, s));
    }
  }
  return MMVAE_OK;
}

}  // namespace mmvae
