#include "conv_ops.hpp"

#include <cstdlib>
#include <cstring>

namespace mmvae {

#define MM_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ < 0) return rc__;  \
  } while (0)

// stride phases of an "up" (transposed) gather: for output phase (ph,pw) the kernel taps with (ph+p-kh) % s == 0
struct UpPhase { int ph, pw; int ntaps; int kh[16], kw[16], dh[16], dw[16]; };
static int up_phases(int k, int s, int p, UpPhase* out) {
  int n = 0;
  for (int ph = 0; ph < s; ++ph)
    for (int pw = 0; pw < s; ++pw) {
      UpPhase& u = out[n++];
      u.ph = ph; u.pw = pw; u.ntaps = 0;
      for (int kh = 0; kh < k; ++kh) {
        const int th = ph + p - kh;
        if (((th % s) + s) % s != 0) continue;
        for (int kw = 0; kw < k; ++kw) {
          const int tw = pw + p - kw;
          if (((tw % s) + s) % s != 0) continue;
          u.kh[u.ntaps] = kh; u.kw[u.ntaps] = kw;
          u.dh[u.ntaps] = th / s; u.dw[u.ntaps] = tw / s;   // exact divisions
          ++u.ntaps;
        }
      }
    }
  return n;
}


int op_pack_down(int dt, const ConvGeom& g, const float* w, void* dst, hipStream_t s, float scale, int fp8, int frag) {
  PackArgs a; std::memset(&a, 0, sizeof(a));
  const int kk = g.k * g.k;
  if (kk > kMaxTaps) { set_error("pack_down: k=%d too large", g.k); return MMVAE_ERR_UNSUPPORTED; }
  a.src = w; a.dst = dst;
  a.cols = g.D0; a.K = g.D1; a.ntaps = kk; a.s_col = g.D1 * kk; a.s_k = kk; a.scale = scale; a.fp8 = fp8; a.frag = frag;
  for (int t = 0; t < kk; ++t) a.tap_off[t] = t;
  return launch_pack(dt, a, s);
}

int op_pack_up(int dt, const ConvGeom& g, const float* w, void* dst, hipStream_t s, float scale, int fp8, int frag) {
  if (g.s > 2 || g.k * g.k > kMaxTaps) { set_error("pack_up: k=%d s=%d unsupported", g.k, g.s); return MMVAE_ERR_UNSUPPORTED; }
  UpPhase ph[4];
  const int np = up_phases(g.k, g.s, g.p, ph);
  const int kk = g.k * g.k;
  const long e = fp8 ? 1 : (long)dtype_size(dt);
  long off = 0;
  for (int i = 0; i < np; ++i) {
    if (ph[i].ntaps == 0) continue;
    PackArgs a; std::memset(&a, 0, sizeof(a));
    a.src = w; a.dst = static_cast<char*>(dst) + off * e;
    a.cols = g.D1; a.K = g.D0; a.ntaps = ph[i].ntaps; a.s_col = kk; a.s_k = g.D1 * kk; a.scale = scale; a.fp8 = fp8; a.frag = frag;
    for (int t = 0; t < ph[i].ntaps; ++t) a.tap_off[t] = ph[i].kh[t] * g.k + ph[i].kw[t];
    MM_TRY(launch_pack(dt, a, s));
    off += (long)g.D1 * ph[i].ntaps * g.D0;
  }
  return MMVAE_OK;
}

// MMVAE_DEEP2_FRAG=0: keep the row-major [cout][tap][cin] packing for deep2_conv_kernel too (A/B of the layout; both are read correctly)
static bool frag_enabled() {
  constexpr bool v = true;
  return v;
}
// does deep2_conv_kernel take this conv's down / up form at this place (large-side map Hl x Wl)?  fp8: its e4m3 form
int op_deep2_down_ok(int dt, const ConvGeom& g, int Hl, int Wl, int fp8) {
  const int Hs = conv_down_size(Hl, g.k, g.s, g.p), Ws = conv_down_size(Wl, g.k, g.s, g.p);
  return deep2_shape_ok(dt, g.D1, g.D0, Hs, Ws, Hl, Wl, g.k * g.k, fp8) ? 1 : 0;
}
int op_deep2_up_ok(int dt, const ConvGeom& g, int Hl, int Wl, int allow_empty_phases, int fp8) {
  if (g.s > 2 || g.k * g.k > kMaxTaps) return 0;
  const int Hs = conv_down_size(Hl, g.k, g.s, g.p), Ws = conv_down_size(Wl, g.k, g.s, g.p);
  UpPhase ph[4];
  const int np = up_phases(g.k, g.s, g.p, ph);
  int ntaps = 0;
  for (int i = 0; i < np; ++i) { if (ph[i].ntaps == 0 && !allow_empty_phases) return 0; ntaps += ph[i].ntaps; }
  return deep2_shape_ok(dt, g.D0, g.D1, (Hl + g.s - 1) / g.s, (Wl + g.s - 1) / g.s, Hs, Ws, ntaps, fp8) ? 1 : 0;
}
// whether the bf16 position-major kernel (conv_pos.inc) takes this layer's FORWARD launch -- the conditions of try_pos (conv_gemm.hip).  The
// fp8 mode keeps such a layer in bf16: pos_conv_kernel at bf16 beats the e4m3 form of deep2_conv_kernel on these shapes (round 4).
bool op_pos_fwd_takes(const ConvGeom& g, int Hl, bool transposed) {
  const int Hs = conv_down_size(Hl, g.k, g.s, g.p);
  const int Hi = transposed ? Hs : Hl, Ho = transposed ? Hl : Hs;
  const int Cin = transposed ? g.D0 : g.D1, Cout = transposed ? g.D1 : g.D0;
  if (Cout % 32 != 0 || Cin % 8 != 0 || (Hi >= 8 && Cout < 128)) return false;
  if (transposed && g.s > 1 && g.k < g.s) return false;
  const int nw = Cout / 32 < 8 ? Cout / 32 : 8;
  if ((Cout / 32) % nw != 0 || (64 * nw) % (Cin / 8) != 0) return false;
  return pos_conv_takes(g.k, g.s, g.p, transposed ? 1 : 0, Hi, Ho, Cin);
}

int op_frag_down(int dt, const ConvGeom& g, int Hl, int Wl, int fp8) { return frag_enabled() ? op_deep2_down_ok(dt, g, Hl, Wl, fp8) : 0; }
int op_frag_up(int dt, const ConvGeom& g, int Hl, int Wl, int allow_empty_phases, int fp8) {
  return frag_enabled() ? op_deep2_up_ok(dt, g, Hl, Wl, allow_empty_phases, fp8) : 0;
}

int op_run_down(int dt, int out_dt, const ConvGeom& g, const void* packed, int N, const void* L, int Hl, int Wl, void* S, int Hs, int Ws,
                const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, hipStream_t s, const SecondSrc& x2) {
  GatherArgs a; std::memset(&a, 0, sizeof(a));
  if (g.k * g.k > kMaxTaps) { set_error("run_down: k=%d too large", g.k); return MMVAE_ERR_UNSUPPORTED; }
  if (x2.x2) { a.x2 = x2.x2; a.w2 = x2.w2; a.Cin2 = x2.Cin2; a.x2_ph = 0; a.x2_pw = 0; }
  a.fp8 = x2.fp8; a.wfrag = x2.wfrag; a.wfrag2 = x2.wfrag2;
  a.x = L; a.w = packed; a.y = S;
  a.pro_scale = pro_s; a.pro_shift = pro_b; a.pro_relu = relu; a.stats = stats; a.accumulate = accumulate;
  a.N = N; a.Hi = Hl; a.Wi = Wl; a.Cin = g.D1; a.Ho = Hs; a.Wo = Ws; a.Cout = g.D0; a.SI = g.s; a.SO = 1;
  a.gk = g.k; a.gs = g.s; a.gp = g.p; a.gup = 1;
  a.nphase = 1;
  a.phases[0] = Phase{0, 0, Hs, Ws, g.k * g.k, 0, 0};
  for (int kh = 0; kh < g.k; ++kh)
    for (int kw = 0; kw < g.k; ++kw) a.taps[kh * g.k + kw] = Tap{kh - g.p, kw - g.p};
  return launch_gather_gemm(dt, out_dt, a, s);
}

int op_run_up(int dt, const ConvGeom& g, const void* packed, int N, const void* S, int Hs, int Ws, void* L, int Hl, int Wl,
              const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, hipStream_t s, const SecondSrc& x2) {
  if (g.s > 2 || g.k * g.k > kMaxTaps) { set_error("run_up: k=%d s=%d unsupported", g.k, g.s); return MMVAE_ERR_UNSUPPORTED; }
  GatherArgs a; std::memset(&a, 0, sizeof(a));
  if (x2.x2) { a.x2 = x2.x2; a.w2 = x2.w2; a.Cin2 = x2.Cin2; a.x2_ph = 0; a.x2_pw = 0; }
  a.fp8 = x2.fp8; a.wfrag = x2.wfrag; a.wfrag2 = x2.wfrag2;
  a.x = S; a.w = packed; a.y = L;
  a.pro_scale = pro_s; a.pro_shift = pro_b; a.pro_relu = relu; a.stats = stats; a.accumulate = accumulate;
  a.N = N; a.Hi = Hs; a.Wi = Ws; a.Cin = g.D0; a.Ho = Hl; a.Wo = Wl; a.Cout = g.D1; a.SI = 1; a.SO = g.s;
  a.gk = g.k; a.gs = g.s; a.gp = g.p; a.gup = 2;
  UpPhase ph[4];
  const int np = up_phases(g.k, g.s, g.p, ph);
  long off = 0; int tap0 = 0; a.nphase = 0;
  for (int i = 0; i < np; ++i) {
    const int Hq = Hl > ph[i].ph ? (Hl - ph[i].ph + g.s - 1) / g.s : 0;
    const int Wq = Wl > ph[i].pw ? (Wl - ph[i].pw + g.s - 1) / g.s : 0;
    const bool skip = (ph[i].ntaps == 0 && accumulate) || Hq == 0 || Wq == 0;
    if (!skip) {
      a.phases[a.nphase++] = Phase{ph[i].ph, ph[i].pw, Hq, Wq, ph[i].ntaps, tap0, off};
      for (int t = 0; t < ph[i].ntaps; ++t) a.taps[tap0 + t] = Tap{ph[i].dh[t], ph[i].dw[t]};
      tap0 += ph[i].ntaps;
    }
    off += (long)g.D1 * ph[i].ntaps * g.D0;
  }
  if (a.nphase == 0) return 1;
  return launch_gather_gemm(dt, dt, a, s);
}

int op_run_wgrad(int dt, const ConvGeom& g, int N, const void* P, int Hs, int Ws, const float* proP_s, const float* proP_b, int proP_relu,
                 const void* G, int Hl, int Wl, const float* proG_s, const float* proG_b, int proG_relu, float* dW, hipStream_t s,
                 float* scratch, float scale, WgradReduceArgs* defer) {
  WgradArgs a; std::memset(&a, 0, sizeof(a));
  const int kk = g.k * g.k;
  if (kk > 25) { set_error("wgrad: k=%d too large", g.k); return MMVAE_ERR_UNSUPPORTED; }
  if (defer) defer->nparts = 0;
  a.P = P; a.G = G; a.dW = dW; a.scratch = scratch; a.defer = defer;
  a.proP_scale = proP_s; a.proP_shift = proP_b; a.proP_relu = proP_relu;
  a.proG_scale = proG_s; a.proG_shift = proG_b; a.proG_relu = proG_relu;
  a.N = N; a.Hp = Hs; a.Wp = Ws; a.Ca = g.D0; a.Hg = Hl; a.Wg = Wl; a.Cb = g.D1; a.Cb_valid = g.D1;
  a.stride = g.s; a.pad = g.p; a.ksz = g.k; a.sA = g.D1 * kk; a.sB = kk; a.ntaps = kk; a.scale = scale;
  for (int t = 0; t < kk; ++t) a.tap_off[t] = t;
  return launch_wgrad(dt, a, s);
}

static void wgrad_args(WgradArgs& a, const ConvGeom& g, int N, const void* P, int Hs, int Ws, const float* proP_s, const float* proP_b, int proP_relu,
                       const void* G, int Hl, int Wl, const float* proG_s, const float* proG_b, int proG_relu, float* dW, float* scratch, float scale) {
  std::memset(&a, 0, sizeof(a));
  const int kk = g.k * g.k;
  a.P = P; a.G = G; a.dW = dW; a.scratch = scratch;
  a.proP_scale = proP_s; a.proP_shift = proP_b; a.proP_relu = proP_relu;
  a.proG_scale = proG_s; a.proG_shift = proG_b; a.proG_relu = proG_relu;
  a.N = N; a.Hp = Hs; a.Wp = Ws; a.Ca = g.D0; a.Hg = Hl; a.Wg = Wl; a.Cb = g.D1; a.Cb_valid = g.D1; a.Ca_valid = g.D0;
  a.stride = g.s; a.pad = g.p; a.ksz = g.k; a.sA = g.D1 * kk; a.sB = kk; a.ntaps = kk; a.scale = scale;
  for (int t = 0; t < kk && t < 25; ++t) a.tap_off[t] = t;
}

// A 3x3 stride-2 conv and the 1x1 stride-2 shortcut on the same input (a residual block's conv1 / downsample.0): both weight gradients from one
// pass over the block input.  Returns 1 when taken, 0 when the shapes are not the stream kernel's (the caller runs the two separately), <0 on error.
int op_run_wgrad_pair(int dt, const ConvGeom& g, const ConvGeom& gs, int N, const void* P, const void* P2, int Hs, int Ws, const void* G, int Hl, int Wl,
                      const float* proG_s, const float* proG_b, int proG_relu, float* dW, float* dW2, hipStream_t s, float* scratch, float scale,
                      float scale2) {
  if (g.k != 3 || g.p != 1 || gs.k != 1 || gs.p != 0 || gs.s != g.s || gs.D0 != g.D0 || gs.D1 != g.D1) return 0;
  WgradArgs a;
  wgrad_args(a, g, N, P, Hs, Ws, nullptr, nullptr, 0, G, Hl, Wl, proG_s, proG_b, proG_relu, dW, scratch, scale);
  return try_wgrad_stream_pair(dt, a, P2, dW2, scale2, s);
}

// whether op_run_wgrad runs this layer on the per-wave stream kernel (whose partial images are <= 19 MB and whose reduce can be deferred)
bool op_wgrad_is_stream(int dt, const ConvGeom& g, int N, int Hs, int Ws, int Hl, int Wl, bool proP, bool proG) {
  if (g.k * g.k > 25) return false;
  WgradArgs a;
  float dummy = 0.f;
  wgrad_args(a, g, N, nullptr, Hs, Ws, proP ? &dummy : nullptr, proP ? &dummy : nullptr, 1, nullptr, Hl, Wl, proG ? &dummy : nullptr, proG ? &dummy : nullptr, 1,
             nullptr, &dummy, 1.f);
  return wgrad_stream_shape(dt, a);
}

bool op_bwd_fusable(int dt, const ConvGeom& g, int N, int Hs, int Ws, int Hl, int Wl) {
  if (g.k * g.k > 25) return false;
  WgradArgs a;
  float dummy = 0.f;
  wgrad_args(a, g, N, nullptr, Hs, Ws, nullptr, nullptr, 0, nullptr, Hl, Wl, nullptr, nullptr, 0, nullptr, &dummy, 1.f);
  return dgrad_wgrad_stream_shape(dt, a);
}

int op_run_bwd_fused(int dt, const ConvGeom& g, int N, const void* P, int Hs, int Ws, const float* proP_s, const float* proP_b, int proP_relu,
                     const void* G, int Hl, int Wl, const void* packed_down, void* dP, const void* x2, const void* w2_packed, float* dW,
                     hipStream_t s, float* scratch, float scale, float* bn_part, float* dW2, float scale2, const JoinGrad* jg) {
  if (g.k * g.k > 25) return 0;
  WgradArgs a;
  wgrad_args(a, g, N, P, Hs, Ws, proP_s, proP_b, proP_relu, G, Hl, Wl, nullptr, nullptr, 0, dW, scratch, scale);
  a.M = a.N * a.Hp * a.Wp;
  return try_dgrad_wgrad_stream(dt, a, packed_down, dP, x2, w2_packed, bn_part, s, dW2, scale2, jg);
}

// whether op_run_bwd_fused takes this layer with a JoinGrad (see kernels.hpp)
bool op_bwd_fusable_jg(int dt, const ConvGeom& g, int N, int Hs, int Ws, int Hl, int Wl, bool proP, bool has_x2, bool bn_sums) {
  if (g.k * g.k > 25) return false;
  WgradArgs a;
  float dummy = 0.f;
  wgrad_args(a, g, N, nullptr, Hs, Ws, proP ? &dummy : nullptr, proP ? &dummy : nullptr, 1, nullptr, Hl, Wl, nullptr, nullptr, 0, nullptr, &dummy, 1.f);
  return dgrad_wgrad_stream_jg_shape(dt, a, has_x2, bn_sums);
}

}  // namespace mmvae
