// Geometry-level conv operations shared by the network orchestrator and the single-op C ABI.
// A conv-like weight [D0][D1][k][k] relates a "small" tensor S [N,Hs,Ws,D0] and a "large" tensor L [N,Hl,Wl,D1]
// with Hs = floor((Hl + 2p - k)/s) + 1:
//   Conv2d          weight (out,in,k,k):  forward = down(L->S), dgrad = up(S->L),   wgrad(P = dy (S), G = x (L))
//   ConvTranspose2d weight (in,out,k,k):  forward = up(S->L),   dgrad = down(L->S), wgrad(P = x (S),  G = dy (L))
#pragma once
#include "kernels.hpp"

namespace mmvae {

struct ConvGeom { int D0, D1, k, s, p; };

inline int conv_down_size(int H, int k, int s, int p) { return (H + 2 * p - k) / s + 1; }
// packed element counts (both equal numel(weight)); 16-byte alignment is the caller's job
// scale multiplies the weights (fp8 layers: the static per-layer scale); fp8 = 1 writes e4m3 bytes (batched packing only)
// frag = 1: fragment-major order (PackArgs::frag) -- what deep2_conv_kernel reads; op_frag_down / op_frag_up say whether the run_down /
// run_up launch of this geometry (large-side map Hl x Wl) goes to that kernel, i.e. whether to pack and run with the flag set.
int op_pack_down(int dt, const ConvGeom& g, const float* w, void* dst, hipStream_t s, float scale = 1.f, int fp8 = 0, int frag = 0);
int op_pack_up(int dt, const ConvGeom& g, const float* w, void* dst, hipStream_t s, float scale = 1.f, int fp8 = 0, int frag = 0);
int op_frag_down(int dt, const ConvGeom& g, int Hl, int Wl, int fp8 = 0);
// the same geometry test without the MMVAE_DEEP2_FRAG developer switch: does deep2_conv_kernel (fp8: its e4m3 form) take this launch?
int op_deep2_down_ok(int dt, const ConvGeom& g, int Hl, int Wl, int fp8 = 0);
int op_deep2_up_ok(int dt, const ConvGeom& g, int Hl, int Wl, int allow_empty_phases = 0, int fp8 = 0);
// allow_empty_phases: the launch accumulates (or is a second source), so stride phases without a tap are skipped, not zero-filled
int op_frag_up(int dt, const ConvGeom& g, int Hl, int Wl, int allow_empty_phases = 0, int fp8 = 0);
// x2 / w2 / Cin2 (optional): a second tensor on the q grid (= S for run_down, = the S-resolution grid for run_up) whose 1x1
// convolution with the packed [Cout][Cin2] matrix w2 is added into the result (phase (0,0) of run_up) in the same kernel.
struct SecondSrc {
  const void* x2 = nullptr; const void* w2 = nullptr; int Cin2 = 0;
  int fp8 = 0;             // the forward conv runs on the fp8 MFMA (e4m3 packed weights)
  int wfrag = 0, wfrag2 = 0;   // packed (w2) is fragment-major
};
int op_run_down(int dt, int out_dt, const ConvGeom& g, const void* packed, int N, const void* L, int Hl, int Wl, void* S, int Hs, int Ws,
                const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, hipStream_t s, const SecondSrc& x2 = SecondSrc());
int op_run_up(int dt, const ConvGeom& g, const void* packed, int N, const void* S, int Hs, int Ws, void* L, int Hl, int Wl,
              const float* pro_s, const float* pro_b, int relu, float* stats, int accumulate, hipStream_t s, const SecondSrc& x2 = SecondSrc());
int op_run_wgrad(int dt, const ConvGeom& g, int N, const void* P, int Hs, int Ws, const float* proP_s, const float* proP_b, int proP_relu,
                 const void* G, int Hl, int Wl, const float* proG_s, const float* proG_b, int proG_relu, float* dW, hipStream_t s,
                 float* scratch = nullptr, float scale = 1.f, WgradReduceArgs* defer = nullptr);
bool op_wgrad_is_stream(int dt, const ConvGeom& g, int N, int Hs, int Ws, int Hl, int Wl, bool proP, bool proG);
int op_run_wgrad_pair(int dt, const ConvGeom& g, const ConvGeom& gs, int N, const void* P, const void* P2, int Hs, int Ws, const void* G, int Hl, int Wl,
                      const float* proG_s, const float* proG_b, int proG_relu, float* dW, float* dW2, hipStream_t s, float* scratch, float scale,
                      float scale2);

// Weight gradient AND the data gradient w.r.t. the small-side tensor P in ONE pass over G (wgrad_stream_kernel with DG; the 16 -> 16
// channel k4 s2 layers on 32x32 -> 64x64 maps, bf16):  dW += P^T (x) G;  dP = down(G) with the conv's packed down form (+ x2 (x) w2,
// the 1x1 shortcut's share, x2 on P's grid with 16 channels, w2 its packed up form).  Returns 1 when taken, 0 when the shape is not
// that kernel's (callers then run op_run_down + op_run_wgrad), <0 on error.  bn_part (optional; P prologue'd, no x2): the pass also
// leaves the BatchNorm-backward partial sums of P's BatchNorm, rows [return value][2][16] -- the layout launch_bn_bwd_reduce writes.
// Taken: returns the number of rows (> 0).  scratch: kWgradScratchBytes, not shared with a concurrent wgrad.
// dW2 (optional, with x2): the weight gradient of the 1x1 conv itself, [16][D0] row-major (Conv2d (out, in, 1, 1)) += scale2 * x2^T (x) pro(P):
// both rows are in LDS for the pass anyway.
bool op_pos_fwd_takes(const ConvGeom& g, int Hl, bool transposed);   // the bf16 position-major kernel takes the layer's forward launch
bool op_bwd_fusable(int dt, const ConvGeom& g, int N, int Hs, int Ws, int Hl, int Wl);
int op_run_bwd_fused(int dt, const ConvGeom& g, int N, const void* P, int Hs, int Ws, const float* proP_s, const float* proP_b, int proP_relu,
                     const void* G, int Hl, int Wl, const void* packed_down, void* dP, const void* x2, const void* w2_packed, float* dW,
                     hipStream_t s, float* scratch, float scale = 1.f, float* bn_part = nullptr, float* dW2 = nullptr, float scale2 = 1.f,
                     const JoinGrad* jg = nullptr);
// jg (optional): G is the masked gradient of a residual join's output and the layer's dy is evaluated on load (kernels.hpp, JoinGrad) -- no
// bn_bwd_apply pass, no dy tensor.  op_bwd_fusable_jg: whether that form exists for the layer (16-wide maps; conv2 with prologue + BatchNorm sums,
// or the 32-channel shortcut with the 1x1 conv's second source).
bool op_bwd_fusable_jg(int dt, const ConvGeom& g, int N, int Hs, int Ws, int Hl, int Wl, bool proP, bool has_x2, bool bn_sums);

}  // namespace mmvae
