// PixelCNN orchestration (reference model.py:212-255); see pixel_net.cpp.
#pragma once
#include <vector>

#include "kernels.hpp"

namespace mmvae {

struct PixelPlan {
  int N = -1, S = -1;
  size_t bytes = 0;
  long x0, st0;                       // instance-normalised input (NHWC, 16 padded channels), its (mean, istd) per (n, c)
  long h[16], a[16], st[16];          // per layer: conv output, relu(IN(.)) and the InstanceNorm statistics
  long g[2];                          // backward: gradient ping-pong buffers
  long packed, bias_pad, partials, wscratch;
};

class PixelNet {
 public:
  PixelNet(int in_ch, int mid, int out_ch, int layers, int dtype);
  int in_ch, mid, out_ch, layers, dtype;
  long n_params = 0;                  // flat f32 parameters: per layer weight (cout, cin, 7, 7) then bias (cout)
  struct Layer { int cin, cout, cin_p, cout_p, ntaps; long w_off, b_off, packF, packB; };
  std::vector<Layer> lay;
  const PixelPlan& plan(int N, int S);
  size_t workspace_bytes(int N, int S) { return plan(N, S).bytes; }
  // x [N][in_ch][S][S] f32 -> out [N][out_ch][S][S] f32 (logits)
  int forward(int N, int S, const float* x, const float* params, void* ws, size_t ws_bytes, float* out, hipStream_t s);
  // grads (same layout as params) +=; d_x (nullable) [N][in_ch][S][S] f32.  x: the forward's input (its statistics are in the workspace).
  int backward(int N, int S, const float* x, const float* d_out, const float* params, float* grads, void* ws, size_t ws_bytes, float* d_x, hipStream_t s);

 private:
  long n_packed = 0;
  PixelPlan plan_;
  int pack(const float* params, char* base, hipStream_t s);
};

}  // namespace mmvae
