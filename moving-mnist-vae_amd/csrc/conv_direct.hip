// im2col of the 1-channel stem input, encoder.conv1 = Conv2d(1 -> 32, k5 s2 p2) (model.py:94): the operand of the stem's weight
// gradient at image sizes the one-pass stem backward (stem_bwd.hip) does not take.  (The direct VALU stem / tail kernels of round 1
// lived here; nothing reached them any more and they were removed in round 3.)
#include "kernels.hpp"

namespace mmvae {

static int grid_for(long work, int threads, int cap = 4096) {
  long b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ---------------------------------------------------------------- stem im2col (for the MFMA weight gradient)
template <typename T>
__global__ void stem_im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int H, int W, int Ho, int Wo) {
  constexpr int VE = Elem<T>::kVec;
  constexpr int NV = 32 / VE;        // 16-byte vectors per output row
  const long total = (long)N * Ho * Wo * NV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / NV;
    const int v = (int)(i - m * NV);
    const int n = (int)(m / (Ho * Wo)), rem = (int)(m - (long)n * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    float f[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      const int tap = v * VE + j;
      float val = 0.f;
      if (tap < 25) {
        const int kh = tap / 5, kw = tap - kh * 5;
        const int hi = 2 * ho - 2 + kh, wi = 2 * wo - 2 + kw;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) val = Elem<T>::load(x + ((long)n * H + hi) * W + wi);
      }
      f[j] = val;
    }
    reinterpret_cast<Vec16*>(col)[i] = Elem<T>::pack(f);
  }
}

int launch_stem_im2col(int dt, const void* x, void* col, int N, int H, int W, int Ho, int Wo, hipStream_t s) {
  const long total = (long)N * Ho * Wo * (dt == DT_F32 ? 8 : 4);
  if (total <= 0) return MMVAE_OK;
  const int blocks = grid_for(total, 256);
  if (dt == DT_F32) hipLaunchKernelGGL((stem_im2col_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)col, N, H, W, Ho, Wo);
  else hipLaunchKernelGGL((stem_im2col_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)col, N, H, W, Ho, Wo);
  return check_launch("stem_im2col");
}

}  // namespace mmvae
