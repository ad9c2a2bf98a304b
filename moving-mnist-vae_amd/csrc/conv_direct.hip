// Direct (non-MFMA) bandwidth kernels for the two layers that do not map to MFMA tiles:
//   stem  encoder.conv1  Conv2d(1 -> 32, k5 s2 p2)       (model.py:94)
//   tail  decoder.conv2  Conv2d(16 -> out_ch, k3 s1 p1)  (model.py:172), f32 NCHW output feeding the output BatchNorm
#include "kernels.hpp"

namespace mmvae {

constexpr int kStemC = 32;   // max stem output channels
constexpr int kTailC = 16;   // tail input channels (out_channels in {1,2,3,4,8})

static int grid_for(long work, int threads, int cap = 4096) {
  long b = (work + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ---------------------------------------------------------------- stem forward
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                       int N, int H, int W, int Ho, int Wo, int Cout) {
  constexpr int VE = Elem<T>::kVec;
  __shared__ float sw[25 * kStemC];
  for (int i = threadIdx.x; i < 25 * kStemC; i += blockDim.x) {
    const int tap = i / kStemC, co = i - tap * kStemC;
    sw[i] = co < Cout ? w[co * 25 + tap] : 0.f;
  }
  __syncthreads();
  const long M = (long)N * Ho * Wo;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    const int n = (int)(m / (Ho * Wo)), rem = (int)(m - (long)n * Ho * Wo);
    const int ho = rem / Wo, wo = rem - ho * Wo;
    float acc[kStemC];
#pragma unroll
    for (int c = 0; c < kStemC; ++c) acc[c] = 0.f;
    const T* xi = x + (long)n * H * W;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      const int hi = 2 * ho - 2 + kh;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int wi = 2 * wo - 2 + kw;
        if (wi < 0 || wi >= W) continue;
        const float xv = Elem<T>::load(xi + hi * W + wi);
        const float* wr = sw + (kh * 5 + kw) * kStemC;
#pragma unroll
        for (int c = 0; c < kStemC; ++c) acc[c] += xv * wr[c];
      }
    }
    T* yo = y + m * Cout;
#pragma unroll
    for (int c = 0; c < kStemC; c += VE)
      if (c < Cout) *reinterpret_cast<Vec16*>(yo + c) = Elem<T>::pack(acc + c);
  }
}

int launch_stem_fwd(int dt, const void* x, const float* w, void* y, int N, int H, int W, int Ho, int Wo, int Cout, hipStream_t s) {
  if (Cout > kStemC || Cout % 8) { set_error("stem_fwd: Cout=%d unsupported", Cout); return MMVAE_ERR_UNSUPPORTED; }
  const long M = (long)N * Ho * Wo;
  if (M <= 0) return MMVAE_OK;
  const int blocks = grid_for(M, 256);
  if (dt == DT_F32) hipLaunchKernelGGL((stem_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)x, w, (float*)y, N, H, W, Ho, Wo, Cout);
  else hipLaunchKernelGGL((stem_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, w, (bf16_t*)y, N, H, W, Ho, Wo, Cout);
  return check_launch("stem_fwd");
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

// ---------------------------------------------------------------- tail forward
template <typename T, int OC>
__global__ __launch_bounds__(256) void tail_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int N, int H, int W) {
  constexpr int VE = Elem<T>::kVec;
  __shared__ float sw[9 * kTailC * OC];   // [tap][ci][oc]
  for (int i = threadIdx.x; i < 9 * kTailC * OC; i += blockDim.x) {
    const int oc = i % OC, q = i / OC, ci = q % kTailC, tap = q / kTailC;
    sw[i] = w[(oc * kTailC + ci) * 9 + tap];
  }
  __syncthreads();
  const long M = (long)N * H * W;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    const int n = (int)(m / (H * W)), rem = (int)(m - (long)n * H * W);
    const int h = rem / W, ww = rem - h * W;
    float acc[OC];
#pragma unroll
    for (int o = 0; o < OC; ++o) acc[o] = bias ? bias[o] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hi = h - 1 + kh;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wi = ww - 1 + kw;
        if (wi < 0 || wi >= W) continue;
        const T* px = x + (((long)n * H + hi) * W + wi) * kTailC;
        float f[kTailC];
#pragma unroll
        for (int v = 0; v < kTailC / VE; ++v) Elem<T>::unpack(reinterpret_cast<const Vec16*>(px)[v], f + v * VE);
        const float* wr = sw + (kh * 3 + kw) * kTailC * OC;
#pragma unroll
        for (int ci = 0; ci < kTailC; ++ci)
#pragma unroll
          for (int o = 0; o < OC; ++o) acc[o] += f[ci] * wr[ci * OC + o];
      }
    }
#pragma unroll
    for (int o = 0; o < OC; ++o) y[((long)n * OC + o) * H * W + rem] = acc[o];
  }
}

#define MMVAE_OC_SWITCH(OCV, CALL)                 \
  switch (OCV) {                                   \
    case 1: { constexpr int OC_ = 1; CALL; } break; \
    case 2: { constexpr int OC_ = 2; CALL; } break; \
    case 3: { constexpr int OC_ = 3; CALL; } break; \
    case 4: { constexpr int OC_ = 4; CALL; } break; \
    case 8: { constexpr int OC_ = 8; CALL; } break; \
    default: set_error("tail: out_channels=%d unsupported (1,2,3,4,8)", OCV); return MMVAE_ERR_UNSUPPORTED; \
  }

int launch_tail_fwd(int dt, const void* x, const float* w, const float* bias, float* y, int N, int H, int W, int OC, hipStream_t s) {
  const long M = (long)N * H * W;
  if (M <= 0) return MMVAE_OK;
  const int blocks = grid_for(M, 256);
  if (dt == DT_F32) { MMVAE_OC_SWITCH(OC, hipLaunchKernelGGL((tail_fwd_kernel<float, OC_>), dim3(blocks), dim3(256), 0, s, (const float*)x, w, bias, y, N, H, W)) }
  else { MMVAE_OC_SWITCH(OC, hipLaunchKernelGGL((tail_fwd_kernel<bf16_t, OC_>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, w, bias, y, N, H, W)) }
  return check_launch("tail_fwd");
}

// ---------------------------------------------------------------- tail data gradient
template <typename T, int OC>
__global__ __launch_bounds__(256) void tail_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                         int N, int H, int W) {
  constexpr int VE = Elem<T>::kVec;
  __shared__ float sw[9 * OC * kTailC];   // [tap][oc][ci]
  for (int i = threadIdx.x; i < 9 * OC * kTailC; i += blockDim.x) {
    const int ci = i % kTailC, q = i / kTailC, oc = q % OC, tap = q / OC;
    sw[i] = w[(oc * kTailC + ci) * 9 + tap];
  }
  __syncthreads();
  const long M = (long)N * H * W;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    const int n = (int)(m / (H * W)), rem = (int)(m - (long)n * H * W);
    const int h = rem / W, ww = rem - h * W;
    float acc[kTailC];
#pragma unroll
    for (int c = 0; c < kTailC; ++c) acc[c] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ho = h + 1 - kh;
      if (ho < 0 || ho >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wo = ww + 1 - kw;
        if (wo < 0 || wo >= W) continue;
#pragma unroll
        for (int o = 0; o < OC; ++o) {
          const float g = dy[((long)n * OC + o) * H * W + ho * W + wo];
          const float* wr = sw + ((kh * 3 + kw) * OC + o) * kTailC;
#pragma unroll
          for (int c = 0; c < kTailC; ++c) acc[c] += g * wr[c];
        }
      }
    }
    T* po = dx + m * kTailC;
#pragma unroll
    for (int v = 0; v < kTailC / VE; ++v) reinterpret_cast<Vec16*>(po)[v] = Elem<T>::pack(acc + v * VE);
  }
}

int launch_tail_dgrad(int dt, const float* dy, const float* w, void* dx, int N, int H, int W, int OC, hipStream_t s) {
  const long M = (long)N * H * W;
  if (M <= 0) return MMVAE_OK;
  const int blocks = grid_for(M, 256);
  if (dt == DT_F32) { MMVAE_OC_SWITCH(OC, hipLaunchKernelGGL((tail_dgrad_kernel<float, OC_>), dim3(blocks), dim3(256), 0, s, dy, w, (float*)dx, N, H, W)) }
  else { MMVAE_OC_SWITCH(OC, hipLaunchKernelGGL((tail_dgrad_kernel<bf16_t, OC_>), dim3(blocks), dim3(256), 0, s, dy, w, (bf16_t*)dx, N, H, W)) }
  return check_launch("tail_dgrad");
}

// ---------------------------------------------------------------- tail weight gradient
// One thread per INPUT pixel q: x[q][0..15] is loaded once and scattered into the 9 taps that see it.
template <typename T>
__global__ __launch_bounds__(256) void tail_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dW,
                                                         float* __restrict__ dbias, int N, int H, int W, int OC, int oc) {
  constexpr int VE = Elem<T>::kVec;
  __shared__ float sRed[4 * (9 * kTailC + 1)];
  float acc[9][kTailC];
  float bsum = 0.f;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int c = 0; c < kTailC; ++c) acc[tp][c] = 0.f;
  const long M = (long)N * H * W;
  for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
    const int n = (int)(m / (H * W)), rem = (int)(m - (long)n * H * W);
    const int hq = rem / W, wq = rem - hq * W;
    float f[kTailC];
    const T* px = x + m * kTailC;
#pragma unroll
    for (int v = 0; v < kTailC / VE; ++v) Elem<T>::unpack(reinterpret_cast<const Vec16*>(px)[v], f + v * VE);
    const float* dplane = dy + ((long)n * OC + oc) * H * W;
    bsum += dplane[rem];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = hq + 1 - kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w2 = wq + 1 - kw;
        if (w2 < 0 || w2 >= W) continue;
        const float g = dplane[h * W + w2];
#pragma unroll
        for (int c = 0; c < kTailC; ++c) acc[kh * 3 + kw][c] += g * f[c];
      }
    }
  }
  // block reduction of 145 values
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int c = 0; c < kTailC; ++c) {
      const float sred = wave_sum(acc[tp][c]);
      if (lane == 0) sRed[wid * (9 * kTailC + 1) + tp * kTailC + c] = sred;
    }
  bsum = wave_sum(bsum);
  if (lane == 0) sRed[wid * (9 * kTailC + 1) + 9 * kTailC] = bsum;
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * kTailC + 1; i += blockDim.x) {
    const float sred = sRed[i] + sRed[(9 * kTailC + 1) + i] + sRed[2 * (9 * kTailC + 1) + i] + sRed[3 * (9 * kTailC + 1) + i];
    if (i < 9 * kTailC) {
      const int tp = i / kTailC, c = i - tp * kTailC;
      atomicAdd(dW + (oc * kTailC + c) * 9 + tp, sred);
    } else if (dbias) {
      atomicAdd(dbias + oc, sred);
    }
  }
}

int launch_tail_wgrad(int dt, const void* x, const float* dy, float* dW, float* dbias, int N, int H, int W, int OC, hipStream_t s) {
  const long M = (long)N * H * W;
  if (M <= 0) return MMVAE_OK;
  int blocks = grid_for(M, 256 * 8, 1024);
  for (int oc = 0; oc < OC; ++oc) {
    if (dt == DT_F32) hipLaunchKernelGGL((tail_wgrad_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)x, dy, dW, dbias, N, H, W, OC, oc);
    else hipLaunchKernelGGL((tail_wgrad_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, dy, dW, dbias, N, H, W, OC, oc);
  }
  return check_launch("tail_wgrad");
}

}  // namespace mmvae
