// PixelCNN pieces that are not convolutions (reference model.py:227-255): InstanceNorm2d (affine = False, no running statistics: instance
// statistics in train AND eval mode, eps 1e-5) forward / backward, fused with the ReLU behind it, and the boundary layout changes between
// the reference's NCHW f32 tensors and the NHWC storage-type tensors (channels zero-padded to 16) the convolution kernels work on.
// One block per image: its H*W*C elements are reduced in a first sweep and re-read (L2-hot) in a second.  The masked 7x7 convolutions
// themselves run on the generic conv kernels: a type-A / type-B mask (model.py:216-220) keeps exactly the first 24 / 25 taps of the 7x7
// kernel in row-major order, i.e. a convolution with that tap list.
#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

constexpr float kInEps = 1e-5f;

// block-wide sums of NV doubles per thread, result broadcast to every thread (blockDim = 256)
template <int NV>
__device__ __forceinline__ void block_sum_d(double (&v)[NV], double* smem) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum_d(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) smem[wid * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (smem[i] + smem[NV + i]) + (smem[2 * NV + i] + smem[3 * NV + i]);
}

// x [N][C][HW] f32  ->  xn [N][HW][16] of T = instance-normalised x in channels < C, zeros above; stats [N][C][2] = (mean, istd)
template <typename T>
__global__ __launch_bounds__(256) void inorm_planar_fwd_kernel(const float* __restrict__ x, T* __restrict__ xn, float* __restrict__ stats, int C, int HW) {
  __shared__ double sRed[8];
  __shared__ float sCoef[2 * 16];
  const int n = blockIdx.x;
  for (int c = 0; c < C; ++c) {
    const float* p = x + ((long)n * C + c) * HW;
    double v[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < HW; i += 256) { const double e = p[i]; v[0] += e; v[1] += e * e; }
    block_sum_d<2>(v, sRed);
    if (threadIdx.x == 0) {
      const double mean = v[0] / HW;
      double var = v[1] / HW - mean * mean;
      if (var < 0.0) var = 0.0;
      const float istd = (float)(1.0 / sqrt(var + (double)kInEps));
      sCoef[2 * c] = (float)mean; sCoef[2 * c + 1] = istd;
      stats[((long)n * C + c) * 2] = (float)mean; stats[((long)n * C + c) * 2 + 1] = istd;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < HW; i += 256) {
    float f[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = c < C ? (x[((long)n * C + c) * HW + i] - sCoef[2 * c]) * sCoef[2 * c + 1] : 0.f;
    T* o = xn + ((long)n * HW + i) * 16;
    constexpr int VE = Elem<T>::kVec;
#pragma unroll
    for (int v = 0; v < 16 / VE; ++v) reinterpret_cast<Vec16*>(o)[v] = Elem<T>::pack(f + v * VE);
  }
}

// d_xn [N][HW][16] of T (gradient w.r.t. the normalised input), x, stats -> d_x [N][C][HW] f32:  dx = istd (g - mean(g) - xhat mean(g xhat))
template <typename T>
__global__ __launch_bounds__(256) void inorm_planar_bwd_kernel(const T* __restrict__ g16, const float* __restrict__ x, const float* __restrict__ stats,
                                                                float* __restrict__ dx, int C, int HW) {
  __shared__ double sRed[8];
  const int n = blockIdx.x;
  for (int c = 0; c < C; ++c) {
    const float mean = stats[((long)n * C + c) * 2], istd = stats[((long)n * C + c) * 2 + 1];
    const float* p = x + ((long)n * C + c) * HW;
    double v[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float g = Elem<T>::load(g16 + ((long)n * HW + i) * 16 + c), xh = (p[i] - mean) * istd;
      v[0] += g; v[1] += (double)g * xh;
    }
    block_sum_d<2>(v, sRed);
    const float m1 = (float)(v[0] / HW), m2 = (float)(v[1] / HW);
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float g = Elem<T>::load(g16 + ((long)n * HW + i) * 16 + c), xh = (p[i] - mean) * istd;
      dx[((long)n * C + c) * HW + i] = istd * (g - m1 - xh * m2);
    }
    __syncthreads();
  }
}

// h [N][HW][C] of T -> a = relu?(instance_norm(h)) (same layout), stats [N][C][2].  Thread t owns the channel group t % cvecs (VE channels).
template <typename T>
__global__ __launch_bounds__(256) void inorm_nhwc_fwd_kernel(const T* __restrict__ h, T* __restrict__ a, float* __restrict__ stats, int C, int HW, int relu) {
  constexpr int VE = Elem<T>::kVec;
  extern __shared__ float smem[];           // [256][2][VE] partial sums, then [C][2] coefficients
  const int cvecs = C / VE, threads = 256 - 256 % cvecs, t = threadIdx.x, n = blockIdx.x;
  const long nvec = (long)HW * cvecs;
  const Vec16* src = reinterpret_cast<const Vec16*>(h) + (long)n * nvec;
  float acc[2][VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  if (t < threads)
    for (long v = t; v < nvec; v += threads) {
      float f[VE];
      Elem<T>::unpack(src[v], f);
#pragma unroll
      for (int j = 0; j < VE; ++j) { acc[0][j] += f[j]; acc[1][j] += f[j] * f[j]; }
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < VE; ++j) smem[(t * 2 + q) * VE + j] = t < threads ? acc[q][j] : 0.f;
  __syncthreads();
  float* coef = smem + 256 * 2 * VE;        // [C][2]
  for (int c = t; c < C; c += 256) {
    const int cg = c / VE, j = c - cg * VE;
    double s1 = 0.0, s2 = 0.0;
    for (int tt = cg; tt < threads; tt += cvecs) { s1 += smem[(tt * 2) * VE + j]; s2 += smem[(tt * 2 + 1) * VE + j]; }
    const double mean = s1 / HW;
    double var = s2 / HW - mean * mean;
    if (var < 0.0) var = 0.0;
    const float istd = (float)(1.0 / sqrt(var + (double)kInEps));
    coef[2 * c] = (float)mean; coef[2 * c + 1] = istd;
    stats[((long)n * C + c) * 2] = (float)mean; stats[((long)n * C + c) * 2 + 1] = istd;
  }
  __syncthreads();
  if (t < threads) {
    const int cg = t % cvecs;
    float mean[VE], istd[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { mean[j] = coef[2 * (cg * VE + j)]; istd[j] = coef[2 * (cg * VE + j) + 1]; }
    Vec16* dst = reinterpret_cast<Vec16*>(a) + (long)n * nvec;
    for (long v = t; v < nvec; v += threads) {
      float f[VE];
      Elem<T>::unpack(src[v], f);
#pragma unroll
      for (int j = 0; j < VE; ++j) { const float y = (f[j] - mean[j]) * istd[j]; f[j] = relu ? fmaxf(y, 0.f) : y; }
      dst[v] = Elem<T>::pack(f);
    }
  }
}

// g = gradient w.r.t. a = relu?(yhat), yhat = (h - mean) istd  ->  dh = istd (gm - mean(gm) - yhat mean(gm yhat)), gm = g [yhat > 0]
template <typename T>
__global__ __launch_bounds__(256) void inorm_nhwc_bwd_kernel(const T* __restrict__ g, const T* __restrict__ h, const float* __restrict__ stats,
                                                              T* __restrict__ dh, int C, int HW, int relu) {
  constexpr int VE = Elem<T>::kVec;
  extern __shared__ float smem[];
  const int cvecs = C / VE, threads = 256 - 256 % cvecs, t = threadIdx.x, n = blockIdx.x;
  const long nvec = (long)HW * cvecs;
  const Vec16* gs = reinterpret_cast<const Vec16*>(g) + (long)n * nvec;
  const Vec16* hs = reinterpret_cast<const Vec16*>(h) + (long)n * nvec;
  const int cg = t % cvecs;
  float mean[VE], istd[VE], acc[2][VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    mean[j] = stats[((long)n * C + cg * VE + j) * 2]; istd[j] = stats[((long)n * C + cg * VE + j) * 2 + 1];
    acc[0][j] = 0.f; acc[1][j] = 0.f;
  }
  if (t < threads)
    for (long v = t; v < nvec; v += threads) {
      float fg[VE], fh[VE];
      Elem<T>::unpack(gs[v], fg);
      Elem<T>::unpack(hs[v], fh);
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float y = (fh[j] - mean[j]) * istd[j];
        const float gm = (relu && !(y > 0.f)) ? 0.f : fg[j];
        acc[0][j] += gm; acc[1][j] += gm * y;
      }
    }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < VE; ++j) smem[(t * 2 + q) * VE + j] = t < threads ? acc[q][j] : 0.f;
  __syncthreads();
  float* coef = smem + 256 * 2 * VE;        // [C][2] = (m1, m2)
  for (int c = t; c < C; c += 256) {
    const int cgc = c / VE, j = c - cgc * VE;
    double s1 = 0.0, s2 = 0.0;
    for (int tt = cgc; tt < threads; tt += cvecs) { s1 += smem[(tt * 2) * VE + j]; s2 += smem[(tt * 2 + 1) * VE + j]; }
    coef[2 * c] = (float)(s1 / HW); coef[2 * c + 1] = (float)(s2 / HW);
  }
  __syncthreads();
  if (t < threads) {
    float m1[VE], m2[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) { m1[j] = coef[2 * (cg * VE + j)]; m2[j] = coef[2 * (cg * VE + j) + 1]; }
    Vec16* dst = reinterpret_cast<Vec16*>(dh) + (long)n * nvec;
    for (long v = t; v < nvec; v += threads) {
      float fg[VE], fh[VE];
      Elem<T>::unpack(gs[v], fg);
      Elem<T>::unpack(hs[v], fh);
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float y = (fh[j] - mean[j]) * istd[j];
        const float gm = (relu && !(y > 0.f)) ? 0.f : fg[j];
        fg[j] = istd[j] * (gm - m1[j] - y * m2[j]);
      }
      dst[v] = Elem<T>::pack(fg);
    }
  }
}

// o [N][HW][16] of T -> out [N][C][HW] f32 (channels < C), and the reverse with zero padding
template <typename T>
__global__ void nhwc16_to_planar_kernel(const T* __restrict__ o, float* __restrict__ out, long npix, int C, int HW) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long n = i / HW, p = i - n * HW;
    for (int c = 0; c < C; ++c) out[(n * C + c) * HW + p] = Elem<T>::load(o + i * 16 + c);
  }
}
template <typename T, typename TI>
__global__ void planar_to_nhwc16_kernel(const TI* __restrict__ in, T* __restrict__ o, long npix, int C, int HW) {
  constexpr int VE = Elem<T>::kVec;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long n = i / HW, p = i - n * HW;
    float f[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = c < C ? Elem<TI>::load(in + (n * C + c) * HW + p) : 0.f;
#pragma unroll
    for (int v = 0; v < 16 / VE; ++v) reinterpret_cast<Vec16*>(o + i * 16)[v] = Elem<T>::pack(f + v * VE);
  }
}

static int ew_blocks(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

int launch_inorm_planar_fwd(int dt, const float* x, void* xn16, float* stats, int N, int C, int HW, hipStream_t s) {
  if (C < 1 || C > 16) { set_error("inorm_planar: C=%d out of range (1..16)", C); return MMVAE_ERR_UNSUPPORTED; }
  if (dt == DT_F32) hipLaunchKernelGGL((inorm_planar_fwd_kernel<float>), dim3(N), dim3(256), 0, s, x, (float*)xn16, stats, C, HW);
  else hipLaunchKernelGGL((inorm_planar_fwd_kernel<bf16_t>), dim3(N), dim3(256), 0, s, x, (bf16_t*)xn16, stats, C, HW);
  return check_launch("inorm_planar_fwd");
}
int launch_inorm_planar_bwd(int dt, const void* g16, const float* x, const float* stats, float* dx, int N, int C, int HW, hipStream_t s) {
  if (C < 1 || C > 16) { set_error("inorm_planar: C=%d out of range (1..16)", C); return MMVAE_ERR_UNSUPPORTED; }
  if (dt == DT_F32) hipLaunchKernelGGL((inorm_planar_bwd_kernel<float>), dim3(N), dim3(256), 0, s, (const float*)g16, x, stats, dx, C, HW);
  else hipLaunchKernelGGL((inorm_planar_bwd_kernel<bf16_t>), dim3(N), dim3(256), 0, s, (const bf16_t*)g16, x, stats, dx, C, HW);
  return check_launch("inorm_planar_bwd");
}
int launch_inorm_nhwc_fwd(int dt, const void* h, void* a, float* stats, int N, int C, int HW, int relu, hipStream_t s) {
  const int VE = dt == DT_F32 ? 4 : 8;
  if (C % VE || C > 256) { set_error("inorm_nhwc: C=%d unsupported", C); return MMVAE_ERR_UNSUPPORTED; }
  const size_t sm = (size_t)(256 * 2 * VE + 2 * C) * sizeof(float);
  if (dt == DT_F32) hipLaunchKernelGGL((inorm_nhwc_fwd_kernel<float>), dim3(N), dim3(256), sm, s, (const float*)h, (float*)a, stats, C, HW, relu);
  else hipLaunchKernelGGL((inorm_nhwc_fwd_kernel<bf16_t>), dim3(N), dim3(256), sm, s, (const bf16_t*)h, (bf16_t*)a, stats, C, HW, relu);
  return check_launch("inorm_nhwc_fwd");
}
int launch_inorm_nhwc_bwd(int dt, const void* g, const void* h, const float* stats, void* dh, int N, int C, int HW, int relu, hipStream_t s) {
  const int VE = dt == DT_F32 ? 4 : 8;
  if (C % VE || C > 256) { set_error("inorm_nhwc: C=%d unsupported", C); return MMVAE_ERR_UNSUPPORTED; }
  const size_t sm = (size_t)(256 * 2 * VE + 2 * C) * sizeof(float);
  if (dt == DT_F32) hipLaunchKernelGGL((inorm_nhwc_bwd_kernel<float>), dim3(N), dim3(256), sm, s, (const float*)g, (const float*)h, stats, (float*)dh, C, HW, relu);
  else hipLaunchKernelGGL((inorm_nhwc_bwd_kernel<bf16_t>), dim3(N), dim3(256), sm, s, (const bf16_t*)g, (const bf16_t*)h, stats, (bf16_t*)dh, C, HW, relu);
  return check_launch("inorm_nhwc_bwd");
}
int launch_nhwc16_to_planar(int dt, const void* o16, float* out, int N, int C, int HW, hipStream_t s) {
  const long npix = (long)N * HW;
  if (dt == DT_F32) hipLaunchKernelGGL((nhwc16_to_planar_kernel<float>), dim3(ew_blocks(npix)), dim3(256), 0, s, (const float*)o16, out, npix, C, HW);
  else hipLaunchKernelGGL((nhwc16_to_planar_kernel<bf16_t>), dim3(ew_blocks(npix)), dim3(256), 0, s, (const bf16_t*)o16, out, npix, C, HW);
  return check_launch("nhwc16_to_planar");
}
int launch_planar_to_nhwc16(int dt, const float* in, void* o16, int N, int C, int HW, hipStream_t s) {
  const long npix = (long)N * HW;
  if (dt == DT_F32) hipLaunchKernelGGL((planar_to_nhwc16_kernel<float, float>), dim3(ew_blocks(npix)), dim3(256), 0, s, in, (float*)o16, npix, C, HW);
  else hipLaunchKernelGGL((planar_to_nhwc16_kernel<bf16_t, float>), dim3(ew_blocks(npix)), dim3(256), 0, s, in, (bf16_t*)o16, npix, C, HW);
  return check_launch("planar_to_nhwc16");
}
// the same from a planar tensor of the storage type (the encoder's staged image, in_channels > 1)
int launch_planar_to_nhwc16(int dt, const void* in, int in_dt, void* o16, int N, int C, int HW, hipStream_t s) {
  if (in_dt != dt) { set_error("planar_to_nhwc16: input and output types differ"); return MMVAE_ERR_UNSUPPORTED; }
  if (dt == DT_F32) return launch_planar_to_nhwc16(dt, static_cast<const float*>(in), o16, N, C, HW, s);
  const long npix = (long)N * HW;
  hipLaunchKernelGGL((planar_to_nhwc16_kernel<bf16_t, bf16_t>), dim3(ew_blocks(npix)), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)o16, npix, C, HW);
  return check_launch("planar_to_nhwc16");
}

}  // namespace mmvae
