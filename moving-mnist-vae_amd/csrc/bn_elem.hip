// Train-mode BatchNorm pieces and the residual joins, as bandwidth kernels (16-byte NHWC vectors).
//
// Forward BN-apply(+ReLU) is normally fused into the consumer conv's load prologue (conv_gemm.hip);
// what is here: per-channel statistics, the finalize step (scale/shift, running stats), the two-input
// residual join, and the backward reductions / dy materialisation  dy = A*g + B*y + C  per channel.
#include "kernels.hpp"
#include "tile_common.hpp"
#include <cstdlib>

namespace mmvae {

// Grid cap of the grid-stride elementwise kernels: 768 = three blocks per CU.  Per kernel it makes no difference (bn_bwd_apply 62 us,
// affine_join 33 us at 512 .. 2048 blocks), the STEP gains 0.06 ms over 2048 (7.91 vs 7.98 ms, twice on one box; 512: 7.92, 640: 7.96,
// 896: 8.00): fewer resident blocks leave the weight-gradient kernels of the side stream more of every CU.
constexpr int kElemMaxBlocks = 768;

__host__ __device__ inline int block_threads_for(int cvecs) {
  // a block size that is a multiple of cvecs, so a thread's channel group never changes in a grid-stride loop
  int b = 256 - (256 % cvecs);
  return b < cvecs ? cvecs : b;
}

static int elem_blocks(long nvec, int threads) {
  constexpr int cap = kElemMaxBlocks;
  long b = (nvec + (long)threads * 4 - 1) / ((long)threads * 4);
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// Per-thread channel-group partials -> per-block per-channel sums.  vals[NV][VE] per thread; threads with equal
// (threadIdx.x % cvecs) share channels.  Writes out[v*C + cg*VE + j] for v<NV.
template <int NV, int VE>
__device__ __forceinline__ void block_channel_reduce(float (&vals)[NV][VE], int cvecs, int C, float* smem, float* out) {
  // smem: blockDim.x * NV * VE floats
  const int t = threadIdx.x;
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int j = 0; j < VE; ++j) smem[(t * NV + v) * VE + j] = vals[v][j];
  __syncthreads();
  for (int o = t; o < cvecs * NV * VE; o += blockDim.x) {
    const int cg = o / (NV * VE), rem = o - cg * (NV * VE);
    float s = 0.f;
    for (int tt = cg; tt < (int)blockDim.x; tt += cvecs) s += smem[tt * NV * VE + rem];
    const int v = rem / VE, j = rem - v * VE;
    out[(long)v * C + cg * VE + j] = s;
  }
}

// ---------------------------------------------------------------- channel statistics (NHWC)
template <typename T>
__global__ void chan_stats_nhwc_kernel(const T* __restrict__ y, long nvec, int C, float* __restrict__ partials) {
  constexpr int VE = Elem<T>::kVec;
  extern __shared__ float smem[];
  const int cvecs = C / VE;
  float acc[2][VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  const long stride = (long)gridDim.x * blockDim.x;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    float f[VE];
    Elem<T>::unpack(reinterpret_cast<const Vec16*>(y)[v], f);
#pragma unroll
    for (int j = 0; j < VE; ++j) { acc[0][j] += f[j]; acc[1][j] += f[j] * f[j]; }
  }
  block_channel_reduce<2, VE>(acc, cvecs, C, smem, partials + (long)blockIdx.x * 2 * C);
}

int chan_stats_parts(long, int) { return 1024; }   // upper bound on partial rows of every reduction kernel

int launch_chan_stats_nhwc(int dt, const void* y, long npix, int C, float* partials, hipStream_t s) {
  const int VE = dt == DT_F32 ? 4 : 8;
  if (C % VE) { set_error("chan_stats: C=%d not a multiple of %d", C, VE); return MMVAE_ERR_UNSUPPORTED; }
  const int cvecs = C / VE, threads = block_threads_for(cvecs);
  const long nvec = npix * cvecs;
  int blocks = elem_blocks(nvec, threads);
  if (blocks > 1024) blocks = 1024;
  const size_t sm = (size_t)threads * 2 * VE * sizeof(float);
  if (dt == DT_F32) hipLaunchKernelGGL((chan_stats_nhwc_kernel<float>), dim3(blocks), dim3(threads), sm, s, (const float*)y, nvec, C, partials);
  else hipLaunchKernelGGL((chan_stats_nhwc_kernel<bf16_t>), dim3(blocks), dim3(threads), sm, s, (const bf16_t*)y, nvec, C, partials);
  int rc = check_launch("chan_stats_nhwc");
  return rc ? rc : blocks;
}

// ---------------------------------------------------------------- channel statistics (NCHW f32)
// planes = N*C; block handles planes blockIdx.x, +gridDim.x, ...; partials [gridDim.x][NR][C]
template <int NR>
__global__ __launch_bounds__(256) void plane_reduce_nchw_kernel(const float* __restrict__ a, const float* __restrict__ b, int N, int C, int HW,
                                                                 float* __restrict__ partials) {
  // NR=2, b==null: (sum a, sum a^2);  NR=2, b!=null: (sum a, sum a*b).  No barrier per plane: every wave reduces its
  // share of a plane into its OWN per-channel LDS accumulators (single writer, fixed order -> bit-reproducible sums);
  // the four waves are combined once at the end.
  extern __shared__ float sm[];
  float* sAcc = sm;             // [4 waves][NR][C]
  const int wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * NR * C; i += blockDim.x) sAcc[i] = 0.f;
  __syncthreads();
  const int planes = N * C;
  for (int p = blockIdx.x; p < planes; p += gridDim.x) {
    const int c = p % C;
    const float* pa = a + (long)p * HW;
    const float* pb = b ? b + (long)p * HW : nullptr;
    float v0 = 0.f, v1 = 0.f;
    if ((HW & 3) == 0) {
      for (int i = threadIdx.x * 4; i < HW; i += 4 * blockDim.x) {
        const float4 x = *reinterpret_cast<const float4*>(pa + i);
        float4 y = x;
        if (pb) y = *reinterpret_cast<const float4*>(pb + i);
        v0 += (x.x + x.y) + (x.z + x.w);
        v1 += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float x = pa[i];
        v0 += x;
        v1 += pb ? x * pb[i] : x * x;
      }
    }
    v0 = wave_sum(v0); v1 = wave_sum(v1);
    if ((threadIdx.x & 63) == 0) { sAcc[wid * NR * C + c] += v0; sAcc[wid * NR * C + C + c] += v1; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NR * C; i += blockDim.x)
    partials[(long)blockIdx.x * NR * C + i] = (sAcc[i] + sAcc[NR * C + i]) + (sAcc[2 * NR * C + i] + sAcc[3 * NR * C + i]);
}

static int nchw_parts(int N, int C) { int p = N * C; return p > 1024 ? 1024 : (p < 1 ? 1 : p); }


int launch_bn_bwd_reduce_nchw(const float* dout, const float* y, int N, int C, int HW, float* partials, hipStream_t s) {
  const int blocks = nchw_parts(N, C);
  const size_t sm = (size_t)4 * 2 * C * sizeof(float);
  hipLaunchKernelGGL((plane_reduce_nchw_kernel<2>), dim3(blocks), dim3(256), sm, s, dout, y, N, C, HW, partials);
  int rc = check_launch("bn_bwd_reduce_nchw");
  return rc ? rc : blocks;
}

// ---------------------------------------------------------------- SyncBN: partial rows -> one row (the all-reduce operand)
// partials [nparts][rows][C] -> out [rows * C]: the finalize kernels take the summed row as a single partial row afterwards.
__global__ __launch_bounds__(256) void partial_rowsum_kernel(const float* __restrict__ partials, int nparts, int stride, float* __restrict__ out) {
  __shared__ double sRed[4];
  const int col = blockIdx.x;
  double s = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) s += (double)partials[(long)p * stride + col];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[col] = (float)((sRed[0] + sRed[1]) + (sRed[2] + sRed[3]));
}
int launch_partial_rowsum(const float* partials, int nparts, int width, float* out, hipStream_t s, int row_stride) {
  hipLaunchKernelGGL(partial_rowsum_kernel, dim3(width), dim3(256), 0, s, partials, nparts, row_stride > 0 ? row_stride : width, out);
  return check_launch("partial_rowsum");
}

// ---------------------------------------------------------------- finalize (training)
// Two strided column sums over per-block partial rows (finalize kernels: one block per channel).  The loads of U rows are issued before the
// first add, so a block pays nparts / (U * blockDim) memory latencies instead of nparts / blockDim -- these launches sit between the big
// kernels of the critical stream.  Fixed order per thread -> bit-reproducible.
template <int U>
__device__ __forceinline__ void strided_pair_sum(const float* __restrict__ pa, const float* __restrict__ pb, long stride, int nparts, double& s0, double& s1) {
  const int bd = blockDim.x;
  int p = threadIdx.x;
  for (; p + (U - 1) * bd < nparts; p += U * bd) {
    float va[U], vb[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { va[k] = pa[(long)(p + k * bd) * stride]; vb[k] = pb[(long)(p + k * bd) * stride]; }
#pragma unroll
    for (int k = 0; k < U; ++k) { s0 += (double)va[k]; s1 += (double)vb[k]; }
  }
  for (; p < nparts; p += bd) { s0 += (double)pa[(long)p * stride]; s1 += (double)pb[(long)p * stride]; }
}

__global__ void bn_finalize_kernel(BnFinalizeArgs a) {
  __shared__ double sRed[2 * 4];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  strided_pair_sum<8>(a.partials + c, a.partials + a.C + c, 2L * a.C, a.nparts, s1, s2);
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sRed[wid] = s1; sRed[4 + wid] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    s1 = 0.0; s2 = 0.0;
    for (int w = 0; w < nw; ++w) { s1 += sRed[w]; s2 += sRed[4 + w]; }
    const double n = a.count;
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double isc = (double)a.in_scale;
    const double istd = 1.0 / sqrt(var + (double)a.eps * isc * isc);
    const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
    a.mean[c] = (float)mean;
    a.istd[c] = (float)istd;
    const float sc = (float)((double)g * istd);
    a.scale[c] = sc;
    a.shift[c] = (float)((double)b - mean * (double)g * istd);
    if (a.running_mean) {
      const double unbiased = (n > 1.0 ? var * n / (n - 1.0) : var) / (isc * isc);
      a.running_mean[c] = (float)((1.0 - (double)a.momentum) * (double)a.running_mean[c] + (double)a.momentum * mean / isc);
      a.running_var[c] = (float)((1.0 - (double)a.momentum) * (double)a.running_var[c] + (double)a.momentum * unbiased);
    }
    if (c == 0 && a.nbt) *a.nbt += 1;
  }
}

int launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(a.C), dim3(256), 0, s, a);
  return check_launch("bn_finalize");
}

// Inference (model.eval(), reference model.py:353-362): every BatchNorm of an entry point folded into its per-channel (scale, shift) pair from the
// running statistics in ONE launch -- the consumers' prologues apply them, no statistics, no per-layer finalize.
__global__ void bn_fold_eval_kernel(BnFoldTable t, const float* __restrict__ params, const float* __restrict__ bnbuf, float* __restrict__ bnws, float eps) {
  const BnFoldEntry e = t.e[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
    const float istd = 1.0f / sqrtf(bnbuf[e.rv_off + c] + eps);
    const float sc = params[e.g_off + c] * istd;
    bnws[e.scale_off + c] = sc / e.in_scale;              // the conv output in memory is in_scale * y (fp8 weight scale)
    bnws[e.shift_off + c] = params[e.b_off + c] - bnbuf[e.rm_off + c] * sc;
  }
}
int launch_bn_fold_eval(const BnFoldEntry* entries, int n, const float* params, const float* bnbuf, float* bnws, float eps, hipStream_t s) {
  for (int i = 0; i < n; i += kBnFoldMax) {
    BnFoldTable t;
    const int m = n - i < kBnFoldMax ? n - i : kBnFoldMax;
    for (int j = 0; j < m; ++j) t.e[j] = entries[i + j];
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3(m), dim3(256), 0, s, t, params, bnbuf, bnws, eps);
  }
  return check_launch("bn_fold_eval");
}

// ---------------------------------------------------------------- forward elementwise
template <typename T, bool TWO>
__global__ void affine_join_kernel(const T* __restrict__ a, const float* __restrict__ sa, const float* __restrict__ ba,
                                   const T* __restrict__ b, const float* __restrict__ sb, const float* __restrict__ bb,
                                   int relu, T* __restrict__ out, long nvec, int C) {
  constexpr int VE = Elem<T>::kVec;
  const int cvecs = C / VE;
  const long stride = (long)gridDim.x * blockDim.x;
  const int cg = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) % cvecs);   // constant: blockDim % cvecs == 0
  float s0[VE], b0[VE], s1[VE], b1[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    s0[j] = sa[cg * VE + j]; b0[j] = ba[cg * VE + j];
    s1[j] = TWO ? sb[cg * VE + j] : 0.f; b1[j] = TWO ? bb[cg * VE + j] : 0.f;
  }
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    float fa[VE], fb[VE];
    Elem<T>::unpack(reinterpret_cast<const Vec16*>(a)[v], fa);
    if (TWO) Elem<T>::unpack(reinterpret_cast<const Vec16*>(b)[v], fb);
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      float x = fa[j] * s0[j] + b0[j];
      if (TWO) x += fb[j] * s1[j] + b1[j];
      fa[j] = relu ? fmaxf(x, 0.f) : x;
    }
    reinterpret_cast<Vec16*>(out)[v] = Elem<T>::pack(fa);
  }
}

template <typename T>
static int launch_affine_join_t(const void* a, const float* sa, const float* ba, const void* b, const float* sb, const float* bb,
                                int relu, void* out, long npix, int C, hipStream_t s) {
  constexpr int VE = Elem<T>::kVec;
  if (C % VE) { set_error("affine/join: C=%d not a multiple of %d", C, VE); return MMVAE_ERR_UNSUPPORTED; }
  const int cvecs = C / VE, threads = block_threads_for(cvecs);
  const long nvec = npix * cvecs;
  const int blocks = elem_blocks(nvec, threads);
  if (b) hipLaunchKernelGGL((affine_join_kernel<T, true>), dim3(blocks), dim3(threads), 0, s, (const T*)a, sa, ba, (const T*)b, sb, bb, relu, (T*)out, nvec, C);
  else hipLaunchKernelGGL((affine_join_kernel<T, false>), dim3(blocks), dim3(threads), 0, s, (const T*)a, sa, ba, (const T*)nullptr, sb, bb, relu, (T*)out, nvec, C);
  note_launch_bytes((double)npix * C * sizeof(T) * (b ? 3 : 2));
  return check_launch("affine_join");
}

int launch_join_fwd(int dt, const void* a, const float* sa, const float* ba, const void* b, const float* sb, const float* bb,
                    void* out, long npix, int C, hipStream_t s) {
  return dt == DT_F32 ? launch_affine_join_t<float>(a, sa, ba, b, sb, bb, 1, out, npix, C, s)
                      : launch_affine_join_t<bf16_t>(a, sa, ba, b, sb, bb, 1, out, npix, C, s);
}
int launch_affine_act(int dt, const void* a, const float* sa, const float* ba, int relu, void* out, long npix, int C, hipStream_t s) {
  return dt == DT_F32 ? launch_affine_join_t<float>(a, sa, ba, nullptr, nullptr, nullptr, relu, out, npix, C, s)
                      : launch_affine_join_t<bf16_t>(a, sa, ba, nullptr, nullptr, nullptr, relu, out, npix, C, s);
}

// ---------------------------------------------------------------- backward reductions
// MODE 0: mask from out>0 ; MODE 1: mask from y0*ms+mb>0 ; MODE 2: no mask ;
// MODE 3: mask from (y0*ms+mb) + (y1*ms1+mb1) > 0  (the residual join recomputed -> `out` is not read at all)
template <typename T, int NY, int MODE>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ out, const float* __restrict__ ms,
                                     const float* __restrict__ mb, const float* __restrict__ ms1, const float* __restrict__ mb1,
                                     const T* __restrict__ y0, const T* __restrict__ y1,
                                     long nvec, int C, float* __restrict__ partials) {
  constexpr int VE = Elem<T>::kVec;
  extern __shared__ float smem[];
  const int cvecs = C / VE;
  const int cg = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) % cvecs);
  float msc[VE], msh[VE], msc1[VE], msh1[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    msc[j] = (MODE == 1 || MODE == 3) ? ms[cg * VE + j] : 0.f; msh[j] = (MODE == 1 || MODE == 3) ? mb[cg * VE + j] : 0.f;
    msc1[j] = MODE == 3 ? ms1[cg * VE + j] : 0.f; msh1[j] = MODE == 3 ? mb1[cg * VE + j] : 0.f;
  }
  float acc[1 + NY][VE];
#pragma unroll
  for (int q = 0; q < 1 + NY; ++q)
#pragma unroll
    for (int j = 0; j < VE; ++j) acc[q][j] = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    float g[VE], o[VE], a0[VE], a1[VE];
    Elem<T>::unpack(reinterpret_cast<const Vec16*>(dout)[v], g);
    Elem<T>::unpack(reinterpret_cast<const Vec16*>(y0)[v], a0);
    if (MODE == 0) Elem<T>::unpack(reinterpret_cast<const Vec16*>(out)[v], o);
    if (NY == 2) Elem<T>::unpack(reinterpret_cast<const Vec16*>(y1)[v], a1);
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      bool on = true;
      if (MODE == 0) on = o[j] > 0.f;
      if (MODE == 1) on = (a0[j] * msc[j] + msh[j]) > 0.f;
      if (MODE == 3) { float x = a0[j] * msc[j] + msh[j]; x += a1[j] * msc1[j] + msh1[j]; on = x > 0.f; }
      const float gg = on ? g[j] : 0.f;
      acc[0][j] += gg;
      acc[1][j] += gg * a0[j];
      if (NY == 2) acc[2][j] += gg * a1[j];
    }
  }
  block_channel_reduce<1 + NY, VE>(acc, cvecs, C, smem, partials + (long)blockIdx.x * (1 + NY) * C);
}

template <typename T>
static int launch_bn_bwd_reduce_t(const void* dout, const void* out, const float* ms, const float* mb, const float* ms1,
                                  const float* mb1, const void* y0, const void* y1, long npix, int C, float* partials, hipStream_t s) {
  constexpr int VE = Elem<T>::kVec;
  if (C % VE) { set_error("bn_bwd_reduce: C=%d not a multiple of %d", C, VE); return MMVAE_ERR_UNSUPPORTED; }
  const int cvecs = C / VE, threads = block_threads_for(cvecs);
  const long nvec = npix * cvecs;
  int blocks = elem_blocks(nvec, threads);
  if (blocks > 1024) blocks = 1024;
  const int ny = y1 ? 2 : 1;
  const size_t sm = (size_t)threads * (1 + ny) * VE * sizeof(float);
  const int mode = out ? 0 : (ms ? ((ms1 && ny == 2) ? 3 : 1) : 2);
#define MMVAE_LAUNCH(NY, MODE) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, NY, MODE>), dim3(blocks), dim3(threads), sm, s, \
    (const T*)dout, (const T*)out, ms, mb, ms1, mb1, (const T*)y0, (const T*)y1, nvec, C, partials)
  if (ny == 2) { if (mode == 0) MMVAE_LAUNCH(2, 0); else if (mode == 1) MMVAE_LAUNCH(2, 1); else if (mode == 3) MMVAE_LAUNCH(2, 3); else MMVAE_LAUNCH(2, 2); }
  else { if (mode == 0) MMVAE_LAUNCH(1, 0); else if (mode == 1) MMVAE_LAUNCH(1, 1); else MMVAE_LAUNCH(1, 2); }
#undef MMVAE_LAUNCH
  note_launch_bytes((double)npix * C * sizeof(T) * (1 + ny + (out ? 1 : 0)));
  int rc = check_launch("bn_bwd_reduce");
  return rc ? rc : blocks;
}
int launch_bn_bwd_reduce(int dt, const void* dout, const void* out, const float* msk_scale, const float* msk_shift,
                         const void* y0, const void* y1, long npix, int C, float* partials, hipStream_t s,
                         const float* msk_scale1, const float* msk_shift1) {
  return dt == DT_F32 ? launch_bn_bwd_reduce_t<float>(dout, out, msk_scale, msk_shift, msk_scale1, msk_shift1, y0, y1, npix, C, partials, s)
                      : launch_bn_bwd_reduce_t<bf16_t>(dout, out, msk_scale, msk_shift, msk_scale1, msk_shift1, y0, y1, npix, C, partials, s);
}

// blockIdx.y selects the BatchNorm: the two branches of a residual join (same partials, which = 0 / 1) finalise in one launch
__global__ void bn_bwd_finalize_kernel(BnBwdFinalizeArgs a0, BnBwdFinalizeArgs a1) {
  const BnBwdFinalizeArgs& a = blockIdx.y ? a1 : a0;
  __shared__ double sRed[2 * 4];
  const int c = blockIdx.x;
  const int rows = 1 + a.ny;
  double s0 = 0.0, s1 = 0.0;
  strided_pair_sum<8>(a.partials + c, a.partials + (long)(1 + a.which) * a.C + c, (long)rows * a.C, a.nparts, s0, s1);
  s0 = wave_sum_d(s0); s1 = wave_sum_d(s1);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sRed[wid] = s0; sRed[4 + wid] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    s0 = 0.0; s1 = 0.0;
    for (int w = 0; w < nw; ++w) { s0 += sRed[w]; s1 += sRed[4 + w]; }
    const double mean = a.mean[c], istd = a.istd[c], g = a.gamma ? a.gamma[c] : 1.0;
    const double sgy = istd * (s1 - mean * s0);       // sum g * yhat
    if (a.dgamma) a.dgamma[c] += (float)sgy;
    if (a.dbeta) a.dbeta[c] += (float)s0;
    const double m1 = s0 / a.count, m2 = sgy / a.count;
    const double A = g * istd;
    const float Af = (float)A, Bf = (float)(-A * m2 * istd), Cf = (float)(-A * m1 + A * m2 * istd * mean);
    a.coefA[c] = Af;
    a.coefB[c] = Bf;
    a.coefC[c] = Cf;
    if (a.dbias_conv) a.dbias_conv[c] += (float)(((double)Af * s0 + (double)Bf * (mean * a.count) + (double)Cf * a.count) * (double)a.dbias_scale);
  }
}
int launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(a.C), dim3(256), 0, s, a, a);
  return check_launch("bn_bwd_finalize");
}
int launch_bn_bwd_finalize2(const BnBwdFinalizeArgs& a0, const BnBwdFinalizeArgs& a1, hipStream_t s) {
  if (a0.C != a1.C) { set_error("bn_bwd_finalize2: channel counts differ"); return MMVAE_ERR_ARG; }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(a0.C, 2), dim3(256), 0, s, a0, a1);
  return check_launch("bn_bwd_finalize2");
}

template <typename T, int NY, int MODE>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out, const float* __restrict__ ms,
                                    const float* __restrict__ mb, const float* __restrict__ ms1, const float* __restrict__ mb1,
                                    const T* __restrict__ y0, const float* __restrict__ A0,
                                    const float* __restrict__ B0, const float* __restrict__ C0, T* __restrict__ dy0,
                                    const T* __restrict__ y1, const float* __restrict__ A1, const float* __restrict__ B1,
                                    const float* __restrict__ C1, T* __restrict__ dy1, long nvec, int C) {
  constexpr int VE = Elem<T>::kVec;
  const int cvecs = C / VE;
  const int cg = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) % cvecs);
  float msc[VE], msh[VE], msc1[VE], msh1[VE], a0[VE], b0[VE], c0[VE], a1[VE], b1[VE], c1[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    const int ch = cg * VE + j;
    msc[j] = (MODE == 1 || MODE == 3) ? ms[ch] : 0.f; msh[j] = (MODE == 1 || MODE == 3) ? mb[ch] : 0.f;
    msc1[j] = MODE == 3 ? ms1[ch] : 0.f; msh1[j] = MODE == 3 ? mb1[ch] : 0.f;
    a0[j] = A0[ch]; b0[j] = B0[ch]; c0[j] = C0[ch];
    a1[j] = NY == 2 ? A1[ch] : 0.f; b1[j] = NY == 2 ? B1[ch] : 0.f; c1[j] = NY == 2 ? C1[ch] : 0.f;
  }
  const long stride = (long)gridDim.x * blockDim.x;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    float g[VE], o[VE], f0[VE], f1[VE], r0[VE], r1[VE];
    Elem<T>::unpack(reinterpret_cast<const Vec16*>(dout)[v], g);
    Elem<T>::unpack(reinterpret_cast<const Vec16*>(y0)[v], f0);
    if (MODE == 0) Elem<T>::unpack(reinterpret_cast<const Vec16*>(out)[v], o);
    if (NY == 2) Elem<T>::unpack(reinterpret_cast<const Vec16*>(y1)[v], f1);
#pragma unroll
    for (int j = 0; j < VE; ++j) {
      bool on = true;
      if (MODE == 0) on = o[j] > 0.f;
      if (MODE == 1) on = (f0[j] * msc[j] + msh[j]) > 0.f;
      if (MODE == 3) { float x = f0[j] * msc[j] + msh[j]; x += f1[j] * msc1[j] + msh1[j]; on = x > 0.f; }
      const float gg = on ? g[j] : 0.f;
      r0[j] = a0[j] * gg + b0[j] * f0[j] + c0[j];
      if (NY == 2) r1[j] = a1[j] * gg + b1[j] * f1[j] + c1[j];
    }
    reinterpret_cast<Vec16*>(dy0)[v] = Elem<T>::pack(r0);
    if (NY == 2) reinterpret_cast<Vec16*>(dy1)[v] = Elem<T>::pack(r1);
  }
}

template <typename T>
static int launch_bn_bwd_apply_t(const void* dout, const void* out, const float* ms, const float* mb, const float* ms1, const float* mb1,
                                 const void* y0,
                                 const float* A0, const float* B0, const float* C0, void* dy0, const void* y1, const float* A1,
                                 const float* B1, const float* C1, void* dy1, long npix, int C, hipStream_t s) {
  constexpr int VE = Elem<T>::kVec;
  if (C % VE) { set_error("bn_bwd_apply: C=%d not a multiple of %d", C, VE); return MMVAE_ERR_UNSUPPORTED; }
  const int cvecs = C / VE, threads = block_threads_for(cvecs);
  const long nvec = npix * cvecs;
  const int blocks = elem_blocks(nvec, threads);
  const int ny = y1 ? 2 : 1;
  const int mode = out ? 0 : (ms ? ((ms1 && ny == 2) ? 3 : 1) : 2);
#define MMVAE_LAUNCH(NY, MODE) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, NY, MODE>), dim3(blocks), dim3(threads), 0, s, \
    (const T*)dout, (const T*)out, ms, mb, ms1, mb1, (const T*)y0, A0, B0, C0, (T*)dy0, (const T*)y1, A1, B1, C1, (T*)dy1, nvec, C)
  if (ny == 2) { if (mode == 0) MMVAE_LAUNCH(2, 0); else if (mode == 1) MMVAE_LAUNCH(2, 1); else if (mode == 3) MMVAE_LAUNCH(2, 3); else MMVAE_LAUNCH(2, 2); }
  else { if (mode == 0) MMVAE_LAUNCH(1, 0); else if (mode == 1) MMVAE_LAUNCH(1, 1); else MMVAE_LAUNCH(1, 2); }
#undef MMVAE_LAUNCH
  note_launch_bytes((double)npix * C * sizeof(T) * (1 + 2 * ny + (out ? 1 : 0)));
  return check_launch("bn_bwd_apply");
}
int launch_bn_bwd_apply(int dt, const void* dout, const void* out, const float* msk_scale, const float* msk_shift,
                        const void* y0, const float* A0, const float* B0, const float* C0, void* dy0,
                        const void* y1, const float* A1, const float* B1, const float* C1, void* dy1,
                        long npix, int C, hipStream_t s, const float* msk_scale1, const float* msk_shift1) {
  return dt == DT_F32 ? launch_bn_bwd_apply_t<float>(dout, out, msk_scale, msk_shift, msk_scale1, msk_shift1, y0, A0, B0, C0, dy0, y1, A1, B1, C1, dy1, npix, C, s)
                      : launch_bn_bwd_apply_t<bf16_t>(dout, out, msk_scale, msk_shift, msk_scale1, msk_shift1, y0, A0, B0, C0, dy0, y1, A1, B1, C1, dy1, npix, C, s);
}

// ---------------------------------------------------------------- last up-block: join backward fused with the tail dgrad
// The gradient entering the last up-block is a 3x3 convolution of the reconstruction gradient d_raw (out_ch <= 8 planes,
// f32 NCHW) with the tail weights:  g[n,h,w,ci] = sum_{oc,kh,kw} d_raw[n,oc,h+1-kh,w+1-kw] * w[oc][ci][kh][kw].
// Materialising it costs a 671 MB write and two 671 MB reads (reduce + apply) at N = 5120; recomputing it from the 8x
// smaller d_raw inside the two BatchNorm-backward passes costs 72 FMAs per 16-byte vector and almost no HBM traffic.
// C = 16 channels (the tail conv's input).  Same mask / coefficient arithmetic as bn_bwd_reduce/apply MODE 3, NY = 2.
//
// A block walks tiles of 256 consecutive 16-byte vectors = PT = 256/cvecs consecutive pixels = R = PT/W whole image rows
// (the host checks W is a power of two <= PT and H*W % PT == 0, so a tile never straddles images).  The R+2 rows of d_raw a
// tile needs go through a double-buffered LDS tile with zero halo columns: one barrier per tile, no per-thread divisions.
// OC == 1 (Moving MNIST) keeps the thread's 9 x VE weights in registers (bf16 pairs when T is bf16: the same rounding the
// MFMA dgrad applies); OC > 1 reads them from LDS.
struct TailG { const float* d_raw; const float* w; int OC, H, W, wshift; int ntiles; };

template <typename T> struct TailW;
template <> struct TailW<float> {
  float v[9][4];
  // the LDS form of the same table: word j of (tap t, channel group c0/4)
  __device__ static __forceinline__ uint32_t word(const float* w, int c0, int t, int j) { return __float_as_uint(w[(c0 + j) * 9 + t]); }
  __device__ static __forceinline__ void fma_words(const uint4& q, float d, float (&g)[4]) {
    g[0] += d * __uint_as_float(q.x); g[1] += d * __uint_as_float(q.y); g[2] += d * __uint_as_float(q.z); g[3] += d * __uint_as_float(q.w);
  }
  __device__ __forceinline__ void load(const float* w, int c0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[t][j] = w[(c0 + j) * 9 + t];
  }
  __device__ __forceinline__ void fma(int t, float d, float (&g)[4]) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] += d * v[t][j];
  }
  __device__ __forceinline__ float dot(int t, const float (&x)[4]) const {
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) r += x[j] * v[t][j];
    return r;
  }
};
template <> struct TailW<bf16_t> {
  uint32_t v[9][4];
  __device__ static __forceinline__ uint32_t word(const float* w, int c0, int t, int j) {
    return pack2_bf16(w[(c0 + 2 * j) * 9 + t], w[(c0 + 2 * j + 1) * 9 + t]);
  }
  __device__ static __forceinline__ void fma_words(const uint4& q, float d, float (&g)[8]) {
    const uint32_t pw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[2 * j] += d * __uint_as_float(pw[j] << 16);
      g[2 * j + 1] += d * __uint_as_float(pw[j] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void load(const float* w, int c0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[t][j] = pack2_bf16(w[(c0 + 2 * j) * 9 + t], w[(c0 + 2 * j + 1) * 9 + t]);
  }
  __device__ __forceinline__ void fma(int t, float d, float (&g)[8]) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t pw = v[t][j];
      asm volatile("" : "+v"(pw));        // keep the pair packed across iterations: a hoisted unpack doubles the registers
      g[2 * j] += d * __uint_as_float(pw << 16);
      g[2 * j + 1] += d * __uint_as_float(pw & 0xffff0000u);
    }
  }
  __device__ __forceinline__ float dot(int t, const float (&x)[8]) const {
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t pw = v[t][j];
      asm volatile("" : "+v"(pw));
      r += x[2 * j] * __uint_as_float(pw << 16);
      r += x[2 * j + 1] * __uint_as_float(pw & 0xffff0000u);
    }
    return r;
  }
};

// WG (reduce pass, OC == 1 only): also accumulates the tail conv's weight gradient dW[ci][tap] = sum x[ci] * d_raw[tap] with the
// joined activation x = max(bn(y0) + bn(y1), 0) it has in registers anyway -> wpartials[block][144], tail_wgrad_finalize_kernel.
template <typename T, bool OC1, bool APPLY, bool WG = false>
__global__ __launch_bounds__(256, (OC1 && !WG) ? 3 : 2) void tail_join_bwd_kernel(TailG tg, const float* __restrict__ ms, const float* __restrict__ mb,
                                                            const float* __restrict__ ms1, const float* __restrict__ mb1,
                                                            const T* __restrict__ y0, const float* __restrict__ A0, const float* __restrict__ B0,
                                                            const float* __restrict__ C0, T* __restrict__ dy0, const T* __restrict__ y1,
                                                            const float* __restrict__ A1, const float* __restrict__ B1,
                                                            const float* __restrict__ C1, T* __restrict__ dy1, float* __restrict__ partials,
                                                            float* __restrict__ wpartials) {
  static_assert(!WG || (OC1 && !APPLY), "the fused weight gradient rides on the one-plane reduce pass");
  constexpr int VE = Elem<T>::kVec;
  constexpr int CV = 16 / VE, PT = 256 / CV;
  extern __shared__ float smem[];
  const int W = tg.W, H = tg.H, hw = H * W, OC = tg.OC;
  const int R = PT >> tg.wshift, pitch = W + 2, rows = R + 2;
  const int tile_floats = OC * rows * pitch;
  float* sD = smem;                                  // [2][OC][rows][pitch]
  float* sW = smem + 2 * tile_floats;                // [OC][9][16] (OC > 1 only)
  const int tid = threadIdx.x, cg = tid % CV, p = tid / CV;
  const int pr = p >> tg.wshift, pc = p & (W - 1);
  for (int i = tid; i < 2 * tile_floats; i += 256) sD[i] = 0.f;     // halo columns stay zero for the kernel's lifetime
  TailW<T> wr;
  if (OC1 && WG) {                                   // register budget goes to the 9 x VE weight-gradient accumulators: table in LDS
    uint32_t* sWq = reinterpret_cast<uint32_t*>(sW);
    for (int i = tid; i < 9 * CV * 4; i += 256) sWq[i] = TailW<T>::word(tg.w, ((i >> 2) % CV) * VE, i / (CV * 4), i & 3);
  } else if (OC1) wr.load(tg.w, cg * VE);
  else
    for (int i = tid; i < OC * 144; i += 256) {      // w[oc][ci][tap] -> sW[oc][tap][ci]
      const int oc = i / 144, rem = i - oc * 144, ci = rem / 9, t = rem - ci * 9;
      float wv = tg.w[i];
      if (VE == 8) wv = bf16_bits_to_f32(f32_to_bf16_bits(wv));      // the rounding the one-plane table (and the MFMA dgrad) applies
      sW[(oc * 9 + t) * 16 + ci] = wv;
    }
  float msc[VE], msh[VE], msc1[VE], msh1[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) { const int ch = cg * VE + j; msc[j] = ms[ch]; msh[j] = mb[ch]; msc1[j] = ms1[ch]; msh1[j] = mb1[ch]; }
  float a0[VE], b0[VE];
  float acc[3][VE];
  float* sC = sW + (OC1 ? (WG ? 9 * CV * 4 : 0) : OC * 144);            // [4][16] C0, C1, A1, B1 (APPLY: keeps the kernel under 168 VGPRs)
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    const int ch = cg * VE + j;
    if (APPLY) { a0[j] = A0[ch]; b0[j] = B0[ch]; }
    else { acc[0][j] = 0.f; acc[1][j] = 0.f; acc[2][j] = 0.f; }
  }
  float wacc[WG ? 9 : 1][VE];
#pragma unroll
  for (int t = 0; t < (WG ? 9 : 1); ++t)
#pragma unroll
    for (int j = 0; j < VE; ++j) wacc[t][j] = 0.f;
  if (APPLY && tid < 64) sC[tid] = (tid < 16 ? C0 : (tid < 32 ? C1 : (tid < 48 ? A1 : B1)))[tid & 15];
  __syncthreads();
  const int stage = rows * W;                        // d_raw elements per plane per tile (<= 3 * PT = 2 per thread)
  int buf = 0;
  Vec16 q0, q1;                                      // the tile's y vectors and (OC == 1) d_raw rows, loaded one tile ahead
  float dpre[2] = {0.f, 0.f};
  auto fetch = [&](int t) {
    const long v = (long)t * 256 + tid;
    q0 = load_nt(reinterpret_cast<const Vec16*>(y0) + v);
    q1 = load_nt(reinterpret_cast<const Vec16*>(y1) + v);
    if (OC1) {
      const int pix0 = t * PT, n = pix0 / hw, h0 = (pix0 - n * hw) >> tg.wshift;      // uniform
      const float* dp = tg.d_raw + (long)n * hw;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + k * 256, r = e >> tg.wshift, c = e & (W - 1), h = h0 - 1 + r;
        dpre[k] = (e < stage && h >= 0 && h < H) ? dp[h * W + c] : 0.f;
      }
    }
  };
  if ((int)blockIdx.x < tg.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < tg.ntiles; t += gridDim.x, buf ^= 1) {
    float* sT = sD + buf * tile_floats;
    // rows h0-1 .. h0+R of every plane -> LDS
    if (OC1) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + k * 256, r = e >> tg.wshift, c = e & (W - 1);
        if (e < stage) sT[r * pitch + c + 1] = dpre[k];
      }
    } else {
      const int pix0 = t * PT, n = pix0 / hw, h0 = (pix0 - n * hw) >> tg.wshift;        // uniform
      for (int oc = 0; oc < OC; ++oc) {
        const float* dp = tg.d_raw + ((long)n * OC + oc) * hw;
        for (int e = tid; e < stage; e += 256) {
          const int r = e >> tg.wshift, c = e & (W - 1), h = h0 - 1 + r;
          sT[(oc * rows + r) * pitch + c + 1] = (h >= 0 && h < H) ? dp[h * W + c] : 0.f;
        }
      }
    }
    const long v = (long)t * 256 + tid;
    float f0[VE], f1[VE];
    Elem<T>::unpack(q0, f0);
    Elem<T>::unpack(q1, f1);
    if (t + (int)gridDim.x < tg.ntiles) fetch(t + gridDim.x);
    __syncthreads();
    float g[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) g[j] = 0.f;
    float xj[VE];
    if (OC1) {
      if (WG) {
#pragma unroll
        for (int j = 0; j < VE; ++j) xj[j] = fmaxf((f0[j] * msc[j] + msh[j]) + (f1[j] * msc1[j] + msh1[j]), 0.f);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float d = sT[(pr + 2 - kh) * pitch + pc + 2 - kw];
          if (WG) TailW<T>::fma_words(reinterpret_cast<const uint4*>(sW)[(kh * 3 + kw) * CV + cg], d, g);
          else wr.fma(kh * 3 + kw, d, g);
          if (WG) {
#pragma unroll
            for (int j = 0; j < VE; ++j) wacc[kh * 3 + kw][j] += d * xj[j];
          }
        }
    } else {
      for (int oc = 0; oc < OC; ++oc)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float d = sT[(oc * rows + pr + 2 - kh) * pitch + pc + 2 - kw];
            const float* wp = sW + (oc * 9 + kh * 3 + kw) * 16 + cg * VE;
#pragma unroll
            for (int j = 0; j < VE; ++j) g[j] += d * wp[j];
          }
    }
    if (APPLY) {
      float r0[VE], r1[VE];
      int co = cg * VE;
      asm volatile("" : "+v"(co));          // re-read per tile: hoisted, the 2 x VE values cost the registers LDS was meant to save
      const float* c0 = sC + co;
      const float *c1 = sC + 16 + co, *a1 = sC + 32 + co, *b1 = sC + 48 + co;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float x = (f0[j] * msc[j] + msh[j]) + (f1[j] * msc1[j] + msh1[j]);
        const float gg = x > 0.f ? g[j] : 0.f;
        r0[j] = a0[j] * gg + b0[j] * f0[j] + c0[j];
        r1[j] = a1[j] * gg + b1[j] * f1[j] + c1[j];
      }
      store_nt(reinterpret_cast<Vec16*>(dy0) + v, Elem<T>::pack(r0));
      store_nt(reinterpret_cast<Vec16*>(dy1) + v, Elem<T>::pack(r1));
    } else {
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float x = WG ? xj[j] : (f0[j] * msc[j] + msh[j]) + (f1[j] * msc1[j] + msh1[j]);     // max(x, 0) > 0 <=> x > 0
        const float gg = x > 0.f ? g[j] : 0.f;
        acc[0][j] += gg; acc[1][j] += gg * f0[j]; acc[2][j] += gg * f1[j];
      }
    }
  }
  if (!APPLY) {
    __syncthreads();
    block_channel_reduce<3, VE>(acc, CV, 16, sC, partials + (long)blockIdx.x * 3 * 16);
  }
  if (WG) {
    // lanes with equal (lane % CV) share channels: butterfly over the other lane bits, then the four waves through LDS
    __syncthreads();
    float* sR = sC;                                  // [4 waves][CV][9 * VE]
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        float v = wacc[t][j];
#pragma unroll
        for (int m = CV; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane < CV) sR[(wv * CV + lane) * 9 * VE + t * VE + j] = v;
      }
    __syncthreads();
    if (tid < 144) {
      const int ci = tid / 9, t = tid - ci * 9, cgo = ci / VE, j = ci - cgo * VE;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += sR[(w * CV + cgo) * 9 * VE + t * VE + j];
      wpartials[(long)blockIdx.x * 144 + tid] = v;
    }
  }
}

// The reduce pass (bf16, one plane, W >= 32, fused weight gradient) on the matrix cores.  tail_join_bwd_kernel<.., WG> spends ~420
// VALU instructions per wave and 128-pixel tile (72 + 72 FMAs and as many bf16 unpacks per thread) and is issue-bound at 2.9 TB/s.
// Here the lane <-> data mapping is the MFMA D fragment's: lane (r = lane & 15, gq = lane >> 4) owns channels 4gq .. 4gq+3 of pixel r of a
// 16-pixel group (8-byte loads; a wave covers 32 pixels = two groups per tile).
//   data gradient  g[ci][pixel] = sum_tap w[ci][tap] * d[pixel (+) tap]: ONE 16x16x32 MFMA per group -- A = the bf16 weights [ci][k],
//     B[k][pixel] = the 9 shifted d_raw values of the pixel, as bf16 hi parts in k = 0..8 and lo parts (d - hi) in k = 16..24 against the
//     same weights, so d_raw keeps (nearly) its f32 precision like in the VALU form;
//   weight gradient dW[ci][tap] = sum_pixel x[pixel][ci] * d[pixel (+) tap], K = the wave's 32 pixels: this conv feeds a BatchNorm, so its true
//     weight gradient is the small remainder of terms that cancel ~1000-fold and plain bf16 operands leave nothing of it (rel-L2 1.95 at
//     config 2's size).  Both operands are therefore split hi + lo (two bf16 each, ~2^-17 relative) and the product is three bf16
//     MFMAs per wave and tile, xh dh + xh dl + xl dh (eight 16x16x4 f32 MFMAs give the same result and cost 70 us more: 256 vs 48 cycles of
//     the matrix pipe per wave and tile).  x hi / lo go through two wave-private bf16 LDS patches [pixel][16] and come back channel-major
//     (ds_read_b64_tr_b16); B = 8 consecutive d_raw values per lane (tap = r < 9).
// Same tiles, d_raw staging, partial-row layouts and fixed summation order as the VALU form.
// F8I: y0 / y1 are e4m3 bytes (fp8 mode's storage of these two tensors): a lane's 4 channels of a pixel are one dword
template <bool F8I>
__global__ __launch_bounds__(256, 4) void tail_reduce_mfma_kernel(TailG tg, const float* __restrict__ ms, const float* __restrict__ mb,
                                                                  const float* __restrict__ ms1, const float* __restrict__ mb1,
                                                                  const bf16_t* __restrict__ y0, const bf16_t* __restrict__ y1,
                                                                  float* __restrict__ partials, float* __restrict__ wpartials) {
  constexpr int PT = 128;
  extern __shared__ float smem[];
  const int W = tg.W, H = tg.H, hw = H * W;
  const int R = PT >> tg.wshift, pitch = W + 2, rows = R + 2;
  const int tile_floats = ((rows * pitch + 3) & ~3) + 8;               // + 8 zeros: where lanes without a tap read their operand
  const int zoff = tile_floats - 8;
  float* sD = smem;                                                    // [2 buffers][hi, lo][rows][pitch]: d_raw split at staging time
  float* sX = smem + 4 * tile_floats;                                 // [4 waves][hi, lo][32 pixels][16 channels] bf16
  float* sR = sX + 4 * 512;                                           // [4][48] sums, [4][144] weight-gradient fragments
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
  const bool odd = gq & 1, lo_half = gq >= 2;
  for (int i = tid; i < 4 * tile_floats; i += 256) sD[i] = 0.f;       // halo columns stay zero for the kernel's lifetime
  // A of the data gradient: [ci = r][k = 8gq + t] = w[ci][tap = 8 (gq & 1) + t] (taps >= 9: zero), the same for the hi and the lo k range
  Vec16 wA;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t0 = 8 * (gq & 1) + 2 * k;
    wA.w[k] = pack2_bf16(t0 < 9 ? tg.w[r * 9 + t0] : 0.f, t0 + 1 < 9 ? tg.w[r * 9 + t0 + 1] : 0.f);
  }
  float msc[4], msh[4], msc1[4], msh1[4], acc[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = 4 * gq + j;
    msc[j] = ms[ch]; msh[j] = mb[ch] + mb1[ch]; msc1[j] = ms1[ch]; msh1[j] = 0.f;      // (the join's two shifts added up front)
    acc[0][j] = 0.f; acc[1][j] = 0.f; acc[2][j] = 0.f;
  }
  (void)msh1;
  f32x4 macc = {0.f, 0.f, 0.f, 0.f};
  char* sXw = reinterpret_cast<char*>(sX + wv * 512);
  const int p0 = 32 * wv, pr = p0 >> tg.wshift, c0 = p0 & (W - 1);     // the wave's 32 pixels: one row (W >= 32)
  // weight gradient operands: A = x^T out of the patches; B: lane (tap = r, k slice gq) reads d_raw of pixels c0 + 8gq .. + 7 shifted by its tap
  const int a0 = (8 * gq + (r >> 2)) * 32 + (r & 3) * 8, a1 = a0 + 4 * 32;
  const int tw = r < 9 ? r : 8, twh = tw / 3, tww = tw - 3 * twh;
  const int wofs = r < 9 ? (pr + 2 - twh) * pitch + c0 + 8 * gq + 2 - tww : zoff;
  const int stage = rows * W;
  int buf = 0;
  uint2 q0[2], q1[2];
  float dpre[2] = {0.f, 0.f};
  auto fetch = [&](int t) {
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const long e = ((long)t * PT + p0 + 16 * pt + r) * 16 + 4 * gq;
      if constexpr (F8I) {
        q0[pt] = make_uint2(*reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(y0) + e), 0u);
        q1[pt] = make_uint2(*reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(y1) + e), 0u);
      } else {
        q0[pt] = *reinterpret_cast<const uint2*>(y0 + e);
        q1[pt] = *reinterpret_cast<const uint2*>(y1 + e);
      }
    }
    const int pix0 = t * PT, n = pix0 / hw, h0 = (pix0 - n * hw) >> tg.wshift;      // uniform
    const float* dp = tg.d_raw + (long)n * hw;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + k * 256, rr = e >> tg.wshift, c = e & (W - 1), h = h0 - 1 + rr;
      dpre[k] = (e < stage && h >= 0 && h < H) ? dp[h * W + c] : 0.f;
    }
  };
  __syncthreads();
  if ((int)blockIdx.x < tg.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < tg.ntiles; t += gridDim.x, buf ^= 1) {
    float* sT = sD + buf * 2 * tile_floats;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + k * 256, rr = e >> tg.wshift, c = e & (W - 1);
      const float hi = __uint_as_float(pack2_bf16(dpre[k], 0.f) << 16);       // d = hi + lo, hi an exact bf16; lo is cut to bf16 when packed
      if (e < stage) { sT[rr * pitch + c + 1] = hi; sT[tile_floats + rr * pitch + c + 1] = dpre[k] - hi; }
    }
    const float* img = sT + (lo_half ? tile_floats : 0);
    uint2 c0v[2] = {q0[0], q0[1]}, c1v[2] = {q1[0], q1[1]};
    if (t + (int)gridDim.x < tg.ntiles) fetch(t + gridDim.x);
    __syncthreads();
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int base = (pr + 2) * pitch + c0 + 16 * pt + r + 2;
      // B[k = 8gq + t][pixel r]: taps 0..7 (even gq) / tap 8 in slot 0 (odd gq); hi parts (gq < 2) / lo parts (gq >= 2)
      // (odd gq: only k-slot 0 = tap 8 meets a non-zero weight in wA; its other seven values are finite staged data and multiply zeros --
      // no per-element selects, constant offsets from one base)
      float dv[8];
      dv[0] = img[base - (odd ? 2 * pitch + 2 : 0)];
#pragma unroll
      for (int k = 1; k < 8; ++k) dv[k] = img[base - (k / 3) * pitch - (k % 3)];
      Vec16 bf;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        bf.w[k] = __builtin_amdgcn_perm(__float_as_uint(dv[2 * k + 1]), __float_as_uint(dv[2 * k]), 0x07060302u);     // the two upper halves
      const f32x4 g = mma_bf16(wA, bf, (f32x4){0.f, 0.f, 0.f, 0.f});
      float f0[4], f1[4];
      if constexpr (F8I) { unpack4_fp8(c0v[pt].x, f0); unpack4_fp8(c1v[pt].x, f1); }
      else { unpack4_bf16(c0v[pt], f0); unpack4_bf16(c1v[pt], f1); }
      float x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = fmaxf(f1[j] * msc1[j] + (f0[j] * msc[j] + msh[j]), 0.f);
        const float gg = x[j] > 0.f ? g[j] : 0.f;
        acc[0][j] += gg; acc[1][j] += gg * f0[j]; acc[2][j] += gg * f1[j];
      }
      const uint32_t h0 = pack2_bf16(x[0], x[1]), h1 = pack2_bf16(x[2], x[3]);
      const uint32_t l0 = pack2_bf16(x[0] - __uint_as_float(h0 << 16), x[1] - __uint_as_float(h0 & 0xffff0000u));
      const uint32_t l1 = pack2_bf16(x[2] - __uint_as_float(h1 << 16), x[3] - __uint_as_float(h1 & 0xffff0000u));
      *reinterpret_cast<uint2*>(sXw + (16 * pt + r) * 32 + gq * 8) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(sXw + 1024 + (16 * pt + r) * 32 + gq * 8) = make_uint2(l0, l1);
    }
    {
      Vec16 bh, bl;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bh.w[k] = __builtin_amdgcn_perm(__float_as_uint(sT[wofs + 2 * k + 1]), __float_as_uint(sT[wofs + 2 * k]), 0x07060302u);
        bl.w[k] = __builtin_amdgcn_perm(__float_as_uint(sT[tile_floats + wofs + 2 * k + 1]), __float_as_uint(sT[tile_floats + wofs + 2 * k]), 0x07060302u);
      }
      const Vec16 ah = FragOps<bf16_t>::load(sXw, a0, a1), al = FragOps<bf16_t>::load(sXw + 1024, a0, a1);
      macc = mma_bf16(ah, bh, macc);
      macc = mma_bf16(ah, bl, macc);
      macc = mma_bf16(al, bh, macc);
    }
  }
  // ---- block sums: the 16 pixel lanes of a row (DPP), then the four waves in order through LDS
#pragma unroll
  for (int v = 0; v < 3; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sum = row16_sum(acc[v][j]);
      if (r == 0) sR[wv * 48 + v * 16 + 4 * gq + j] = sum;
    }
  if (r < 9) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sR[192 + wv * 144 + (4 * gq + j) * 9 + r] = macc[j];      // D fragment: rows ci = 4gq + j, column tap = r
  }
  __syncthreads();
  if (tid < 48) partials[(long)blockIdx.x * 48 + tid] = (sR[tid] + sR[48 + tid]) + (sR[96 + tid] + sR[144 + tid]);
  if (tid < 144) wpartials[(long)blockIdx.x * 144 + tid] = (sR[192 + tid] + sR[336 + tid]) + (sR[480 + tid] + sR[624 + tid]);
}

// The apply pass in the same form: g from one hi/lo-split MFMA per 16 pixels, dy0 = A0 g + B0 y0 + C0 and dy1 = A1 g + B1 y1 + C1 for the
// lane's four channels, 8-byte loads and stores.
__global__ __launch_bounds__(256, 4) void tail_apply_mfma_kernel(TailG tg, const float* __restrict__ ms, const float* __restrict__ mb,
                                                                 const float* __restrict__ ms1, const float* __restrict__ mb1,
                                                                 const bf16_t* __restrict__ y0, const float* __restrict__ A0, const float* __restrict__ B0,
                                                                 const float* __restrict__ C0, bf16_t* __restrict__ dy0, const bf16_t* __restrict__ y1,
                                                                 const float* __restrict__ A1, const float* __restrict__ B1,
                                                                 const float* __restrict__ C1, bf16_t* __restrict__ dy1) {
  constexpr int PT = 128;
  extern __shared__ float smem[];
  const int W = tg.W, H = tg.H, hw = H * W;
  const int R = PT >> tg.wshift, pitch = W + 2, rows = R + 2;
  const int tile_floats = (rows * pitch + 3) & ~3;
  float* sD = smem;                                                    // [2][rows][pitch]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
  const bool odd = gq & 1, lo_half = gq >= 2;
  for (int i = tid; i < 2 * tile_floats; i += 256) sD[i] = 0.f;
  Vec16 wA;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t0 = 8 * (gq & 1) + 2 * k;
    wA.w[k] = pack2_bf16(t0 < 9 ? tg.w[r * 9 + t0] : 0.f, t0 + 1 < 9 ? tg.w[r * 9 + t0 + 1] : 0.f);
  }
  float msc[4], msh[4], msc1[4], msh1[4], ca0[4], cb0[4], cc0[4], ca1[4], cb1[4], cc1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = 4 * gq + j;
    msc[j] = ms[ch]; msh[j] = mb[ch]; msc1[j] = ms1[ch]; msh1[j] = mb1[ch];
    ca0[j] = A0[ch]; cb0[j] = B0[ch]; cc0[j] = C0[ch]; ca1[j] = A1[ch]; cb1[j] = B1[ch]; cc1[j] = C1[ch];
  }
  const int p0 = 32 * wv, pr = p0 >> tg.wshift, c0 = p0 & (W - 1);
  const int stage = rows * W;
  int buf = 0;
  uint2 q0[2], q1[2];
  float dpre[2] = {0.f, 0.f};
  auto fetch = [&](int t) {
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const long e = ((long)t * PT + p0 + 16 * pt + r) * 16 + 4 * gq;
      q0[pt] = load_nt(reinterpret_cast<const uint2*>(y0 + e));
      q1[pt] = load_nt(reinterpret_cast<const uint2*>(y1 + e));
    }
    const int pix0 = t * PT, n = pix0 / hw, h0 = (pix0 - n * hw) >> tg.wshift;      // uniform
    const float* dp = tg.d_raw + (long)n * hw;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + k * 256, rr = e >> tg.wshift, c = e & (W - 1), h = h0 - 1 + rr;
      dpre[k] = (e < stage && h >= 0 && h < H) ? dp[h * W + c] : 0.f;
    }
  };
  __syncthreads();
  if ((int)blockIdx.x < tg.ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < tg.ntiles; t += gridDim.x, buf ^= 1) {
    float* sT = sD + buf * tile_floats;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + k * 256, rr = e >> tg.wshift, c = e & (W - 1);
      if (e < stage) sT[rr * pitch + c + 1] = dpre[k];
    }
    uint2 c0v[2] = {q0[0], q0[1]}, c1v[2] = {q1[0], q1[1]};
    if (t + (int)gridDim.x < tg.ntiles) fetch(t + gridDim.x);
    __syncthreads();
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int base = (pr + 2) * pitch + c0 + 16 * pt + r + 2;
      float dv[8];
      dv[0] = sT[base - (odd ? 2 * pitch + 2 : 0)];
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        const float v = sT[base - (k / 3) * pitch - (k % 3)];
        dv[k] = odd ? 0.f : v;
      }
      Vec16 bf;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t hi = pack2_bf16(dv[2 * k], dv[2 * k + 1]);
        const uint32_t lo = pack2_bf16(dv[2 * k] - __uint_as_float(hi << 16), dv[2 * k + 1] - __uint_as_float(hi & 0xffff0000u));
        bf.w[k] = lo_half ? lo : hi;
      }
      const f32x4 g = mma_bf16(wA, bf, (f32x4){0.f, 0.f, 0.f, 0.f});
      const float f0[4] = {__uint_as_float(c0v[pt].x << 16), __uint_as_float(c0v[pt].x & 0xffff0000u), __uint_as_float(c0v[pt].y << 16),
                           __uint_as_float(c0v[pt].y & 0xffff0000u)};
      const float f1[4] = {__uint_as_float(c1v[pt].x << 16), __uint_as_float(c1v[pt].x & 0xffff0000u), __uint_as_float(c1v[pt].y << 16),
                           __uint_as_float(c1v[pt].y & 0xffff0000u)};
      float r0[4], r1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = (f0[j] * msc[j] + msh[j]) + (f1[j] * msc1[j] + msh1[j]);
        const float gg = x > 0.f ? g[j] : 0.f;
        r0[j] = ca0[j] * gg + cb0[j] * f0[j] + cc0[j];
        r1[j] = ca1[j] * gg + cb1[j] * f1[j] + cc1[j];
      }
      const long e = ((long)t * PT + p0 + 16 * pt + r) * 16 + 4 * gq;
      store_nt(reinterpret_cast<uint2*>(dy0 + e), make_uint2(pack2_bf16(r0[0], r0[1]), pack2_bf16(r0[2], r0[3])));
      store_nt(reinterpret_cast<uint2*>(dy1 + e), make_uint2(pack2_bf16(r1[0], r1[1]), pack2_bf16(r1[2], r1[3])));
    }
  }
}

// Whether the tile walk above covers an (OC, H, W) output with element type dt; the caller keeps the separate dgrad otherwise.
bool tail_join_fusable(int dt, int OC, int N, int H, int W) {
  const int cv = dt == DT_F32 ? 4 : 2, pt = 256 / cv;
  if (OC < 1 || OC > 8 || W < 1 || (W & (W - 1)) || W > pt) return false;
  if (((long)H * W) % pt) return false;
  return (long)N * H * W / pt < (1L << 30);
}

template <bool APPLY>
static int launch_tail_join(int dt, const float* d_raw, const float* w, int OC, int N, int H, int W, const float* ms, const float* mb,
                            const float* ms1, const float* mb1, const void* y0, const float* A0, const float* B0, const float* C0, void* dy0,
                            const void* y1, const float* A1, const float* B1, const float* C1, void* dy1, float* partials, float* wpartials,
                            hipStream_t s, int f8in = 0) {
  if (f8in && (APPLY || !wpartials || dt == DT_F32 || OC != 1 || W < 32)) { set_error("tail_join_bwd: e4m3 y2 / ys only in the one-plane MFMA reduce pass"); return MMVAE_ERR_UNSUPPORTED; }
  if (wpartials && (APPLY || OC != 1)) { set_error("tail_join_bwd: fused wgrad needs the one-plane reduce pass"); return MMVAE_ERR_UNSUPPORTED; }
  if (!tail_join_fusable(dt, OC, N, H, W)) { set_error("tail_join_bwd: OC=%d H=%d W=%d not supported", OC, H, W); return MMVAE_ERR_UNSUPPORTED; }
  const int VE = dt == DT_F32 ? 4 : 8, cv = 16 / VE, pt = 256 / cv;
  int wshift = 0;
  while ((1 << wshift) < W) ++wshift;
  const int ntiles = (int)((long)N * H * W / pt);
  // apply: three 256-thread blocks fit a CU (launch bounds), so 768 blocks are exactly one resident round on 256 CUs; 2048 left the
  // last 2/3 round to a third of the chip (535 -> 506 us at 5120 frames).  The reduce pass (two per CU) keeps its two full rounds.
  const int cap = APPLY ? (OC == 1 ? 768 : kElemMaxBlocks) : 1024;
  int blocks = ntiles < cap ? ntiles : cap;
  const TailG tg{d_raw, w, OC, H, W, wshift, ntiles};
  const int tile_floats = OC * (pt / W + 2) * (W + 2);
  const size_t sm = ((size_t)2 * tile_floats + (OC == 1 ? (wpartials ? 9 * cv * 4 : 0) : OC * 144) + (APPLY ? 64 : 256 * 3 * VE)) * sizeof(float);
#define MMVAE_LAUNCH(T, OC1, WG) hipLaunchKernelGGL((tail_join_bwd_kernel<T, OC1, APPLY, WG>), dim3(blocks), dim3(256), sm, s, tg, ms, mb, ms1, mb1, \
    (const T*)y0, A0, B0, C0, (T*)dy0, (const T*)y1, A1, B1, C1, (T*)dy1, partials, wpartials)
  if constexpr (APPLY) {
    if (dt != DT_F32 && OC == 1 && W >= 32) {
      // measured at 5120 frames: 512 blocks 462 us, 640: 499, 704: 460, 768: 443, 896 / 1024: 521, 1280: 506 (five fit a CU) -- whole
      // multiples of the CU count, and no more concurrent read + write streams than the memory system likes
      const int nb = ntiles < 768 ? ntiles : 768;
      const int tf = ((pt / W + 2) * (W + 2) + 3) & ~3;
      hipLaunchKernelGGL(tail_apply_mfma_kernel, dim3(nb), dim3(256), (size_t)2 * tf * sizeof(float), s, tg, ms, mb, ms1, mb1, (const bf16_t*)y0, A0, B0, C0,
                         (bf16_t*)dy0, (const bf16_t*)y1, A1, B1, C1, (bf16_t*)dy1);
      note_launch_bytes((double)N * H * W * (4 * 16 * 2.0 + 4.0));
      const int rcm = check_launch("tail_apply_mfma");
      return rcm ? rcm : nb;
    }
  }
  if constexpr (!APPLY) {
    if (wpartials && dt != DT_F32 && OC == 1 && W >= 32) {
      const int nb = ntiles < 1024 ? ntiles : 1024;                  // four blocks per CU (114 VGPRs): one resident round (314 us; 768: 345, 512: 411)
      const int tf = ((pt / W + 2) * (W + 2) + 3) & ~3;
      const size_t lds = ((size_t)4 * (tf + 8) + 4 * 512 + 192 + 576) * sizeof(float);
      if (f8in) hipLaunchKernelGGL(tail_reduce_mfma_kernel<true>, dim3(nb), dim3(256), lds, s, tg, ms, mb, ms1, mb1, (const bf16_t*)y0, (const bf16_t*)y1, partials, wpartials);
      else hipLaunchKernelGGL(tail_reduce_mfma_kernel<false>, dim3(nb), dim3(256), lds, s, tg, ms, mb, ms1, mb1, (const bf16_t*)y0, (const bf16_t*)y1, partials, wpartials);
      note_launch_bytes((double)N * H * W * (2 * 16 * (f8in ? 1.0 : 2.0) + 4.0));
      const int rcm = check_launch("tail_reduce_mfma");
      return rcm ? rcm : nb;
    }
    if (wpartials) { if (dt == DT_F32) MMVAE_LAUNCH(float, true, true); else MMVAE_LAUNCH(bf16_t, true, true); }
  }
  if (!wpartials) {
    if (dt == DT_F32) { if (OC == 1) MMVAE_LAUNCH(float, true, false); else MMVAE_LAUNCH(float, false, false); }
    else { if (OC == 1) MMVAE_LAUNCH(bf16_t, true, false); else MMVAE_LAUNCH(bf16_t, false, false); }
  }
#undef MMVAE_LAUNCH
  const int rc = check_launch(APPLY ? "tail_join_bwd_apply" : "tail_join_bwd_reduce");
  return rc ? rc : blocks;
}

// wpartials != NULL (OC == 1): the pass also leaves the tail conv's weight-gradient partials [blocks][144] there; finish with
// launch_tail_wgrad_finalize(wpartials, <returned block count>, dW).
int launch_tail_join_bwd_reduce(int dt, const float* d_raw, const float* w, int OC, int N, int H, int W, const float* ms, const float* mb,
                                const float* ms1, const float* mb1, const void* y0, const void* y1, float* partials, hipStream_t s,
                                float* wpartials, int f8in) {
  return launch_tail_join<false>(dt, d_raw, w, OC, N, H, W, ms, mb, ms1, mb1, y0, nullptr, nullptr, nullptr, nullptr, y1, nullptr, nullptr, nullptr,
                                 nullptr, partials, wpartials, s, f8in);
}

int launch_tail_join_bwd_apply(int dt, const float* d_raw, const float* w, int OC, int N, int H, int W, const float* ms, const float* mb,
                               const float* ms1, const float* mb1, const void* y0, const float* A0, const float* B0, const float* C0, void* dy0,
                               const void* y1, const float* A1, const float* B1, const float* C1, void* dy1, hipStream_t s) {
  const int rc = launch_tail_join<true>(dt, d_raw, w, OC, N, H, W, ms, mb, ms1, mb1, y0, A0, B0, C0, dy0, y1, A1, B1, C1, dy1, nullptr, nullptr, s);
  return rc < 0 ? rc : MMVAE_OK;
}

// ---------------------------------------------------------------- last up-block: residual join fused with the tail conv (forward)
// r_raw[n,h,w] = bias + sum_{ci,kh,kw} x[n,h+kh-1,w+kw-1,ci] * w[ci][kh][kw],  x = relu(bn(y2) + bn(ys))   (model.py:193, one plane)
// The joined activation x (671 MB at N = 5120) is never written: one block walks one image top to bottom, RP = PT/W rows per
// pass.  Each thread turns its 16-byte vector of x into 9 per-tap partial dot products, the CV lanes of a pixel combine them
// by shuffle and park them in a 3-pass (4 when RP = 1) LDS ring S[tap][row][col]; an output pixel is then the sum of 9 ring entries.  Every
// input row is read exactly once (2 x 671 MB in, 84 MB out, against 4 x 671 MB for join + conv).  One barrier per pass: pass
// k+1 overwrites the ring rows of pass k-2, whose last readers (the outputs of pass k-1) are behind barrier k.
// Per-block (sum, sum of squares) of r_raw go to stats[block][2] for the output BatchNorm.
struct TailFwd { const float* w; const float* bias; float* r_raw; float* stats; int H, W, wshift; };

template <typename T>
__global__ __launch_bounds__(256, 3) void tail_join_fwd_kernel(TailFwd a, const T* __restrict__ y0, const float* __restrict__ ms,
                                                               const float* __restrict__ mb, const T* __restrict__ y1,
                                                               const float* __restrict__ ms1, const float* __restrict__ mb1) {
  constexpr int VE = Elem<T>::kVec;
  constexpr int CV = 16 / VE, PT = 256 / CV;
  extern __shared__ float smem[];
  const int W = a.W, H = a.H;
  const int RP = PT >> a.wshift, NRING = RP >= 2 ? 3 : 4, RR = NRING * RP, pitch = W + 2;   // outputs reach 2 rows back: 1 pass, or 2 when RP = 1
  float* sS = smem;                                  // [9][RR][pitch]
  const int tid = threadIdx.x, cg = tid % CV, p = tid / CV;
  const int pr = p >> a.wshift, pc = p & (W - 1);
  for (int i = tid; i < 9 * RR * pitch; i += 256) sS[i] = 0.f;      // halo columns stay zero
  TailW<T> wr;
  wr.load(a.w, cg * VE);
  float msc[VE], msh[VE], msc1[VE], msh1[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) { const int ch = cg * VE + j; msc[j] = ms[ch]; msh[j] = mb[ch]; msc1[j] = ms1[ch]; msh1[j] = mb1[ch]; }
  const float bias = a.bias[0];
  const long n = blockIdx.x;
  const Vec16* v0 = reinterpret_cast<const Vec16*>(y0) + n * H * W * CV + tid;
  const Vec16* v1 = reinterpret_cast<const Vec16*>(y1) + n * H * W * CV + tid;
  float* out = a.r_raw + n * H * W;
  const int npass = H / RP;
  Vec16 q0 = v0[0], q1 = v1[0];
  float s1 = 0.f, s2 = 0.f;
  __syncthreads();
  int k3 = 0;                                        // k mod NRING
  for (int k = 0; k <= npass; ++k) {
    const int slot = k3 * RP + pr;
    float sp[9];
    if (k < npass) {
      float f0[VE], f1[VE], x[VE];
      Elem<T>::unpack(q0, f0);
      Elem<T>::unpack(q1, f1);
      if (k + 1 < npass) { q0 = v0[(long)(k + 1) * 256]; q1 = v1[(long)(k + 1) * 256]; }   // default policy: the tail of y2 / ys is still in the Infinity Cache
#pragma unroll
      for (int j = 0; j < VE; ++j) x[j] = fmaxf((f0[j] * msc[j] + msh[j]) + (f1[j] * msc1[j] + msh1[j]), 0.f);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float v = wr.dot(t, x);
#pragma unroll
        for (int m = 1; m < CV; m <<= 1) v += __shfl_xor(v, m);
        sp[t] = v;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) sp[t] = 0.f;       // the row below the image
    }
    if (cg == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t) sS[(t * RR + slot) * pitch + pc + 1] = sp[t];
    }
    __syncthreads();
    if (tid < PT) {
      const int ro = tid >> a.wshift, col = tid & (W - 1);
      const int h = k * RP - 1 + ro;
      if (h >= 0 && h < H) {
        float v = bias;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int hh = h + kh - 1;
          int sl = k3 * RP + ro + kh - 2 + RR;       // ring slot of row hh
          if (sl >= RR) sl -= RR;
          if (hh >= 0 && hh < H) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) v += sS[((kh * 3 + kw) * RR + sl) * pitch + col + kw];
          }
        }
        out[h * W + col] = v;
        s1 += v; s2 += v * v;
      }
    }
    k3 = k3 == NRING - 1 ? 0 : k3 + 1;
  }
  if (a.stats) {
    __shared__ float sRed[2][4];
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((tid & 63) == 0) { sRed[0][tid >> 6] = s1; sRed[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid < 2) a.stats[(long)blockIdx.x * 2 + tid] = (sRed[tid][0] + sRed[tid][1]) + (sRed[tid][2] + sRed[tid][3]);
  }
}

// Whether the image walk above covers an H x W output with element type dt (one output plane only).
bool tail_fwd_fusable(int dt, int OC, int N, int H, int W) {
  const int cv = dt == DT_F32 ? 4 : 2, pt = 256 / cv;
  if (OC != 1 || W < 1 || (W & (W - 1)) || W > pt) return false;
  return H % (pt / W) == 0 && N >= 1;
}

// returns the number of stats partials (= N) or a negative error
int launch_tail_join_fwd(int dt, const void* y0, const float* ms, const float* mb, const void* y1, const float* ms1, const float* mb1,
                         const float* w, const float* bias, float* r_raw, float* stats, int N, int H, int W, hipStream_t s) {
  if (!tail_fwd_fusable(dt, 1, N, H, W)) { set_error("tail_join_fwd: H=%d W=%d not supported", H, W); return MMVAE_ERR_UNSUPPORTED; }
  const int cv = dt == DT_F32 ? 4 : 2, pt = 256 / cv;
  int wshift = 0;
  while ((1 << wshift) < W) ++wshift;
  const TailFwd a{w, bias, r_raw, stats, H, W, wshift};
  const int rp = pt / W;
  const size_t sm = (size_t)9 * (rp >= 2 ? 3 : 4) * rp * (W + 2) * sizeof(float);
  if (dt == DT_F32) hipLaunchKernelGGL((tail_join_fwd_kernel<float>), dim3(N), dim3(256), sm, s, a, (const float*)y0, ms, mb, (const float*)y1, ms1, mb1);
  else hipLaunchKernelGGL((tail_join_fwd_kernel<bf16_t>), dim3(N), dim3(256), sm, s, a, (const bf16_t*)y0, ms, mb, (const bf16_t*)y1, ms1, mb1);
  const int rc = check_launch("tail_join_fwd");
  return rc ? rc : N;
}


// ---------------------------------------------------------------- NCHW f32 elementwise (output BN)
__global__ __launch_bounds__(64) void tail_wgrad_finalize_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ dW) {
  double s = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 64) s += (double)partials[(long)p * 144 + blockIdx.x];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) dW[blockIdx.x] += (float)s;
}

int launch_tail_wgrad_finalize(const float* wpartials, int nparts, float* dW, hipStream_t s) {
  hipLaunchKernelGGL(tail_wgrad_finalize_kernel, dim3(144), dim3(64), 0, s, wpartials, nparts, dW);
  return check_launch("tail_wgrad_finalize");
}

__global__ void affine_nchw_kernel(const float* __restrict__ raw, const float* __restrict__ scale, const float* __restrict__ shift,
                                   float* __restrict__ out, long total, int C, int HW) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    out[i] = raw[i] * scale[c] + shift[c];
  }
}
int launch_affine_nchw(const float* raw, const float* scale, const float* shift, float* out, int N, int C, int HW, hipStream_t s) {
  const long total = (long)N * C * HW;
  const int blocks = elem_blocks(total / 4 + 1, 256);
  hipLaunchKernelGGL(affine_nchw_kernel, dim3(blocks), dim3(256), 0, s, raw, scale, shift, out, total, C, HW);
  note_launch_bytes((double)total * 8.0);
  return check_launch("affine_nchw");
}
// ---- Gaussian reconstruction loss fused into the output BatchNorm's backward (reference model.py:193, :403): the loss gradient
//   d[n,c,i] = k * (scale[c] * raw[n,c,i] + shift[c] - target[n,c,i]),  k = coef / sigma^2 (* the upstream gradient, a device scalar)
// is a function of the conv output `raw` and the target alone, so neither d_recon nor a second read of recon is needed: the reduce pass
// sums (d, d * raw) per channel straight from raw and target, the apply pass writes d_raw = A d + B raw + C.  Two passes over the plane
// (2 reads each, 1 write) instead of gauss_nll_bwd + plane_reduce + apply (6 reads, 2 writes).  Same block / summation structure as
// plane_reduce_nchw_kernel (bit-reproducible).
__global__ __launch_bounds__(256) void gauss_tail_reduce_kernel(const float* __restrict__ raw, const float* __restrict__ tgt, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float k, const float* __restrict__ gs, int N, int C,
                                                                 int HW, float* __restrict__ partials) {
  extern __shared__ float sm[];
  float* sAcc = sm;             // [4 waves][2][C]
  const int wid = threadIdx.x >> 6;
  if (gs) k *= gs[0];
  for (int i = threadIdx.x; i < 4 * 2 * C; i += blockDim.x) sAcc[i] = 0.f;
  __syncthreads();
  const int planes = N * C;
  for (int p = blockIdx.x; p < planes; p += gridDim.x) {
    const int c = p % C;
    const float sc = scale[c], sh = shift[c];
    const float* pa = raw + (long)p * HW;
    const float* pt = tgt + (long)p * HW;
    float v0 = 0.f, v1 = 0.f;
    if ((HW & 3) == 0) {
      for (int i = threadIdx.x * 4; i < HW; i += 4 * blockDim.x) {
        const float4 x = *reinterpret_cast<const float4*>(pa + i);
        const float4 t = *reinterpret_cast<const float4*>(pt + i);
        const float d0 = k * ((sc * x.x + sh) - t.x), d1 = k * ((sc * x.y + sh) - t.y), d2 = k * ((sc * x.z + sh) - t.z), d3 = k * ((sc * x.w + sh) - t.w);
        v0 += (d0 + d1) + (d2 + d3);
        v1 += (d0 * x.x + d1 * x.y) + (d2 * x.z + d3 * x.w);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float x = pa[i], d = k * ((sc * x + sh) - pt[i]);
        v0 += d; v1 += d * x;
      }
    }
    v0 = wave_sum(v0); v1 = wave_sum(v1);
    if ((threadIdx.x & 63) == 0) { sAcc[wid * 2 * C + c] += v0; sAcc[wid * 2 * C + C + c] += v1; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x)
    partials[(long)blockIdx.x * 2 * C + i] = (sAcc[i] + sAcc[2 * C + i]) + (sAcc[2 * 2 * C + i] + sAcc[3 * 2 * C + i]);
}
int launch_gauss_tail_reduce(const float* raw, const float* target, const float* scale, const float* shift, float sigma, float coef, const float* gscale,
                             int N, int C, int HW, float* partials, hipStream_t s) {
  const int blocks = nchw_parts(N, C);
  const size_t sm = (size_t)4 * 2 * C * sizeof(float);
  hipLaunchKernelGGL(gauss_tail_reduce_kernel, dim3(blocks), dim3(256), sm, s, raw, target, scale, shift, coef / (sigma * sigma), gscale, N, C, HW, partials);
  note_launch_bytes((double)N * C * HW * 8.0);
  const int rc = check_launch("gauss_tail_reduce");
  return rc ? rc : blocks;
}
__global__ __launch_bounds__(256) void gauss_tail_apply_kernel(const float* __restrict__ raw, const float* __restrict__ tgt, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float k, const float* __restrict__ gs,
                                                                const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cc,
                                                                float* __restrict__ dy, int planes, int C, int HW) {
  if (gs) k *= gs[0];
  for (int p = blockIdx.x; p < planes; p += gridDim.x) {
    const int c = p % C;
    const float sc = scale[c], sh = shift[c], ca = A[c], cb = B[c], cc = Cc[c];
    const long base = (long)p * HW;
    if ((HW & 3) == 0) {
      for (int i = threadIdx.x * 4; i < HW; i += 1024) {
        const float4 x = *reinterpret_cast<const float4*>(raw + base + i);
        const float4 t = *reinterpret_cast<const float4*>(tgt + base + i);
        float4 o;
        o.x = ca * (k * ((sc * x.x + sh) - t.x)) + cb * x.x + cc; o.y = ca * (k * ((sc * x.y + sh) - t.y)) + cb * x.y + cc;
        o.z = ca * (k * ((sc * x.z + sh) - t.z)) + cb * x.z + cc; o.w = ca * (k * ((sc * x.w + sh) - t.w)) + cb * x.w + cc;
        *reinterpret_cast<float4*>(dy + base + i) = o;
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += 256) {
        const float x = raw[base + i];
        dy[base + i] = ca * (k * ((sc * x + sh) - tgt[base + i])) + cb * x + cc;
      }
    }
  }
}
int launch_gauss_tail_apply(const float* raw, const float* target, const float* scale, const float* shift, float sigma, float coef, const float* gscale,
                            const float* A, const float* B, const float* Cc, float* dy, int N, int C, int HW, hipStream_t s) {
  if (C > 64) { set_error("gauss_tail_apply: C=%d > 64", C); return MMVAE_ERR_UNSUPPORTED; }
  const int planes = N * C;
  const int blocks = planes < 4096 ? planes : 4096;
  hipLaunchKernelGGL(gauss_tail_apply_kernel, dim3(blocks), dim3(256), 0, s, raw, target, scale, shift, coef / (sigma * sigma), gscale, A, B, Cc, dy, planes, C, HW);
  note_launch_bytes((double)N * C * HW * 12.0);
  return check_launch("gauss_tail_apply");
}

// dy = A[c]*dout + B[c]*y + C[c] on NCHW f32 planes.  Block = whole planes (n, c).
__global__ __launch_bounds__(256) void bn_bwd_apply_nchw_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                                const float* __restrict__ A, const float* __restrict__ B,
                                                                const float* __restrict__ Cc, float* __restrict__ dy, int planes, int C,
                                                                int HW) {
  for (int p = blockIdx.x; p < planes; p += gridDim.x) {
    const int c = p % C;
    const float ca = A[c], cb = B[c], cc = Cc[c];
    const long base = (long)p * HW;
    if ((HW & 3) == 0) {
      for (int i = threadIdx.x * 4; i < HW; i += 1024) {
        const float4 d = *reinterpret_cast<const float4*>(dout + base + i);
        const float4 v = *reinterpret_cast<const float4*>(y + base + i);
        float4 o;
        o.x = ca * d.x + cb * v.x + cc; o.y = ca * d.y + cb * v.y + cc; o.z = ca * d.z + cb * v.z + cc; o.w = ca * d.w + cb * v.w + cc;
        *reinterpret_cast<float4*>(dy + base + i) = o;
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += 256) dy[base + i] = ca * dout[base + i] + cb * y[base + i] + cc;
    }
  }
}
int launch_bn_bwd_apply_nchw(const float* dout, const float* y, const float* A, const float* B, const float* Cc, float* dy,
                             int N, int C, int HW, hipStream_t s) {
  if (C > 64) { set_error("bn_bwd_apply_nchw: C=%d > 64", C); return MMVAE_ERR_UNSUPPORTED; }
  const int planes = N * C;
  const int blocks = planes < 4096 ? planes : 4096;
  hipLaunchKernelGGL(bn_bwd_apply_nchw_kernel, dim3(blocks), dim3(256), 0, s, dout, y, A, B, Cc, dy, planes, C, HW);
  return check_launch("bn_bwd_apply_nchw");
}

}  // namespace mmvae
