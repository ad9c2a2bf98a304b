// Latent-space and loss kernels: reparameterisation, KL, Gaussian NLL / weighted cross-entropy, RBF-MMD,
// plus the flat Adam step and small conversion helpers.  All scalar reductions accumulate in f64 through one
// atomicAdd per block into a caller-zeroed accumulator.  Every block's contribution is first rounded to a multiple of 2^-16: all
// running sums (< 2^37) are then exactly representable, the f64 additions are exact and the total is the same bits in ANY arrival
// order -- the step stays bit-reproducible without a second reduction pass (rounding: <= 8e-6 per block, 1e-10 of a loss sum).
#include "kernels.hpp"

namespace mmvae {

static int rblocks(long n, int cap = 1024) {
  long b = (n + 1023) / 1024;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

__device__ __forceinline__ double exact_quantum(double s) { return rint(s * 65536.0) * (1.0 / 65536.0); }
__device__ __forceinline__ void block_atomic_add_d(double v, double* out) {
  __shared__ double sred[4];
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sred[wid] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sred[w];
    atomicAdd(out, exact_quantum(s));
  }
}

// ---------------------------------------------------------------- reparameterisation (model.py:148-150)
template <typename T>
__global__ void rsample_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                                   float* __restrict__ enc, T* __restrict__ enc_t, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float e = mu[i] + eps[i] * expf(lv[i] * 0.5f);
    enc[i] = e;
    if (enc_t) Elem<T>::store(enc_t + i, e);
  }
}
int launch_rsample_fwd(int dt, const float* mu, const float* logvar, const float* eps, float* enc_f32, void* enc_t, long n, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  if (dt == DT_F32) hipLaunchKernelGGL((rsample_fwd_kernel<float>), dim3(rblocks(n)), dim3(256), 0, s, mu, logvar, eps, enc_f32, (float*)enc_t, n);
  else hipLaunchKernelGGL((rsample_fwd_kernel<bf16_t>), dim3(rblocks(n)), dim3(256), 0, s, mu, logvar, eps, enc_f32, (bf16_t*)enc_t, n);
  return check_launch("rsample_fwd");
}
__global__ void rsample_bwd_kernel(const float* __restrict__ d_enc, const float* __restrict__ lv, const float* __restrict__ eps,
                                   float* __restrict__ d_mu, float* __restrict__ d_lv, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = d_enc[i];
    d_mu[i] = g;
    d_lv[i] = g * eps[i] * 0.5f * expf(lv[i] * 0.5f);
  }
}
int launch_rsample_bwd(const float* d_enc, const float* logvar, const float* eps, float* d_mu, float* d_logvar, long n, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(rsample_bwd_kernel, dim3(rblocks(n)), dim3(256), 0, s, d_enc, logvar, eps, d_mu, d_logvar, n);
  return check_launch("rsample_bwd");
}

// ---------------------------------------------------------------- KL (model.py:364-365)
__global__ void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, long n, double* out) {
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float l = lv[i], m = mu[i];
    acc += (double)(l - expf(l) - m * m + 1.0f);
  }
  block_atomic_add_d(-0.5 * acc, out);
}
int launch_kl_fwd(const float* mu, const float* logvar, long n, double* out, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(rblocks(n, 256)), dim3(256), 0, s, mu, logvar, n, out);
  return check_launch("kl_fwd");
}
__global__ void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, float coef, const float* __restrict__ gs,
                              float* __restrict__ d_mu, float* __restrict__ d_lv, long n) {
  if (gs) coef *= gs[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    d_mu[i] = coef * mu[i];
    d_lv[i] = coef * 0.5f * (expf(lv[i]) - 1.0f);
  }
}
int launch_kl_bwd(const float* mu, const float* logvar, float coef, const float* gscale, float* d_mu, float* d_logvar, long n, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(rblocks(n)), dim3(256), 0, s, mu, logvar, coef, gscale, d_mu, d_logvar, n);
  return check_launch("kl_bwd");
}

// ---------------------------------------------------------------- Gaussian NLL (model.py:403)
__global__ void gauss_nll_fwd_kernel(const float* __restrict__ r, const float* __restrict__ t, long n, float inv2var, float cst,
                                     double* out) {
  double acc = 0.0;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(r)[i], b = reinterpret_cast<const float4*>(t)[i];
    const float d0 = b.x - a.x, d1 = b.y - a.y, d2 = b.z - a.z, d3 = b.w - a.w;
    acc += (double)((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long i = n4 << 2; i < n; ++i) { const float d = t[i] - r[i]; acc += (double)(d * d); }
  block_atomic_add_d(acc * (double)inv2var, out);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out, exact_quantum((double)cst * (double)n));
}
int launch_gauss_nll_fwd(const float* r, const float* t, long n, float sigma, double* out, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  const float var = sigma * sigma;
  const float cst = logf(sigma) + (float)log(sqrt(2.0 * 3.14159265358979323846));
  hipLaunchKernelGGL(gauss_nll_fwd_kernel, dim3(rblocks(n / 4 + 1)), dim3(256), 0, s, r, t, n, 1.0f / (2.0f * var), cst, out);
  return check_launch("gauss_nll_fwd");
}
__global__ void gauss_nll_bwd_kernel(const float* __restrict__ r, const float* __restrict__ t, long n, float k, const float* __restrict__ gs,
                                     float* __restrict__ d_r) {
  if (gs) k *= gs[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d_r[i] = k * (r[i] - t[i]);
}
int launch_gauss_nll_bwd(const float* r, const float* t, long n, float sigma, float coef, const float* gscale, float* d_r, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(gauss_nll_bwd_kernel, dim3(rblocks(n, 2048)), dim3(256), 0, s, r, t, n, coef / (sigma * sigma), gscale, d_r);
  return check_launch("gauss_nll_bwd");
}

// ---------------------------------------------------------------- weighted cross entropy (model.py:400-401)
template <bool BWD>
__global__ void ce_kernel(const float* __restrict__ r, const long long* __restrict__ t, const float* __restrict__ w, int N, int Q, int HW,
                          float coef, const float* __restrict__ gs, double* out, float* __restrict__ d_r) {
  double acc = 0.0;
  if (BWD && gs) coef *= gs[0];
  const long total = (long)N * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW), p = (int)(i - (long)n * HW);
    const float* rp = r + (long)n * Q * HW + p;
    const int tg = (int)t[i];
    float mx = rp[0];
    for (int q = 1; q < Q; ++q) mx = fmaxf(mx, rp[(long)q * HW]);
    float se = 0.f;
    for (int q = 0; q < Q; ++q) se += expf(rp[(long)q * HW] - mx);
    const float lse = mx + logf(se);
    const float wt = w ? w[tg] : 1.f;
    if (!BWD) {
      acc += (double)(wt * (lse - rp[(long)tg * HW]));
    } else {
      float* dp = d_r + (long)n * Q * HW + p;
      for (int q = 0; q < Q; ++q) {
        const float sm = expf(rp[(long)q * HW] - lse);
        dp[(long)q * HW] = coef * wt * (sm - (q == tg ? 1.f : 0.f));
      }
    }
  }
  if (!BWD) block_atomic_add_d(acc, out);
}
int launch_ce_fwd(const float* r, const long long* t, const float* w, int N, int Q, int HW, double* out, hipStream_t s) {
  if ((long)N * HW <= 0) return MMVAE_OK;
  hipLaunchKernelGGL((ce_kernel<false>), dim3(rblocks((long)N * HW)), dim3(256), 0, s, r, t, w, N, Q, HW, 0.f, (const float*)nullptr, out, (float*)nullptr);
  return check_launch("ce_fwd");
}
int launch_ce_bwd(const float* r, const long long* t, const float* w, int N, int Q, int HW, float coef, const float* gscale, float* d_r,
                  hipStream_t s) {
  if ((long)N * HW <= 0) return MMVAE_OK;
  hipLaunchKernelGGL((ce_kernel<true>), dim3(rblocks((long)N * HW, 2048)), dim3(256), 0, s, r, t, w, N, Q, HW, coef, gscale, (double*)nullptr, d_r);
  return check_launch("ce_bwd");
}

// ---------------------------------------------------------------- RBF MMD (model.py:367-383), never materialises (N,N,d)
// 64x64 pair tile per block, 4x4 pairs per thread, d staged through LDS in chunks of 32.
__global__ __launch_bounds__(256) void mmd_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, int d, double* out) {
  __shared__ float sA[64][33];
  __shared__ float sB[64][33];
  const int which = blockIdx.z;     // 0: xx, 1: yy, 2: xy
  const float* A = which == 1 ? y : x;
  const float* B = which == 0 ? x : y;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int k0 = 0; k0 < d; k0 += 32) {
    __syncthreads();
    for (int v = threadIdx.x; v < 64 * 32; v += 256) {
      const int rr = v >> 5, kk = v & 31;
      sA[rr][kk] = (i0 + rr < n && k0 + kk < d) ? A[(long)(i0 + rr) * d + k0 + kk] : 0.f;
      sB[rr][kk] = (j0 + rr < n && k0 + kk < d) ? B[(long)(j0 + rr) * d + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = sA[ti + 16 * a][kk];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = sB[tj + 16 * b][kk];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { const float df = av[a] - bv[b]; acc[a][b] += df * df; }
    }
  }
  const float inv = 1.0f / ((float)d * (float)d);
  double sum = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (i0 + ti + 16 * a < n && j0 + tj + 16 * b < n) sum += (double)expf(-acc[a][b] * inv);
  block_atomic_add_d(which == 2 ? -2.0 * sum : sum, out);
}
// VAE.compute_kernel (model.py:367-376): the (n, m) matrix k[i][j] = exp(-mean_d((x_i - y_j)^2) / d) itself (a helper of the
// reference surface; the loss never materialises it).  Same 64x64 tile / 4x4 pairs per thread scheme as mmd_fwd_kernel.
__global__ __launch_bounds__(256) void rbf_matrix_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, int m, int d,
                                                         float* __restrict__ out) {
  __shared__ float sA[64][33];
  __shared__ float sB[64][33];
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int k0 = 0; k0 < d; k0 += 32) {
    __syncthreads();
    for (int v = threadIdx.x; v < 64 * 32; v += 256) {
      const int rr = v >> 5, kk = v & 31;
      sA[rr][kk] = (i0 + rr < n && k0 + kk < d) ? x[(long)(i0 + rr) * d + k0 + kk] : 0.f;
      sB[rr][kk] = (j0 + rr < m && k0 + kk < d) ? y[(long)(j0 + rr) * d + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = sA[ti + 16 * a][kk];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = sB[tj + 16 * b][kk];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { const float df = av[a] - bv[b]; acc[a][b] += df * df; }
    }
  }
  const float inv = 1.0f / ((float)d * (float)d);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + ti + 16 * a, j = j0 + tj + 16 * b;
      if (i < n && j < m) out[(long)i * m + j] = expf(-acc[a][b] * inv);
    }
}
int launch_rbf_matrix(const float* x, const float* y, int n, int m, int d, float* out, hipStream_t s) {
  if (n <= 0 || m <= 0 || d <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(rbf_matrix_kernel, dim3((m + 63) / 64, (n + 63) / 64), dim3(256), 0, s, x, y, n, m, d, out);
  return check_launch("rbf_matrix");
}
// MFMA version (exact f32, v_mfma_f32_16x16x4_f32): a 64x64 tile of pairs per block, each wave a 32x32 quadrant as 2x2
// MFMA tiles; S = A B^T accumulated over d in LDS chunks of 32, then k = exp(-(|a|^2 + |b|^2 - 2S)/d^2).
// The symmetric xx / yy sums visit only tiles with j-tile >= i-tile (off-diagonal tiles count twice).
__global__ void row_sqnorm_kernel(const float* __restrict__ x, int n, int d, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  float a = 0.f;
  for (int k = lane; k < d; k += 64) { const float v = x[(long)row * d + k]; a += v * v; }
  a = wave_sum(a);
  if (lane == 0) out[row] = a;
}

// Persistent MFMA kernel.  The work list holds only the 64 x 64 tiles that count: the symmetric xx / yy sums pair tile row p
// with row tiles-1-p (tiles+1 upper-triangle entries per pair), xy takes the full square; a block walks the list with stride
// gridDim.x and issues ONE double atomic at the end (one per tile serialised 12.9 k same-address atomics at N = 5120).
// The 64 x 64 x d inner products run on the bf16 MFMA with each f32 operand split into hi + lo bf16 halves,
//   a.b ~= ah.bh + ah.bl + al.bh      (inputs kept to 16 mantissa bits: |error| <= 2^-15 |a||b|, i.e. < 1e-7 of the exponent
// d2/d^2), 3 x v_mfma_f32_16x16x32_bf16 per 32 k instead of 8 x v_mfma_f32_16x16x4_f32 at a sixteenth of the rate; the squared
// norms stay exact f32 (row_sqnorm_kernel), so k(a, a) = exp(-(2|a|^2 - 2 a.a)/d^2) stays within the same bound of 1.
struct MmdEntry { int which, it, jt; bool valid; };
__device__ __forceinline__ MmdEntry mmd_entry(int e, int tiles, int S) {
  MmdEntry m;
  m.valid = true;
  if (e >= 2 * S) { m.which = 2; e -= 2 * S; m.it = e / tiles; m.jt = e - m.it * tiles; return m; }
  m.which = e >= S ? 1 : 0;
  if (m.which) e -= S;
  const int p = e / (tiles + 1), c = e - p * (tiles + 1);
  if (c < tiles - p) { m.it = p; m.jt = p + c; }
  else { m.it = tiles - 1 - p; m.jt = m.it + (c - (tiles - p)); m.valid = m.it != p; }     // odd tile counts: the middle row pairs with itself
  return m;
}

__global__ __launch_bounds__(256) void mmd_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ nx, const float* __restrict__ ny, int n, int d,
                                                           int tiles, double* out) {
  constexpr int kPitch = 40;                        // bf16 per row: 32 + 8 pad = 80 B, 16-byte fragment reads hit all banks once
  __shared__ __attribute__((aligned(16))) uint16_t sT[4][64 * kPitch];      // A hi, A lo, B hi, B lo
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = lane >> 4, r = lane & 15;
  const int wi = (wv >> 1) * 32, wj = (wv & 1) * 32;   // this wave's 32x32 quadrant
  const float inv = 1.0f / ((float)d * (float)d);
  const float nscale = -inv * 1.44269504088896341f;    // exp(-d2 * inv) = exp2(d2 * nscale)
  const bool vec4 = (d & 3) == 0;
  const int S = ((tiles + 1) / 2) * (tiles + 1);
  const int total = 2 * S + tiles * tiles;
  const int srow = tid >> 3, scol = (tid & 7) * 4;    // staging: rows srow and srow + 32, k columns scol .. scol + 3
  double sum = 0.0;
  for (int e = blockIdx.x; e < total; e += gridDim.x) {
    const MmdEntry m = mmd_entry(e, tiles, S);        // uniform per block
    if (!m.valid) continue;
    const float* A = m.which == 1 ? y : x;
    const float* B = m.which == 0 ? x : y;
    const float* nA = m.which == 1 ? ny : nx;
    const float* nB = m.which == 0 ? nx : ny;
    const int i0 = m.it * 64, j0 = m.jt * 64;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
    for (int k0 = 0; k0 < d; k0 += 32) {
      float va[2][4], vb[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rr = srow + 32 * h, kk = k0 + scol;
        const float* pa = A + (long)(i0 + rr) * d + kk;
        const float* pb = B + (long)(j0 + rr) * d + kk;
        if (vec4) {
          const float4 qa = (kk < d && i0 + rr < n) ? *reinterpret_cast<const float4*>(pa) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 qb = (kk < d && j0 + rr < n) ? *reinterpret_cast<const float4*>(pb) : make_float4(0.f, 0.f, 0.f, 0.f);
          va[h][0] = qa.x; va[h][1] = qa.y; va[h][2] = qa.z; va[h][3] = qa.w;
          vb[h][0] = qb.x; vb[h][1] = qb.y; vb[h][2] = qb.z; vb[h][3] = qb.w;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            va[h][c] = (kk + c < d && i0 + rr < n) ? pa[c] : 0.f;
            vb[h][c] = (kk + c < d && j0 + rr < n) ? pb[c] : 0.f;
          }
        }
      }
      __syncthreads();                               // the previous chunk's fragment reads are done
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float a0 = va[h][2 * c], a1 = va[h][2 * c + 1], b0 = vb[h][2 * c], b1 = vb[h][2 * c + 1];
          ah[c] = pack2_bf16(a0, a1);
          bh[c] = pack2_bf16(b0, b1);
          al[c] = pack2_bf16(a0 - __uint_as_float(ah[c] << 16), a1 - __uint_as_float(ah[c] & 0xffff0000u));
          bl[c] = pack2_bf16(b0 - __uint_as_float(bh[c] << 16), b1 - __uint_as_float(bh[c] & 0xffff0000u));
        }
        const int off = (srow + 32 * h) * kPitch + scol;
        *reinterpret_cast<uint2*>(&sT[0][off]) = make_uint2(ah[0], ah[1]);
        *reinterpret_cast<uint2*>(&sT[1][off]) = make_uint2(al[0], al[1]);
        *reinterpret_cast<uint2*>(&sT[2][off]) = make_uint2(bh[0], bh[1]);
        *reinterpret_cast<uint2*>(&sT[3][off]) = make_uint2(bl[0], bl[1]);
      }
      __syncthreads();
      bf16x8 fa[2][2], fb[2][2];                      // [tile][hi / lo]: row (lane & 15), k = 8 * (lane >> 4) .. + 7
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          fa[t][q] = *reinterpret_cast<const bf16x8*>(&sT[q][(wi + 16 * t + r) * kPitch + 8 * g]);
          fb[t][q] = *reinterpret_cast<const bf16x8*>(&sT[2 + q][(wj + 16 * t + r) * kPitch + 8 * g]);
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);      // small terms first
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
        }
    }
    float tsum = 0.f;                                 // <= 16 terms in (0, 1]: f32 is exact enough before the double accumulator
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = j0 + wj + 16 * b + r;
        const float nbj = j < n ? nB[j] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + wi + 16 * a + 4 * g + q;      // D layout: row = 4*(lane>>4) + reg, col = lane&15
          if (i < n && j < n) tsum += exp2f(fmaxf(nA[i] + nbj - 2.0f * acc[a][b][q], 0.f) * nscale);
        }
      }
    const double w = m.which == 2 ? -2.0 : ((m.jt > m.it) ? 2.0 : 1.0);
    sum += w * (double)tsum;
  }
  block_atomic_add_d(sum, out);
}

int launch_mmd_fwd(const float* x, const float* y, int n, int d, double* out, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  const int tiles = (n + 63) / 64;
  hipLaunchKernelGGL(mmd_fwd_kernel, dim3(tiles, tiles, 3), dim3(256), 0, s, x, y, n, d, out);
  return check_launch("mmd_fwd");
}
// scratch: 2*n floats (row squared norms of x and y)
int launch_mmd_fwd_mfma(const float* x, const float* y, int n, int d, float* scratch, double* out, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  const int tiles = (n + 63) / 64;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n + 3) / 4), dim3(256), 0, s, x, n, d, scratch);
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n + 3) / 4), dim3(256), 0, s, y, n, d, scratch + n);
  const int total = 2 * ((tiles + 1) / 2) * (tiles + 1) + tiles * tiles;
  hipLaunchKernelGGL(mmd_fwd_mfma_kernel, dim3(total < 1024 ? total : 1024), dim3(256), 0, s, x, y, scratch, scratch + n, n, d, tiles, out);
  return check_launch("mmd_fwd_mfma");
}

// d mmd / d y_j = -(4/d^2) sum_i k(y_i,y_j)(y_j - y_i) + (4/d^2) sum_i k(x_i,y_j)(y_j - x_i).  One block per j.
__global__ __launch_bounds__(256) void mmd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, int d, float coef,
                                                      const float* __restrict__ gs, float* __restrict__ d_y) {
  extern __shared__ float sm[];
  if (gs) coef *= gs[0];
  float* syj = sm;            // [d]
  float* wy = sm + d;         // [chunk]
  float* wx = wy + 1024;      // [chunk]
  const int j = blockIdx.x;
  for (int k = threadIdx.x; k < d; k += blockDim.x) syj[k] = y[(long)j * d + k];
  const float inv = 1.0f / ((float)d * (float)d);
  float accv[2] = {0.f, 0.f};   // up to 512 dims with 256 threads
  for (int i0 = 0; i0 < n; i0 += 1024) {
    __syncthreads();
    for (int ii = threadIdx.x; ii < 1024; ii += blockDim.x) {
      const int i = i0 + ii;
      float ky = 0.f, kx = 0.f;
      if (i < n) {
        float dy2 = 0.f, dx2 = 0.f;
        for (int k = 0; k < d; ++k) {
          const float a = y[(long)i * d + k] - syj[k], b = x[(long)i * d + k] - syj[k];
          dy2 += a * a; dx2 += b * b;
        }
        ky = expf(-dy2 * inv); kx = expf(-dx2 * inv);
      }
      wy[ii] = ky; wx[ii] = kx;
    }
    __syncthreads();
    const int cnt = min(1024, n - i0);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = threadIdx.x + 256 * q;
      if (k < d) {
        float a = 0.f;
        const float yj = syj[k];
        for (int ii = 0; ii < cnt; ++ii) {
          const long row = (long)(i0 + ii) * d + k;
          a += -wy[ii] * (yj - y[row]) + wx[ii] * (yj - x[row]);
        }
        accv[q] += a;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = threadIdx.x + 256 * q;
    if (k < d) d_y[(long)j * d + k] += coef * 4.0f * inv * accv[q];
  }
}
int launch_mmd_bwd(const float* x, const float* y, int n, int d, float coef, const float* gscale, float* d_y, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  if (d > 512) { set_error("mmd_bwd: latent dim %d > 512", d); return MMVAE_ERR_UNSUPPORTED; }
  const size_t smb = (size_t)(d + 2048) * sizeof(float);
  hipLaunchKernelGGL(mmd_bwd_kernel, dim3(n), dim3(256), smb, s, x, y, n, d, coef, gscale, d_y);
  return check_launch("mmd_bwd");
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam defaults, main.py:468)
__global__ void adam_kernel(AdamArgs a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    float g = a.g[i] * a.grad_scale;
    float p = a.p[i];
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    float m = a.m[i], v = a.v[i];
    m = m + (g - m) * (1.0f - a.beta1);                 // exp_avg.lerp_(grad, 1-beta1)
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(v) / a.bc2 + a.eps;       // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
    p = p - (a.lr / a.bc1) * (m / denom);               // param.addcdiv_(exp_avg, denom, value=-step_size)
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
  }
}
int launch_adam(const AdamArgs& a, hipStream_t s) {
  if (a.n <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(rblocks(a.n, 2048)), dim3(256), 0, s, a);
  return check_launch("adam");
}
// Graph-capturable form: the step count lives on the device (a captured launch replays its scalar arguments, so host-side bias
// corrections would freeze).  state[0] = step count (as a double), incremented first; bias corrections from it.
__global__ void adam_dev_kernel(AdamArgs a, double* state) {
  __shared__ float sbc[2];
  if (threadIdx.x == 0) {
    const double t = state[0] + 1.0;          // every block computes the same corrections from the not-yet-incremented count
    sbc[0] = (float)(1.0 - pow((double)a.beta1, t));
    sbc[1] = (float)sqrt(1.0 - pow((double)a.beta2, t));
  }
  __syncthreads();
  const float bc1 = sbc[0], bc2 = sbc[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    float g = a.g[i] * a.grad_scale;
    float p = a.p[i];
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    float m = a.m[i], v = a.v[i];
    m = m + (g - m) * (1.0f - a.beta1);
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;
    const float denom = sqrtf(v) / bc2 + a.eps;
    p = p - (a.lr / bc1) * (m / denom);
    a.p[i] = p; a.m[i] = m; a.v[i] = v;
  }
}
__global__ void adam_tick_kernel(double* state) { state[0] += 1.0; }
int launch_adam_dev(const AdamArgs& a, double* state, hipStream_t s) {
  if (a.n <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(rblocks(a.n, 2048)), dim3(256), 0, s, a, state);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, state);      // after every block has read the old count
  return check_launch("adam_dev");
}

// ---------------------------------------------------------------- input normalisation (main.py:383-387)
// Four labels per thread (32-byte / 4-byte loads, 16-byte f32 store, 8-byte bf16 store); TL = int64 (what the reference's loader yields)
// or uint8 (a label transport an eighth of the size); both outputs optional.
template <typename T, typename TL>
__global__ void normalise_kernel(const TL* __restrict__ labels, long n, long nq, float mean, float stdv, T* __restrict__ img, float* __restrict__ img32) {
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    float v[4];
    if constexpr (sizeof(TL) == 8) {
      const longlong2 a = reinterpret_cast<const longlong2*>(labels)[2 * q], b = reinterpret_cast<const longlong2*>(labels)[2 * q + 1];
      v[0] = (float)a.x; v[1] = (float)a.y; v[2] = (float)b.x; v[3] = (float)b.y;
    } else {
      const uchar4 a = reinterpret_cast<const uchar4*>(labels)[q];
      v[0] = (float)a.x; v[1] = (float)a.y; v[2] = (float)a.z; v[3] = (float)a.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (v[j] - mean) / stdv;
    if (img32) reinterpret_cast<float4*>(img32)[q] = make_float4(v[0], v[1], v[2], v[3]);
    if (img) {
      if constexpr (sizeof(T) == 4) reinterpret_cast<float4*>(img)[q] = make_float4(v[0], v[1], v[2], v[3]);
      else reinterpret_cast<uint2*>(img)[q] = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
    }
  }
  // tail (n not a multiple of 4)
  for (long i = (nq << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = ((float)labels[i] - mean) / stdv;
    if (img) Elem<T>::store(img + i, v);
    if (img32) img32[i] = v;
  }
}
int launch_normalise(int dt, const void* labels, int label_bytes, long n, float mean, float stdv, void* img_t, float* img_f32, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  if (label_bytes != 8 && label_bytes != 1) { set_error("normalise: labels must be int64 or uint8"); return MMVAE_ERR_ARG; }
  // the vector path needs 16-byte (int64) / 4-byte (uint8) aligned labels and 16- / 8-byte aligned outputs: a contiguous but misaligned
  // view (labels[1:], an odd uint8 offset) takes the scalar loop for the whole range
  const bool aligned = reinterpret_cast<uintptr_t>(labels) % (label_bytes == 8 ? 16 : 4) == 0 && reinterpret_cast<uintptr_t>(img_f32) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(img_t) % (dt == DT_F32 ? 16 : 8) == 0;
  const long nq = aligned ? n >> 2 : 0;
  const dim3 g(rblocks(aligned ? (n >> 2) + 1 : n, 2048)), b(256);
  if (label_bytes == 8) {
    if (dt == DT_F32) hipLaunchKernelGGL((normalise_kernel<float, long long>), g, b, 0, s, (const long long*)labels, n, nq, mean, stdv, (float*)img_t, img_f32);
    else hipLaunchKernelGGL((normalise_kernel<bf16_t, long long>), g, b, 0, s, (const long long*)labels, n, nq, mean, stdv, (bf16_t*)img_t, img_f32);
  } else {
    if (dt == DT_F32) hipLaunchKernelGGL((normalise_kernel<float, unsigned char>), g, b, 0, s, (const unsigned char*)labels, n, nq, mean, stdv, (float*)img_t, img_f32);
    else hipLaunchKernelGGL((normalise_kernel<bf16_t, unsigned char>), g, b, 0, s, (const unsigned char*)labels, n, nq, mean, stdv, (bf16_t*)img_t, img_f32);
  }
  return check_launch("normalise");
}

// k-means quantiser of the input pipeline (main.py:21-38: ToTensor -> kmeans.predict per pixel; utils.py:279-309) fused with
// the normalisation of main.py:383-387: label = argmin_k (x/255 - centre_k)^2 (lowest index wins ties), image = (label-mean)/std
__global__ void quantise_normalise_kernel(const unsigned char* __restrict__ frames, long n, const float* __restrict__ centres, int q,
                                          float mean, float stdv, long long* __restrict__ labels, float* __restrict__ image) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = (float)frames[i] / 255.0f;
    int best = 0;
    float bd = (x - centres[0]) * (x - centres[0]);
    for (int k = 1; k < q; ++k) { const float d = (x - centres[k]) * (x - centres[k]); if (d < bd) { bd = d; best = k; } }
    if (labels) labels[i] = best;
    if (image) image[i] = ((float)best - mean) / stdv;
  }
}
int launch_quantise_normalise(const unsigned char* frames, long n, const float* centres, int q, float mean, float stdv,
                              long long* labels, float* image, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  if (q < 1 || q > 256) { set_error("quantise: q=%d out of range", q); return MMVAE_ERR_ARG; }
  hipLaunchKernelGGL(quantise_normalise_kernel, dim3(rblocks(n, 2048)), dim3(256), 0, s, frames, n, centres, q, mean, stdv, labels, image);
  return check_launch("quantise_normalise");
}

template <typename TI, typename TOo>
__global__ void convert_kernel(const TI* __restrict__ in, TOo* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    Elem<TOo>::store(out + i, Elem<TI>::load(in + i));
}
int launch_convert(int dt_in, int dt_out, const void* in, void* out, long n, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  const dim3 g(rblocks(n, 2048)), b(256);
  if (dt_in == DT_F32 && dt_out == DT_F32) hipLaunchKernelGGL((convert_kernel<float, float>), g, b, 0, s, (const float*)in, (float*)out, n);
  else if (dt_in == DT_F32) hipLaunchKernelGGL((convert_kernel<float, bf16_t>), g, b, 0, s, (const float*)in, (bf16_t*)out, n);
  else if (dt_out == DT_F32) hipLaunchKernelGGL((convert_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)in, (float*)out, n);
  else hipLaunchKernelGGL((convert_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)in, (bf16_t*)out, n);
  return check_launch("convert");
}

// out[r][0..ca) = a[r][:], out[r][ca..ca+cb) = b[r][:]   (f32 -> T); b may be null (cb = 0)
__global__ void fill_f32_kernel(float* p, float v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int launch_fill_f32(float* p, float v, long n, hipStream_t s) {
  if (n <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n);
  return check_launch("fill_f32");
}

template <typename T>
__global__ void concat2_kernel(const float* __restrict__ a, const float* __restrict__ b, int rows, int ca, int cb, T* __restrict__ out) {
  const long total = (long)rows * (ca + cb);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / (ca + cb)), c = (int)(i - (long)r * (ca + cb));
    const float v = c < ca ? a[(long)r * ca + c] : b[(long)r * cb + (c - ca)];
    Elem<T>::store(out + i, v);
  }
}
int launch_concat2_to_t(int dt, const float* a, const float* b, int rows, int ca, int cb, void* out, hipStream_t s) {
  const long total = (long)rows * (ca + cb);
  if (total <= 0) return MMVAE_OK;
  if (dt == DT_F32) hipLaunchKernelGGL((concat2_kernel<float>), dim3(rblocks(total)), dim3(256), 0, s, a, b, rows, ca, cb, (float*)out);
  else hipLaunchKernelGGL((concat2_kernel<bf16_t>), dim3(rblocks(total)), dim3(256), 0, s, a, b, rows, ca, cb, (bf16_t*)out);
  return check_launch("concat2");
}

// out = { (nll*px + klc*kl + mmdc*mmd)/N, nll*px/N, kl/N, mmd/N }   (model.py:405-406); acc = {px, kl, mmd}
__global__ void loss_finish_kernel(const double* acc, float* out, float nll, float klc, float mmdc, float n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double px = (double)nll * acc[0], kl = acc[1], mmd = acc[2];
    out[0] = (float)((px + (double)klc * kl + (double)mmdc * mmd) / (double)n);
    out[1] = (float)(px / (double)n);
    out[2] = (float)(kl / (double)n);
    out[3] = (float)(mmd / (double)n);
  }
}
int launch_loss_finish(const double* acc, float* out, float nll, float klc, float mmdc, float n, hipStream_t s) {
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, acc, out, nll, klc, mmdc, n);
  return check_launch("loss_finish");
}

__global__ void partials_add_kernel(const float* partials, int nparts, int row_stride, int C, float* out) {
  const int c = blockIdx.x;
  float v = 0.f;
  for (int p = threadIdx.x; p < nparts; p += blockDim.x) v += partials[(long)p * row_stride + c];
  v = wave_sum(v);
  if (threadIdx.x == 0) out[c] += v;
}
int launch_partials_add(const float* partials, int nparts, int row_stride, int C, float* out, hipStream_t s) {
  if (C <= 0) return MMVAE_OK;
  hipLaunchKernelGGL(partials_add_kernel, dim3(C), dim3(64), 0, s, partials, nparts, row_stride, C, out);
  return check_launch("partials_add");
}


}  // namespace mmvae
