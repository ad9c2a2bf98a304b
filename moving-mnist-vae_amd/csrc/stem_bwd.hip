// Backward of the stem  Conv2d(1 -> 32, k5 s2 p2) -> BatchNorm -> ReLU  (model.py:94-96,115-117) in ONE pass over the
// incoming gradient g and the stored pre-BatchNorm output y0 (plus the 1-channel image).
//
// The stem needs no data gradient, only dW / dgamma / dbeta, and BatchNorm-backward is affine in its sums:
//     dy = A * (gm - m1 - m2 * yhat),   gm = g * [y0*s + b > 0],  yhat = (y0 - mean) * istd,
//     m1 = sum(gm) / count,  m2 = sum(gm * yhat) / count,  A = gamma * istd
// so   dW[c][t] = sum_pix dy[c] * patch[t] = A[c] * (W1[c][t] - m2[c] * W2[c][t] - m1[c] * W3[t])
// with W1 = sum gm (x) patch,  W2 = sum yhat (x) patch,  W3 = sum patch -- three pixel reductions that do NOT depend on
// m1 / m2.  One kernel therefore produces the BatchNorm sums AND the three images (one MFMA reduction with 80 "channels":
// gm | yhat | ones, against the 25 taps of the 5x5 patch gathered on the fly from an LDS copy of the image rows), and a
// finalize kernel combines them.  Replaces bn_bwd_reduce + bn_bwd_apply + im2col + wgrad: 2.7 GB -> 0.7 GB of HBM
// traffic per step at N = 5120, and nothing of it waits on a grid-wide reduction.
//
// Tile = rpt whole output rows of one image (<= 128 pixels); wave w owns the 32-pixel k-slice w of the tile.
// LDS: [P tile 128 x 80 T][image patch prow x (W+4) f32]; the flush staging aliases the P tile.
#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

constexpr int kStemPA = 80;                        // P-tile channels: gm 0..31 | yhat 32..63 | ones 64 | zero 65..79
constexpr int kStemPartFloats = 64 + kStemPA * 32; // per-block partial: S0[32] S1[32] | image [80][32]

struct StemBwdArgs {
  const void* g; const void* y0; const void* x;
  const float* ms; const float* mb; const float* mean; const float* istd;
  float* partials;
  int N, H, W, Ho, Wo, rpt, tiles_per_img, ntiles, prow, pw;
};

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_bwd_kernel(StemBwdArgs a) {
  constexpr int VE = Elem<T>::kVec, ES = sizeof(T);
  constexpr int CV = 32 / VE;                      // 16-byte vectors per pixel of g / y0
  constexpr int NS = 128 * CV / 256;               // staging slots per thread and tensor
  constexpr int PPS = 256 / CV;                    // pixels per slot round
  constexpr int PITCH = kStemPA * ES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sP = smem;
  float* sX = reinterpret_cast<float*>(sP + 128 * PITCH);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, r = lane & 15;
  const T* __restrict__ G = reinterpret_cast<const T*>(a.g);
  const T* __restrict__ Y = reinterpret_cast<const T*>(a.y0);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const int cv = t % CV, pp0 = t / CV;
  float ms[VE], mb[VE], mean[VE], istd[VE], s0[VE], s1[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    ms[j] = a.ms[cv * VE + j]; mb[j] = a.mb[cv * VE + j]; mean[j] = a.mean[cv * VE + j]; istd[j] = a.istd[cv * VE + j];
    s0[j] = 0.f; s1[j] = 0.f;
  }
  // image slots (tile-invariant): patch element idx = t + 256k -> (row, col)
  const int nx = a.prow * a.W;
  int xo[3], xg[3], xr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int idx = t + 256 * k;
    const int row = idx / a.W, col = idx - row * a.W;
    xo[k] = row * a.pw + col + 2; xg[k] = row * a.W + col; xr[k] = idx < nx ? row : (1 << 28);
  }
  // zero the P tile (its padding channels stay zero) and the image patch (its halo columns stay zero)
  for (int i = t; i < 128 * PITCH / 16; i += 256) reinterpret_cast<Vec16*>(sP)[i] = Vec16{{0, 0, 0, 0}};
  for (int i = t; i < a.prow * a.pw; i += 256) sX[i] = 0.f;
  // fragment addressing of this lane (tile-invariant)
  const int tile_pix = a.rpt * a.Wo;
  int tapo[2];                                      // patch offset of tap 16tb + r, or -1
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) { const int tap = 16 * tb + r; tapo[tb] = tap < 25 ? (tap / 5) * a.pw + tap % 5 : -1; }
  int offP[2], offX[8];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) offP[b] = (32 * wv + 16 * b + 4 * gq + (r >> 2)) * PITCH + (r & 3) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = 32 * wv + 16 * (j >> 2) + 4 * gq + (j & 3);
      offX[j] = q < tile_pix ? (2 * (q / a.Wo)) * a.pw + 2 * (q % a.Wo) : 0;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = 32 * wv + 4 * j + gq;
      offX[j] = q < tile_pix ? (2 * (q / a.Wo)) * a.pw + 2 * (q % a.Wo) : 0;
    }
    offP[0] = offP[1] = 0;
  }
  f32x4 acc[5][2];
#pragma unroll
  for (int ta = 0; ta < 5; ++ta) { acc[ta][0] = (f32x4){0, 0, 0, 0}; acc[ta][1] = (f32x4){0, 0, 0, 0}; }

  // ---- software pipeline: registers of the NEXT tile
  Vec16 gv[NS], yv[NS];
  float xv[3];
  int nvalid_c = 0, r0_c = 0;
  auto issue = [&](int tile) {
    const int n = tile / a.tiles_per_img, h0 = (tile - n * a.tiles_per_img) * a.rpt;
    nvalid_c = min(a.rpt, a.Ho - h0) * a.Wo;
    r0_c = 2 * h0 - 2;
    const long base = (((long)n * a.Ho + h0) * a.Wo) * 32 + cv * VE;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int p = pp0 + k * PPS;
      gv[k] = Vec16{{0, 0, 0, 0}}; yv[k] = Vec16{{0, 0, 0, 0}};
      if (p < nvalid_c) {
        gv[k] = *reinterpret_cast<const Vec16*>(G + base + (long)p * 32);
        yv[k] = *reinterpret_cast<const Vec16*>(Y + base + (long)p * 32);
      }
    }
    const T* xi = X + ((long)n * a.H + r0_c) * a.W;       // may point before the image: only in-range rows are read
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      xv[k] = 0.f;
      if ((unsigned)(r0_c + xr[k]) < (unsigned)a.H) xv[k] = Elem<T>::load(xi + xg[k]);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int p = pp0 + k * PPS;
      float fg[VE], fy[VE];
      Elem<T>::unpack(gv[k], fg);
      Elem<T>::unpack(yv[k], fy);
      const bool ok = p < nvalid_c;
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const float gm = (ok && fy[j] * ms[j] + mb[j] > 0.f) ? fg[j] : 0.f;
        s0[j] += gm;
        s1[j] += gm * fy[j];
        fg[j] = gm;
        fy[j] = ok ? (fy[j] - mean[j]) * istd[j] : 0.f;
      }
      *reinterpret_cast<Vec16*>(sP + p * PITCH + cv * 16) = Elem<T>::pack(fg);
      *reinterpret_cast<Vec16*>(sP + p * PITCH + 32 * ES + cv * 16) = Elem<T>::pack(fy);
      if (cv == 0) {
        float one[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) one[j] = 0.f;
        one[0] = ok ? 1.f : 0.f;
        *reinterpret_cast<Vec16*>(sP + p * PITCH + 64 * ES) = Elem<T>::pack(one);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (xr[k] < (1 << 28)) sX[xo[k]] = xv[k];
  };

#pragma unroll
  for (int j = 0; j < VE; ++j) { asm volatile("" ::"v"(ms[j])); asm volatile("" ::"v"(mb[j])); asm volatile("" ::"v"(mean[j])); asm volatile("" ::"v"(istd[j])); }
  int tile = blockIdx.x;
  if (tile < a.ntiles) issue(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();                                // the previous tile's fragment reads are done (first: the zero fill)
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);
    if constexpr (sizeof(T) == 2) {
      Vec16 bf[2];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = tapo[tb] >= 0 ? sX[offX[j] + tapo[tb]] : 0.f;
        bf[tb] = Elem<bf16_t>::pack(f);
      }
#pragma unroll
      for (int ta = 0; ta < 5; ++ta) {
        const Vec16 af = FragOps<bf16_t>::load(sP, offP[0] + ta * 32, offP[1] + ta * 32);
        acc[ta][0] = mma_bf16(af, bf[0], acc[ta][0]);
        acc[ta][1] = mma_bf16(af, bf[1], acc[ta][1]);
      }
    } else {
      // exact-f32 mode: 8 steps of v_mfma_f32_16x16x4_f32, step j covers pixels 32wv + 4j + gq
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = 32 * wv + 4 * j + gq;
        float bvv[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) bvv[tb] = tapo[tb] >= 0 ? sX[offX[j] + tapo[tb]] : 0.f;
#pragma unroll
        for (int ta = 0; ta < 5; ++ta) {
          const float av = *reinterpret_cast<const float*>(sP + p * PITCH + (16 * ta + r) * 4);
          acc[ta][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bvv[0], acc[ta][0], 0, 0, 0);
          acc[ta][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bvv[1], acc[ta][1], 0, 0, 0);
        }
      }
    }
  }
  // ---- flush: fixed summation order (wave 0, 1, 2, 3) -> bit-reproducible partials
  // BatchNorm sums: lanes of a wave with equal channel vector first (xor shuffles), then the waves in order
#pragma unroll
  for (int j = 0; j < VE; ++j) {
#pragma unroll
    for (int o = CV; o < 64; o <<= 1) { s0[j] += __shfl_xor(s0[j], o, 64); s1[j] += __shfl_xor(s1[j], o, 64); }
  }
  __syncthreads();
  float* sRed = reinterpret_cast<float*>(smem);
  for (int i = t; i < kStemPartFloats; i += 256) sRed[i] = 0.f;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wv == w) {
      if (lane < CV) {
#pragma unroll
        for (int j = 0; j < VE; ++j) { sRed[lane * VE + j] += s0[j]; sRed[32 + lane * VE + j] += s1[j]; }
      }
#pragma unroll
      for (int ta = 0; ta < 5; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) sRed[64 + (16 * ta + 4 * gq + jj) * 32 + 16 * tb + r] += acc[ta][tb][jj];
    }
  }
  __syncthreads();
  float* dst = a.partials + (long)blockIdx.x * kStemPartFloats;
  for (int i = t; i < kStemPartFloats; i += 256) dst[i] = sRed[i];
}

// ---------------------------------------------------------------- finalize: one block per stem channel
struct StemBwdFinArgs {
  const float* partials; int nparts;
  const float* global_sums;      // SyncBN: [2][32] sums over the global batch (coefficients); NULL: the local sums
  double count;                  // pixels behind the sums the coefficients use
  const float* gamma; const float* mean; const float* istd;
  float* dgamma; float* dbeta; float* dW;       // dW [32][25]  (+=)
};

__global__ __launch_bounds__(256) void stem_bwd_finalize_kernel(StemBwdFinArgs a) {
  __shared__ double sB[2][4];
  __shared__ double sW[3][8][32];
  const int c = blockIdx.x, t = threadIdx.x;
  double b0 = 0.0, b1 = 0.0;
  for (int p = t; p < a.nparts; p += 256) {
    const float* row = a.partials + (long)p * kStemPartFloats;
    b0 += (double)row[c]; b1 += (double)row[32 + c];
  }
  b0 = wave_sum_d(b0); b1 = wave_sum_d(b1);
  if ((t & 63) == 0) { sB[0][t >> 6] = b0; sB[1][t >> 6] = b1; }
  const int pl = t >> 5, tap = t & 31;
  double w1 = 0.0, w2 = 0.0, w3 = 0.0;
  for (int p = pl; p < a.nparts; p += 8) {
    const float* img = a.partials + (long)p * kStemPartFloats + 64;
    w1 += (double)img[c * 32 + tap]; w2 += (double)img[(32 + c) * 32 + tap]; w3 += (double)img[64 * 32 + tap];
  }
  sW[0][pl][tap] = w1; sW[1][pl][tap] = w2; sW[2][pl][tap] = w3;
  __syncthreads();
  const double S0 = (sB[0][0] + sB[0][1]) + (sB[0][2] + sB[0][3]), S1 = (sB[1][0] + sB[1][1]) + (sB[1][2] + sB[1][3]);
  const double mean = a.mean[c], istd = a.istd[c], g = a.gamma ? a.gamma[c] : 1.0;
  const double sgy_local = istd * (S1 - mean * S0);
  if (t == 0) {
    if (a.dgamma) a.dgamma[c] += (float)sgy_local;
    if (a.dbeta) a.dbeta[c] += (float)S0;
  }
  double G0 = S0, sgy = sgy_local;
  if (a.global_sums) { G0 = a.global_sums[c]; sgy = istd * ((double)a.global_sums[32 + c] - mean * G0); }
  const double m1 = G0 / a.count, m2 = sgy / a.count, A = g * istd;
  if (t < 25) {
    double v1 = 0.0, v2 = 0.0, v3 = 0.0;
    for (int q = 0; q < 8; ++q) { v1 += sW[0][q][t]; v2 += sW[1][q][t]; v3 += sW[2][q][t]; }
    a.dW[c * 25 + t] += (float)(A * (v1 - m2 * v2 - m1 * v3));
  }
}

bool stem_bwd_fusable(int S) { return S >= 9 && S <= 64; }

int stem_bwd_part_floats() { return kStemPartFloats; }

// Returns the number of partial rows (> 0) or an error.  partials: rows * stem_bwd_part_floats() floats.
int launch_stem_bwd(int dt, const void* g, const void* y0, const void* x, const float* ms, const float* mb, const float* mean,
                    const float* istd, float* partials, long partials_cap_floats, int N, int S, int Ho, int Wo, hipStream_t s) {
  if (!stem_bwd_fusable(S) || Wo > 128 || Wo < 1) { set_error("stem_bwd: image size %d unsupported", S); return MMVAE_ERR_UNSUPPORTED; }
  StemBwdArgs a;
  a.g = g; a.y0 = y0; a.x = x; a.ms = ms; a.mb = mb; a.mean = mean; a.istd = istd; a.partials = partials;
  a.N = N; a.H = S; a.W = S; a.Ho = Ho; a.Wo = Wo;
  a.rpt = 128 / Wo; if (a.rpt > Ho) a.rpt = Ho; if (a.rpt < 1) a.rpt = 1;
  a.tiles_per_img = (Ho + a.rpt - 1) / a.rpt;
  a.ntiles = N * a.tiles_per_img;
  a.prow = 2 * a.rpt + 3; a.pw = S + 4;
  if (a.prow * S > 768) { set_error("stem_bwd: patch of %d x %d exceeds the staging slots", a.prow, S); return MMVAE_ERR_UNSUPPORTED; }
  int gx = 512;
  if (gx > a.ntiles) gx = a.ntiles;
  if ((long)gx * kStemPartFloats > partials_cap_floats) gx = (int)(partials_cap_floats / kStemPartFloats);
  if (gx < 1) { set_error("stem_bwd: partials buffer too small"); return MMVAE_ERR_WORKSPACE; }
  const size_t es = dtype_size(dt);
  const size_t lds = (size_t)128 * kStemPA * es + (size_t)a.prow * a.pw * 4;
  const size_t need = lds > (size_t)kStemPartFloats * 4 ? lds : (size_t)kStemPartFloats * 4;
  if (dt == DT_F32) hipLaunchKernelGGL((stem_bwd_kernel<float>), dim3(gx), dim3(256), need, s, a);
  else hipLaunchKernelGGL((stem_bwd_kernel<bf16_t>), dim3(gx), dim3(256), need, s, a);
  const int rc = check_launch("stem_bwd");
  return rc ? rc : gx;
}

int launch_stem_bwd_finalize(const float* partials, int nparts, const float* global_sums, double count, const float* gamma, const float* mean,
                             const float* istd, float* dgamma, float* dbeta, float* dW, hipStream_t s) {
  StemBwdFinArgs a{partials, nparts, global_sums, count, gamma, mean, istd, dgamma, dbeta, dW};
  hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3(32), dim3(256), 0, s, a);
  return check_launch("stem_bwd_finalize");
}

}  // namespace mmvae
