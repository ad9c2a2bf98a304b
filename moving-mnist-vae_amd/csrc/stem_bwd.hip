// Backward of the stem  Conv2d(1 -> 32, k5 s2 p2) -> BatchNorm -> ReLU  (model.py:94-96,115-117) in ONE pass over the
// incoming gradient g and the stored pre-BatchNorm output y0 (plus the 1-channel image).
//
// The stem needs no data gradient, only dW / dgamma / dbeta, and BatchNorm-backward is affine in its sums:
//     dy = A * (gm - m1 - m2 * yhat),   gm = g * [y0*s + b > 0],  yhat = (y0 - mean) * istd,
//     m1 = sum(gm) / count,  m2 = sum(gm * yhat) / count,  A = gamma * istd
// so   dW[c][t] = sum_pix dy[c] * patch[t] = A[c] * (W1[c][t] - m2[c] * istd[c] * (W2[c][t] - mean[c] * W3[t]) - m1[c] * W3[t])
// with W1 = sum gm (x) patch,  W2 = sum y0 (x) patch,  W3 = sum patch -- three pixel reductions that do NOT depend on
// m1 / m2.  W2 and W3 do not even depend on the gradient: y0 = w * patch, so W2 = w x R with R = sum patch (x) patch, the
// 25 x 25 gram matrix of the batch's patches (W3 is its ones row) -- a 42 MB read of the image, off the critical path
// ("gram" mode of the kernel, on the side stream).  On the critical path one pass over g and y0 ("grad" mode) produces the
// BatchNorm sums and W1 (MFMA reduction over pixels against the 25 taps gathered on the fly from an LDS copy of the image
// rows); a 32-block finalize combines everything in double precision.  Replaces bn_bwd_reduce + bn_bwd_apply + im2col + wgrad: 2.7 GB -> 0.7 GB of HBM
// traffic per step at N = 5120, and nothing of it waits on a grid-wide reduction.
//
// Work unit = a 32-pixel slab (whole output rows of one image) owned by ONE wave: private LDS, no block barrier in the loop.
// LDS per wave: [P slab 32 x 32 T][image patch prow x (W+8) f32]; the flush staging aliases the front.
#include <stdlib.h>
#include <string.h>

#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

// MODE 0 ("grad"): P tile = gm (32 channels) built from g and y0; also the BatchNorm sums S0 = sum gm, S1 = sum gm * y0.
// MODE 1 ("gram"): P tile = the im2col of the image itself (taps 0..24, a ones column at 25, zeros beyond): R = sum patch (x) patch
//                  (row 25 = sum patch).  Reads only the image; runs off the critical path.
constexpr int kStemPartFloats = 64 + 32 * 32;      // per-block partial: S0[32] S1[32] | image [32][32]

struct StemBwdArgs {
  // DG (round 4): g is not read -- it is recomputed slab by slab as the data gradient of encoder.layer1's conv1 (3x3 s2 p1, 32 -> 32) plus its
  // 1x1 s2 shortcut from their dy tensors (dy1, dys: [N][Ho/2][Wo/2][32]) and the packed transposed-form weights (op_pack_up order)
  const void* dy1; const void* dys; const void* wd1; const void* wds;
  const void* g; const void* y0; const void* x;
  const float* ms; const float* mb;
  float* partials;
  int N, H, W, Ho, Wo, wshift, rps, slabs_per_img, nslabs, prow, pw;
};

// Every WAVE streams its own 32-pixel slabs (rps whole output rows of one image) through wave-private LDS: no block barrier in
// the loop, 16 independent software pipelines per CU.  Lane roles: staging = 16-byte vector `lane + 64k` of the slab (contiguous
// in memory), fragments = the usual 16x16x32 / 16x16x4 maps over the slab's 32 pixels.
template <typename T, int MODE, bool DG = false>
__global__ __launch_bounds__(256, DG ? 3 : sizeof(T) == 2 ? 4 : 2) void stem_bwd_kernel(StemBwdArgs a) {
  static_assert(!DG || (MODE == 0 && sizeof(T) == 2), "fused layer1 data gradient: bf16 grad mode");
  constexpr int VE = Elem<T>::kVec, ES = sizeof(T);
  constexpr int CV = 32 / VE;                      // 16-byte vectors per pixel of g / y0 / the P slab
  constexpr int NS = MODE == 0 ? 32 * CV / 64 : 1; // g / y0 vectors per lane and slab
  constexpr int NXV = sizeof(T) == 2 ? 1 : 2;      // image vectors per lane and slab
  constexpr int PITCH = 32 * ES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  float* sC = reinterpret_cast<float*>(smem);                          // [2][32] mask coefficients (block-shared, read-only)
  constexpr int kDyPitch = 96;                                         // bytes per staged dy pixel: 64 + 32 (conflict-free ds_read_b128 of 16 pixels)
  constexpr int kDyRow = 17 * kDyPitch;                                // 16 pixels + one zero pixel on the right
  const int wave_bytes = 32 * PITCH + a.prow * a.pw * 4 + (DG ? 2 * kDyRow : 0);
  char* sP = smem + 256 + wv * wave_bytes;                             // this wave's P slab [32][32] T
  float* sX = reinterpret_cast<float*>(sP + 32 * PITCH);               // this wave's image patch [prow][pw] f32, 4 halo columns a side
  char* sDy = reinterpret_cast<char*>(sX) + a.prow * a.pw * 4;         // DG: two dy rows [17][96 B]
  char* sWf = smem + 256 + 4 * wave_bytes;                             // DG: 20 A fragments of 1 KB: [9 conv1 taps + shortcut][2 ci halves][lane]
  const T* __restrict__ G = reinterpret_cast<const T*>(a.g);
  const T* __restrict__ Y = reinterpret_cast<const T*>(a.y0);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  if (MODE == 0 && t < 64) sC[t] = t < 32 ? a.ms[t] : a.mb[t - 32];
  for (int i = lane; i < a.prow * a.pw; i += 64) sX[i] = 0.f;          // the halo columns stay zero
  if constexpr (DG) {
    // A fragment (tap slot f, ci half m): lane (gq, r) = row ci 16 m + r, k = co 8 gq ..: out of op_pack_up's row-major [ci][tap][co] matrices of
    // the four stride phases (taps per phase 1, 2, 2, 4; element offsets 0, 1024, 3072, 5120)
    for (int i = t; i < 20 * 64; i += 256) {
      const int fm = i >> 6, l = i & 63, f = fm >> 1, m = fm & 1, rr = l & 15, g4 = l >> 4;
      const T* src;
      if (f < 9) {
        const int pbase = f < 1 ? 0 : f < 3 ? 1024 : f < 5 ? 3072 : 5120, ntp = f < 1 ? 1 : f < 5 ? 2 : 4, tp = f < 1 ? 0 : f < 3 ? f - 1 : f < 5 ? f - 3 : f - 5;
        src = reinterpret_cast<const T*>(a.wd1) + pbase + ((16 * m + rr) * ntp + tp) * 32 + 8 * g4;
      } else {
        src = reinterpret_cast<const T*>(a.wds) + (16 * m + rr) * 32 + 8 * g4;
      }
      *reinterpret_cast<Vec16*>(sWf + i * 16) = *reinterpret_cast<const Vec16*>(src);
    }
    if (lane < 12) {                                                   // the zero pixel (16) of both rows
      const int row = lane / 6, v = lane % 6;
      *reinterpret_cast<Vec16*>(sDy + row * kDyRow + 16 * kDyPitch + v * 16) = Vec16{{0, 0, 0, 0}};
    }
  }
  __syncthreads();
  const int cv = lane % CV;
  float s1[VE];                                     // sum gm * y0 (sum gm comes out of the MFMA: ones column of B)
#pragma unroll
  for (int j = 0; j < VE; ++j) s1[j] = 0.f;
  // image vector slots: vector idx = lane + 64k of the patch rows -> (row, column vector)
  const int xvr = a.W / VE;                         // vectors per image row
  int xrow[NXV], xcol[NXV];
#pragma unroll
  for (int k = 0; k < NXV; ++k) {
    const int idx = lane + 64 * k;
    xrow[k] = idx < a.prow * xvr ? idx / xvr : (1 << 28);
    xcol[k] = (idx % xvr) * VE;
  }
  // fragment addressing (slab-invariant)
  int tapo[2];                                      // patch offset of tap 16tb + r, or -1 (tap 25: the ones column -> sum of the P rows)
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) { const int tap = 16 * tb + r; tapo[tb] = tap < 25 ? (tap / 5) * a.pw + tap % 5 : -1; }
  const float b_fill = r == 9 ? 1.f : 0.f;          // value of B column 16 + r outside the 25 taps (tb = 1 only)
  auto pix_off = [&](int q) { return ((q >> a.wshift) * 2) * a.pw + (q & (a.Wo - 1)) * 2 + 2; };   // tap (0,0) of slab pixel q
  int offP[2], offXb[2];
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) { offP[b] = (16 * b + 4 * gq + (r >> 2)) * PITCH + (r & 3) * 8; offXb[b] = pix_off(16 * b + 4 * gq); }
  } else { offP[0] = offP[1] = 0; offXb[0] = offXb[1] = 0; }
  // MODE 1: this lane builds half a row of the im2col slab: pixel lane >> 1, taps 16 * (lane & 1) ..
  const int bq = lane >> 1, bh = lane & 1;
  const int boff = pix_off(bq);
  f32x4 acc[2][2];
#pragma unroll
  for (int ta = 0; ta < 2; ++ta) { acc[ta][0] = (f32x4){0, 0, 0, 0}; acc[ta][1] = (f32x4){0, 0, 0, 0}; }

  // ---- software pipeline of this wave: registers of the NEXT slab
  Vec16 gv[NS], yv[NS], xv[NXV];
  // DG: the two dy rows of the next slab (16 pixels x 32 channels = one vector per lane).  Native vectors: as `Vec16` (a struct around an array)
  // these copy-only rows stayed in scratch memory (32 bytes per lane: load, wait, scratch store, reload -- no prefetch)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 dv[2];
  int r0_c = 0, ph_n = 0;
  auto issue = [&](int slab) {
    const int n = slab / a.slabs_per_img, h0 = (slab - n * a.slabs_per_img) * a.rps;
    r0_c = 2 * h0 - 2;
    if constexpr (DG) {
      // output row h0 of the 32 x 32 grid: phase ph = h0 & 1, q-row hq = h0 >> 1.  ph = 0: taps kh = 1 (dy1 row hq) + the shortcut (dys row hq);
      // ph = 1: kh = 2 (dy1 row hq) and kh = 0 (dy1 row hq + 1, zero below the map)
      const int hq = h0 >> 1, Hs = a.Ho >> 1;
      ph_n = h0 & 1;
      const long rb = ((long)n * Hs + hq) * (long)(a.Wo >> 1) * 32 + lane * VE;
      dv[0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.dy1) + rb);
      dv[1] = (u32x4){0, 0, 0, 0};
      if (!ph_n) dv[1] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.dys) + rb);
      else if (hq + 1 < Hs) dv[1] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.dy1) + rb + (long)(a.Wo >> 1) * 32);
      const long base = (((long)n * a.Ho + h0) * a.Wo) * 32 + lane * VE;
#pragma unroll
      for (int k = 0; k < NS; ++k) yv[k] = *reinterpret_cast<const Vec16*>(Y + base + k * 64 * VE);
    } else if constexpr (MODE == 0) {
      const long base = (((long)n * a.Ho + h0) * a.Wo) * 32 + lane * VE;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        gv[k] = *reinterpret_cast<const Vec16*>(G + base + k * 64 * VE);
        yv[k] = *reinterpret_cast<const Vec16*>(Y + base + k * 64 * VE);
      }
    }
    const long xbase = ((long)n * a.H + r0_c) * a.W;      // may be negative: only in-range rows are read
#pragma unroll
    for (int k = 0; k < NXV; ++k) {
      xv[k] = Vec16{{0, 0, 0, 0}};
      if ((unsigned)(r0_c + xrow[k]) < (unsigned)a.H) xv[k] = *reinterpret_cast<const Vec16*>(X + (xbase + (long)xrow[k] * a.W + xcol[k]));
    }
  };
  int ph_c = 0;                                     // DG: phase of the slab being committed
  auto commit = [&]() {
    if constexpr (DG) {
      // ---- g slab = data gradient of conv1 (+ shortcut) for this output row, on the matrix cores: D[ci][pixel wq of column phase pw]
      *reinterpret_cast<u32x4*>(sDy + (lane >> 2) * kDyPitch + (lane & 3) * 16) = dv[0];
      *reinterpret_cast<u32x4*>(sDy + kDyRow + (lane >> 2) * kDyPitch + (lane & 3) * 16) = dv[1];
      const char* b0 = sDy + r * kDyPitch + gq * 16;               // B fragment: pixel r (+ dw), channels (co) 8 gq ..
      const Vec16 r0d0 = *reinterpret_cast<const Vec16*>(b0), r0d1 = *reinterpret_cast<const Vec16*>(b0 + kDyPitch);
      const Vec16 r1d0 = *reinterpret_cast<const Vec16*>(b0 + kDyRow), r1d1 = *reinterpret_cast<const Vec16*>(b0 + kDyRow + kDyPitch);
      auto wf = [&](int f, int m) { return *reinterpret_cast<const Vec16*>(sWf + ((f * 2 + m) * 64 + lane) * 16); };
      f32x4 d[2][2];                                               // [pw][ci half]
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        d[0][m] = (f32x4){0, 0, 0, 0}; d[1][m] = (f32x4){0, 0, 0, 0};
        if (ph_c == 0) {
          // phase (0,0): tap (1,1) dw 0 [slot 0] + shortcut [slot 9] on dys (row buffer 1);  phase (0,1): taps (1,0) dw 1 [1], (1,2) dw 0 [2]
          d[0][m] = mma_bf16(wf(0, m), r0d0, d[0][m]);
          d[0][m] = mma_bf16(wf(9, m), r1d0, d[0][m]);
          d[1][m] = mma_bf16(wf(1, m), r0d1, d[1][m]);
          d[1][m] = mma_bf16(wf(2, m), r0d0, d[1][m]);
        } else {
          // phase (1,0): taps (0,1) dh 1 [3], (2,1) dh 0 [4];  phase (1,1): (0,0) dh 1 dw 1 [5], (0,2) dh 1 dw 0 [6], (2,0) dh 0 dw 1 [7], (2,2) [8]
          d[0][m] = mma_bf16(wf(3, m), r1d0, d[0][m]);
          d[0][m] = mma_bf16(wf(4, m), r0d0, d[0][m]);
          d[1][m] = mma_bf16(wf(5, m), r1d1, d[1][m]);
          d[1][m] = mma_bf16(wf(6, m), r1d0, d[1][m]);
          d[1][m] = mma_bf16(wf(7, m), r0d1, d[1][m]);
          d[1][m] = mma_bf16(wf(8, m), r0d0, d[1][m]);
        }
      }
      // lane (gq, r): channels 16 m + 4 gq .. + 3 of pixel 2 r + pw -> the slab [pixel][32 channels] as bf16 (what the stored gradient was)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          *reinterpret_cast<uint2*>(sP + (2 * r + pw) * PITCH + (16 * m + 4 * gq) * 2) =
              make_uint2(pack2_bf16(d[pw][m][0], d[pw][m][1]), pack2_bf16(d[pw][m][2], d[pw][m][3]));
#pragma unroll
      for (int k = 0; k < NS; ++k) gv[k] = *reinterpret_cast<const Vec16*>(sP + (lane + 64 * k) * 16);
    }
    if constexpr (MODE == 0) {
      float ms[VE], mb[VE];
#pragma unroll
      for (int j = 0; j < VE; j += 4) {
        const float4 u = *reinterpret_cast<const float4*>(sC + cv * VE + j), w = *reinterpret_cast<const float4*>(sC + 32 + cv * VE + j);
        ms[j] = u.x; ms[j + 1] = u.y; ms[j + 2] = u.z; ms[j + 3] = u.w;
        mb[j] = w.x; mb[j + 1] = w.y; mb[j + 2] = w.z; mb[j + 3] = w.w;
      }
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float fg[VE], fy[VE];
        Elem<T>::unpack(gv[k], fg);
        Elem<T>::unpack(yv[k], fy);
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          const float gm = (fy[j] * ms[j] + mb[j] > 0.f) ? fg[j] : 0.f;
          s1[j] += gm * fy[j];
          fg[j] = gm;
        }
        *reinterpret_cast<Vec16*>(sP + (lane + 64 * k) * 16) = Elem<T>::pack(fg);     // slab vector index == LDS vector index
      }
    }
#pragma unroll
    for (int k = 0; k < NXV; ++k) {
      if (xrow[k] < (1 << 28)) {
        float f[VE];
        Elem<T>::unpack(xv[k], f);
        float* d = sX + xrow[k] * a.pw + 4 + xcol[k];
#pragma unroll
        for (int j = 0; j < VE; j += 4) *reinterpret_cast<float4*>(d + j) = make_float4(f[j], f[j + 1], f[j + 2], f[j + 3]);
      }
    }
  };

  const int gw = blockIdx.x * 4 + wv, gstride = gridDim.x * 4;
  int slab = gw;
  if (slab < a.nslabs) issue(slab);
  for (; slab < a.nslabs; slab += gstride) {
    ph_c = ph_n;
    commit();
    if constexpr (MODE == 1) {
      // im2col slab of the image itself: taps 16bh .. 16bh+15 of pixel bq (tap 25 = 1, taps > 25 = 0)
      constexpr int NV = 16 / VE;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float f[VE];
#pragma unroll
        for (int j = 0; j < VE; ++j) {
          // bh is per lane: both constant tap sets are read, one is kept
          const int tlo = v * VE + j, thi = 16 + v * VE + j;
          const float lo = sX[boff + (tlo / 5) * a.pw + tlo % 5];
          const float hi = thi < 25 ? sX[boff + (thi / 5) * a.pw + thi % 5] : (thi == 25 ? 1.f : 0.f);
          f[j] = bh ? hi : lo;
        }
        *reinterpret_cast<Vec16*>(sP + bq * PITCH + (16 * bh + v * VE) * ES) = Elem<T>::pack(f);
      }
    }
    if (slab + gstride < a.nslabs) issue(slab + gstride);
    if constexpr (sizeof(T) == 2) {
      Vec16 bf[2];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = tapo[tb] >= 0 ? sX[offXb[j >> 2] + 2 * (j & 3) + tapo[tb]] : b_fill;
        bf[tb] = Elem<bf16_t>::pack(f);
      }
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        const Vec16 af = FragOps<bf16_t>::load(sP, offP[0] + ta * 32, offP[1] + ta * 32);
        acc[ta][0] = mma_bf16(af, bf[0], acc[ta][0]);
        acc[ta][1] = mma_bf16(af, bf[1], acc[ta][1]);
      }
    } else {
      // exact-f32 mode: 8 steps of v_mfma_f32_16x16x4_f32, step j covers slab pixels 4j + gq
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = 4 * j + gq;
        const int po = pix_off(p);
        float bvv[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) bvv[tb] = tapo[tb] >= 0 ? sX[po + tapo[tb]] : b_fill;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
          const float av = *reinterpret_cast<const float*>(sP + p * PITCH + (16 * ta + r) * 4);
          acc[ta][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bvv[0], acc[ta][0], 0, 0, 0);
          acc[ta][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bvv[1], acc[ta][1], 0, 0, 0);
        }
      }
    }
  }
  // ---- flush: fixed summation order (wave 0, 1, 2, 3) -> bit-reproducible partials
  // sum gm * y0: lanes of a wave with equal channel vector first (xor shuffles), then the waves in order
#pragma unroll
  for (int j = 0; j < VE; ++j) {
#pragma unroll
    for (int o = CV; o < 64; o <<= 1) s1[j] += __shfl_xor(s1[j], o, 64);
  }
  __syncthreads();
  float* sRed = reinterpret_cast<float*>(smem);
  for (int i = t; i < kStemPartFloats; i += 256) sRed[i] = 0.f;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wv == w) {
      if (lane < CV) {
#pragma unroll
        for (int j = 0; j < VE; ++j) sRed[32 + lane * VE + j] += s1[j];
      }
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) sRed[64 + (16 * ta + 4 * gq + jj) * 32 + 16 * tb + r] += acc[ta][tb][jj];
    }
  }
  __syncthreads();
  if (t < 32) sRed[t] = sRed[64 + t * 32 + 25];   // S0[c] = image[c][25]
  __syncthreads();
  float* dst = a.partials + (long)blockIdx.x * kStemPartFloats;
  for (int i = t; i < kStemPartFloats; i += 256) dst[i] = sRed[i];
}

// R[i] = sum over the gram kernel's partial rows, in double (row 25 = sum patch)
__global__ __launch_bounds__(256) void stem_gram_sum_kernel(const float* __restrict__ partials, int nparts, double* __restrict__ R) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per element, 0 .. 1023
  double s = 0.0;
  for (int p = lane; p < nparts; p += 64) s += (double)partials[(long)p * kStemPartFloats + 64 + i];
  s = wave_sum_d(s);
  if (lane == 0) R[i] = s;
}

// ---------------------------------------------------------------- finalize: one block per stem channel
struct StemBwdFinArgs {
  const float* partials; int nparts;
  const double* R;               // [32][32] patch gram matrix of the batch; row 25 = sum patch
  const float* w;                // stem weights [32][25] f32 (W2 = w x R: sum y0 (x) patch without reading y0 again)
  const float* global_sums;      // SyncBN: [2][32] sums over the global batch (coefficients); NULL: the local sums
  double count;                  // pixels behind the sums the coefficients use
  const float* gamma; const float* mean; const float* istd;
  float* dgamma; float* dbeta; float* dW;       // dW [32][25]  (+=)
};

// 1024 threads: 32 part-lanes x 32 taps; every thread's loads are independent (24 rows each at 768 partial rows, 8 accumulators), so the
// kernel costs a few memory latencies instead of a dependent chain of them (it sits at the very end of the step's critical stream)
__global__ __launch_bounds__(1024) void stem_bwd_finalize_kernel(StemBwdFinArgs a) {
  __shared__ double sB[2][16];
  __shared__ double sW[32][32];
  __shared__ double sR[26 * 32];
  __shared__ float sw[25];
  const int c = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < 26 * 32; i += 1024) sR[i] = a.R[i];
  if (t < 25) sw[t] = a.w[c * 25 + t];
  double b0 = 0.0, b1 = 0.0;
  for (int p = t; p < a.nparts; p += 1024) {
    const float* row = a.partials + (long)p * kStemPartFloats;
    b0 += (double)row[c]; b1 += (double)row[32 + c];
  }
  b0 = wave_sum_d(b0); b1 = wave_sum_d(b1);
  if ((t & 63) == 0) { sB[0][t >> 6] = b0; sB[1][t >> 6] = b1; }
  // W1[c][tap]: 32 part-lanes per tap, 8 independent loads in flight each
  const int pl = t >> 5, tap = t & 31;
  const float* src = a.partials + 64 + c * 32 + tap;
  double w1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) w1[k] = 0.0;
  int p = pl;
  for (; p + 224 < a.nparts; p += 256) {
#pragma unroll
    for (int k = 0; k < 8; ++k) w1[k] += (double)src[(long)(p + 32 * k) * kStemPartFloats];
  }
  for (; p < a.nparts; p += 32) w1[0] += (double)src[(long)p * kStemPartFloats];
  sW[pl][tap] = ((w1[0] + w1[1]) + (w1[2] + w1[3])) + ((w1[4] + w1[5]) + (w1[6] + w1[7]));
  __syncthreads();
  double S0 = 0.0, S1 = 0.0;
  for (int q = 0; q < 16; ++q) { S0 += sB[0][q]; S1 += sB[1][q]; }
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
    double v1 = 0.0;
    for (int q = 0; q < 32; ++q) v1 += sW[q][t];
    double v2 = 0.0;                                   // sum_pix y0[c] * patch[t] = sum_t' w[c][t'] * R[t'][t]
    for (int u = 0; u < 25; ++u) v2 += (double)sw[u] * sR[u * 32 + t];
    const double v3 = sR[25 * 32 + t];
    a.dW[c * 25 + t] += (float)(A * (v1 - m2 * istd * (v2 - mean * v3) - m1 * v3));
  }
}

// 32-pixel slabs of whole output rows: the output width S/2 must divide 32; image rows are read as 16-byte vectors
bool stem_bwd_fusable(int S) { return S == 64 || S == 32 || S == 16; }

int stem_bwd_part_floats() { return kStemPartFloats; }

static int stem_geom(StemBwdArgs& a, int dt, int N, int S, int Ho, int Wo) {
  if (!stem_bwd_fusable(S) || Wo * 2 != S || Ho != Wo) { set_error("stem_bwd: image size %d unsupported", S); return MMVAE_ERR_UNSUPPORTED; }
  a.N = N; a.H = S; a.W = S; a.Ho = Ho; a.Wo = Wo;
  a.wshift = 0; while ((1 << a.wshift) < Wo) ++a.wshift;
  a.rps = 32 / Wo;
  a.slabs_per_img = Ho / a.rps;
  a.nslabs = N * a.slabs_per_img;
  a.prow = 2 * a.rps + 3; a.pw = S + 8;
  const int ve = dt == DT_F32 ? 4 : 8;
  if (a.prow * (S / ve) > 64 * (dt == DT_F32 ? 2 : 1)) { set_error("stem_bwd: patch of %d x %d exceeds the staging slots", a.prow, S); return MMVAE_ERR_UNSUPPORTED; }
  return MMVAE_OK;
}
static size_t stem_lds(const StemBwdArgs& a, int dt) {
  const size_t lds = 256 + 4 * ((size_t)32 * 32 * dtype_size(dt) + (size_t)a.prow * a.pw * 4);
  return lds > (size_t)kStemPartFloats * 4 ? lds : (size_t)kStemPartFloats * 4;
}

// ---------------------------------------------------------------- stem forward as the same kind of stream (bf16 storage)
// encoder.conv1 (1 -> 32, k5 s2 p2) is 25 multiply-adds per output: the patch-tile kernel stages the 1-channel image as 8 padded
// channels and moves the layer at 1.6 TB/s (212 us at N = 5120).  Here every wave walks its own 32-pixel slabs: image rows in a
// wave-private f32 LDS patch (loaded once, next slab in flight), the B fragment of a pixel tile = its 25 taps gathered from that patch
// (7 zero taps pad K to 32: ONE v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels), A = the weights, built once from the f32
// master (no packing launch), a lane ends with 8 consecutive channels of a pixel = one 16-byte store; BatchNorm sums from the f32
// results, per-block partial rows in fixed order.
struct StemFwdArgs {
  const void* x; const float* w; void* y; float* stats;
  int N, H, W, Ho, Wo, wshift, rps, slabs_per_img, nslabs, prow, pw;
};

__global__ __launch_bounds__(256, 4) void stem_fwd_stream_kernel(StemFwdArgs a) {
  typedef bf16_t T;
  constexpr int VE = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  const int wave_bytes = a.prow * a.pw * 4;
  float* sX = reinterpret_cast<float*>(smem + 1024 + wv * wave_bytes);   // this wave's image patch [prow][pw] f32, 4 halo columns a side
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  for (int i = lane; i < a.prow * a.pw; i += 64) sX[i] = 0.f;            // the halo columns stay zero
  // A fragments: row r of fragment m is channel 8*(r/4) + 4m + r%4 (a lane's 2 x 4 results are then 8 consecutive channels);
  // k = taps 8gq .. 8gq+7 (taps >= 25 are zero)
  Vec16 wA[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int c = 8 * (r >> 2) + 4 * m + (r & 3);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int tap = 8 * gq + j; f[j] = tap < 25 ? a.w[c * 25 + tap] : 0.f; }
    wA[m] = Elem<bf16_t>::pack(f);
  }
  // B gather: this lane's 8 taps as patch offsets (or -1)
  int toff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { const int tap = 8 * gq + j; toff[j] = tap < 25 ? (tap / 5) * a.pw + tap % 5 : -1; }
  auto pix_off = [&](int q) { return ((q >> a.wshift) * 2) * a.pw + (q & (a.Wo - 1)) * 2 + 2; };   // tap (0,0) of slab pixel q
  const int po[2] = {pix_off(r), pix_off(16 + r)};
  const int xvr = a.W / VE;
  const int idx = lane;
  const int xrow = idx < a.prow * xvr ? idx / xvr : (1 << 28), xcol = (idx % xvr) * VE;
  float st1[8], st2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { st1[j] = 0.f; st2[j] = 0.f; }
  Vec16 xv = Vec16{{0, 0, 0, 0}};
  auto issue = [&](int slab) {
    const int n = slab / a.slabs_per_img, h0 = (slab - n * a.slabs_per_img) * a.rps;
    const int r0 = 2 * h0 - 2;
    xv = Vec16{{0, 0, 0, 0}};
    if ((unsigned)(r0 + xrow) < (unsigned)a.H) xv = *reinterpret_cast<const Vec16*>(X + (((long)n * a.H + r0 + xrow) * a.W + xcol));
  };
  const int gw = blockIdx.x * 4 + wv, gstride = gridDim.x * 4;
  int slab = gw;
  if (slab < a.nslabs) issue(slab);
  for (; slab < a.nslabs; slab += gstride) {
    if (xrow < (1 << 28)) {
      float f[VE];
      Elem<T>::unpack(xv, f);
      float* d = sX + xrow * a.pw + 4 + xcol;
      *reinterpret_cast<float4*>(d) = make_float4(f[0], f[1], f[2], f[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
    if (slab + gstride < a.nslabs) issue(slab + gstride);
    __builtin_amdgcn_sched_barrier(0);
    bf16_t* yrow = reinterpret_cast<bf16_t*>(a.y) + ((long)slab * 32) * 32 + 8 * gq;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = toff[j] >= 0 ? sX[po[pt] + toff[j]] : 0.f;
      const Vec16 bf = Elem<bf16_t>::pack(f);
      const f32x4 z = (f32x4){0, 0, 0, 0};
      const f32x4 d0 = mma_bf16(wA[0], bf, z), d1 = mma_bf16(wA[1], bf, z);
      const float v[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
      for (int j = 0; j < 8; ++j) { st1[j] += v[j]; st2[j] += v[j] * v[j]; }
      dstore8<bf16_t>(yrow + (16 * pt + r) * 32, v, false);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- BatchNorm sums: the 16 pixel-lanes of a row (DPP), then the four waves in order; one partial row [2][32] per block
#pragma unroll
  for (int j = 0; j < 8; ++j) { st1[j] = row16_sum(st1[j]); st2[j] = row16_sum(st2[j]); }
  float* sb = reinterpret_cast<float*>(smem);                            // [4 waves][64]
  if (r == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sb[wv * 64 + 8 * gq + j] = st1[j]; sb[wv * 64 + 32 + 8 * gq + j] = st2[j]; }
  }
  __syncthreads();
  if (a.stats && t < 64) a.stats[(long)blockIdx.x * 64 + t] = (sb[t] + sb[64 + t]) + (sb[128 + t] + sb[192 + t]);
}

bool stem_fwd_stream_ok(int dt, int S) {
  constexpr bool env = true;
  return env && dt == DT_BF16 && stem_bwd_fusable(S);
}
// y [N][S/2][S/2][32] bf16 = conv1(x), stats (nullable): partial rows [rows][2][32]; returns rows (> 0) or an error
int launch_stem_fwd_stream(int dt, const void* x, const float* w, void* y, float* stats, int N, int S, hipStream_t s) {
  if (!stem_fwd_stream_ok(dt, S)) { set_error("stem_fwd_stream: bf16, image size 64 / 32 / 16"); return MMVAE_ERR_UNSUPPORTED; }
  StemBwdArgs g; memset(&g, 0, sizeof(g));
  const int Ho = S / 2;
  const int rc0 = stem_geom(g, dt, N, S, Ho, Ho);
  if (rc0 < 0) return rc0;
  StemFwdArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.y = y; a.stats = stats;
  a.N = N; a.H = S; a.W = S; a.Ho = Ho; a.Wo = Ho; a.wshift = g.wshift; a.rps = g.rps; a.slabs_per_img = g.slabs_per_img; a.nslabs = g.nslabs;
  a.prow = g.prow; a.pw = g.pw;
  int gx = 1024;
  if (gx * 4 > a.nslabs) gx = (a.nslabs + 3) / 4;
  const size_t lds = 1024 + 4 * (size_t)a.prow * a.pw * 4;
  hipLaunchKernelGGL(stem_fwd_stream_kernel, dim3(gx), dim3(256), lds, s, a);
  note_launch_bytes((double)N * ((double)S * S * 2.0 + (double)(S / 2) * (S / 2) * 32 * 2.0));
  const int rc = check_launch("stem_fwd_stream");
  return rc ? rc : gx;
}

// Patch gram matrix of the batch (input only): scratch = rows * stem_bwd_part_floats() floats, R = 1024 doubles.
int launch_stem_gram(int dt, const void* x, float* scratch, long scratch_cap_floats, double* R, int N, int S, int Ho, int Wo, hipStream_t s) {
  StemBwdArgs a; memset(&a, 0, sizeof(a));
  const int rc0 = stem_geom(a, dt, N, S, Ho, Wo);
  if (rc0 < 0) return rc0;
  a.x = x; a.partials = scratch;
  int gx = 1024;
  if (gx * 4 > a.nslabs) gx = (a.nslabs + 3) / 4;
  if ((long)gx * kStemPartFloats > scratch_cap_floats) gx = (int)(scratch_cap_floats / kStemPartFloats);
  if (gx < 1) { set_error("stem_gram: scratch too small"); return MMVAE_ERR_WORKSPACE; }
  if (dt == DT_F32) hipLaunchKernelGGL((stem_bwd_kernel<float, 1>), dim3(gx), dim3(256), stem_lds(a, dt), s, a);
  else hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, 1>), dim3(gx), dim3(256), stem_lds(a, dt), s, a);
  int rc = check_launch("stem_gram");
  if (rc) return rc;
  hipLaunchKernelGGL(stem_gram_sum_kernel, dim3(256), dim3(256), 0, s, scratch, gx, R);
  return check_launch("stem_gram_sum");
}

// Returns the number of partial rows (> 0) or an error.  partials: rows * stem_bwd_part_floats() floats.
int launch_stem_bwd(int dt, const void* g, const void* y0, const void* x, const float* ms, const float* mb, float* partials,
                    long partials_cap_floats, int N, int S, int Ho, int Wo, hipStream_t s) {
  StemBwdArgs a; memset(&a, 0, sizeof(a));
  const int rc0 = stem_geom(a, dt, N, S, Ho, Wo);
  if (rc0 < 0) return rc0;
  a.g = g; a.y0 = y0; a.x = x; a.ms = ms; a.mb = mb; a.partials = partials;
  int gx = 1024;
  if (gx * 4 > a.nslabs) gx = (a.nslabs + 3) / 4;
  if ((long)gx * kStemPartFloats > partials_cap_floats) gx = (int)(partials_cap_floats / kStemPartFloats);
  if (gx < 1) { set_error("stem_bwd: partials buffer too small"); return MMVAE_ERR_WORKSPACE; }
  if (dt == DT_F32) hipLaunchKernelGGL((stem_bwd_kernel<float, 0>), dim3(gx), dim3(256), stem_lds(a, dt), s, a);
  else hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, 0>), dim3(gx), dim3(256), stem_lds(a, dt), s, a);
  note_launch_bytes((double)N * ((double)S * S * 2.0 + 2.0 * Ho * Wo * 32 * 2.0));     // x, g, y0
  const int rc = check_launch("stem_bwd");
  return rc ? rc : gx;
}

// The same pass with the incoming gradient recomputed from encoder.layer1's dy tensors (see StemBwdArgs): bf16, S = 64 (one output row per slab)
bool stem_bwd_dg_ok(int dt, int S) { return dt == DT_BF16 && S == 64; }
int launch_stem_bwd_dg(const void* dy1, const void* dys, const void* wd1, const void* wds, const void* y0, const void* x, const float* ms, const float* mb,
                       float* partials, long partials_cap_floats, int N, int S, int Ho, int Wo, hipStream_t s) {
  StemBwdArgs a; memset(&a, 0, sizeof(a));
  const int rc0 = stem_geom(a, DT_BF16, N, S, Ho, Wo);
  if (rc0 < 0) return rc0;
  if (a.rps != 1 || Wo != 32) { set_error("stem_bwd_dg: needs 32-pixel output rows"); return MMVAE_ERR_UNSUPPORTED; }
  a.dy1 = dy1; a.dys = dys; a.wd1 = wd1; a.wds = wds; a.y0 = y0; a.x = x; a.ms = ms; a.mb = mb; a.partials = partials;
  int gx = 768;                                          // three 256-thread blocks per CU (47 KB of LDS each): one resident round
  if (gx * 4 > a.nslabs) gx = (a.nslabs + 3) / 4;
  if ((long)gx * kStemPartFloats > partials_cap_floats) gx = (int)(partials_cap_floats / kStemPartFloats);
  if (gx < 1) { set_error("stem_bwd: partials buffer too small"); return MMVAE_ERR_WORKSPACE; }
  const size_t lds = 256 + 4 * ((size_t)32 * 32 * 2 + (size_t)a.prow * a.pw * 4 + 2 * 17 * 96) + 20 * 1024;
  hipLaunchKernelGGL((stem_bwd_kernel<bf16_t, 0, true>), dim3(gx), dim3(256), lds > (size_t)kStemPartFloats * 4 ? lds : (size_t)kStemPartFloats * 4, s, a);
  note_launch_bytes((double)N * ((double)S * S * 2.0 + Ho * Wo * 32 * 2.0 + 2.0 * (Ho / 2) * (Wo / 2) * 32 * 2.0));     // x, y0, dy1, dys
  const int rc = check_launch("stem_bwd_dg");
  return rc ? rc : gx;
}

int launch_stem_bwd_finalize(const float* partials, int nparts, const double* R, const float* w, const float* global_sums, double count,
                             const float* gamma, const float* mean, const float* istd, float* dgamma, float* dbeta, float* dW, hipStream_t s) {
  StemBwdFinArgs a{partials, nparts, R, w, global_sums, count, gamma, mean, istd, dgamma, dbeta, dW};
  hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3(32), dim3(1024), 0, s, a);
  return check_launch("stem_bwd_finalize");
}

}  // namespace mmvae
