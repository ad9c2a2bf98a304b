// wgrad_stream_kernel: weight gradient of the stride-2 k4 layers on the big, thin feature maps (ConvTranspose2d 16/32 -> 16
// channels, 32x32 -> 64x64: decoder.uplayer5.conv2 / .upsample, 839 MB of operands each at N = 5120) as a barrier-free stream.
//
//   dW[a][b][kh][kw] = sum_{n,h,w} P[n,h,w,a] * G[n, S*h + kh - PAD, S*w + kw - PAD, b]        (P: small grid, G: large grid)
//
// wgrad2_kernel (conv_wgrad.inc) stages a 128-pixel P tile and its G patch per block behind two barriers and moves these layers
// at 2.0-2.9 TB/s.  Here every WAVE walks strips of P rows on its own: a ring of G rows and the current P row live in
// wave-private LDS (no block barrier anywhere in the loop), every G row is loaded from memory exactly once per strip, the
// next step's rows are in flight in registers while the current step multiplies, and all KS*KS tap accumulators of the
// (Ca x Cb) tile stay in registers.  One step = 32 P pixels (one row) = one MFMA K-step: the A fragment (P^T) and the 16 B
// fragments (G rows S*h+kh-PAD, columns S*w+kw-PAD) come out of the natural [pixel][channel] LDS images through
// ds_read_b64_tr_b16.  Out-of-range rows and the two border columns are zeros in LDS.  At the end the block's waves add their
// accumulators in LDS and store ONE partial image [tap][a][b] per block (summed by wgrad_reduce_kernel: no atomics, fixed order).
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

struct WStreamArgs {
  const void* P; const void* G; float* part;
  const float* proP_scale; const float* proP_shift; int proP_relu;
  const float* proG_scale; const float* proG_shift; int proG_relu;
  int N, Hp, Hg, Wg;
  int HS;                    // P rows per strip (divides Hp)
  int nunits;                // N * Hp / HS
  // fused data gradient (DG): dx[n,h,w,a] = sum_{kh,kw,b} G[n, S*h+kh-PAD, S*w+kw-PAD, b] * wd[a][kh*KS+kw][b]  (+ x2[n,h,w,:] . w2[a][:])
  const void* wd;            // packed [Ca][KS*KS][Cb] of T (the conv's "down" form)
  void* dx;                  // [N][Hp][WP][Ca] of T
  const void* x2; const void* w2;   // optional second source on the P grid: x2 [N][Hp][WP][16], w2 packed [Ca][16]
  // (X2) the pass also accumulates the 1x1 conv's own weight gradient, dW2[c2][a] = sum x2[n,h,w,c2] * pro(P[n,h,w,a]) -- both rows are
  // in LDS anyway -- as one more image of the block's partial: [NT + 1][16][16]
  // BatchNorm-backward partial sums of the BN that produced P (DG + PRO_P): g = dx * (P*scale+shift > 0); rows [block][2][16]: sum g, sum g * P
  float* bn_part;
};

// KS x KS taps, stride S, padding PAD; WP = P row width (32: one row per MFMA K-step); CA16, CB16: channel tiles of P and G
// DG: the same pass also produces the data gradient w.r.t. P (the G rows a P row needs for its weight gradient are exactly the rows
// its data gradient needs): the dy tensor is read ONCE for both.  Two taps (2 x 16 G channels) make one 16x16x32 MFMA K-step; the A
// operand is the conv's packed "down" weights (8 fragments, loaded once), B fragments are plain 16-byte reads of ring pixels.
// X2 (with DG): + the 1x1 shortcut's share, x2 (x) w2, from a second row staged per step (the block-input gradient in one kernel).
// ST (with DG and PRO_P): P is a pre-BatchNorm tensor whose BN+ReLU is the prologue; the pass also reduces that BatchNorm's backward sums
// (sum g, sum g*P over all pixels, g = dx masked by the ReLU) from the data gradient it has just computed -- the bn_bwd_reduce launch
// that would re-read dx and P disappears.
template <int KS, int S, int PAD, int WP, int CA16, int CB16, bool PRO_P, bool PRO_G, bool DG, bool X2, bool ST>
__global__ __launch_bounds__(256, 2) void wgrad_stream_kernel(WStreamArgs a) {
  static_assert(!ST || (DG && PRO_P && !X2), "BatchNorm sums ride on the data gradient of a prologue'd P");
  static_assert(WP == 32, "one P row per K-step");
  static_assert(!DG || (CA16 == 1 && CB16 == 1 && KS == 4), "fused data gradient: 16 -> 16 channels, 4x4 taps");
  static_assert(!X2 || DG, "the second source rides on the data gradient");
  constexpr int CAB = CA16 * 32, CBB = CB16 * 32;          // bytes per P / G pixel
  constexpr int WL = S * (WP - 1) + KS;                    // G columns a P row reaches: -PAD .. S*(WP-1)+KS-1-PAD
  constexpr int ROWB = WL * CBB;                           // bytes per LDS G row
  constexpr int NSLOT = KS + S;                            // ring: the KS rows being multiplied + the S rows arriving
  constexpr int PB = WP * CAB;                             // bytes of a P row
  constexpr int X2B = X2 ? WP * 32 : 0;                    // bytes of an x2 row (16 channels)
  constexpr int RAWB = ST ? PB : 0;                        // raw copy of the P row (before the prologue) for the BatchNorm sums
  constexpr int WAVE_LDS = NSLOT * ROWB + PB + X2B + RAWB;
  constexpr int NT = KS * KS;
  constexpr int WSIZE = (NT + (X2 ? 1 : 0)) * CA16 * 16 * CB16 * 16;   // floats of a partial image (X2: + the 1x1 conv's 16 x 16)
  static_assert(4 * WAVE_LDS >= WSIZE * 4 + 512, "the flush image (+ 128 floats of BatchNorm sums) aliases the rings");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* ring = smem + wv * WAVE_LDS;
  char* prow = ring + NSLOT * ROWB;
  char* x2row = prow + PB;
  char* rawrow = x2row + X2B;
  constexpr int Wg = S * WP;                               // G row width (the launcher checks it)
  constexpr int grow_bytes = Wg * CBB;                     // bytes of a G row in memory
  constexpr int GV = (S * grow_bytes + 1023) / 1024;       // 16-byte vectors per lane for the S rows of a step
  constexpr int PV = PB / 1024;                            // ... for the P row
  static_assert(PB % 1024 == 0, "P row = whole wave loads");

  // ---- border columns are zero for the whole kernel: columns [0, PAD) and [PAD + Wg, WL) of every slot
  for (int s = 0; s < NSLOT; ++s) {
    for (int i = lane; i < (PAD * CBB) / 16; i += 64) reinterpret_cast<Vec16*>(ring + s * ROWB)[i] = Vec16{{0, 0, 0, 0}};
    for (int i = lane; i < ((WL - PAD) * CBB - grow_bytes) / 16; i += 64)
      reinterpret_cast<Vec16*>(ring + s * ROWB + PAD * CBB + grow_bytes)[i] = Vec16{{0, 0, 0, 0}};
  }
  // ---- prologue coefficients of the channels this lane stages (8 consecutive channels per 16-byte vector)
  float psc[8], psh[8], gsc[8], gsh[8];
  if (PRO_P) {
    const int c = (lane % (CAB / 16)) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { psc[j] = a.proP_scale[c + j]; psh[j] = a.proP_shift[c + j]; }
  }
  if (PRO_G) {
    const int c = (lane % (CBB / 16)) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gsc[j] = a.proG_scale[c + j]; gsh[j] = a.proG_shift[c + j]; }
  }
  const float p_lo = a.proP_relu ? 0.f : -__builtin_inff(), g_lo = a.proG_relu ? 0.f : -__builtin_inff();
  // ---- fragment offsets (tile-invariant): k-slice gq = P pixels 8gq .. 8gq+7, two 4-pixel blocks; lane i = r supplies the
  // address of pixel (i >> 2) of the block, channel quad (i & 3)
  int offA[2], offB[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int w = 8 * gq + 4 * blk + (r >> 2);
    offA[blk] = w * CAB + (r & 3) * 8;
    offB[blk] = (S * w) * CBB + (r & 3) * 8;               // + kw * CBB + slot * ROWB (+ 32 per channel tile)
  }
  f32x4 acc[NT][CA16][CB16];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int ca = 0; ca < CA16; ++ca)
#pragma unroll
      for (int cb = 0; cb < CB16; ++cb) acc[k][ca][cb] = (f32x4){0, 0, 0, 0};

  f32x4 acc2 = (f32x4){0, 0, 0, 0};                         // X2: the 1x1 conv's weight gradient [x2 channel][P channel]
  float bsc[4] = {0, 0, 0, 0}, bsh[4] = {0, 0, 0, 0}, bs0[4] = {0, 0, 0, 0}, bs1[4] = {0, 0, 0, 0};
  if constexpr (ST) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { bsc[j] = a.proP_scale[4 * gq + j]; bsh[j] = a.proP_shift[4 * gq + j]; }
  }
  Vec16 wdA[DG ? KS * KS / 2 : 1], w2A = Vec16{{0, 0, 0, 0}};
  if constexpr (DG) {
    // A[row a = r][k = 8gq ..]: taps (kh, 2kp) and (kh, 2kp + 1), 16 G channels each = 64 contiguous bytes of wd[a][tap][b]
#pragma unroll
    for (int pr = 0; pr < KS * KS / 2; ++pr)
      wdA[pr] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.wd) + r * (KS * KS * 32) + pr * 64 + gq * 16);
    if (X2 && gq < 2) w2A = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.w2) + r * 32 + gq * 16);
  }
  const bf16_t* __restrict__ Pm = reinterpret_cast<const bf16_t*>(a.P);
  const bf16_t* __restrict__ Gm = reinterpret_cast<const bf16_t*>(a.G);
  // XCD-aware unit walk: block b runs on XCD b % 8; every XCD sweeps its own eighth of the units front to back
  const int nblk = gridDim.x;
  int u_first, u_step, u_end;
  if ((nblk & 7) == 0) {
    const int per = (a.nunits + 7) >> 3;
    const int lo = (blockIdx.x & 7) * per;
    u_first = lo + (blockIdx.x >> 3) * 4 + wv; u_step = (nblk >> 3) * 4; u_end = min(a.nunits, lo + per);
  } else { u_first = blockIdx.x * 4 + wv; u_step = nblk * 4; u_end = a.nunits; }
  const int nstrips = a.Hp / a.HS;
  const int nq = a.HS + 1;                                  // steps of a unit: the priming step + one per P row

  // registers of the step in flight
  Vec16 gv[GV], pv[PV], xv = Vec16{{0, 0, 0, 0}};
  auto issue = [&](int u, int q) {
    const int n = u / nstrips, h0 = (u - n * nstrips) * a.HS;
    // G rows [top - S + 1, top], top = last row of P row h0 + q - 1
    const int top = S * (h0 + q - 1) - PAD + KS - 1;
#pragma unroll
    for (int k = 0; k < GV; ++k) {
      const int byte = (lane + 64 * k) * 16;
      const int rr = byte / grow_bytes, off = byte - rr * grow_bytes;
      const int row = top - S + 1 + rr;
      gv[k] = Vec16{{0, 0, 0, 0}};
      if (rr < S && row >= 0 && row < a.Hg)
        gv[k] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(Gm) + (((long)n * a.Hg + row) * Wg) * CBB + off);
    }
    if (q > 0) {
#pragma unroll
      for (int k = 0; k < PV; ++k)
        pv[k] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(Pm) + (((long)n * a.Hp + (h0 + q - 1)) * WP) * CAB + (lane + 64 * k) * 16);
      if constexpr (X2) xv = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.x2) + (((long)n * a.Hp + (h0 + q - 1)) * WP) * 32 + lane * 16);
    }
  };
  auto commit = [&](int u, int q) {
    const int n = u / nstrips, h0 = (u - n * nstrips) * a.HS;
    const int top = S * (h0 + q - 1) - PAD + KS - 1;
#pragma unroll
    for (int k = 0; k < GV; ++k) {
      const int byte = (lane + 64 * k) * 16;
      const int rr = byte / grow_bytes, off = byte - rr * grow_bytes;
      if (rr < S) {
        const int row = top - S + 1 + rr;
        Vec16 v = gv[k];
        if (PRO_G && row >= 0 && row < a.Hg) {
          float f[8];
          Elem<bf16_t>::unpack(v, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * gsc[j] + gsh[j], g_lo);
          v = Elem<bf16_t>::pack(f);
        }
        const int slot = (row + 4 * NSLOT) % NSLOT;
        *reinterpret_cast<Vec16*>(ring + slot * ROWB + PAD * CBB + off) = v;
      }
    }
    if (q > 0) {
#pragma unroll
      for (int k = 0; k < PV; ++k) {
        Vec16 v = pv[k];
        if constexpr (ST) *reinterpret_cast<Vec16*>(rawrow + (lane + 64 * k) * 16) = v;
        if (PRO_P) {
          float f[8];
          Elem<bf16_t>::unpack(v, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * psc[j] + psh[j], p_lo);
          v = Elem<bf16_t>::pack(f);
        }
        *reinterpret_cast<Vec16*>(prow + (lane + 64 * k) * 16) = v;
      }
      if constexpr (X2) *reinterpret_cast<Vec16*>(x2row + lane * 16) = xv;
    }
  };

  int u = u_first, q = 0;
  if (u < u_end) issue(u, 0);
  while (u < u_end) {
    commit(u, q);
    // the next step of the flat (unit, step) sequence goes in flight
    int un = u, qn = q + 1;
    if (qn == nq) { un = u + u_step; qn = 0; }
    if (un < u_end) issue(un, qn);
    __builtin_amdgcn_sched_barrier(0);
    if (q > 0) {
      const int n = u / nstrips, h0 = (u - n * nstrips) * a.HS;
      const int first = S * (h0 + q - 1) - PAD;             // G row of tap row kh = 0
      Vec16 af[CA16];
#pragma unroll
      for (int ca = 0; ca < CA16; ++ca) af[ca] = FragOps<bf16_t>::load(prow, offA[0] + 32 * ca, offA[1] + 32 * ca);
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        const char* rowp = ring + ((first + kh + 4 * NSLOT) % NSLOT) * ROWB;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
          for (int cb = 0; cb < CB16; ++cb) {
            const Vec16 bf = FragOps<bf16_t>::load(rowp, offB[0] + kw * CBB + 32 * cb, offB[1] + kw * CBB + 32 * cb);
#pragma unroll
            for (int ca = 0; ca < CA16; ++ca) acc[kh * KS + kw][ca][cb] = mma_bf16(af[ca], bf, acc[kh * KS + kw][ca][cb]);
          }
        }
      }
      if constexpr (X2) {
        // x2^T (16 x 32 pixels) times the prologue'd P row (32 pixels x 16): the same fragment form as af, from the x2 row
        const Vec16 ax = FragOps<bf16_t>::load(x2row, offA[0], offA[1]);
        acc2 = mma_bf16(ax, af[0], acc2);
      }
      if constexpr (DG) {
        // D[a][pixel]: lane (r = pixel of the 16-pixel tile, gq) ends with channels a = 4gq .. 4gq+3
        f32x4 dacc[WP / 16];
#pragma unroll
        for (int pt = 0; pt < WP / 16; ++pt) dacc[pt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
          const char* rowp = ring + ((first + kh + 4 * NSLOT) % NSLOT) * ROWB;
#pragma unroll
          for (int kp = 0; kp < KS / 2; ++kp)
#pragma unroll
            for (int pt = 0; pt < WP / 16; ++pt) {
              const Vec16 b = *reinterpret_cast<const Vec16*>(rowp + (S * (16 * pt + r) + 2 * kp + (gq >> 1)) * CBB + (gq & 1) * 16);
              dacc[pt] = mma_bf16(wdA[kh * (KS / 2) + kp], b, dacc[pt]);
            }
        }
        if constexpr (X2) {
#pragma unroll
          for (int pt = 0; pt < WP / 16; ++pt) {
            const Vec16 b = *reinterpret_cast<const Vec16*>(x2row + (16 * pt + r) * 32 + (gq & 1) * 16);     // k >= 16: w2A is zero there
            dacc[pt] = mma_bf16(w2A, b, dacc[pt]);
          }
        }
        if constexpr (ST) {
#pragma unroll
          for (int pt = 0; pt < WP / 16; ++pt) {
            const uint2 yr = *reinterpret_cast<const uint2*>(rawrow + (16 * pt + r) * CAB + gq * 8);      // P[pixel][4gq .. 4gq+3]
            const float y[4] = {__uint_as_float(yr.x << 16), __uint_as_float(yr.x & 0xffff0000u), __uint_as_float(yr.y << 16), __uint_as_float(yr.y & 0xffff0000u)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = (y[j] * bsc[j] + bsh[j] > 0.f) ? dacc[pt][j] : 0.f;
              bs0[j] += g; bs1[j] += g * y[j];
            }
          }
        }
        bf16_t* drow = reinterpret_cast<bf16_t*>(a.dx) + (((long)n * a.Hp + (h0 + q - 1)) * WP) * 16 + 4 * gq;
#pragma unroll
        for (int pt = 0; pt < WP / 16; ++pt) {
          float v[4] = {dacc[pt][0], dacc[pt][1], dacc[pt][2], dacc[pt][3]};
          dstore4<bf16_t>(drow + (16 * pt + r) * 16, v, false);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    u = un; q = qn;
  }

  // ---- flush: the four waves add their accumulators in LDS (wave order: deterministic), the block stores one partial image.
  // D fragment: lane (r, gq) holds D[a = 4gq + j][b = r]
  float* img = reinterpret_cast<float*>(smem);
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int ca = 0; ca < CA16; ++ca)
#pragma unroll
          for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* p = img + (k * CA16 * 16 + ca * 16 + 4 * gq + j) * (CB16 * 16) + cb * 16 + r;
              *p = (w == 0 ? 0.f : *p) + acc[k][ca][cb][j];
            }
      if constexpr (X2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* p = img + (NT * 16 + 4 * gq + j) * 16 + r;
          *p = (w == 0 ? 0.f : *p) + acc2[j];
        }
      }
    }
    __syncthreads();
  }
  if constexpr (ST) {
    // per-channel sums: the 16 pixel-lanes of a row (DPP), then the four waves in order through LDS (behind the partial image)
    float* sb = img + WSIZE;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs0[j] = row16_sum(bs0[j]); bs1[j] = row16_sum(bs1[j]); }
    if (r == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sb[wv * 32 + 4 * gq + j] = bs0[j]; sb[wv * 32 + 16 + 4 * gq + j] = bs1[j]; }
    }
    __syncthreads();
    if (t < 32) a.bn_part[(long)blockIdx.x * 32 + t] = (sb[t] + sb[32 + t]) + (sb[64 + t] + sb[96 + t]);
  }
  float* dst = a.part + (long)blockIdx.x * WSIZE;
  for (int i = t; i < WSIZE / 4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(img)[i];
}

template <int CA16, bool PRO_P, bool PRO_G, bool DG, bool X2, bool ST = false>
static int launch_wstream_t(const WStreamArgs& a, int gx, hipStream_t s) {
  constexpr int KS = 4, S = 2, WP = 32, CB16 = 1;
  constexpr int WL = S * (WP - 1) + KS;
  constexpr size_t lds = 4 * (size_t)((KS + S) * WL * CB16 * 32 + WP * CA16 * 32 + (X2 ? WP * 32 : 0) + (ST ? WP * CA16 * 32 : 0));
  auto kern = &wgrad_stream_kernel<KS, S, 1, WP, CA16, CB16, PRO_P, PRO_G, DG, X2, ST>;
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("wgrad_stream: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MMVAE_ERR_HIP; }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, s, a);
  return check_launch("wgrad_stream");
}

static bool wstream_enabled() {
  static const int enabled = [] { const char* e = getenv("MMVAE_WSTREAM"); return e ? atoi(e) : 1; }();
  return enabled != 0;
}
static bool wstream_shape(int dt, const WgradArgs& a) {
  if (!wstream_enabled() || dt != DT_BF16 || a.P_planar || a.G_planar || !a.scratch) return false;
  if (a.ksz != 4 || a.stride != 2 || a.pad != 1 || a.ntaps != 16) return false;
  if (a.Wp != 32 || a.Wg != 64 || a.Hg != 2 * a.Hp || a.Cb != 16 || (a.Ca != 16 && a.Ca != 32)) return false;
  if (a.Cb_valid != a.Cb || a.Ca_valid != a.Ca) return false;
  for (int t = 0; t < 16; ++t) if (a.tap_off[t] != t) return false;
  return true;
}
static int wstream_fill(const WgradArgs& a, WStreamArgs& b, bool x2 = false) {
  memset(&b, 0, sizeof(b));
  b.P = a.P; b.G = a.G; b.part = a.scratch;
  b.proP_scale = a.proP_scale; b.proP_shift = a.proP_shift; b.proP_relu = a.proP_relu;
  b.proG_scale = a.proG_scale; b.proG_shift = a.proG_shift; b.proG_relu = a.proG_relu;
  b.N = a.N; b.Hp = a.Hp; b.Hg = a.Hg; b.Wg = a.Wg;
  b.HS = a.Hp % 16 == 0 ? 16 : a.Hp;
  b.nunits = a.N * (a.Hp / b.HS);
  int gx = 512;                                             // two 4-wave blocks per CU
  while (gx > 8 && (long)gx * 4 > b.nunits) gx -= 8;
  const int wsize = (16 + (x2 ? 1 : 0)) * a.Ca * a.Cb;
  if ((size_t)gx * wsize * 4 > kWgradScratchBytes) return 0;
  return gx;
}
// dW2 (optional, X2 passes): the 1x1 conv's weight (16 outputs = x2 channels, 16 inputs = P channels, row-major) receives the 17th image
static int wstream_reduce(const WgradArgs& a, int gx, hipStream_t s, bool x2 = false, float* dW2 = nullptr, float scale2 = 1.f) {
  WgradReduceArgs u; memset(&u, 0, sizeof(u));
  u.part = a.scratch; u.dW = a.dW; u.Ca = a.Ca; u.Cb = a.Cb; u.ntaps = a.ntaps; u.nparts = gx;
  u.Ca_valid = a.Ca_valid; u.Cb_valid = a.Cb_valid; u.sA = a.sA; u.sB = a.sB; u.scale = a.scale;
  u.part_stride = (long)(a.ntaps + (x2 ? 1 : 0)) * a.Ca * a.Cb;
  for (int t = 0; t < 25; ++t) u.tap_off[t] = a.tap_off[t];
  const int rc = launch_wgrad_reduce(u, s);
  if (rc < 0 || !x2 || !dW2) return rc;
  WgradReduceArgs v; memset(&v, 0, sizeof(v));
  v.part = a.scratch + (long)a.ntaps * a.Ca * a.Cb; v.part_stride = u.part_stride; v.dW = dW2; v.Ca = 16; v.Cb = a.Ca; v.ntaps = 1; v.nparts = gx;
  v.Ca_valid = 16; v.Cb_valid = a.Ca; v.sA = a.Ca; v.sB = 1; v.scale = scale2;
  return launch_wgrad_reduce(v, s);
}

// Returns 1 when the launch was taken (kernel + reduce enqueued), 0 when the shape is not this kernel's, <0 on error.  MMVAE_WSTREAM=0: off
int try_wgrad_stream(int dt, const WgradArgs& a, hipStream_t s) {
  if (!wstream_shape(dt, a)) return 0;
  WStreamArgs b;
  const int gx = wstream_fill(a, b);
  if (gx <= 0) return 0;
  const bool pp = a.proP_scale != nullptr, pg = a.proG_scale != nullptr;
  int rc;
  if (a.Ca == 16) rc = pp ? (pg ? launch_wstream_t<1, true, true, false, false>(b, gx, s) : launch_wstream_t<1, true, false, false, false>(b, gx, s))
                          : (pg ? launch_wstream_t<1, false, true, false, false>(b, gx, s) : launch_wstream_t<1, false, false, false, false>(b, gx, s));
  else rc = pp ? (pg ? launch_wstream_t<2, true, true, false, false>(b, gx, s) : launch_wstream_t<2, true, false, false, false>(b, gx, s))
               : (pg ? launch_wstream_t<2, false, true, false, false>(b, gx, s) : launch_wstream_t<2, false, false, false, false>(b, gx, s));
  if (rc < 0) return rc;
  const int rc2 = wstream_reduce(a, gx, s);
  return rc2 < 0 ? rc2 : 1;
}

// Weight gradient AND data gradient (w.r.t. P) of a 16 -> 16 channel k4 s2 layer in one pass over G (MMVAE_WSTREAM_DG=0: off).
//   wd: the conv's packed down form [Ca][16][Cb]; dx [N][Hp][32][16]; x2 / w2 (optional): the 1x1 shortcut's operand on the P grid and its
//   packed [Ca][16] matrix; bn_part (optional, needs the prologue and no x2): [blocks][2][16] BatchNorm-backward sums of P's BatchNorm.
//   Returns the number of blocks (> 0) when taken, 0 when the shape is not this kernel's, <0 on error.
bool dgrad_wgrad_stream_shape(int dt, const WgradArgs& a) {
  static const int enabled = [] { const char* e = getenv("MMVAE_WSTREAM_DG"); return e ? atoi(e) : 1; }();
  return enabled != 0 && wstream_shape(dt, a) && a.Ca == 16 && !a.proG_scale;
}
int try_dgrad_wgrad_stream(int dt, const WgradArgs& a, const void* wd, void* dx, const void* x2, const void* w2, float* bn_part, hipStream_t s,
                           float* dW2, float scale2) {
  if (!dgrad_wgrad_stream_shape(dt, a) || !wd || !dx || ((x2 != nullptr) != (w2 != nullptr))) return 0;
  if (bn_part && (x2 || !a.proP_scale)) { set_error("dgrad_wgrad_stream: BatchNorm sums need the prologue'd P and no second source"); return MMVAE_ERR_ARG; }
  WStreamArgs b;
  const int gx = wstream_fill(a, b, x2 != nullptr);
  if (gx <= 0) return 0;
  b.wd = wd; b.dx = dx; b.x2 = x2; b.w2 = w2; b.bn_part = bn_part;
  const bool pp = a.proP_scale != nullptr;
  int rc;
  if (x2) rc = pp ? launch_wstream_t<1, true, false, true, true>(b, gx, s) : launch_wstream_t<1, false, false, true, true>(b, gx, s);
  else if (bn_part) rc = launch_wstream_t<1, true, false, true, false, true>(b, gx, s);
  else rc = pp ? launch_wstream_t<1, true, false, true, false>(b, gx, s) : launch_wstream_t<1, false, false, true, false>(b, gx, s);
  if (rc < 0) return rc;
  const int rc2 = wstream_reduce(a, gx, s, x2 != nullptr, dW2, scale2);
  return rc2 < 0 ? rc2 : gx;             // taken: the number of blocks = rows of bn_part
}

}  // namespace mmvae
