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
  // (P2) a second P operand on the same grid whose layer reads the SAME G pixels through its only tap, the centre one: the 1x1 stride-S
  // shortcut next to a 3x3 stride-S conv (encoder.layer1: conv1 and downsample.0 from one read of the block input).  Its weight gradient
  // dW2[a][b] = sum P2[n,h,w,a] * G[n, S*h, S*w, b] is one more image of the partial, [NT + 1][CA][CB]
  const void* P2;
  // (JG) G is the dy of a BatchNorm after a residual join, evaluated on load instead of read: G = jA[c] * dout + jB[c] * Jy + jC[c], dout
  // (a.G) the gradient w.r.t. the join's output ALREADY masked by its ReLU, Jy the branch's pre-BatchNorm output -- the bn_bwd_apply launch
  // that would write dy (and the read of it here) disappears
  const void* Jy; const float* jA; const float* jB; const float* jC;
};

// KS x KS taps, stride S, padding PAD; WP = P row width: one step = 32 P pixels = one MFMA K-step = ONE row of 32 or TWO rows of 16
// (RP rows; they are contiguous in memory, their G rows are S rows apart in the ring); CA16, CB16: channel tiles of P and G; NW waves
// per block (each with its own ring: a 3x3 stride-2 ring of 32-channel rows is 21 KB, two waves per block keep six waves per CU).
// DG: the same pass also produces the data gradient w.r.t. P (the G rows a P row needs for its weight gradient are exactly the rows
// its data gradient needs): the dy tensor is read ONCE for both.  Two taps (2 x 16 G channels) make one 16x16x32 MFMA K-step; the A
// operand is the conv's packed "down" weights (8 fragments per 16 P channels: in registers for 16 channels, in block-shared LDS for
// 32), B fragments are plain 16-byte reads of ring pixels.
// X2 (with DG): + the 1x1 shortcut's share, x2 (x) w2, from a second row staged per step (the block-input gradient in one kernel),
// and that 1x1 conv's own weight gradient x2^T (x) pro(P) as one more image of the partial.
// ST (with DG and PRO_P): P is a pre-BatchNorm tensor whose BN+ReLU is the prologue; the pass also reduces that BatchNorm's backward sums
// (sum g, sum g*P over all pixels, g = dx masked by the ReLU) from the data gradient it has just computed -- the bn_bwd_reduce launch
// that would re-read dx and P disappears.
template <int KS, int S, int PAD, int WP, int CA16, int CB16, bool PRO_P, bool PRO_G, bool DG, bool X2, bool ST, int NW = 4, bool P2 = false, bool JG = false>
__global__ __launch_bounds__(64 * NW, 2) void wgrad_stream_kernel(WStreamArgs a) {
  static_assert(!JG || (DG && !PRO_G), "the join-gradient loader rides on the fused data-gradient pass");
  static_assert(WP == 32 || WP == 16, "32 P pixels per step: one row of 32 or two rows of 16");
  static_assert(!ST || (DG && PRO_P && !X2 && CA16 == 1), "BatchNorm sums ride on the data gradient of a prologue'd 16-channel P");
  static_assert(!DG || (CB16 == 1 && KS == 4 && S == 2), "fused data gradient: 16 G channels, 4x4 taps, stride 2");
  static_assert(!X2 || DG, "the second source rides on the data gradient");
  static_assert(!P2 || (!DG && !X2 && !ST && PAD < KS), "the centre-tap companion rides on the plain weight-gradient pass");
  constexpr int RP = 32 / WP;                              // P rows per step
  constexpr int CA = CA16 * 16, CB = CB16 * 16;
  constexpr int CAB = CA16 * 32, CBB = CB16 * 32;          // bytes per P / G pixel
  constexpr int WL = S * (WP - 1) + KS;                    // G columns a P row reaches: -PAD .. S*(WP-1)+KS-1-PAD
  constexpr int ROWB = WL * CBB;                           // bytes per LDS G row
  constexpr int GR = S * RP;                               // G rows arriving per step
  constexpr int NSLOT = S * (RP - 1) + KS + GR;            // ring: the rows being multiplied + the rows arriving
  static_assert(KS - S <= GR, "the priming step brings the rows above the first step's own");
  constexpr int PB = 32 * CAB;                             // bytes of a step's P rows
  constexpr int X2B = X2 ? 32 * 32 : 0;                    // bytes of the x2 rows (16 channels)
  constexpr int RAWB = ST ? PB : 0;                        // raw copy of the P rows (before the prologue) for the BatchNorm sums
  constexpr int P2B = P2 ? PB : 0;                         // the companion layer's P rows
  constexpr int WAVE_LDS = NSLOT * ROWB + PB + X2B + RAWB + P2B;
  constexpr int NT = KS * KS;
  constexpr int WSIZE = NT * CA * CB + (X2 ? 16 * CA : 0) + (P2 ? CA * CB : 0);   // floats of a partial image (X2: + the 1x1 conv's [16][CA]; P2: + [CA][CB])
  constexpr bool WD_LDS = DG && CA16 > 1;                  // the data gradient's A fragments live in block-shared LDS
  constexpr int WDB = WD_LDS ? CA16 * (NT / 2) * 1024 : 0;
  // (the flush image, WSIZE floats + 128 floats of BatchNorm sums, aliases the rings: the launcher allocates the larger of the two)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* ring = smem + WDB + wv * WAVE_LDS;
  char* prow = ring + NSLOT * ROWB;
  char* x2row = prow + PB;
  char* rawrow = x2row + X2B;
  char* prow2 = rawrow + RAWB;
  constexpr int Wg = S * WP;                               // G row width (the launcher checks it)
  constexpr int grow_bytes = Wg * CBB;                     // bytes of a G row in memory
  constexpr int GV = (GR * grow_bytes + 1023) / 1024;      // 16-byte vectors per lane for the GR rows of a step
  constexpr int PV = PB / 1024;                            // ... for the P rows
  static_assert(PB % 1024 == 0, "P rows = whole wave loads");

  // ---- border columns are zero for the whole kernel: columns [0, PAD) and [PAD + Wg, WL) of every slot
  for (int sl = 0; sl < NSLOT; ++sl) {
    for (int i = lane; i < (PAD * CBB) / 16; i += 64) reinterpret_cast<Vec16*>(ring + sl * ROWB)[i] = Vec16{{0, 0, 0, 0}};
    for (int i = lane; i < ((WL - PAD) * CBB - grow_bytes) / 16; i += 64)
      reinterpret_cast<Vec16*>(ring + sl * ROWB + PAD * CBB + grow_bytes)[i] = Vec16{{0, 0, 0, 0}};
  }
  if constexpr (WD_LDS) {
    // A[row a = 16 at + r][k = 8gq ..]: taps (kh, 2kp) and (kh, 2kp + 1), 16 G channels each = 64 contiguous bytes of wd[a][tap][b]
    for (int i = t; i < CA16 * (NT / 2) * 64; i += 64 * NW) {
      const int ln = i & 63, pr = (i >> 6) % (NT / 2), at = (i >> 6) / (NT / 2);
      reinterpret_cast<Vec16*>(smem)[i] =
          *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.wd) + (16 * at + (ln & 15)) * (NT * 32) + pr * 64 + (ln >> 4) * 16);
    }
    __syncthreads();
  }
  // ---- prologue coefficients of the channels this lane stages (8 consecutive channels per 16-byte vector)
  float psc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, psh[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gsc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gsh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (PRO_P) {
    const int c = (lane % (CAB / 16)) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { psc[j] = a.proP_scale[c + j]; psh[j] = a.proP_shift[c + j]; }
  }
  // (G_TAB, the centre-tap companion form: G's prologue coefficients live in an LDS table behind the rings, read where a row is committed -- 16
  // registers fewer across the loop: with them the form spilled 8 registers at the 256 limit)
  constexpr bool G_TAB = PRO_G && P2;
  float* gtab = reinterpret_cast<float*>(smem + WDB + NW * WAVE_LDS + (JG ? 3 * CB * 4 : 0));
  if constexpr (PRO_G && !G_TAB) {
    const int c = (lane % (CBB / 16)) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gsc[j] = a.proG_scale[c + j]; gsh[j] = a.proG_shift[c + j]; }
  }
  if constexpr (G_TAB) {
    if (t < CB) { gtab[t] = a.proG_scale[t]; gtab[CB + t] = a.proG_shift[t]; }
    __syncthreads();
  }
  // (JG) the coefficient table [3][CB] sits behind the rings in LDS and is read where a row is committed: 24 registers live for a few
  // instructions instead of the whole loop (the 32-channel second-source form is at the 256-register limit)
  float* jtab = reinterpret_cast<float*>(smem + WDB + NW * WAVE_LDS);
  if constexpr (JG) {
    if (t < CB) { jtab[t] = a.jA[t]; jtab[CB + t] = a.jB[t]; jtab[2 * CB + t] = a.jC[t]; }
    __syncthreads();
  }
  const float p_lo = a.proP_relu ? 0.f : -__builtin_inff(), g_lo = a.proG_relu ? 0.f : -__builtin_inff();
  // ---- fragment offsets (tile-invariant): k-slice gq = the step's P pixels 8gq .. 8gq+7, two 4-pixel blocks; lane i = r supplies the
  // address of pixel (i >> 2) of the block, channel quad (i & 3).  With two rows per step the slice's row is (8 gq) / WP.
  int offA[2], offB[2], offX[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int p = 8 * gq + 4 * blk + (r >> 2);
    offA[blk] = p * CAB + (r & 3) * 8;
    offX[blk] = p * 32 + (r & 3) * 8;
    offB[blk] = (S * (p % WP)) * CBB + (r & 3) * 8;        // + kw * CBB + slot * ROWB (+ 32 per channel tile)
  }
  const int lrow = (8 * gq) / WP;
  f32x4 acc[NT][CA16][CB16];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int ca = 0; ca < CA16; ++ca)
#pragma unroll
      for (int cb = 0; cb < CB16; ++cb) acc[k][ca][cb] = (f32x4){0, 0, 0, 0};

  f32x4 accD[P2 ? CA16 : 1][P2 ? CB16 : 1];                 // P2: the centre-tap companion's weight gradient
#pragma unroll
  for (int ca = 0; ca < (P2 ? CA16 : 1); ++ca)
#pragma unroll
    for (int cb = 0; cb < (P2 ? CB16 : 1); ++cb) accD[ca][cb] = (f32x4){0, 0, 0, 0};
  f32x4 acc2[CA16];                                         // X2: the 1x1 conv's weight gradient [x2 channel][P channel]
#pragma unroll
  for (int ca = 0; ca < CA16; ++ca) acc2[ca] = (f32x4){0, 0, 0, 0};
  float bsc[4] = {0, 0, 0, 0}, bsh[4] = {0, 0, 0, 0}, bs0[4] = {0, 0, 0, 0}, bs1[4] = {0, 0, 0, 0};
  if constexpr (ST) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { bsc[j] = a.proP_scale[4 * gq + j]; bsh[j] = a.proP_shift[4 * gq + j]; }
  }
  Vec16 wdA[(DG && !WD_LDS) ? NT / 2 : 1], w2A[CA16];
#pragma unroll
  for (int at = 0; at < CA16; ++at) w2A[at] = Vec16{{0, 0, 0, 0}};
  if constexpr (DG) {
    if constexpr (!WD_LDS) {
#pragma unroll
      for (int pr = 0; pr < NT / 2; ++pr)
        wdA[pr] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.wd) + r * (NT * 32) + pr * 64 + gq * 16);
    }
    if (X2 && gq < 2) {
#pragma unroll
      for (int at = 0; at < CA16; ++at) w2A[at] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.w2) + (16 * at + r) * 32 + gq * 16);
    }
  }
  const bf16_t* __restrict__ Pm = reinterpret_cast<const bf16_t*>(a.P);
  const bf16_t* __restrict__ Gm = reinterpret_cast<const bf16_t*>(a.G);
  // XCD-aware unit walk: block b runs on XCD b % 8; every XCD sweeps its own eighth of the units front to back
  const int nblk = gridDim.x;
  int u_first, u_step, u_end;
  if ((nblk & 7) == 0) {
    const int per = (a.nunits + 7) >> 3;
    const int lo = (blockIdx.x & 7) * per;
    u_first = lo + (blockIdx.x >> 3) * NW + wv; u_step = (nblk >> 3) * NW; u_end = min(a.nunits, lo + per);
  } else { u_first = blockIdx.x * NW + wv; u_step = nblk * NW; u_end = a.nunits; }
  const int nstrips = a.Hp / a.HS;
  const int nq = a.HS / RP + 1;                             // steps of a unit: the priming step + one per RP P rows

  // registers of the step in flight
  // (the P / P2 / x2 rows in flight are native vectors: as `Vec16` -- a struct around an array -- rows that are only copied, not unpacked, stayed
  // in scratch memory in every form without a P prologue: global load, wait, scratch store, scratch reload, LDS store -- no prefetch, 32-80 bytes
  // of scratch per lane and step)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  Vec16 gv[GV], jv[JG ? GV : 1];
  u32x4 pv[PV], pv2[P2 ? PV : 1], xv = (u32x4){0, 0, 0, 0};
  // LATE_P (the centre-tap companion form: 40 accumulator fragments; the 32-channel second-source form): the next step's P rows are requested AFTER
  // the step's MFMA section, not before it -- live across that section they do not fit 256 registers and the compiler parked them in scratch memory (load, wait, store to
  // scratch, reload: no prefetch at all and 94 MB of scratch write-backs per launch in the round-4 profile)
  constexpr bool LATE_P = P2 || (X2 && CA16 > 1);
  auto issueP = [&](int u, int q) {
    // (no branch: on the priming step, q = 0, the strip's first rows are loaded and dropped -- with the loads under `if (q > 0)` the arrays
    // stayed in scratch memory in every 3x3 form: load, wait, scratch store, reload)
    const int n = u / nstrips, h0 = (u - n * nstrips) * a.HS;
    const int hrow = h0 + RP * (q > 0 ? q - 1 : 0);
#pragma unroll
    for (int k = 0; k < PV; ++k)
      pv[k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Pm) + (((long)n * a.Hp + hrow) * WP) * CAB + (lane + 64 * k) * 16);
    if constexpr (X2) xv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.x2) + (((long)n * a.Hp + hrow) * WP) * 32 + lane * 16);
    if constexpr (P2) {
#pragma unroll
      for (int k = 0; k < PV; ++k)
        pv2[k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.P2) + (((long)n * a.Hp + hrow) * WP) * CAB + (lane + 64 * k) * 16);
    }
  };
  auto issue = [&](int u, int q) {
    const int n = u / nstrips, h0 = (u - n * nstrips) * a.HS;
    const int hrow = h0 + RP * (q - 1);                     // first P row of the step (q = 0: the rows above the strip's first step)
    // G rows [top - GR + 1, top], top = last row the step's last P row reaches
    const int top = S * (hrow + RP - 1) - PAD + KS - 1;
#pragma unroll
    for (int k = 0; k < GV; ++k) {
      const int byte = (lane + 64 * k) * 16;
      const int rr = byte / grow_bytes, off = byte - rr * grow_bytes;
      const int row = top - GR + 1 + rr;
      gv[k] = Vec16{{0, 0, 0, 0}};
      if constexpr (JG) jv[k] = Vec16{{0, 0, 0, 0}};
      if (rr < GR && row >= 0 && row < a.Hg) {
        gv[k] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(Gm) + (((long)n * a.Hg + row) * Wg) * CBB + off);
        if constexpr (JG) jv[k] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.Jy) + (((long)n * a.Hg + row) * Wg) * CBB + off);
      }
    }
    if constexpr (!LATE_P) issueP(u, q);
  };
  auto commit = [&](int u, int q) {
    const int n = u / nstrips, h0 = (u - n * nstrips) * a.HS;
    const int top = S * (h0 + RP * (q - 1) + RP - 1) - PAD + KS - 1;
#pragma unroll
    for (int k = 0; k < GV; ++k) {
      const int byte = (lane + 64 * k) * 16;
      const int rr = byte / grow_bytes, off = byte - rr * grow_bytes;
      if (rr < GR) {
        const int row = top - GR + 1 + rr;
        Vec16 v = gv[k];
        if constexpr (PRO_G) if (row >= 0 && row < a.Hg) {
          float f[8];
          Elem<bf16_t>::unpack(v, f);
          if constexpr (G_TAB) {
            const int c = (lane % (CBB / 16)) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 cs = *reinterpret_cast<const float4*>(gtab + c + 4 * h), cb = *reinterpret_cast<const float4*>(gtab + CB + c + 4 * h);
              f[4 * h + 0] = fmaxf(f[4 * h + 0] * cs.x + cb.x, g_lo);
              f[4 * h + 1] = fmaxf(f[4 * h + 1] * cs.y + cb.y, g_lo);
              f[4 * h + 2] = fmaxf(f[4 * h + 2] * cs.z + cb.z, g_lo);
              f[4 * h + 3] = fmaxf(f[4 * h + 3] * cs.w + cb.w, g_lo);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * gsc[j] + gsh[j], g_lo);
          }
          v = Elem<bf16_t>::pack(f);
        }
        if constexpr (JG) if (row >= 0 && row < a.Hg) {        // (rows outside the map stay zero: they are the conv's padding)
          float f[8], y[8];
          Elem<bf16_t>::unpack(v, f);
          Elem<bf16_t>::unpack(jv[k], y);
          const int c = (lane % (CBB / 16)) * 8;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float4 cA = *reinterpret_cast<const float4*>(jtab + c + 4 * h), cB = *reinterpret_cast<const float4*>(jtab + CB + c + 4 * h),
                         cC = *reinterpret_cast<const float4*>(jtab + 2 * CB + c + 4 * h);
            f[4 * h + 0] = cA.x * f[4 * h + 0] + cB.x * y[4 * h + 0] + cC.x;
            f[4 * h + 1] = cA.y * f[4 * h + 1] + cB.y * y[4 * h + 1] + cC.y;
            f[4 * h + 2] = cA.z * f[4 * h + 2] + cB.z * y[4 * h + 2] + cC.z;
            f[4 * h + 3] = cA.w * f[4 * h + 3] + cB.w * y[4 * h + 3] + cC.w;
          }
          v = Elem<bf16_t>::pack(f);
        }
        const int slot = (row + 4 * NSLOT) % NSLOT;
        *reinterpret_cast<Vec16*>(ring + slot * ROWB + PAD * CBB + off) = v;
      }
    }
    if (q > 0) {
#pragma unroll
      for (int k = 0; k < PV; ++k) {
        u32x4 v = pv[k];
        if constexpr (ST) *reinterpret_cast<u32x4*>(rawrow + (lane + 64 * k) * 16) = v;
        if constexpr (PRO_P) {
          float f[8];
          Elem<bf16_t>::unpack(__builtin_bit_cast(Vec16, v), f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * psc[j] + psh[j], p_lo);
          v = __builtin_bit_cast(u32x4, Elem<bf16_t>::pack(f));
        }
        *reinterpret_cast<u32x4*>(prow + (lane + 64 * k) * 16) = v;
      }
      if constexpr (X2) *reinterpret_cast<u32x4*>(x2row + lane * 16) = xv;
      if constexpr (P2) {
#pragma unroll
        for (int k = 0; k < PV; ++k) *reinterpret_cast<u32x4*>(prow2 + (lane + 64 * k) * 16) = pv2[k];
      }
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
      const int hrow = h0 + RP * (q - 1);
      const int first = S * hrow - PAD;                     // G row of tap row kh = 0 of the step's first P row
      Vec16 af[CA16];
#pragma unroll
      for (int ca = 0; ca < CA16; ++ca) af[ca] = FragOps<bf16_t>::load(prow, offA[0] + 32 * ca, offA[1] + 32 * ca);
      Vec16 af2[P2 ? CA16 : 1];
      if constexpr (P2) {
#pragma unroll
        for (int ca = 0; ca < CA16; ++ca) af2[ca] = FragOps<bf16_t>::load(prow2, offA[0] + 32 * ca, offA[1] + 32 * ca);
      }
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        // (two rows per step: the lane's k-slice belongs to P row lrow, whose G rows are S further down the ring)
        const char* rowp = ring + ((first + kh + S * (RP > 1 ? lrow : 0) + 4 * NSLOT) % NSLOT) * ROWB;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
          for (int cb = 0; cb < CB16; ++cb) {
            const Vec16 bf = FragOps<bf16_t>::load(rowp, offB[0] + kw * CBB + 32 * cb, offB[1] + kw * CBB + 32 * cb);
#pragma unroll
            for (int ca = 0; ca < CA16; ++ca) acc[kh * KS + kw][ca][cb] = mma_bf16(af[ca], bf, acc[kh * KS + kw][ca][cb]);
            if constexpr (P2) {
              if (kh == PAD && kw == PAD) {                  // the centre tap reads G[S*h][S*w]: the 1x1 stride-S layer's only pixel
#pragma unroll
                for (int ca = 0; ca < CA16; ++ca) accD[ca][cb] = mma_bf16(af2[ca], bf, accD[ca][cb]);
              }
            }
          }
        }
      }
      if constexpr (X2) {
        // x2^T (16 x 32 pixels) times the prologue'd P rows (32 pixels x CA): the same fragment form as af, from the x2 rows
        const Vec16 ax = FragOps<bf16_t>::load(x2row, offX[0], offX[1]);
#pragma unroll
        for (int ca = 0; ca < CA16; ++ca) acc2[ca] = mma_bf16(ax, af[ca], acc2[ca]);
      }
      if constexpr (DG) {
        // D[a][pixel]: lane (r = pixel of the 16-pixel tile, gq) ends with channels a = 16 at + 4gq .. + 3; tile pt = pixels 16pt .. of the step
        f32x4 dacc[CA16][2];
#pragma unroll
        for (int at = 0; at < CA16; ++at)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) dacc[at][pt] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            const char* rowp = ring + ((first + kh + S * ((16 * pt) / WP) + 4 * NSLOT) % NSLOT) * ROWB;
            const int col0 = (16 * pt) % WP;
#pragma unroll
            for (int kp = 0; kp < KS / 2; ++kp) {
              const Vec16 b = *reinterpret_cast<const Vec16*>(rowp + (S * (col0 + r) + 2 * kp + (gq >> 1)) * CBB + (gq & 1) * 16);
#pragma unroll
              for (int at = 0; at < CA16; ++at) {
                Vec16 wA;
                if constexpr (WD_LDS) wA = reinterpret_cast<const Vec16*>(smem)[(at * (NT / 2) + kh * (KS / 2) + kp) * 64 + lane];
                else wA = wdA[kh * (KS / 2) + kp];
                dacc[at][pt] = mma_bf16(wA, b, dacc[at][pt]);
              }
            }
          }
        }
        if constexpr (X2) {
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            const Vec16 b = *reinterpret_cast<const Vec16*>(x2row + (16 * pt + r) * 32 + (gq & 1) * 16);     // k >= 16: w2A is zero there
#pragma unroll
            for (int at = 0; at < CA16; ++at) dacc[at][pt] = mma_bf16(w2A[at], b, dacc[at][pt]);
          }
        }
        if constexpr (ST) {
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            const uint2 yr = *reinterpret_cast<const uint2*>(rawrow + (16 * pt + r) * CAB + gq * 8);      // P[pixel][4gq .. 4gq+3]
            const float y[4] = {__uint_as_float(yr.x << 16), __uint_as_float(yr.x & 0xffff0000u), __uint_as_float(yr.y << 16), __uint_as_float(yr.y & 0xffff0000u)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = (y[j] * bsc[j] + bsh[j] > 0.f) ? dacc[0][pt][j] : 0.f;
              bs0[j] += g; bs1[j] += g * y[j];
            }
          }
        }
        if constexpr (CA16 == 1) {
          // a pixel's 32 bytes come from its four gq lanes: whole 32-byte sectors
          bf16_t* drow = reinterpret_cast<bf16_t*>(a.dx) + (((long)n * a.Hp + hrow) * WP) * CA + 4 * gq;
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            float v[4] = {dacc[0][pt][0], dacc[0][pt][1], dacc[0][pt][2], dacc[0][pt][3]};
            dstore4<bf16_t>(drow + (16 * pt + r) * CA, v, false);
          }
        } else {
          // 32 channels: a lane's two quads are 32 bytes apart -- through the (now free) P rows in LDS, then whole 16-byte vectors
          bf16_t* stg = reinterpret_cast<bf16_t*>(prow) + 4 * gq;
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int at = 0; at < CA16; ++at) {
              float v[4] = {dacc[at][pt][0], dacc[at][pt][1], dacc[at][pt][2], dacc[at][pt][3]};
              dstore4<bf16_t>(stg + (16 * pt + r) * CA + 16 * at, v, false);
            }
          char* dbase = reinterpret_cast<char*>(a.dx) + (((long)n * a.Hp + hrow) * WP) * CAB;
#pragma unroll
          for (int k = 0; k < PV; ++k) *reinterpret_cast<Vec16*>(dbase + (lane + 64 * k) * 16) = *reinterpret_cast<const Vec16*>(prow + (lane + 64 * k) * 16);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LATE_P) { if (un < u_end) issueP(un, qn); }
    u = un; q = qn;
  }

  // ---- flush: the waves add their accumulators in LDS (wave order: deterministic), the block stores one partial image.
  // D fragment: lane (r, gq) holds D[a = 4gq + j][b = r]
  float* img = reinterpret_cast<float*>(smem);
  __syncthreads();
  for (int w = 0; w < NW; ++w) {
    if (wv == w) {
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int ca = 0; ca < CA16; ++ca)
#pragma unroll
          for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* p = img + (k * CA + ca * 16 + 4 * gq + j) * CB + cb * 16 + r;
              *p = (w == 0 ? 0.f : *p) + acc[k][ca][cb][j];
            }
      if constexpr (X2) {
#pragma unroll
        for (int ca = 0; ca < CA16; ++ca)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float* p = img + NT * CA * CB + (4 * gq + j) * CA + ca * 16 + r;
            *p = (w == 0 ? 0.f : *p) + acc2[ca][j];
          }
      }
      if constexpr (P2) {
#pragma unroll
        for (int ca = 0; ca < CA16; ++ca)
#pragma unroll
          for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float* p = img + (NT * CA + ca * 16 + 4 * gq + j) * CB + cb * 16 + r;
              *p = (w == 0 ? 0.f : *p) + accD[ca][cb][j];
            }
      }
    }
    __syncthreads();
  }
  if constexpr (ST) {
    // per-channel sums: the 16 pixel-lanes of a row (DPP), then the waves in order through LDS (behind the partial image)
    float* sb = img + WSIZE;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs0[j] = row16_sum(bs0[j]); bs1[j] = row16_sum(bs1[j]); }
    if (r == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sb[wv * 32 + 4 * gq + j] = bs0[j]; sb[wv * 32 + 16 + 4 * gq + j] = bs1[j]; }
    }
    __syncthreads();
    if (t < 32) {
      float sum = sb[t];
      for (int w = 1; w < NW; ++w) sum += sb[w * 32 + t];
      a.bn_part[(long)blockIdx.x * 32 + t] = sum;
    }
  }
  float* dst = a.part + (long)blockIdx.x * WSIZE;
  for (int i = t; i < WSIZE / 4; i += 64 * NW) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(img)[i];
}

template <int KS, int S, int PAD, int WP, int CA16, int CB16, bool PRO_P, bool PRO_G, bool DG, bool X2, bool ST = false, int NW = 4, bool P2 = false, bool JG = false>
static int launch_wstream_t(const WStreamArgs& a, int gx, hipStream_t s) {
  constexpr int RP = 32 / WP;
  constexpr int WL = S * (WP - 1) + KS;
  constexpr int NSLOT = S * (RP - 1) + KS + S * RP;
  constexpr size_t rings = NW * (size_t)(NSLOT * WL * CB16 * 32 + 32 * CA16 * 32 + (X2 ? 32 * 32 : 0) + (ST ? 32 * CA16 * 32 : 0) + (P2 ? 32 * CA16 * 32 : 0)) +
                           ((DG && CA16 > 1) ? CA16 * (KS * KS / 2) * 1024 : 0) + (JG ? 3 * CB16 * 16 * 4 : 0) + ((PRO_G && P2) ? 2 * CB16 * 16 * 4 : 0);
  constexpr size_t flush = ((size_t)KS * KS * CA16 * 16 * CB16 * 16 + (X2 ? 16 * CA16 * 16 : 0) + (P2 ? CA16 * 16 * CB16 * 16 : 0)) * 4 + 512;
  constexpr size_t lds = rings > flush ? rings : flush;
  static_assert(lds <= 160 * 1024, "one block must fit the CU's LDS");
  auto kern = &wgrad_stream_kernel<KS, S, PAD, WP, CA16, CB16, PRO_P, PRO_G, DG, X2, ST, NW, P2, JG>;
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("wgrad_stream: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MMVAE_ERR_HIP; }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx), dim3(64 * NW), lds, s, a);
  return check_launch("wgrad_stream");
}

static bool wstream_enabled() {
  constexpr int enabled = 1;
  return enabled != 0;
}
// The geometries with an instantiation (kind):
//   1  ConvTranspose2d k4 s2 p1 on 32-wide P rows, G 16 channels, P 16 / 32 channels      (decoder.uplayer5.conv2 / .upsample)
//   2  the same on 16-wide P rows (two rows per step)                                       (decoder.uplayer4.conv2 / .upsample)
//   3  Conv2d 3x3 s1 p1, 32 -> 32 channels, 16-wide maps                                    (encoder.layer1.conv2)
//   4  Conv2d 3x3 s2 p1, 32 -> 32 channels, 16-wide P (dy) rows, 32-wide G rows             (encoder.layer1.conv1)
static int wstream_kind(int dt, const WgradArgs& a) {
  if (!wstream_enabled() || dt != DT_BF16 || a.P_planar || a.G_planar || !a.scratch) return 0;
  if (a.ntaps != a.ksz * a.ksz || a.Cb_valid != a.Cb || a.Ca_valid != a.Ca) return 0;
  for (int t = 0; t < a.ntaps; ++t) if (a.tap_off[t] != t) return 0;
  if (a.Wg != a.stride * a.Wp || a.Hg != a.stride * a.Hp || a.Hp != a.Wp) return 0;
  if (a.ksz == 4 && a.stride == 2 && a.pad == 1 && a.Cb == 16 && (a.Ca == 16 || a.Ca == 32)) return a.Wp == 32 ? 1 : (a.Wp == 16 ? 2 : 0);
  if (a.ksz == 3 && a.pad == 1 && a.Wp == 16 && a.Ca == 32 && a.Cb == 32 && !(a.proP_scale && a.proG_scale)) return a.stride == 1 ? 3 : (a.stride == 2 ? 4 : 0);
  return 0;
}
static int wstream_fill(const WgradArgs& a, WStreamArgs& b, int kind, bool x2 = false, bool p2 = false) {
  memset(&b, 0, sizeof(b));
  b.P = a.P; b.G = a.G; b.part = a.scratch;
  b.proP_scale = a.proP_scale; b.proP_shift = a.proP_shift; b.proP_relu = a.proP_relu;
  b.proG_scale = a.proG_scale; b.proG_shift = a.proG_shift; b.proG_relu = a.proG_relu;
  b.N = a.N; b.Hp = a.Hp; b.Hg = a.Hg; b.Wg = a.Wg;
  b.HS = a.Hp % 16 == 0 ? 16 : a.Hp;
  b.nunits = a.N * (a.Hp / b.HS);
  const int nw = kind == 4 ? 2 : 4;
  int gx = kind == 4 ? 768 : 512;                           // two 4-wave (three 2-wave) blocks per CU
  while (gx > 8 && (long)gx * nw > b.nunits) gx -= 8;
  const int wsize = (a.ntaps + ((x2 || p2) ? 1 : 0)) * a.Ca * a.Cb;
  if ((size_t)gx * wsize * 4 > kWgradScratchBytes) return 0;
  return gx;
}
// dW2 (optional, X2 passes): the 1x1 conv's weight (16 outputs = x2 channels, Ca inputs = P channels, row-major) receives the extra image
static int wstream_reduce(const WgradArgs& a, int gx, hipStream_t s, bool x2 = false, float* dW2 = nullptr, float scale2 = 1.f) {
  WgradReduceArgs u; memset(&u, 0, sizeof(u));
  u.part = a.scratch; u.dW = a.dW; u.Ca = a.Ca; u.Cb = a.Cb; u.ntaps = a.ntaps; u.nparts = gx;
  u.Ca_valid = a.Ca_valid; u.Cb_valid = a.Cb_valid; u.sA = a.sA; u.sB = a.sB; u.scale = a.scale;
  u.part_stride = (long)a.ntaps * a.Ca * a.Cb + (x2 ? 16L * a.Ca : 0);
  for (int t = 0; t < 25; ++t) u.tap_off[t] = a.tap_off[t];
  if (a.defer && !x2) { *a.defer = u; return 0; }
  const int rc = launch_wgrad_reduce(u, s);
  if (rc < 0 || !x2 || !dW2) return rc;
  WgradReduceArgs v; memset(&v, 0, sizeof(v));
  v.part = a.scratch + (long)a.ntaps * a.Ca * a.Cb; v.part_stride = u.part_stride; v.dW = dW2; v.Ca = 16; v.Cb = a.Ca; v.ntaps = 1; v.nparts = gx;
  v.Ca_valid = 16; v.Cb_valid = a.Ca; v.sA = a.Ca; v.sB = 1; v.scale = scale2;
  return launch_wgrad_reduce(v, s);
}

// whether try_wgrad_stream takes this weight gradient (callers that hand it a smaller scratch or defer its reduce ask first)
bool wgrad_stream_shape(int dt, const WgradArgs& a) {
  return wstream_kind(dt, a) != 0 && !(a.proP_scale && a.proG_scale);
}

// Returns 1 when the launch was taken (kernel + reduce enqueued), 0 when the shape is not this kernel's, <0 on error.  MMVAE_WSTREAM=0: off
int try_wgrad_stream(int dt, const WgradArgs& a, hipStream_t s) {
  const int kind = wstream_kind(dt, a);
  if (!kind) return 0;
  WStreamArgs b;
  const int gx = wstream_fill(a, b, kind);
  if (gx <= 0) return 0;
  const bool pp = a.proP_scale != nullptr, pg = a.proG_scale != nullptr;
  int rc = 0;
#define MMVAE_WS(KS, S, WP, CA, CB, NW)                                                                                      \
  rc = pp ? (pg ? 0 : launch_wstream_t<KS, S, 1, WP, CA, CB, true, false, false, false, false, NW>(b, gx, s))                \
          : (pg ? launch_wstream_t<KS, S, 1, WP, CA, CB, false, true, false, false, false, NW>(b, gx, s)                     \
                : launch_wstream_t<KS, S, 1, WP, CA, CB, false, false, false, false, false, NW>(b, gx, s))
  if (pp && pg) return 0;                                   // one prologue register set
  if (kind == 1) { if (a.Ca == 16) MMVAE_WS(4, 2, 32, 1, 1, 4); else MMVAE_WS(4, 2, 32, 2, 1, 4); }
  else if (kind == 2) { if (a.Ca == 16) MMVAE_WS(4, 2, 16, 1, 1, 4); else MMVAE_WS(4, 2, 16, 2, 1, 4); }
  else if (kind == 3) MMVAE_WS(3, 1, 16, 2, 2, 4);
  else MMVAE_WS(3, 2, 16, 2, 2, 2);
#undef MMVAE_WS
  if (rc < 0) return rc;
  const int rc2 = wstream_reduce(a, gx, s);
  return rc2 < 0 ? rc2 : 1;
}

// encoder.layer1: conv1 (3x3 s2 p1, 32 -> 32) and the 1x1 s2 shortcut read the same block input: both weight gradients from ONE pass over it
// (P = d conv1 output, P2 = d shortcut output, same grid).  Returns 1 when taken, 0 when the shape is not this kernel's, <0 on error.
int try_wgrad_stream_pair(int dt, const WgradArgs& a, const void* P2, float* dW2, float scale2, hipStream_t s) {
  const int kind = wstream_kind(dt, a);
  if (kind != 4 || a.proP_scale || !P2 || !dW2) return 0;
  WStreamArgs b;
  const int gx = wstream_fill(a, b, kind, false, true);
  if (gx <= 0) return 0;
  b.P2 = P2;
  note_launch_bytes((double)a.N * 2.0 * ((double)a.Hp * a.Wp * 2 * a.Ca + (double)a.Hg * a.Wg * a.Cb));
  const int rc = a.proG_scale ? launch_wstream_t<3, 2, 1, 16, 2, 2, false, true, false, false, false, 2, true>(b, gx, s)
                              : launch_wstream_t<3, 2, 1, 16, 2, 2, false, false, false, false, false, 2, true>(b, gx, s);
  if (rc < 0) return rc;
  WgradReduceArgs u; memset(&u, 0, sizeof(u));
  u.part = a.scratch; u.dW = a.dW; u.Ca = a.Ca; u.Cb = a.Cb; u.ntaps = a.ntaps; u.nparts = gx;
  u.Ca_valid = a.Ca_valid; u.Cb_valid = a.Cb_valid; u.sA = a.sA; u.sB = a.sB; u.scale = a.scale;
  u.part_stride = (long)(a.ntaps + 1) * a.Ca * a.Cb;
  for (int t = 0; t < 25; ++t) u.tap_off[t] = a.tap_off[t];
  int rc2 = launch_wgrad_reduce(u, s);
  if (rc2 < 0) return rc2;
  WgradReduceArgs v; memset(&v, 0, sizeof(v));
  v.part = a.scratch + (long)a.ntaps * a.Ca * a.Cb; v.part_stride = u.part_stride; v.dW = dW2; v.Ca = a.Ca; v.Cb = a.Cb; v.ntaps = 1; v.nparts = gx;
  v.Ca_valid = a.Ca; v.Cb_valid = a.Cb; v.sA = a.Cb; v.sB = 1; v.scale = scale2;
  rc2 = launch_wgrad_reduce(v, s);
  return rc2 < 0 ? rc2 : 1;
}

// Weight gradient AND data gradient (w.r.t. P) of a ConvTranspose2d k4 s2 p1 layer with 16 output channels in one pass over G
// (MMVAE_WSTREAM_DG=0: off).
//   wd: the conv's packed down form [Ca][16][Cb]; dx [N][Hp][Wp][Ca]; x2 / w2 (optional): the 1x1 shortcut's operand on the P grid and its
//   packed [Ca][16] matrix; bn_part (optional, needs the prologue, Ca = 16 and no x2): [blocks][2][16] BatchNorm-backward sums of P's BatchNorm.
//   Returns the number of blocks (> 0) when taken, 0 when the shape is not this kernel's, <0 on error.
bool dgrad_wgrad_stream_shape(int dt, const WgradArgs& a) {
  constexpr int enabled = 1;
  const int kind = wstream_kind(dt, a);
  return enabled != 0 && (kind == 1 || kind == 2) && !a.proG_scale;
}
// jg (optional): G is the join's masked output gradient and dy = jg->A * G + jg->B * jg->y + jg->C is evaluated on load (JG above); the
// instantiated forms are the two a DeconvBottleneck on 16-wide maps needs: (Ca 16, prologue, BatchNorm sums) and (Ca 32, second source).
bool dgrad_wgrad_stream_jg_shape(int dt, const WgradArgs& a, bool has_x2, bool bn_sums) {
  if (!dgrad_wgrad_stream_shape(dt, a) || wstream_kind(dt, a) != 2) return false;
  return (a.Ca == 16 && a.proP_scale && bn_sums && !has_x2) || (a.Ca == 32 && !a.proP_scale && has_x2 && !bn_sums);
}
int try_dgrad_wgrad_stream(int dt, const WgradArgs& a, const void* wd, void* dx, const void* x2, const void* w2, float* bn_part, hipStream_t s,
                           float* dW2, float scale2, const JoinGrad* jg) {
  if (!dgrad_wgrad_stream_shape(dt, a) || !wd || !dx || ((x2 != nullptr) != (w2 != nullptr))) return 0;
  if (jg && !dgrad_wgrad_stream_jg_shape(dt, a, x2 != nullptr, bn_part != nullptr)) { set_error("dgrad_wgrad_stream: no join-gradient form of this shape"); return MMVAE_ERR_UNSUPPORTED; }
  if (bn_part && (x2 || !a.proP_scale || a.Ca != 16)) { set_error("dgrad_wgrad_stream: BatchNorm sums need the prologue'd 16-channel P and no second source"); return MMVAE_ERR_ARG; }
  const int kind = wstream_kind(dt, a);
  WStreamArgs b;
  const int gx = wstream_fill(a, b, kind, x2 != nullptr);
  if (gx <= 0) return 0;
  b.wd = wd; b.dx = dx; b.x2 = x2; b.w2 = w2; b.bn_part = bn_part;
  if (jg) { b.Jy = jg->y; b.jA = jg->A; b.jB = jg->B; b.jC = jg->C; }
  note_launch_bytes((double)a.N * 2.0 * ((double)a.Hp * a.Wp * (2 * a.Ca + (x2 ? 16 : 0)) + (double)a.Hg * a.Wg * a.Cb * (jg ? 2 : 1)));   // P, dx, x2, G (+ Jy) (bf16)
  const bool pp = a.proP_scale != nullptr;
  int rc;
#define MMVAE_WSD(WP, CA)                                                                                                    \
  do {                                                                                                                        \
    if (x2) rc = pp ? launch_wstream_t<4, 2, 1, WP, CA, 1, true, false, true, true>(b, gx, s)                                 \
                    : launch_wstream_t<4, 2, 1, WP, CA, 1, false, false, true, true>(b, gx, s);                               \
    else rc = pp ? launch_wstream_t<4, 2, 1, WP, CA, 1, true, false, true, false>(b, gx, s)                                   \
                 : launch_wstream_t<4, 2, 1, WP, CA, 1, false, false, true, false>(b, gx, s);                                 \
  } while (0)
  if (jg) rc = bn_part ? launch_wstream_t<4, 2, 1, 16, 1, 1, true, false, true, false, true, 4, false, true>(b, gx, s)
                       : launch_wstream_t<4, 2, 1, 16, 2, 1, false, false, true, true, false, 4, false, true>(b, gx, s);
  else if (bn_part) rc = kind == 1 ? launch_wstream_t<4, 2, 1, 32, 1, 1, true, false, true, false, true>(b, gx, s)
                              : launch_wstream_t<4, 2, 1, 16, 1, 1, true, false, true, false, true>(b, gx, s);
  else if (kind == 1) { if (a.Ca == 16) MMVAE_WSD(32, 1); else MMVAE_WSD(32, 2); }
  else { if (a.Ca == 16) MMVAE_WSD(16, 1); else MMVAE_WSD(16, 2); }
#undef MMVAE_WSD
  if (rc < 0) return rc;
  const int rc2 = wstream_reduce(a, gx, s, x2 != nullptr, dW2, scale2);
  return rc2 < 0 ? rc2 : gx;             // taken: the number of blocks = rows of bn_part
}

}  // namespace mmvae
