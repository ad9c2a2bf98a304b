// The last up-block's backward in ONE pass over its two branch outputs (decoder.uplayer5 at 64x64: y2, ys = 671 MB each at N = 5120).
//
// Round 2 ran it as  tail_apply_mfma (reads y2, ys; WRITES dy2, dys)  ->  wgrad_stream<DG, ST> over dy2  ->  bn1 apply  ->
// wgrad_stream<DG, X2> over dys: the two dy tensors exist only to be written once and read once (2.7 GB of the step's traffic).
// join_bwd_stream_kernel never stores them: every wave walks strips of P rows (the 32x32 grid) and PRODUCES the two dy rows a P row
// brings into reach, straight into two LDS rings private to a pair of waves:
//     g    = conv^T(d_raw, w_tail)            one 16x16x32 MFMA per 16 pixels, K = 9 taps x {hi, lo} halves of d_raw (f32 kept)
//     mask = bn2(y2) + bns(ys) > 0            the residual join recomputed (its output was never stored either)
//     dy2  = A2 g mask + B2 y2 + C2,  dys = As g mask + Bs ys + Cs        (BatchNorm backward of both branches, coefficients from the
//                                                                          reduce pass that ran before)
// and then does, from those rings, what the two wgrad_stream passes did: the weight gradients of conv2 and upsample (16 taps x 16 x 16
// each, accumulators in registers), both data gradients (d_a1 with bn1's backward sums, and the upsample branch's share of the block-input
// gradient), one P row per MFMA K-step.  y2 / ys / d_raw rows are prefetched one step ahead in registers in the MFMA D-fragment mapping
// (lane = 4 channels of one pixel: what the producer computes in); nothing is read twice, nothing but the two small data gradients
// (168 MB each) is written.  The 1x1 conv's share of the block-input gradient needs bn1's batch sums, i.e. a grid-wide dependency:
// it is added by conv1_bwd_stream_kernel below after the finalize (dy1 = bn1 backward of d_a1, g_in += dy1 (x) W1, dW1 = dy1^T (x) xin).
//
// Two waves share a strip and split the work (see the kernel): 64 accumulators per wave, two waves per SIMD; LDS: 2 rings x 4 slots x
// 66 pixels x 32 B + rows = 21 KB per wave pair, the two convs' data-gradient A fragments (16 KB) block-shared.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

struct JoinBwdArgs {
  const float* d_raw; const float* w_tail;              // [N][Hg][Wg] f32 (one plane); [16][9] f32
  const void* y2; const void* ys;                       // [N][Hg][Wg][16] bf16, pre-BatchNorm
  const float* ms2; const float* mb2; const float* mss; const float* mbs;      // forward scale / shift of the two BatchNorms (mask)
  const float* A2; const float* B2; const float* C2; const float* As; const float* Bs; const float* Cs;   // backward coefficients
  const void* y1; const float* p1s; const float* p1b;   // conv2's input, pre-BatchNorm, with bn1's scale / shift (ReLU)
  const void* wd2; void* da1; float* part2; float* bn_part;
  const void* xin; const float* pxs; const float* pxb;  // the block input (optional BatchNorm+ReLU prologue)
  const void* wds; void* gin; float* parts;
  int N, Hp, Hg;
  int nunits;                                           // N: one unit = one image, all Hp rows
};

// Measured and NOT kept: a third ring with dy2 - bf16(dy2) and a second set of data-gradient MFMAs on it (hi/lo split of dy2 for conv2's
// data gradient).  bn1's backward sums are the remainder of a sum that cancels almost completely (BatchNorm-backward outputs have zero
// mean), and the bf16 rounding of 21 M dy2 elements is 35 % / 58 % of bn1's dgamma / dbeta at config 2's size at initialisation (3 % /
// 10 % after 60 steps); with hi + lo: 10 % / 19 % -- at +0.12 ms per step (7.13 -> 7.25), for an error that is below the gradient's
// own minibatch sampling noise either way (tests/test_config2_gpu.py).  Doing the lo part on only half of d_a1's pixels made things
// worse downstream: sums and applied tensor must stay consistent.
//
// Two waves share a strip (a "pair"): both PRODUCE -- wave 0 the first, wave 1 the second of the step's two new dy rows, for both
// branches, into the pair's two rings -- then wave 0 consumes the dy2 ring (conv2: weight gradient, d_a1, bn1 sums) and wave 1 the dys
// ring (upsample: weight gradient, its share of the block-input gradient).  64 accumulators per wave instead of 128: two waves per SIMD
// (the one-wave-per-strip form needed the whole 512-register file and ran at 2.3 TB/s).  Two block barriers per step order the
// pair's LDS traffic: [d_raw rows, P rows] -> barrier -> [produce] -> barrier -> [consume]; the loads of the next step are in flight
// during the consume phase.  A block = two pairs; every pair runs the same number of steps (idle ones past its last unit).
// F8I: y2 / ys are e4m3 bytes (fp8 mode's storage of these two tensors): a lane's 4 channels of a pixel are one dword
template <bool PRO_X, int PAIRS, bool F8I = false>
__global__ __launch_bounds__(128 * PAIRS, 2) void join_bwd_stream_kernel(JoinBwdArgs a) {
  constexpr int KS = 4, S = 2, PAD = 1, WP = 32, Wg = 64;
  constexpr int WL = S * (WP - 1) + KS;                    // 66 ring columns: -1 .. 64
  constexpr int ROWB = WL * 32;
  constexpr int NSLOT = 4;                                 // the four dy rows a P row multiplies; the two new ones replace the two oldest
  constexpr int DP = 68;                                   // d_raw ring pitch (floats): column c at index c + 1, zero halo at 0 and 65
  constexpr int NT = 16;
  constexpr int PAIR_LDS = 2 * NSLOT * ROWB + 3 * 1024 + 2 * 4 * DP * 4;
  constexpr int WDB = 2 * (NT / 2) * 1024;                 // the two convs' data-gradient A fragments
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  const int pair = wv >> 1, role = wv & 1;                 // role 0: conv2 (main path), role 1: upsample (shortcut)
  char* ringA = smem + WDB + pair * PAIR_LDS;              // dy2 rows
  char* ringB = ringA + NSLOT * ROWB;                      // dys rows
  char* prowA = ringB + NSLOT * ROWB;                      // relu(bn1(y1)) row
  char* rawA = prowA + 1024;                               // y1 row as stored (bn1 backward sums)
  char* prowB = rawA + 1024;                               // block-input row
  float* dring = reinterpret_cast<float*>(prowB + 1024);   // [hi, lo][4][DP] d_raw rows, split ONCE on the way in: hi = the value rounded to bf16
                                                           // (as an f32), lo = d - hi; a B fragment is then four v_perm of upper halves
  const bool odd = gq & 1, lo_half = gq >= 2;

  // ---- zero: ring border columns, the d ring (halo columns stay zero for the kernel's lifetime) -- each wave its half
  for (int sl = role * NSLOT; sl < (role + 1) * NSLOT; ++sl) {
    for (int i = lane; i < (PAD * 32) / 16; i += 64) reinterpret_cast<Vec16*>(ringA + sl * ROWB)[i] = Vec16{{0, 0, 0, 0}};
    for (int i = lane; i < ((WL - PAD) * 32 - Wg * 32) / 16; i += 64)
      reinterpret_cast<Vec16*>(ringA + sl * ROWB + PAD * 32 + Wg * 32)[i] = Vec16{{0, 0, 0, 0}};
  }
  if (role == 0) for (int i = lane; i < 2 * 4 * DP; i += 64) dring[i] = 0.f;
  // ---- block-shared A fragments of the two data gradients: [conv][pair of taps][lane]
  for (int i = t; i < 2 * (NT / 2) * 64; i += 128 * PAIRS) {
    const int ln = i & 63, pr = (i >> 6) % (NT / 2), cv = (i >> 6) / (NT / 2);
    const char* wd = reinterpret_cast<const char*>(cv ? a.wds : a.wd2);
    reinterpret_cast<Vec16*>(smem)[i] = *reinterpret_cast<const Vec16*>(wd + (ln & 15) * (NT * 32) + pr * 64 + (ln >> 4) * 16);
  }
  __syncthreads();
  // ---- the producer's constants (lane = channels 4gq .. 4gq+3)
  Vec16 wA;   // A of g: [ci = r][k = 8gq + t] = w[ci][tap = 8 (gq & 1) + t] (taps >= 9: zero), the same for the hi and the lo k range
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t0 = 8 * (gq & 1) + 2 * k;
    wA.w[k] = pack2_bf16(t0 < 9 ? a.w_tail[r * 9 + t0] : 0.f, t0 + 1 < 9 ? a.w_tail[r * 9 + t0 + 1] : 0.f);
  }
  // (channel pairs as 2-vectors: the producer's affine arithmetic runs on v_pk_fma_f32; the join's two shifts are added up front)
  f32x2_t msc[2], msh[2], msc1[2], ca0[2], cb0[2], cc0[2], ca1[2], cb1[2], cc1[2];
  float bsc[4], bsh[4];
  float bs0[4] = {0, 0, 0, 0}, bs1[4] = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ch = 4 * gq + j;
    msc[j >> 1][j & 1] = a.ms2[ch]; msh[j >> 1][j & 1] = a.mb2[ch] + a.mbs[ch]; msc1[j >> 1][j & 1] = a.mss[ch];
    ca0[j >> 1][j & 1] = a.A2[ch]; cb0[j >> 1][j & 1] = a.B2[ch]; cc0[j >> 1][j & 1] = a.C2[ch];
    ca1[j >> 1][j & 1] = a.As[ch]; cb1[j >> 1][j & 1] = a.Bs[ch]; cc1[j >> 1][j & 1] = a.Cs[ch];
    bsc[j] = a.p1s[ch]; bsh[j] = a.p1b[ch];
  }
  // ---- prologue coefficients of this wave's P row (8 consecutive channels per 16-byte vector: lane & 1 selects the half);
  // role 0: bn1 of y1 (ReLU); role 1: the block input's optional BatchNorm+ReLU
  float psc[8], psh[8];
  {
    const int c = (lane & 1) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      psc[j] = role == 0 ? a.p1s[c + j] : (PRO_X ? a.pxs[c + j] : 1.f);
      psh[j] = role == 0 ? a.p1b[c + j] : (PRO_X ? a.pxb[c + j] : 0.f);
    }
  }
  const bool pro = role == 0 || PRO_X;
  // ---- fragment offsets (as in wgrad_stream_kernel, one 32-pixel P row per step)
  int offA[2], offB[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int p = 8 * gq + 4 * blk + (r >> 2);
    offA[blk] = p * 32 + (r & 3) * 8;
    offB[blk] = (S * p) * 32 + (r & 3) * 8;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) acc[k] = (f32x4){0, 0, 0, 0};

  const bf16_t* __restrict__ Y2 = reinterpret_cast<const bf16_t*>(a.y2);
  const bf16_t* __restrict__ YS = reinterpret_cast<const bf16_t*>(a.ys);
  // XCD-aware unit walk over PAIRS (block b runs on XCD b % 8); every pair of a block runs `iters` steps
  const int nblk = gridDim.x;
  int u_first, u_step, u_end;
  if ((nblk & 7) == 0) {
    const int per = (a.nunits + 7) >> 3;
    const int lo = (blockIdx.x & 7) * per;
    u_first = lo + (blockIdx.x >> 3) * PAIRS + pair; u_step = (nblk >> 3) * PAIRS; u_end = min(a.nunits, lo + per);
  } else { u_first = blockIdx.x * PAIRS + pair; u_step = nblk * PAIRS; u_end = a.nunits; }
  const int nq = a.Hp + 1;                                  // the priming step + one per P row
  // units of the block's first pair >= units of its second: the first pair's count bounds both
  int my_units = 0, max_units = 0;
  {
    const int f0 = u_first - pair;
    max_units = f0 < u_end ? (u_end - f0 + u_step - 1) / u_step : 0;
    my_units = u_first < u_end ? (u_end - u_first + u_step - 1) / u_step : 0;
  }
  const int iters = max_units * nq;

  // ---- registers of the step in flight: this wave's new dy row's operands in the D-fragment mapping, the new d_raw rows (role 0),
  // this wave's P row
  uint2 q2[4], qs[4];
  float dv4[4] = {0, 0, 0, 0};
  Vec16 pv = Vec16{{0, 0, 0, 0}};
  const char* Pmine = reinterpret_cast<const char*>(role == 0 ? a.y1 : a.xin);
  auto issue = [&](int n, int q) {
    const int h = q - 1;                                    // P row (q = 0: the rows above the first step's own)
    const int top = S * h - PAD + KS - 1;                   // the step's new dy rows: top - 1 (wave 0), top (wave 1)
    {
      const int row = top - 1 + role;
      const bool ok = row >= 0 && row < a.Hg;
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const long e = (((long)n * a.Hg + (ok ? row : 0)) * Wg + 16 * pt + r) * 16 + 4 * gq;
        if constexpr (F8I) {
          q2[pt] = make_uint2(ok ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(Y2) + e)) : 0u, 0u);
          qs[pt] = make_uint2(ok ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(YS) + e)) : 0u, 0u);
        } else {
          q2[pt] = ok ? load_nt(reinterpret_cast<const uint2*>(Y2 + e)) : make_uint2(0, 0);
          qs[pt] = ok ? load_nt(reinterpret_cast<const uint2*>(YS + e)) : make_uint2(0, 0);
        }
      }
    }
    // d_raw rows (wave 0): a dy row needs its neighbours above and below.  Priming: rows top - 2 .. top + 1 (four rows, one float4
    // per lane); afterwards the two new ones, top and top + 1 (lanes 0 .. 31: one float4 each)
    if (role == 0) {
      const float* dp = a.d_raw + (long)n * a.Hg * Wg;
      const int row = q == 0 ? top - 2 + (lane >> 4) : top + ((lane >> 4) & 1), c = (lane & 15) * 4;
      const float4 v = ((q == 0 || lane < 32) && row >= 0 && row < a.Hg) ? *reinterpret_cast<const float4*>(dp + row * Wg + c) : make_float4(0, 0, 0, 0);
      dv4[0] = v.x; dv4[1] = v.y; dv4[2] = v.z; dv4[3] = v.w;
    }
    if (q > 0) pv = *reinterpret_cast<const Vec16*>(Pmine + (((long)n * a.Hp + h) * WP) * 32 + lane * 16);
  };
  auto commit_rows = [&](int q) {
    const int h = q - 1;
    const int top = S * h - PAD + KS - 1;
    if (role == 0) {
      const int row = q == 0 ? top - 2 + (lane >> 4) : top + ((lane >> 4) & 1);
      if (q == 0 || lane < 32) {
        float* dst = dring + ((row + 8) & 3) * DP + (lane & 15) * 4 + 1;
        const uint32_t h01 = pack2_bf16(dv4[0], dv4[1]), h23 = pack2_bf16(dv4[2], dv4[3]);
        const float h0 = __uint_as_float(h01 << 16), h1 = __uint_as_float(h01 & 0xffff0000u), h2 = __uint_as_float(h23 << 16), h3 = __uint_as_float(h23 & 0xffff0000u);
        dst[0] = h0; dst[1] = h1; dst[2] = h2; dst[3] = h3;
        const uint32_t l01 = pack2_bf16(dv4[0] - h0, dv4[1] - h1), l23 = pack2_bf16(dv4[2] - h2, dv4[3] - h3);      // lo rounded (not cut) to bf16
        dst[4 * DP] = __uint_as_float(l01 << 16); dst[4 * DP + 1] = __uint_as_float(l01 & 0xffff0000u);
        dst[4 * DP + 2] = __uint_as_float(l23 << 16); dst[4 * DP + 3] = __uint_as_float(l23 & 0xffff0000u);
      }
    }
    if (q > 0) {
      char* prow = role == 0 ? prowA : prowB;
      if (role == 0) *reinterpret_cast<Vec16*>(rawA + lane * 16) = pv;
      Vec16 v = pv;
      if (pro) {
        float f[8];
        Elem<bf16_t>::unpack(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * psc[j] + psh[j], 0.f);
        v = Elem<bf16_t>::pack(f);
      }
      *reinterpret_cast<Vec16*>(prow + lane * 16) = v;
    }
  };
  auto produce = [&](int q) {
    const int h = q - 1;
    const int top = S * h - PAD + KS - 1;
    // this wave's new dy row of both branches (rows outside the image are the convolution's zero padding)
    const int row = top - 1 + role;
    const bool ok = row >= 0 && row < a.Hg;
    const int slot = (row + 8) & 3;
    const float* dimg = dring + (lo_half ? 4 * DP : 0);      // k = 0..15: hi parts, k = 16..31: lo parts against the same weights
    const float* d0 = dimg + ((row + 8) & 3) * DP;           // d_raw rows `row`, row - 1, row + 1 (ring slots mod 4)
    const float* dm = dimg + ((row + 7) & 3) * DP;
    const float* dq = dimg + ((row + 9) & 3) * DP;
    if (!ok) {                                               // (wave-uniform) rows outside the image are the convolution's zero padding
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        *reinterpret_cast<uint2*>(ringA + slot * ROWB + (PAD + 16 * pt + r) * 32 + gq * 8) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(ringB + slot * ROWB + (PAD + 16 * pt + r) * 32 + gq * 8) = make_uint2(0u, 0u);
      }
      return;
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int c = 16 * pt + r + 1;                         // ring index of the pixel's column
      // B[k = 8gq + t][pixel r]: tap t = 3 kh + kw reads d(row + 1 - kh, col + 1 - kw); taps 0..7 on even gq, tap 8 in slot 0 on odd gq
      float dv[8];
      dv[0] = odd ? dm[c - 1] : dq[c + 1];
      dv[1] = dq[c]; dv[2] = dq[c - 1]; dv[3] = d0[c + 1]; dv[4] = d0[c]; dv[5] = d0[c - 1]; dv[6] = dm[c + 1]; dv[7] = dm[c];
      // (odd gq: only k-slot 0 = tap 8 meets a non-zero weight in wA; the other seven values are finite and multiply zeros)
      Vec16 bf;
#pragma unroll
      for (int k = 0; k < 4; ++k) bf.w[k] = __builtin_amdgcn_perm(__float_as_uint(dv[2 * k + 1]), __float_as_uint(dv[2 * k]), 0x07060302u);
      const f32x4 g = mma_bf16(wA, bf, (f32x4){0.f, 0.f, 0.f, 0.f});
      const uint2 v2 = q2[pt], vs = qs[pt];
      float f0[4], f1[4];
      if constexpr (F8I) { unpack4_fp8(v2.x, f0); unpack4_fp8(vs.x, f1); }
      else { unpack4_bf16(v2, f0); unpack4_bf16(vs, f1); }
      float r0[4], r1[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2_t v0 = {f0[2 * h], f0[2 * h + 1]}, v1 = {f1[2 * h], f1[2 * h + 1]};
        const f32x2_t x = __builtin_elementwise_fma(v1, msc1[h], __builtin_elementwise_fma(v0, msc[h], msh[h]));
        const f32x2_t gg = {x[0] > 0.f ? g[2 * h] : 0.f, x[1] > 0.f ? g[2 * h + 1] : 0.f};
        const f32x2_t q0 = __builtin_elementwise_fma(ca0[h], gg, __builtin_elementwise_fma(cb0[h], v0, cc0[h]));
        const f32x2_t q1 = __builtin_elementwise_fma(ca1[h], gg, __builtin_elementwise_fma(cb1[h], v1, cc1[h]));
        r0[2 * h] = q0[0]; r0[2 * h + 1] = q0[1]; r1[2 * h] = q1[0]; r1[2 * h + 1] = q1[1];
      }
      *reinterpret_cast<uint2*>(ringA + slot * ROWB + (PAD + 16 * pt + r) * 32 + gq * 8) = make_uint2(pack2_bf16(r0[0], r0[1]), pack2_bf16(r0[2], r0[3]));
      *reinterpret_cast<uint2*>(ringB + slot * ROWB + (PAD + 16 * pt + r) * 32 + gq * 8) = make_uint2(pack2_bf16(r1[0], r1[1]), pack2_bf16(r1[2], r1[3]));
    }
  };

  const Vec16* wdL = reinterpret_cast<const Vec16*>(smem) + role * (NT / 2) * 64;
  const char* ringM = role == 0 ? ringA : ringB;            // the ring this wave consumes
  const char* prowM = role == 0 ? prowA : prowB;
  bf16_t* dxM = reinterpret_cast<bf16_t*>(role == 0 ? a.da1 : a.gin);
  int u = u_first, q = 0, done = 0;
  if (my_units > 0) issue(u, 0);
  for (int it = 0; it < iters; ++it) {
    const bool active = done < my_units;
    if (active) commit_rows(q);
    __syncthreads();
    if (active) produce(q);
    int un = u, qn = q + 1, dn = done;
    if (qn == nq) { un = u + u_step; qn = 0; dn = done + 1; }
    if (active && dn < my_units) issue(un, qn);
    __syncthreads();
    if (active && q > 0) {
      const int h = q - 1;
      const int first = S * h - PAD;                        // dy row of tap row kh = 0
      const Vec16 af = FragOps<bf16_t>::load(prowM, offA[0], offA[1]);
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        const char* rowp = ringM + ((first + kh + 8) & 3) * ROWB;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const Vec16 bf = FragOps<bf16_t>::load(rowp, offB[0] + kw * 32, offB[1] + kw * 32);
          acc[kh * KS + kw] = mma_bf16(af, bf, acc[kh * KS + kw]);
        }
      }
      f32x4 dacc[2] = {(f32x4){0, 0, 0, 0}, (f32x4){0, 0, 0, 0}};
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        const char* rowp = ringM + ((first + kh + 8) & 3) * ROWB;
#pragma unroll
        for (int kp = 0; kp < KS / 2; ++kp) {
          const Vec16 wAd = wdL[(kh * (KS / 2) + kp) * 64 + lane];
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            const Vec16 b = *reinterpret_cast<const Vec16*>(rowp + (S * (16 * pt + r) + 2 * kp + (gq >> 1)) * 32 + (gq & 1) * 16);
            dacc[pt] = mma_bf16(wAd, b, dacc[pt]);
          }
        }
      }
      if (role == 0) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          const uint2 yr = *reinterpret_cast<const uint2*>(rawA + (16 * pt + r) * 32 + gq * 8);
          const float y[4] = {__uint_as_float(yr.x << 16), __uint_as_float(yr.x & 0xffff0000u), __uint_as_float(yr.y << 16), __uint_as_float(yr.y & 0xffff0000u)};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float g = (y[j] * bsc[j] + bsh[j] > 0.f) ? dacc[pt][j] : 0.f;
            bs0[j] += g; bs1[j] += g * y[j];
          }
        }
      }
      bf16_t* drow = dxM + (((long)u * a.Hp + h) * WP) * 16 + 4 * gq;
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        float v[4] = {dacc[pt][0], dacc[pt][1], dacc[pt][2], dacc[pt][3]};
        dstore4<bf16_t>(drow + (16 * pt + r) * 16, v, false);
      }
    }
    u = un; q = qn; done = dn;
  }

  // ---- flush: the two pairs add their accumulators in LDS (pair order), the block stores one partial image per conv
  float* img = reinterpret_cast<float*>(smem);
  __syncthreads();
  for (int p = 0; p < PAIRS; ++p) {
    if (pair == p) {
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* o = img + role * (NT * 256) + (k * 16 + 4 * gq + j) * 16 + r;
          *o = (p == 0 ? 0.f : *o) + acc[k][j];
        }
    }
    __syncthreads();
  }
  {
    float* sb = img + 2 * NT * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs0[j] = row16_sum(bs0[j]); bs1[j] = row16_sum(bs1[j]); }
    if (r == 0 && role == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sb[pair * 32 + 4 * gq + j] = bs0[j]; sb[pair * 32 + 16 + 4 * gq + j] = bs1[j]; }
    }
    __syncthreads();
    if (t < 32) a.bn_part[(long)blockIdx.x * 32 + t] = PAIRS == 2 ? sb[t] + sb[32 + t] : sb[t];
  }
  float* dst2 = a.part2 + (long)blockIdx.x * (NT * 256);
  float* dsts = a.parts + (long)blockIdx.x * (NT * 256);
  for (int i = t; i < NT * 256 / 4; i += 128 * PAIRS) {
    reinterpret_cast<float4*>(dst2)[i] = reinterpret_cast<const float4*>(img)[i];
    reinterpret_cast<float4*>(dsts)[i] = reinterpret_cast<const float4*>(img + NT * 256)[i];
  }
}

bool join_bwd_stream_ok(int dt, int OC, int C, int Hp, int Hg) {
  constexpr int enabled = 1;
  return enabled != 0 && dt == DT_BF16 && OC == 1 && C == 16 && Hp == 32 && Hg == 64;
}

// Returns the number of blocks (= partial images per conv = rows of bn_part), <0 on error.
int launch_join_bwd_stream(const JoinBwdLaunch& L, hipStream_t s) {
  JoinBwdArgs a; memset(&a, 0, sizeof(a));
  a.d_raw = L.d_raw; a.w_tail = L.w_tail; a.y2 = L.y2; a.ys = L.ys;
  a.ms2 = L.ms2; a.mb2 = L.mb2; a.mss = L.mss; a.mbs = L.mbs;
  a.A2 = L.A2; a.B2 = L.B2; a.C2 = L.C2; a.As = L.As; a.Bs = L.Bs; a.Cs = L.Cs;
  a.y1 = L.y1; a.p1s = L.p1s; a.p1b = L.p1b; a.wd2 = L.wd2; a.da1 = L.da1; a.part2 = L.part2; a.bn_part = L.bn_part;
  a.xin = L.xin; a.pxs = L.pxs; a.pxb = L.pxb; a.wds = L.wds; a.gin = L.gin; a.parts = L.parts;
  a.N = L.N; a.Hp = 32; a.Hg = 64; a.nunits = L.N;
  if (!L.p1s || !L.p1b) { set_error("join_bwd_stream: conv2's input needs bn1's scale / shift"); return MMVAE_ERR_ARG; }
  // measured (config 2, ms per step): two pairs per block x 512 blocks 7.17; 384 blocks 7.34; 768 blocks 7.24; one pair per block
  // (128 threads, 1024 blocks) 9.8 -- its register cap spills the accumulators
  // Phase ablation (isolated launch with its three helper launches, us): all 604; without the producer 451; without the weight-gradient
  // MFMAs 544; without the data-gradient MFMAs 535; without bn1's sums 587; without any arithmetic 397 -- the streaming skeleton (loads,
  // LDS commits, two barriers per step, stores) is ~280 us of the kernel's ~485, the producer ~155, the consumers ~125.
  // ... (round 3, after the producer's instruction diet: one pair per block x 1024 blocks, four blocks per CU by LDS, 7.006 / 7.007 against
  // 6.990 / 7.041 -- the block barrier that couples the two pairs is not the limit; the kernel is bound by its instruction issue, not by
  // memory: e4m3 storage of y2 / ys (fp8 mode, 1.4 GB fewer bytes) leaves its 600 us unchanged)
  constexpr int pairs = 2;
  int gx = 512;                                             // two blocks of two wave pairs per CU (LDS: 59 KB each)
  while (gx > 8 && (long)gx * pairs > a.nunits) gx -= 8;
  constexpr size_t lds = 2 * 8 * 1024 + pairs * (size_t)(2 * 4 * 66 * 32 + 3 * 1024 + 2 * 4 * 68 * 4);
  static_assert(lds >= 2 * 16 * 256 * 4 + 512, "the flush images alias the rings");
  note_launch_bytes((double)L.N * (2.0 * 64 * 64 * 16 * (L.f8in ? 1 : 2) + 64 * 64 * 4.0 + 4.0 * 32 * 32 * 16 * 2));   // y2, ys, d_raw; y1, xin, d_a1, g_in
#define MMVAE_JB(PX, PR, F8)                                                                                                        \
  do {                                                                                                                               \
    static bool attr_set = false;                                                                                                    \
    if (!attr_set) {                                                                                                                 \
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&join_bwd_stream_kernel<PX, PR, F8>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
      if (e != hipSuccess) { set_error("join_bwd_stream: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MMVAE_ERR_HIP; }    \
      attr_set = true;                                                                                                               \
    }                                                                                                                                \
    hipLaunchKernelGGL((join_bwd_stream_kernel<PX, PR, F8>), dim3(gx), dim3(128 * PR), lds, s, a);                                   \
  } while (0)
  if (L.f8in) { if (L.pxs) MMVAE_JB(true, 2, true); else MMVAE_JB(false, 2, true); }
  else { if (L.pxs) MMVAE_JB(true, 2, false); else MMVAE_JB(false, 2, false); }
#undef MMVAE_JB
  const int rc = check_launch("join_bwd_stream");
  return rc ? rc : gx;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The 1x1 conv of a DeconvBottleneck (conv1: block input -> 16 channels) backward, after bn1's sums are final:
//   dy1 = A1 (d_a1 [bn1(y1) > 0]) + B1 y1 + C1;   g_in += dy1 (x) W1 (in place on the shortcut's share);   dW1 += dy1^T (x) pro(xin)
// One row of 32 pixels per wave and step, rows prefetched one step ahead, both products on the matrix cores (the forms of the
// X2 part of wgrad_stream_kernel).  Cin = 16.
struct Conv1BwdArgs {
  const void* da1; const void* y1; const float* ms; const float* mb; const float* A; const float* B; const float* C;
  const void* xin; const float* pxs; const float* pxb;
  const void* w1u;                                          // conv1's packed "up" form [Cin = 16][16]
  void* gin; float* part;
  long nrows;                                               // N * Hp rows of 32 pixels
};

// MASK: g_in leaves multiplied by [pro(xin) > 0] -- xin is the output of the block below's join ReLU, so that block's BatchNorm backward
// needs no mask pass (its reduce runs unmasked, its dy is evaluated by the consumers' loaders: wgrad_stream_kernel JG)
template <bool PRO_X, bool MASK>
__global__ __launch_bounds__(256, 4) void conv1_bwd_stream_kernel(Conv1BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* dyrow = smem + wv * 2048;                           // dy1 row [32][16] bf16
  char* xrow = dyrow + 1024;                                // pro(xin) row
  float sc[8], sh[8], cA[8], cB[8], cC[8], xs[8], xb[8];
  {
    const int c = (lane & 1) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = a.ms[c + j]; sh[j] = a.mb[c + j]; cA[j] = a.A[c + j]; cB[j] = a.B[c + j]; cC[j] = a.C[c + j];
      xs[j] = PRO_X ? a.pxs[c + j] : 1.f; xb[j] = PRO_X ? a.pxb[c + j] : 0.f;
    }
  }
  Vec16 w2A = Vec16{{0, 0, 0, 0}};
  if (gq < 2) w2A = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.w1u) + r * 32 + gq * 16);
  int offA[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) offA[blk] = (8 * gq + 4 * blk + (r >> 2)) * 32 + (r & 3) * 8;
  f32x4 acc = (f32x4){0, 0, 0, 0};
  const long wstep = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wv;
  Vec16 vd = Vec16{{0, 0, 0, 0}}, vy = vd, vx = vd;
  uint2 gv[2] = {make_uint2(0, 0), make_uint2(0, 0)};
  auto issue = [&](long rw) {
    vd = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.da1) + rw * 1024 + lane * 16);
    vy = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.y1) + rw * 1024 + lane * 16);
    vx = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.xin) + rw * 1024 + lane * 16);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) gv[pt] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.gin) + rw * 1024 + (16 * pt + r) * 32 + gq * 8);
  };
  if (row < a.nrows) issue(row);
  while (row < a.nrows) {
    {
      float d[8], y[8], x[8];
      Elem<bf16_t>::unpack(vd, d);
      Elem<bf16_t>::unpack(vy, y);
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = cA[j] * ((y[j] * sc[j] + sh[j] > 0.f) ? d[j] : 0.f) + cB[j] * y[j] + cC[j];
      *reinterpret_cast<Vec16*>(dyrow + lane * 16) = Elem<bf16_t>::pack(d);
      if constexpr (PRO_X) {
        Elem<bf16_t>::unpack(vx, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = fmaxf(x[j] * xs[j] + xb[j], 0.f);
        *reinterpret_cast<Vec16*>(xrow + lane * 16) = Elem<bf16_t>::pack(x);
      } else {
        *reinterpret_cast<Vec16*>(xrow + lane * 16) = vx;
      }
    }
    const uint2 g0[2] = {gv[0], gv[1]};
    const long cur = row;
    row += wstep;
    if (row < a.nrows) issue(row);
    __builtin_amdgcn_sched_barrier(0);
    const Vec16 ad = FragOps<bf16_t>::load(dyrow, offA[0], offA[1]);
    const Vec16 ax = FragOps<bf16_t>::load(xrow, offA[0], offA[1]);
    acc = mma_bf16(ad, ax, acc);                            // dW1[c1 out][c1 in] += dy1^T (x) x
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const Vec16 b = *reinterpret_cast<const Vec16*>(dyrow + (16 * pt + r) * 32 + (gq & 1) * 16);      // k >= 16: w2A is zero there
      const f32x4 dg = mma_bf16(w2A, b, (f32x4){0, 0, 0, 0});
      const uint2 e = g0[pt];
      float v[4] = {dg[0] + __uint_as_float(e.x << 16), dg[1] + __uint_as_float(e.x & 0xffff0000u), dg[2] + __uint_as_float(e.y << 16),
                    dg[3] + __uint_as_float(e.y & 0xffff0000u)};
      if constexpr (MASK) {
        const uint2 xr = *reinterpret_cast<const uint2*>(xrow + (16 * pt + r) * 32 + gq * 8);          // pro(xin)[pixel][4gq .. 4gq+3]
        const float xm[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u), __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = xm[j] > 0.f ? v[j] : 0.f;
      }
      dstore4<bf16_t>(reinterpret_cast<bf16_t*>(a.gin) + cur * 512 + (16 * pt + r) * 16 + 4 * gq, v, false);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float* img = reinterpret_cast<float*>(smem);
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* p = img + (4 * gq + j) * 16 + r;
        *p = (w == 0 ? 0.f : *p) + acc[j];
      }
    }
    __syncthreads();
  }
  if (t < 64) reinterpret_cast<float4*>(a.part + (long)blockIdx.x * 256)[t] = reinterpret_cast<const float4*>(img)[t];
}

int launch_conv1_bwd_stream(const Conv1BwdLaunch& L, hipStream_t s) {
  Conv1BwdArgs a; memset(&a, 0, sizeof(a));
  a.da1 = L.da1; a.y1 = L.y1; a.ms = L.ms; a.mb = L.mb; a.A = L.A; a.B = L.B; a.C = L.C; a.xin = L.xin; a.pxs = L.pxs; a.pxb = L.pxb;
  a.w1u = L.w1u; a.gin = L.gin; a.part = L.part; a.nrows = L.nrows;
  int gx = 1024;
  while (gx > 8 && (long)gx * 4 > a.nrows) gx -= 8;
  note_launch_bytes((double)L.nrows * 1024.0 * 5);         // d_a1, y1, xin, g_in read; g_in written
  if (L.pxs) {
    if (L.mask_out) hipLaunchKernelGGL((conv1_bwd_stream_kernel<true, true>), dim3(gx), dim3(256), 4 * 2048, s, a);
    else hipLaunchKernelGGL((conv1_bwd_stream_kernel<true, false>), dim3(gx), dim3(256), 4 * 2048, s, a);
  } else {
    if (L.mask_out) hipLaunchKernelGGL((conv1_bwd_stream_kernel<false, true>), dim3(gx), dim3(256), 4 * 2048, s, a);
    else hipLaunchKernelGGL((conv1_bwd_stream_kernel<false, false>), dim3(gx), dim3(256), 4 * 2048, s, a);
  }
  const int rc = check_launch("conv1_bwd_stream");
  return rc ? rc : gx;
}

}  // namespace mmvae
