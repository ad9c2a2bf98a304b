// conv3_stream_kernel: forward of the 32 -> 32 channel 3x3 convolutions of the first encoder block (encoder.layer1: conv1 stride 2 on
// the 32x32 stem activation together with the block's 1x1 stride-2 shortcut, conv2 stride 1 on 16x16) as a barrier-free stream.
//
// The patch-tile kernel moves these layers at 1.5-2.2 TB/s, and conv1 and the shortcut each read the same 335 MB input (at N = 5120).
// Here every WAVE walks strips of output rows on its own: a ring of input rows (fused BN+ReLU applied once, on the way in) lives in
// wave-private LDS, the next rows are in flight in registers, one output row = 16 pixels = one MFMA column tile, so a 3x3 tap is ONE
// v_mfma_f32_16x16x32_bf16 per 16 output channels (K = the 32 input channels), B fragments are plain 16-byte reads of ring pixels, the
// A fragments (9 taps x 2, + 2 for the shortcut: the centre tap's pixel IS the shortcut's pixel) stay in registers for the whole
// kernel, a lane ends with 8 consecutive output channels of its pixel (one 16-byte store per tensor), BatchNorm sums come from the
// f32 results.  With SC the input is read ONCE for conv1 and the shortcut.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

struct Conv3StreamArgs {
  const void* x; const void* w; const void* wsc; void* y; void* ysc;
  const float* pro_scale; const float* pro_shift; int pro_relu;
  float* stats; float* stats_sc;          // partial rows [blocks][2][32] (nullable)
  // BS (a data gradient run as a forward conv over dy): `stats` receives the BatchNorm-BACKWARD sums of the BatchNorm + ReLU that produced
  // this conv's output side in the forward pass: rows [sum g, sum g * ym] with g = the (bf16-rounded) result masked by ym * ms + mb > 0
  const void* ym; const float* ms; const float* mb;
  int N, Ho, Hi;
  int HS, nunits;                         // output rows per strip, N * Ho / HS
};

// S: stride (1 or 2); SC: also the 1x1 stride-S shortcut (second weight set, second output); 32 -> 32 channels, 16 output columns
template <int S, bool SC, bool PRO, bool BS = false>
__global__ __launch_bounds__(256, 2) void conv3_stream_kernel(Conv3StreamArgs a) {
  static_assert(!BS || (S == 1 && !SC && !PRO), "backward sums: the stride-1 data gradient");
  constexpr int KS = 3, PAD = 1, WO = 16, WI = S * WO, CB = 64;            // bytes per pixel (32 channels of bf16)
  constexpr int WL = WI + 2;                                                // ring row: one zero pixel on each side
  constexpr int ROWB = WL * CB;
  constexpr int NSLOT = KS + S;
  constexpr int WAVE_LDS = NSLOT * ROWB;
  constexpr int RV = S * WI * CB / 1024;                                    // 16-byte vectors per lane for the S rows of a step (S = 2: 4, S = 1: 1)
  constexpr int NPRIME = (KS - S + S - 1) / S;                              // steps that only load (S = 2: 1, S = 1: 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* ring = smem + 2048 + wv * WAVE_LDS;
  for (int s = 0; s < NSLOT; ++s) {
    if (lane < 4) *reinterpret_cast<Vec16*>(ring + s * ROWB + lane * 16) = Vec16{{0, 0, 0, 0}};
    else if (lane < 8) *reinterpret_cast<Vec16*>(ring + s * ROWB + (WL - 1) * CB + (lane - 4) * 16) = Vec16{{0, 0, 0, 0}};
  }
  // A fragments: row r of fragment m is output channel 8*(r/4) + 4m + r%4; k = input channels 8gq .. 8gq+7 of the tap: 16 bytes of
  // the packed [cout][tap][cin] weights
  Vec16 wA[KS * KS][2], wS[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int c = 8 * (r >> 2) + 4 * m + (r & 3);
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap)
      wA[tap][m] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.w) + ((c * KS * KS + tap) * 32 + 8 * gq) * 2);
    wS[m] = SC ? *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.wsc) + (c * 32 + 8 * gq) * 2) : Vec16{{0, 0, 0, 0}};
  }
  float psc[8], psh[8];
  if (PRO) {
    const int c = (lane & 3) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { psc[j] = a.pro_scale[c + j]; psh[j] = a.pro_shift[c + j]; }
  }
  const float lo = a.pro_relu ? 0.f : -__builtin_inff();
  float bms[8], bmb[8];
  if (BS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { bms[j] = a.ms[8 * gq + j]; bmb[j] = a.mb[8 * gq + j]; }
  }
  Vec16 yv = Vec16{{0, 0, 0, 0}};
  float st[SC ? 4 : 2][8];
#pragma unroll
  for (int q = 0; q < (SC ? 4 : 2); ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) st[q][j] = 0.f;

  const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(a.x);
  const int nblk = gridDim.x;
  int u_first, u_step, u_end;
  if ((nblk & 7) == 0) {                                                    // XCD-aware walk (see conv_wstream.hip)
    const int per = (a.nunits + 7) >> 3;
    const int lo_u = (blockIdx.x & 7) * per;
    u_first = lo_u + (blockIdx.x >> 3) * 4 + wv; u_step = (nblk >> 3) * 4; u_end = min(a.nunits, lo_u + per);
  } else { u_first = blockIdx.x * 4 + wv; u_step = nblk * 4; u_end = a.nunits; }
  const int nstrips = a.Ho / a.HS;
  const int nq = a.HS + NPRIME;

  Vec16 xv[RV];
  auto issue = [&](int u, int q) {
    const int n = u / nstrips, oh0 = (u - n * nstrips) * a.HS;
    const int top = S * (oh0 + q - NPRIME) + 1;                             // last input row of output row oh0 + q - NPRIME
#pragma unroll
    for (int k = 0; k < RV; ++k) {
      const int byte = (lane + 64 * k) * 16;
      const int rr = byte / (WI * CB), off = byte - rr * (WI * CB);
      const int row = top - S + 1 + rr;
      xv[k] = Vec16{{0, 0, 0, 0}};
      if (row >= 0 && row < a.Hi) xv[k] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(X) + (((long)n * a.Hi + row) * WI) * CB + off);
    }
    if (BS && q >= NPRIME)     // the mask / sum operand of the output row this step computes: this lane's 8 channels of pixel r
      yv = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.ym) + ((((long)n * a.Ho + oh0 + q - NPRIME) * WO + r) * 32 + 8 * gq) * 2);
  };
  auto commit = [&](int u, int q) {
    const int n = u / nstrips, oh0 = (u - n * nstrips) * a.HS;
    const int top = S * (oh0 + q - NPRIME) + 1;
#pragma unroll
    for (int k = 0; k < RV; ++k) {
      const int byte = (lane + 64 * k) * 16;
      const int rr = byte / (WI * CB), off = byte - rr * (WI * CB);
      const int row = top - S + 1 + rr;
      Vec16 v = xv[k];
      if (PRO && row >= 0 && row < a.Hi) {
        float f[8];
        Elem<bf16_t>::unpack(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * psc[j] + psh[j], lo);
        v = Elem<bf16_t>::pack(f);
      }
      *reinterpret_cast<Vec16*>(ring + ((row + 4 * NSLOT) % NSLOT) * ROWB + CB + off) = v;
    }
    (void)n;
  };

  int u = u_first, q = 0;
  if (u < u_end) issue(u, 0);
  while (u < u_end) {
    commit(u, q);
    const Vec16 ycur = yv;                                                  // (BS) loaded for THIS step by the previous issue
    int un = u, qn = q + 1;
    if (qn == nq) { un = u + u_step; qn = 0; }
    if (un < u_end) issue(un, qn);
    __builtin_amdgcn_sched_barrier(0);
    if (q >= NPRIME) {
      const int n = u / nstrips, oh = (u - n * nstrips) * a.HS + q - NPRIME;
      const int first = S * oh - PAD;
      f32x4 acc[2], asc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) { acc[m] = (f32x4){0, 0, 0, 0}; asc[m] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        const char* rowp = ring + ((first + kh + 4 * NSLOT) % NSLOT) * ROWB;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const Vec16 b = *reinterpret_cast<const Vec16*>(rowp + (S * r + kw) * CB + gq * 16);
          acc[0] = mma_bf16(wA[kh * KS + kw][0], b, acc[0]);
          acc[1] = mma_bf16(wA[kh * KS + kw][1], b, acc[1]);
          if (SC && kh == 1 && kw == 1) { asc[0] = mma_bf16(wS[0], b, asc[0]); asc[1] = mma_bf16(wS[1], b, asc[1]); }
        }
      }
      const long po = (((long)n * a.Ho + oh) * WO + r) * 32 + 8 * gq;
      {
        const float v[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
        if constexpr (BS) {
          float ym[8], vr[8];
          Elem<bf16_t>::unpack(ycur, ym);
          Elem<bf16_t>::unpack(Elem<bf16_t>::pack(v), vr);                // the gradient as the stored tensor holds it
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float g = (ym[j] * bms[j] + bmb[j] > 0.f) ? vr[j] : 0.f;
            st[0][j] += g; st[1][j] += g * ym[j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { st[0][j] += v[j]; st[1][j] += v[j] * v[j]; }
        }
        dstore8<bf16_t>(reinterpret_cast<bf16_t*>(a.y) + po, v, false);
      }
      if constexpr (SC) {
        const float v[8] = {asc[0][0], asc[0][1], asc[0][2], asc[0][3], asc[1][0], asc[1][1], asc[1][2], asc[1][3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) { st[2][j] += v[j]; st[3][j] += v[j] * v[j]; }
        dstore8<bf16_t>(reinterpret_cast<bf16_t*>(a.ysc) + po, v, false);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    u = un; q = qn;
  }
  // ---- BatchNorm sums: the 16 pixel-lanes of a row (DPP), the four waves in order; partial rows [2][32] per block and tensor
  float* sb = reinterpret_cast<float*>(smem);                               // [4 waves][128]
#pragma unroll
  for (int qq = 0; qq < (SC ? 4 : 2); ++qq)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = row16_sum(st[qq][j]);
      if (r == 0) sb[wv * 128 + qq * 32 + 8 * gq + j] = v;
    }
  __syncthreads();
  if (t < (SC ? 128 : 64)) {
    const float v = (sb[t] + sb[128 + t]) + (sb[256 + t] + sb[384 + t]);
    float* dst = t < 64 ? a.stats : a.stats_sc;
    if (dst) dst[(long)blockIdx.x * 64 + (t & 63)] = v;
  }
}

bool conv3_stream_ok(int dt, int Cin, int Cout, int k, int s, int p, int Hin, int Win) {
  constexpr bool env = true;
  return env && dt == DT_BF16 && Cin == 32 && Cout == 32 && k == 3 && p == 1 && (s == 1 || s == 2) && Hin == Win && Win == 16 * s;
}

// y [N][Ho][16][32] = conv3x3(x [N][s*Ho][s*16][32], stride s, pad 1) with the packed [32][9][32] weights; wsc / ysc / stats_sc (optional, s = 2):
// the 1x1 stride-2 shortcut on the same input, packed [32][32].  Returns the number of partial rows (> 0) or an error.
int launch_conv3_stream(int dt, int stride, const void* x, const void* w, const void* wsc, void* y, void* ysc, const float* pro_scale,
                        const float* pro_shift, int pro_relu, float* stats, float* stats_sc, int N, int Ho, hipStream_t s) {
  if (dt != DT_BF16 || (stride != 1 && stride != 2) || (wsc && stride != 2) || ((wsc != nullptr) != (ysc != nullptr))) {
    set_error("conv3_stream: bf16, stride 1 or 2, shortcut only with stride 2"); return MMVAE_ERR_UNSUPPORTED;
  }
  Conv3StreamArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.w = w; a.wsc = wsc; a.y = y; a.ysc = ysc; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.pro_relu = pro_relu;
  a.stats = stats; a.stats_sc = stats_sc; a.N = N; a.Ho = Ho; a.Hi = stride * Ho;
  a.HS = Ho % 8 == 0 ? 8 : Ho;
  a.nunits = N * (Ho / a.HS);
  // blocks: measured per kernel at 5120 frames : stride 1 62 us at 512, 55 at 768 / 1024; stride 2 flat
  int gx = stride == 1 ? 768 : 512;
  while (gx > 8 && (long)gx * 4 > a.nunits) gx -= 8;
  const bool pro = pro_scale != nullptr;
  const size_t lds = 2048 + 4 * (size_t)((3 + stride) * (stride * 16 + 2) * 64);
#define MMVAE_C3(S_, SC_, P_) hipLaunchKernelGGL((conv3_stream_kernel<S_, SC_, P_>), dim3(gx), dim3(256), lds, s, a)
  if (stride == 2) { if (wsc) { if (pro) MMVAE_C3(2, true, true); else MMVAE_C3(2, true, false); } else { if (pro) MMVAE_C3(2, false, true); else MMVAE_C3(2, false, false); } }
  else { if (pro) MMVAE_C3(1, false, true); else MMVAE_C3(1, false, false); }
#undef MMVAE_C3
  note_launch_bytes((double)N * 32 * 2.0 * ((double)a.Hi * a.Hi + (double)Ho * Ho * (wsc ? 2 : 1)));
  const int rc = check_launch("conv3_stream");
  return rc ? rc : gx;
}

// The data gradient of a 3x3 stride-1 32 -> 32 Conv2d as a forward conv over dy (w: the conv's weights packed [cin][flipped tap][cout]) with
// the backward sums of the BatchNorm + ReLU in front of that conv from the same pass (rows [sum g][sum g * ym], g masked by ym * ms + mb > 0):
// encoder.layer1.conv2's dgrad and bn1's reduce in one kernel.  Returns the number of partial rows (> 0) or an error.
int launch_conv3_stream_bwd(int dt, const void* dy, const void* w_flipped, void* dx, const void* ym, const float* ms, const float* mb, float* sums,
                            int N, int H, hipStream_t s) {
  if (dt != DT_BF16 || H != 16 || !ym || !ms || !mb || !sums) { set_error("conv3_stream_bwd: bf16, 16x16 maps, mask operand and sums required"); return MMVAE_ERR_UNSUPPORTED; }
  Conv3StreamArgs a; memset(&a, 0, sizeof(a));
  a.x = dy; a.w = w_flipped; a.y = dx; a.stats = sums; a.ym = ym; a.ms = ms; a.mb = mb; a.N = N; a.Ho = H; a.Hi = H;
  a.HS = 8; a.nunits = N * (H / a.HS);
  int gx = 768;
  while (gx > 8 && (long)gx * 4 > a.nunits) gx -= 8;
  const size_t lds = 2048 + 4 * (size_t)(4 * 18 * 64);
  hipLaunchKernelGGL((conv3_stream_kernel<1, false, false, true>), dim3(gx), dim3(256), lds, s, a);
  note_launch_bytes((double)N * 32 * 2.0 * 3.0 * H * H);
  const int rc = check_launch("conv3_stream_bwd");
  return rc ? rc : gx;
}

// ---------------------------------------------------------------- last up-block join + tail conv (one output plane) as a stream
// tail_fwd_stream_kernel: r_raw[n,0,h,w] = bias + sum_{kh,kw,c} x[n, h+kh-1, w+kw-1, c] * w[0][c][kh][kw] with the joined activation
// x = relu(y2*s2+b2 + ys*ss+bs) recomputed on the way into a wave-private ring of rows (never stored).  tail_join_fwd_kernel (bn_elem.hip)
// walks one image per block with per-tap dot products on the VALU and reads its 1.34 GB at 3.6 TB/s; here every wave streams strips of
// rows, the next rows are in flight in registers, and the 16 x 9 reduction per pixel runs on the MFMA: two taps (2 x 16 channels) are one
// K = 32 step, 6 steps per 16-pixel tile, the 15 unused output rows of the fragment cost nothing.  BatchNorm sums of the output
// (one channel) are kept per wave over all its rows; one partial row per block.
struct TailFwdStreamArgs {
  const void* y2; const void* ys; const float* s2; const float* b2; const float* ss; const float* bs;
  const float* w; const float* bias; float* r_raw; float* stats;
  int N, H, HS, nunits;
};

// F8I: y2 / ys are e4m3 bytes (16 per pixel): a lane stages one whole pixel per row instead of two half pixels
template <bool F8I>
__global__ __launch_bounds__(256, 2) void tail_fwd_stream_kernel(TailFwdStreamArgs a) {
  constexpr int KS = 3, W = 64, CB = 32;                                   // bytes per pixel (16 channels of bf16)
  constexpr int NC = F8I ? 16 : 8;                                         // channels a lane stages
  constexpr int WL = W + 2, ROWB = WL * CB, NSLOT = KS + 1, WAVE_LDS = NSLOT * ROWB, NPRIME = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* ring = smem + 64 + wv * WAVE_LDS;
  for (int s = 0; s < NSLOT; ++s) {
    if (lane < 2) *reinterpret_cast<Vec16*>(ring + s * ROWB + lane * 16) = Vec16{{0, 0, 0, 0}};
    else if (lane < 4) *reinterpret_cast<Vec16*>(ring + s * ROWB + (WL - 1) * CB + (lane - 2) * 16) = Vec16{{0, 0, 0, 0}};
  }
  // A fragments, per-tap-column form (see up5_tail_fwd_kernel): D[kw][column] over two joined rows per MFMA (kh = 0, 1) + the third row
  // (kh = 2) in a second one; rows m = kw, every joined pixel is read once per output row
  Vec16 wA[2];
  {
    float f0[8], f1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 8 * (gq & 1) + j;
      f0[j] = r < 3 ? a.w[c * 9 + (gq >> 1) * 3 + r] : 0.f;
      f1[j] = (r < 3 && gq < 2) ? a.w[c * 9 + 2 * 3 + r] : 0.f;
    }
    wA[0] = Elem<bf16_t>::pack(f0); wA[1] = Elem<bf16_t>::pack(f1);
  }
  const float bias = a.bias ? a.bias[0] : 0.f;
  // join coefficients of the 8 channels this lane stages (vector lane + 64k of a row: channels 8 * (lane & 1) ..)
  float c2s[NC], c2b[NC], css[NC];
  {
    const int c = F8I ? 0 : (lane & 1) * 8;
#pragma unroll
    for (int j = 0; j < NC; ++j) { c2s[j] = a.s2[c + j]; c2b[j] = a.b2[c + j] + a.bs[c + j]; css[j] = a.ss[c + j]; }
  }
  float st1 = 0.f, st2 = 0.f;
  const int nblk = gridDim.x;
  int u_first, u_step, u_end;
  if ((nblk & 7) == 0) {
    const int per = (a.nunits + 7) >> 3;
    const int lo_u = (blockIdx.x & 7) * per;
    u_first = lo_u + (blockIdx.x >> 3) * 4 + wv; u_step = (nblk >> 3) * 4; u_end = min(a.nunits, lo_u + per);
  } else { u_first = blockIdx.x * 4 + wv; u_step = nblk * 4; u_end = a.nunits; }
  const int nstrips = a.H / a.HS;
  const int nq = a.HS + NPRIME;
  Vec16 v2[2], vs[2];
  auto issue = [&](int u, int q) {
    const int n = u / nstrips, oh0 = (u - n * nstrips) * a.HS;
    const int row = oh0 + q - NPRIME + 1;                                   // the row arriving at step q
#pragma unroll
    for (int k = 0; k < (F8I ? 1 : 2); ++k) {
      v2[k] = Vec16{{0, 0, 0, 0}}; vs[k] = Vec16{{0, 0, 0, 0}};
      if (row >= 0 && row < a.H) {
        const long off = (((long)n * a.H + row) * W) * (F8I ? 16 : CB) + (lane + 64 * k) * 16;
        v2[k] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.y2) + off);
        vs[k] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.ys) + off);
      }
    }
  };
  auto commit = [&](int u, int q) {
    const int n = u / nstrips, oh0 = (u - n * nstrips) * a.HS;
    const int row = oh0 + q - NPRIME + 1;
    const bool in = row >= 0 && row < a.H;
    if constexpr (F8I) {
      float f2[16], fs[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) { unpack4_fp8(v2[0].w[q4], f2 + 4 * q4); unpack4_fp8(vs[0].w[q4], fs + 4 * q4); }
#pragma unroll
      for (int j = 0; j < 16; ++j) f2[j] = in ? fmaxf(f2[j] * c2s[j] + c2b[j] + fs[j] * css[j], 0.f) : 0.f;
      char* dstp = ring + ((row + 4 * NSLOT) % NSLOT) * ROWB + CB + lane * 32;
      *reinterpret_cast<Vec16*>(dstp) = Elem<bf16_t>::pack(f2);
      *reinterpret_cast<Vec16*>(dstp + 16) = Elem<bf16_t>::pack(f2 + 8);
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float f2[8], fs[8];
        Elem<bf16_t>::unpack(v2[k], f2);
        Elem<bf16_t>::unpack(vs[k], fs);
#pragma unroll
        for (int j = 0; j < 8; ++j) f2[j] = in ? fmaxf(f2[j] * c2s[j] + c2b[j] + fs[j] * css[j], 0.f) : 0.f;
        *reinterpret_cast<Vec16*>(ring + ((row + 4 * NSLOT) % NSLOT) * ROWB + CB + (lane + 64 * k) * 16) = Elem<bf16_t>::pack(f2);
      }
    }
    (void)n;
  };
  int u = u_first, q = 0;
  if (u < u_end) issue(u, 0);
  while (u < u_end) {
    commit(u, q);
    int un = u, qn = q + 1;
    if (qn == nq) { un = u + u_step; qn = 0; }
    if (un < u_end) issue(un, qn);
    __builtin_amdgcn_sched_barrier(0);
    if (q >= NPRIME) {
      const int n = u / nstrips, oh = (u - n * nstrips) * a.HS + q - NPRIME;
      const char* row01 = ring + ((oh - 1 + (gq >> 1) + 4 * NSLOT) % NSLOT) * ROWB;       // this lane's k half: joined row oh - 1 or oh
      const char* row2 = ring + ((oh + 1 + 4 * NSLOT) % NSLOT) * ROWB;
      float d0[4], d1[4], d2[4];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const int po = (16 * pt + r + 1) * CB + (gq & 1) * 16;                             // column 16 pt + r (ring index + 1: the halo)
        f32x4 acc = mma_bf16(wA[0], *reinterpret_cast<const Vec16*>(row01 + po), (f32x4){0, 0, 0, 0});
        acc = mma_bf16(wA[1], *reinterpret_cast<const Vec16*>(row2 + po), acc);
        d0[pt] = acc[0]; d1[pt] = acc[1]; d2[pt] = acc[2];
        asm volatile("" : "+v"(d0[pt]), "+v"(d1[pt]), "+v"(d2[pt]));                       // (pinned: see up5_tail_fwd_kernel)
      }
      // lanes gq = 0: out[p] = D[0][p - 1] + D[1][p] + D[2][p + 1], neighbours through DPP row shifts (tile edges from the adjacent tile)
      float outv[4];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        float left = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d0[pt]), 0x111, 0xf, 0xf, true));          // row_shr:1
        if (pt > 0) {
          const float pl = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d0[pt - 1]), 0x121, 0xf, 0xf, true)); // row_ror:1
          left = r == 0 ? pl : left;
        }
        float right = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d2[pt]), 0x101, 0xf, 0xf, true));         // row_shl:1
        if (pt < 3) {
          const float nr = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d2[pt + 1]), 0x12f, 0xf, 0xf, true)); // row_ror:15
          right = r == 15 ? nr : right;
        }
        outv[pt] = (left + d1[pt]) + right + bias;
      }
      if (gq == 0) {
        float* orow = a.r_raw + ((long)n * a.H + oh) * W + r;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          orow[16 * pt] = outv[pt];
          st1 += outv[pt]; st2 += outv[pt] * outv[pt];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    u = un; q = qn;
  }
  // sums: lanes gq = 0 of a wave (16 lanes = one DPP row), then the four waves in order
  st1 = row16_sum(st1); st2 = row16_sum(st2);
  float* sb = reinterpret_cast<float*>(smem);
  if (lane == 0) { sb[wv * 2] = st1; sb[wv * 2 + 1] = st2; }
  __syncthreads();
  if (a.stats && t < 2) a.stats[(long)blockIdx.x * 2 + t] = (sb[t] + sb[2 + t]) + (sb[4 + t] + sb[6 + t]);
}

// ---------------------------------------------------------------- ConvTranspose2d(16 -> 16, k4 s2 p1) forward as a stream
// convT4_stream_kernel: the decoder's widest layers (uplayer5.conv2 / .upsample: 32x32 -> 64x64, 839 MB each at N = 5120; uplayer4.conv2:
// 16x16 -> 32x32).  The patch-tile kernel moves them at 1.5-3.5 TB/s.  Here a wave keeps a ring of INPUT rows (fused BN+ReLU on the way
// in) and produces the two output rows 2q, 2q+1 of input row q: per output row and column parity the 2 x 2 contributing taps x 16
// channels are two K = 32 MFMA steps per 16-pixel tile (A = the packed per-phase weights, 8 fragments resident; B = 16-byte reads of
// ring pixels), the results of both parities are interleaved into a wave-private LDS row and leave as whole 16-byte-per-lane stores;
// BatchNorm sums from the f32 results.
struct ConvT4StreamArgs {
  const void* x; const void* w; void* y;
  const float* pro_scale; const float* pro_shift; int pro_relu;
  float* stats;                           // partial rows [blocks][2][16] (nullable)
  int N, Hi, HS, nunits;
};

// F8O (fp8 mode, the 32x32 -> 64x64 layers): the output leaves as e4m3 bytes, 16 per pixel (the statistics are still those of the f32 results)
template <int WIN, bool PRO, bool F8O = false>
__global__ __launch_bounds__(256, 3) void convT4_stream_kernel(ConvT4StreamArgs a) {
  constexpr int CB = 32, WL = WIN + 2, ROWB = WL * CB, NSLOT = 4, NPRIME = 2;
  constexpr int CBO = F8O ? 16 : 32;                                        // bytes per output pixel
  constexpr int OROWB = 2 * WIN * CBO;                                      // one output row
  constexpr int WAVE_LDS = NSLOT * ROWB + OROWB;
  constexpr int XV = (WIN * CB + 1023) / 1024;                              // input-row vectors per lane (WIN = 32: 1; 16: 1, half the lanes)
  constexpr int NPT = WIN / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* ring = smem + 1024 + wv * WAVE_LDS;
  char* orow = ring + NSLOT * ROWB;
  for (int s = 0; s < NSLOT; ++s) {
    if (lane < 2) *reinterpret_cast<Vec16*>(ring + s * ROWB + lane * 16) = Vec16{{0, 0, 0, 0}};
    else if (lane < 4) *reinterpret_cast<Vec16*>(ring + s * ROWB + (WL - 1) * CB + (lane - 2) * 16) = Vec16{{0, 0, 0, 0}};
  }
  // A[phase = 2*ph + pw][th]: row r = output channel, k = 8gq ..: tap (th, tw = gq >> 1) of the phase, input channels 8*(gq & 1) ..
  Vec16 wA[4][2];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int th = 0; th < 2; ++th)
      wA[p][th] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.w) + p * 2048 + r * 128 + th * 64 + gq * 16);
  // input offsets of the taps (op_pack_up's order): rows ph = 0: dh = {0, -1}, ph = 1: {+1, 0}; columns likewise
  const int dwl[2] = {(gq >> 1) ? -1 : 0, (gq >> 1) ? 0 : 1};               // this lane's column offset for pw = 0 / 1
  float psc[8], psh[8];
  if (PRO) {
    const int c = (lane & 1) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { psc[j] = a.pro_scale[c + j]; psh[j] = a.pro_shift[c + j]; }
  }
  const float lo = a.pro_relu ? 0.f : -__builtin_inff();
  float st1[4] = {0, 0, 0, 0}, st2[4] = {0, 0, 0, 0};
  const int nblk = gridDim.x;
  int u_first, u_step, u_end;
  if ((nblk & 7) == 0) {
    const int per = (a.nunits + 7) >> 3;
    const int lo_u = (blockIdx.x & 7) * per;
    u_first = lo_u + (blockIdx.x >> 3) * 4 + wv; u_step = (nblk >> 3) * 4; u_end = min(a.nunits, lo_u + per);
  } else { u_first = blockIdx.x * 4 + wv; u_step = nblk * 4; u_end = a.nunits; }
  const int nstrips = a.Hi / a.HS;
  const int nq = a.HS + NPRIME;
  const bool xlane = lane * 16 < WIN * CB;                                  // WIN = 16: a row is 512 bytes, lanes 0-31 stage it
  Vec16 xv = Vec16{{0, 0, 0, 0}};
  auto issue = [&](int u, int q) {
    const int n = u / nstrips, q0 = (u - n * nstrips) * a.HS;
    const int row = q0 + q - NPRIME + 1;
    xv = Vec16{{0, 0, 0, 0}};
    if (xlane && row >= 0 && row < a.Hi) xv = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.x) + (((long)n * a.Hi + row) * WIN) * CB + lane * 16);
  };
  auto commit = [&](int u, int q) {
    const int n = u / nstrips, q0 = (u - n * nstrips) * a.HS;
    const int row = q0 + q - NPRIME + 1;
    if (xlane) {
      Vec16 v = xv;
      if (PRO && row >= 0 && row < a.Hi) {
        float f[8];
        Elem<bf16_t>::unpack(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * psc[j] + psh[j], lo);
        v = Elem<bf16_t>::pack(f);
      }
      *reinterpret_cast<Vec16*>(ring + ((row + 4 * NSLOT) % NSLOT) * ROWB + CB + lane * 16) = v;
    }
    (void)n; (void)XV;
  };
  int u = u_first, q = 0;
  if (u < u_end) issue(u, 0);
  while (u < u_end) {
    commit(u, q);
    int un = u, qn = q + 1;
    if (qn == nq) { un = u + u_step; qn = 0; }
    if (un < u_end) issue(un, qn);
    __builtin_amdgcn_sched_barrier(0);
    if (q >= NPRIME) {
      const int n = u / nstrips, iq = (u - n * nstrips) * a.HS + q - NPRIME;          // input row
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          f32x4 acc[NPT];
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) acc[pt] = (f32x4){0, 0, 0, 0};
#pragma unroll
          for (int th = 0; th < 2; ++th) {
            const int dh = ph == 0 ? (th == 0 ? 0 : -1) : (th == 0 ? 1 : 0);
            const char* rowp = ring + ((iq + dh + 4 * NSLOT) % NSLOT) * ROWB;
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
              const Vec16 b = *reinterpret_cast<const Vec16*>(rowp + (16 * pt + r + dwl[pw] + 1) * CB + (gq & 1) * 16);
              acc[pt] = mma_bf16(wA[2 * ph + pw][th], b, acc[pt]);
            }
          }
          // lane (r = input column p, gq) holds output channels 4gq .. 4gq+3 of output pixel 2p + pw
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { st1[j] += acc[pt][j]; st2[j] += acc[pt][j] * acc[pt][j]; }
            if constexpr (F8O) {
              *reinterpret_cast<uint32_t*>(orow + (2 * (16 * pt + r) + pw) * CBO + gq * 4) = pack4_fp8(acc[pt][0], acc[pt][1], acc[pt][2], acc[pt][3]);
            } else {
              uint2 o;
              o.x = pack2_bf16(acc[pt][0], acc[pt][1]); o.y = pack2_bf16(acc[pt][2], acc[pt][3]);
              *reinterpret_cast<uint2*>(orow + (2 * (16 * pt + r) + pw) * CB + gq * 8) = o;
            }
          }
        }
        // the finished output row 2*iq + ph leaves as 16 bytes per lane (compiler fence: stored as 4- / 8-byte vectors, read back as 16-byte ones)
        asm volatile("" ::: "memory");
        char* dst = reinterpret_cast<char*>(a.y) + (((long)n * (2 * a.Hi) + 2 * iq + ph) * (2 * WIN)) * CBO;
#pragma unroll
        for (int k = 0; k < OROWB / 1024; ++k) *reinterpret_cast<Vec16*>(dst + (lane + 64 * k) * 16) = *reinterpret_cast<const Vec16*>(orow + (lane + 64 * k) * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    u = un; q = qn;
  }
  float* sb = reinterpret_cast<float*>(smem);                               // [4 waves][32]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v1 = row16_sum(st1[j]), v2 = row16_sum(st2[j]);
    if (r == 0) { sb[wv * 32 + 4 * gq + j] = v1; sb[wv * 32 + 16 + 4 * gq + j] = v2; }
  }
  __syncthreads();
  if (a.stats && t < 32) a.stats[(long)blockIdx.x * 32 + t] = (sb[t] + sb[32 + t]) + (sb[64 + t] + sb[96 + t]);
}

// ---------------------------------------------------------------- last up-block: join + tail conv with the branch outputs RECOMPUTED
// up5_tail_fwd_kernel: tail_fwd_stream_kernel reads the two 64x64x16 branch outputs (1.34 GB at 5120 frames) that convT4_stream_kernel wrote a
// moment ago; the forward kernels of this block are memory-bound (e4m3 storage of those tensors makes them 28-35 % faster, DESIGN lesson 36).
// Here a wave keeps rings of the two ConvTranspose2d INPUT rows (32x32x16: 0.34 GB together) and recomputes both branch outputs for the
// two output rows of an input row exactly like convT4_stream_kernel does (same MFMA sequence, the results rounded to bf16 like the stored
// tensors), joins them in the D-fragment mapping (lane = 4 channels of one output pixel) into the ring of joined rows the tail conv reads.
// y2 / ys are still written by convT4_stream_kernel (statistics; the backward pass reads them): this kernel only stops READING them.
struct Up5TailFwdArgs {
  const void* y1; const float* p1s; const float* p1b; const void* w2;      // conv2 branch: input (pre-bn1), bn1 scale / shift, packed up weights
  const void* xin; const float* pxs; const float* pxb; const void* wu;     // upsample branch: block input (+ optional BatchNorm+ReLU), packed up weights
  const float* s2; const float* b2; const float* ss; const float* bs;      // join coefficients (forward scale / shift of bn2 and of the shortcut's BatchNorm)
  const float* w; const float* bias; float* r_raw; float* stats;
  int N, HS, nunits;
};

template <bool PRO_X>
__global__ __launch_bounds__(256, 2) void up5_tail_fwd_kernel(Up5TailFwdArgs a) {
  constexpr int WIN = 32, Hi = 32, W = 64, H = 64, CB = 32;
  constexpr int IWL = WIN + 2, IROWB = IWL * CB, NSLOT = 4;                // input rings: 4 rows of 34 pixels
  constexpr int JWL = W + 2, JROWB = JWL * CB;                              // joined ring: 4 rows of 66 pixels
  constexpr int WAVE_LDS = 2 * NSLOT * IROWB + NSLOT * JROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* ringA = smem + 64 + wv * WAVE_LDS;                                  // relu(bn1(y1)) rows
  char* ringX = ringA + NSLOT * IROWB;                                      // block-input rows
  char* ringJ = ringX + NSLOT * IROWB;                                      // joined rows
  for (int sl = 0; sl < NSLOT; ++sl) {
    if (lane < 2) { *reinterpret_cast<Vec16*>(ringA + sl * IROWB + lane * 16) = Vec16{{0, 0, 0, 0}}; *reinterpret_cast<Vec16*>(ringX + sl * IROWB + lane * 16) = Vec16{{0, 0, 0, 0}}; }
    else if (lane < 4) { *reinterpret_cast<Vec16*>(ringA + sl * IROWB + (IWL - 1) * CB + (lane - 2) * 16) = Vec16{{0, 0, 0, 0}}; *reinterpret_cast<Vec16*>(ringX + sl * IROWB + (IWL - 1) * CB + (lane - 2) * 16) = Vec16{{0, 0, 0, 0}}; }
    else if (lane < 6) *reinterpret_cast<Vec16*>(ringJ + sl * JROWB + (lane - 4) * 16) = Vec16{{0, 0, 0, 0}};
    else if (lane < 8) *reinterpret_cast<Vec16*>(ringJ + sl * JROWB + (JWL - 1) * CB + (lane - 6) * 16) = Vec16{{0, 0, 0, 0}};
  }
  // ConvTranspose2d A fragments of both branches (as in convT4_stream_kernel): [phase = 2 ph + pw][th]
  Vec16 w2A[4][2], wuA[4][2];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int th = 0; th < 2; ++th) {
      w2A[p][th] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.w2) + p * 2048 + r * 128 + th * 64 + gq * 16);
      wuA[p][th] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.wu) + p * 2048 + r * 128 + th * 64 + gq * 16);
    }
  const int dwl[2] = {(gq >> 1) ? -1 : 0, (gq >> 1) ? 0 : 1};
  // tail conv, per-tap-column form: D[kw][n] = sum_{kh, c} w[c][kh][kw] J[oh - 1 + kh][column n][c] -- rows m = kw (3 of 16 real), K = two
  // joined rows x 16 channels per MFMA (kh = 0, 1), the third row (kh = 2) in a second MFMA whose upper K half is zero.  Every joined pixel
  // is read ONCE per output row (2 B-operand reads per 16-pixel tile instead of 6); the three columns a pixel needs are combined across
  // lanes with DPP row shifts: out[p] = D[0][p - 1] + D[1][p] + D[2][p + 1].
  Vec16 wT[2];
  {
    float f0[8], f1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 8 * (gq & 1) + j;
      f0[j] = r < 3 ? a.w[c * 9 + (gq >> 1) * 3 + r] : 0.f;                  // k = (kh = gq >> 1, c)
      f1[j] = (r < 3 && gq < 2) ? a.w[c * 9 + 2 * 3 + r] : 0.f;              // k = (kh = 2, c) in the lower half, zeros above
    }
    wT[0] = Elem<bf16_t>::pack(f0); wT[1] = Elem<bf16_t>::pack(f1);
  }
  const float bias = a.bias ? a.bias[0] : 0.f;
  // prologue coefficients of the 8 channels this lane stages per input row; join coefficients of this lane's 4 output channels
  float p1s[8], p1b[8], pxs[8], pxb[8];
  {
    const int c = (lane & 1) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p1s[j] = a.p1s[c + j]; p1b[j] = a.p1b[c + j]; pxs[j] = PRO_X ? a.pxs[c + j] : 1.f; pxb[j] = PRO_X ? a.pxb[c + j] : 0.f; }
  }
  float js2[4], jb[4], jss[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int c = 4 * gq + j; js2[j] = a.s2[c]; jb[j] = a.b2[c] + a.bs[c]; jss[j] = a.ss[c]; }
  float st1 = 0.f, st2 = 0.f;
  const int nblk = gridDim.x;
  int u_first, u_step, u_end;
  if ((nblk & 7) == 0) {
    const int per = (a.nunits + 7) >> 3;
    const int lo_u = (blockIdx.x & 7) * per;
    u_first = lo_u + (blockIdx.x >> 3) * 4 + wv; u_step = (nblk >> 3) * 4; u_end = min(a.nunits, lo_u + per);
  } else { u_first = blockIdx.x * 4 + wv; u_step = nblk * 4; u_end = a.nunits; }
  const int nstrips = Hi / a.HS;
  // steps of a unit: q = 0 primes input rows q0 - 2, ... ; step q >= 2 runs input row iq = q0 - 1 + (q - 2), i.e. iq = q0 - 1 .. q0 + HS
  constexpr int NPRIME = 2;
  const int nq = a.HS + 2 + NPRIME;
  Vec16 va = Vec16{{0, 0, 0, 0}}, vx = Vec16{{0, 0, 0, 0}};
  auto issue = [&](int u, int q) {
    const int n = u / nstrips, q0 = (u - n * nstrips) * a.HS;
    const int row = q0 - 1 + q - NPRIME + 1;                                // the input row arriving at step q: one ahead of the row computed
    va = Vec16{{0, 0, 0, 0}}; vx = Vec16{{0, 0, 0, 0}};
    if (row >= 0 && row < Hi) {
      const long off = (((long)n * Hi + row) * WIN) * CB + lane * 16;
      va = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.y1) + off);
      vx = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.xin) + off);
    }
  };
  auto commit = [&](int u, int q) {
    const int n = u / nstrips, q0 = (u - n * nstrips) * a.HS;
    const int row = q0 - 1 + q - NPRIME + 1;
    Vec16 ca = va, cx = vx;
    if (row >= 0 && row < Hi) {
      float f[8];
      Elem<bf16_t>::unpack(ca, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * p1s[j] + p1b[j], 0.f);
      ca = Elem<bf16_t>::pack(f);
      if (PRO_X) {
        Elem<bf16_t>::unpack(cx, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * pxs[j] + pxb[j], 0.f);
        cx = Elem<bf16_t>::pack(f);
      }
    }
    const int slot = (row + 4 * NSLOT) % NSLOT;
    *reinterpret_cast<Vec16*>(ringA + slot * IROWB + CB + lane * 16) = ca;
    *reinterpret_cast<Vec16*>(ringX + slot * IROWB + CB + lane * 16) = cx;
    (void)n;
  };
  int u = u_first, q = 0;
  if (u < u_end) issue(u, 0);
  while (u < u_end) {
    commit(u, q);
    int un = u, qn = q + 1;
    if (qn == nq) { un = u + u_step; qn = 0; }
    if (un < u_end) issue(un, qn);
    __builtin_amdgcn_sched_barrier(0);
    if (q >= NPRIME) {
      const int n = u / nstrips, q0 = (u - n * nstrips) * a.HS;
      const int iq = q0 - 1 + q - NPRIME;                                   // input row of this step: q0 - 1 .. q0 + HS
      const bool inside = iq >= 0 && iq < Hi;                               // (wave-uniform) rows outside the image: the tail conv's zero padding
      // ---- the two joined rows 2 iq, 2 iq + 1
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        char* jrow = ringJ + ((2 * iq + ph + 8 * NSLOT) % NSLOT) * JROWB;
        if (!inside) {
          for (int i = lane; i < (W * CB) / 16; i += 64) *reinterpret_cast<Vec16*>(jrow + CB + i * 16) = Vec16{{0, 0, 0, 0}};
          continue;
        }
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          f32x4 acc2[2], accS[2];
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) { acc2[pt] = (f32x4){0, 0, 0, 0}; accS[pt] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
          for (int th = 0; th < 2; ++th) {
            const int dh = ph == 0 ? (th == 0 ? 0 : -1) : (th == 0 ? 1 : 0);
            const int sl = (iq + dh + 4 * NSLOT) % NSLOT;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
              const int off = sl * IROWB + (16 * pt + r + dwl[pw] + 1) * CB + (gq & 1) * 16;
              acc2[pt] = mma_bf16(w2A[2 * ph + pw][th], *reinterpret_cast<const Vec16*>(ringA + off), acc2[pt]);
              accS[pt] = mma_bf16(wuA[2 * ph + pw][th], *reinterpret_cast<const Vec16*>(ringX + off), accS[pt]);
            }
          }
          // lane (r = input column p, gq) holds output channels 4gq .. 4gq+3 of output pixel 2p + pw: round like the stored tensors, join
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(acc2[pt][j] * js2[j] + jb[j] + accS[pt][j] * jss[j], 0.f);
            *reinterpret_cast<uint2*>(jrow + (1 + 2 * (16 * pt + r) + pw) * CB + gq * 8) = make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
          }
        }
      }
      // (the joined rows were stored as 8-byte vectors and are read back as 16-byte ones: without a compiler-level fence type-based alias
      // analysis may hoist those loads above the stores)
      asm volatile("" ::: "memory");
      // ---- the tail conv's output rows that have all three joined rows now: 2 iq - 1 and 2 iq (inside the strip and the image)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int oh = 2 * iq - 1 + k;
        if (oh < 2 * q0 || oh >= 2 * (q0 + a.HS) || oh < 0 || oh >= H) continue;
        f32x4 acc[4];
        const char* row01 = ringJ + ((oh - 1 + (gq >> 1) + 8 * NSLOT) % NSLOT) * JROWB;      // this lane's k half: joined row oh - 1 or oh
        const char* row2 = ringJ + ((oh + 1 + 8 * NSLOT) % NSLOT) * JROWB;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          const int po = (16 * pt + r + 1) * CB + (gq & 1) * 16;                             // column 16 pt + r (ring index + 1: the halo)
          acc[pt] = mma_bf16(wT[0], *reinterpret_cast<const Vec16*>(row01 + po), (f32x4){0, 0, 0, 0});
          acc[pt] = mma_bf16(wT[1], *reinterpret_cast<const Vec16*>(row2 + po), acc[pt]);
        }
        // lanes gq = 0 hold D[kw = 0..2][column 16 pt + r] in acc[pt][0..2]
        float d0[4], d1[4], d2[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          d0[pt] = acc[pt][0]; d1[pt] = acc[pt][1]; d2[pt] = acc[pt][2];
          // (hipcc 7.2 otherwise feeds element 0 to the DPP moves of element 2 as well -- seen in the ISA; pin the three values)
          asm volatile("" : "+v"(d0[pt]), "+v"(d1[pt]), "+v"(d2[pt]));
        }
        float outv[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          // D[0] of column p - 1: lane r - 1 of this tile, lane 15 of the previous one (or the zero halo column)
          float left = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d0[pt]), 0x111, 0xf, 0xf, true));          // row_shr:1
          if (pt > 0) {
            const float pl = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d0[pt - 1]), 0x121, 0xf, 0xf, true)); // row_ror:1
            left = r == 0 ? pl : left;
          }
          // D[2] of column p + 1: lane r + 1 of this tile, lane 0 of the next one
          float right = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d2[pt]), 0x101, 0xf, 0xf, true));         // row_shl:1
          if (pt < 3) {
            const float nr = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d2[pt + 1]), 0x12f, 0xf, 0xf, true)); // row_ror:15
            right = r == 15 ? nr : right;
          }
          outv[pt] = (left + d1[pt]) + right + bias;
        }
        if (gq == 0) {
          float* orow = a.r_raw + ((long)n * H + oh) * W + r;
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) {
            orow[16 * pt] = outv[pt];
            st1 += outv[pt]; st2 += outv[pt] * outv[pt];
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    u = un; q = qn;
  }
  st1 = row16_sum(st1); st2 = row16_sum(st2);
  float* sb = reinterpret_cast<float*>(smem);
  if (lane == 0) { sb[wv * 2] = st1; sb[wv * 2 + 1] = st2; }
  __syncthreads();
  if (a.stats && t < 2) a.stats[(long)blockIdx.x * 2 + t] = (sb[t] + sb[2 + t]) + (sb[4 + t] + sb[6 + t]);
}

bool up5_tail_fwd_ok(int dt, int OC, int C, int Cin, int Hin, int Hout) {
  constexpr bool env = true;
  return env && dt == DT_BF16 && OC == 1 && C == 16 && Cin == 16 && Hin == 32 && Hout == 64;
}
// returns the number of partial rows [rows][2] (> 0) or an error
int launch_up5_tail_fwd(const void* y1, const float* p1s, const float* p1b, const void* w2_up, const void* xin, const float* pxs, const float* pxb,
                        const void* wu_up, const float* s2, const float* b2, const float* ss, const float* bs, const float* w, const float* bias,
                        float* r_raw, float* stats, int N, hipStream_t s) {
  Up5TailFwdArgs a; memset(&a, 0, sizeof(a));
  a.y1 = y1; a.p1s = p1s; a.p1b = p1b; a.w2 = w2_up; a.xin = xin; a.pxs = pxs; a.pxb = pxb; a.wu = wu_up;
  a.s2 = s2; a.b2 = b2; a.ss = ss; a.bs = bs; a.w = w; a.bias = bias; a.r_raw = r_raw; a.stats = stats;
  a.N = N; a.HS = 16; a.nunits = N * (32 / a.HS);
  int gx = 512;                                              // two blocks per CU (LDS), one resident round; rows of `stats` <= N
  while (gx > 1 && (long)gx * 4 > a.nunits) gx -= gx > 8 ? 8 : 1;
  const size_t lds = 64 + 4 * (size_t)(2 * 4 * 34 * 32 + 4 * 66 * 32);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&up5_tail_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&up5_tail_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("up5_tail_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MMVAE_ERR_HIP; }
    attr_set = true;
  }
  if (pxs) hipLaunchKernelGGL(up5_tail_fwd_kernel<true>, dim3(gx), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(up5_tail_fwd_kernel<false>, dim3(gx), dim3(256), lds, s, a);
  note_launch_bytes((double)N * (2.0 * 32 * 32 * 16 * 2 + 64 * 64 * 4.0));
  const int rc = check_launch("up5_tail_fwd");
  return rc ? rc : gx;
}

bool convT4_stream_ok(int dt, int Cin, int Cout, int k, int s, int p, int Hin, int Win) {
  constexpr bool env = true;
  return env && dt == DT_BF16 && Cin == 16 && Cout == 16 && k == 4 && s == 2 && p == 1 && Hin == Win && (Win == 32 || Win == 16);
}
// y [N][2H][2H][16] = conv_transpose2d(x [N][H][H][16]) with the packed per-phase "up" weights; returns stats rows (> 0) or an error
int launch_convT4_stream(int dt, const void* x, const void* w_up, void* y, const float* pro_scale, const float* pro_shift, int pro_relu, float* stats,
                         int N, int Hin, hipStream_t s, int f8out) {
  if (!convT4_stream_ok(dt, 16, 16, 4, 2, 1, Hin, Hin)) { set_error("convT4_stream: bf16, 16 -> 16 channels, k4 s2 p1, 16x16 or 32x32 input"); return MMVAE_ERR_UNSUPPORTED; }
  ConvT4StreamArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.w = w_up; a.y = y; a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.pro_relu = pro_relu; a.stats = stats;
  a.N = N; a.Hi = Hin; a.HS = Hin % 8 == 0 ? 8 : Hin; a.nunits = N * (Hin / a.HS);
  // blocks : 32x32 inputs 324 / 293 us (the pair of uplayer5, on two streams) at 768, 308 / 253 at 1024,
  // 318 / 306 at 1280; 16x16 inputs 123 us at 768, 130 at 1024
  int gx = Hin == 32 ? 1024 : 768;
  while (gx > 8 && (long)gx * 4 > a.nunits) gx -= 8;
  const bool pro = pro_scale != nullptr;
  const size_t lds = 1024 + 4 * (size_t)(4 * (Hin + 2) * 32 + 2 * Hin * 32);
  if (f8out && Hin != 32) { set_error("convT4_stream: e4m3 output only for the 32x32 -> 64x64 layers"); return MMVAE_ERR_UNSUPPORTED; }
  if (f8out) { if (pro) hipLaunchKernelGGL((convT4_stream_kernel<32, true, true>), dim3(gx), dim3(256), lds, s, a); else hipLaunchKernelGGL((convT4_stream_kernel<32, false, true>), dim3(gx), dim3(256), lds, s, a); }
  else if (Hin == 32) { if (pro) hipLaunchKernelGGL((convT4_stream_kernel<32, true>), dim3(gx), dim3(256), lds, s, a); else hipLaunchKernelGGL((convT4_stream_kernel<32, false>), dim3(gx), dim3(256), lds, s, a); }
  else { if (pro) hipLaunchKernelGGL((convT4_stream_kernel<16, true>), dim3(gx), dim3(256), lds, s, a); else hipLaunchKernelGGL((convT4_stream_kernel<16, false>), dim3(gx), dim3(256), lds, s, a); }
  note_launch_bytes((double)N * 16 * (2.0 + (f8out ? 4.0 : 8.0)) * Hin * Hin);   // x + y = (1 + 4) Hin^2 pixels of 16 channels (bf16; y e4m3 with f8out)
  const int rc = check_launch("convT4_stream");
  return rc ? rc : gx;
}

bool tail_fwd_stream_ok(int dt, int OC, int H, int W) {
  constexpr bool env = true;
  return env && dt == DT_BF16 && OC == 1 && W == 64 && H == 64;
}
// returns the number of partial rows [rows][2][1] (> 0) or an error
int launch_tail_fwd_stream(int dt, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs, const float* w,
                           const float* bias, float* r_raw, float* stats, int N, int H, int W, hipStream_t s, int f8in) {
  if (!tail_fwd_stream_ok(dt, 1, H, W)) { set_error("tail_fwd_stream: bf16, one output plane, 64x64"); return MMVAE_ERR_UNSUPPORTED; }
  TailFwdStreamArgs a; memset(&a, 0, sizeof(a));
  a.y2 = y2; a.ys = ys; a.s2 = s2; a.b2 = b2; a.ss = ss; a.bs = bs; a.w = w; a.bias = bias; a.r_raw = r_raw; a.stats = stats;
  a.N = N; a.H = H; a.HS = 16; a.nunits = N * (H / a.HS);
  // blocks : 485 us at 256, 353 at 512, 333 at 768, 346 at 1024, 379 at 1280
  int gx = 768;                                              // rows of `stats` <= N: the entry point's contract is an [N][2] buffer
  while (gx > 1 && (long)gx * 4 > a.nunits) gx -= gx > 8 ? 8 : 1;
  const size_t lds = 64 + 4 * (size_t)(4 * 66 * 32);
  if (f8in) hipLaunchKernelGGL(tail_fwd_stream_kernel<true>, dim3(gx), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(tail_fwd_stream_kernel<false>, dim3(gx), dim3(256), lds, s, a);
  note_launch_bytes((double)N * H * W * (2 * 16 * (f8in ? 1.0 : 2.0) + 4.0));
  const int rc = check_launch("tail_fwd_stream");
  return rc ? rc : gx;
}

}  // namespace mmvae
