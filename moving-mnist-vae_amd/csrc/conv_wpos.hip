// wgrad_pos_kernel: weight gradients of the channel-heavy layers whose small-side map is up to 4x4 and large-side map up to 8x8 (bf16; round 4).
//
//   dW[a][b][tap] = sum_n sum_{pos} P[n, pos, a] * G[n, pix(pos, tap), b]      (P: small-side tensor, G: large-side tensor, conv_ops.hpp)
//
// as a split-K GEMM whose K index is the IMAGE: for a (position, tap) pair the two operands are the same pixel pair of 16 consecutive
// images, so pairs whose tap falls into the zero padding are skipped outright (3x3 p1 on a 2x2 map: 16 of the 36 pairs are real) and
// there is no im2col, no tap table and no halo.  wgrad2_kernel (conv_wgrad.inc) tiles PIXELS: on 2x2 / 4x4 maps its K-steps are mostly
// padding and it re-stages the patch per tap group: 8 % of the MFMA peak on encoder.layer4.conv2.
//
//   * block = one 64 x 64 tile of (a, b) channels, ALL taps, one contiguous range of images; wave w owns tap w: a 64 x 64 accumulator
//     tile = 2 x 2 fragments of v_mfma_f32_32x32x16_bf16 (64 registers) that stays in registers over the whole image range and is
//     stored ONCE as the block's partial image [split][tap][a][b] (f32), summed in a fixed order by wgrad_reduce_kernel.
//   * per stage of KI images the block stages P[KI][npos][64] and G[KI][npix][64] (BatchNorm + ReLU of the activation operand applied
//     on the way in) into one of two LDS buffers while the other is being multiplied (register prefetch, one barrier per stage).
//   * operand fragments (32 channels x 16 images) come out of the natural [image][pixel][channel] LDS layout with two
//     ds_read_b64_tr_b16 per lane; image pitch = 64 (mod 256) bytes makes every such read conflict-free (4 images x 64 bytes
//     per 32-lane half = all 64 banks).
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct WposArgs {
  const void* P; const void* G; float* part;
  const float* proP_scale; const float* proP_shift; int proP_relu;
  const float* proG_scale; const float* proG_shift; int proG_relu;
  int N, Ca, Cb, HS, HL, K, S, Pd;
  int nsplit, imgs_per_split;        // grid.x; images per block (a multiple of KI)
  // bands (round 4b): a stage holds KI images x (prow P rows, grow G rows) -- band b = P rows [b prow, (b + 1) prow) and the window of
  // `grow` G rows its taps reach, clamped into the map; nb = 1, prow = HS, grow = HL is the whole-image stage
  int nb, prow, grow;
};

// KI images per stage; NTHR threads (one wave per tap); kMaxV staging vectors per thread and stage; NFB: 32-channel fragments of the G tile
// (2: 64 x 64 channel tile, 1: 64 x 32 -- half the G bytes per stage, twice the tiles)
template <int KI, int NTHR, int kMaxV, int NFB>
__global__ __launch_bounds__(NTHR, 1) void wgrad_pos_kernel(WposArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int nthr = NTHR;
  constexpr int TB = 32 * NFB, PXG = TB * 2, VG = TB / 8;              // G tile channels, bytes and 16-byte vectors per staged G pixel
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int npos = a.HS * a.HS, npix = a.HL * a.HL;
  const int nPb = a.prow * a.HS, nGb = a.grow * a.HL;                  // pixels of a band's P rows / G window
  // image pitches = 64 (mod 256) bytes
  const int ipP = ((nPb * 128 + 191) & ~255) + 64, ipG = ((nGb * PXG + 191) & ~255) + 64;
  const int stageB = KI * (ipP + ipG);
  const int tiles_b = a.Cb / TB;
  const int a0 = (blockIdx.y / tiles_b) << 6, b0 = (blockIdx.y % tiles_b) * TB;
  const int n_begin = blockIdx.x * a.imgs_per_split;
  const int n_end = min(a.N, n_begin + a.imgs_per_split);
  const bf16_t* __restrict__ Pp = reinterpret_cast<const bf16_t*>(a.P);
  const bf16_t* __restrict__ Gp = reinterpret_cast<const bf16_t*>(a.G);
  const int kh = wv / a.K, kw = wv - kh * a.K;
  auto win0 = [&](int band) {                          // first G row of the band's window
    int g = band * a.prow * a.S - a.Pd;
    g = g < 0 ? 0 : g;
    return g > a.HL - a.grow ? a.HL - a.grow : g;
  };

  // ---- staging plan: a thread owns one (operand, pixel of the band, 8-channel group) slot and walks the stage's images in rounds of `ipr`
  // images; only the image index (and the band's pixel offset) changes between its vectors
  const int slots = nPb * 8 + nGb * VG;               // 16-byte vectors per image (both operands)
  const int ipr = nthr / slots;                       // images per round (launcher: >= 1, KI / ipr <= kMaxV)
  const bool active = t < ipr * slots;
  const int img_r = t / slots, rem = t - img_r * slots;
  const bool isP = rem < nPb * 8;
  const int px = isP ? rem >> 3 : (rem - nPb * 8) / VG, cg = isP ? rem & 7 : (rem - nPb * 8) % VG;
  // (at most one operand carries a fused BatchNorm + ReLU: the activation; the other one is a gradient -- the launcher's check)
  const bool pro = isP ? a.proP_scale != nullptr : a.proG_scale != nullptr;
  // its (scale, shift) rows for this block's channels wait in LDS (behind the two stage buffers): 16 registers the K loop keeps free
  float* sPro = reinterpret_cast<float*>(smem + 2 * stageB);
  if (t < 128) {
    const bool p = a.proP_scale != nullptr;
    const int c = t & 63, nc = p ? 64 : TB;
    const float* sc = p ? a.proP_scale + a0 : a.proG_scale ? a.proG_scale + b0 : nullptr;
    const float* sh = p ? a.proP_shift + a0 : a.proG_scale ? a.proG_shift + b0 : nullptr;
    sPro[t] = (!sc || c >= nc) ? (t < 64 ? 1.f : 0.f) : t < 64 ? sc[c] : sh[c];
  }
  const float lo = (isP ? a.proP_relu : a.proG_relu) ? 0.f : -__builtin_inff();
  const long istride = isP ? (long)npos * a.Ca : (long)npix * a.Cb;                       // elements per image of this thread's operand
  const bf16_t* src0 = isP ? Pp + (long)px * a.Ca + a0 + cg * 8 : Gp + (long)px * a.Cb + b0 + cg * 8;
  const int ip_t = isP ? ipP : ipG;
  const int dst0 = (isP ? 0 : KI * ipP) + img_r * ip_t + px * (isP ? 128 : PXG) + cg * 16;
  Vec16 q[kMaxV];
  auto fetch = [&](int n0, int band) {
    const long boff = isP ? (long)band * nPb * a.Ca : (long)win0(band) * a.HL * a.Cb;     // the band's first pixel
#pragma unroll
    for (int k = 0; k < kMaxV; ++k) {
      const int img = k * ipr + img_r;
      q[k] = Vec16{{0, 0, 0, 0}};
      if (active && img < KI && n0 + img < n_end) {
        q[k] = *reinterpret_cast<const Vec16*>(src0 + (long)(n0 + img) * istride + boff);
        if (pro) {
          float f[8];
          Elem<bf16_t>::unpack(q[k], f);
          const float4 s0 = *reinterpret_cast<const float4*>(sPro + cg * 8), s1 = *reinterpret_cast<const float4*>(sPro + cg * 8 + 4);
          const float4 h0 = *reinterpret_cast<const float4*>(sPro + 64 + cg * 8), h1 = *reinterpret_cast<const float4*>(sPro + 64 + cg * 8 + 4);
          const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], lo);
          q[k] = Elem<bf16_t>::pack(f);
        }
      }
    }
  };
  auto commit = [&](char* buf) {
#pragma unroll
    for (int k = 0; k < kMaxV; ++k)
      if (active && k * ipr + img_r < KI) *reinterpret_cast<Vec16*>(buf + dst0 + k * ipr * ip_t) = q[k];
  };

  f32x16 acc[2][NFB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NFB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // lane -> fragment address: channel (lane & 15) + 16 * ((lane >> 4) & 1) of a 32-channel fragment, images 8 * (lane >> 5) + {0..3 | 4..7};
  // a ds_read_b64_tr_b16 lane supplies the address of row (i >> 2), channel quad (i & 3) of its 16-lane group's 4 x 16 block
  const int li = lane & 15;
  const int lrow = ((lane >> 5) << 3) + (li >> 2);                    // image row of the first read (+4: second read)
  const int lcol = (((lane >> 4) & 1) << 5) + ((li & 3) << 3);        // byte offset inside the 64-byte fragment row
  __syncthreads();                                    // sPro
  // stages in order: (image chunk 0, band 0), (chunk 0, band 1), ..., (chunk 1, band 0), ...
  const int nchunk = (n_end - n_begin + KI - 1) / KI, nstage = nchunk * a.nb;
  fetch(n_begin, 0);
  commit(smem);
  __syncthreads();
  int cur = 0, band = 0, n0 = n_begin;
  for (int st = 0; st < nstage; ++st) {
    int bn = band + 1, nn = n0;
    if (bn == a.nb) { bn = 0; nn = n0 + KI; }
    const bool more = st + 1 < nstage;
    if (more) fetch(nn, bn);
    const char* sP = smem + cur * stageB;
    const char* sG = sP + KI * ipP;
    const int g0 = win0(band);
    for (int hr = 0; hr < a.prow; ++hr) {
      const int hs = band * a.prow + hr;
      const int hl = hs * a.S - a.Pd + kh;
      if (hl < g0 || hl >= g0 + a.grow || hl < 0 || hl >= a.HL) continue;
      for (int ws = 0; ws < a.HS; ++ws) {
        const int wl = ws * a.S - a.Pd + kw;
        if (wl < 0 || wl >= a.HL) continue;
        const char* pb = sP + (hr * a.HS + ws) * 128 + lcol;
        const char* gb = sG + ((hl - g0) * a.HL + wl) * PXG + lcol;
#pragma unroll
        for (int ks = 0; ks < KI / 16; ++ks) {
          bf16x8 fa[2], fb[NFB];
          typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const char* pa = pb + (ks * 16 + lrow) * ipP + f * 64;
            const s16x4 a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
            const s16x4 a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * ipP));
            fa[f] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
#pragma unroll
          for (int f = 0; f < NFB; ++f) {
            const char* pg = gb + (ks * 16 + lrow) * ipG + f * 64;
            const s16x4 g_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pg));
            const s16x4 g_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pg + 4 * ipG));
            fb[f] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(g_lo, g_hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NFB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (more) commit(smem + (cur ^ 1) * stageB);
    __syncthreads();
    cur ^= 1; band = bn; n0 = nn;
  }
  // ---- the block's partial image [split][tap][a][b]: acc[i][j][v] = D[row 8 (v / 4) + 4 (lane >> 5) + v % 4][col lane & 31]
  const int ntaps = a.K * a.K;
  float* out = a.part + (((long)blockIdx.x * ntaps + wv) * a.Ca + a0) * a.Cb + b0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NFB; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = 32 * i + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3), col = 32 * j + (lane & 31);
        out[(long)row * a.Cb + col] = acc[i][j][v];
      }
}

// Returns 1 when taken (kernel + ordered reduce launched), 0 when the shape is not this kernel's, < 0 on error.
int try_wgrad_pos(int dt, const WgradArgs& a, hipStream_t s) {
  if (dt != DT_BF16 || a.P_planar || a.G_planar || !a.scratch || (a.proP_scale && a.proG_scale)) return 0;
  if (a.Hp != a.Wp || a.Hg != a.Wg || a.Hp < 1 || a.Hp > 4 || a.Hg > 8 || a.Ca % 64 != 0 || a.Cb % 32 != 0) return 0;
  if (a.ksz < 1 || a.ksz > 4 || a.ntaps != a.ksz * a.ksz || a.ksz * a.ksz < 4) return 0;       // one wave per tap: 4, 9 or 16 waves
  if ((a.Ca_valid && a.Ca_valid != a.Ca) || (a.Cb_valid && a.Cb_valid != a.Cb)) return 0;
  for (int t = 0; t < a.ntaps; ++t) if (a.tap_off[t] != t) return 0;
  const int npos = a.Hp * a.Hp, npix = a.Hg * a.Hg;
  if (npos + npix < 8) return 0;                      // 1x1 | 2x2 (decoder.conv1): 22 us here, 20 on wgrad2_kernel
  const int nthr = 64 * a.ntaps;
  // 64 x 64 channel tiles where they give >= 8 tiles (>= 256 blocks at <= 32 image splits), else 64 x 32: twice the tiles, half the G bytes
  const int nfb = (a.Cb % 64 == 0 && (a.Ca / 64) * (a.Cb / 64) >= 8) ? 2 : 1;
  const int TB = 32 * nfb, VG = TB / 8;
  const int tiles = (a.Ca / 64) * (a.Cb / TB);
  // the smallest number of row bands whose two stages fit the LDS and whose staging fits the threads' vector slots
  int KI = 16, nb = 0, prow = 0, grow = 0, rounds = 0;
  size_t stageB = 0;
  const int maxv = nthr == 1024 ? 4 : 8;
  for (int cand = 1; cand <= a.Hp && !nb; cand *= 2) {
    if (a.Hp % cand) continue;
    const int pr = a.Hp / cand;
    int gr = (pr - 1) * a.stride + a.ksz; if (gr > a.Hg) gr = a.Hg;
    const int nPb = pr * a.Hp, nGb = gr * a.Hg;
    const int ipP = ((nPb * 128 + 191) & ~255) + 64, ipG = ((nGb * TB * 2 + 191) & ~255) + 64;
    const int slots = nPb * 8 + nGb * VG;
    const int ipr = nthr / slots;
    if (ipr < 1) continue;
    for (int ki = (cand == 1 && npos + npix <= 8) ? 32 : 16; ki >= 16; ki -= 16) {
      const size_t sb = (size_t)ki * (ipP + ipG);
      const int rd = (ki + ipr - 1) / ipr;
      if (2 * sb + 512 <= 160 * 1024 && rd <= maxv) { KI = ki; nb = cand; prow = pr; grow = gr; stageB = sb; rounds = rd; break; }
    }
  }
  if (!nb) return 0;
  WposArgs b; std::memset(&b, 0, sizeof(b));
  b.P = a.P; b.G = a.G; b.part = a.scratch;
  b.proP_scale = a.proP_scale; b.proP_shift = a.proP_shift; b.proP_relu = a.proP_relu;
  b.proG_scale = a.proG_scale; b.proG_shift = a.proG_shift; b.proG_relu = a.proG_relu;
  b.N = a.N; b.Ca = a.Ca; b.Cb = a.Cb; b.HS = a.Hp; b.HL = a.Hg; b.K = a.ksz; b.S = a.stride; b.Pd = a.pad;
  b.nb = nb; b.prow = prow; b.grow = grow;
  const long wsize = (long)a.Ca * a.Cb * a.ntaps;
  // image splits: ~256 blocks; at most 32 partial images (64 while they stay below 32 MB in all), each block at least two image chunks,
  // partial images within the scratch
  int nsplit = (256 + tiles - 1) / tiles;
  const int max_split = wsize * 4 * 64 <= (32L << 20) ? 64 : 32;
  if (nsplit > max_split) nsplit = max_split;
  const long cap = (long)(kWgradScratchBytes / 4) / wsize;
  if (cap < 1) return 0;
  if (nsplit > cap) nsplit = (int)cap;
  int per = (a.N + nsplit - 1) / nsplit;
  per = ((per + KI - 1) / KI) * KI;
  if (per < 2 * KI) per = 2 * KI;
  nsplit = (a.N + per - 1) / per;
  b.nsplit = nsplit; b.imgs_per_split = per;
  const size_t lds = 2 * stageB + 512;
  dim3 grid(nsplit, tiles), block(nthr);
  bool launched = false;
#define MMVAE_WPOS(KI_, NT_, MV_, NF_) \
  if (!launched && KI == KI_ && nthr == NT_ && nfb == NF_ && (MV_ == 8 || rounds <= 4)) { \
    static bool attr_set = false; \
    if (!attr_set) { \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pos_kernel<KI_, NT_, MV_, NF_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { \
        set_error("wgrad_pos: hipFuncSetAttribute"); return MMVAE_ERR_HIP; } \
      attr_set = true; } \
    hipLaunchKernelGGL((wgrad_pos_kernel<KI_, NT_, MV_, NF_>), grid, block, lds, s, b); launched = true; }
  // (vectors per thread and stage: 4 where the rounds allow it -- 16 registers fewer; the 16-wave blocks have 128 registers per lane)
  MMVAE_WPOS(32, 256, 8, 2) MMVAE_WPOS(32, 576, 4, 2) MMVAE_WPOS(32, 576, 8, 2) MMVAE_WPOS(32, 1024, 4, 2)
  MMVAE_WPOS(16, 256, 8, 2) MMVAE_WPOS(16, 576, 4, 2) MMVAE_WPOS(16, 576, 8, 2) MMVAE_WPOS(16, 1024, 4, 2)
  MMVAE_WPOS(16, 256, 8, 1) MMVAE_WPOS(16, 576, 4, 1) MMVAE_WPOS(16, 576, 8, 1) MMVAE_WPOS(16, 1024, 4, 1)
#undef MMVAE_WPOS
  if (!launched) return 0;
  const int rc = check_launch("wgrad_pos");
  if (rc) return rc;
  WgradReduceArgs u; std::memset(&u, 0, sizeof(u));
  u.part = a.scratch; u.dW = a.dW; u.Ca = a.Ca; u.Cb = a.Cb; u.ntaps = a.ntaps; u.nparts = nsplit;
  u.Ca_valid = a.Ca; u.Cb_valid = a.Cb; u.sA = a.sA; u.sB = a.sB; u.scale = a.scale;
  for (int t = 0; t < 25; ++t) u.tap_off[t] = a.tap_off[t];
  const int rr = launch_wgrad_reduce(u, s);
  return rr < 0 ? rr : 1;
}

}  // namespace mmvae
