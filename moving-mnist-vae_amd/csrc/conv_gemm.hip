// Implicit-GEMM convolution kernels on CDNA4 matrix cores (gfx950).
//
//  * gather_gemm_kernel : Conv2d forward, ConvTranspose2d forward and every data-gradient,
//    as ONE gather-form GEMM  D[cout][pixel] = W[cout][k] * X[k][pixel]  (k = tap x cin).
//    MFMA orientation is "weights x pixels" so each lane ends up with 4 consecutive output
//    channels of one pixel -> 8/16-byte NHWC stores.
//  * wgrad_kernel : weight gradients, D[a][b] = sum_pixels P[pixel][a] * G[pixel + tap][b].
//
// Both stage 128-byte K-rows (8 x 16-byte "kvecs") through LDS with an XOR swizzle and feed
// v_mfma_f32_16x16x32_bf16 (bf16 storage) or v_mfma_f32_16x16x4_f32 (exact f32 mode).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kernels.hpp"

namespace mmvae {

constexpr int KV = 8;  // kvecs (16 B) per LDS row

__device__ __forceinline__ int lds_slot(int row, int kv) { return row * KV + (kv ^ ((row >> 1) & 7)); }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ f32x4 run(const Vec16& a, const Vec16& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // lane group g = lane>>4 holds k = 4g..4g+3 of a 16-wide chunk in BOTH operands; the four
  // 16x16x4 MFMAs consume element j of every group, so the k-permutation cancels.
  __device__ static __forceinline__ f32x4 run(const Vec16& a, const Vec16& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w[j]), __uint_as_float(b.w[j]), c, 0, 0, 0);
    return c;
  }
};

template <typename TO> __device__ __forceinline__ void store4(TO* p, const float* v, bool acc);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v, bool acc) {
  float4 o = make_float4(v[0], v[1], v[2], v[3]);
  if (acc) { float4 e = *reinterpret_cast<const float4*>(p); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
  *reinterpret_cast<float4*>(p) = o;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v, bool acc) {
  float f[4] = {v[0], v[1], v[2], v[3]};
  if (acc) {
    uint2 e = *reinterpret_cast<const uint2*>(p);
    f[0] += __uint_as_float(e.x << 16); f[1] += __uint_as_float(e.x & 0xffff0000u);
    f[2] += __uint_as_float(e.y << 16); f[3] += __uint_as_float(e.y & 0xffff0000u);
  }
  uint2 o;
  o.x = pack2_bf16(f[0], f[1]);
  o.y = pack2_bf16(f[2], f[3]);
  *reinterpret_cast<uint2*>(p) = o;
}

// ============================================================================ gather GEMM
template <typename T, typename TO, int CT16>
__global__ __launch_bounds__(256) void gather_gemm_kernel(GatherArgs a) {
  constexpr int VE = Elem<T>::kVec;
  constexpr int PT = 128;
  constexpr int CT = CT16 * 16;
  __shared__ Vec16 sX[PT * KV];
  __shared__ Vec16 sW[CT * KV];
  __shared__ float sPro[2 * 512];
  __shared__ int sTap[2 * kMaxTaps];
  __shared__ float sStat[4 * 2 * CT];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, g = lane >> 4, r = lane & 15;
  const int cbase = blockIdx.y * CT;
  const Phase P = a.phases[blockIdx.z];
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wt = reinterpret_cast<const T*>(a.w) + P.w_off;
  TO* __restrict__ Y = reinterpret_cast<TO*>(a.y);
  const bool has_pro = a.pro_scale != nullptr;
  if (has_pro) {
    for (int i = t; i < a.Cin; i += 256) { sPro[i] = a.pro_scale[i]; sPro[512 + i] = a.pro_shift[i]; }
  }
  if (t < P.ntaps) { sTap[2 * t] = a.taps[P.tap0 + t].dh; sTap[2 * t + 1] = a.taps[P.tap0 + t].dw; }
  __syncthreads();

  const int kvecs = P.ntaps * a.cin_vecs;
  const int Ktot = kvecs * VE;
  const int nks = (kvecs + KV - 1) / KV;
  const int HqWq = P.Hq * P.Wq;
  const int M = a.N * HqWq;
  const int ntiles = (M + PT - 1) / PT;
  const int kv_l = t & 7;           // this thread's kvec column in every K-step
  const int row_l = t >> 3;         // rows row_l + 32*i

  float st1[CT16][4], st2[CT16][4];
#pragma unroll
  for (int c = 0; c < CT16; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) { st1[c][j] = 0.f; st2[c][j] = 0.f; }

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- per-thread pixel decode for the 4 rows this thread stages
    int pbase[4], ph0[4], pw0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = tile * PT + row_l + 32 * i;
      if (m < M) {
        const int n = m / HqWq, rem = m - n * HqWq;
        const int hq = rem / P.Wq, wq = rem - hq * P.Wq;
        pbase[i] = n * a.Hi * a.Wi;
        ph0[i] = hq * a.SI;
        pw0[i] = wq * a.SI;
      } else {
        pbase[i] = -1; ph0[i] = 0; pw0[i] = 0;
      }
    }
    f32x4 acc[CT16][2];
#pragma unroll
    for (int c = 0; c < CT16; ++c) { acc[c][0] = (f32x4){0, 0, 0, 0}; acc[c][1] = (f32x4){0, 0, 0, 0}; }

    for (int ks = 0; ks < nks; ++ks) {
      // ---- stage X (gathered, optional affine+relu prologue, zero padding afterwards)
      const int gk = ks * KV + kv_l;
      int dh = 0, dw = 0, c0 = 0;
      const bool kvalid = gk < kvecs;
      if (kvalid) {
        const int tap = gk / a.cin_vecs;
        c0 = (gk - tap * a.cin_vecs) * VE;
        dh = sTap[2 * tap]; dw = sTap[2 * tap + 1];
      }
      Vec16 xv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int hi = ph0[i] + dh, wi = pw0[i] + dw;
        const bool ok = kvalid && pbase[i] >= 0 && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
        if (ok) {
          xv[i] = *reinterpret_cast<const Vec16*>(X + ((long)(pbase[i] + hi * a.Wi + wi) * a.Cin + c0));
          if (has_pro) {
            float f[VE];
            Elem<T>::unpack(xv[i], f);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
              float v = f[j] * sPro[c0 + j] + sPro[512 + c0 + j];
              f[j] = a.pro_relu ? fmaxf(v, 0.f) : v;
            }
            xv[i] = Elem<T>::pack(f);
          }
        } else {
          xv[i] = Vec16{{0, 0, 0, 0}};
        }
      }
      // ---- stage W
      Vec16 wvv[(CT * KV + 255) / 256];
#pragma unroll
      for (int i = 0; i < (CT * KV + 255) / 256; ++i) {
        const int v = t + 256 * i;
        const int wr = v >> 3;
        const bool ok = (v < CT * KV) && kvalid && (cbase + wr) < a.Cout;
        wvv[i] = ok ? *reinterpret_cast<const Vec16*>(Wt + ((long)(cbase + wr) * Ktot + (long)gk * VE)) : Vec16{{0, 0, 0, 0}};
      }
      __syncthreads();   // previous K-step's fragment reads are done
#pragma unroll
      for (int i = 0; i < 4; ++i) sX[lds_slot(row_l + 32 * i, kv_l)] = xv[i];
#pragma unroll
      for (int i = 0; i < (CT * KV + 255) / 256; ++i) {
        const int v = t + 256 * i;
        if (v < CT * KV) sW[lds_slot(v >> 3, kv_l)] = wvv[i];
      }
      __syncthreads();
      // ---- MFMA
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int kvs = 4 * k2 + g;
        const Vec16 b0 = sX[lds_slot(32 * wv + r, kvs)];
        const Vec16 b1 = sX[lds_slot(32 * wv + 16 + r, kvs)];
#pragma unroll
        for (int c = 0; c < CT16; ++c) {
          const Vec16 af = sW[lds_slot(16 * c + r, kvs)];
          acc[c][0] = Mma<T>::run(af, b0, acc[c][0]);
          acc[c][1] = Mma<T>::run(af, b1, acc[c][1]);
        }
      }
    }
    // ---- epilogue: lane holds couts (16c + 4g + j), pixel (32wv + 16pt + r)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int m = tile * PT + 32 * wv + 16 * pt + r;
      if (m < M) {
        const int n = m / HqWq, rem = m - n * HqWq;
        const int hq = rem / P.Wq, wq = rem - hq * P.Wq;
        const long obase = ((long)(n * a.Ho + hq * a.SO + P.ph) * a.Wo + (wq * a.SO + P.pw)) * a.Cout;
#pragma unroll
        for (int c = 0; c < CT16; ++c) {
          const int co = cbase + 16 * c + 4 * g;
          if (co < a.Cout) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = acc[c][pt][j] + (a.bias ? a.bias[co + j] : 0.f);
              st1[c][j] += v[j];
              st2[c][j] += v[j] * v[j];
            }
            store4<TO>(Y + obase + co, v, a.accumulate != 0);
          }
        }
      }
    }
  }
  if (a.stats) {
    // reduce over the 16 pixel-lanes that share g, then over the 4 waves
#pragma unroll
    for (int c = 0; c < CT16; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          st1[c][j] += __shfl_xor(st1[c][j], o, 64);
          st2[c][j] += __shfl_xor(st2[c][j], o, 64);
        }
      }
    __syncthreads();
    if (r == 0) {
#pragma unroll
      for (int c = 0; c < CT16; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sStat[wv * 2 * CT + 16 * c + 4 * g + j] = st1[c][j];
          sStat[wv * 2 * CT + CT + 16 * c + 4 * g + j] = st2[c][j];
        }
    }
    __syncthreads();
    if (t < 2 * CT) {
      const float s = sStat[t] + sStat[2 * CT + t] + sStat[4 * CT + t] + sStat[6 * CT + t];
      const int which = t / CT, cl = t - which * CT;
      if (cbase + cl < a.Cout) a.stats[((long)blockIdx.z * gridDim.x + blockIdx.x) * 2 * a.Cout + (long)which * a.Cout + cbase + cl] = s;
    }
  }
}

template <typename T, typename TO>
static int launch_gather_t(GatherArgs& a, int gx, hipStream_t s) {
  int ct16 = a.Cout >= 128 ? 8 : (a.Cout > 32 ? 4 : (a.Cout > 16 ? 2 : 1));
  const int CT = ct16 * 16;
  dim3 grid(gx, (a.Cout + CT - 1) / CT, a.nphase), block(256);
  switch (ct16) {
    case 1: hipLaunchKernelGGL((gather_gemm_kernel<T, TO, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((gather_gemm_kernel<T, TO, 2>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((gather_gemm_kernel<T, TO, 4>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((gather_gemm_kernel<T, TO, 8>), grid, block, 0, s, a); break;
  }
  int rc = check_launch("gather_gemm");
  return rc ? rc : gx * a.nphase;
}

int conv_xcd_walk() {
  return 1;      // XCD-aware tile order (measured neutral to slightly positive; the switch is gone)
}

// (round 4, all sixteen channel-heavy layers through gather_gemm_kernel again for reference: 1.1-5.7x slower than the kernel each runs on
// today, forward and data gradient, except uplayer3.conv1's forward -- 19.6 vs 24.1 us; the switch stays a constant)
bool conv_force_v1() {
  constexpr int v = 0;
  return v != 0;
}

bool make_tile_geom(TileGeom& g, int N, int Hq, int Wq, int Hi, int Wi, int SI, int oh, int ow, int span_h, int span_w, int TP, int sub) {
  if (Hq <= 0 || Wq <= 0 || Wq > TP || N <= 0) return false;
  if (sub > 1 && (TP != 128 || (Hq * Wq >= 128 ? 128 % Wq != 0 : 128 % (Hq * Wq) != 0))) return false;   // sub-tiles must be whole rows / images
  g.N = N; g.Hq = Hq; g.Wq = Wq; g.Hi = Hi; g.Wi = Wi; g.SI = SI; g.oh = oh; g.ow = ow; g.TP = TP * sub; g.sub = sub;
  if (Hq * Wq >= TP) {
    g.segs = 1; g.sub_j = TP / Wq; g.sub_seg = 0;
    g.qr = g.sub_j * sub; if (g.qr > Hq) g.qr = Hq;
    g.tiles_per_img = (Hq + g.qr - 1) / g.qr;
    g.ntiles = N * g.tiles_per_img;
  } else {
    g.sub_seg = TP / (Hq * Wq); g.sub_j = 0;
    g.segs = g.sub_seg * sub; g.qr = Hq; g.tiles_per_img = 0;
    g.ntiles = (N + g.segs - 1) / g.segs;
  }
  g.PR = (g.qr - 1) * SI + span_h;
  g.PW = (Wq - 1) * SI + span_w;
  g.sub_pix = g.tiles_per_img > 0 ? g.sub_j * SI * g.PW : g.sub_seg * g.PR * g.PW;
  return g.PR < 256 && g.PW < 256 && g.segs <= 512;
}

constexpr size_t kV2MaxLds = 60 * 1024;

// Pipelined all-phases patch kernel: eligible when the weights of every phase together with one patch fit LDS.
// Returns >0 (stats rows) when it ran, 0 when not eligible, <0 on error.
static int try_patch(int dt, int out_dt, const GatherArgs& a, hipStream_t s) {
  constexpr int enabled = 1;
  if (!enabled || a.Cout > 64 || a.Cin > 256) return 0;
  if (a.x2 && (a.x_planar || a.y_planes || a.accumulate || out_dt != dt || a.Cin2 % (dt == DT_F32 ? 4 : 8) != 0 || a.Cin2 > 128 || a.Cout > 32)) return 0;
  if ((long)a.N * a.Ho * a.Wo * (a.y_planes ? a.y_planes : a.Cout) >= (1L << 30)) return 0;     // 32-bit output offsets
  if (!a.x_planar && (long)a.N * a.Hi * a.Wi * a.Cin * (long)dtype_size(dt) >= (1L << 31)) return 0;   // buffer-load range
  const int VE = dt == DT_F32 ? 4 : 8;
  if (a.x_planar && (a.Cin > 16 || a.x_planes > a.Cin || a.pro_scale)) return 0;
  if (a.y_planes && (a.Cout != 16 || a.y_planes > 16)) return 0;
  const int cin_vecs = a.Cin / VE;
  if (!a.x_planar && (256 % cin_vecs) != 0) return 0;
  PatchArgs b; memset(&b, 0, sizeof(b));
  b.x = a.x; b.w = a.w; b.y = a.y; b.pro_scale = a.pro_scale; b.pro_shift = a.pro_shift; b.pro_relu = a.pro_relu;
  b.bias = a.bias; b.stats = a.stats; b.accumulate = a.accumulate;
  b.Cin = a.Cin; b.Cout = a.Cout; b.Ho = a.Ho; b.Wo = a.Wo; b.SO = a.SO; b.nphase = a.nphase;
  b.x_planar = a.x_planar; b.x_planes = a.x_planes; b.y_planes = a.y_planes;
  int ct16 = (a.Cout + 15) / 16; if (ct16 == 3) ct16 = 4;
  const int CT = ct16 * 16;
  int dh0 = 0, dh1 = 0, dw0 = 0, dw1 = 0, Hq = 0, Wq = 0;
  bool first = true;
  for (int p = 0; p < a.nphase; ++p) {
    const Phase& ph = a.phases[p];
    if (ph.ntaps <= 0) return 0;
    for (int t = 0; t < ph.ntaps; ++t) {
      const Tap tp = a.taps[ph.tap0 + t];
      if (first) { dh0 = dh1 = tp.dh; dw0 = dw1 = tp.dw; first = false; }
      dh0 = tp.dh < dh0 ? tp.dh : dh0; dh1 = tp.dh > dh1 ? tp.dh : dh1;
      dw0 = tp.dw < dw0 ? tp.dw : dw0; dw1 = tp.dw > dw1 ? tp.dw : dw1;
    }
    Hq = ph.Hq > Hq ? ph.Hq : Hq; Wq = ph.Wq > Wq ? ph.Wq : Wq;
    PatchPhase& q = b.phases[p];
    q.ph = ph.ph; q.pw = ph.pw; q.Hq = ph.Hq; q.Wq = ph.Wq; q.ntaps = ph.ntaps; q.tap0 = ph.tap0; q.w_off = ph.w_off;
    const int kvp = (ph.ntaps * cin_vecs + 3) & ~3;
    q.w_vec0 = b.w_vecs; q.koff0 = b.koff_total;
    b.w_vecs += CT * (kvp + 1); b.koff_total += kvp;
  }
  if ((size_t)b.w_vecs * 16 > 40 * 1024) return 0;
  for (int t = 0; t < kMaxTaps; ++t) b.taps[t] = a.taps[t];
  b.x2_phase = -1;
  if (a.x2) {
    // the second source lives on the q grid: its phase must span the whole q grid
    for (int p = 0; p < a.nphase; ++p)
      if (a.phases[p].ph == a.x2_ph && a.phases[p].pw == a.x2_pw && a.phases[p].Hq == Hq && a.phases[p].Wq == Wq) b.x2_phase = p;
    if (b.x2_phase < 0) return 0;
    if ((long)a.N * Hq * Wq * a.Cin2 * (long)dtype_size(dt) >= (1L << 31)) return 0;
    b.x2 = a.x2; b.w2 = a.w2; b.Cin2 = a.Cin2;
    b.kvp2 = (a.Cin2 / VE + 3) & ~3;
    b.x2_bytes = (unsigned)((long)a.N * Hq * Wq * a.Cin2 * (long)dtype_size(dt));
  }
  // tile size: 128 q-pixels, or 256 / 512 (whole rows, power-of-two width >= 16) for the 16- and 32-channel layers:
  // bigger tiles amortise barriers, tile decode and the patch halo
  constexpr int sub_env = 0, uni_env = 1;
  int max_sub = ct16 == 1 ? (a.x_planar ? 4 : 2) : 1;       // measured per layer class (tools/sweep_env.sh MMVAE_PATCH_SUB)
  if (sub_env > 0) { const int inst = ct16 == 1 ? 4 : (ct16 == 2 ? 2 : 1); max_sub = sub_env < inst ? sub_env : inst; }
  // uniform geometry (whole rows, power-of-two width >= 16, every (ph, pw) phase present and full-size): LDS epilogue
  const bool pow2w = Wq >= 16 && (Wq & (Wq - 1)) == 0;
  bool uni_ok = uni_env && pow2w && Hq * Wq >= 128 && a.nphase == a.SO * a.SO && a.SO <= 2 && (!a.y_planes || a.SO == 1);
  for (int i = 0; i < 4; ++i) b.phase_of[i] = -1;
  for (int p = 0; p < a.nphase && uni_ok; ++p) {
    const Phase& ph = a.phases[p];
    if (ph.Hq != Hq || ph.Wq != Wq || ph.ph < 0 || ph.ph >= a.SO || ph.pw < 0 || ph.pw >= a.SO || b.phase_of[ph.ph * a.SO + ph.pw] >= 0) uni_ok = false;
    else b.phase_of[ph.ph * a.SO + ph.pw] = p;
  }
  if (uni_ok && (Hq * a.SO != a.Ho || Wq * a.SO != a.Wo || (!a.y_planes && a.Cout % 16 != 0))) uni_ok = false;
  const size_t out_es = out_dt == DT_F32 ? 4 : dtype_size(dt);
  size_t lds = 0;
  int slots = 0;
  bool ok = false;
  TileGeom g1;
  const bool have1 = make_tile_geom(g1, a.N, Hq, Wq, a.Hi, a.Wi, a.SI, dh0, dw0, dh1 - dh0 + 1, dw1 - dw0 + 1, 128, 1);
  if (!have1) return 0;
  for (int sub = uni_ok ? max_sub : 1; sub >= 1 && !ok; sub >>= 1) {
    if (sub > 1 && Hq * Wq < 128 * sub) continue;
    if (!make_tile_geom(b.g, a.N, Hq, Wq, a.Hi, a.Wi, a.SI, dh0, dw0, dh1 - dh0 + 1, dw1 - dw0 + 1, 128, sub)) continue;
    if (sub > 1 && (b.g.tiles_per_img == 0 || b.g.qr * Wq != 128 * sub)) continue;
    if (sub > 1 && b.g.ntiles < 1024 && g1.ntiles >= 1024) continue;       // keep >= 1024 tiles when the problem has them
    b.npt = 2 * sub;
    b.uni = (uni_ok && b.g.tiles_per_img > 0 && b.g.qr * Wq == 128 * sub) ? 1 : 0;
    if (sub > 1 && !b.uni) continue;
    // staging buffer of one wave: NCHW planes, or channel-quad planes [npt][SO][ct16][4] of (16 q x 4 channels + 16 B skew)
    b.out_wave_bytes = !b.uni ? 0 : (int)(a.y_planes ? (size_t)a.y_planes * b.npt * 16 * 4 : (size_t)b.npt * a.SO * ct16 * 4 * (16 * 4 * out_es + 16));
    b.out_wave_bytes = (b.out_wave_bytes + 15) & ~15;
    if (a.x2) {
      if (b.npt != 2) continue;                                                 // second source: 128-pixel tiles only
      b.x2_slots = (b.g.segs * b.g.qr * Wq * (a.Cin2 / VE) + 255) / 256;
      if (b.x2_slots > 4) continue;
    }
    lds = patch_conv_lds_bytes(b, dt);
    slots = patch_conv_slots(b, dt);
    ok = lds <= (sub > 1 ? 48 * 1024 : kV2MaxLds) && slots <= 12;
    if (!ok && b.uni && sub == 1) {          // the staging buffers did not fit: general epilogue
      b.uni = 0; b.out_wave_bytes = 0;
      lds = patch_conv_lds_bytes(b, dt);
      ok = lds <= kV2MaxLds && slots <= 12;
    }
    if (ok) { b.wq_shift = 0; while ((1 << b.wq_shift) < Wq) ++b.wq_shift; }
  }
  if (!ok) return 0;
  int occ = (int)((160 * 1024) / lds);
  const int occ_regs = slots <= 4 ? (ct16 == 1 ? 4 : (ct16 == 2 ? 3 : 2)) : ((slots <= 8 && ct16 == 1 && !a.x_planar) ? 3 : 2);
  if (occ > occ_regs) occ = occ_regs;
  if (occ < 1) occ = 1;
  int gx = 256 * occ;
  if (b.npt >= 4 && slots <= 4 && occ > occ_regs - 1 && occ > 1) occ = occ_regs - 1;
  gx = 256 * occ;
  if (gx > b.g.ntiles) gx = b.g.ntiles;
  if (gx > kGatherMaxGridX) gx = kGatherMaxGridX;
  if (gx >= 8) gx &= ~7;
  b.xcd_walk = conv_xcd_walk();
  b.dbg = 0;
  b.x_bytes = a.x_planar ? 0u : (unsigned)((long)a.N * a.Hi * a.Wi * a.Cin * (long)dtype_size(dt));
  if (a.x2) {
    if (ct16 > 2) return 0;
    patch_conv_x2_carve(b, dt);
    if (occ > 2 && ct16 == 1 && slots <= 4) { occ = occ_regs - 1 > 1 ? occ_regs - 1 : 1; gx = 256 * occ; if (gx > b.g.ntiles) gx = b.g.ntiles; if (gx > kGatherMaxGridX) gx = kGatherMaxGridX; if (gx >= 8) gx &= ~7; }
  }
  return launch_patch_conv(dt, out_dt, b, gx, s);
}

// deep2_conv_kernel (conv_deep2.inc): weights from L2 straight into MFMA fragments, no barrier in the K loop.  MMVAE_DEEP2=0: off
static bool deep2_enabled() {
  constexpr int enabled = 1;
  return enabled != 0 && !conv_force_v1();
}
// LDS bytes of a tile of npt*16 q-pixels (whole images)
static size_t deep2_lds_for(int dt, int Cin, int Cout, int hw, int HiWi, int ntaps_all, int npt, int fp8 = 0) {
  DeepArgs b; memset(&b, 0, sizeof(b));
  b.fp8 = fp8;
  b.Cin = Cin; b.Hi = HiWi; b.Wi = 1; b.ipt = npt * 16 / hw; b.npt = npt; b.ntaps_all = ntaps_all; b.nw = Cout / 32 < 8 ? Cout / 32 : 8;
  return deep2_conv_lds_bytes(b, dt);
}
bool deep2_shape_ok(int dt, int Cin, int Cout, int Hq, int Wq, int Hi, int Wi, int ntaps_all, int fp8) {
  if (!deep2_enabled()) return false;
  if (fp8 && dt != DT_BF16) return false;
  const int ES = fp8 ? 1 : dt == DT_F32 ? 4 : 2;
  const int cpt = Cin * ES / 64;
  const int c0 = ES == 4 ? 2 : 1;                   // 64-byte chunks of a tap at Cin = 32 (fp8: at Cin = 64)
  if (Cin < (fp8 ? 64 : 32) || (Cin * ES) % 64 != 0 || (cpt != c0 && cpt != 2 * c0 && cpt != 4 * c0 && cpt != 8 * c0)) return false;   // Cin 32, 64, 128, 256
  if (Cout < 32 || Cout % 32 != 0 || (Cout > 256 && Cout % 256 != 0)) return false;
  if (Cin < 64 && Cout < 64) return false;          // the 16/32-channel layers are the patch-tile kernel's (weights resident in LDS)
  // (32 -> 32 with 16 taps -- decoder.uplayer3.conv2, whose weights + patch do not fit the patch-tile kernel's LDS -- measured here in round 4:
  // forward 84 us against patch_conv's 90, data gradient 84 us against the first-generation gather_gemm's 45: left where it was)
  const int hw = Hq * Wq;
  if (hw < 1 || hw > 128 || Hq > 255 || Wq > 255 || ntaps_all < 1 || ntaps_all > kMaxTaps) return false;
  const int npt_min = hw > 64 ? 8 : hw > 32 ? 4 : 2;
  if (fp8 && cpt > 4) return false;
  return deep2_lds_for(dt, Cin, Cout, hw, Hi * Wi, ntaps_all, npt_min, fp8) <= 150 * 1024;
}
static int try_deep2(int dt, int out_dt, const GatherArgs& a, hipStream_t s) {
  constexpr int npt_env = 0;
  if (a.x_planar || a.y_planes || a.x2) return 0;
  if (a.fp8 && (!a.wfrag || a.accumulate || dt != DT_BF16 || out_dt != DT_BF16)) return 0;
  int Hq = 0, Wq = 0, ntaps_all = 0;
  for (int p = 0; p < a.nphase; ++p) {
    const Phase& ph = a.phases[p];
    if (ph.ntaps <= 0) return 0;
    Hq = ph.Hq > Hq ? ph.Hq : Hq; Wq = ph.Wq > Wq ? ph.Wq : Wq;
    ntaps_all = ph.tap0 + ph.ntaps > ntaps_all ? ph.tap0 + ph.ntaps : ntaps_all;
  }
  if (!deep2_shape_ok(dt, a.Cin, a.Cout, Hq, Wq, a.Hi, a.Wi, ntaps_all, a.fp8)) return 0;
  if ((long)a.N * a.Ho * a.Wo * a.Cout >= (1L << 32)) return 0;
  const int ES = dt == DT_F32 ? 4 : 2, VE = 16 / ES;
  const int cpt = a.Cin * (a.fp8 ? 1 : ES) / 64;
  DeepArgs b; memset(&b, 0, sizeof(b));
  b.fp8 = a.fp8;
  b.x = a.x; b.w = a.w; b.y = a.y; b.pro_scale = a.pro_scale; b.pro_shift = a.pro_shift; b.pro_relu = a.pro_relu;
  b.bias = a.bias; b.stats = a.stats; b.accumulate = a.accumulate; b.wfrag = a.wfrag;
  b.N = a.N; b.Hi = a.Hi; b.Wi = a.Wi; b.Cin = a.Cin; b.Ho = a.Ho; b.Wo = a.Wo; b.Cout = a.Cout; b.SI = a.SI; b.SO = a.SO;
  b.nphase = a.nphase; b.Hq = Hq; b.Wq = Wq; b.ntaps_all = ntaps_all;
  for (int p = 0; p < a.nphase; ++p) {
    const Phase& ph = a.phases[p];
    b.phases[p] = DeepPhase{ph.ph, ph.pw, ph.Hq, ph.Wq, ph.ntaps, ph.tap0, ph.w_off};
  }
  for (int t = 0; t < kMaxTaps; ++t) b.taps[t] = a.taps[t];
  for (b.cpt_log2 = 0; (1 << b.cpt_log2) < cpt; ++b.cpt_log2) {}
  constexpr int nw_env = 0;
  b.nw = a.Cout / 32 < 8 ? a.Cout / 32 : 8;
  if (nw_env && nw_env <= b.nw && (a.Cout / 32) % nw_env == 0) b.nw = nw_env;
  if ((64 * b.nw) % (a.Cin / VE) != 0) return 0;
  const int gy = a.Cout / (32 * b.nw);
  const int hw = Hq * Wq;
  // the largest tile (fewest re-reads of the weight matrix) that fits 80 KB of LDS (two blocks per CU) and leaves no SIMD idle;
  // else the largest that fits at all
  int best = 0;
  for (int pass = 0; pass < 2 && !best; ++pass)
    for (int npt = 8; npt >= 2 && !best; npt >>= 1) {
      if (npt_env && npt != npt_env && pass == 0) continue;
      if (hw > npt * 16) break;
      const size_t lds = deep2_lds_for(dt, a.Cin, a.Cout, hw, a.Hi * a.Wi, ntaps_all, npt, a.fp8);
      int ipt = npt * 16 / hw; if (ipt > a.N) ipt = a.N;
      const long ntiles = (a.N + ipt - 1) / ipt;
      if (pass == 0) {
        if (lds > 80 * 1024) continue;
        // measured (tools/deep_sweep.sh): below one wave per SIMD a smaller tile wins, above it the larger tile's halved weight traffic does
        if (!npt_env && npt > 2 && ntiles * gy * b.nw < 256 * 4) continue;
      } else if (lds > 150 * 1024) continue;
      best = npt;
    }
  if (!best) return 0;
  b.npt = best; b.ipt = best * 16 / hw;
  if (b.ipt > a.N) b.ipt = a.N;
  b.ntiles = (a.N + b.ipt - 1) / b.ipt;
  int gx = b.ntiles < kGatherMaxGridX ? b.ntiles : kGatherMaxGridX;
#ifdef MMVAE_DEEP2_TS
  {
    static long long* tsbuf = nullptr; static int tscount = 0;
    if (!tsbuf && hipMalloc(&tsbuf, kGatherMaxGridX * 16 * 8) != hipSuccess) return MMVAE_ERR_HIP;
    (void)hipMemsetAsync(tsbuf, 0, kGatherMaxGridX * 16 * 8, s);
    b.ts = tsbuf;
    const int rc = launch_deep2_conv(dt, out_dt, b, gx, s);
    if (++tscount % 22 == 3) {
      (void)hipStreamSynchronize(s);
      static long long h[kGatherMaxGridX * 16];
      (void)hipMemcpy(h, tsbuf, sizeof(h), hipMemcpyDeviceToHost);
      fprintf(stderr, "DEEP2TS Cin=%d Cout=%d taps=%d Hi=%d ipt=%d npt=%d nw=%d ntiles=%d gx=%d lds=%zu frag=%d\n", b.Cin, b.Cout, b.ntaps_all, b.Hi, b.ipt, b.npt, b.nw, b.ntiles, gx, deep2_conv_lds_bytes(b, dt), b.wfrag);
      for (int blk : {0, 1, gx / 2, gx - 1}) {
        fprintf(stderr, "  blk %4d:", blk);
        for (int i = 1; i < 16 && h[blk * 16 + i]; ++i) fprintf(stderr, " %7lld", h[blk * 16 + i] - h[blk * 16 + i - 1]);
        fprintf(stderr, "\n");
      }
    }
    return rc;
  }
#endif
  return launch_deep2_conv(dt, out_dt, b, gx, s);
}

// pos_conv_kernel (conv_pos.inc): q-grids up to 4x4, MFMA columns = images, padded (position, tap) pairs not computed
static int try_pos(int dt, int out_dt, const GatherArgs& a, hipStream_t s) {
  if (conv_force_v1() || dt != DT_BF16 || out_dt != DT_BF16 || !a.wfrag || a.fp8 || a.gup == 0) return 0;
  if (a.x_planar || a.y_planes || a.x2 || a.bias || a.Hi != a.Wi || a.Ho != a.Wo || a.Cout % 32 != 0) return 0;
  const int up = a.gup == 2 ? 1 : 0;
  if (!pos_conv_takes(a.gk, a.gs, a.gp, up, a.Hi, a.Ho, a.Cin)) return 0;
  if (up && a.gs > 1 && a.gk < a.gs && !a.accumulate) return 0;      // stride phases without a tap would have to be zero-filled
  if (a.Hi >= 8 && a.Cout < 128) return 0;      // an 8x8 input tile is 131 KB of LDS (one block per CU): with two waves per block deep2 wins (measured)
  if ((long)a.N * a.Ho * a.Wo * a.Cout >= (1L << 32)) return 0;
  PosArgs b; memset(&b, 0, sizeof(b));
  b.x = a.x; b.w = a.w; b.y = a.y; b.pro_scale = a.pro_scale; b.pro_shift = a.pro_shift; b.pro_relu = a.pro_relu;
  b.stats = a.stats; b.accumulate = a.accumulate; b.N = a.N; b.Cout = a.Cout;
  b.nw = a.Cout / 32 < 8 ? a.Cout / 32 : 8;
  if ((a.Cout / 32) % b.nw != 0 || (64 * b.nw) % (a.Cin / 8) != 0) return 0;
  return launch_pos_conv(a.gk, a.gs, a.gp, up, a.Hi, a.Ho, a.Cin, b, s);
}

// the second source as its own accumulate launch (shapes the patch-tile kernel does not merge)
static int launch_x2_separately(int dt, int out_dt, const GatherArgs& a, hipStream_t s) {
  GatherArgs b; memset(&b, 0, sizeof(b));
  int Hq = 0, Wq = 0;
  for (int p = 0; p < a.nphase; ++p) if (a.phases[p].ph == a.x2_ph && a.phases[p].pw == a.x2_pw) { Hq = a.phases[p].Hq; Wq = a.phases[p].Wq; }
  if (Hq <= 0) { set_error("gather_gemm: the second source's phase (%d,%d) is not part of the launch", a.x2_ph, a.x2_pw); return MMVAE_ERR_ARG; }
  b.x = a.x2; b.w = a.w2; b.y = a.y; b.accumulate = 1; b.wfrag = a.wfrag2;
  b.N = a.N; b.Hi = Hq; b.Wi = Wq; b.Cin = a.Cin2; b.Ho = a.Ho; b.Wo = a.Wo; b.Cout = a.Cout; b.SI = 1; b.SO = a.SO;
  b.nphase = 1; b.phases[0] = Phase{a.x2_ph, a.x2_pw, Hq, Wq, 1, 0, 0}; b.taps[0] = Tap{0, 0};
  b.gk = 1; b.gs = a.SO; b.gp = 0; b.gup = 2;      // a 1x1 stride-SO transposed conv into phase (0, 0)
  return launch_gather_gemm(dt, out_dt, b, s);
}

int launch_gather_gemm(int dt, int out_dt, GatherArgs a, hipStream_t s) {
  const int VE = dt == DT_F32 ? 4 : 8;
  note_launch_bytes((double)a.N * ((double)a.Hi * a.Wi * (a.x_planar ? a.x_planes : a.Cin) * (a.x_planar == 1 ? 4 : dtype_size(dt)) +
                                   (double)a.Ho * a.Wo * (a.y_planes > 0 ? a.y_planes : a.Cout) * (a.y_planes > 0 ? 4 : dtype_size(out_dt)) +
                                   (a.x2 ? (double)a.phases[0].Hq * a.phases[0].Wq * a.Cin2 * dtype_size(dt) : 0.0)));
  if (a.fp8) {
    // fp8 forward convolutions exist in the deep-layer kernel only (the packed weights are e4m3 bytes: no other kernel can read them)
    if (dt != DT_BF16 || out_dt != DT_BF16 || a.x2 || a.accumulate) { set_error("gather_gemm: fp8 needs the bf16 forward path"); return MMVAE_ERR_UNSUPPORTED; }
    a.cin_vecs = a.Cin / VE;
    const int rc2 = try_deep2(dt, out_dt, a, s);
    if (rc2 != 0) return rc2;
    if (a.wfrag) { set_error("gather_gemm: fragment-major fp8 weights (Cin=%d Cout=%d) need the deep2 kernel", a.Cin, a.Cout); return MMVAE_ERR_UNSUPPORTED; }
    set_error("gather_gemm: fp8 layer Cin=%d Cout=%d is not eligible for the deep-layer kernel", a.Cin, a.Cout); return MMVAE_ERR_UNSUPPORTED;
  }
  if (a.x2) {
    // fragment-major weights are read by deep2_conv_kernel only, which takes one source: two launches then
    constexpr bool merge = true;
    int rc = (merge && !conv_force_v1() && !a.wfrag && !a.wfrag2) ? try_patch(dt, out_dt, a, s) : 0;
    if (rc != 0) return rc;
    GatherArgs m = a; m.x2 = nullptr; m.w2 = nullptr; m.Cin2 = 0; m.wfrag2 = 0;
    rc = launch_gather_gemm(dt, out_dt, m, s);
    if (rc < 0) return rc;
    const int rc2 = launch_x2_separately(dt, out_dt, a, s);
    return rc2 < 0 ? rc2 : rc;
  }
  if (a.Cin % VE != 0 || a.Cout % 4 != 0 || a.nphase < 1 || a.nphase > kMaxPhases) {
    set_error("gather_gemm: unsupported Cin=%d Cout=%d nphase=%d", a.Cin, a.Cout, a.nphase);
    return MMVAE_ERR_UNSUPPORTED;
  }
  if (a.pro_scale && a.Cin > 512) { set_error("gather_gemm: prologue needs Cin<=512"); return MMVAE_ERR_UNSUPPORTED; }
  a.cin_vecs = a.Cin / VE;
  int max_tiles = 0, ntap = 0;
  for (int p = 0; p < a.nphase; ++p) {
    const long M = (long)a.N * a.phases[p].Hq * a.phases[p].Wq;
    if (M > 0x7fffffffL / 2) { set_error("gather_gemm: too many pixels"); return MMVAE_ERR_UNSUPPORTED; }
    const int tiles = (int)((M + 127) / 128);
    if (tiles > max_tiles) max_tiles = tiles;
    ntap += a.phases[p].ntaps;
  }
  if (ntap > kMaxTaps) { set_error("gather_gemm: %d taps > %d", ntap, kMaxTaps); return MMVAE_ERR_UNSUPPORTED; }
  if (max_tiles <= 0) return 1;
  if (a.wfrag && conv_force_v1()) { set_error("gather_gemm: fragment-major weights with MMVAE_CONV_V1"); return MMVAE_ERR_UNSUPPORTED; }
  if (!conv_force_v1()) {
    const int rcp = a.wfrag ? 0 : try_patch(dt, out_dt, a, s);
    if (rcp != 0) return rcp;
    const int rcq = try_pos(dt, out_dt, a, s);
    if (rcq != 0) return rcq;
    const int rcd2 = try_deep2(dt, out_dt, a, s);
    if (rcd2 != 0) return rcd2;
    if (a.wfrag) { set_error("gather_gemm: fragment-major weights (Cin=%d Cout=%d) need the deep2 kernel, which does not take this launch", a.Cin, a.Cout); return MMVAE_ERR_UNSUPPORTED; }
  }
  {
    // thin layers: barrier-free streaming kernel (weights in LDS, pixels straight from global memory)
    constexpr int g3_maxk = 160;
    constexpr int g3_dbg = 0;
    a.dbg = g3_dbg;
    int maxk = 0;
    for (int p = 0; p < a.nphase; ++p) if (a.phases[p].ntaps * a.Cin > maxk) maxk = a.phases[p].ntaps * a.Cin;
    const bool fits32 = (long)a.N * a.Hi * a.Wi * a.Cin < (1L << 31) && (long)a.N * a.Ho * a.Wo * a.Cout < (1L << 31);   // 32-bit element offsets
    if (!conv_force_v1() && !a.x_planar && !a.y_planes && !a.bias && out_dt == dt && a.Cout <= 64 && maxk <= g3_maxk && fits32) {
      int ct16 = (a.Cout + 15) / 16; if (ct16 == 3) ct16 = 4;
      if (gather3_lds_bytes(a, dt, ct16 * 16) <= kV2MaxLds) {
        long wt = 0;
        for (int p = 0; p < a.nphase; ++p) { const long m = (long)a.N * a.phases[p].Hq * a.phases[p].Wq; if ((m + 31) / 32 > wt) wt = (m + 31) / 32; }
        long gx3 = (wt + 3) / 4; if (gx3 > 2048 / a.nphase) gx3 = 2048 / a.nphase; if (gx3 < 1) gx3 = 1;
        if (gx3 > kGatherMaxGridX) gx3 = kGatherMaxGridX;
        return launch_gather3(dt, out_dt, a, (int)gx3, s);
      }
    }
    if (a.x_planar || a.y_planes) { set_error("gather_gemm: planar boundary layouts need the patch-tile kernel"); return MMVAE_ERR_UNSUPPORTED; }
  }
  const int gx = max_tiles < kGatherMaxGridX ? max_tiles : kGatherMaxGridX;
  if (dt == DT_F32) return launch_gather_t<float, float>(a, gx, s);
  if (out_dt == DT_F32) return launch_gather_t<bf16_t, float>(a, gx, s);
  return launch_gather_t<bf16_t, bf16_t>(a, gx, s);
}

// ============================================================================ wgrad
// LDS rows are CHANNELS, kvecs are groups of VE consecutive pixels (transposed staging).
template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  constexpr int VE = Elem<T>::kVec;
  constexpr int PK = KV * VE;       // pixels per K-step (64 bf16 / 32 f32)
  __shared__ Vec16 sP[64 * KV];     // [TA<=64][8 kvecs]
  __shared__ Vec16 sG[256 * KV];    // [TG*TB<=256][8 kvecs]
  __shared__ int sPix[PK * 3];
  __shared__ float sProP[2 * 256];
  __shared__ float sProG[2 * 256];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, g = lane >> 4, r = lane & 15;
  const int TA = a.TA16 * 16, TB = a.TB16 * 16, TG = a.TG;
  const int nb_tiles = (a.Cb + TB - 1) / TB;
  const int a0 = (blockIdx.y / nb_tiles) * TA, b0 = (blockIdx.y % nb_tiles) * TB;
  const int tap0 = blockIdx.z * TG;
  const int tg_n = min(TG, a.ntaps - tap0);
  const T* __restrict__ P = reinterpret_cast<const T*>(a.P);
  const T* __restrict__ G = reinterpret_cast<const T*>(a.G);
  T* sPe = reinterpret_cast<T*>(sP);
  T* sGe = reinterpret_cast<T*>(sG);
  const bool proP = a.proP_scale != nullptr, proG = a.proG_scale != nullptr;
  if (proP) for (int i = t; i < TA; i += 256) {
    const bool ok = a0 + i < a.Ca;
    sProP[i] = ok ? a.proP_scale[a0 + i] : 0.f; sProP[256 + i] = ok ? a.proP_shift[a0 + i] : 0.f;
  }
  if (proG) for (int i = t; i < TB; i += 256) {
    const bool ok = b0 + i < a.Cb;
    sProG[i] = ok ? a.proG_scale[b0 + i] : 0.f; sProG[256 + i] = ok ? a.proG_shift[b0 + i] : 0.f;
  }

  const int units = a.TA16 * a.TB16 * tg_n;
  f32x4 acc[16];
  int rowA[16], rowB[16];   // LDS fragment rows of unit u = wv + 4*i -> (tg, tb, ta)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc[i] = (f32x4){0, 0, 0, 0};
    const int u = wv + 4 * i;
    const int tg = u % tg_n, q = u / tg_n;
    const int tb = q % a.TB16, ta = q / a.TB16;
    rowA[i] = 16 * ta + r;
    rowB[i] = tg * TB + 16 * tb + r;
  }

  const long m_begin = (long)blockIdx.x * a.pix_per_block;
  const long m_end = min((long)a.M, m_begin + a.pix_per_block);
  const int HpWp = a.Hp * a.Wp;
  const int pvecs = TA / VE, gvecs = TB / VE;

  for (long m0 = m_begin; m0 < m_end; m0 += PK) {
    __syncthreads();   // previous step's fragment reads done; sPix reusable
    if (t < PK) {
      const long m = m0 + t;
      if (m < m_end) {
        const int n = (int)(m / HpWp), rem = (int)(m - (long)n * HpWp);
        const int hp = rem / a.Wp, wp = rem - hp * a.Wp;
        sPix[3 * t] = n * a.Hg * a.Wg;
        sPix[3 * t + 1] = hp * a.stride - a.pad;
        sPix[3 * t + 2] = wp * a.stride - a.pad;
      } else {
        sPix[3 * t] = -1; sPix[3 * t + 1] = 0; sPix[3 * t + 2] = 0;
      }
    }
    __syncthreads();
    // ---- stage P^T : [TA rows][PK pixels]
    for (int v = t; v < PK * pvecs; v += 256) {
      const int pk = v / pvecs, cv = v - pk * pvecs;
      const int ch = cv * VE;
      float f[VE];
      const bool ok = (m0 + pk) < m_end && (a0 + ch) < a.Ca;
      if (ok) {
        const Vec16 q = *reinterpret_cast<const Vec16*>(P + ((m0 + pk) * a.Ca + a0 + ch));
        Elem<T>::unpack(q, f);
        if (proP) {
#pragma unroll
          for (int j = 0; j < VE; ++j) {
            float x = f[j] * sProP[ch + j] + sProP[256 + ch + j];
            f[j] = a.proP_relu ? fmaxf(x, 0.f) : x;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) f[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const int row = ch + j;
        Elem<T>::store(sPe + (long)lds_slot(row, pk / VE) * VE + (pk % VE), f[j]);
      }
    }
    // ---- stage G_tap^T : [tg][TB rows][PK pixels]
    for (int v = t; v < tg_n * PK * gvecs; v += 256) {
      const int tg = v / (PK * gvecs), rem = v - tg * (PK * gvecs);
      const int pk = rem / gvecs, cv = rem - pk * gvecs;
      const int ch = cv * VE;
      const int tap = tap0 + tg, kh = tap / a.ksz, kw = tap - kh * a.ksz;
      const int base = sPix[3 * pk], hg = sPix[3 * pk + 1] + kh, wg = sPix[3 * pk + 2] + kw;
      const bool ok = base >= 0 && hg >= 0 && hg < a.Hg && wg >= 0 && wg < a.Wg && (b0 + ch) < a.Cb;
      float f[VE];
      if (ok) {
        const Vec16 q = *reinterpret_cast<const Vec16*>(G + ((long)(base + hg * a.Wg + wg) * a.Cb + b0 + ch));
        Elem<T>::unpack(q, f);
        if (proG) {
#pragma unroll
          for (int j = 0; j < VE; ++j) {
            float x = f[j] * sProG[ch + j] + sProG[256 + ch + j];
            f[j] = a.proG_relu ? fmaxf(x, 0.f) : x;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < VE; ++j) f[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < VE; ++j) {
        const int row = tg * TB + ch + j;
        Elem<T>::store(sGe + (long)lds_slot(row, pk / VE) * VE + (pk % VE), f[j]);
      }
    }
    __syncthreads();
    // ---- MFMA: unit u = wv + 4*i -> (tg, tb, ta)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int u = wv + 4 * i;
      if (u < units) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int kvs = 4 * k2 + g;
          const Vec16 af = sP[lds_slot(rowA[i], kvs)];
          const Vec16 bf = sG[lds_slot(rowB[i], kvs)];
          acc[i] = Mma<T>::run(af, bf, acc[i]);
        }
      }
    }
  }
  // ---- epilogue: lane holds a = 16ta + 4g + j, b = 16tb + r
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int u = wv + 4 * i;
    if (u < units) {
      const int tg = u % tg_n, q = u / tg_n;
      const int tb = q % a.TB16, ta = q / a.TB16;
      const int bb = b0 + 16 * tb + r;
      if (bb < a.Cb_valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int aa = a0 + 16 * ta + 4 * g + j;
          // partial image [pixel chunk = blockIdx.x][tap][a][b]: one owner per element, plain stores (summed by wgrad_reduce_kernel)
          if (aa < a.Ca) a.scratch[(((long)blockIdx.x * a.ntaps + tap0 + tg) * a.Ca + aa) * a.Cb + bb] = acc[i][j];
        }
      }
    }
  }
}

// Returns 1 when the v2 kernel ran, 0 when not eligible, <0 on error.
static int try_wgrad2(int dt, const WgradArgs& a, hipStream_t s) {
  if ((conv_force_v1() && !a.P_planar && !a.G_planar) || a.Ca % 16 || a.Cb % 16 || a.ntaps > 25) return 0;
  if (a.proP_scale && a.proG_scale) return 0;          // one prologue register set in the kernel
  if (a.P_planar && (a.Ca != 16 || a.P_planes > 16 || a.proP_scale)) return 0;
  if (a.G_planar && (a.Cb != 16 || a.Ca != 32 || a.proG_scale || a.ntaps < 4)) return 0;
  auto pick = [](int c) { return c >= 64 ? 64 : c; };
  int TA = pick(a.Ca), TB = pick(a.Cb);
  if (!(TA == 16 || TA == 32 || TA == 64) || !(TB == 16 || TB == 32 || TB == 64)) return 0;
  if (a.Ca % TA || a.Cb % TB) return 0;
  // a 1x1 stride-2 layer on 4x4 maps (encoder.layer3's shortcut): the 128-pixel tile's G patch is 8 images x 7x7 pixels, 50 KB at 64
  // channels -- it fits with a 32-channel G tile (the k-split mode cannot shrink the pixel tile instead)
  if (a.ntaps < 4 && TB == 64 && !a.P_planar && !a.G_planar) {
    Wgrad2Args probe; memset(&probe, 0, sizeof(probe));
    probe.TG = wgrad2_taps_per_block(TA / 16, 4, a.ntaps); probe.nw = 4;
    if (make_tile_geom(probe.g, a.N, a.Hp, a.Wp, a.Hg, a.Wg, a.stride, -a.pad, -a.pad, a.ksz, a.ksz, 128, 1) &&
        (wgrad2_lds_bytes(probe, dt, TA, 64) > kV2MaxLds || wgrad2_patch_slots(probe, dt, 64) > 16)) TB = 32;
  }
  const int ta16 = TA / 16, tb16 = TB / 16;
  Wgrad2Args b; memset(&b, 0, sizeof(b));
  b.P = a.P; b.G = a.G; b.dW = a.dW;
  b.proP_scale = a.proP_scale; b.proP_shift = a.proP_shift; b.proP_relu = a.proP_relu;
  b.proG_scale = a.proG_scale; b.proG_shift = a.proG_shift; b.proG_relu = a.proG_relu;
  b.TG = wgrad2_taps_per_block(ta16, tb16, a.ntaps);
  b.nw = 4;
  // deep layers (>= 32x64 channel tiles, 3x3 / 4x4 kernels): one tap per wave, every tap in one block -> the tile is
  // staged once instead of once per tap group
  constexpr int nw_env = 1;
  // (only where the 4-wave block needs several tap groups, and not for 64x64 tiles with 16 waves: 128 VGPRs spill)
  if (nw_env && !a.P_planar && !a.G_planar && ta16 >= 2 && tb16 >= 2 && (a.ntaps == 9 || a.ntaps == 16) && b.TG < a.ntaps &&
      !(a.ntaps == 16 && ta16 * tb16 >= 16)) {
    b.nw = a.ntaps; b.TG = a.ntaps;
  } else if (nw_env && !a.P_planar && !a.G_planar && a.ntaps == 16 && ta16 == 4 && tb16 == 4) {
    b.nw = 8; b.TG = 16;               // 64x64 tiles, 16 taps: 8 waves x 2 taps (16 waves would spill at 128 VGPRs)
  }
  bool fits = false;
  size_t lds = 0;
  for (int TP = 128; TP >= 32 && !fits; TP >>= 1) {     // shrink the pixel tile until patch + P tile fit in LDS
    if (TP < 128 && a.ntaps < 4) break;                 // the k-split mode (1x1 convs) needs all four k-steps
    if (!make_tile_geom(b.g, a.N, a.Hp, a.Wp, a.Hg, a.Wg, a.stride, -a.pad, -a.pad, a.ksz, a.ksz, TP, 1)) continue;
    lds = wgrad2_lds_bytes(b, dt, TA, TB);
    fits = lds <= kV2MaxLds && wgrad2_patch_slots(b, dt, TB) <= (b.nw == 4 ? 16 : 8);
  }
  if (!fits) return 0;
  // ("big" 256 / 512-pixel tiles for the 16x16 channel tile were measured in round 2 without a gain; the kernel keeps the template
  // parameter for the planar-P tail weight gradient, which uses it)
  b.Ca = a.Ca; b.Cb = a.Cb; b.Cb_valid = a.Cb_valid; b.Ca_valid = a.Ca_valid; b.ksz = a.ksz; b.ntaps = a.ntaps;
  b.sA = a.sA; b.sB = a.sB; b.scale = a.scale; b.P_planar = a.P_planar; b.P_planes = a.P_planes; b.G_planar = a.G_planar;
  for (int t = 0; t < 25; ++t) b.tap_off[t] = a.tap_off[t];
  const int tiles_ab = (a.Ca / TA) * (a.Cb / TB);
  const int zg = (a.ntaps + b.TG - 1) / b.TG;
  const long wsize = (long)a.Ca * a.Cb * a.ntaps;
  // grid: ~176 persistent blocks in all, each flushing one partial image tile.  Measured in round 4 (bench.py, ms per step): one block per
  // LDS slot (up to 1536 blocks) 6.59, 256 blocks 6.47 / 6.40, 192 6.35 / 6.41, 160 6.39, 128 6.36, 96 6.55, 64 6.96 -- these launches run on
  // the side stream beside the data-gradient chain: fewer blocks leave CUs to the critical path and write fewer partial images (the reduce
  // that follows moves as many bytes as the blocks wrote)
  // (the channel-heavy layers only -- operands below 200 MB; the big thin layers that still come here -- the deeper variant's identity
  // blocks at 32x32 / 64x64, the PixelCNN's 7x7 convs -- stream 0.3 .. 1.3 GB and keep one block per LDS slot)
  const long op_bytes = ((long)a.N * a.Hp * a.Wp * a.Ca + (long)a.N * a.Hg * a.Wg * a.Cb) * (long)dtype_size(dt);
  int occ = (int)((160 * 1024) / lds); if (occ > 6) occ = 6; if (occ < 1) occ = 1;
  long gx = (op_bytes < (200L << 20) ? 176L : 256L * occ) / ((long)tiles_ab * zg); if (gx < 1) gx = 1;
  if (!a.scratch) { set_error("wgrad: the partial-image scratch is required (no atomic flush path)"); return MMVAE_ERR_ARG; }
  const bool partial = true;
  if (partial) {
    // keep the partial images (written once, read once) below the bytes of the operands themselves
    const long in_bytes = ((long)a.N * a.Hp * a.Wp * a.Ca + (long)a.N * a.Hg * a.Wg * a.Cb) * (long)dtype_size(dt);
    long budget = in_bytes > (8L << 20) ? in_bytes : (8L << 20);
    if (budget > (long)kWgradScratchBytes) budget = (long)kWgradScratchBytes;
    const long cap = budget / (wsize * 4);
    if (cap < 1) return 0;
    if (gx > cap) gx = cap;
  } else {                                              // direct flush: keep the scattered atomics of a launch below ~3M
    long cap = (3L << 20) / (wsize > 0 ? wsize : 1);
    if (cap < 2) cap = 2;
    if (gx > cap) gx = cap;
  }
  if (gx > b.g.ntiles) gx = b.g.ntiles;
  if (gx >= 8) gx &= ~7L;
  b.xcd_walk = conv_xcd_walk();
  b.dbg = 0;
  if (partial) { b.dW = a.scratch; b.partial = 1; }
  const int rc = launch_wgrad2(dt, b, (int)gx, tiles_ab, zg, ta16, tb16, s);
  if (rc < 0) return rc;
  if (partial) {
    WgradReduceArgs u; memset(&u, 0, sizeof(u));
    u.part = a.scratch; u.dW = a.dW; u.Ca = a.Ca; u.Cb = a.Cb; u.ntaps = a.ntaps; u.nparts = (int)gx;
    u.Ca_valid = a.Ca_valid; u.Cb_valid = a.Cb_valid; u.sA = a.sA; u.sB = a.sB; u.scale = a.scale;
    for (int t = 0; t < 25; ++t) u.tap_off[t] = a.tap_off[t];
    const int rc2 = launch_wgrad_reduce(u, s);
    if (rc2 < 0) return rc2;
  }
  return 1;
}

#define MM_CHECK_RC(expr) do { const int rc__ = (expr); if (rc__ < 0) return rc__; } while (0)
int launch_wgrad(int dt, WgradArgs a, hipStream_t s) {
  const int VE = dt == DT_F32 ? 4 : 8;
  if (a.Ca % VE != 0 || a.Cb % VE != 0 || a.ntaps > 25 || a.ntaps < 1) {
    set_error("wgrad: unsupported Ca=%d Cb=%d ntaps=%d", a.Ca, a.Cb, a.ntaps);
    return MMVAE_ERR_UNSUPPORTED;
  }
  a.M = a.N * a.Hp * a.Wp;
  note_launch_bytes((double)a.N * ((double)a.Hp * a.Wp * (a.P_planar ? a.P_planes * 4.0 : a.Ca * (double)dtype_size(dt)) +
                                   (double)a.Hg * a.Wg * (a.G_planar ? 1 : a.Cb) * (double)dtype_size(dt)));
  if (a.Cb_valid <= 0 || a.Cb_valid > a.Cb) a.Cb_valid = a.Cb;
  if (a.Ca_valid <= 0 || a.Ca_valid > a.Ca) a.Ca_valid = a.Ca;
  if (a.M <= 0) return MMVAE_OK;
  {
    const int rcs = try_wgrad_stream(dt, a, s);
    if (rcs != 0) return rcs < 0 ? rcs : MMVAE_OK;
    const int rcp = try_wgrad_pos(dt, a, s);
    if (rcp != 0) return rcp < 0 ? rcp : MMVAE_OK;
    const int rc2 = try_wgrad2(dt, a, s);
    if (rc2 != 0) return rc2 < 0 ? rc2 : MMVAE_OK;
    if (a.P_planar || a.G_planar) { set_error("wgrad: planar operands need the patch-tile kernel"); return MMVAE_ERR_UNSUPPORTED; }
  }
  const int TA = a.Ca >= 64 ? 64 : ((a.Ca + 15) / 16) * 16;
  const int TB = a.Cb >= 64 ? 64 : ((a.Cb + 15) / 16) * 16;
  a.TA16 = TA / 16; a.TB16 = TB / 16;
  int tg = 256 / TB;                                   // LDS rows for G
  const int max_units_tg = 64 / (a.TA16 * a.TB16);     // <= 16 accumulators per wave
  if (tg > max_units_tg) tg = max_units_tg;
  if (tg > a.ntaps) tg = a.ntaps;
  if (tg < 1) tg = 1;
  a.TG = tg;
  const int tiles = ((a.Ca + TA - 1) / TA) * ((a.Cb + TB - 1) / TB);
  const int zg = (a.ntaps + tg - 1) / tg;
  const int PK = KV * VE;
  const long ksteps = ((long)a.M + PK - 1) / PK;
  long want = 1024 / ((long)tiles * zg);               // pixel chunks so that ~1024 blocks exist
  const long wsize = (long)a.Ca * a.Cb * a.ntaps;
  if (!a.scratch) { set_error("wgrad: the partial-image scratch is required (no atomic flush path)"); return MMVAE_ERR_ARG; }
  {                                                    // ...whose partial images fit the scratch (and stay below ~16M floats)
    long cap = (long)(kWgradScratchBytes / 4) / (wsize > 0 ? wsize : 1);
    const long cap2 = (16L << 20) / (wsize > 0 ? wsize : 1);
    if (cap > cap2 && cap2 >= 8) cap = cap2;
    if (cap < 1) { set_error("wgrad: one partial image (%ld floats) exceeds the scratch", wsize); return MMVAE_ERR_UNSUPPORTED; }
    if (want > cap) want = cap;
  }
  if (want < 1) want = 1;
  if (want > ksteps) want = ksteps;
  long steps_per = (ksteps + want - 1) / want;
  a.pix_per_block = (int)(steps_per * PK);
  const int gx = (int)(((long)a.M + a.pix_per_block - 1) / a.pix_per_block);
  dim3 grid(gx, tiles, zg), block(256);
  if (dt == DT_F32) hipLaunchKernelGGL((wgrad_kernel<float>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((wgrad_kernel<bf16_t>), grid, block, 0, s, a);
  MM_CHECK_RC(check_launch("wgrad"));
  WgradReduceArgs u; memset(&u, 0, sizeof(u));
  u.part = a.scratch; u.dW = a.dW; u.Ca = a.Ca; u.Cb = a.Cb; u.ntaps = a.ntaps; u.nparts = gx;
  u.Ca_valid = a.Ca_valid; u.Cb_valid = a.Cb_valid; u.sA = a.sA; u.sB = a.sB; u.scale = a.scale;
  for (int t = 0; t < 25; ++t) u.tap_off[t] = a.tap_off[t];
  return launch_wgrad_reduce(u, s);
}

// ============================================================================ weight packing
// fragment-major position of element (col, kk) of a [cols][Ktot] matrix (PackArgs::frag)
template <typename T>
__device__ __forceinline__ long frag_index(int col, int kk, int Ktot) {
  constexpr int VE = Elem<T>::kVec, CK = 4 * VE;     // elements per 16 bytes, per 64-byte chunk
  const int chunk = kk / CK, in = kk - chunk * CK;
  // column q of a wave's 32 is row 4*(q/8) + q%4 of its fragment (q/4)%2 (a lane's results are then 8 consecutive columns)
  const int q = col & 31;
  const int blk = 2 * (col >> 5) + ((q >> 2) & 1), r = 4 * (q >> 3) + (q & 3);
  return (((long)blk * (Ktot / CK) + chunk) * 64 + (in / VE) * 16 + r) * VE + (in % VE);
}
// the same for e4m3 bytes: a 64-byte chunk is 64 k-values = two MFMA k-blocks; a lane's 16 bytes = its 8 bytes of block 0, then of block 1
__device__ __forceinline__ long frag_index_f8(int col, int kk, int Ktot) {
  const int chunk = kk >> 6, in = kk & 63, half = in >> 5, w = in & 31;
  const int q = col & 31;
  const int blk = 2 * (col >> 5) + ((q >> 2) & 1), r = 4 * (q >> 3) + (q & 3);
  return (((long)blk * (Ktot >> 6) + chunk) * 64 + (w >> 3) * 16 + r) * 16 + half * 8 + (w & 7);
}
template <typename T>
__global__ void pack_kernel(PackArgs a) {
  const long total = (long)a.cols * a.ntaps * a.K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % a.K);
    const long q = i / a.K;
    const int tp = (int)(q % a.ntaps);
    const int col = (int)(q / a.ntaps);
    const bool real = (a.cols_valid <= 0 || col < a.cols_valid) && (a.K_valid <= 0 || k < a.K_valid);
    const float v = real ? a.src[(long)col * a.s_col + (long)k * a.s_k + a.tap_off[tp]] * a.scale : 0.f;
    Elem<T>::store(reinterpret_cast<T*>(a.dst) + (a.frag ? frag_index<T>(col, tp * a.K + k, a.ntaps * a.K) : i), v);
  }
}

struct PackJob {
  const float* src; void* dst; int cols, K, ntaps, s_col, s_k, cols_valid, K_valid; float scale; unsigned char tap_off[26]; unsigned char frag; unsigned char fp8;
};
__device__ __forceinline__ unsigned char f32_to_e4m3(float v) {
  v = fminf(fmaxf(v, -448.f), 448.f);
  return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
}
constexpr int kPackJobsPerLaunch = 48;
struct PackMulti { int njobs; int pad; PackJob jobs[kPackJobsPerLaunch]; };

template <typename T>
__global__ void pack_multi_kernel(PackMulti m) {
  const PackJob& a = m.jobs[blockIdx.y];
  const long total = (long)a.cols * a.ntaps * a.K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % a.K);
    const long q = i / a.K;
    const int tp = (int)(q % a.ntaps);
    const int col = (int)(q / a.ntaps);
    const bool real = (a.cols_valid <= 0 || col < a.cols_valid) && (a.K_valid <= 0 || k < a.K_valid);
    const float v = real ? a.src[(long)col * a.s_col + (long)k * a.s_k + a.tap_off[tp]] * a.scale : 0.f;
    if (a.fp8) reinterpret_cast<unsigned char*>(a.dst)[a.frag ? frag_index_f8(col, tp * a.K + k, a.ntaps * a.K) : i] = f32_to_e4m3(v);
    else Elem<T>::store(reinterpret_cast<T*>(a.dst) + (a.frag ? frag_index<T>(col, tp * a.K + k, a.ntaps * a.K) : i), v);
  }
}

static thread_local bool g_pack_batching = false;
static thread_local int g_pack_n = 0;
static thread_local std::vector<PackJob> g_pack_jobs;     // grows with the net (blocks_per_stage), never shrinks

void pack_batch_begin() { g_pack_batching = true; g_pack_n = 0; }

int pack_batch_flush(int dt, hipStream_t s) {
  g_pack_batching = false;
  for (int j0 = 0; j0 < g_pack_n; j0 += kPackJobsPerLaunch) {
    PackMulti m; m.njobs = g_pack_n - j0 < kPackJobsPerLaunch ? g_pack_n - j0 : kPackJobsPerLaunch; m.pad = 0;
    for (int j = 0; j < m.njobs; ++j) m.jobs[j] = g_pack_jobs[j0 + j];
    // 128 blocks per job: the largest weight (256 x 256 x 9) sets the launch's duration, its gather reads are latency-bound
    dim3 grid(128, m.njobs), block(256);
    if (dt == DT_F32) hipLaunchKernelGGL((pack_multi_kernel<float>), grid, block, 0, s, m);
    else hipLaunchKernelGGL((pack_multi_kernel<bf16_t>), grid, block, 0, s, m);
    int rc = check_launch("pack_multi");
    if (rc) { g_pack_n = 0; return rc; }
  }
  g_pack_n = 0;
  return MMVAE_OK;
}

int launch_pack(int dt, const PackArgs& a, hipStream_t s) {
  const long total = (long)a.cols * a.ntaps * a.K;
  if (total <= 0) return MMVAE_OK;
  if (g_pack_batching) {
    if (a.ntaps > 26) { set_error("pack: %d taps > 26", a.ntaps); return MMVAE_ERR_ARG; }
    if ((size_t)g_pack_n >= g_pack_jobs.size()) g_pack_jobs.resize(g_pack_jobs.size() + 256);
    PackJob& j = g_pack_jobs[g_pack_n++];
    j.src = a.src; j.dst = a.dst; j.cols = a.cols; j.K = a.K; j.ntaps = a.ntaps; j.s_col = a.s_col; j.s_k = a.s_k;
    j.cols_valid = a.cols_valid; j.K_valid = a.K_valid; j.scale = a.scale; j.fp8 = (unsigned char)(a.fp8 ? 1 : 0); j.frag = (unsigned char)(a.frag ? 1 : 0);
    for (int t = 0; t < a.ntaps; ++t) j.tap_off[t] = (unsigned char)a.tap_off[t];
    return MMVAE_OK;
  }
  if (a.fp8) { set_error("pack: fp8 packing is available in batched mode only"); return MMVAE_ERR_UNSUPPORTED; }
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (dt == DT_F32) hipLaunchKernelGGL((pack_kernel<float>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((pack_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, a);
  return check_launch("pack");
}

}  // namespace mmvae
