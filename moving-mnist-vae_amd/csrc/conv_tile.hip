// "Patch-tile" convolution kernels (v2) for the pixel-heavy, small-channel layers that dominate HBM traffic.
//
// A tile = up to 128 output-grid ("q") pixels: either `qr` consecutive q-rows of one image, or several whole
// images.  The input patch every tap of the tile touches is staged in LDS ONCE, in natural NHWC layout
// (16-byte vector writes, zero-filled padding, optional fused BN+ReLU), and MFMA fragments are gathered from it
// with per-lane LDS addresses:
//   * gather2_kernel  (Conv2d fwd / ConvT fwd / dgrads): B fragment = 8 consecutive channels of one tap
//     -> ds_read_b128 at patch(pixel) + tap offset.  The phase's whole weight matrix sits in LDS for the block's
//     lifetime (persistent tile loop), so HBM sees each input and output element once.
//   * wgrad2_kernel   (weight gradients): the reduction runs over pixels, so both operands need pixel-major
//     fragments; they come from the SAME natural-layout LDS images through ds_read_b64_tr_b16 (hardware
//     transpose, bf16) or plain ds_read_b32 (f32 mode).  Every tap reuses the staged patch by adding a constant
//     byte offset.  Accumulators live in registers across the persistent loop; the flush goes through LDS so
//     that global float atomics are issued on consecutive addresses (256 B per wave instruction).
#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

// Stage the patch of `Cs` channels [c0, c0+Cs) of X (tensor channels C) into LDS: layout [segs*PR][PW][Cs].
// Loads are issued in register batches of NB (all NB in flight, then NB LDS stores): staging is latency-bound.
template <typename T, int NB>
__device__ __forceinline__ void stage_patch(const TileGeom& g, int tile, const T* __restrict__ X, int C, int c0, int Cs,
                                            const float* sPro, int relu, Vec16* sPatch) {
  constexpr int VE = Elem<T>::kVec;
  const int cvec = Cs / VE;
  const int rowvecs = g.PW * cvec;
  const int total = g.segs * g.PR * rowvecs;
  int n1, hq01;
  tile_origin(g, tile, 0, n1, hq01);
  const bool multi = g.tiles_per_img == 0;
  const int h_base = hq01 * g.SI + g.oh;
  for (int base = threadIdx.x; base < total; base += 256 * NB) {
    Vec16 q[NB];
    int cvs[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int v = base + 256 * k;
      q[k] = Vec16{{0, 0, 0, 0}};
      cvs[k] = -1;
      if (v < total) {
        const int row = v / rowvecs, rv = v - row * rowvecs;
        const int pc = rv / cvec, cv = rv - pc * cvec;
        const int seg = row / g.PR, pr = row - seg * g.PR;
        const int n = multi ? n1 + seg : n1;
        const int hi = h_base + pr, wi = g.ow + pc;
        if (n < g.N && hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi) {
          q[k] = *reinterpret_cast<const Vec16*>(X + ((long)(n * g.Hi + hi) * g.Wi + wi) * C + c0 + cv * VE);
          cvs[k] = cv;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int v = base + 256 * k;
      if (v < total) {
        if (sPro && cvs[k] >= 0) {               // padding stays exactly zero
          float f[VE];
          Elem<T>::unpack(q[k], f);
#pragma unroll
          for (int j = 0; j < VE; ++j) {
            const float x = f[j] * sPro[cvs[k] * VE + j] + sPro[512 + cvs[k] * VE + j];
            f[j] = relu ? fmaxf(x, 0.f) : x;
          }
          q[k] = Elem<T>::pack(f);
        }
        sPatch[v] = q[k];
      }
    }
  }
}

// Same patch, but the source is PLANAR ([N][planes][Hi][Wi], element type ST = float or T) with only `planes` (<= 16)
// real channels: every patch pixel becomes a 16-channel NHWC LDS pixel whose channels >= planes are zero, so the
// 1-channel image (stem) and the 1-2-channel reconstruction gradient (tail) run through the same MFMA kernels.
template <typename T, typename ST>
__device__ __forceinline__ void stage_patch_planar(const TileGeom& g, int tile, const ST* __restrict__ X, int planes, int Cpad,
                                                   Vec16* sPatch) {
  constexpr int VE = Elem<T>::kVec;
  const int NV = Cpad / VE;                 // vec16 per staged pixel (Cpad in {VE, 8, 16})
  const int total = g.segs * g.PR * g.PW;
  int n1, hq01;
  tile_origin(g, tile, 0, n1, hq01);
  const bool multi = g.tiles_per_img == 0;
  const int h_base = hq01 * g.SI + g.oh;
  const long plane = (long)g.Hi * g.Wi;
  for (int v = threadIdx.x; v < total; v += 256) {
    const int row = v / g.PW, pc = v - row * g.PW;
    const int seg = row / g.PR, pr = row - seg * g.PR;
    const int n = multi ? n1 + seg : n1;
    const int hi = h_base + pr, wi = g.ow + pc;
    float f[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) f[c] = 0.f;
    if (n < g.N && hi >= 0 && hi < g.Hi && wi >= 0 && wi < g.Wi) {
      const ST* src = X + (long)n * planes * plane + (long)hi * g.Wi + wi;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < planes) f[c] = Elem<ST>::load(src + c * plane);
    }
#pragma unroll
    for (int k = 0; k < 16 / VE; ++k)
      if (k < NV) sPatch[v * NV + k] = Elem<T>::pack(f + k * VE);
  }
}

template <typename TO> __device__ __forceinline__ void store4v(TO* p, const float* v, bool acc);
template <> __device__ __forceinline__ void store4v<float>(float* p, const float* v, bool acc) {
  float4 o = make_float4(v[0], v[1], v[2], v[3]);
  if (acc) { const float4 e = *reinterpret_cast<const float4*>(p); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
  *reinterpret_cast<float4*>(p) = o;
}
template <> __device__ __forceinline__ void store4v<bf16_t>(bf16_t* p, const float* v, bool acc) {
  float f[4] = {v[0], v[1], v[2], v[3]};
  if (acc) {
    const uint2 e = *reinterpret_cast<const uint2*>(p);
    f[0] += __uint_as_float(e.x << 16); f[1] += __uint_as_float(e.x & 0xffff0000u);
    f[2] += __uint_as_float(e.y << 16); f[3] += __uint_as_float(e.y & 0xffff0000u);
  }
  uint2 o;
  o.x = pack2_bf16(f[0], f[1]);
  o.y = pack2_bf16(f[2], f[3]);
  *reinterpret_cast<uint2*>(p) = o;
}

// ============================================================================ gather2
// LDS carve: [weights: Cout rows x (kvp+1) vec][patch][sKoff: kvp ints][sPro: 1024 floats][sStat: 8*CT floats]

// ============================================================================ gather3: barrier-free streaming gather
// For thin layers (small K, small Cout) LDS staging of the input is pure overhead: no wave shares its pixels with
// another, only halos overlap and those are served by L1/L2.  Each wave owns 32 output pixels per iteration, loads its
// MFMA B-fragments (8 consecutive channels of one tap per lane) straight from global memory into registers, and reads
// the A-fragments from a read-only LDS copy of the phase's weight matrix (filled once per block; the only barrier).
// These launches are bound by index arithmetic, not by memory (measured: 0.32 of 0.62 ms remain with loads and stores
// disabled), so the address math is stripped down: power-of-two pixel decode by shifts, one LDS int4 per k-vector with
// the precomputed element offset and tap displacement, 32-bit element offsets.
// LDS carve: [weights: CT rows x (kvp+1) vec][sK: kvp int4][sPro: 1024 floats][sStat: 8*CT floats]
// NKS > 0: the number of 4-k-vector steps is a compile-time constant (fully unrolled, every load of a wave-tile in
// flight at once); NKS == 0: runtime loop.
template <typename T, typename TO, int CT16, int NKS>
__global__ __launch_bounds__(256) void gather3_kernel(GatherArgs a) {
  constexpr int VE = Elem<T>::kVec;
  constexpr int CT = CT16 * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Phase P = a.phases[blockIdx.z];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, r = lane & 15;
  const int cin_vecs = a.cin_vecs;
  const int kvecs = P.ntaps * cin_vecs;
  const int kvp = (kvecs + 3) & ~3;
  const int wrow = kvp + 1;
  Vec16* sW = reinterpret_cast<Vec16*>(smem);
  int4* sK = reinterpret_cast<int4*>(sW + CT * wrow);
  float* sPro = reinterpret_cast<float*>(sK + kvp);
  float* sStat = sPro + 1024;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wt = reinterpret_cast<const T*>(a.w) + P.w_off;
  TO* __restrict__ Y = reinterpret_cast<TO*>(a.y);
  const bool has_pro = a.pro_scale != nullptr;
  for (int v = t; v < CT * kvp; v += 256) {
    const int row = v / kvp, kv = v - row * kvp;
    sW[row * wrow + kv] = (kv < kvecs && row < a.Cout) ? *reinterpret_cast<const Vec16*>(Wt + ((long)row * kvecs + kv) * VE) : Vec16{{0, 0, 0, 0}};
  }
  for (int v = t; v < kvp; v += 256) {
    int4 e = make_int4(0, 1 << 28, 1 << 28, 0);       // padding k-vectors: displacement far out of range -> zero operand
    if (v < kvecs) {
      const int tap = v / cin_vecs, c0 = (v - tap * cin_vecs) * VE;
      const Tap tp = a.taps[P.tap0 + tap];
      e = make_int4((tp.dh * a.Wi + tp.dw) * a.Cin + c0, tp.dh, tp.dw, c0);
    }
    sK[v] = e;
  }
  if (has_pro) for (int i = t; i < a.Cin; i += 256) { sPro[i] = a.pro_scale[i]; sPro[512 + i] = a.pro_shift[i]; }
  __syncthreads();

  float st1[CT16][4], st2[CT16][4];
#pragma unroll
  for (int c = 0; c < CT16; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) { st1[c][j] = 0.f; st2[c][j] = 0.f; }
  const int HqWq = P.Hq * P.Wq;
  const int M = a.N * HqWq;
  const int nwt = (M + 31) >> 5;
  const int nks = kvp >> 2;
  const int sh_w = (P.Wq & (P.Wq - 1)) == 0 ? __builtin_ctz(P.Wq) : -1;       // power-of-two fast path
  const int sh_hw = (HqWq & (HqWq - 1)) == 0 ? __builtin_ctz(HqWq) : -1;
  const unsigned uHi = (unsigned)a.Hi, uWi = (unsigned)a.Wi;
  const int img = a.Hi * a.Wi;

  for (int wt = blockIdx.x * 4 + wv; wt < nwt; wt += gridDim.x * 4) {
    int xoff[2], ph0[2], pw0[2], yoff[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int m = wt * 32 + 16 * pt + r;
      int n, hq, wq;
      if (sh_w >= 0 && sh_hw >= 0) { n = m >> sh_hw; const int rem = m & (HqWq - 1); hq = rem >> sh_w; wq = rem & (P.Wq - 1); }
      else { n = m / HqWq; const int rem = m - n * HqWq; hq = rem / P.Wq; wq = rem - hq * P.Wq; }
      ph0[pt] = hq * a.SI;
      pw0[pt] = wq * a.SI;
      xoff[pt] = (n * img + ph0[pt] * a.Wi + pw0[pt]) * a.Cin;
      yoff[pt] = ((n * a.Ho + hq * a.SO + P.ph) * a.Wo + (wq * a.SO + P.pw)) * a.Cout;
      if (m >= M) ph0[pt] = 1 << 28;                 // every tap lands out of range -> zero operand, no store
    }
    f32x4 acc[CT16][2];
#pragma unroll
    for (int c = 0; c < CT16; ++c) { acc[c][0] = (f32x4){0, 0, 0, 0}; acc[c][1] = (f32x4){0, 0, 0, 0}; }
    auto kstep = [&](int ks) {
      const int kv = 4 * ks + gq;
      const int4 e = sK[kv];
      Vec16 b[2];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        // branch-free: always load (from offset 0 when the tap is out of range), then select zero
        const bool ok = (unsigned)(ph0[pt] + e.y) < uHi && (unsigned)(pw0[pt] + e.z) < uWi;
        const unsigned off = ok ? (unsigned)(xoff[pt] + e.x) : 0u;
        Vec16 v = *reinterpret_cast<const Vec16*>(X + off);
        if (has_pro) {                                  // uniform branch
          float f[VE];
          Elem<T>::unpack(v, f);
#pragma unroll
          for (int j = 0; j < VE; ++j) {
            const float x = f[j] * sPro[e.w + j] + sPro[512 + e.w + j];
            f[j] = a.pro_relu ? fmaxf(x, 0.f) : x;
          }
          v = Elem<T>::pack(f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) b[pt].w[q] = ok ? v.w[q] : 0u;
      }
#pragma unroll
      for (int c = 0; c < CT16; ++c) {
        const Vec16 af = sW[(16 * c + r) * wrow + kv];
        acc[c][0] = mma_vec<T>(af, b[0], acc[c][0]);
        acc[c][1] = mma_vec<T>(af, b[1], acc[c][1]);
      }
    };
    if constexpr (NKS > 0) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) kstep(ks);
    } else {
      for (int ks = 0; ks < nks; ++ks) kstep(ks);
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      if (ph0[pt] < (1 << 28)) {
#pragma unroll
        for (int c = 0; c < CT16; ++c) {
          const int co = 16 * c + 4 * gq;
          if (co < a.Cout) {
            float v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              v[jj] = acc[c][pt][jj];
              st1[c][jj] += v[jj];
              st2[c][jj] += v[jj] * v[jj];
            }
            store4v<TO>(Y + (unsigned)(yoff[pt] + co), v, a.accumulate != 0);
          }
        }
      }
    }
  }
  if (a.stats) {
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
          sStat[wv * 2 * CT + 16 * c + 4 * gq + j] = st1[c][j];
          sStat[wv * 2 * CT + CT + 16 * c + 4 * gq + j] = st2[c][j];
        }
    }
    __syncthreads();
    if (t < 2 * CT) {
      const float s = sStat[t] + sStat[2 * CT + t] + sStat[4 * CT + t] + sStat[6 * CT + t];
      const int which = t / CT, cl = t - which * CT;
      if (cl < a.Cout) a.stats[((long)blockIdx.z * gridDim.x + blockIdx.x) * 2 * a.Cout + (long)which * a.Cout + cl] = s;
    }
  }
}

size_t gather3_lds_bytes(const GatherArgs& a, int dt, int CT) {
  int worst = 0;
  for (int p = 0; p < a.nphase; ++p) {
    const int kvecs = a.phases[p].ntaps * a.cin_vecs, kvp = (kvecs + 3) & ~3;
    if (kvp > worst) worst = kvp;
  }
  return (size_t)CT * (worst + 1) * 16 + (size_t)worst * 16 + 1024 * 4 + (size_t)8 * CT * 4;
}

template <typename T, typename TO, int CT16>
static void launch_gather3_nks(const GatherArgs& a, dim3 grid, size_t lds, int nks, hipStream_t s) {
  dim3 block(256);
  switch (nks) {
    case 1: hipLaunchKernelGGL((gather3_kernel<T, TO, CT16, 1>), grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL((gather3_kernel<T, TO, CT16, 2>), grid, block, lds, s, a); break;
    case 4: hipLaunchKernelGGL((gather3_kernel<T, TO, CT16, 4>), grid, block, lds, s, a); break;
    case 8: hipLaunchKernelGGL((gather3_kernel<T, TO, CT16, 8>), grid, block, lds, s, a); break;
    case 9: hipLaunchKernelGGL((gather3_kernel<T, TO, CT16, 9>), grid, block, lds, s, a); break;
    default: hipLaunchKernelGGL((gather3_kernel<T, TO, CT16, 0>), grid, block, lds, s, a); break;
  }
}

template <typename T, typename TO>
static int launch_gather3_t(const GatherArgs& a, int dt, int gx, hipStream_t s) {
  int ct16 = (a.Cout + 15) / 16;
  if (ct16 == 3) ct16 = 4;
  const size_t lds = gather3_lds_bytes(a, dt, ct16 * 16);
  dim3 grid(gx, 1, a.nphase);
  // all phases of a launch must share the unrolled step count, else the runtime-loop instance is used
  int nks = -1;
  for (int p = 0; p < a.nphase; ++p) {
    const int k = (((a.phases[p].ntaps * a.cin_vecs) + 3) & ~3) >> 2;
    nks = (nks < 0 || nks == k) ? k : 0;
  }
  switch (ct16) {
    case 1: launch_gather3_nks<T, TO, 1>(a, grid, lds, nks, s); break;
    case 2: launch_gather3_nks<T, TO, 2>(a, grid, lds, nks, s); break;
    case 4: launch_gather3_nks<T, TO, 4>(a, grid, lds, nks, s); break;
    default: set_error("gather3: Cout=%d too large", a.Cout); return MMVAE_ERR_UNSUPPORTED;
  }
  int rc = check_launch("gather3");
  return rc ? rc : gx * a.nphase;
}

int launch_gather3(int dt, int out_dt, const GatherArgs& a, int gx, hipStream_t s) {
  if (dt == DT_F32) return launch_gather3_t<float, float>(a, dt, gx, s);
  if (out_dt != dt) { set_error("gather3: mixed output type unsupported"); return MMVAE_ERR_UNSUPPORTED; }
  return launch_gather3_t<bf16_t, bf16_t>(a, dt, gx, s);
}

}  // namespace mmvae
