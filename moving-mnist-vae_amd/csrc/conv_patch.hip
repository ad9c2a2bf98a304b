// patch_conv_kernel: pipelined patch-tile convolution for the thin, pixel-heavy layers (Conv2d fwd, ConvT fwd, dgrads)
// whose complete weight set (all stride-phases) fits LDS.
//
// A tile = up to 128 q-pixels (rows of one image or several whole images); q is the coarse grid: the output grid of a
// strided Conv2d / ConvT-dgrad, the input-resolution grid of a ConvT / strided-conv dgrad (each q-pixel then owns
// SO x SO output pixels, one per stride-phase).  Per tile the block
//   1. commits the input patch (prefetched into registers during the previous tile's MFMA phase) to LDS in natural
//      NHWC layout, zero-filled padding, fused BN+ReLU applied ONCE per element;
//   2. issues the global loads of the next tile;
//   3. runs every stride-phase from the one staged patch: B fragment = 8 consecutive channels of one tap
//      (ds_read_b128 at pixel base + per-k-vector offset), A fragment = LDS copy of the phase's weight matrix,
//      D[cout][pixel] so that a lane ends with 4 consecutive output channels of one pixel (8/16-byte NHWC stores);
//      per-channel sum / sum^2 for BatchNorm are accumulated from the f32 results.
// The staging slots of a thread are tile-invariant (element offset + one range check), its channel vector is fixed,
// so the BN scale/shift of the prologue sit in registers and the per-tile index arithmetic is a handful of VALU ops --
// these layers are bound by per-tile latency and index VALU, not by HBM or MFMA.
//
// LDS carve: [weights of all phases: CT rows x (kvp_p+1) vec16 each][patch][sKoff: sum kvp_p ints][sStat: 8*CT floats]
#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

template <typename TO> __device__ __forceinline__ void pstore4(TO* p, const float* v, bool acc);
template <> __device__ __forceinline__ void pstore4<float>(float* p, const float* v, bool acc) {
  float4 o = make_float4(v[0], v[1], v[2], v[3]);
  if (acc) { const float4 e = *reinterpret_cast<const float4*>(p); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
  *reinterpret_cast<float4*>(p) = o;
}
template <> __device__ __forceinline__ void pstore4<bf16_t>(bf16_t* p, const float* v, bool acc) {
  float f[4] = {v[0], v[1], v[2], v[3]};
  if (acc) {
    const uint2 e = *reinterpret_cast<const uint2*>(p);
    f[0] += __uint_as_float(e.x << 16); f[1] += __uint_as_float(e.x & 0xffff0000u);
    f[2] += __uint_as_float(e.y << 16); f[3] += __uint_as_float(e.y & 0xffff0000u);
  }
  uint2 o;
  o.x = (uint32_t)f32_to_bf16_bits(f[0]) | ((uint32_t)f32_to_bf16_bits(f[1]) << 16);
  o.y = (uint32_t)f32_to_bf16_bits(f[2]) | ((uint32_t)f32_to_bf16_bits(f[3]) << 16);
  *reinterpret_cast<uint2*>(p) = o;
}

constexpr int kPatchSlotInvalid = 0x40000000;
constexpr int kPlanarPrefetch = 2;      // planar inputs with <= 2 planes are prefetched like NHWC ones

template <typename T, typename TO, int CT16, int MAXG, bool PLANAR>
__global__ __launch_bounds__(256, MAXG <= 4 ? (CT16 == 1 ? 4 : (CT16 == 2 ? 3 : 2)) : 2) void patch_conv_kernel(PatchArgs a) {
  constexpr int VE = Elem<T>::kVec;
  constexpr int ES = sizeof(T);
  constexpr int CT = CT16 * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TileGeom g = a.g;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, gq = lane >> 4, r = lane & 15;
  const int cin_vecs = a.Cin / VE;
  const int npatch = g.segs * g.PR * g.PW;
  Vec16* sW = reinterpret_cast<Vec16*>(smem);
  Vec16* sPatch = sW + a.w_vecs;
  int* sKoff = reinterpret_cast<int*>(sPatch + npatch * cin_vecs);
  float* sStat = reinterpret_cast<float*>(sKoff + a.koff_total);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  TO* __restrict__ Y = reinterpret_cast<TO*>(a.y);
  const bool has_pro = a.pro_scale != nullptr;
  const bool multi = g.tiles_per_img == 0;
  constexpr bool planar = PLANAR;
  const bool planar_pf = planar && a.x_planes <= kPlanarPrefetch;

  // ---- block prologue: weight matrices and k-offset tables of every phase
  for (int p = 0; p < a.nphase; ++p) {
    const PatchPhase P = a.phases[p];
    const int kvecs = P.ntaps * cin_vecs, kvp = (kvecs + 3) & ~3, wrow = kvp + 1;
    const T* Wt = reinterpret_cast<const T*>(a.w) + P.w_off;
    Vec16* sWp = sW + P.w_vec0;
    for (int v = t; v < CT * kvp; v += 256) {
      const int row = v / kvp, kv = v - row * kvp;
      sWp[row * wrow + kv] = (kv < kvecs && row < a.Cout) ? *reinterpret_cast<const Vec16*>(Wt + ((long)row * kvecs + kv) * VE) : Vec16{{0, 0, 0, 0}};
    }
    for (int v = t; v < kvp; v += 256) {
      int off = 0;
      if (v < kvecs) {
        const int tap = v / cin_vecs, cv = v - tap * cin_vecs;
        const Tap tp = a.taps[P.tap0 + tap];
        off = (((tp.dh - g.oh) * g.PW + (tp.dw - g.ow)) * a.Cin + cv * VE) * ES;
      }
      sKoff[P.koff0 + v] = off;
    }
  }
  // ---- this thread's staging slots (tile-invariant)
  //  NHWC:   slot k = 16-byte vector `cv` of patch pixel pix0 + k*PS
  //  planar: slot k = patch pixel t + 256k (x_planes scalars, expanded to Cin zero-padded channels in LDS)
  const int cv = planar ? 0 : t % cin_vecs, pix0 = planar ? t : t / cin_vecs, PS = planar ? 256 : 256 / cin_vecs;
  float psc[VE], psh[VE];
#pragma unroll
  for (int j = 0; j < VE; ++j) {
    psc[j] = has_pro ? a.pro_scale[cv * VE + j] : 1.f;
    psh[j] = has_pro ? a.pro_shift[cv * VE + j] : 0.f;
  }
  int goff[MAXG], gchk[MAXG];
  const long plane = (long)g.Hi * g.Wi;
#pragma unroll
  for (int k = 0; k < MAXG; ++k) {
    const int pp = pix0 + k * PS;
    goff[k] = 0; gchk[k] = kPatchSlotInvalid;
    if (pp < npatch) {
      const int row = pp / g.PW, pc = pp - row * g.PW;
      const int seg = row / g.PR, pr = row - seg * g.PR;
      const int wi = g.ow + pc;
      bool ok = wi >= 0 && wi < g.Wi;
      if (multi) ok = ok && g.oh + pr >= 0 && g.oh + pr < g.Hi;
      goff[k] = planar ? (int)(seg * a.x_planes * plane) + pr * g.Wi + wi : ((seg * g.Hi + pr) * g.Wi + wi) * a.Cin + cv * VE;
      if (ok) gchk[k] = multi ? seg : pr;
    }
  }
  const unsigned lim = multi ? g.N : g.Hi;
  // ---- tile-independent decode of this lane's two q-pixels
  const int npix_tile = g.segs * g.qr * g.Wq;
  int pbase[2], pseg[2], pj[2], pwq[2];
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int p = 32 * wv + 16 * pt + r;
    const int pc = p < npix_tile ? p : 0;
    pbase[pt] = patch_index(g, pc) * a.Cin * ES;
    const int per_seg = g.qr * g.Wq;
    pseg[pt] = pc / per_seg;
    const int rem = pc - pseg[pt] * per_seg;
    pj[pt] = rem / g.Wq;
    pwq[pt] = rem - pj[pt] * g.Wq;
    if (p >= npix_tile) pseg[pt] = -1;
  }
  float st1[CT16][4], st2[CT16][4];
#pragma unroll
  for (int c = 0; c < CT16; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) { st1[c][j] = 0.f; st2[c][j] = 0.f; }
  const char* patch_bytes = reinterpret_cast<const char*>(sPatch);

  // ---- software pipeline: registers of the NEXT tile
  Vec16 gv[MAXG];
  int tb_c = 0;
  auto issue = [&](int tile) {
    int n, hq0;
    tile_origin(g, tile, 0, n, hq0);
    const int h_base = hq0 * g.SI + g.oh;
    tb_c = multi ? n : h_base;
    if constexpr (!planar) {
      const T* src = X + ((long)n * g.Hi + h_base) * g.Wi * a.Cin;
#pragma unroll
      for (int k = 0; k < MAXG; ++k) {
        gv[k] = Vec16{{0, 0, 0, 0}};
        if ((unsigned)(tb_c + gchk[k]) < lim) gv[k] = *reinterpret_cast<const Vec16*>(src + goff[k]);
      }
    } else if (planar_pf) {
      const long base = (long)n * a.x_planes * plane + (long)h_base * g.Wi;
#pragma unroll
      for (int k = 0; k < MAXG; ++k) {
        gv[k] = Vec16{{0, 0, 0, 0}};
        if ((unsigned)(tb_c + gchk[k]) < lim) {
#pragma unroll
          for (int c = 0; c < kPlanarPrefetch; ++c) {
            if (c < a.x_planes) {
              const long idx = base + goff[k] + c * plane;
              const float f = a.x_planar == 1 ? reinterpret_cast<const float*>(a.x)[idx] : Elem<T>::load(X + idx);
              gv[k].w[c] = __float_as_uint(f);
            }
          }
        }
      }
    }
  };
  auto commit = [&](int tile) {
    if constexpr (!planar) {
#pragma unroll
      for (int k = 0; k < MAXG; ++k) {
        const int pp = pix0 + k * PS;
        if (pp < npatch) {
          Vec16 q = gv[k];
          if (has_pro && (unsigned)(tb_c + gchk[k]) < lim) {       // padding stays exactly zero
            float f[VE];
            Elem<T>::unpack(q, f);
#pragma unroll
            for (int j = 0; j < VE; ++j) {
              const float x = f[j] * psc[j] + psh[j];
              f[j] = a.pro_relu ? fmaxf(x, 0.f) : x;
            }
            q = Elem<T>::pack(f);
          }
          sPatch[pp * cin_vecs + cv] = q;
        }
      }
    } else {
      // planar source: every patch pixel becomes a Cin-channel NHWC LDS pixel whose channels >= x_planes are zero
      int n, hq0;
      tile_origin(g, tile, 0, n, hq0);
      const long base = (long)n * a.x_planes * plane + (long)(hq0 * g.SI + g.oh) * g.Wi;
#pragma unroll
      for (int k = 0; k < MAXG; ++k) {
        const int pp = pix0 + k * PS;
        if (pp < npatch) {
          if (planar_pf) {
            float f[VE];
#pragma unroll
            for (int c = 0; c < VE; ++c) f[c] = c < kPlanarPrefetch ? __uint_as_float(gv[k].w[c]) : 0.f;
            sPatch[pp * cin_vecs] = Elem<T>::pack(f);
            for (int q = 1; q < cin_vecs; ++q) sPatch[pp * cin_vecs + q] = Vec16{{0, 0, 0, 0}};
          } else {
            const bool ok = (unsigned)(tb_c + gchk[k]) < lim;
            for (int q = 0; q < cin_vecs; ++q) {
              float f[VE];
#pragma unroll
              for (int c = 0; c < VE; ++c) {
                f[c] = 0.f;
                if (ok && q * VE + c < a.x_planes) {
                  const long idx = base + goff[k] + (long)(q * VE + c) * plane;
                  f[c] = a.x_planar == 1 ? reinterpret_cast<const float*>(a.x)[idx] : Elem<T>::load(X + idx);
                }
              }
              sPatch[pp * cin_vecs + q] = Elem<T>::pack(f);
            }
          }
        }
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < g.ntiles) issue(tile);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                 // previous tile's fragment reads are done (also orders the block prologue)
    commit(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < g.ntiles) issue(tile + gridDim.x);     // in flight during the MFMA phase
    // output bases of this lane's two q-pixels
    long obase[2]; int ohq[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      int n, hq0;
      tile_origin(g, tile, pseg[pt] < 0 ? 0 : pseg[pt], n, hq0);
      ohq[pt] = (pseg[pt] >= 0 && n < g.N) ? hq0 + pj[pt] : (1 << 29);
      obase[pt] = a.y_planes > 0 ? (long)n * a.y_planes * a.Ho * a.Wo + (long)ohq[pt] * a.SO * a.Wo + pwq[pt] * a.SO
                                 : (((long)n * a.Ho + (long)ohq[pt] * a.SO) * a.Wo + pwq[pt] * a.SO) * a.Cout;
    }
    for (int p = 0; p < a.nphase; ++p) {
      const PatchPhase P = a.phases[p];
      const int kvp = (P.ntaps * cin_vecs + 3) & ~3, wrow = kvp + 1, nks = kvp >> 2;
      const Vec16* sWp = sW + P.w_vec0;
      const int* sK = sKoff + P.koff0;
      f32x4 acc[CT16][2];
#pragma unroll
      for (int c = 0; c < CT16; ++c) { acc[c][0] = (f32x4){0, 0, 0, 0}; acc[c][1] = (f32x4){0, 0, 0, 0}; }
      for (int ks = 0; ks < nks; ++ks) {
        const int kv = 4 * ks + gq;
        const int koff = sK[kv];
        const Vec16 b0 = *reinterpret_cast<const Vec16*>(patch_bytes + pbase[0] + koff);
        const Vec16 b1 = *reinterpret_cast<const Vec16*>(patch_bytes + pbase[1] + koff);
#pragma unroll
        for (int c = 0; c < CT16; ++c) {
          const Vec16 af = sWp[(16 * c + r) * wrow + kv];
          acc[c][0] = mma_vec<T>(af, b0, acc[c][0]);
          acc[c][1] = mma_vec<T>(af, b1, acc[c][1]);
        }
      }
      // ---- epilogue: lane holds couts 16c + 4gq + j of q-pixel 32wv + 16pt + r, output pixel (hq*SO+ph, wq*SO+pw)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        if (ohq[pt] < P.Hq && pwq[pt] < P.Wq) {
          if (a.y_planes > 0) {
            // NCHW f32 output with y_planes (<= 16) real channels: the reconstruction layout of the reference
            if (gq * 4 < a.y_planes) {
              float* Yp = reinterpret_cast<float*>(a.y);
              const long hw = (long)a.Ho * a.Wo;
              const long pix = obase[pt] + (long)P.ph * a.Wo + P.pw;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const int co = 4 * gq + jj;
                if (co < a.y_planes) {
                  const float v = acc[0][pt][jj] + (a.bias ? a.bias[co] : 0.f);
                  st1[0][jj] += v;
                  st2[0][jj] += v * v;
                  Yp[pix + co * hw] = v;
                }
              }
            }
          } else {
            const long ob = obase[pt] + ((long)P.ph * a.Wo + P.pw) * a.Cout;
#pragma unroll
            for (int c = 0; c < CT16; ++c) {
              const int co = 16 * c + 4 * gq;
              if (co < a.Cout) {
                float v[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                  v[jj] = acc[c][pt][jj] + (a.bias ? a.bias[co + jj] : 0.f);
                  st1[c][jj] += v[jj];
                  st2[c][jj] += v[jj] * v[jj];
                }
                pstore4<TO>(Y + ob + co, v, a.accumulate != 0);
              }
            }
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
      const int cs = a.y_planes > 0 ? a.y_planes : a.Cout;      // channel count of the statistics rows
      if (cl < cs) a.stats[(long)blockIdx.x * 2 * cs + (long)which * cs + cl] = s;
    }
  }
}

size_t patch_conv_lds_bytes(const PatchArgs& a, int dt) {
  const int VE = dt == DT_F32 ? 4 : 8;
  const int cin_vecs = a.Cin / VE;
  int ct16 = (a.Cout + 15) / 16; if (ct16 == 3) ct16 = 4;
  return (size_t)a.w_vecs * 16 + (size_t)a.g.segs * a.g.PR * a.g.PW * cin_vecs * 16 + (size_t)a.koff_total * 4 + (size_t)8 * ct16 * 16 * 4;
}

int patch_conv_slots(const PatchArgs& a, int dt) {
  const int VE = dt == DT_F32 ? 4 : 8;
  const int npatch = a.g.segs * a.g.PR * a.g.PW;
  const int per_round = a.x_planar ? 256 : 256 / (a.Cin / VE);
  return (npatch + per_round - 1) / per_round;
}

template <typename T, typename TO>
static int launch_patch_t(const PatchArgs& a, int dt, int gx, hipStream_t s) {
  int ct16 = (a.Cout + 15) / 16;
  if (ct16 == 3) ct16 = 4;
  const size_t lds = patch_conv_lds_bytes(a, dt);
  const int slots = patch_conv_slots(a, dt);
  dim3 grid(gx), block(256);
#define MMVAE_PC(C_) do { \
    if (a.x_planar) { if (slots <= 4) hipLaunchKernelGGL((patch_conv_kernel<T, TO, C_, 4, true>), grid, block, lds, s, a); \
                      else hipLaunchKernelGGL((patch_conv_kernel<T, TO, C_, 12, true>), grid, block, lds, s, a); } \
    else { if (slots <= 4) hipLaunchKernelGGL((patch_conv_kernel<T, TO, C_, 4, false>), grid, block, lds, s, a); \
           else hipLaunchKernelGGL((patch_conv_kernel<T, TO, C_, 12, false>), grid, block, lds, s, a); } } while (0)
  if (slots > 12) { set_error("patch_conv: %d staging slots > 12", slots); return MMVAE_ERR_UNSUPPORTED; }
  switch (ct16) {
    case 1: MMVAE_PC(1); break;
    case 2: MMVAE_PC(2); break;
    case 4: MMVAE_PC(4); break;
    default: set_error("patch_conv: Cout=%d too large", a.Cout); return MMVAE_ERR_UNSUPPORTED;
  }
#undef MMVAE_PC
  const int rc = check_launch("patch_conv");
  return rc ? rc : gx;
}

int launch_patch_conv(int dt, int out_dt, const PatchArgs& a, int gx, hipStream_t s) {
  if (dt == DT_F32) return launch_patch_t<float, float>(a, dt, gx, s);
  if (out_dt == DT_F32) return launch_patch_t<bf16_t, float>(a, dt, gx, s);
  return launch_patch_t<bf16_t, bf16_t>(a, dt, gx, s);
}

}  // namespace mmvae
