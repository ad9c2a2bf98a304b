// Host side of the patch-tile weight-gradient path + the reduction of the per-block partial images.
#include <cstdlib>
#include "kernels.hpp"

namespace mmvae {

int launch_wgrad2_bf16(const Wgrad2Args& a, dim3 grid, int ta16, int tb16, int maxg, hipStream_t s);
int launch_wgrad2_f32(const Wgrad2Args& a, dim3 grid, int ta16, int tb16, int maxg, hipStream_t s);

int wgrad2_taps_per_block(int ta16, int tb16, int ntaps) {
  int mt = 12 / (ta16 * tb16); if (mt < 1) mt = 1; if (mt > 4) mt = 4;
  const int tg = ntaps >= 4 ? 4 * mt : mt;
  return tg < ntaps ? tg : ntaps;
}

size_t wgrad2_lds_bytes(const Wgrad2Args& a, int dt, int TA, int TB) {
  const size_t es = dtype_size(dt);
  const size_t stage = (size_t)a.g.TP * TA * es + (size_t)a.g.segs * a.g.PR * a.g.PW * TB * es + 32 * 4;
  const size_t acc1 = (size_t)TA * TB * 4;              // per-tap flush image
  return stage > acc1 ? stage : acc1;
}

// staging slots per thread for the G patch (16-byte vectors / 256 threads)
int wgrad2_patch_slots(const Wgrad2Args& a, int dt, int TB) {
  const int cvg = TB * (int)dtype_size(dt) / 16;
  const int nt = 64 * (a.nw > 0 ? a.nw : 4);
  const int per_round = a.G_planar ? nt : nt / cvg;
  const int npatch = a.g.segs * a.g.PR * a.g.PW;
  return (npatch + per_round - 1) / per_round;
}

int launch_wgrad2(int dt, const Wgrad2Args& a, int gx, int tiles_ab, int zg, int ta16, int tb16, hipStream_t s) {
  dim3 grid(gx, tiles_ab, zg);
  const int slots = wgrad2_patch_slots(a, dt, tb16 * 16);
  if (slots > 16) { set_error("wgrad2: %d staging slots > 16", slots); return MMVAE_ERR_UNSUPPORTED; }
  return dt == DT_F32 ? launch_wgrad2_f32(a, grid, ta16, tb16, slots, s) : launch_wgrad2_bf16(a, grid, ta16, tb16, slots, s);
}

// dW[a*sA + b*sB + tap_off[t]] += scale * sum_{p < nparts} part[p][t][a][b]      -- no atomics, fixed summation order
// A block owns E = 4*LANES consecutive elements of the image and ALL parts of them: thread (rg, l) sums the float4 column l of the
// parts rg, rg + RG, ... (four independent accumulators, 16-byte loads), the RG row groups are combined by a fixed binary tree in
// LDS, row group 0 adds the result into the weight layout with a plain read-modify-write (it is the only writer of its elements).
// LANES is chosen from the image size alone (>= ~256 blocks where the image allows), so the order of every addition is a function of
// the shapes: the same inputs give the same bits.  Taps that share a destination (the encoder heads: the 2x2 pooled positions all
// accumulate into one 1x1 weight) are folded into the part axis by the launcher ([part][tap] is contiguous).
template <int LANES>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceArgs a) {
  constexpr int RG = 256 / LANES;
  __shared__ float4 sSum[256];
  const int ab = a.Ca * a.Cb;
  const long wsize = a.part_stride;
  const int l = threadIdx.x % LANES, rg = threadIdx.x / LANES;
  const float* src = a.part + (long)blockIdx.x * (4 * LANES) + l * 4;
  float4 s[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = rg;
  for (; p + 3 * RG < a.nparts; p += 4 * RG) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(src + (long)(p + k * RG) * wsize);
      s[k].x += v.x; s[k].y += v.y; s[k].z += v.z; s[k].w += v.w;
    }
  }
  for (; p < a.nparts; p += RG) {
    const float4 v = *reinterpret_cast<const float4*>(src + (long)p * wsize);
    s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
  }
  sSum[threadIdx.x] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                                  (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
#pragma unroll
  for (int st = RG / 2; st >= 1; st >>= 1) {
    if (rg < st) {
      const float4 o = sSum[threadIdx.x + st * LANES];
      float4& m = sSum[threadIdx.x];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
    }
    __syncthreads();
  }
  if (rg == 0) {
    const float4 m = sSum[l];
    const float v[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = (long)blockIdx.x * (4 * LANES) + l * 4 + j;
      const int t = (int)(i / ab), rem = (int)(i - (long)t * ab);
      const int aa = rem / a.Cb, bb = rem - aa * a.Cb;
      if (aa < a.Ca_valid && bb < a.Cb_valid) a.dW[(long)aa * a.sA + (long)bb * a.sB + a.tap_off[t]] += v[j] * a.scale;
    }
  }
}

int launch_wgrad_reduce(WgradReduceArgs a, hipStream_t s) {
  if (a.nparts < 1) return MMVAE_OK;
  // taps that share a destination become extra parts of a one-tap image (all of them must then share it)
  bool distinct = true;
  for (int t = 0; t < a.ntaps && distinct; ++t)
    for (int u = 0; u < t; ++u) if (a.tap_off[u] == a.tap_off[t]) { distinct = false; break; }
  if (a.part_stride <= 0) a.part_stride = (long)a.ntaps * a.Ca * a.Cb;
  if (!distinct) {
    for (int t = 1; t < a.ntaps; ++t)
      if (a.tap_off[t] != a.tap_off[0]) { set_error("wgrad_reduce: taps share some destinations but not all"); return MMVAE_ERR_UNSUPPORTED; }
    if (a.part_stride != (long)a.ntaps * a.Ca * a.Cb) { set_error("wgrad_reduce: shared destinations need densely packed partial images"); return MMVAE_ERR_UNSUPPORTED; }
    a.nparts *= a.ntaps; a.ntaps = 1; a.part_stride = (long)a.Ca * a.Cb;
  }
  const long wsize = (long)a.ntaps * a.Ca * a.Cb;
  if (wsize % 256) { set_error("wgrad_reduce: %ld elements not a multiple of 256", wsize); return MMVAE_ERR_ARG; }
  a.exclusive = 1;
  note_launch_bytes((double)a.nparts * a.part_stride * 4.0);
  // elements per block: 256 .. 32 (whole 128-byte lines per part), aiming at >= 256 blocks
  int lanes = 64;
  while (lanes > 8 && wsize / (4 * lanes) < 256) lanes >>= 1;
  const dim3 grid((unsigned)(wsize / (4 * lanes)));
  switch (lanes) {
    case 64: hipLaunchKernelGGL(wgrad_reduce_kernel<64>, grid, dim3(256), 0, s, a); break;
    case 32: hipLaunchKernelGGL(wgrad_reduce_kernel<32>, grid, dim3(256), 0, s, a); break;
    case 16: hipLaunchKernelGGL(wgrad_reduce_kernel<16>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(wgrad_reduce_kernel<8>, grid, dim3(256), 0, s, a); break;
  }
  return check_launch("wgrad_reduce");
}

// ---------------------------------------------------------------- patch_conv host side
int launch_patch_conv_f32(const PatchArgs& a, int gx, hipStream_t s);
int launch_patch_conv_bf16(const PatchArgs& a, int gx, hipStream_t s);
int launch_patch_conv_bf16_f32(const PatchArgs& a, int gx, hipStream_t s);

size_t patch_conv_lds_bytes(const PatchArgs& a, int dt) {
  const int VE = dt == DT_F32 ? 4 : 8;
  const int cin_vecs = a.Cin / VE;
  int ct16 = (a.Cout + 15) / 16; if (ct16 == 3) ct16 = 4;
  size_t b = (size_t)a.w_vecs * 16 + (size_t)a.g.segs * a.g.PR * a.g.PW * cin_vecs * 16 + (size_t)a.koff_total * 4 + (size_t)8 * ct16 * 16 * 4 +
             (a.uni ? (size_t)4 * a.out_wave_bytes : 0);
  b = (b + 15) & ~(size_t)15;
  if (a.x2) b += (size_t)ct16 * 16 * (a.kvp2 + 1) * 16 + ((size_t)a.g.segs * a.g.qr * a.g.Wq * (a.Cin2 / VE) + 8) * 16;
  return b;
}
// Vec16 offsets (from the start of LDS) of the second source's weights / tile: right behind everything else
void patch_conv_x2_carve(PatchArgs& a, int dt) {
  PatchArgs z = a; z.x2 = nullptr;
  const size_t base = patch_conv_lds_bytes(z, dt);
  int ct16 = (a.Cout + 15) / 16; if (ct16 == 3) ct16 = 4;
  a.w2_vec0 = (int)(base / 16);
  a.x2_vec0 = a.w2_vec0 + ct16 * 16 * (a.kvp2 + 1);
}

int patch_conv_slots(const PatchArgs& a, int dt) {
  const int VE = dt == DT_F32 ? 4 : 8;
  const int npatch = a.g.segs * a.g.PR * a.g.PW;
  const int per_round = a.x_planar ? 256 : 256 / (a.Cin / VE);
  return (npatch + per_round - 1) / per_round;
}

int launch_patch_conv(int dt, int out_dt, const PatchArgs& a, int gx, hipStream_t s) {
  if (dt == DT_F32) return launch_patch_conv_f32(a, gx, s);
  if (out_dt == DT_F32) return launch_patch_conv_bf16_f32(a, gx, s);
  return launch_patch_conv_bf16(a, gx, s);
}

// ---------------------------------------------------------------- deep2_conv host side
int launch_deep2_conv_f32(const DeepArgs& a, int gx, hipStream_t s);
int launch_deep2_conv_bf16(const DeepArgs& a, int gx, hipStream_t s);
int launch_deep2_conv_bf16_f32(const DeepArgs& a, int gx, hipStream_t s);

size_t deep2_conv_lds_bytes(const DeepArgs& a, int dt) {
  const size_t pitch = a.fp8 ? (size_t)a.Cin + 16 : (size_t)a.Cin * dtype_size(dt) + 32;
  return (size_t)(a.ipt * a.Hi * a.Wi + 1) * pitch + (size_t)a.npt * 16 * 4 + (size_t)a.nw * 64 * 4 + (size_t)a.ntaps_all * a.npt * 16 * 2;
}

int launch_deep2_conv(int dt, int out_dt, const DeepArgs& a, int gx, hipStream_t s) {
  if (dt == DT_F32) return launch_deep2_conv_f32(a, gx, s);
  if (out_dt == DT_F32) return launch_deep2_conv_bf16_f32(a, gx, s);
  return launch_deep2_conv_bf16(a, gx, s);
}

}  // namespace mmvae
