// Host side of the patch-tile weight-gradient path + the reduction of the per-block partial images.
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

// dW[a*sA + b*sB + tap_off[t]] += scale * sum_{p < nparts} part[p][t][a][b]
// Block = 256 consecutive elements x one chunk of parts: lane l of every wave owns the float4 column l, wave w sums the
// parts w, w+4, ... of the chunk (8 loads in flight), the four waves combine through LDS.  grid = (wsize/256, chunks);
// with one chunk (large gradients) the result is added with a plain read-modify-write, else with float atomics.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceArgs a) {
  __shared__ float4 sSum[4][64];
  const int ab = a.Ca * a.Cb;
  const int wsize = a.ntaps * ab;                 // multiple of 256 (Ca, Cb multiples of 16)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per = (a.nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(a.nparts, p0 + per);
  const float* src = a.part + (long)blockIdx.x * 256 + lane * 4;
  float4 s[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = p0 + wv;
  for (; p + 12 < p1; p += 16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(src + (long)(p + 4 * k) * wsize);
      s[k].x += v.x; s[k].y += v.y; s[k].z += v.z; s[k].w += v.w;
    }
  }
  for (; p < p1; p += 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + (long)p * wsize);
    s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
  }
  sSum[wv][lane] = make_float4((s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y),
                               (s[0].z + s[1].z) + (s[2].z + s[3].z), (s[0].w + s[1].w) + (s[2].w + s[3].w));
  __syncthreads();
  const float* f = reinterpret_cast<const float*>(sSum);
  const int j = threadIdx.x;
  const float sum = (f[j] + f[256 + j]) + (f[512 + j] + f[768 + j]);
  const int i = blockIdx.x * 256 + j;
  const int t = i / ab, rem = i - t * ab;
  const int aa = rem / a.Cb, bb = rem - aa * a.Cb;
  if (p1 > p0 && aa < a.Ca_valid && bb < a.Cb_valid) {
    float* dst = a.dW + (long)aa * a.sA + (long)bb * a.sB + a.tap_off[t];
    if (a.exclusive) *dst += sum * a.scale;
    else atomicAdd(dst, sum * a.scale);
  }
}

int launch_wgrad_reduce(WgradReduceArgs a, hipStream_t s) {
  const int wsize = a.ntaps * a.Ca * a.Cb;
  if (wsize % 256) { set_error("wgrad_reduce: %d elements not a multiple of 256", wsize); return MMVAE_ERR_ARG; }
  const int bx = wsize / 256;
  int gy = (512 + bx - 1) / bx;                  // >= ~512 blocks ...
  if (gy > a.nparts / 16) gy = a.nparts / 16;    // ... of >= 16 parts (4 per wave) each
  if (gy < 1) gy = 1;
  // one chunk and distinct destinations (no two taps share a weight): plain read-modify-write
  bool distinct = true;
  for (int t = 0; t < a.ntaps && distinct; ++t)
    for (int u = 0; u < t; ++u) if (a.tap_off[u] == a.tap_off[t]) { distinct = false; break; }
  a.exclusive = (gy == 1 && distinct) ? 1 : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, gy), dim3(256), 0, s, a);
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

// ---------------------------------------------------------------- deep_conv host side
int launch_deep_conv_f32(const DeepArgs& a, int gx, hipStream_t s);
int launch_deep_conv_bf16(const DeepArgs& a, int gx, hipStream_t s);
int launch_deep_conv_bf16_f32(const DeepArgs& a, int gx, hipStream_t s);

size_t deep_conv_lds_bytes(const DeepArgs& a, int dt) {
  const size_t es = dtype_size(dt);
  const int CT = a.ct16 * 16;
  const int TPX = (8 / (a.ct16 / 4)) * a.npt * 16;
  const size_t pitch = (size_t)a.Cin * (a.fp8 ? 1 : es) + 16;
  return (size_t)(a.ipt * a.Hi * a.Wi + 1) * pitch + (size_t)2 * CT * (a.fp8 ? 5 : 9) * 16 + (size_t)a.ntaps_all * TPX * 2 + 1024 * 4;
}

int launch_deep_conv(int dt, int out_dt, const DeepArgs& a, int gx, hipStream_t s) {
  if (dt == DT_F32) return launch_deep_conv_f32(a, gx, s);
  if (out_dt == DT_F32) return launch_deep_conv_bf16_f32(a, gx, s);
  return launch_deep_conv_bf16(a, gx, s);
}

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
