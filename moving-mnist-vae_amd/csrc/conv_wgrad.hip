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
  const int per_round = 256 / cvg;
  const int npatch = a.g.segs * a.g.PR * a.g.PW;
  return (npatch + per_round - 1) / per_round;
}

int launch_wgrad2(int dt, const Wgrad2Args& a, int gx, int tiles_ab, int zg, int ta16, int tb16, hipStream_t s) {
  dim3 grid(gx, tiles_ab, zg);
  const int slots = wgrad2_patch_slots(a, dt, tb16 * 16);
  if (slots > 16) { set_error("wgrad2: %d staging slots > 16", slots); return MMVAE_ERR_UNSUPPORTED; }
  return dt == DT_F32 ? launch_wgrad2_f32(a, grid, ta16, tb16, slots, s) : launch_wgrad2_bf16(a, grid, ta16, tb16, slots, s);
}

// dW[a*sA + b*sB + tap_off[t]] += scale * sum_{p < nparts} part[p][t][a][b];  grid = (wsize/256, part chunks)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceArgs a) {
  const int ab = a.Ca * a.Cb;
  const int wsize = a.ntaps * ab;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= wsize) return;
  const int per = (a.nparts + gridDim.y - 1) / gridDim.y;
  const int p0 = blockIdx.y * per, p1 = min(a.nparts, p0 + per);
  const float* src = a.part + (long)p0 * wsize + i;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = p0;
  for (; p + 4 <= p1; p += 4, src += 4L * wsize) {
    s0 += src[0]; s1 += src[wsize]; s2 += src[2L * wsize]; s3 += src[3L * wsize];
  }
  for (; p < p1; ++p, src += wsize) s0 += src[0];
  const float sum = (s0 + s1) + (s2 + s3);
  const int t = i / ab, rem = i - t * ab;
  const int aa = rem / a.Cb, bb = rem - aa * a.Cb;
  // atomic: part chunks add up, and several taps may alias one destination (the pooled 1x1 heads share one weight)
  if (p1 > p0 && aa < a.Ca_valid && bb < a.Cb_valid) atomicAdd(a.dW + (long)aa * a.sA + (long)bb * a.sB + a.tap_off[t], sum * a.scale);
}

int launch_wgrad_reduce(const WgradReduceArgs& a, hipStream_t s) {
  const int wsize = a.ntaps * a.Ca * a.Cb;
  const int bx = (wsize + 255) / 256;
  int gy = (2048 + bx - 1) / bx;                 // ~2048 blocks in flight
  if (gy > a.nparts / 4) gy = a.nparts / 4;
  if (gy < 1) gy = 1;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, gy), dim3(256), 0, s, a);
  return check_launch("wgrad_reduce");
}

}  // namespace mmvae
