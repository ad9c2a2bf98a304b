// join_conv1_fwd_kernel: the residual join of a DeconvBottleneck fused with the NEXT block's 1x1 conv1 (reference model.py:70-88: `out += shortcut;
// relu` of block i, then `conv1` of block i + 1), as a per-wave stream (round 4).
//
//   out[p][c]  = relu(s2[c] y2[p][c] + b2[c] + ss[c] ys[p][c] + bs[c])          (the join; written: the upsample branch and the backward pass read it)
//   y1[p][co]  = sum_c W1[co][c] out[p][c]                                       (the next block's conv1, 16 output channels, no bias)
//   stats      = per-channel (sum, sum of squares) of y1 from the f32 accumulators (bn1's batch statistics), one partial row per block
//
// affine_join_kernel wrote `out` and a patch-tile conv launch read it back: 3 + 2 tensor passes.  Here one step = 32 pixels: the two branch rows
// come in as whole wave loads (next step in flight in registers), the joined row goes out as whole wave stores AND into wave-private LDS, from
// where it is the B operand of one (CIN = 32) / one half-empty (CIN = 16) 16x16x32 MFMA per 16 pixels -- 2 + 2 passes, one launch.
#include <hip/hip_runtime.h>

#include <string.h>

#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmvae {

struct JoinConv1Args {
  const void* y2; const void* ys; const float* s2; const float* b2; const float* ss; const float* bs;
  const void* w1d;            // the next block's conv1, packed "down" form [16][CIN] bf16
  void* out; void* y1; float* stats;
  long nsteps;                // steps of 32 pixels
};

template <int CIN16>
__global__ __launch_bounds__(256, 4) void join_conv1_fwd_kernel(JoinConv1Args a) {
  constexpr int PXB = CIN16 * 32;              // bytes per joined pixel
  constexpr int STEPB = 32 * PXB;              // bytes per step (32 pixels)
  constexpr int NV = STEPB / 1024;             // 16-byte vectors per lane and tensor
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), gq = lane >> 4, r = lane & 15;
  char* xrow = smem + wv * STEPB;
  float c2s[8], c2b[8], css[8], csb[8];
  {
    const int c = (lane % (PXB / 16)) * 8;     // (the same channel group for every vector of the lane: 64 lanes x 16 B = whole pixels)
#pragma unroll
    for (int j = 0; j < 8; ++j) { c2s[j] = a.s2[c + j]; c2b[j] = a.b2[c + j]; css[j] = a.ss[c + j]; csb[j] = a.bs[c + j]; }
  }
  // A = W1[co = r][k = 8 gq ..]: CIN = 16 fills k < 16 (gq < 2), the rest of the K-step is zero
  Vec16 wA = Vec16{{0, 0, 0, 0}};
  if (CIN16 == 2 || gq < 2) wA = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.w1d) + r * PXB + gq * 16);
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  const long wstep = (long)gridDim.x * 4;
  long step = (long)blockIdx.x * 4 + wv;
  Vec16 vy[NV], vs[NV];
  auto issue = [&](long st) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      vy[v] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.y2) + st * STEPB + (lane + 64 * v) * 16);
      vs[v] = *reinterpret_cast<const Vec16*>(reinterpret_cast<const char*>(a.ys) + st * STEPB + (lane + 64 * v) * 16);
    }
  };
  if (step < a.nsteps) issue(step);
  while (step < a.nsteps) {
    const long cur = step;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float y[8], sh[8];
      Elem<bf16_t>::unpack(vy[v], y);
      Elem<bf16_t>::unpack(vs[v], sh);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = y[j] * c2s[j] + c2b[j];
        x += sh[j] * css[j] + csb[j];
        y[j] = fmaxf(x, 0.f);
      }
      const Vec16 o = Elem<bf16_t>::pack(y);
      *reinterpret_cast<Vec16*>(reinterpret_cast<char*>(a.out) + cur * STEPB + (lane + 64 * v) * 16) = o;
      *reinterpret_cast<Vec16*>(xrow + (lane + 64 * v) * 16) = o;
    }
    step += wstep;
    if (step < a.nsteps) issue(step);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const Vec16 b = *reinterpret_cast<const Vec16*>(xrow + (16 * pt + r) * PXB + (CIN16 == 2 ? gq : (gq & 1)) * 16);
      const f32x4 d = mma_bf16(wA, b, (f32x4){0, 0, 0, 0});          // D[co = 4 gq + j][pixel = 16 pt + r]
      float v[4] = {d[0], d[1], d[2], d[3]};
      dstore4<bf16_t>(reinterpret_cast<bf16_t*>(a.y1) + cur * 512 + (16 * pt + r) * 16 + 4 * gq, v, false);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] += v[j]; s1[j] += v[j] * v[j]; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (a.stats) {
    // the 16 pixel-lanes of a row (DPP), then the waves in order through LDS: one partial row [2][16] per block, fixed order
    float* sb = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s0[j] = row16_sum(s0[j]); s1[j] = row16_sum(s1[j]); }
    __syncthreads();                             // (the x rows are dead)
    if (r == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { sb[wv * 32 + 4 * gq + j] = s0[j]; sb[wv * 32 + 16 + 4 * gq + j] = s1[j]; }
    }
    __syncthreads();
    if (t < 32) {
      float sum = sb[t];
      for (int w = 1; w < 4; ++w) sum += sb[w * 32 + t];
      a.stats[(long)blockIdx.x * 32 + t] = sum;
    }
  }
}

bool join_conv1_fwd_ok(int dt, int C, int Cout_next, long npix) {
  return dt == DT_BF16 && (C == 16 || C == 32) && Cout_next == 16 && npix > 0 && npix % 32 == 0;
}

// returns the number of blocks (= rows of `stats`, [2][16] each: sum, sum of squares of y1) or <0
int launch_join_conv1_fwd(int C, const void* y2, const float* s2, const float* b2, const void* ys, const float* ss, const float* bs, const void* w1_down,
                          void* out, void* y1, float* stats, long npix, hipStream_t s) {
  JoinConv1Args a; memset(&a, 0, sizeof(a));
  a.y2 = y2; a.ys = ys; a.s2 = s2; a.b2 = b2; a.ss = ss; a.bs = bs; a.w1d = w1_down; a.out = out; a.y1 = y1; a.stats = stats; a.nsteps = npix / 32;
  int gx = 1024;
  while (gx > 8 && (long)gx * 4 > a.nsteps) gx -= 8;
  note_launch_bytes((double)npix * (3.0 * C + 16.0) * 2.0);        // y2, ys read; out, y1 written (bf16)
  if (C == 16) hipLaunchKernelGGL(join_conv1_fwd_kernel<1>, dim3(gx), dim3(256), 4 * 1024, s, a);
  else hipLaunchKernelGGL(join_conv1_fwd_kernel<2>, dim3(gx), dim3(256), 4 * 2048, s, a);
  const int rc = check_launch("join_conv1_fwd");
  return rc ? rc : gx;
}

}  // namespace mmvae
