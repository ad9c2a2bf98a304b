// Device helpers shared by the patch-tile kernels (conv_tile.hip, conv_wgrad_*.hip).
#pragma once
#include "kernels.hpp"

namespace mmvae {

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ f32x4 mma_bf16(const Vec16& a, const Vec16& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma_f32v(const Vec16& a, const Vec16& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w[j]), __uint_as_float(b.w[j]), c, 0, 0, 0);
  return c;
}
template <typename T> __device__ __forceinline__ f32x4 mma_vec(const Vec16& a, const Vec16& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma_vec<bf16_t>(const Vec16& a, const Vec16& b, f32x4 c) { return mma_bf16(a, b, c); }
template <> __device__ __forceinline__ f32x4 mma_vec<float>(const Vec16& a, const Vec16& b, f32x4 c) { return mma_f32v(a, b, c); }

// pixel-major MFMA fragments out of a natural-layout [pixel][channel] LDS image (weight-gradient style reductions over pixels)
template <typename T> struct FragOps;
template <> struct FragOps<bf16_t> {
  // lane (g = lane>>4, i = lane&15): 8 pixels = two 4-pixel blocks; lane supplies the address of pixel-row q = i>>2,
  // channel quad p = i&3 of its group's block and receives channel i of the 4 pixels (ds_read_b64_tr_b16).
  __device__ static __forceinline__ Vec16 load(const char* base, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off1));
    Vec16 v;
    v.w[0] = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    v.w[1] = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    v.w[2] = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    v.w[3] = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return v;
  }
};

__device__ __forceinline__ void tile_origin(const TileGeom& g, int tile, int seg, int& n, int& hq0) {
  if (g.tiles_per_img > 0) {
    n = tile / g.tiles_per_img;
    hq0 = (tile - n * g.tiles_per_img) * g.qr;
  } else {
    n = tile * g.segs + seg;
    hq0 = 0;
  }
}

// patch pixel index (origin-relative, in pixels) of tile pixel p; p must be < segs*qr*Wq
__device__ __forceinline__ int patch_index(const TileGeom& g, int p) {
  const int per_seg = g.qr * g.Wq;
  const int seg = p / per_seg, rem = p - seg * per_seg;
  const int j = rem / g.Wq, wq = rem - j * g.Wq;
  return (seg * g.PR + j * g.SI) * g.PW + wq * g.SI;
}

// Blocks are dispatched round-robin over the 8 XCDs (block b runs on XCD b % 8; every XCD has its own L2).  Give each
// XCD one contiguous eighth of the tiles and let blocks of neighbouring rank walk neighbouring tiles, so that the halo
// rows two adjacent tiles share are served by one L2 instead of being fetched into two.  Falls back to the plain
// grid-stride walk when the grid is not a multiple of 8.
struct TileWalk { int first, step, end; };        // tiles first, first + step, ... < end
__device__ __forceinline__ TileWalk xcd_tile_walk(int ntiles, int enable) {
  const int B = gridDim.x, b = blockIdx.x;
  if (!enable || (B & 7) != 0) return TileWalk{b, B, ntiles};
  const int per = (ntiles + 7) >> 3;
  const int lo = (b & 7) * per;
  return TileWalk{lo + (b >> 3), B >> 3, min(ntiles, lo + per)};
}

}  // namespace mmvae
