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

// 4 consecutive output channels of one pixel (an MFMA D-fragment column), optionally accumulated onto what is stored
template <typename TO> __device__ __forceinline__ void dstore4(TO* p, const float* v, bool acc);
template <> __device__ __forceinline__ void dstore4<float>(float* p, const float* v, bool acc) {
  float4 o = make_float4(v[0], v[1], v[2], v[3]);
  if (acc) { const float4 e = *reinterpret_cast<const float4*>(p); o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w; }
  *reinterpret_cast<float4*>(p) = o;
}
template <> __device__ __forceinline__ void dstore4<bf16_t>(bf16_t* p, const float* v, bool acc) {
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

// fp8 forward (BASELINE configs[4]): 8 f32 -> 8 OCP e4m3 bytes, saturating at the largest finite value (448)
__device__ __forceinline__ uint2 pack8_fp8(const float* f) {
  uint2 o; o.x = 0; o.y = 0;
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = fminf(fmaxf(f[j], -448.f), 448.f);
  o.x = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], o.x, false);
  o.x = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], o.x, true);
  o.y = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], o.y, false);
  o.y = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], o.y, true);
  return o;
}

// e4m3 STORAGE of the two largest activations in fp8 mode (decoder.uplayer5's branch outputs): 4 channels of a pixel <-> one dword
__device__ __forceinline__ uint32_t pack4_fp8(float f0, float f1, float f2, float f3) {
  uint32_t o = 0;
  o = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(f0, -448.f), 448.f), fminf(fmaxf(f1, -448.f), 448.f), o, false);
  o = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(f2, -448.f), 448.f), fminf(fmaxf(f3, -448.f), 448.f), o, true);
  return o;
}
__device__ __forceinline__ void unpack4_fp8(uint32_t v, float* f) {
  const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(v, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(v, true);
  f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}
// 4 bf16 channels (8 bytes) of a pixel
__device__ __forceinline__ void unpack4_bf16(uint2 v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u); f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}

// 8 consecutive output channels of one pixel
template <typename TO> __device__ __forceinline__ void dstore8(TO* p, const float* v, bool acc);
template <> __device__ __forceinline__ void dstore8<float>(float* p, const float* v, bool acc) {
  dstore4<float>(p, v, acc);
  dstore4<float>(p + 4, v + 4, acc);
}
template <> __device__ __forceinline__ void dstore8<bf16_t>(bf16_t* p, const float* v, bool acc) {
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = v[j];
  if (acc) {
    const uint4 e = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] += __uint_as_float(w[j] << 16); f[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u); }
  }
  uint4 o;
  o.x = pack2_bf16(f[0], f[1]); o.y = pack2_bf16(f[2], f[3]); o.z = pack2_bf16(f[4], f[5]); o.w = pack2_bf16(f[6], f[7]);
  *reinterpret_cast<uint4*>(p) = o;
}
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15); every lane receives it
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  return v;
}
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
