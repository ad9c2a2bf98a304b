// Shared device helpers for the gfx950 (CDNA4) conv-VAE kernels.
// Wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmvae.h"

namespace mmvae {

constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- storage element types -------------------------------------------------
// Activations / packed weights are stored either as f32 or bf16 ("T"); all
// accumulation is f32.  A "kvec" is one 16-byte vector of T (4 f32 or 8 bf16).
struct alignas(16) Vec16 { uint32_t w[4]; };
// Once-read / once-written streams (tensors several times the Infinity Cache): non-temporal 16-byte accesses
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__device__ __forceinline__ Vec16 load_nt(const Vec16* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  Vec16 r; r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  return r;
}
__device__ __forceinline__ void store_nt(Vec16* p, const Vec16& q) {
  u32x4_t v; v.x = q.w[0]; v.y = q.w[1]; v.z = q.w[2]; v.w = q.w[3];
  __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(p));
}

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
__device__ __forceinline__ uint2 load_nt(const uint2* p) {
  const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void store_nt(uint2* p, const uint2& q) {
  u32x2_t v; v.x = q.x; v.y = q.y;
  __builtin_nontemporal_store(v, reinterpret_cast<u32x2_t*>(p));
}

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
// round-to-nearest-even, NaN preserved (plain cast lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}

// two floats -> one dword of two bf16 (round-to-nearest-even): a single v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kVec = 4;          // elements per 16 B
  static constexpr int kDtype = 0;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
  __device__ static __forceinline__ void unpack(const Vec16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v.w[i]);
  }
  __device__ static __forceinline__ Vec16 pack(const float* f) {
    Vec16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = __float_as_uint(f[i]);
    return v;
  }
};
struct bf16_t { uint16_t bits; };
template <> struct Elem<bf16_t> {
  static constexpr int kVec = 8;
  static constexpr int kDtype = 1;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_bits_to_f32(p->bits); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { p->bits = f32_to_bf16_bits(v); }
  __device__ static __forceinline__ void unpack(const Vec16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v.w[i] << 16);
      f[2 * i + 1] = __uint_as_float(v.w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ Vec16 pack(const float* f) {
    Vec16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v.w[i] = pack2_bf16(f[2 * i], f[2 * i + 1]);
    return v;
  }
};

// ---- wave / block reductions -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of NV floats per thread; result valid in thread 0.  smem: NV * (blockDim/64) floats
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) smem[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += smem[w * NV + i];
      v[i] = s;
    }
  }
}

// ---- error plumbing (host); the MMVAE_* codes come from the public header
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// algorithmic bytes (every operand touched once) of the launch the next check_launch() closes -- recorded by the launch census only
void note_launch_bytes(double bytes);

}  // namespace mmvae
