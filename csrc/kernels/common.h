// Shared device helpers for the gfx950 (CDNA4) kernels.
// Wavefront = 64 lanes; all vector memory traffic is 16 B per lane where possible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

#define VPSA_EXPORT extern "C" __attribute__((visibility("default")))

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t u) {
  return __uint_as_float(((uint32_t)u) << 16);
}

// Unpack a dword holding two bf16 (lo = element 0).
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// Pack two floats to a bf16x2 dword (round-to-nearest-even, NaN preserving: the
// compiler lowers the cast to v_cvt_pk_bf16_f32 on gfx950).
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  bf16x2 v;
  v[0] = (bf16_t)a;
  v[1] = (bf16_t)b;
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// KV-cache page geometry. One page holds PAGE tokens of one layer for every KV head:
//   K part: [n_kv_heads][PAGE][HEAD_DIM]      (row = one key, 256 B)
//   V part: [n_kv_heads][HEAD_DIM][PAGE]      (transposed: row = one channel, 128 B)
// so both the QK^T A-operand (8 contiguous channels of a key) and the PV A-operand
// (8 contiguous keys of a channel) are plain 16-byte reads with no LDS transpose.
#define PAGE 64
#define HEAD_DIM 128
