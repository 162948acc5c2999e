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

// KV-cache page geometry. One page holds PAGE tokens of one layer for every KV head, stored so
// that every MFMA fragment load of the decode kernel (16 rows x 64 B) is one contiguous KiB, i.e.
// covers whole 128-byte lines (a row-major K page made every fragment load touch half of each of
// 16 lines, and the second halves re-missed in L2: 9 % more HBM requests than K/V bytes):
//   K part: [n_kv_heads][HEAD_DIM / 32][PAGE][32]     channel block cb, key, 32 channels (64 B)
//   V part: [n_kv_heads][PAGE / 32][HEAD_DIM][32]     key half, channel, 32 keys (64 B)
// The QK^T A-operand (8 contiguous channels of a key) and the PV A-operand (8 contiguous keys of
// a channel) are still plain 16-byte reads with no transpose.
#define PAGE 64
#define HEAD_DIM 128

// Element offset inside one KV head's K part / V part.
__device__ __forceinline__ int k_elem_off(int key, int ch) { return (ch >> 5) * (PAGE * 32) + key * 32 + (ch & 31); }
__device__ __forceinline__ int v_elem_off(int ch, int key) { return (key >> 5) * (HEAD_DIM * 32) + ch * 32 + (key & 31); }

// Staging a K / V tile (16 KiB each) into the LDS images of the prefill kernels, 256 threads x 4
// chunks of 16 B each.  Thread `tid`, chunk i reads global chunk  *_stage_tid(tid) + *_stage_i(i)
// (in elements) and holds, for K, chunk c of key `key` (LDS image [key][16 chunks]) and, for V,
// chunk cv of channel d (LDS image [channel][8 chunks]).  Eight consecutive lanes hold the eight
// chunks of one row's 128-byte bank window (conflict-free ds_write_b128), and a wave reads two
// runs of 512 contiguous bytes.
__device__ __forceinline__ int k_stage_tid(int tid) { return (((tid >> 2) & 1) * 256 + (tid >> 3) * 4 + (tid & 3)) * 8; }
__device__ __forceinline__ int k_stage_i(int i) { return ((i >> 1) * 512 + (i & 1) * 128) * 8; }
__device__ __forceinline__ int k_stage_key(int tid, int i) { return ((i & 1) * 256 + tid) >> 3; }
__device__ __forceinline__ int k_stage_chunk(int tid, int i) { return (i >> 1) * 8 + (tid & 7); }
__device__ __forceinline__ int v_stage_tid(int tid) { return (((tid >> 2) & 1) * 512 + (tid >> 3) * 4 + (tid & 3)) * 8; }
__device__ __forceinline__ int v_stage_i(int i) { return i * 128 * 8; }
__device__ __forceinline__ int v_stage_ch(int tid, int i) { return (i * 256 + tid) >> 3; }
__device__ __forceinline__ int v_stage_chunk(int tid, int i) { return tid & 7; }
