// Single-query attention over dense (unpaged) K/V for 64-channel heads: the decoder of a
// speech model attends with ONE new query per window to its own few hundred earlier positions
// and to the 1500 encoder positions of the window (cross-attention), both held as plain
// [window][position][head * 64] tensors.
//
// The work is a GEMV pair per (window, head): 2 x T x 64 bf16 values read once, no reuse, so the
// kernel is a streaming reduction, not an MFMA kernel.  One workgroup per (head, window); eight
// lanes share a key (16 bytes of the key row each: one 128-byte line per key and group), every
// group keeps four keys in flight, and each group runs its own online softmax (running maximum,
// denominator, 8 output channels per lane).  Groups are merged at the end, inside a wave with
// cross-lane moves, between waves through 2.5 KB of LDS.  Scores, probabilities and the output
// stay in fp32 registers; nothing but Q, K, V and the 64 outputs touches memory, where the same
// step written with library calls is five launches with [window, head, T] fp32 round trips.
//
// The number of valid positions is read from device memory (position counter + 1), so a captured
// HIP graph replays the same launch for every decode step.
#include "common.h"

#define DA_KEYS 4  // keys in flight per 8-lane group

__device__ __forceinline__ void da_merge(float& m, float& l, float (&acc)[8], float m2, float l2,
                                         const float (&acc2)[8]) {
  const float mn = fmaxf(m, m2);
  const float a = __builtin_amdgcn_exp2f(m - mn), b = __builtin_amdgcn_exp2f(m2 - mn);
  l = l * a + l2 * b;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = acc[i] * a + acc2[i] * b;
  m = mn;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void dense_decode_attn_kernel(
    const uint16_t* __restrict__ q, long q_bs, const uint16_t* __restrict__ k,
    const uint16_t* __restrict__ v, long kv_bs, long kv_ts, const long* __restrict__ pos, int T,
    uint16_t* __restrict__ out, long o_bs, float scale_log2e) {
  constexpr int SLOTS = THREADS / 8, WAVES = THREADS / 64;
  __shared__ float merge[WAVES][8][10];
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int c8 = tid & 7, slot = tid >> 3;
  int n = T;
  if (pos != nullptr) {
    const long p = pos[0] + 1;
    n = p < T ? (int)p : T;
  }
  float qf[8];
  {
    const u32x4 qw = *(const u32x4*)(q + b * q_bs + h * 64 + c8 * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qf[2 * i] = bf16lo(qw[i]) * scale_log2e;
      qf[2 * i + 1] = bf16hi(qw[i]) * scale_log2e;
    }
  }
  const uint16_t* kb = k + b * kv_bs + h * 64 + c8 * 8;
  const uint16_t* vb = v + b * kv_bs + h * 64 + c8 * 8;
  float m = -1e30f, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < n; base += SLOTS * DA_KEYS) {
    u32x4 kk[DA_KEYS], vv[DA_KEYS];
    bool ok[DA_KEYS];
#pragma unroll
    for (int j = 0; j < DA_KEYS; ++j) {
      const int key = base + j * SLOTS + slot;
      ok[j] = key < n;
      const long off = (long)(ok[j] ? key : n - 1) * kv_ts;
      kk[j] = *(const u32x4*)(kb + off);
      vv[j] = *(const u32x4*)(vb + off);
    }
#pragma unroll
    for (int j = 0; j < DA_KEYS; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s = fmaf(qf[2 * i], bf16lo(kk[j][i]), s);
        s = fmaf(qf[2 * i + 1], bf16hi(kk[j][i]), s);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (ok[j]) {
        const float mn = fmaxf(m, s);
        const float a = __builtin_amdgcn_exp2f(m - mn), p = __builtin_amdgcn_exp2f(s - mn);
        l = l * a + p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] = acc[2 * i] * a + p * bf16lo(vv[j][i]);
          acc[2 * i + 1] = acc[2 * i + 1] * a + p * bf16hi(vv[j][i]);
        }
        m = mn;
      }
    }
  }
  // the 8 groups of a wave: lanes that differ in bits 3..5 hold the same channels
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
    float acc2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc2[i] = __shfl_xor(acc[i], off, 64);
    da_merge(m, l, acc, m2, l2, acc2);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 8) {
    merge[wave][lane][0] = m;
    merge[wave][lane][1] = l;
#pragma unroll
    for (int i = 0; i < 8; ++i) merge[wave][lane][2 + i] = acc[i];
  }
  __syncthreads();
  if (tid < 8) {
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
      float acc2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc2[i] = merge[w][tid][2 + i];
      da_merge(m, l, acc, merge[w][tid][0], merge[w][tid][1], acc2);
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(acc[2 * i] * inv, acc[2 * i + 1] * inv);
    *(u32x4*)(out + b * o_bs + h * 64 + tid * 8) = o;
  }
}

// q [B][heads * 64] (row stride q_bs elements), k / v [B][T][heads * 64] (strides kv_bs, kv_ts),
// out [B][heads * 64] (row stride o_bs); every row start 16-byte aligned.  pos: device pointer to
// the index of the newest valid position (positions 0..pos[0] are attended), or null for all T.
// Returns -1 for arguments the kernel does not cover.
VPSA_EXPORT int vpsa_dense_decode_attention(const void* q, long q_bs, const void* k, const void* v,
                                            long kv_bs, long kv_ts, const void* pos, int B, int heads,
                                            int T, void* out, long o_bs, float scale,
                                            hipStream_t stream) {
  if (B <= 0 || heads <= 0) return 0;
  if (T <= 0 || B > 65535 || (q_bs & 7) || (kv_bs & 7) || (kv_ts & 7) || (o_bs & 7)) return -1;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return -1;
  const float sl = scale * 1.4426950408889634f;
  hipLaunchKernelGGL((dense_decode_attn_kernel<512>), dim3(heads, B), dim3(512), 0, stream,
                     (const uint16_t*)q, q_bs, (const uint16_t*)k, (const uint16_t*)v, kv_bs, kv_ts,
                     (const long*)pos, T, (uint16_t*)out, o_bs, sl);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
