// Batched multi-adapter LoRA for gfx950: every row of the batch may use a different adapter
// (or none), one launch covers them all.
//
//   shrink:  h[t, :]  = A[slot(t)] . x[t]            A: [slots, R, in]   (R = stacked ranks)
//   expand:  y[t, off + o] += B[slot(t)][o, :] . h[t, hoff : hoff + r_pad]
//                                                     B: [slots, out, r_pad]  (scale folded in)
//
// Rows are processed in tiles of 16 tokens on MFMA 16x16x32 bf16 with the adapter matrix as
// the A operand and the token tile as the B operand.  A tile usually belongs to one sequence
// and therefore one adapter; a mixed tile (decode batches) is handled by looping over the
// adapters present in it with the other rows' operands zeroed.  Rows with slot < 0 are left
// untouched.  No atomics: every output element is owned by exactly one lane.
#include "common.h"

#define LORA_MAX_SLOTS 64

// grid (ceil(T / 16), R / 16), block 256: the four waves split the reduction dimension.
// Decode batches have a handful of token tiles, so the rank blocks are spread over the grid
// (more workgroups in flight) instead of looped over inside one.
__global__ __launch_bounds__(256) void lora_shrink_kernel(
    const uint16_t* __restrict__ x, long x_stride, const uint16_t* __restrict__ a,
    const int* __restrict__ slots, uint16_t* __restrict__ h, int tokens, int in_dim, int rank_total,
    int n_slots) {
  __shared__ f32x4 red[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const int t = blockIdx.x * 16 + c16;
  const int my_slot = t < tokens ? slots[t] : -1;
  const int r0 = blockIdx.y * 16;
  const int slice = in_dim >> 2;
  const uint16_t* xp = x + (long)min(t, tokens - 1) * x_stride + wave * slice + 8 * g;
  for (int s = 0; s < n_slots; ++s) {
    if (!__any(my_slot == s)) continue;  // same answer in all four waves: they see the same tile
    const bool mine = my_slot == s;
    const uint16_t* ap = a + ((long)s * rank_total + r0 + c16) * in_dim + wave * slice + 8 * g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < slice; k += 128) {
      u32x4 av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = *reinterpret_cast<const u32x4*>(ap + k + 32 * u);
        bv[u] = *reinterpret_cast<const u32x4*>(xp + k + 32 * u);
        if (!mine) bv[u] = (u32x4){0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[u]),
                                                      __builtin_bit_cast(bf16x8, bv[u]), acc, 0, 0, 0);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && mine) {
      const f32x4 v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
      u32x2 o;
      o[0] = pack_bf16x2(v[0], v[1]);
      o[1] = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<u32x2*>(h + (long)t * rank_total + r0 + 4 * g) = o;
    }
    __syncthreads();
  }
}

// One launch covers the sub-modules of a fused projection (q | k | v, gate | up).
struct LoraSubs {
  const uint16_t* b[3];  // [slots, out, r_pad]
  int out_dim[3];
  int y_off[3];
  int h_off[3];
  int blk_end[3];  // exclusive prefix of 256-column blocks over the sub-modules
  int n;
};

// grid (ceil(T / 16), sum_i out_i / 256), block 256: wave w owns 64 consecutive output
// columns.  The 16 A rows of MFMA j are the columns 16 * (m / 4) + 4 * j + (m % 4) of that
// span, so that after the four MFMAs lane (token c16, group g) holds the 16 consecutive
// columns 16 g .. 16 g + 15: the read-modify-write of y is two 16-byte accesses per lane and
// one full 128-byte line per token row and wave.
template <int KSTEPS>
__global__ __launch_bounds__(256) void lora_expand_kernel(
    const uint16_t* __restrict__ h, int h_stride, LoraSubs subs, const int* __restrict__ slots,
    uint16_t* __restrict__ y, long y_stride, int tokens, int n_slots) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const int t = blockIdx.x * 16 + c16;
  const int my_slot = t < tokens ? slots[t] : -1;
  const int r_pad = 32 * KSTEPS;
  int sub = 0, blk = blockIdx.y;
  if (subs.n > 1 && blk >= subs.blk_end[0]) { sub = 1; blk = blockIdx.y - subs.blk_end[0]; }
  if (subs.n > 2 && blockIdx.y >= subs.blk_end[1]) { sub = 2; blk = blockIdx.y - subs.blk_end[1]; }
  const int out_dim = subs.out_dim[sub];
  const int o0 = blk * 256 + 64 * wave;
  if (o0 >= out_dim) return;  // out_dim is a multiple of 64: a wave's span is all in or all out
  const uint16_t* hp = h + (long)min(t, tokens - 1) * h_stride + subs.h_off[sub] + 8 * g;
  for (int s = 0; s < n_slots; ++s) {
    if (!__any(my_slot == s)) continue;
    const bool mine = my_slot == s;
    u32x4 hv[KSTEPS];
#pragma unroll
    for (int u = 0; u < KSTEPS; ++u) {
      hv[u] = *reinterpret_cast<const u32x4*>(hp + 32 * u);
      if (!mine) hv[u] = (u32x4){0u, 0u, 0u, 0u};
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = o0 + 16 * (c16 >> 2) + 4 * j + (c16 & 3);
      const uint16_t* bp = subs.b[sub] + ((long)s * out_dim + col) * r_pad + 8 * g;
      acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < KSTEPS; ++u) {
        const u32x4 bv = *reinterpret_cast<const u32x4*>(bp + 32 * u);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bv),
                                                         __builtin_bit_cast(bf16x8, hv[u]), acc[j], 0, 0, 0);
      }
    }
    if (mine) {
      uint16_t* yp = y + (long)t * y_stride + subs.y_off[sub] + o0 + 16 * g;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const u32x4 old = *reinterpret_cast<const u32x4*>(yp + 8 * half);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 v = acc[2 * half + q];
          o[2 * q] = pack_bf16x2(bf16lo(old[2 * q]) + v[0], bf16hi(old[2 * q]) + v[1]);
          o[2 * q + 1] = pack_bf16x2(bf16lo(old[2 * q + 1]) + v[2], bf16hi(old[2 * q + 1]) + v[3]);
        }
        *reinterpret_cast<u32x4*>(yp + 8 * half) = o;
      }
    }
  }
}

VPSA_EXPORT int vpsa_lora_shrink(const void* x, long x_stride, const void* a, const int* slots,
                                 void* h, int tokens, int in_dim, int rank_total, int n_slots,
                                 hipStream_t stream) {
  if (tokens <= 0) return 0;
  if (in_dim % 512 != 0 || rank_total % 16 != 0 || n_slots > LORA_MAX_SLOTS || x_stride % 8 != 0)
    return -1;
  hipLaunchKernelGGL(lora_shrink_kernel, dim3((tokens + 15) / 16, rank_total / 16), dim3(256), 0,
                     stream, (const uint16_t*)x, x_stride, (const uint16_t*)a, slots, (uint16_t*)h,
                     tokens, in_dim, rank_total, n_slots);
  return 0;
}

// b / out_dim / y_off / h_off: n_sub (1..3) entries each
VPSA_EXPORT int vpsa_lora_expand(const void* h, int h_stride, const void* const* b, const int* out_dim,
                                 const int* y_off, const int* h_off, int n_sub, const int* slots,
                                 void* y, long y_stride, int tokens, int r_pad, int n_slots,
                                 hipStream_t stream) {
  if (tokens <= 0) return 0;
  if (n_sub < 1 || n_sub > 3 || (r_pad != 32 && r_pad != 64) || n_slots > LORA_MAX_SLOTS ||
      h_stride % 8 != 0 || y_stride % 8 != 0)
    return -1;
  LoraSubs subs;
  int blocks = 0;
  for (int i = 0; i < 3; ++i) {
    const int j = i < n_sub ? i : 0;
    if (out_dim[j] % 64 != 0 || y_off[j] % 8 != 0 || h_off[j] % 8 != 0) return -1;
    subs.b[i] = (const uint16_t*)b[j];
    subs.out_dim[i] = out_dim[j];
    subs.y_off[i] = y_off[j];
    subs.h_off[i] = h_off[j];
    if (i < n_sub) blocks += (out_dim[j] + 255) / 256;
    subs.blk_end[i] = blocks;
  }
  subs.n = n_sub;
  const dim3 grid((tokens + 15) / 16, blocks);
  if (r_pad == 32)
    hipLaunchKernelGGL(lora_expand_kernel<1>, grid, dim3(256), 0, stream, (const uint16_t*)h, h_stride,
                       subs, slots, (uint16_t*)y, y_stride, tokens, n_slots);
  else
    hipLaunchKernelGGL(lora_expand_kernel<2>, grid, dim3(256), 0, stream, (const uint16_t*)h, h_stride,
                       subs, slots, (uint16_t*)y, y_stride, tokens, n_slots);
  return 0;
}
