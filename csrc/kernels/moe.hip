// Mixture-of-experts routing kernels for gfx950 (Mixtral-style top-k softmax routing).
//   1. moe_topk_softmax: router logits -> top-k expert ids + renormalised weights
//   2. moe_align:        counting sort of the (token, k) pairs by expert -> per-expert
//                        contiguous row ranges for the expert GEMMs
//   3. moe_permute:      gather activation rows into expert-sorted order
//   4. moe_combine:      out[t] = sum_k w[t,k] * y[pos[t,k]]  (gather form: no atomics, so
//                        the result is bitwise reproducible)
// The expert GEMMs themselves run on the matrix cores through the library GEMM on each
// expert's contiguous row range.
#include "common.h"

#define MAX_EXPERTS 64
#define MAX_TOPK 8

__global__ __launch_bounds__(256) void moe_topk_softmax_kernel(const uint16_t* __restrict__ logits,
                                                               float* __restrict__ weights,
                                                               int* __restrict__ ids, int tokens,
                                                               int n_experts, int topk,
                                                               int renorm) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= tokens) return;
  float v[MAX_EXPERTS];
  float mx = -INFINITY;
  for (int e = 0; e < n_experts; ++e) {
    v[e] = bf16_bits_to_f32(logits[(long)t * n_experts + e]);
    mx = fmaxf(mx, v[e]);
  }
  float den = 0.f;
  for (int e = 0; e < n_experts; ++e) {
    v[e] = __expf(v[e] - mx);
    den += v[e];
  }
  float wsum = 0.f;
  float wk[MAX_TOPK];
  int ik[MAX_TOPK];
  for (int k = 0; k < topk; ++k) {
    float best = -1.f;
    int bi = 0;
    for (int e = 0; e < n_experts; ++e)
      if (v[e] > best) { best = v[e]; bi = e; }
    v[bi] = -2.f;
    wk[k] = best / den;
    ik[k] = bi;
    wsum += wk[k];
  }
  for (int k = 0; k < topk; ++k) {
    weights[(long)t * topk + k] = renorm ? wk[k] / wsum : wk[k];
    ids[(long)t * topk + k] = ik[k];
  }
}

// Single-workgroup counting sort.  offsets has n_experts + 1 entries; pos[i] is the sorted
// row of pair i = t*topk + k; sorted_tok[row] = token of that row.  Pairs of one expert keep
// their token order (stable), so results do not depend on scheduling.
__global__ __launch_bounds__(1024) void moe_align_kernel(const int* __restrict__ ids,
                                                         int* __restrict__ offsets,
                                                         int* __restrict__ pos,
                                                         int* __restrict__ sorted_tok, int n_pairs,
                                                         int n_experts, int topk) {
  __shared__ int cnt[MAX_EXPERTS + 1];
  __shared__ int base[MAX_EXPERTS + 1];
  const int tid = threadIdx.x;
  if (tid <= n_experts) cnt[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n_pairs; i += 1024) atomicAdd(&cnt[ids[i]], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int e = 0; e < n_experts; ++e) {
      base[e] = run;
      offsets[e] = run;
      run += cnt[e];
    }
    offsets[n_experts] = run;
  }
  __syncthreads();
  // stable placement: expert e is handled by wave (e % 16); the wave walks the pair list in
  // order and ranks its matches with a ballot prefix count
  const int wave = tid >> 6, lane = tid & 63;
  for (int e = wave; e < n_experts; e += 16) {
    int run = base[e];
    for (int i0 = 0; i0 < n_pairs; i0 += 64) {
      const int i = i0 + lane;
      const bool hit = i < n_pairs && ids[i] == e;
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        pos[i] = run + rank;
        sorted_tok[run + rank] = i / topk;
      }
      run += __popcll(m);
    }
  }
}

__global__ __launch_bounds__(256) void moe_permute_kernel(const u32x4* __restrict__ x,
                                                          const int* __restrict__ sorted_tok,
                                                          u32x4* __restrict__ xs, int vec_per_row) {
  const int row = blockIdx.x;
  const int tok = sorted_tok[row];
  for (int i = threadIdx.x; i < vec_per_row; i += 256)
    xs[(long)row * vec_per_row + i] = x[(long)tok * vec_per_row + i];
}

__global__ __launch_bounds__(256) void moe_combine_kernel(const uint16_t* __restrict__ y,
                                                          const float* __restrict__ weights,
                                                          const int* __restrict__ pos,
                                                          uint16_t* __restrict__ out, int hidden,
                                                          int topk) {
  const int t = blockIdx.x;
  for (int i = threadIdx.x * 8; i < hidden; i += 256 * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < topk; ++k) {
      const float w = weights[(long)t * topk + k];
      const long row = pos[(long)t * topk + k];
      const u32x4 a = *reinterpret_cast<const u32x4*>(y + row * hidden + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += w * bf16lo(a[j]);
        acc[2 * j + 1] += w * bf16hi(a[j]);
      }
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(acc[2 * j], acc[2 * j + 1]);
    *reinterpret_cast<u32x4*>(out + (long)t * hidden + i) = o;
  }
}

VPSA_EXPORT int vpsa_moe_topk_softmax(const void* logits, float* weights, int* ids, int tokens,
                                      int n_experts, int topk, int renorm, hipStream_t stream) {
  if (tokens <= 0) return 0;
  if (n_experts > MAX_EXPERTS || topk > MAX_TOPK || topk > n_experts) return -1;
  hipLaunchKernelGGL(moe_topk_softmax_kernel, dim3((tokens + 255) / 256), dim3(256), 0, stream,
                     (const uint16_t*)logits, weights, ids, tokens, n_experts, topk, renorm);
  return 0;
}

VPSA_EXPORT int vpsa_moe_align(const int* ids, int* offsets, int* pos, int* sorted_tok,
                               int n_pairs, int n_experts, int topk, hipStream_t stream) {
  if (n_experts > MAX_EXPERTS) return -1;
  hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(1024), 0, stream, ids, offsets, pos,
                     sorted_tok, n_pairs, n_experts, topk);
  return 0;
}

VPSA_EXPORT int vpsa_moe_permute(const void* x, const int* sorted_tok, void* xs, int rows,
                                 int hidden, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (hidden % 8 != 0) return -1;
  hipLaunchKernelGGL(moe_permute_kernel, dim3(rows), dim3(256), 0, stream, (const u32x4*)x,
                     sorted_tok, (u32x4*)xs, hidden / 8);
  return 0;
}

VPSA_EXPORT int vpsa_moe_combine(const void* y, const float* weights, const int* pos, void* out,
                                 int tokens, int hidden, int topk, hipStream_t stream) {
  if (tokens <= 0) return 0;
  if (hidden % 8 != 0) return -1;
  hipLaunchKernelGGL(moe_combine_kernel, dim3(tokens), dim3(256), 0, stream, (const uint16_t*)y,
                     weights, pos, (uint16_t*)out, hidden, topk);
  return 0;
}

// ---------------------------------------------------------------------------------
// Small-batch expert MLP (decode): only the experts a token was routed to are read.
// Pair p = t * topk + k uses expert e = ids[p] on token t.  Both kernels are weight-streaming
// GEMVs: a wave owns ROWS output rows, every lane walks the reduction dimension in 16-byte
// chunks (ROWS (x2 for gate/up) independent 16-byte loads in flight per step), partial sums
// are combined with wave shuffles.  No host sync and no data-dependent launch shape, so the
// step stays capturable in a HIP graph.
//   gate_up: act[p, n] = silu(W13[e][n, :] . x[t]) * (W13[e][I + n, :] . x[t])     n < I
//   down:    y[p, h]   = W2[e][h, :] . act[p, :]                                    h < H
// grid = (ceil(rows / (4 * ROWS)), pairs), block = 256.
// ---------------------------------------------------------------------------------
#define GEMV_ROWS 4

__device__ __forceinline__ float dot8(u32x4 w, u32x4 a) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s += bf16lo(w[j]) * bf16lo(a[j]);
    s += bf16hi(w[j]) * bf16hi(a[j]);
  }
  return s;
}

__global__ __launch_bounds__(256) void moe_gemv_gate_up_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ w13, const int* __restrict__ ids,
    uint16_t* __restrict__ act, int hidden, int inter, int topk) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int p = blockIdx.y;
  const int t = p / topk;
  const int e = ids[p];
  const int n0 = (blockIdx.x * 4 + wave) * GEMV_ROWS;
  if (n0 >= inter) return;
  const int chunks = hidden >> 3;
  const u32x4* xv = reinterpret_cast<const u32x4*>(x + (long)t * hidden);
  const uint16_t* wbase = w13 + (long)e * 2 * inter * hidden;
  float g[GEMV_ROWS], u[GEMV_ROWS];
#pragma unroll
  for (int r = 0; r < GEMV_ROWS; ++r) g[r] = u[r] = 0.f;
  for (int c = lane; c < chunks; c += 64) {
    const u32x4 a = xv[c];
    u32x4 wg[GEMV_ROWS], wu[GEMV_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) {
      const int n = n0 + r < inter ? n0 + r : inter - 1;
      wg[r] = reinterpret_cast<const u32x4*>(wbase + (long)n * hidden)[c];
      wu[r] = reinterpret_cast<const u32x4*>(wbase + (long)(inter + n) * hidden)[c];
    }
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) {
      g[r] += dot8(wg[r], a);
      u[r] += dot8(wu[r], a);
    }
  }
#pragma unroll
  for (int r = 0; r < GEMV_ROWS; ++r) {
    g[r] = wave_reduce_sum(g[r]);
    u[r] = wave_reduce_sum(u[r]);
  }
  if (lane < GEMV_ROWS && n0 + lane < inter) {
    float gv = 0.f, uv = 0.f;
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r)
      if (r == lane) { gv = g[r]; uv = u[r]; }
    // the library path rounds gate/up to bf16 before the activation: do the same
    gv = bf16_bits_to_f32(__builtin_bit_cast(uint16_t, (bf16_t)gv));
    uv = bf16_bits_to_f32(__builtin_bit_cast(uint16_t, (bf16_t)uv));
    const float s = gv / (1.f + __expf(-gv)) * uv;
    act[(long)p * inter + n0 + lane] = __builtin_bit_cast(uint16_t, (bf16_t)s);
  }
}

__global__ __launch_bounds__(256) void moe_gemv_down_kernel(
    const uint16_t* __restrict__ act, const uint16_t* __restrict__ w2, const int* __restrict__ ids,
    uint16_t* __restrict__ y, int hidden, int inter) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int p = blockIdx.y;
  const int e = ids[p];
  const int h0 = (blockIdx.x * 4 + wave) * GEMV_ROWS;
  if (h0 >= hidden) return;
  const int chunks = inter >> 3;
  const u32x4* av = reinterpret_cast<const u32x4*>(act + (long)p * inter);
  const uint16_t* wbase = w2 + (long)e * hidden * inter;
  float acc[GEMV_ROWS];
#pragma unroll
  for (int r = 0; r < GEMV_ROWS; ++r) acc[r] = 0.f;
  for (int c = lane; c < chunks; c += 64) {
    const u32x4 a = av[c];
    u32x4 w[GEMV_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) {
      const int h = h0 + r < hidden ? h0 + r : hidden - 1;
      w[r] = reinterpret_cast<const u32x4*>(wbase + (long)h * inter)[c];
    }
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) acc[r] += dot8(w[r], a);
  }
#pragma unroll
  for (int r = 0; r < GEMV_ROWS; ++r) acc[r] = wave_reduce_sum(acc[r]);
  if (lane < GEMV_ROWS && h0 + lane < hidden) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r)
      if (r == lane) v = acc[r];
    y[(long)p * hidden + h0 + lane] = __builtin_bit_cast(uint16_t, (bf16_t)v);
  }
}

// x [tokens, hidden], w13 [E, 2*inter, hidden], w2 [E, hidden, inter], ids [tokens*topk]
// act [tokens*topk, inter] and y [tokens*topk, hidden] are caller-provided workspaces.
VPSA_EXPORT int vpsa_moe_gemv(const void* x, const void* w13, const void* w2, const int* ids,
                              void* act, void* y, int tokens, int topk, int hidden, int inter,
                              hipStream_t stream) {
  const int pairs = tokens * topk;
  if (pairs <= 0) return 0;
  if (hidden % 8 != 0 || inter % 8 != 0) return -1;
  const int per_block = 4 * GEMV_ROWS;
  hipLaunchKernelGGL(moe_gemv_gate_up_kernel, dim3((inter + per_block - 1) / per_block, pairs),
                     dim3(256), 0, stream, (const uint16_t*)x, (const uint16_t*)w13, ids,
                     (uint16_t*)act, hidden, inter, topk);
  hipLaunchKernelGGL(moe_gemv_down_kernel, dim3((hidden + per_block - 1) / per_block, pairs),
                     dim3(256), 0, stream, (const uint16_t*)act, (const uint16_t*)w2, ids,
                     (uint16_t*)y, hidden, inter);
  return 0;
}
