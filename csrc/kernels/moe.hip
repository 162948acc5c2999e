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
