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
      if (w == 0.f) continue;  // a pair of an expert held by another rank: its row was never written
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
      // whole lines, read once per step: streaming loads (the matrix-core kernels below read half
      // lines per step and need the cache to hold the other half: plain loads there, measured)
      wg[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (long)n * hidden) + c);
      wu[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (long)(inter + n) * hidden) + c);
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
      w[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (long)h * inter) + c);
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

// ---------------------------------------------------------------------------------
// Decode-batch expert MLP on the matrix cores: every expert's weights are streamed ONCE and
// multiplied with all tokens routed to it (the GEMV kernels above re-read an expert per pair;
// running all experts on all rows multiplies the FLOPs by E / top-k).  Works on the
// expert-sorted pair list of moe_align (offsets / sorted_tok), so there is no host sync and
// the launch shape does not depend on the routing: HIP-graph capturable.
//
// MFMA 16x16x32 bf16 with the weights as the A operand (16 weight rows x 32 reduction
// elements: one 16-byte global load per lane, consumed straight from registers) and up to
// TT tiles of 16 tokens as B operands (token rows are read through L2: they are tiny next to
// the weights).  D: lane (c16, g) holds weight rows 4g..4g+3 of token column c16, so the
// epilogue writes 4 consecutive bf16 per lane.
//   gate_up: grid (I / 64, E); wave w of a block owns gate rows i0..i0+15 and the matching
//            up rows, fuses silu(gate) * up, writes act[sorted row, i]
//   down:    grid (H / 16, E); the block's 4 waves split the reduction (I) in four slices
//            and combine through LDS, write y[sorted row, h]
// Experts without tokens return at once; an expert with more than 16 * TT tokens loops
// (its weights are then streamed once per 16 * TT tokens).
// ---------------------------------------------------------------------------------
#define MOE_KU 4  // reduction steps (of 32) whose loads are issued together

template <int TT>
__global__ __launch_bounds__(256) void moe_mfma_gate_up_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ w13,
    const int* __restrict__ offsets, const int* __restrict__ sorted_tok,
    uint16_t* __restrict__ act, int hidden, int inter) {
  const int e = blockIdx.y;
  const int r0 = offsets[e];
  const int n_e = offsets[e + 1] - r0;
  if (n_e <= 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const int i0 = (blockIdx.x * 4 + wave) * 16;
  const uint16_t* wg = w13 + ((long)e * 2 * inter + i0 + c16) * hidden + 8 * g;
  const uint16_t* wu = wg + (long)inter * hidden;
  for (int t0 = 0; t0 < n_e; t0 += 16 * TT) {
    const int nt = min(TT, (n_e - t0 + 15) >> 4);
    const uint16_t* xp[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int j = min(t0 + tt * 16 + c16, n_e - 1);  // padded columns repeat the last token
      xp[tt] = x + (long)sorted_tok[r0 + j] * hidden + 8 * g;
    }
    f32x4 accg[TT], accu[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) accg[tt] = accu[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < hidden; k += 32 * MOE_KU) {
      u32x4 ag[MOE_KU], au[MOE_KU];
#pragma unroll
      for (int u = 0; u < MOE_KU; ++u) {
        ag[u] = *reinterpret_cast<const u32x4*>(wg + k + 32 * u);
        au[u] = *reinterpret_cast<const u32x4*>(wu + k + 32 * u);
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        if (tt < nt) {
          u32x4 b[MOE_KU];
#pragma unroll
          for (int u = 0; u < MOE_KU; ++u) b[u] = *reinterpret_cast<const u32x4*>(xp[tt] + k + 32 * u);
#pragma unroll
          for (int u = 0; u < MOE_KU; ++u) {
            accg[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, ag[u]), __builtin_bit_cast(bf16x8, b[u]), accg[tt], 0, 0, 0);
            accu[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, au[u]), __builtin_bit_cast(bf16x8, b[u]), accu[tt], 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int j = t0 + tt * 16 + c16;
      if (tt < nt && j < n_e) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gt = accg[tt][r];
          v[r] = gt / (1.f + __expf(-gt)) * accu[tt][r];
        }
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(act + (long)(r0 + j) * inter + i0 + 4 * g) = o;
      }
    }
  }
}

template <int TT>
__global__ __launch_bounds__(256) void moe_mfma_down_kernel(
    const uint16_t* __restrict__ act, const uint16_t* __restrict__ w2,
    const int* __restrict__ offsets, uint16_t* __restrict__ y, int hidden, int inter) {
  __shared__ f32x4 red[4][TT][64];
  const int e = blockIdx.y;
  const int r0 = offsets[e];
  const int n_e = offsets[e + 1] - r0;
  if (n_e <= 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c16 = lane & 15, g = lane >> 4;
  const int h0 = blockIdx.x * 16;
  const int slice = inter >> 2;  // per wave; a multiple of 32 * MOE_KU (checked on the host)
  const uint16_t* wp = w2 + ((long)e * hidden + h0 + c16) * inter + wave * slice + 8 * g;
  for (int t0 = 0; t0 < n_e; t0 += 16 * TT) {
    const int nt = min(TT, (n_e - t0 + 15) >> 4);
    const uint16_t* ap[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int j = min(t0 + tt * 16 + c16, n_e - 1);
      ap[tt] = act + (long)(r0 + j) * inter + wave * slice + 8 * g;
    }
    f32x4 acc[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < slice; k += 32 * MOE_KU) {
      u32x4 a[MOE_KU];
#pragma unroll
      for (int u = 0; u < MOE_KU; ++u) a[u] = *reinterpret_cast<const u32x4*>(wp + k + 32 * u);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
        if (tt < nt) {
          u32x4 b[MOE_KU];
#pragma unroll
          for (int u = 0; u < MOE_KU; ++u) b[u] = *reinterpret_cast<const u32x4*>(ap[tt] + k + 32 * u);
#pragma unroll
          for (int u = 0; u < MOE_KU; ++u)
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, a[u]), __builtin_bit_cast(bf16x8, b[u]), acc[tt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      if (tt < nt) red[wave][tt][lane] = acc[tt];
    __syncthreads();
    // wave w finishes token tile w (TT <= 4)
    if (wave < nt) {
      const f32x4 s = red[0][wave][lane] + red[1][wave][lane] + red[2][wave][lane] + red[3][wave][lane];
      const int j = t0 + wave * 16 + c16;
      if (j < n_e) {
        u32x2 o;
        o[0] = pack_bf16x2(s[0], s[1]);
        o[1] = pack_bf16x2(s[2], s[3]);
        *reinterpret_cast<u32x2*>(y + (long)(r0 + j) * hidden + h0 + 4 * g) = o;
      }
    }
    __syncthreads();
  }
}

template <int TT>
static void launch_moe_mfma(const void* x, const void* w13, const void* w2, const int* offsets,
                            const int* sorted_tok, void* act, void* y, int n_experts, int hidden,
                            int inter, hipStream_t stream) {
  hipLaunchKernelGGL(moe_mfma_gate_up_kernel<TT>, dim3(inter / 64, n_experts), dim3(256), 0, stream,
                     (const uint16_t*)x, (const uint16_t*)w13, offsets, sorted_tok, (uint16_t*)act,
                     hidden, inter);
  hipLaunchKernelGGL(moe_mfma_down_kernel<TT>, dim3(hidden / 16, n_experts), dim3(256), 0, stream,
                     (const uint16_t*)act, (const uint16_t*)w2, offsets, (uint16_t*)y, hidden, inter);
}

// x [tokens, hidden]; w13 [E, 2*inter, hidden]; w2 [E, hidden, inter]; offsets [E+1] and
// sorted_tok [tokens*topk] from vpsa_moe_align; act [tokens*topk, inter] and
// y [tokens*topk, hidden] are workspaces in expert-sorted row order.
VPSA_EXPORT int vpsa_moe_mfma(const void* x, const void* w13, const void* w2, const int* offsets,
                              const int* sorted_tok, void* act, void* y, int tokens, int topk,
                              int n_experts, int hidden, int inter, hipStream_t stream) {
  if (tokens * topk <= 0) return 0;
  if (n_experts > MAX_EXPERTS) return -1;
  if (hidden % (32 * MOE_KU) != 0 || hidden % 16 != 0 || inter % 64 != 0 ||
      (inter / 4) % (32 * MOE_KU) != 0)
    return -1;
  // an expert sees each token at most once: tokens bounds its row count
  if (tokens <= 16)
    launch_moe_mfma<1>(x, w13, w2, offsets, sorted_tok, act, y, n_experts, hidden, inter, stream);
  else if (tokens <= 32)
    launch_moe_mfma<2>(x, w13, w2, offsets, sorted_tok, act, y, n_experts, hidden, inter, stream);
  else
    launch_moe_mfma<4>(x, w13, w2, offsets, sorted_tok, act, y, n_experts, hidden, inter, stream);
  return 0;
}
