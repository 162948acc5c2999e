// Batched stochastic sampling for gfx950: temperature, top-k, top-p and the draw itself for
// every sampled row of a decode step in ONE launch (one workgroup of 1024 threads per row),
// with no sort and no host round trip per row.
//
// The logits are bf16, so a logit has only 65536 possible values.  Mapping the bf16 bit
// pattern to an order-preserving 16-bit key turns both filters into a search for a key
// threshold:
//   top-k:  the largest key t with  count(key >= t) >= k
//   top-p:  the largest key t with  mass(key >= t)  >= p * mass(kept by top-k)
// and each search is a 16-step bisection whose step is one pass over the row (256 KB for a
// 128k vocabulary: L2-resident) with a block reduction.  All reductions run in a fixed order
// (lane shuffles, then waves in index order), so a row with the same logits, parameters and
// uniform number always yields the same token: seeded requests are reproducible.
// Tokens whose logits tie exactly with the threshold are all kept (a sort-based filter would
// cut through the tie in index order); apart from that the kept set equals
// softmax -> sort -> cumulative-sum filtering.
// The draw: thread t owns a contiguous slice of the row; an exclusive scan of the slices' kept
// mass locates the slice containing u * Z, and that thread walks its slice to the token.
#include "common.h"

#define SAMPLE_THREADS 1024

__device__ __forceinline__ uint32_t bf16_key(uint32_t b) {
  return (b & 0x8000u) ? (~b & 0xffffu) : (b | 0x8000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
  const uint32_t b = (k & 0x8000u) ? (k & 0x7fffu) : (~k & 0xffffu);
  return __uint_as_float(b << 16);
}

struct SampleRed {
  float f[SAMPLE_THREADS / 64];
  uint32_t u[SAMPLE_THREADS / 64];
};

__device__ __forceinline__ float block_sum(float v, SampleRed& r) {
  v = wave_reduce_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) r.f[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) s += r.f[w];
  return s;
}

__device__ __forceinline__ uint32_t block_sum_u(uint32_t v, SampleRed& r) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) r.u[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t s = 0;
#pragma unroll
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) s += r.u[w];
  return s;
}

__device__ __forceinline__ uint32_t block_max_u(uint32_t v, SampleRed& r) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) r.u[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t s = 0;
#pragma unroll
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) s = max(s, r.u[w]);
  return s;
}

// Visit the thread's slice [begin, end) of the row: f(key, index).  begin is a multiple of 8,
// so whole groups of 8 are 16-byte loads.
template <typename F>
__device__ __forceinline__ void for_slice(const uint16_t* __restrict__ row, int begin, int end, F f) {
  int i = begin;
  for (; i + 8 <= end; i += 8) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(row + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f(bf16_key(v[j] & 0xffffu), i + 2 * j);
      f(bf16_key(v[j] >> 16), i + 2 * j + 1);
    }
  }
  for (; i < end; ++i) f(bf16_key(row[i]), i);
}

// params [n, 3] = temperature, top_p, top_k (as float; <= 0 means off); uniform [n] in [0, 1);
// rows [n] = row of `logits` each sample reads (null: row i); out [n] token ids.
// stats (optional, [n, 3]) receives threshold key, kept mass relative to the largest
// probability, kept count: for tests.
__global__ __launch_bounds__(SAMPLE_THREADS) void sample_kernel(
    const uint16_t* __restrict__ logits, long stride, int vocab, const int* __restrict__ rows,
    const float* __restrict__ params, const float* __restrict__ uniform, long* __restrict__ out,
    float* __restrict__ stats) {
  __shared__ SampleRed red;
  __shared__ float scan_w[SAMPLE_THREADS / 64];
  __shared__ int winner;
  const int r = blockIdx.x;
  const uint16_t* row = logits + (long)(rows ? rows[r] : r) * stride;
  const float temperature = fmaxf(params[r * 3], 1e-5f);
  const float top_p = params[r * 3 + 1];
  const int top_k = (int)params[r * 3 + 2];
  const int per = (((vocab + SAMPLE_THREADS - 1) / SAMPLE_THREADS) + 7) & ~7;
  const int begin = min((int)threadIdx.x * per, vocab);
  const int end = min(begin + per, vocab);

  uint32_t kmax = 0;
  for_slice(row, begin, end, [&](uint32_t k, int) { kmax = max(kmax, k); });
  kmax = block_max_u(kmax, red);
  const float vmax = key_value(kmax);
  const float scale = 1.4426950408889634f / temperature;
  auto prob = [&](uint32_t k) { return __builtin_amdgcn_exp2f((key_value(k) - vmax) * scale); };
  auto count_ge = [&](uint32_t t) {
    uint32_t c = 0;
    for_slice(row, begin, end, [&](uint32_t k, int) { c += k >= t ? 1u : 0u; });
    return block_sum_u(c, red);
  };
  auto mass_ge = [&](uint32_t t) {
    float s = 0.f;
    for_slice(row, begin, end, [&](uint32_t k, int) { s += k >= t ? prob(k) : 0.f; });
    return block_sum(s, red);
  };

  uint32_t thr = 0;
  if (top_k > 0 && top_k < vocab) {
    uint32_t lo = 0, hi = kmax + 1;  // count_ge(lo) >= k > count_ge(hi) (= 0)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (count_ge(mid) >= (uint32_t)top_k) lo = mid;
      else hi = mid;
    }
    thr = lo;
  }
  float z = mass_ge(thr);
  if (top_p > 0.f && top_p < 1.f) {
    const float target = top_p * z;
    uint32_t lo = thr, hi = kmax + 1;  // mass_ge(lo) >= target > mass_ge(hi) (= 0)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (mass_ge(mid) >= target) lo = mid;
      else hi = mid;
    }
    thr = lo;
    z = mass_ge(thr);
  }
  if (stats != nullptr) {
    const uint32_t kept = count_ge(thr);
    if (threadIdx.x == 0) {
      stats[r * 3] = (float)thr;
      stats[r * 3 + 1] = z;
      stats[r * 3 + 2] = (float)kept;
    }
  }

  // the draw
  float mine = 0.f;
  for_slice(row, begin, end, [&](uint32_t k, int) { mine += k >= thr ? prob(k) : 0.f; });
  // exclusive scan over threads: lanes, then waves in order
  float incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float v = __shfl_up(incl, off, 64);
    if ((int)(threadIdx.x & 63) >= off) incl += v;
  }
  if ((threadIdx.x & 63) == 63) scan_w[threadIdx.x >> 6] = incl;
  if (threadIdx.x == 0) winner = -1;
  __syncthreads();
  float base = 0.f, total = 0.f;
#pragma unroll
  for (int w = 0; w < SAMPLE_THREADS / 64; ++w) {
    if (w < (int)(threadIdx.x >> 6)) base += scan_w[w];
    total += scan_w[w];
  }
  const float before = base + incl - mine;
  const float target = fminf(uniform[r], 0.99999994f) * total;
  // exactly one thread has before <= target < before + mine; rounding at the very end of the
  // row is caught by the fallback below
  if (mine > 0.f && before <= target && target < before + mine) atomicMax(&winner, (int)threadIdx.x);
  __syncthreads();
  if (winner < 0) {  // target fell into a rounding gap: the last thread with kept mass takes it
    if (mine > 0.f) atomicMax(&winner, (int)threadIdx.x);
    __syncthreads();
  }
  if ((int)threadIdx.x == winner) {
    float acc = before;
    int tok = -1, last = -1;
    for_slice(row, begin, end, [&](uint32_t k, int i) {
      if (k >= thr) {
        acc += prob(k);
        last = i;
        if (tok < 0 && acc > target) tok = i;
      }
    });
    out[r] = tok >= 0 ? tok : last;
  }
}

VPSA_EXPORT int vpsa_sample(const void* logits, long stride, int vocab, const int* rows,
                            const float* params, const float* uniform, long* out, float* stats, int n,
                            hipStream_t stream) {
  if (n <= 0) return 0;
  if (vocab <= 0 || (stride & 7)) return -1;
  hipLaunchKernelGGL(sample_kernel, dim3(n), dim3(SAMPLE_THREADS), 0, stream, (const uint16_t*)logits,
                     stride, vocab, rows, params, uniform, out, stats);
  return 0;
}
