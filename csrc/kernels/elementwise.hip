// Memory-bound per-token kernels of the decoder layer for gfx950:
//   rmsnorm (+ fused residual add), rotary embedding fused with the paged KV-cache
//   write, SwiGLU activation, greedy argmax sampling.
// Each is a single pass over its operand with 16-byte (8 x bf16) accesses per lane and
// fp32 arithmetic; none has cross-block reuse, so the grid is simply one block per
// token (or a capped grid-stride for the flat elementwise case).
#include "common.h"

// ---------------------------------------------------------------------------------
// RMSNorm:  y = x * rsqrt(mean(x^2) + eps) * w
// With `residual` != nullptr:  x' = x + residual;  residual <- x';  y = norm(x') * w.
// One 256-thread block per token row; a thread keeps up to 4 chunks of 8 values in
// registers (hidden <= 8192), so the row is read exactly once.
// ---------------------------------------------------------------------------------
template <int CHUNKS>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const uint16_t* __restrict__ x,
                                                      uint16_t* __restrict__ residual,
                                                      const uint16_t* __restrict__ w,
                                                      uint16_t* __restrict__ y, int hidden,
                                                      long x_stride, long y_stride, float eps) {
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const uint16_t* xr = x + (long)row * x_stride;
  uint16_t* rr = residual ? residual + (long)row * hidden : nullptr;
  float v[CHUNKS][8];
  u32x4 ww[CHUNKS];  // fetched with the row: a load issued after the reduction would add its latency
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int i = (c * 256 + tid) * 8;
    if (i < hidden) {
      u32x4 a = *reinterpret_cast<const u32x4*>(xr + i);
      ww[c] = *reinterpret_cast<const u32x4*>(w + i);
      float t[8] = {bf16lo(a[0]), bf16hi(a[0]), bf16lo(a[1]), bf16hi(a[1]),
                    bf16lo(a[2]), bf16hi(a[2]), bf16lo(a[3]), bf16hi(a[3])};
      if (rr) {
        u32x4 b = *reinterpret_cast<const u32x4*>(rr + i);
        float u[8] = {bf16lo(b[0]), bf16hi(b[0]), bf16lo(b[1]), bf16hi(b[1]),
                      bf16lo(b[2]), bf16hi(b[2]), bf16lo(b[3]), bf16hi(b[3])};
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          t[2 * k] += u[2 * k];
          t[2 * k + 1] += u[2 * k + 1];
          o[k] = pack_bf16x2(t[2 * k], t[2 * k + 1]);
        }
        *reinterpret_cast<u32x4*>(rr + i) = o;
        // the normalised value is computed from the rounded residual stream so that the
        // result matches a reference that reads the bf16 residual back
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          t[2 * k] = bf16lo(o[k]);
          t[2 * k + 1] = bf16hi(o[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[c][k] = t[k];
        ss += t[k] * t[k];
      }
    }
  }
  __shared__ float red[4];
  ss = wave_reduce_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float inv = rsqrtf(tot / (float)hidden + eps);
  uint16_t* yr = y + (long)row * y_stride;
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int i = (c * 256 + tid) * 8;
    if (i < hidden) {
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = pack_bf16x2(v[c][2 * k] * inv * bf16lo(ww[c][k]), v[c][2 * k + 1] * inv * bf16hi(ww[c][k]));
      }
      *reinterpret_cast<u32x4*>(yr + i) = o;
    }
  }
}

VPSA_EXPORT int vpsa_rmsnorm(const void* x, void* residual, const void* w, void* y, int tokens,
                             int hidden, long x_stride, long y_stride, float eps,
                             hipStream_t stream) {
  if (tokens <= 0) return 0;
  if (hidden % 8 != 0 || hidden > 8192) return -1;
  const int chunks = (hidden + 2047) / 2048;
  dim3 grid(tokens), block(256);
#define LAUNCH(C)                                                                             \
  hipLaunchKernelGGL(rmsnorm_kernel<C>, grid, block, 0, stream, (const uint16_t*)x,           \
                     (uint16_t*)residual, (const uint16_t*)w, (uint16_t*)y, hidden, x_stride, \
                     y_stride, eps)
  switch (chunks) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    default: LAUNCH(4); break;
  }
#undef LAUNCH
  return 0;
}

// ---------------------------------------------------------------------------------
// Rotary embedding (NeoX pairing: channel i rotates with i + 64) applied in place to the
// Q and K heads of a fused QKV activation, fused with the paged KV-cache write.
//   qkv        [tokens, (n_q + 2 n_kv) * 128] bf16, row stride qkv_stride
//   positions  [tokens] int32
//   slots      [tokens] int64   global slot = page * 64 + offset; < 0 skips the cache write
//   cos_sin    [max_pos, 128] fp32: cos[0..63] then sin[0..63]
//   cache      this layer's base pointer; page p starts at cache + p * page_stride;
//              K part at +0, V part at +v_offset (elements)
// One block per token; 32 lanes per head (each lane owns channels 2j,2j+1 and 64+2j,65+2j).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_cache_kernel(
    uint16_t* __restrict__ qkv, const int* __restrict__ positions, const long* __restrict__ slots,
    const float* __restrict__ cos_sin, uint16_t* __restrict__ cache, long qkv_stride, int n_q,
    int n_kv, long page_stride, long v_offset, int rotary) {
  const int tok = blockIdx.x;
  const int tid = threadIdx.x;
  const int j = tid & 31;
  const int hs = tid >> 5;
  const int pos = positions[tok];
  const long slot = slots ? slots[tok] : -1;
  uint16_t* row = qkv + (long)tok * qkv_stride;
  const float2 c = *reinterpret_cast<const float2*>(cos_sin + (long)pos * HEAD_DIM + 2 * j);
  const float2 s = *reinterpret_cast<const float2*>(cos_sin + (long)pos * HEAD_DIM + 64 + 2 * j);
  const int total = n_q + 2 * n_kv;
  const long page = slot >= 0 ? slot / PAGE : 0;
  const int off = slot >= 0 ? (int)(slot % PAGE) : 0;
  uint16_t* kpage = cache + page * page_stride;
  uint16_t* vpage = kpage + v_offset;
  for (int h = hs; h < total; h += 8) {
    uint32_t* p = reinterpret_cast<uint32_t*>(row + h * HEAD_DIM);
    uint32_t lo = p[j], hi = p[32 + j];
    if (h < n_q + n_kv) {
      if (rotary) {
        const float x0 = bf16lo(lo), x1 = bf16hi(lo), y0 = bf16lo(hi), y1 = bf16hi(hi);
        lo = pack_bf16x2(x0 * c.x - y0 * s.x, x1 * c.y - y1 * s.y);
        hi = pack_bf16x2(y0 * c.x + x0 * s.x, y1 * c.y + x1 * s.y);
        p[j] = lo;
        p[32 + j] = hi;
      }
      if (h >= n_q && slot >= 0) {
        const int kvh = h - n_q;
        // channels 2j, 2j+1 and 64 + 2j, 65 + 2j of key `off` (K part: [channel block][key][32])
        uint16_t* kd = kpage + (long)kvh * PAGE * HEAD_DIM;
        *reinterpret_cast<uint32_t*>(kd + k_elem_off(off, 2 * j)) = lo;
        *reinterpret_cast<uint32_t*>(kd + k_elem_off(off, 64 + 2 * j)) = hi;
      }
    } else if (slot >= 0) {
      const int kvh = h - n_q - n_kv;
      uint16_t* vd = vpage + (long)kvh * HEAD_DIM * PAGE;  // V part: [key half][channel][32 keys]
      vd[v_elem_off(2 * j, off)] = (uint16_t)(lo & 0xffff);
      vd[v_elem_off(2 * j + 1, off)] = (uint16_t)(lo >> 16);
      vd[v_elem_off(64 + 2 * j, off)] = (uint16_t)(hi & 0xffff);
      vd[v_elem_off(65 + 2 * j, off)] = (uint16_t)(hi >> 16);
    }
  }
}

VPSA_EXPORT int vpsa_rope_cache(void* qkv, const int* positions, const long* slots,
                                const float* cos_sin, void* cache, int tokens, long qkv_stride,
                                int n_q, int n_kv, long page_stride, long v_offset, int rotary,
                                hipStream_t stream) {
  if (tokens <= 0) return 0;
  hipLaunchKernelGGL(rope_cache_kernel, dim3(tokens), dim3(256), 0, stream, (uint16_t*)qkv,
                     positions, slots, cos_sin, (uint16_t*)cache, qkv_stride, n_q, n_kv,
                     page_stride, v_offset, rotary);
  return 0;
}

// ---------------------------------------------------------------------------------
// K/V of prefill rows -> paged cache, 32 consecutive rows of one KV head per workgroup.  The
// query heads are not touched (the prefill attention kernel rotates them as it loads them) and
// the rotated K goes to the cache only, not back into qkv.
//
// The per-token kernel above writes a token's V as 128 two-byte stores 64 bytes apart (the V
// page keeps 32 keys of a channel together): 0.75 TB/s at 8192 rows.  Here V goes through a
// transposing LDS tile so that a store instruction covers 32 consecutive keys of two channels:
// 128 contiguous bytes when the rows' slots are consecutive, which they are within a sequence.
// K: one thread per (row, 16-byte chunk c and its rotation partner c + 64 channels); the eight
// chunks of a row and the rows of a tile land in contiguous runs of the [channel block][key][32]
// page layout.
// ---------------------------------------------------------------------------------
#define KVT_ROWS 32
#define KVT_LD 130  // LDS row stride in elements: 65 dwords, so 32 rows of one channel hit 32 banks

__global__ __launch_bounds__(256) void kv_cache_tile_kernel(
    const uint16_t* __restrict__ qkv, const int* __restrict__ positions, const long* __restrict__ slots,
    const float* __restrict__ cos_sin, uint16_t* __restrict__ cache, long qkv_stride, int n_q,
    int n_kv, long page_stride, long v_offset, int rotary, int n_rows) {
  __shared__ uint16_t vt[KVT_ROWS * KVT_LD];
  const int tid = threadIdx.x, kvh = blockIdx.y;
  const int t0 = blockIdx.x * KVT_ROWS;
  {
    const int tok = t0 + (tid >> 3), c8 = tid & 7;
    const long slot = tok < n_rows ? slots[tok] : -1;
    if (slot >= 0) {
      const uint16_t* src = qkv + (long)tok * qkv_stride + (n_q + kvh) * HEAD_DIM + 8 * c8;
      u32x4 lo = *reinterpret_cast<const u32x4*>(src);
      u32x4 hi = *reinterpret_cast<const u32x4*>(src + 64);
      if (rotary) {
        const float* cs = cos_sin + (long)positions[tok] * HEAD_DIM + 8 * c8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(cs + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(cs + 64);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(cs + 68);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float ca = w < 2 ? c0[2 * w] : c1[2 * w - 4], cb = w < 2 ? c0[2 * w + 1] : c1[2 * w - 3];
          const float sa = w < 2 ? s0[2 * w] : s1[2 * w - 4], sb = w < 2 ? s0[2 * w + 1] : s1[2 * w - 3];
          const float x0 = bf16lo(lo[w]), x1 = bf16hi(lo[w]), y0 = bf16lo(hi[w]), y1 = bf16hi(hi[w]);
          lo[w] = pack_bf16x2(x0 * ca - y0 * sa, x1 * cb - y1 * sb);
          hi[w] = pack_bf16x2(y0 * ca + x0 * sa, y1 * cb + x1 * sb);
        }
      }
      const long page = slot / PAGE;
      const int off = (int)(slot % PAGE);
      uint16_t* kd = cache + page * page_stride + (long)kvh * PAGE * HEAD_DIM;
      *reinterpret_cast<u32x4*>(kd + k_elem_off(off, 8 * c8)) = lo;
      *reinterpret_cast<u32x4*>(kd + k_elem_off(off, 64 + 8 * c8)) = hi;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lt = (tid >> 4) + 16 * i, c = tid & 15;
    const int tok = t0 + lt;
    u32x4 v = (u32x4){0, 0, 0, 0};
    if (tok < n_rows)
      v = *reinterpret_cast<const u32x4*>(qkv + (long)tok * qkv_stride + (n_q + n_kv + kvh) * HEAD_DIM + 8 * c);
    uint32_t* d = reinterpret_cast<uint32_t*>(vt + lt * KVT_LD + 8 * c);
    d[0] = v[0];
    d[1] = v[1];
    d[2] = v[2];
    d[3] = v[3];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, lt = lane & 31;
  const int tok = t0 + lt;
  const long slot = tok < n_rows ? slots[tok] : -1;
  if (slot >= 0) {
    const long page = slot / PAGE;
    const int off = (int)(slot % PAGE);
    uint16_t* vd = cache + page * page_stride + v_offset + (long)kvh * HEAD_DIM * PAGE;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int ch = it * 8 + wave * 2 + (lane >> 5);
      vd[v_elem_off(ch, off)] = vt[lt * KVT_LD + ch];
    }
  }
}

// Rows [0, n_rows) of qkv (already offset to the first prefill row, as are positions and slots):
// K heads rotated and written to the cache, V heads written to the cache; qkv is left unchanged.
VPSA_EXPORT int vpsa_kv_cache_write(const void* qkv, const int* positions, const long* slots,
                                    const float* cos_sin, void* cache, int n_rows, long qkv_stride,
                                    int n_q, int n_kv, long page_stride, long v_offset, int rotary,
                                    hipStream_t stream) {
  if (n_rows <= 0) return 0;
  if ((qkv_stride & 7) || ((uintptr_t)qkv & 15) || n_kv > 65535) return -1;
  hipLaunchKernelGGL(kv_cache_tile_kernel, dim3((n_rows + KVT_ROWS - 1) / KVT_ROWS, n_kv), dim3(256), 0,
                     stream, (const uint16_t*)qkv, positions, slots, cos_sin, (uint16_t*)cache, qkv_stride,
                     n_q, n_kv, page_stride, v_offset, rotary, n_rows);
  return 0;
}

// ---------------------------------------------------------------------------------
// SwiGLU: out[t, i] = silu(x[t, i]) * x[t, inter + i];   x is [tokens, 2*inter].
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ x,
                                                       uint16_t* __restrict__ out, long tokens,
                                                       int inter) {
  const long chunks_per_row = inter / 8;
  const long total = tokens * chunks_per_row;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long t = idx / chunks_per_row;
    const long i = (idx - t * chunks_per_row) * 8;
    const uint16_t* xr = x + t * 2 * (long)inter;
    u32x4 g = *reinterpret_cast<const u32x4*>(xr + i);
    u32x4 u = *reinterpret_cast<const u32x4*>(xr + inter + i);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g0 = bf16lo(g[k]), g1 = bf16hi(g[k]);
      const float r0 = g0 / (1.f + __expf(-g0)) * bf16lo(u[k]);
      const float r1 = g1 / (1.f + __expf(-g1)) * bf16hi(u[k]);
      o[k] = pack_bf16x2(r0, r1);
    }
    *reinterpret_cast<u32x4*>(out + t * (long)inter + i) = o;
  }
}

VPSA_EXPORT int vpsa_silu_mul(const void* x, void* out, long tokens, int inter,
                              hipStream_t stream) {
  if (tokens <= 0) return 0;
  if (inter % 8 != 0) return -1;
  long total = tokens * (long)(inter / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(silu_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     (const uint16_t*)x, (uint16_t*)out, tokens, inter);
  return 0;
}

// ---------------------------------------------------------------------------------
// Greedy sampling: out[b] = argmax_v logits[b, v] (lowest index wins ties).
// One 1024-thread block per row, 8 bf16 per lane per step.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_bf16_kernel(const uint16_t* __restrict__ logits,
                                                           long* __restrict__ out, int vocab,
                                                           long row_stride) {
  const int row = blockIdx.x;
  const uint16_t* lr = logits + (long)row * row_stride;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  const int nvec = vocab / 8;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    u32x4 a = *reinterpret_cast<const u32x4*>(lr + (long)c * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v0 = bf16lo(a[k]), v1 = bf16hi(a[k]);
      if (v0 > best) { best = v0; bidx = c * 8 + 2 * k; }
      if (v1 > best) { best = v1; bidx = c * 8 + 2 * k + 1; }
    }
  }
  for (int i = nvec * 8 + threadIdx.x; i < vocab; i += blockDim.x) {
    const float v = bf16_bits_to_f32(lr[i]);
    if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bidx, off, 64);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
  __shared__ float sb[16];
  __shared__ int si[16];
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sb[wid] = best; si[wid] = bidx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    for (int i = 1; i < nw; ++i) {
      if (sb[i] > best || (sb[i] == best && si[i] < bidx)) { best = sb[i]; bidx = si[i]; }
    }
    out[row] = bidx == 0x7fffffff ? 0 : bidx;
  }
}

VPSA_EXPORT int vpsa_argmax_bf16(const void* logits, long* out, int rows, int vocab,
                                 long row_stride, hipStream_t stream) {
  if (rows <= 0) return 0;
  if (row_stride % 8 != 0) return -1;
  hipLaunchKernelGGL(argmax_bf16_kernel, dim3(rows), dim3(1024), 0, stream,
                     (const uint16_t*)logits, out, vocab, row_stride);
  return 0;
}
