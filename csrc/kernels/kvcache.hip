// KV-cache movement kernels for gfx950: page gather/scatter between the paged HBM pool
// and a contiguous staging buffer (the unit handed to the pinned-host copy engine, the
// remote cache server or a peer GPU), and the CacheGen-style lossy codec front end
// (anchor-delta + per-group quantisation; the entropy-coding back end runs on the host).
//
// The HBM pool is [page][layer][K|V][kv_head][64*128] bf16, so one page (all layers) is a
// single contiguous run of `page_elems` elements: gather/scatter are pure 16-byte-per-lane
// streaming copies whose only indirection is the page id.
#include "common.h"

__global__ __launch_bounds__(256) void kv_gather_pages_kernel(const u32x4* __restrict__ pool,
                                                              const long* __restrict__ page_ids,
                                                              u32x4* __restrict__ staging,
                                                              long vec_per_page) {
  const long page = page_ids[blockIdx.y];
  const u32x4* src = pool + page * vec_per_page;
  u32x4* dst = staging + (long)blockIdx.y * vec_per_page;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < vec_per_page; i += (long)gridDim.x * 256)
    dst[i] = __builtin_nontemporal_load(src + i);
}

__global__ __launch_bounds__(256) void kv_scatter_pages_kernel(const u32x4* __restrict__ staging,
                                                               const long* __restrict__ page_ids,
                                                               u32x4* __restrict__ pool,
                                                               long vec_per_page) {
  const long page = page_ids[blockIdx.y];
  u32x4* dst = pool + page * vec_per_page;
  const u32x4* src = staging + (long)blockIdx.y * vec_per_page;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < vec_per_page; i += (long)gridDim.x * 256)
    dst[i] = __builtin_nontemporal_load(src + i);
}

static inline int copy_grid_x(long vec_per_page, int n_pages) {
  long per = (vec_per_page + 255) / 256;
  long want = 4096 / (n_pages > 0 ? n_pages : 1);
  if (want < 1) want = 1;
  if (per > want) per = want;
  return (int)per;
}

VPSA_EXPORT int vpsa_kv_gather_pages(const void* pool, const long* page_ids, void* staging,
                                     int n_pages, long page_bytes, hipStream_t stream) {
  if (n_pages <= 0) return 0;
  if (page_bytes % 16 != 0) return -1;
  const long vpp = page_bytes / 16;
  hipLaunchKernelGGL(kv_gather_pages_kernel, dim3(copy_grid_x(vpp, n_pages), n_pages), dim3(256),
                     0, stream, (const u32x4*)pool, page_ids, (u32x4*)staging, vpp);
  return 0;
}

VPSA_EXPORT int vpsa_kv_scatter_pages(const void* staging, const long* page_ids, void* pool,
                                      int n_pages, long page_bytes, hipStream_t stream) {
  if (n_pages <= 0) return 0;
  if (page_bytes % 16 != 0) return -1;
  const long vpp = page_bytes / 16;
  hipLaunchKernelGGL(kv_scatter_pages_kernel, dim3(copy_grid_x(vpp, n_pages), n_pages), dim3(256),
                     0, stream, (const u32x4*)staging, page_ids, (u32x4*)pool, vpp);
  return 0;
}

// ---------------------------------------------------------------------------------
// CacheGen-style codec front end.
// The staging buffer is a sequence of units of `unit_elems` = n_kv*64*128 elements that
// alternate K, V per layer:  unit u -> layer (u/2) % n_layers, K when u is even.
// A group is 64 consecutive elements (32 channels of two consecutive keys in a K unit; two
// channels over 32 of the page's tokens in a V unit) and carries one fp32 scale.
//   * K groups are delta-coded against the same channels of the page's first two tokens
//     (token locality), using the anchor's *dequantised* value so errors do not chain;
//   * V groups are quantised directly;
//   * the number of quantisation levels depends on the layer (`levels[layer]`, e.g. 127
//     for early layers and 15 or 7 for late ones): the symbols always occupy one byte
//     here, and the host range coder turns the narrower alphabets into fewer bits.
// Eight lanes own one group (8 elements = 16 B per lane).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}

__device__ __forceinline__ void unpack8(const u32x4& a, float* t) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t[2 * k] = bf16lo(a[k]);
    t[2 * k + 1] = bf16hi(a[k]);
  }
}

__global__ __launch_bounds__(256) void cachegen_encode_kernel(
    const uint16_t* __restrict__ staging, int8_t* __restrict__ symbols, float* __restrict__ scales,
    const int* __restrict__ levels, long n_groups, long groups_per_unit, int n_layers) {
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  const bool ok = gid < n_groups;
  const long g = ok ? gid : 0;
  const long unit = g / groups_per_unit;
  const long gin = g - unit * groups_per_unit;
  const bool is_k = (unit & 1) == 0;
  const float L = (float)levels[(unit >> 1) % n_layers];
  const int tok = is_k ? (int)(gin & 31) : 0;  // key pair within the channel block
  const long ganchor = g - (long)tok;

  float x[8], a[8];
  unpack8(*reinterpret_cast<const u32x4*>(staging + g * 64 + sub * 8), x);
  unpack8(*reinterpret_cast<const u32x4*>(staging + ganchor * 64 + sub * 8), a);
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(a[k]));
  amax = group8_max(amax);
  const float sa = amax > 0.f ? amax / L : 1.f;
  float q[8];
  float s_out = sa;
  if (tok == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = rintf(a[k] / sa);
  } else {
    float d[8], dmax = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      d[k] = x[k] - rintf(a[k] / sa) * sa;
      dmax = fmaxf(dmax, fabsf(d[k]));
    }
    dmax = group8_max(dmax);
    const float sd = dmax > 0.f ? dmax / L : 1.f;
    s_out = sd;
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = rintf(d[k] / sd);
  }
  if (ok) {
    uint32_t w0 = 0, w1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      w0 |= ((uint32_t)(uint8_t)(int8_t)q[k]) << (8 * k);
      w1 |= ((uint32_t)(uint8_t)(int8_t)q[4 + k]) << (8 * k);
    }
    *reinterpret_cast<u32x2*>(symbols + g * 64 + sub * 8) = (u32x2){w0, w1};
    if (sub == 0) scales[g] = s_out;
  }
}

__global__ __launch_bounds__(256) void cachegen_decode_kernel(
    const int8_t* __restrict__ symbols, const float* __restrict__ scales,
    uint16_t* __restrict__ staging, long n_groups, long groups_per_unit) {
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  if (gid >= n_groups) return;
  const long unit = gid / groups_per_unit;
  const long gin = gid - unit * groups_per_unit;
  const bool is_k = (unit & 1) == 0;
  const int tok = is_k ? (int)(gin & 31) : 0;
  const long ganchor = gid - (long)tok;
  const u32x2 qa = *reinterpret_cast<const u32x2*>(symbols + ganchor * 64 + sub * 8);
  const float sa = scales[ganchor];
  float v[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = (float)(int8_t)(qa[0] >> (8 * k)) * sa;
    v[4 + k] = (float)(int8_t)(qa[1] >> (8 * k)) * sa;
  }
  if (tok != 0) {
    const u32x2 qd = *reinterpret_cast<const u32x2*>(symbols + gid * 64 + sub * 8);
    const float sd = scales[gid];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] += (float)(int8_t)(qd[0] >> (8 * k)) * sd;
      v[4 + k] += (float)(int8_t)(qd[1] >> (8 * k)) * sd;
    }
  }
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
  *reinterpret_cast<u32x4*>(staging + gid * 64 + sub * 8) = o;
}

VPSA_EXPORT int vpsa_cachegen_encode(const void* staging, void* symbols, float* scales,
                                     const int* levels, long n_elems, long unit_elems,
                                     int n_layers, hipStream_t stream) {
  if (n_elems <= 0) return 0;
  if (n_elems % 64 != 0 || unit_elems % 8192 != 0) return -1;
  const long n_groups = n_elems / 64;
  const long blocks = (n_groups * 8 + 255) / 256;
  hipLaunchKernelGGL(cachegen_encode_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     (const uint16_t*)staging, (int8_t*)symbols, scales, levels, n_groups,
                     unit_elems / 64, n_layers);
  return 0;
}

VPSA_EXPORT int vpsa_cachegen_decode(const void* symbols, const float* scales, void* staging,
                                     long n_elems, long unit_elems, hipStream_t stream) {
  if (n_elems <= 0) return 0;
  if (n_elems % 64 != 0 || unit_elems % 8192 != 0) return -1;
  const long n_groups = n_elems / 64;
  const long blocks = (n_groups * 8 + 255) / 256;
  hipLaunchKernelGGL(cachegen_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     (const int8_t*)symbols, scales, (uint16_t*)staging, n_groups,
                     unit_elems / 64);
  return 0;
}
