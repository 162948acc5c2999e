// Entropy stage of the CacheGen-style KV codec on the GPU: interleaved rANS (range asymmetric
// numeral systems) over the int8 symbols that the quantiser (kvcache.hip) produces.
//
// Why rANS here: an adaptive binary range coder is a serial dependency chain per symbol (the
// host coder manages ~30 M symbols/s, 0.13 s per 8 MiB page); rANS with a static table is a
// handful of integer operations per symbol and thousands of independent streams can run side
// by side, so a page is coded in tens of microseconds next to the quantiser and the
// compressed bytes are what crosses PCIe.
//
// Layout.  The symbols of a page fall into contexts (one per layer: K and V deltas of a layer
// share statistics), ctx_len symbols each.  A context is cut into spc streams; stream j owns
// symbols j, j + spc, j + 2 spc, ... so that at every step the 64 lanes of a wave touch 64
// consecutive bytes (coalesced loads in the encoder, coalesced stores in the decoder).
// One thread codes one stream.  Per context: a 256-entry frequency table normalised to 2^12,
// built from a histogram of that very page (it travels with the blob: 512 B per context).
// Coder: 32-bit state, byte-wise renormalisation, lower bound 2^23.  The encoder runs over its
// symbols backwards and writes its bytes backwards into a private scratch slot, so the decoder
// reads forwards; vpsa_rans_pack then gathers the slots into one dense byte string at the
// offsets given by the prefix sum of the sizes.
#include "common.h"

#define RANS_BITS 12
#define RANS_M (1u << RANS_BITS)
#define RANS_LOW (1u << 23)

// grid (blocks_per_ctx, n_ctx), block 256.  hist must be zeroed by the caller.
__global__ __launch_bounds__(256) void rans_hist_kernel(const uint8_t* __restrict__ sym,
                                                        uint32_t* __restrict__ hist, long ctx_len) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint8_t* p = sym + (long)blockIdx.y * ctx_len;
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < ctx_len; i += stride) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p + i);
    atomicAdd(&h[w & 0xff], 1u);
    atomicAdd(&h[(w >> 8) & 0xff], 1u);
    atomicAdd(&h[(w >> 16) & 0xff], 1u);
    atomicAdd(&h[w >> 24], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[blockIdx.y * 256 + threadIdx.x], h[threadIdx.x]);
}

// grid n_ctx, block 256.  freq[s] = max(1, round(count * 4096 / total)) for symbols that occur,
// then the entry with the largest frequency (lowest index on ties) absorbs the rounding error
// so that the table sums to exactly 4096.  err[0] is set if that entry would drop below 1.
__global__ __launch_bounds__(256) void rans_normalize_kernel(const uint32_t* __restrict__ hist,
                                                             uint16_t* __restrict__ freq,
                                                             uint16_t* __restrict__ cum, long ctx_len,
                                                             int* __restrict__ err) {
  __shared__ int sf[256];
  __shared__ int si[256];
  __shared__ int scan[256];
  const int t = threadIdx.x;
  const uint32_t c = hist[blockIdx.x * 256 + t];
  int f = 0;
  if (c) {
    f = (int)(((unsigned long long)c * RANS_M + (unsigned long long)(ctx_len / 2)) /
              (unsigned long long)ctx_len);
    if (f < 1) f = 1;
  }
  // sum and arg-max in one tree
  scan[t] = f;
  sf[t] = f;
  si[t] = t;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) {
      scan[t] += scan[t + off];
      const int a = sf[t], b = sf[t + off];
      if (b > a || (b == a && si[t + off] < si[t])) {
        sf[t] = b;
        si[t] = si[t + off];
      }
    }
    __syncthreads();
  }
  const int diff = (int)RANS_M - scan[0];
  const int imax = si[0];
  __syncthreads();
  if (t == imax) {
    f += diff;
    if (f < 1) {
      err[0] = 1;
      f = 1;
    }
  }
  // exclusive scan of the final frequencies
  scan[t] = f;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = t >= off ? scan[t - off] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  freq[blockIdx.x * 256 + t] = (uint16_t)f;
  cum[blockIdx.x * 256 + t] = (uint16_t)(scan[t] - f);
}

// grid (spc / 256, n_ctx), block 256.  cap = scratch bytes per stream (2 * symbols per stream).
__global__ __launch_bounds__(256) void rans_encode_kernel(
    const uint8_t* __restrict__ sym, const uint16_t* __restrict__ freq, const uint16_t* __restrict__ cum,
    uint8_t* __restrict__ scratch, uint32_t* __restrict__ states, uint16_t* __restrict__ sizes,
    long ctx_len, int spc, int cap) {
  __shared__ uint16_t sf[256], sc[256];
  const int ctx = blockIdx.y;
  sf[threadIdx.x] = freq[ctx * 256 + threadIdx.x];
  sc[threadIdx.x] = cum[ctx * 256 + threadIdx.x];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  const long gs = (long)ctx * spc + j;
  const uint8_t* p = sym + (long)ctx * ctx_len + j;
  uint8_t* buf = scratch + gs * cap;
  const int len = (int)(ctx_len / spc);
  int pos = cap;
  uint32_t x = RANS_LOW;
  for (int k = len - 1; k >= 0; --k) {
    const uint32_t s = p[(long)k * spc];
    const uint32_t f = sf[s];
    const uint32_t x_max = ((RANS_LOW >> RANS_BITS) << 8) * f;
    while (x >= x_max && pos > 0) {
      buf[--pos] = (uint8_t)(x & 0xff);
      x >>= 8;
    }
    x = ((x / f) << RANS_BITS) + (x % f) + sc[s];
  }
  states[gs] = x;
  sizes[gs] = (uint16_t)(cap - pos);
}

// One thread per stream: move its bytes from the scratch slot to the dense output.
__global__ __launch_bounds__(256) void rans_pack_kernel(const uint8_t* __restrict__ scratch,
                                                        const uint16_t* __restrict__ sizes,
                                                        const long* __restrict__ offsets,
                                                        uint8_t* __restrict__ out, long n_streams,
                                                        int cap) {
  const long gs = (long)blockIdx.x * 256 + threadIdx.x;
  if (gs >= n_streams) return;
  const int n = sizes[gs];
  const uint8_t* src = scratch + gs * cap + (cap - n);
  uint8_t* dst = out + offsets[gs];
  for (int i = 0; i < n; ++i) dst[i] = src[i];
}

// grid (spc / 256, n_ctx), block 256.  coded_len bounds every read (a damaged blob decodes to
// garbage, never out of bounds).
__global__ __launch_bounds__(256) void rans_decode_kernel(
    const uint8_t* __restrict__ coded, long coded_len, const long* __restrict__ offsets,
    const uint32_t* __restrict__ states, const uint16_t* __restrict__ freq,
    const uint16_t* __restrict__ cum, uint8_t* __restrict__ sym, long ctx_len, int spc) {
  __shared__ uint16_t sf[256], sc[256];
  __shared__ uint8_t slot2sym[RANS_M];
  const int ctx = blockIdx.y;
  {
    const uint32_t f = freq[ctx * 256 + threadIdx.x], c = cum[ctx * 256 + threadIdx.x];
    sf[threadIdx.x] = (uint16_t)f;
    sc[threadIdx.x] = (uint16_t)c;
    for (uint32_t i = c; i < c + f && i < RANS_M; ++i) slot2sym[i] = (uint8_t)threadIdx.x;
  }
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  const long gs = (long)ctx * spc + j;
  uint8_t* out = sym + (long)ctx * ctx_len + j;
  const int len = (int)(ctx_len / spc);
  long rp = offsets[gs];
  uint32_t x = states[gs];
  for (int k = 0; k < len; ++k) {
    const uint32_t slot = x & (RANS_M - 1);
    const uint32_t s = slot2sym[slot];
    out[(long)k * spc] = (uint8_t)s;
    x = (uint32_t)sf[s] * (x >> RANS_BITS) + slot - sc[s];
    while (x < RANS_LOW) {
      const uint32_t b = (rp >= 0 && rp < coded_len) ? coded[rp] : 0u;
      ++rp;
      x = (x << 8) | b;
    }
  }
}

static bool rans_shape_ok(long n, int n_ctx, int spc) {
  if (n <= 0 || n_ctx <= 0 || spc <= 0 || spc % 256 != 0 || n % n_ctx != 0) return false;
  const long ctx_len = n / n_ctx;
  if (ctx_len % spc != 0 || ctx_len % 4 != 0) return false;
  const long len = ctx_len / spc;
  return len >= 1 && 2 * len <= 65535;
}

// Pass 1: histogram -> table -> per-stream coding into scratch.  hist [n_ctx * 256] u32 must be
// zero; scratch holds n_ctx * spc slots of 2 * (ctx_len / spc) bytes.
VPSA_EXPORT int vpsa_rans_encode(const void* sym, long n, int n_ctx, int spc, void* hist, void* freq,
                                 void* cum, void* scratch, void* states, void* sizes, int* err,
                                 hipStream_t stream) {
  if (!rans_shape_ok(n, n_ctx, spc)) return -1;
  const long ctx_len = n / n_ctx;
  const int cap = (int)(2 * (ctx_len / spc));
  int hb = (int)((ctx_len / 4 + 255) / 256);
  if (hb > 64) hb = 64;
  hipLaunchKernelGGL(rans_hist_kernel, dim3(hb, n_ctx), dim3(256), 0, stream, (const uint8_t*)sym,
                     (uint32_t*)hist, ctx_len);
  hipLaunchKernelGGL(rans_normalize_kernel, dim3(n_ctx), dim3(256), 0, stream, (const uint32_t*)hist,
                     (uint16_t*)freq, (uint16_t*)cum, ctx_len, err);
  hipLaunchKernelGGL(rans_encode_kernel, dim3(spc / 256, n_ctx), dim3(256), 0, stream,
                     (const uint8_t*)sym, (const uint16_t*)freq, (const uint16_t*)cum,
                     (uint8_t*)scratch, (uint32_t*)states, (uint16_t*)sizes, ctx_len, spc, cap);
  return 0;
}

// Pass 2: offsets = exclusive prefix sum of sizes (computed by the caller).
VPSA_EXPORT int vpsa_rans_pack(const void* scratch, const void* sizes, const long* offsets, void* out,
                               long n, int n_ctx, int spc, hipStream_t stream) {
  if (!rans_shape_ok(n, n_ctx, spc)) return -1;
  const long n_streams = (long)n_ctx * spc;
  const int cap = (int)(2 * (n / n_ctx / spc));
  hipLaunchKernelGGL(rans_pack_kernel, dim3((unsigned)((n_streams + 255) / 256)), dim3(256), 0, stream,
                     (const uint8_t*)scratch, (const uint16_t*)sizes, offsets, (uint8_t*)out,
                     n_streams, cap);
  return 0;
}

// cum is recomputed here from freq (only freq travels in the blob); cum_ws [n_ctx * 256] u16.
__global__ __launch_bounds__(256) void rans_cum_kernel(const uint16_t* __restrict__ freq,
                                                       uint16_t* __restrict__ cum) {
  __shared__ int scan[256];
  const int t = threadIdx.x;
  const int f = freq[blockIdx.x * 256 + t];
  scan[t] = f;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = t >= off ? scan[t - off] : 0;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  cum[blockIdx.x * 256 + t] = (uint16_t)(scan[t] - f);
}

VPSA_EXPORT int vpsa_rans_decode(const void* coded, long coded_len, const long* offsets,
                                 const void* states, const void* freq, void* cum_ws, void* sym, long n,
                                 int n_ctx, int spc, hipStream_t stream) {
  if (!rans_shape_ok(n, n_ctx, spc)) return -1;
  const long ctx_len = n / n_ctx;
  hipLaunchKernelGGL(rans_cum_kernel, dim3(n_ctx), dim3(256), 0, stream, (const uint16_t*)freq,
                     (uint16_t*)cum_ws);
  hipLaunchKernelGGL(rans_decode_kernel, dim3(spc / 256, n_ctx), dim3(256), 0, stream,
                     (const uint8_t*)coded, coded_len, offsets, (const uint32_t*)states,
                     (const uint16_t*)freq, (const uint16_t*)cum_ws, (uint8_t*)sym, ctx_len, spc);
  return 0;
}
