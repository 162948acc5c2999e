// Read-bandwidth probes: how fast can this card stream HBM, by loads in flight per lane, cache
// policy and access pattern?  The ceiling the KV- and weight-streaming kernels are measured against.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool CHUNKED>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ p, long n16, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  if (CHUNKED) {
    // a workgroup walks contiguous pieces of U * 4 KiB (what a KV page walk looks like)
    const long piece = 256L * U;
    for (long base = (long)blockIdx.x * piece; base + piece <= n16; base += (long)gridDim.x * piece) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = NT ? __builtin_nontemporal_load(p + base + u * 256 + threadIdx.x) : p[base + u * 256 + threadIdx.x];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
  } else {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n16; i += U * stride) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
  }
  if (acc == 0x9e3779b9u && sink != nullptr) sink[0] = acc;
}

extern "C" __attribute__((visibility("default"))) int roof_probe(const void* p, long bytes, void* sink, int blocks,
                                                                   int unroll, int nt, int chunked, hipStream_t s) {
#define L(U, NT, C) hipLaunchKernelGGL((probe<U, NT, C>), dim3(blocks), dim3(256), 0, s, (const u32x4*)p, bytes / 16, (uint32_t*)sink)
#define D(U) if (unroll == U) { if (nt && chunked) L(U, true, true); else if (nt) L(U, true, false); else if (chunked) L(U, false, true); else L(U, false, false); return 0; }
  D(2) D(4) D(8) D(16)
  return -1;
}
