// GPU<->GPU KV transport for disaggregated prefill on one xGMI node.
//
// Reference contract: the prefill replica (kv_producer) hands finished-prompt KV to the decode
// replica (kv_consumer) through NIXL/UCX with CUDA IPC over NVLink, configured by
// LMCACHE_NIXL_* and hostIPC pods (/root/reference/helm/templates/deployment-vllm-multi.yaml:97-100,
// 267-305).  MI355X-native equivalents provided here:
//   (a) hipIpc: the decode replica exports its KV pool once; the prefill replica maps it and
//       writes pages with hipMemcpyAsync on a side stream (SDMA engines over the direct xGMI
//       link between the two GPUs: no compute units are taken from the decode GEMMs);
//   (b) RCCL point-to-point: ncclSend / ncclRecv pairs on a communicator created per (P, D)
//       pair, for deployments where IPC handles cannot be exchanged.
// A page (all layers of 64 tokens) is contiguous, so one message per page (8 MiB for
// Llama-3-8B) keeps an xGMI link (~153 GB/s) busy with a handful of transfers in flight.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#define API extern "C" __attribute__((visibility("default")))

// One process drives one GPU; the HIP current device is per thread (0 in a fresh thread).
// comm_bind_device records the process's device and every entry point that touches a stream
// or device memory moves its calling thread onto it first.
static int g_device = -1;
static inline void bind_thread() {
  if (g_device >= 0) (void)hipSetDevice(g_device);
}
API void comm_bind_device(int device) { g_device = device; }

// ---------------------------------------------------------------- hipIpc
API int comm_ipc_handle_size() { return (int)sizeof(hipIpcMemHandle_t); }

API int comm_ipc_export(void* dev_ptr, void* handle_out) {
  bind_thread();
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, dev_ptr);
  if (e != hipSuccess) return (int)e;
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

// Handle of the ALLOCATION that contains dev_ptr plus dev_ptr's offset inside it: a tensor from a
// caching allocator may sit in the middle of a larger hipMalloc block, and the peer maps the block.
API int comm_ipc_export_base(void* dev_ptr, void* handle_out, long* offset_out) {
  bind_thread();
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  hipError_t e = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dev_ptr);
  if (e != hipSuccess || base == nullptr) return (int)(e != hipSuccess ? e : hipErrorInvalidValue);
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, (void*)base);
  if (e != hipSuccess) return (int)e;
  memcpy(handle_out, &h, sizeof(h));
  *offset_out = (long)((char*)dev_ptr - (char*)base);
  return 0;
}

// A device allocation of its own (not carved from a caching allocator's block), so that the
// exported handle covers exactly `bytes`.
API void* comm_dev_alloc(size_t bytes) {
  bind_thread();
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  return p;
}

API void comm_dev_free(void* p) {
  bind_thread();
  (void)hipFree(p);
}

API void* comm_ipc_open(const void* handle) {
  bind_thread();
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return nullptr;
  return p;
}

API int comm_ipc_close(void* p) { return (int)hipIpcCloseMemHandle(p); }

API int comm_enable_peer(int dev, int peer) {
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) return -1;
  (void)hipSetDevice(dev);
  hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
  return (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? 0 : (int)e;
}

// Copy n pages from the local pool into a mapped peer pool on `stream`.
API int comm_push_pages(const void* src_base, void* dst_base, size_t page_bytes,
                        const int64_t* src_pages, const int64_t* dst_pages, int n, void* stream) {
  bind_thread();
  for (int i = 0; i < n; ++i) {
    const char* s = (const char*)src_base + (size_t)src_pages[i] * page_bytes;
    char* d = (char*)dst_base + (size_t)dst_pages[i] * page_bytes;
    hipError_t e = hipMemcpyAsync(d, s, page_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

// ---------------------------------------------------------------- RCCL p2p
API int comm_rccl_unique_id_size() { return (int)sizeof(ncclUniqueId); }

API int comm_rccl_unique_id(void* out) {
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return (int)r;
  memcpy(out, &id, sizeof(id));
  return 0;
}

// Blocks until all `nranks` processes have called it with the same id (run it on its own thread).
API void* comm_rccl_init(const void* id_bytes, int nranks, int rank, int device) {
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  g_device = device;
  ncclComm_t c = nullptr;
  if (ncclCommInitRank(&c, nranks, id, rank) != ncclSuccess) return nullptr;
  return c;
}

API int comm_rccl_destroy(void* c) { return (int)ncclCommDestroy((ncclComm_t)c); }

// A transfer can be hundreds of pages; one ncclGroupStart/End around all of them would enqueue
// that many point-to-point operations as ONE fused launch, whose size the library bounds.  The
// pages go in groups of at most COMM_RCCL_GROUP operations: sender and receiver cut the same list
// at the same places, so every group's sends meet their receives in order.
#define COMM_RCCL_GROUP 64

API int comm_rccl_group_limit() { return COMM_RCCL_GROUP; }

API int comm_rccl_send_pages(void* c, const void* base, size_t page_bytes, const int64_t* pages, int n,
                             int peer, void* stream) {
  bind_thread();
  for (int g0 = 0; g0 < n; g0 += COMM_RCCL_GROUP) {
    const int g1 = g0 + COMM_RCCL_GROUP < n ? g0 + COMM_RCCL_GROUP : n;
    ncclGroupStart();
    for (int i = g0; i < g1; ++i) {
      ncclResult_t r = ncclSend((const char*)base + (size_t)pages[i] * page_bytes, page_bytes, ncclChar,
                                peer, (ncclComm_t)c, (hipStream_t)stream);
      if (r != ncclSuccess) {
        ncclGroupEnd();
        return (int)r;
      }
    }
    ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) return (int)r;
  }
  return 0;
}

API int comm_rccl_recv_pages(void* c, void* base, size_t page_bytes, const int64_t* pages, int n,
                             int peer, void* stream) {
  bind_thread();
  for (int g0 = 0; g0 < n; g0 += COMM_RCCL_GROUP) {
    const int g1 = g0 + COMM_RCCL_GROUP < n ? g0 + COMM_RCCL_GROUP : n;
    ncclGroupStart();
    for (int i = g0; i < g1; ++i) {
      ncclResult_t r = ncclRecv((char*)base + (size_t)pages[i] * page_bytes, page_bytes, ncclChar, peer,
                                (ncclComm_t)c, (hipStream_t)stream);
      if (r != ncclSuccess) {
        ncclGroupEnd();
        return (int)r;
      }
    }
    ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) return (int)r;
  }
  return 0;
}

// ---------------------------------------------------------------- one-shot all-reduce
// Tensor-parallel decode steps all-reduce a few hundred KB per layer: latency, not bandwidth.
// On an xGMI node every GPU reaches every peer over its own link, so instead of a ring (one
// link in, one link out, world - 1 hops) every rank publishes its operand in a buffer that
// its peers have mapped through hipIpc and then reads all peers' buffers directly: one hop,
// all links busy at once, one kernel.
//
// Protocol (per workgroup b, which owns a fixed slice of the elements):
//   1. copy the slice of the operand into the local staging buffer of this call's phase
//      (two phases alternate, so a rank that is already in call n+1 never overwrites what a
//      peer still reads for call n: it cannot be in call n+2 before every peer has entered
//      call n+1, i.e. finished reading call n);
//   2. release-fence, then store the call's epoch into flag[b][my rank] of every peer;
//   3. wait until flag[b][r] of the local flags has reached the epoch for every peer r, with a
//      wall-clock bound: a peer that never arrives sets the error word instead of hanging the
//      queue; acquire-fence;
//   4. sum the slice over the ranks in rank order (identical result on every rank) in fp32,
//      write bf16 into the operand.
// Epochs live in device memory (one counter per workgroup, advanced by the kernel itself), so
// the launch is identical every time and can be replayed from a HIP graph.
#define AR_MAX_RANKS 8
#define AR_BLOCKS 32
#define AR_THREADS 512
#define AR_TIMEOUT_TICKS 500000000ull  // ~5 s of the 100 MHz wall clock

struct ArShared {                       // layout of every rank's published buffer
  uint32_t flags[AR_BLOCKS][AR_MAX_RANKS];
  uint32_t pad[256 - AR_MAX_RANKS];     // keep the data 16-byte aligned and off the flag lines
};

struct ArCtx {
  int rank, world;
  size_t max_bytes;
  char* local;                 // ArShared + 2 * max_bytes
  char* peers[AR_MAX_RANKS];   // mapped peer buffers (own entry = local)
  uint32_t* epochs;            // [AR_BLOCKS] device counters
  int* err;                    // error word: pinned host memory mapped into the device
  int* err_host;               // the host's view of it (readable without a device synchronisation)
  int fine_grained;            // the published buffer is fine-grained device memory
};

struct ArArgs {
  char* peers[AR_MAX_RANKS];
  uint32_t* epochs;
  int* err;
  size_t max_bytes;
  int rank, world;
};

__device__ __forceinline__ float ar_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float ar_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t ar_pack(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}

// GATHER = false: in-place sum of `data` (n elements) over the ranks.
// GATHER = true: `data` is this rank's n-element shard, `out` receives the shards of all ranks
// in rank order (world * n elements); same publication, handshake and time-out.
template <bool GATHER>
__global__ __launch_bounds__(AR_THREADS) void ar_oneshot_kernel(ArArgs a, uint16_t* data, uint16_t* out, long n) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  __shared__ uint32_t epoch_s;
  __shared__ int failed_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    epoch_s = a.epochs[b] + 1;
    failed_s = 0;
  }
  __syncthreads();
  const uint32_t e = epoch_s;
  const size_t phase_off = sizeof(ArShared) + (size_t)(e & 1u) * a.max_bytes;
  // slice of 8-element (16-byte) groups owned by this workgroup
  const long groups = (n + 7) >> 3;
  const long per = (groups + AR_BLOCKS - 1) / AR_BLOCKS;
  const long g0 = (long)b * per, g1 = g0 + per < groups ? g0 + per : groups;
  u4* mine = reinterpret_cast<u4*>(a.peers[a.rank] + phase_off);
  const u4* src = reinterpret_cast<const u4*>(data);
  for (long g = g0 + tid; g < g1; g += AR_THREADS) mine[g] = src[g];
  __threadfence_system();
  __syncthreads();
  if (tid < a.world) {
    ArShared* peer = reinterpret_cast<ArShared*>(a.peers[tid]);
    __hip_atomic_store(&peer->flags[b][a.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    ArShared* me = reinterpret_cast<ArShared*>(a.peers[a.rank]);
    const unsigned long long t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(&me->flags[b][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      if (wall_clock64() - t0 > AR_TIMEOUT_TICKS) {
        failed_s = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __threadfence_system();
  if (failed_s) {
    if (tid == 0) {
      *a.err = 1;
      a.epochs[b] = e;
    }
    return;
  }
  if (GATHER) {
    for (int r = 0; r < a.world; ++r) {
      const u4* from = reinterpret_cast<const u4*>(a.peers[r] + phase_off);
      u4* to = reinterpret_cast<u4*>(out) + (long)r * groups;
      for (long g = g0 + tid; g < g1; g += AR_THREADS) to[g] = __builtin_nontemporal_load(from + g);
    }
    if (tid == 0) a.epochs[b] = e;
    return;
  }
  u4* dst = reinterpret_cast<u4*>(data);
  for (long g = g0 + tid; g < g1; g += AR_THREADS) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < a.world; ++r) {
      const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(a.peers[r] + phase_off) + g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += ar_lo(v[j]);
        acc[2 * j + 1] += ar_hi(v[j]);
      }
    }
    u4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ar_pack(acc[2 * j], acc[2 * j + 1]);
    dst[g] = o;
  }
  if (tid == 0) a.epochs[b] = e;
}

API void* comm_ar_create(int rank, int world, size_t max_bytes) {
  bind_thread();
  if (world < 2 || world > AR_MAX_RANKS || rank < 0 || rank >= world) return nullptr;
  max_bytes = (max_bytes + 255) & ~(size_t)255;
  ArCtx* c = new ArCtx();
  c->rank = rank;
  c->world = world;
  c->max_bytes = max_bytes;
  const size_t total = sizeof(ArShared) + 2 * max_bytes;
  // hipIpcOpenMemHandle never returns for an allocation whose size modulo 4 GiB is 2 GiB or more
  // (measured, ROCm 7.2): refuse such a buffer here rather than hang the peers in comm_ar_open.
  if ((total & 0xFFFFFFFFull) >= 0x80000000ull) {
    delete c;
    return nullptr;
  }
  // The peers poll this buffer's flags and read its data while kernels of this process write them:
  // fine-grained memory is the allocation kind whose writes are visible to another device inside a
  // running kernel.  Where the runtime cannot export such an allocation over IPC, fall back to an
  // ordinary one (system-scope atomics and fences then carry the visibility, as before).
  c->fine_grained = 0;
  c->local = nullptr;
  if (hipExtMallocWithFlags((void**)&c->local, total, hipDeviceMallocFinegrained) == hipSuccess) {
    hipIpcMemHandle_t probe;
    if (hipIpcGetMemHandle(&probe, c->local) == hipSuccess) {
      c->fine_grained = 1;
    } else {
      (void)hipFree(c->local);
      c->local = nullptr;
    }
  }
  (void)hipGetLastError();
  if (!c->local && hipMalloc((void**)&c->local, total) != hipSuccess) {
    delete c;
    return nullptr;
  }
  (void)hipMemset(c->local, 0, total);
  (void)hipMalloc((void**)&c->epochs, AR_BLOCKS * sizeof(uint32_t));
  (void)hipMemset(c->epochs, 0, AR_BLOCKS * sizeof(uint32_t));
  c->err_host = nullptr;
  c->err = nullptr;
  if (hipHostMalloc((void**)&c->err_host, sizeof(int), hipHostMallocMapped) == hipSuccess &&
      hipHostGetDevicePointer((void**)&c->err, c->err_host, 0) == hipSuccess) {
    *c->err_host = 0;
  } else {
    if (c->err_host) (void)hipHostFree(c->err_host);
    c->err_host = nullptr;
    (void)hipMalloc((void**)&c->err, sizeof(int));
    (void)hipMemset(c->err, 0, sizeof(int));
  }
  (void)hipDeviceSynchronize();
  for (int r = 0; r < AR_MAX_RANKS; ++r) c->peers[r] = nullptr;
  c->peers[rank] = c->local;
  return c;
}

API int comm_ar_export(void* ctx, void* handle_out) {
  return comm_ipc_export(((ArCtx*)ctx)->local, handle_out);
}

// handles: world entries of comm_ipc_handle_size() bytes, in rank order (own entry ignored).
API int comm_ar_open(void* ctx, const void* handles) {
  bind_thread();
  ArCtx* c = (ArCtx*)ctx;
  const size_t hs = sizeof(hipIpcMemHandle_t);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    void* p = comm_ipc_open((const char*)handles + r * hs);
    if (p == nullptr) return -1 - r;
    c->peers[r] = (char*)p;
  }
  return 0;
}

API long comm_ar_max_elems(void* ctx) { return (long)(((ArCtx*)ctx)->max_bytes / 2); }

// In-place sum of `n` bf16 elements (a multiple of 8) at `data` over all ranks; data must be
// 16-byte aligned.  Returns -1 when the operand does not fit (the caller uses RCCL).
API int comm_ar_allreduce_bf16(void* ctx, void* data, long n, void* stream) {
  bind_thread();
  ArCtx* c = (ArCtx*)ctx;
  if (n <= 0) return 0;
  if ((n & 7) || (size_t)n * 2 > c->max_bytes || ((uintptr_t)data & 15)) return -1;
  ArArgs a;
  for (int r = 0; r < AR_MAX_RANKS; ++r) a.peers[r] = c->peers[r];
  a.epochs = c->epochs;
  a.err = c->err;
  a.max_bytes = c->max_bytes;
  a.rank = c->rank;
  a.world = c->world;
  hipLaunchKernelGGL(ar_oneshot_kernel<false>, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, (hipStream_t)stream, a,
                     (uint16_t*)data, (uint16_t*)nullptr, n);
  return 0;
}

// All-gather of `n` bf16 elements per rank (a multiple of 8; both pointers 16-byte aligned):
// out[r * n + i] = rank r's data[i].  Returns -1 when the shard does not fit.
API int comm_ar_allgather_bf16(void* ctx, const void* data, void* out, long n, void* stream) {
  bind_thread();
  ArCtx* c = (ArCtx*)ctx;
  if (n <= 0) return 0;
  if ((n & 7) || (size_t)n * 2 > c->max_bytes || ((uintptr_t)data & 15) || ((uintptr_t)out & 15)) return -1;
  ArArgs a;
  for (int r = 0; r < AR_MAX_RANKS; ++r) a.peers[r] = c->peers[r];
  a.epochs = c->epochs;
  a.err = c->err;
  a.max_bytes = c->max_bytes;
  a.rank = c->rank;
  a.world = c->world;
  hipLaunchKernelGGL(ar_oneshot_kernel<true>, dim3(AR_BLOCKS), dim3(AR_THREADS), 0, (hipStream_t)stream, a,
                     (uint16_t*)data, (uint16_t*)out, n);
  return 0;
}

// 1 if any call since creation gave up waiting for a peer (synchronises the device).
API int comm_ar_failed(void* ctx) {
  bind_thread();
  ArCtx* c = (ArCtx*)ctx;
  int v = 0;
  if (c->err_host) {
    (void)hipDeviceSynchronize();
    return *(volatile int*)c->err_host;
  }
  (void)hipMemcpy(&v, c->err, sizeof(int), hipMemcpyDeviceToHost);
  return v;
}

// The same word without touching the device: what the kernels that have FINISHED reported.  Cheap
// enough for once per engine step (call it after the step's own read-back).  -1: not available.
API int comm_ar_failed_nosync(void* ctx) {
  ArCtx* c = (ArCtx*)ctx;
  return c->err_host ? *(volatile int*)c->err_host : -1;
}

API int comm_ar_fine_grained(void* ctx) { return ((ArCtx*)ctx)->fine_grained; }

API void comm_ar_destroy(void* ctx) {
  bind_thread();
  ArCtx* c = (ArCtx*)ctx;
  for (int r = 0; r < c->world; ++r)
    if (r != c->rank && c->peers[r]) (void)hipIpcCloseMemHandle(c->peers[r]);
  (void)hipFree(c->local);
  (void)hipFree(c->epochs);
  if (c->err_host)
    (void)hipHostFree(c->err_host);
  else
    (void)hipFree(c->err);
  delete c;
}
