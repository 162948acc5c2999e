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
  hipSetDevice(dev);
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

API int comm_rccl_send_pages(void* c, const void* base, size_t page_bytes, const int64_t* pages, int n,
                             int peer, void* stream) {
  bind_thread();
  ncclGroupStart();
  for (int i = 0; i < n; ++i) {
    ncclResult_t r = ncclSend((const char*)base + (size_t)pages[i] * page_bytes, page_bytes, ncclChar,
                              peer, (ncclComm_t)c, (hipStream_t)stream);
    if (r != ncclSuccess) {
      ncclGroupEnd();
      return (int)r;
    }
  }
  return (int)ncclGroupEnd();
}

API int comm_rccl_recv_pages(void* c, void* base, size_t page_bytes, const int64_t* pages, int n,
                             int peer, void* stream) {
  bind_thread();
  ncclGroupStart();
  for (int i = 0; i < n; ++i) {
    ncclResult_t r = ncclRecv((char*)base + (size_t)pages[i] * page_bytes, page_bytes, ncclChar, peer,
                              (ncclComm_t)c, (hipStream_t)stream);
    if (r != ncclSuccess) {
      ncclGroupEnd();
      return (int)r;
    }
  }
  return (int)ncclGroupEnd();
}
