// KV I/O engine for one GPU: pinned host-DRAM page pool, side-stream copy engine
// (HBM <-> pinned host via hipMemcpyAsync, overlapped with compute), and the host half of
// the CacheGen-style codec (adaptive binary-decomposed range coder over the quantised
// symbols produced by the cachegen_encode kernel).
//
// Contract it fulfils (reference launches LMCache for this and only configures it):
// LMCACHE_LOCAL_CPU / LMCACHE_MAX_LOCAL_CPU_SIZE host offload
// (/root/reference/helm/templates/deployment-vllm-multi.yaml:306-313) and the remote
// serde selection naive|cachegen (:321-322).
//
// Design for MI355X: a KV page (all layers of 64 tokens) is one contiguous run in HBM
// (8 MiB for Llama-3-8B), so a page moves with a single large DMA.  The pool is one
// hipHostMalloc'ed slab carved into page-sized slots (no per-transfer registration); copies
// are issued on a dedicated non-blocking stream and fenced with events, never with a
// device-wide synchronise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "rangecoder.hpp"

#define API extern "C" __attribute__((visibility("default")))

namespace {

struct Pool {
  // pinned memory comes in slabs of ~1 GiB: one huge hipHostMalloc can fail where many
  // moderate ones succeed, and a pool that cannot reach its configured size still works
  std::vector<char*> slabs;
  size_t slots_per_slab = 1;
  char* slot_addr(int64_t slot) const {
    return slabs[(size_t)slot / slots_per_slab] + ((size_t)slot % slots_per_slab) * slot_bytes;
  }
  size_t slot_bytes = 0;
  size_t n_slots = 0;
  std::vector<int64_t> free_list;
  std::mutex mu;
  hipStream_t stream = nullptr;
  int device = 0;
};

// One process drives one GPU.  The HIP current device is per thread (0 in a fresh thread), and
// events, streams and async copies follow it: every entry point moves its calling thread to the
// device the pool was created on.
int g_device = -1;
inline void bind_thread() {
  if (g_device >= 0) (void)hipSetDevice(g_device);
}

}  // namespace

// ---------------------------------------------------------------- pinned pool
API void* kvio_pool_create(int device, size_t slot_bytes, size_t n_slots) {
  if (slot_bytes == 0 || n_slots == 0) return nullptr;
  Pool* p = new Pool();
  p->device = device;
  g_device = device;
  p->slot_bytes = slot_bytes;
  p->n_slots = n_slots;
  if (hipSetDevice(device) != hipSuccess) {
    delete p;
    return nullptr;
  }
  p->slots_per_slab = std::max<size_t>(1, ((size_t)1 << 30) / slot_bytes);
  size_t got = 0;
  while (got < n_slots) {
    const size_t n = std::min(p->slots_per_slab, n_slots - got);
    char* slab = nullptr;
    if (hipHostMalloc((void**)&slab, slot_bytes * n, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      // The host's lockable memory is exhausted.  Give some back: the runtime, the GEMM library
      // and the engine's own staging buffers pin memory too and fail hard when none is left.
      for (int k = 0; k < 4 && p->slabs.size() > 4; ++k) {
        hipHostFree(p->slabs.back());
        p->slabs.pop_back();
        got -= p->slots_per_slab;
      }
      break;
    }
    p->slabs.push_back(slab);
    got += n;
  }
  if (got == 0) {
    delete p;
    return nullptr;
  }
  p->n_slots = n_slots = got;
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) {
    for (char* sl : p->slabs) hipHostFree(sl);
    delete p;
    return nullptr;
  }
  p->free_list.reserve(n_slots);
  for (int64_t i = (int64_t)n_slots - 1; i >= 0; --i) p->free_list.push_back(i);
  return p;
}

// Slots actually pinned (<= the requested count when the host refuses more locked memory).
API int64_t kvio_pool_num_slots(void* h) { return h ? (int64_t) static_cast<Pool*>(h)->n_slots : 0; }

API void kvio_pool_destroy(void* h) {
  Pool* p = static_cast<Pool*>(h);
  if (!p) return;
  hipStreamSynchronize(p->stream);
  hipStreamDestroy(p->stream);
  for (char* sl : p->slabs) hipHostFree(sl);
  delete p;
}

API int64_t kvio_pool_alloc(void* h) {
  Pool* p = static_cast<Pool*>(h);
  std::lock_guard<std::mutex> lk(p->mu);
  if (p->free_list.empty()) return -1;
  int64_t s = p->free_list.back();
  p->free_list.pop_back();
  return s;
}

API void kvio_pool_free(void* h, int64_t slot) {
  Pool* p = static_cast<Pool*>(h);
  std::lock_guard<std::mutex> lk(p->mu);
  p->free_list.push_back(slot);
}

API size_t kvio_pool_free_slots(void* h) {
  Pool* p = static_cast<Pool*>(h);
  std::lock_guard<std::mutex> lk(p->mu);
  return p->free_list.size();
}

API void* kvio_pool_slot_ptr(void* h, int64_t slot) {
  Pool* p = static_cast<Pool*>(h);
  if (slot < 0 || (size_t)slot >= p->n_slots) return nullptr;
  return p->slot_addr(slot);
}

API void* kvio_pool_stream(void* h) { return static_cast<Pool*>(h)->stream; }

// ---------------------------------------------------------------- copy engine
// All copies run on the pool's side stream.  `wait_event` (may be null) is an event recorded
// on the compute stream after the producing kernels; the returned event (created here, owned
// by the caller -> kvio_event_destroy) fires when the batch of copies has landed.
API void* kvio_event_create() {
  bind_thread();
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}
API void kvio_event_destroy(void* e) {
  if (e) hipEventDestroy((hipEvent_t)e);
}
API int kvio_event_record(void* e, void* stream) {
  bind_thread();
  return (int)hipEventRecord((hipEvent_t)e, (hipStream_t)stream);
}
API int kvio_event_query(void* e) { return hipEventQuery((hipEvent_t)e) == hipSuccess ? 1 : 0; }
API int kvio_event_sync(void* e) { return (int)hipEventSynchronize((hipEvent_t)e); }
API int kvio_stream_wait_event(void* stream, void* e) {
  bind_thread();
  return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e, 0);
}

// device pages -> pinned slots.  dev_base + page_ids[i] * page_bytes is the source.
API int kvio_offload_pages(void* h, const void* dev_base, size_t page_bytes, const int64_t* page_ids,
                           const int64_t* slots, int n, void* wait_event, void* done_event) {
  bind_thread();
  Pool* p = static_cast<Pool*>(h);
  if (page_bytes > p->slot_bytes) return -1;
  if (wait_event && hipStreamWaitEvent(p->stream, (hipEvent_t)wait_event, 0) != hipSuccess) return -2;
  for (int i = 0; i < n; ++i) {
    const char* src = (const char*)dev_base + (size_t)page_ids[i] * page_bytes;
    char* dst = p->slot_addr(slots[i]);
    if (hipMemcpyAsync(dst, src, page_bytes, hipMemcpyDeviceToHost, p->stream) != hipSuccess) return -3;
  }
  if (done_event && hipEventRecord((hipEvent_t)done_event, p->stream) != hipSuccess) return -4;
  return 0;
}

// pinned slots -> device pages.
API int kvio_load_pages(void* h, void* dev_base, size_t page_bytes, const int64_t* page_ids,
                        const int64_t* slots, int n, void* wait_event, void* done_event) {
  bind_thread();
  Pool* p = static_cast<Pool*>(h);
  if (page_bytes > p->slot_bytes) return -1;
  if (wait_event && hipStreamWaitEvent(p->stream, (hipEvent_t)wait_event, 0) != hipSuccess) return -2;
  for (int i = 0; i < n; ++i) {
    char* dst = (char*)dev_base + (size_t)page_ids[i] * page_bytes;
    const char* src = p->slot_addr(slots[i]);
    if (hipMemcpyAsync(dst, src, page_bytes, hipMemcpyHostToDevice, p->stream) != hipSuccess) return -3;
  }
  if (done_event && hipEventRecord((hipEvent_t)done_event, p->stream) != hipSuccess) return -4;
  return 0;
}

// One piece of every page: `part_bytes` at `dev_base + page * dev_stride` <-> the same number of
// bytes at `slot_off` inside the page's pinned slot (a KV pool allocated as several pools of
// consecutive layers keeps a page as one such piece per pool).  to_host: 1 = offload, 0 = load.
API int kvio_copy_page_parts(void* h, void* dev_base, size_t dev_stride, size_t part_bytes, size_t slot_off,
                             const int64_t* page_ids, const int64_t* slots, int n, int to_host,
                             void* wait_event, void* done_event) {
  bind_thread();
  Pool* p = static_cast<Pool*>(h);
  if (slot_off + part_bytes > p->slot_bytes || part_bytes > dev_stride) return -1;
  if (wait_event && hipStreamWaitEvent(p->stream, (hipEvent_t)wait_event, 0) != hipSuccess) return -2;
  for (int i = 0; i < n; ++i) {
    char* dev = (char*)dev_base + (size_t)page_ids[i] * dev_stride;
    char* host = p->slot_addr(slots[i]) + slot_off;
    const hipError_t e = to_host ? hipMemcpyAsync(host, dev, part_bytes, hipMemcpyDeviceToHost, p->stream)
                                 : hipMemcpyAsync(dev, host, part_bytes, hipMemcpyHostToDevice, p->stream);
    if (e != hipSuccess) return -3;
  }
  if (done_event && hipEventRecord((hipEvent_t)done_event, p->stream) != hipSuccess) return -4;
  return 0;
}

// ---------------------------------------------------------------- entropy coder (host)
// The coder itself is host-only code in rangecoder.hpp (built and fuzzed under sanitizers by the
// CPU test suite); these are its C entry points.
API int64_t kvio_rc_encode(const int8_t* sym, size_t n, uint8_t* out, size_t cap, size_t ctx_period,
                           int n_ctx) {
  return vrc::encode(sym, n, out, cap, ctx_period, n_ctx);
}

// 0 = ok, -1 = the stream ended before `n` symbols were decoded (truncated or not a stream)
API int kvio_rc_decode(const uint8_t* in, size_t len, int8_t* sym, size_t n, size_t ctx_period,
                       int n_ctx) {
  return vrc::decode(in, len, sym, n, ctx_period, n_ctx);
}
