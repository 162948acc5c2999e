"""Lower KV tiers under the HBM page pool: pinned host DRAM and local disk.

Host tier: on a GPU the pool is one ``hipHostMalloc`` slab managed by ``libvpsa_kvio``
(page-sized slots, copies on a dedicated HIP stream fenced by events); on CPU-only hosts
(tests, plumbing replicas) it is a plain tensor with synchronous copies.  Both keep an LRU
``hash -> slot`` index; evicted entries spill to the disk tier when one is configured.
"""

from __future__ import annotations

import ctypes
import logging
import os
from collections import OrderedDict
from ctypes import c_int, c_int64, c_size_t, c_void_p

import torch

from .. import ops
from .serde import kvio

log = logging.getLogger("vpsa.kv")


class _Event:
    """Owns a HIP event created by the copy engine."""

    def __init__(self, lib):
        self.lib = lib
        self.h = c_void_p(lib.kvio_event_create())

    def query(self) -> bool:
        return bool(self.lib.kvio_event_query(self.h))

    def wait(self) -> None:
        self.lib.kvio_event_sync(self.h)

    def __del__(self):
        try:
            if self.h:
                self.lib.kvio_event_destroy(self.h)
                self.h = None
        except Exception:
            pass


class HostKVPool:
    def __init__(self, pool: torch.Tensor, capacity_bytes: int, on_spill=None):
        self.pool = pool
        self.page_bytes = ops.page_numel(pool) * pool.element_size()
        self.n_slots = max(1, int(capacity_bytes // self.page_bytes))
        self.index: OrderedDict[int, int] = OrderedDict()  # hash -> slot (LRU first)
        self.on_spill = on_spill
        self.gpu = pool.is_cuda
        self.stores = self.loads = self.evictions = 0
        if self.gpu:
            lib = kvio()
            lib.kvio_pool_create.restype = c_void_p
            lib.kvio_pool_create.argtypes = [c_int, c_size_t, c_size_t]
            lib.kvio_pool_destroy.argtypes = [c_void_p]
            lib.kvio_pool_slot_ptr.restype = c_void_p
            lib.kvio_pool_slot_ptr.argtypes = [c_void_p, c_int64]
            lib.kvio_pool_stream.restype = c_void_p
            lib.kvio_pool_stream.argtypes = [c_void_p]
            lib.kvio_event_create.restype = c_void_p
            lib.kvio_event_destroy.argtypes = [c_void_p]
            lib.kvio_event_record.argtypes = [c_void_p, c_void_p]
            lib.kvio_event_query.argtypes = [c_void_p]
            lib.kvio_event_sync.argtypes = [c_void_p]
            lib.kvio_stream_wait_event.argtypes = [c_void_p, c_void_p]
            lib.kvio_copy_page_parts.restype = c_int
            lib.kvio_copy_page_parts.argtypes = [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p,
                                                 c_void_p, c_int, c_int, c_void_p, c_void_p]
            self.lib = lib
            self.h = c_void_p(lib.kvio_pool_create(pool.device.index or 0, self.page_bytes, self.n_slots))
            if not self.h:
                raise MemoryError(f"could not pin {self.n_slots * self.page_bytes >> 20} MiB of host memory")
            lib.kvio_pool_num_slots.restype = c_int64
            lib.kvio_pool_num_slots.argtypes = [c_void_p]
            got = int(lib.kvio_pool_num_slots(self.h))
            if got < self.n_slots:
                log.warning("host KV tier: pinned %d of %d requested pages (%.1f of %.1f GiB); the host "
                            "refuses more locked memory", got, self.n_slots, got * self.page_bytes / 2**30,
                            self.n_slots * self.page_bytes / 2**30)
                self.n_slots = got
            self.free_slots = list(range(self.n_slots - 1, -1, -1))
        else:
            self.host = torch.empty((self.n_slots, *pool.shape[1:]), dtype=pool.dtype)
            self.free_slots = list(range(self.n_slots - 1, -1, -1))

    def close(self) -> None:
        if self.gpu and getattr(self, "h", None):
            self.lib.kvio_pool_destroy(self.h)
            self.h = None
        self.host = None

    # ------------------------------------------------------------------ index
    def contains(self, h: int) -> bool:
        return h in self.index

    def used_bytes(self) -> int:
        return len(self.index) * self.page_bytes

    def _slot_for(self, h: int) -> int:
        if not self.free_slots:
            old_h, old_slot = self.index.popitem(last=False)
            self.evictions += 1
            if self.on_spill is not None:
                self.on_spill(old_h, self.slot_tensor(old_slot))
            self.free_slots.append(old_slot)
        slot = self.free_slots.pop()
        self.index[h] = slot
        return slot

    def slot_tensor(self, slot: int) -> torch.Tensor:
        """CPU view of a slot's bytes (GPU pools: drains the copy stream first)."""
        if not self.gpu:
            return self.host[slot]
        ev = _Event(self.lib)
        self.lib.kvio_event_record(ev.h, c_void_p(self.lib.kvio_pool_stream(self.h)))
        ev.wait()
        ptr = self.lib.kvio_pool_slot_ptr(self.h, slot)
        buf = (ctypes.c_char * self.page_bytes).from_address(ptr)
        return torch.frombuffer(buf, dtype=torch.uint8).view(self.pool.dtype).view(self.pool.shape[1:])

    # ------------------------------------------------------------------ transfers
    def store(self, items: list[tuple[int, int]], compute_event=None):
        """HBM pages -> host. items: (hash, page_id). Returns a completion event or None."""
        todo = [(h, p) for h, p in items if h not in self.index]
        for h, _ in items:
            if h in self.index:
                self.index.move_to_end(h)
        if not todo:
            return None
        self.stores += len(todo)
        # While free slots last, the whole batch is one copy-engine call.  Once slots must be
        # recycled, pages go one at a time so that a slot is spilled only after the copy that
        # filled it has been issued (the spill drains the copy stream before reading).
        done = None
        wait = None
        if self.gpu and compute_event is None:
            wait = _Event(self.lib)
            self.lib.kvio_event_record(wait.h, c_void_p(torch.cuda.current_stream(self.pool.device).cuda_stream))
        pos = 0
        while pos < len(todo):
            n = min(len(todo) - pos, len(self.free_slots)) or 1
            batch = todo[pos:pos + n]
            pos += n
            slots = [self._slot_for(h) for h, _ in batch]
            if not self.gpu:
                for (h, p), s in zip(batch, slots):
                    ops.page_read_into(self.pool, p, self.host[s].view(-1))
                continue
            pages = (c_int64 * n)(*[p for _, p in batch])
            sl = (c_int64 * n)(*slots)
            done = self._copy_parts(pages, sl, n, True, wait)
        if done is not None:
            done._keep = wait
        return done

    def load(self, items: list[tuple[int, int]]):
        """host -> HBM pages. items: (hash, page_id), every hash must be resident."""
        slots = []
        for h, _ in items:
            slots.append(self.index[h])
            self.index.move_to_end(h)
        self.loads += len(items)
        if not self.gpu:
            for (h, p), s in zip(items, slots):
                ops.page_assign(self.pool, p, self.host[s].view(-1))
            return None
        n = len(items)
        pages = (c_int64 * n)(*[p for _, p in items])
        sl = (c_int64 * n)(*slots)
        # the target pages may still be read by kernels already queued on the compute stream
        wait = _Event(self.lib)
        self.lib.kvio_event_record(wait.h, c_void_p(torch.cuda.current_stream(self.pool.device).cuda_stream))
        done = self._copy_parts(pages, sl, n, False, wait)
        done._keep = wait
        return done

    def _copy_parts(self, pages, slots, n: int, to_host: bool, wait):
        """Queue the copies of ``n`` pages on the copy stream, one piece per pool the KV pool is made
        of, behind ``wait``; returns the event recorded after the last one."""
        done = _Event(self.lib)
        parts = ops.page_parts(self.pool, 0)
        off = 0
        for i, part in enumerate(parts):
            nbytes = part.numel() * part.element_size()
            rc = self.lib.kvio_copy_page_parts(
                self.h, c_void_p(part.data_ptr()), nbytes, nbytes, off, pages, slots, n, 1 if to_host else 0,
                wait.h if (wait is not None and i == 0) else None, done.h if i == len(parts) - 1 else None)
            if rc != 0:
                raise RuntimeError(f"kvio_copy_page_parts failed ({rc})")
            off += nbytes
        return done

    def make_compute_wait(self, ev) -> None:
        """Make the current compute stream wait for a copy-engine event (GPU-side fence)."""
        if ev is not None and self.gpu:
            self.lib.kvio_stream_wait_event(c_void_p(torch.cuda.current_stream(self.pool.device).cuda_stream), ev.h)


class DiskKVStore:
    """Serialised pages as files ``<dir>/<hash>.kv`` with an LRU size cap."""

    def __init__(self, path: str, capacity_bytes: int, serde):
        self.path = path
        self.capacity = capacity_bytes
        self.serde = serde
        os.makedirs(path, exist_ok=True)
        self.index: OrderedDict[int, int] = OrderedDict()  # hash -> bytes
        self.bytes = 0
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".kv"):
                try:
                    h = int(fn[:-3], 16)
                except ValueError:
                    continue
                sz = os.path.getsize(os.path.join(path, fn))
                self.index[h] = sz
                self.bytes += sz

    def _file(self, h: int) -> str:
        return os.path.join(self.path, f"{h:016x}.kv")

    def contains(self, h: int) -> bool:
        return h in self.index

    def put(self, h: int, page: torch.Tensor) -> None:
        if h in self.index:
            self.index.move_to_end(h)
            return
        blob = self.serde.encode(page)
        with open(self._file(h) + ".tmp", "wb") as f:
            f.write(blob)
        os.replace(self._file(h) + ".tmp", self._file(h))
        self.index[h] = len(blob)
        self.bytes += len(blob)
        while self.bytes > self.capacity and len(self.index) > 1:
            old, sz = self.index.popitem(last=False)
            self.bytes -= sz
            try:
                os.remove(self._file(old))
            except OSError:
                pass

    def get(self, h: int, device: torch.device) -> torch.Tensor | None:
        if h not in self.index:
            return None
        try:
            with open(self._file(h), "rb") as f:
                blob = f.read()
        except OSError:
            self.index.pop(h, None)
            return None
        self.index.move_to_end(h)
        return self.serde.decode(blob, device)
