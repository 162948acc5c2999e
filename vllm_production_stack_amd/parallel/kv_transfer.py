"""Prefill -> decode KV page transfer between engine replicas (disaggregated prefill).

Protocol (control plane = the replicas' own HTTP ports, data plane chosen per pair):
  1. producer  POST <decode>/kv/reserve {hashes}      -> {pages, missing, transports...}
     the decode replica allocates HBM pages for the hashes it does not hold yet;
  2. producer  moves the page bytes:
        ipc   the decode replica's LANDING WINDOW (a device allocation of its own, of a size
              hipIpc can map: LMCACHE_NIXL_BUFFER_SIZE, default 1 GiB, see ipc_safe_size) is mapped through a hipIpc handle and written with
              hipMemcpyAsync on a side stream (SDMA over the direct xGMI link); at commit the
              decode replica copies window slots into its pool pages on the device.  The pool
              itself is not exported: hipIpcOpenMemHandle never returns for allocations
              whose size modulo 4 GiB is 2 GiB or more (benchmarks/ipc_probe.py), and a pool's
              size follows from the memory budget.  A request with more pages than free slots goes in
              rounds (reserve -> push -> commit, repeated),
        rccl  ncclSend / ncclRecv of whole pages on side streams, over a two-rank RCCL
              communicator created per (prefill, decode) pair (the unique id travels over the
              control plane); for GPUs that cannot exchange IPC handles (other node, other
              container namespace).  Opt-in: VPSA_PD_TRANSPORT=rccl,
        tcp   POST <decode>/kv/push with the raw page bytes (any host, CPU tests);
  3. producer  POST <decode>/kv/commit {hashes, pages} -> pages are published in the decode
     replica's prefix cache; the decode request then finds its prompt resident.
Reference behaviour being provided: the NIXL sender/receiver pair configured by
LMCACHE_NIXL_* (/root/reference/helm/templates/deployment-vllm-multi.yaml:267-305) and the
router's two-phase request flow (/root/reference/src/vllm_router/services/request_service/request.py:339-431).
"""

from __future__ import annotations

import ctypes
import json
import logging
import urllib.request
from ctypes import c_int, c_size_t, c_void_p

import torch

from .. import _native

log = logging.getLogger("vpsa.kvxfer")
_comm = None
_api_key: str | None = None  # bearer token of the deployment's replicas (they share VLLM_API_KEY)
_peer_maps: dict[tuple[str, str], int] = {}  # (target url, handle) -> mapped base of its landing window
_rccl_peers: dict[str, tuple[int, str]] = {}  # target url -> (communicator, key on the decode side)


def host_identity() -> str:
    """Same string for two processes iff they can exchange hipIpc handles by construction:
    one kernel instance (boot id) and one hostname."""
    import socket

    try:
        with open("/proc/sys/kernel/random/boot_id") as f:
            boot = f.read().strip()
    except OSError:
        boot = ""
    return f"{socket.gethostname()}/{boot}"


def transport_preference() -> str:
    import os

    return os.environ.get("VPSA_PD_TRANSPORT", "auto").lower()


def comm():
    global _comm
    if _comm is None:
        lib = _native.load("vpsa_comm")
        lib.comm_ipc_handle_size.restype = c_int
        lib.comm_ipc_export.restype = c_int
        lib.comm_ipc_export.argtypes = [c_void_p, c_void_p]
        lib.comm_ipc_export_base.restype = c_int
        lib.comm_ipc_export_base.argtypes = [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_long)]
        lib.comm_ipc_open.restype = c_void_p
        lib.comm_ipc_open.argtypes = [c_void_p]
        lib.comm_dev_alloc.restype = c_void_p
        lib.comm_dev_alloc.argtypes = [c_size_t]
        lib.comm_dev_free.argtypes = [c_void_p]
        lib.comm_ipc_close.argtypes = [c_void_p]
        lib.comm_push_pages.restype = c_int
        lib.comm_push_pages.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p]
        lib.comm_rccl_unique_id_size.restype = c_int
        lib.comm_rccl_unique_id.restype = c_int
        lib.comm_rccl_unique_id.argtypes = [c_void_p]
        lib.comm_rccl_init.restype = c_void_p
        lib.comm_rccl_init.argtypes = [c_void_p, c_int, c_int, c_int]
        lib.comm_rccl_destroy.argtypes = [c_void_p]
        for fn in (lib.comm_rccl_send_pages, lib.comm_rccl_recv_pages):
            fn.restype = c_int
            fn.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p]
        lib.comm_bind_device.argtypes = [c_int]
        if torch.cuda.is_available():
            # this process's GPU; the library moves every calling thread onto it
            from .. import ops

            ops.ensure_device()
            lib.comm_bind_device(c_int(torch.cuda.current_device()))
        _comm = lib
    return _comm


def _post(url: str, payload: dict, raw: bytes | None = None, timeout: float = 60.0) -> dict:
    if raw is None:
        data = json.dumps(payload).encode()
        headers = {"Content-Type": "application/json"}
    else:
        meta = json.dumps(payload).encode()
        data = len(meta).to_bytes(4, "little") + meta + raw
        headers = {"Content-Type": "application/octet-stream"}
    if _api_key:
        headers["Authorization"] = f"Bearer {_api_key}"
    req = urllib.request.Request(url, data=data, headers=headers)
    with urllib.request.urlopen(req, timeout=timeout) as r:
        body = r.read()
    return json.loads(body) if body else {}


def ipc_safe_size(nbytes: int) -> int:
    """Largest size <= ``nbytes`` (whole 4 KiB pages) that hipIpcOpenMemHandle maps.  Measured on
    ROCm 7.2 / MI355X: the importer never returns when the allocation's size modulo 4 GiB is
    2 GiB or more (2, 2.5, 3, 6, 7, 10, 14 GiB hang; 0.5, 1, 4, 4.5, 5, 8, 9, 12, 16, 64, 100 GiB
    map in under a millisecond), as if the size went through a signed 32-bit value somewhere."""
    nbytes = max(int(nbytes), 0) & ~4095
    if nbytes & 0x80000000:
        nbytes = (nbytes & ~0xFFFFFFFF) | (0x80000000 - (2 << 20))
    return nbytes


def export_ipc_handle(pool: torch.Tensor) -> tuple[str, int] | None:
    """(hex hipIpc handle of the device allocation that contains the tensor, byte offset of the
    tensor inside that allocation).  The caching allocator may place a tensor in the middle of a
    larger block; a peer maps the whole block and adds the offset.  Whether the peer can map it
    depends on the BLOCK's size (see ``ipc_safe_size``), which is why the P/D data plane exports
    a window it allocated itself; this function serves benchmarks/ipc_probe.py."""
    if not pool.is_cuda:
        return None
    try:
        lib = comm()
        n = lib.comm_ipc_handle_size()
        buf = ctypes.create_string_buffer(n)
        off = ctypes.c_long(0)
        if lib.comm_ipc_export_base(c_void_p(pool.data_ptr()), buf, ctypes.byref(off)) != 0:
            return None
        return buf.raw.hex(), int(off.value)
    except Exception as e:
        log.warning("hipIpc export failed: %s", e)
        return None


def push_request_pages(engine, target: str, hashes: list[int], pages: list[int]) -> str:
    """Producer side: make ``hashes`` resident on the decode replica at ``target``.  The calls carry
    this replica's API key: the replicas of one deployment are started with the same VLLM_API_KEY
    (/root/reference/helm/templates/deployment-vllm-multi.yaml:216-230), and the decode replica's
    /kv/* endpoints sit behind the same bearer check as its OpenAI endpoints."""
    import os

    global _api_key
    _api_key = getattr(engine.cfg, "api_key", None) or os.environ.get("VLLM_API_KEY") or None
    runner = engine.runner
    pool = runner.pool
    page_bytes = pool.stride(0) * pool.element_size()
    idx = {h: i for i, h in enumerate(hashes)}
    pref = transport_preference()
    used = "resident"
    import time

    rounds, deadline = 0, time.time() + 20.0
    while rounds < len(hashes) + 2:  # one round per landing-window fill; bounded in any case
        res = _post(target + "/kv/reserve", {"hashes": hashes, "page_bytes": page_bytes, "source_rank": _my_rank(),
                                             "host_id": host_identity(), "transport": pref, "gpu": bool(pool.is_cuda)})
        if res.get("busy"):  # the window is full of other producers' pages
            if time.time() > deadline:
                pref = "tcp"  # the next reserve asks for pool pages and the bytes go over the control plane
            time.sleep(0.002)
            continue
        rounds += 1
        dst_pages = res.get("pages", [])
        missing = res.get("missing", [])
        if not missing:
            break
        src = [pages[idx[h]] for h in missing]
        mode = "tcp"
        if pool.is_cuda and res.get("ipc_handle") and res.get("ipc_slots") is not None:
            try:
                _push_ipc(pool, target, res["ipc_handle"], res.get("ipc_offset", 0), page_bytes, src, res["ipc_slots"])
                mode = "ipc"
            except Exception as e:
                log.warning("ipc push failed (%s); falling back", e)
        if mode == "tcp" and pool.is_cuda and pref == "rccl" and res.get("gpu"):
            try:
                _push_rccl(pool, target, page_bytes, src, dst_pages)
                mode = "rccl"
            except Exception as e:
                log.warning("rccl push failed (%s); falling back to tcp", e)
        if mode == "tcp":
            chunk = 8
            for i in range(0, len(src), chunk):
                blob = pool[torch.tensor(src[i:i + chunk], device=pool.device)].contiguous().cpu()
                raw = blob.view(torch.uint8).numpy().tobytes()
                _post(target + "/kv/push", {"pages": dst_pages[i:i + chunk], "page_bytes": page_bytes}, raw)
        _post(target + "/kv/commit", {"hashes": missing, "pages": dst_pages, "landed": mode == "ipc"})
        used = mode
        if not res.get("more"):
            break
    return used


def _my_rank() -> int:
    import torch.distributed as dist

    return dist.get_rank() if dist.is_available() and dist.is_initialized() else -1


def _push_ipc(pool, target, handle_hex, offset, page_bytes, src_pages, dst_slots) -> None:
    """Copy pool pages into slots of the peer's landing window."""
    lib = comm()
    base = _peer_maps.get((target, handle_hex))
    if base is None:
        h = bytes.fromhex(handle_hex)
        base = lib.comm_ipc_open(ctypes.create_string_buffer(h, len(h)))
        if not base:
            raise RuntimeError("hipIpcOpenMemHandle failed")
        for key in [k for k in _peer_maps if k[0] == target]:  # the replica behind this URL restarted
            lib.comm_ipc_close(c_void_p(_peer_maps.pop(key)))
        _peer_maps[(target, handle_hex)] = base
    n = len(src_pages)
    sp = (ctypes.c_int64 * n)(*src_pages)
    dp = (ctypes.c_int64 * n)(*dst_slots)
    stream = torch.cuda.Stream(device=pool.device)
    stream.wait_stream(torch.cuda.current_stream(pool.device))
    rc = lib.comm_push_pages(c_void_p(pool.data_ptr()), c_void_p(base + offset), page_bytes,
                             sp, dp, n, c_void_p(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(f"comm_push_pages failed ({rc})")
    stream.synchronize()


def _push_rccl(pool, target, page_bytes, src_pages, dst_pages) -> None:
    """ncclSend the pages; the decode replica posts the matching ncclRecv first."""
    import threading

    lib = comm()
    ent = _rccl_peers.get(target)
    if ent is None:
        n = lib.comm_rccl_unique_id_size()
        uid = ctypes.create_string_buffer(n)
        if lib.comm_rccl_unique_id(uid) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        key = uid.raw.hex()
        err: list[Exception] = []

        def join_remote():
            try:
                _post(target + "/kv/rccl_init", {"id": key}, timeout=120.0)
            except Exception as e:  # surfaced after our own init returns or times out
                err.append(e)

        th = threading.Thread(target=join_remote, daemon=True)
        th.start()
        c = lib.comm_rccl_init(uid, 2, 0, pool.device.index or 0)
        th.join()
        if err or not c:
            raise RuntimeError(f"RCCL communicator setup failed: {err[0] if err else 'ncclCommInitRank'}")
        ent = (c, key)
        _rccl_peers[target] = ent
    c, key = ent
    n = len(src_pages)
    _post(target + "/kv/rccl_recv", {"comm": key, "pages": dst_pages, "page_bytes": page_bytes})
    sp = (ctypes.c_int64 * n)(*src_pages)
    stream = torch.cuda.Stream(device=pool.device)
    stream.wait_stream(torch.cuda.current_stream(pool.device))
    rc = lib.comm_rccl_send_pages(c_void_p(c), c_void_p(pool.data_ptr()), page_bytes, sp, n, 1,
                                  c_void_p(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(f"ncclSend failed ({rc})")
    stream.synchronize()


# ------------------------------------------------------------------ decode side (engine thread)
class KVReceiver:
    """Runs inside the decode replica; methods are invoked on the engine thread."""

    def __init__(self, engine):
        self.engine = engine
        self.reserved: dict[int, int] = {}  # hash -> page reserved for an in-flight transfer
        self._landing = None  # device pointer of the landing window (0: could not be set up)
        self._landing_handle = None
        self._free_slots: list[int] = []
        self._slot_of: dict[int, int] = {}  # hash -> landing slot of an in-flight ipc transfer
        self._reserved_at: dict[int, float] = {}  # hash -> time of its reserve
        import os

        # a producer that died between reserve and commit must not keep pages and window slots
        self.reserve_ttl = float(os.environ.get("VPSA_PD_RESERVE_TTL_S", "120"))
        self._rccl: dict[str, int] = {}  # unique id (hex) -> communicator with one prefill replica
        self._rccl_stream = None

    def _landing_window(self, page_bytes: int) -> bool:
        """Allocate and export the landing window on first use; False when there is none."""
        if self._landing is None:
            import os

            self._landing = 0
            # the chart's nixlBufferSize (LMCACHE_NIXL_BUFFER_SIZE, 1 GiB in every reference example:
            # /root/reference/.github/values-10-disagg-prefill.yaml:32) is this window's size
            want = getattr(getattr(self.engine.cfg, "kv_offload", None), "pd_buffer_bytes", 1 << 30)
            if os.environ.get("VPSA_PD_LANDING_MIB"):
                want = int(os.environ["VPSA_PD_LANDING_MIB"]) << 20
            nbytes = ipc_safe_size(int(want))
            lib = comm()
            ptr = lib.comm_dev_alloc(c_size_t(nbytes)) if nbytes >= page_bytes else None
            if ptr:
                buf = ctypes.create_string_buffer(lib.comm_ipc_handle_size())
                if lib.comm_ipc_export(c_void_p(ptr), buf) == 0:
                    self._landing, self._landing_handle = ptr, buf.raw.hex()
                    self._free_slots = list(range(nbytes // page_bytes))
                    log.info("P/D landing window: %d MiB, %d page slots", nbytes >> 20, len(self._free_slots))
                else:
                    lib.comm_dev_free(c_void_p(ptr))
            if not self._landing:
                log.warning("no hipIpc landing window (allocation or export failed): pages arrive by RCCL or TCP")
        return bool(self._landing)

    def _expire(self) -> int:
        """Give back pages and window slots of reservations nobody committed within the TTL."""
        import time

        now = time.time()
        old = [h for h, t in self._reserved_at.items() if now - t > self.reserve_ttl]
        for h in old:
            self._reserved_at.pop(h, None)
            page = self.reserved.pop(h, None)
            if page is not None:
                self.engine.bm.free([page])
            slot = self._slot_of.pop(h, None)
            if slot is not None:
                self._free_slots.append(slot)
        if old:
            log.warning("dropped %d KV transfer reservations older than %.0f s", len(old), self.reserve_ttl)
        return len(old)

    def reserve(self, hashes: list[int], page_bytes: int, peer: dict | None = None) -> dict:
        """``peer``: what the producer said about itself (host_id, transport, gpu); a producer on
        this host with a GPU gets landing-window slots, as many pages as there are free slots."""
        bm = self.engine.bm
        pool = self.engine.runner.pool
        mine = pool.stride(0) * pool.element_size()
        if page_bytes != mine:
            raise ValueError(f"page size mismatch: sender {page_bytes}, receiver {mine}")
        if self._reserved_at:
            self._expire()
        missing = [h for h in hashes if not bm.contains(h) and h not in self.reserved]
        peer = peer or {}
        slots = None
        more = False
        by_ipc = bool(missing and pool.is_cuda and peer.get("gpu") and peer.get("host_id") == host_identity() and
                      peer.get("transport", "auto") in ("auto", "ipc") and self._landing_window(page_bytes))
        if by_ipc and not self._free_slots:
            # other producers fill the window right now: their commits free it within milliseconds
            return {"pages": [], "missing": [], "more": True, "busy": True, "host_id": host_identity(),
                    "gpu": True}
        if by_ipc:
            k = min(len(missing), len(self._free_slots))
            more = k < len(missing)
            missing = missing[:k]
        if missing and not bm.can_allocate(len(missing)):
            raise MemoryError("decode replica has no free KV pages for the transfer")
        if by_ipc:
            slots = [self._free_slots.pop() for _ in missing]
        pages = bm.allocate(len(missing)) if missing else []
        import time

        now = time.time()
        for i, (h, p) in enumerate(zip(missing, pages)):
            self.reserved[h] = p
            self._reserved_at[h] = now
            if slots is not None:
                self._slot_of[h] = slots[i]
        return {"pages": pages, "missing": missing, "more": more,
                "ipc_handle": self._landing_handle if slots is not None else None, "ipc_offset": 0,
                "ipc_slots": slots, "host_id": host_identity(), "gpu": bool(pool.is_cuda)}

    def rccl_init(self, key: str) -> dict:
        """Join the two-rank communicator a prefill replica is creating (blocks until it joined)."""
        pool = self.engine.runner.pool
        if not pool.is_cuda:
            raise ValueError("RCCL transport needs a GPU replica")
        if key not in self._rccl:
            uid = bytes.fromhex(key)
            c = comm().comm_rccl_init(ctypes.create_string_buffer(uid, len(uid)), 2, 1, pool.device.index or 0)
            if not c:
                raise RuntimeError("ncclCommInitRank failed")
            self._rccl[key] = c
        return {"ok": True}

    def rccl_recv(self, key: str, pages: list[int], page_bytes: int) -> dict:
        """Post ncclRecv for ``pages`` on the receive stream; ``commit`` waits for completion."""
        pool = self.engine.runner.pool
        c = self._rccl[key]
        if self._rccl_stream is None:
            self._rccl_stream = torch.cuda.Stream(device=pool.device)
        self._rccl_stream.wait_stream(torch.cuda.current_stream(pool.device))
        n = len(pages)
        dp = (ctypes.c_int64 * n)(*pages)
        rc = comm().comm_rccl_recv_pages(c_void_p(c), c_void_p(pool.data_ptr()), page_bytes, dp, n, 0,
                                         c_void_p(self._rccl_stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"ncclRecv failed ({rc})")
        return {"ok": True}

    def push(self, pages: list[int], raw: bytes) -> dict:
        pool = self.engine.runner.pool
        n = len(pages)
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).view(pool.dtype).view(n, *pool.shape[1:])
        pool[torch.tensor(pages, device=pool.device)] = t.to(pool.device)
        return {"ok": True}

    def commit(self, hashes: list[int], pages: list[int], landed: bool = False) -> dict:
        """Publish transferred pages.  ``landed``: the bytes are in the landing window (the
        producer has synchronised its copy stream); they move into the pool pages first.  Slots
        are returned either way."""
        bm = self.engine.bm
        pool = self.engine.runner.pool
        if any(self.reserved.get(h) != p for h, p in zip(hashes, pages)):
            raise ValueError("commit of pages that are not reserved (the reservation expired?)")
        slots = [self._slot_of.pop(h, None) for h in hashes]
        if landed:
            if any(s is None for s in slots):
                raise ValueError("commit of landed pages without landing slots")
            n = len(pages)
            sp = (ctypes.c_int64 * n)(*slots)
            dp = (ctypes.c_int64 * n)(*pages)
            rc = comm().comm_push_pages(c_void_p(self._landing), c_void_p(pool.data_ptr()),
                                        pool.stride(0) * pool.element_size(), sp, dp, n,
                                        c_void_p(torch.cuda.current_stream(pool.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"landing window copy failed ({rc})")
        if self.engine.runner.device.type == "cuda":
            torch.cuda.synchronize()
        self._free_slots.extend(s for s in slots if s is not None)
        for h, p in zip(hashes, pages):
            self.reserved.pop(h, None)
            self._reserved_at.pop(h, None)
            bm.register(p, h)
        bm.free(pages)  # published pages wait in the prefix cache (evictable, LRU)
        return {"ok": True, "published": len(pages)}


def local_push(src_engine, receiver: "KVReceiver", prompt_ids: list[int], window: bool = False) -> int:
    """Same-process P->D transfer (two engines sharing a device or both on the CPU): the
    reserve/push/commit protocol with the HTTP hops removed.  On a GPU the bytes move with the
    same side-stream page copier the hipIpc data plane uses, so this exercises
    ``comm_push_pages`` without a second process; ``window`` sends them through the receiver's
    landing window, in rounds when it has fewer slots than there are pages (its device pointer
    is used directly: a process cannot open its own hipIpc handle).  Returns the pages moved."""
    from ..engine.block_manager import chain_hashes

    bm = src_engine.bm
    hashes = chain_hashes(prompt_ids)
    n_hit = bm.peek_prefix(hashes)
    hashes = hashes[:n_hit]
    if not hashes:
        return 0
    pages = bm.match_prefix(hashes, count_stats=False)  # takes a reference on each page
    moved = 0
    try:
        spool, dpool = src_engine.runner.pool, receiver.engine.runner.pool
        page_bytes = spool.stride(0) * spool.element_size()
        idx = {h: i for i, h in enumerate(hashes)}
        peer = {"gpu": True, "host_id": host_identity(), "transport": "ipc"} if window else None
        for _ in range(len(hashes) + 2):
            res = receiver.reserve(hashes, page_bytes, peer)
            missing, dst = res["missing"], res["pages"]
            if not missing:
                break
            src = [pages[idx[h]] for h in missing]
            landed = res.get("ipc_slots") is not None
            if window and not landed:
                raise RuntimeError("the receiver offered no landing window")
            if spool.is_cuda:
                n = len(src)
                sp = (ctypes.c_int64 * n)(*src)
                dp = (ctypes.c_int64 * n)(*(res["ipc_slots"] if landed else dst))
                base = receiver._landing if landed else dpool.data_ptr()
                stream = torch.cuda.Stream(device=spool.device)
                stream.wait_stream(torch.cuda.current_stream(spool.device))
                rc = comm().comm_push_pages(c_void_p(spool.data_ptr()), c_void_p(base),
                                            page_bytes, sp, dp, n, c_void_p(stream.cuda_stream))
                if rc != 0:
                    raise RuntimeError(f"comm_push_pages failed ({rc})")
                stream.synchronize()
            else:
                dpool[torch.tensor(dst)] = spool[torch.tensor(src)]
            receiver.commit(missing, dst, landed)
            moved += len(missing)
            if not res.get("more"):
                break
        return moved
    finally:
        bm.free(pages)
