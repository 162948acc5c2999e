"""Prefill -> decode KV page transfer between engine replicas (disaggregated prefill).

Protocol (control plane = the replicas' own HTTP ports, data plane chosen per pair):
  1. producer  POST <decode>/kv/reserve {hashes}      -> {pages, missing, transports...}
     the decode replica allocates HBM pages for the hashes it does not hold yet;
  2. producer  moves the page bytes:
        ipc-direct  the decode replica's KV POOL itself is mapped through a hipIpc handle and the
              pages are written where they will live, with hipMemcpyAsync on a side stream (SDMA
              over the direct xGMI link); commit only publishes.  The decode replica of a pair
              allocates its pool as one device allocation of its own, of a size hipIpc can map
              (hipIpcOpenMemHandle never returns for allocations whose size modulo 4 GiB is 2 GiB
              or more: benchmarks/ipc_probe.py, ``ipc_safe_size``; the page count is rounded down
              accordingly: ``ModelRunner._alloc_exported_pool``).  The producers of a deployment
              are trusted with the whole pool, as NIXL peers are with registered memory,
        ipc   fallback when the pool could not be exported: a LANDING WINDOW (a device allocation
              of LMCACHE_NIXL_BUFFER_SIZE bytes, default 1 GiB) is mapped and written the same way,
              and at commit the decode replica copies window slots into its pool pages on the
              device.  A request with more pages than free slots goes in rounds (reserve -> push
              -> commit, repeated),
        rccl  ncclSend / ncclRecv of whole pages on side streams, over a two-rank RCCL
              communicator created per (prefill, decode) pair (the unique id travels over the
              control plane).  Taken automatically when both replicas have a GPU and the decode
              replica offers no landing window to this producer (another host or container
              namespace, or its hipIpc export failed); VPSA_PD_TRANSPORT=rccl forces it,
        tcp   POST <decode>/kv/push with the raw page bytes (any host, CPU tests);
  3. producer  POST <decode>/kv/commit {hashes, pages} -> pages are published in the decode
     replica's prefix cache; the decode request then finds its prompt resident.

Neither side's step loop waits for a transfer, and a transfer does not wait for the end of the
prompt: the prefill replica hands the pages of every finished prefill CHUNK to the push worker of
their target (pages are immutable once full), so a long prompt's KV leaves while its later chunks
are still being computed and only the last chunk's pages are on the time-to-first-token path.  The
engine thread only takes a reference per page, records a stream event and queues the job
(``PagePusher``: one worker thread and copy stream per decode target, so a dead target delays
nobody else's transfer; a target that failed is skipped for a few seconds); the HTTP handler of
the prefill request awaits the future of the request's LAST job before it answers the router, so
the decode request is sent only when the pages are published.  On the decode replica ``commit`` enqueues the window -> pool copy on a side
stream, a worker thread waits for its event (bounded), and only then are the pages registered in
the prefix cache: whoever finds them there finds complete data, and nothing ever calls a
device-wide synchronise.
Reference behaviour being provided: the NIXL sender/receiver pair configured by
LMCACHE_NIXL_* (/root/reference/helm/templates/deployment-vllm-multi.yaml:267-305) and the
router's two-phase request flow (/root/reference/src/vllm_router/services/request_service/request.py:339-431).
"""

from __future__ import annotations

import ctypes
import json
import logging
import urllib.error
import urllib.request
from ctypes import c_int, c_size_t, c_void_p

import torch

from .. import _native

log = logging.getLogger("vpsa.kvxfer")
_comm = None
_api_key: str | None = None  # bearer token of the deployment's replicas (they share VLLM_API_KEY)
_peer_maps: dict[tuple[str, str], int] = {}  # (target url, handle) -> mapped base of its landing window
_rccl_peers: dict[str, tuple[int, str]] = {}  # target url -> (communicator, key on the decode side)
_rccl_broken: set[str] = set()  # targets whose RCCL setup failed once: not tried again
# Test hook: with a stub comm library installed in ``_comm`` the device code paths (landing
# window, RCCL) run against CPU tensors, so their control flow is covered without a GPU.
_FAKE_DEVICE = False


def _on_device(pool) -> bool:
    return bool(pool.is_cuda or _FAKE_DEVICE)


class _DoneEvent:
    def query(self) -> bool:
        return True

    def synchronize(self) -> None:
        return None


def _new_stream(pool):
    return torch.cuda.Stream(device=pool.device) if pool.is_cuda else None


def _stream_ptr(stream):
    return c_void_p(stream.cuda_stream if stream is not None else 0)


def _record(stream):
    if stream is None:
        return _DoneEvent()
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


def _device_index(pool) -> int:
    return (pool.device.index or 0) if pool.is_cuda else 0


def wait_event(ev, timeout_s: float) -> bool:
    """Host-side bounded wait for a stream event (never on the engine thread)."""
    import time

    deadline = time.time() + timeout_s
    while not ev.query():
        if time.time() > deadline:
            return False
        time.sleep(0.0005)
    return True


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


def _post(url: str, payload: dict, raw: bytes | None = None, timeout: float = 10.0) -> dict:
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
    try:
        with urllib.request.urlopen(req, timeout=timeout) as r:
            body = r.read()
    except urllib.error.HTTPError as e:
        # the peer says why in the body ("KV pool exhausted", "pages are not reserved", ...)
        why = ""
        try:
            d = json.loads(e.read() or b"{}")
            why = d.get("message") or (d.get("error") or {}).get("message", "")
        except Exception:  # noqa: BLE001
            pass
        raise RuntimeError(f"{url.rsplit('/', 1)[-1]}: HTTP {e.code} {why}".rstrip()) from None
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


def target_allowed(target: str) -> bool:
    """A push goes to an http(s) URL; when VPSA_PD_ALLOWED_TARGETS (comma-separated URL prefixes or
    host names) is set, only to one of those.  The router overwrites whatever ``kv_transfer_params``
    a client sent; this is the engine's own line of defence against being told to POST KV bytes and
    the deployment's bearer token somewhere else."""
    import os
    from urllib.parse import urlparse

    try:
        u = urlparse(target)
    except ValueError:
        return False
    if u.scheme not in ("http", "https") or not u.hostname:
        return False
    allow = [x.strip() for x in os.environ.get("VPSA_PD_ALLOWED_TARGETS", "").split(",") if x.strip()]
    if not allow:
        return True
    return any(target.startswith(a) or u.hostname == a or f"{u.hostname}:{u.port}" == a for a in allow)


def push_request_pages(engine, target: str, hashes: list[int], pages: list[int], ready=None,
                       stream=None, deadline_s: float = 30.0) -> str:
    """Producer side: make ``hashes`` resident on the decode replica at ``target``.  Runs on the
    pusher's worker thread (``stream``: its side stream, ``ready``: event recorded on the compute
    stream when the request finished; the copies wait for it on the device).  Every control-plane
    call has a short timeout and the whole push one deadline: a slow or dead decode replica costs
    this request its transfer, never the replica its step loop.  The calls carry this replica's
    API key: the replicas of one deployment are started with the same VLLM_API_KEY
    (/root/reference/helm/templates/deployment-vllm-multi.yaml:216-230), and the decode replica's
    /kv/* endpoints sit behind the same bearer check as its OpenAI endpoints."""
    import os
    import time

    global _api_key
    _api_key = getattr(engine.cfg, "api_key", None) or os.environ.get("VLLM_API_KEY") or None
    runner = engine.runner
    pool = runner.pool
    page_bytes = pool.stride(0) * pool.element_size()
    idx = {h: i for i, h in enumerate(hashes)}
    pref = transport_preference()
    used = "resident"
    on_dev = _on_device(pool)
    if on_dev and stream is None:
        stream = _new_stream(pool)
    if stream is not None:
        if ready is not None:
            stream.wait_event(ready)
        else:
            stream.wait_stream(torch.cuda.current_stream(pool.device))

    t_end = time.time() + deadline_s
    busy_until = time.time() + min(10.0, deadline_s / 2)
    rounds = 0
    while rounds < len(hashes) + 2:  # one round per landing-window fill; bounded in any case
        if time.time() > t_end:
            raise TimeoutError(f"KV push to {target} exceeded {deadline_s:.0f} s")
        res = _post(target + "/kv/reserve", {"hashes": hashes, "page_bytes": page_bytes, "source_rank": _my_rank(),
                                             "host_id": host_identity(), "transport": pref, "gpu": on_dev})
        if res.get("busy"):  # the window is full of other producers' pages
            if time.time() > busy_until:
                pref = "tcp" if pref != "rccl" else pref  # the next reserve asks for pool pages
            time.sleep(0.002)
            continue
        rounds += 1
        dst_pages = res.get("pages", [])
        missing = res.get("missing", [])
        if not missing:
            break
        src = [pages[idx[h]] for h in missing]
        mode = "tcp"
        if on_dev and res.get("ipc_handle") and res.get("ipc_slots") is not None:
            try:
                _push_ipc(pool, target, res["ipc_handle"], res.get("ipc_offset", 0), page_bytes, src,
                          res["ipc_slots"], stream)
                mode = "ipc-direct" if res.get("direct") else "ipc"
            except Exception as e:
                log.warning("ipc push failed (%s); falling back", e)
        # RCCL when forced, or when both sides have a GPU and no landing window came back (another
        # host or namespace, or the export failed): a multi-node deployment uses it by itself
        want_rccl = pref == "rccl" or (pref == "auto" and res.get("ipc_slots") is None)
        if mode == "tcp" and on_dev and res.get("gpu") and want_rccl and target not in _rccl_broken:
            try:
                _push_rccl(pool, target, page_bytes, src, dst_pages, stream)
                mode = "rccl"
            except Exception as e:
                _rccl_broken.add(target)
                log.warning("rccl push failed (%s); this pair uses tcp from now on", e)
        if mode == "tcp":
            chunk = 8
            for i in range(0, len(src), chunk):
                if stream is not None:
                    with torch.cuda.stream(stream):
                        blob = pool[torch.tensor(src[i:i + chunk], device=pool.device)].contiguous().cpu()
                else:
                    blob = pool[torch.tensor(src[i:i + chunk])].contiguous()
                raw = blob.view(torch.uint8).numpy().tobytes()
                _post(target + "/kv/push", {"pages": dst_pages[i:i + chunk], "page_bytes": page_bytes}, raw)
        _post(target + "/kv/commit", {"hashes": missing, "pages": dst_pages, "landed": mode == "ipc",
                                      "rccl": mode == "rccl"}, timeout=15.0)
        used = mode
        if not res.get("more"):
            break
    return used


class PagePusher:
    """Moves a prefill replica's pages to their decode replicas.  ``submit`` is called on the
    engine thread, as often per request as it has new full pages (every finished prefill chunk,
    then once more with ``final`` when the request is over), and returns at once with the
    request's future, which completes when the request's LAST job is over.  One worker thread
    with a copy stream of its own per decode target: a target that does not answer costs the
    pushes to that target their deadline and nobody else anything, and is skipped for
    ``COOLDOWN_S`` after a failure (the decode replica recomputes what did not arrive).  Pages
    stay referenced until their job is over; ``drain_released`` (engine thread again) gives the
    references back."""

    COOLDOWN_S = 5.0

    def __init__(self, engine):
        import queue
        import threading

        self.engine = engine
        self._released: "queue.Queue" = queue.Queue()
        self._lock = threading.Lock()
        self._targets: dict[str, "queue.Queue"] = {}
        self._down_until: dict[str, float] = {}
        self._req: dict[str, dict] = {}  # request id -> progress of its transfer

    def submit(self, request_id: str, target: str, hashes: list[int], pages: list[int], final: bool = True):
        import queue
        import threading
        import time
        from concurrent.futures import Future

        pool = self.engine.runner.pool
        bm = self.engine.bm
        bm.pin(pages)  # the request may give its pages back before the copy has read them
        ready = None
        if pool.is_cuda and pages:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(pool.device))
        with self._lock:
            st = self._req.get(request_id)
            if st is None:
                st = self._req[request_id] = {"fut": Future(), "t0": time.time(), "pages": 0, "error": None,
                                              "mode": "resident", "jobs": 0}
            st["jobs"] += 1
            if final:
                st["t_final"] = time.time()
            q = self._targets.get(target)
            if q is None:
                q = self._targets[target] = queue.Queue()
                threading.Thread(target=self._loop, args=(target, q), name="kv-push", daemon=True).start()
        q.put((request_id, list(hashes), list(pages), ready, final))
        return st["fut"]

    def pending(self, request_id: str) -> bool:
        with self._lock:
            return request_id in self._req

    def drain_released(self) -> None:
        """Engine thread: give back the page references of finished pushes."""
        import queue

        while True:
            try:
                pages = self._released.get_nowait()
            except queue.Empty:
                return
            self.engine.bm.free(pages)

    def _loop(self, target: str, jobs) -> None:
        import time

        pool = self.engine.runner.pool
        stream = None
        if pool.is_cuda:
            from .. import ops

            ops.ensure_device()
            torch.cuda.set_device(pool.device)
            stream = torch.cuda.Stream(device=pool.device)
        while True:
            job = jobs.get()
            if job is None:
                return
            request_id, hashes, pages, ready, final = job
            with self._lock:
                st = self._req.get(request_id)
            try:
                if hashes and st is not None and st["error"] is None:
                    if time.time() < self._down_until.get(target, 0.0):
                        raise ConnectionError(f"decode replica {target} failed a moment ago: transfer skipped")
                    mode = push_request_pages(self.engine, target, hashes, pages, ready, stream)
                    st["pages"] += len(pages)
                    if mode != "resident":
                        st["mode"] = mode
            except BaseException as e:  # the future must always complete
                self._down_until[target] = time.time() + self.COOLDOWN_S
                if st is not None:
                    st["error"] = e
            finally:
                self._released.put(pages)
            if final and st is not None:
                with self._lock:
                    self._req.pop(request_id, None)
                if st["error"] is not None:
                    st["fut"].set_exception(st["error"])
                else:
                    now = time.time()
                    # tail_seconds: from the request's end to the publication of its last pages,
                    # the part of the transfer that the time to first token pays for
                    st["fut"].set_result({"mode": st["mode"], "pages": st["pages"], "jobs": st["jobs"],
                                          "seconds": now - st["t0"], "tail_seconds": now - st.get("t_final", st["t0"])})

    def close(self) -> None:
        with self._lock:
            for q in self._targets.values():
                q.put(None)


def _my_rank() -> int:
    import torch.distributed as dist

    return dist.get_rank() if dist.is_available() and dist.is_initialized() else -1


def _push_ipc(pool, target, handle_hex, offset, page_bytes, src_pages, dst_slots, stream=None) -> None:
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
    if stream is None and pool.is_cuda:
        stream = _new_stream(pool)
        stream.wait_stream(torch.cuda.current_stream(pool.device))
    rc = lib.comm_push_pages(c_void_p(pool.data_ptr()), c_void_p(base + offset), page_bytes,
                             sp, dp, n, _stream_ptr(stream))
    if rc != 0:
        raise RuntimeError(f"comm_push_pages failed ({rc})")
    if stream is not None:
        stream.synchronize()  # the caller's thread, never the engine's


def _push_rccl(pool, target, page_bytes, src_pages, dst_pages, stream=None) -> None:
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
                _post(target + "/kv/rccl_init", {"id": key}, timeout=60.0)
            except Exception as e:  # surfaced after our own init returns or times out
                err.append(e)

        th = threading.Thread(target=join_remote, daemon=True)
        th.start()
        c = lib.comm_rccl_init(uid, 2, 0, _device_index(pool))
        th.join()
        if err or not c:
            raise RuntimeError(f"RCCL communicator setup failed: {err[0] if err else 'ncclCommInitRank'}")
        ent = (c, key)
        _rccl_peers[target] = ent
    c, key = ent
    n = len(src_pages)
    # recv before send: the decode replica has its ncclRecv calls enqueued when this returns
    _post(target + "/kv/rccl_recv", {"comm": key, "pages": dst_pages, "page_bytes": page_bytes})
    sp = (ctypes.c_int64 * n)(*src_pages)
    if stream is None and pool.is_cuda:
        stream = _new_stream(pool)
        stream.wait_stream(torch.cuda.current_stream(pool.device))
    rc = lib.comm_rccl_send_pages(c_void_p(c), c_void_p(pool.data_ptr()), page_bytes, sp, n, 1,
                                  _stream_ptr(stream))
    if rc != 0:
        raise RuntimeError(f"ncclSend failed ({rc})")
    if stream is not None and not wait_event(_record(stream), 20.0):
        raise TimeoutError("ncclSend did not complete within 20 s")


# ------------------------------------------------------------------ decode side (engine thread)
class KVReceiver:
    """Runs inside the decode replica; methods are invoked on the engine thread (except
    ``rccl_init``, which blocks in the communicator rendezvous on a worker thread)."""

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
        self._rccl_streams: dict[str, object] = {}  # communicator key -> its receive stream
        self._rccl_events: dict[tuple, tuple] = {}  # pages of a posted receive -> (event, communicator key)
        # pages of receives that never completed: a late sender may still write them, so they stay
        # out of the allocator until their event fires (or for good, with a dead sender)
        self._quarantine: list[tuple[object, list[int]]] = []
        self._copy_stream = None  # window -> pool copies

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

    def _require_reserved(self, pages: list[int]) -> None:
        """Page indices in a push / receive / commit come from another process: they must be pages
        this receiver handed out and still holds for a transfer.  Anything else (a reservation
        that expired and whose page now belongs to a running request, an index outside the pool,
        a client making things up) is refused before a byte is written."""
        live = set(self.reserved.values())
        bad = [p for p in pages if not isinstance(p, int) or isinstance(p, bool) or p not in live]
        if bad:
            raise ValueError(f"pages {bad[:4]} are not reserved for a transfer (the reservation expired?)")

    def reserve(self, hashes: list[int], page_bytes: int, peer: dict | None = None) -> dict:
        """``peer``: what the producer said about itself (host_id, transport, gpu); a producer on
        this host with a GPU gets landing-window slots, as many pages as there are free slots."""
        bm = self.engine.bm
        pool = self.engine.runner.pool
        if not isinstance(pool, torch.Tensor):
            raise ValueError("this replica keeps its KV pool in layer groups and cannot take whole pages: "
                             "start it with its --kv-transfer-config (or --kv-layer-groups 1)")
        mine = pool.stride(0) * pool.element_size()
        if page_bytes != mine:
            raise ValueError(f"page size mismatch: sender {page_bytes}, receiver {mine}")
        if not isinstance(hashes, list) or not all(isinstance(h, int) and not isinstance(h, bool) for h in hashes):
            raise ValueError("hashes: a list of integers")
        if self._reserved_at:
            self._expire()
        if self._quarantine:
            self._sweep_quarantine()
        missing = [h for h in hashes if not bm.contains(h) and h not in self.reserved]
        peer = peer or {}
        slots = None
        more = False
        on_dev = _on_device(pool)
        same_host_gpu = bool(missing and on_dev and peer.get("gpu") and peer.get("host_id") == host_identity()
                             and peer.get("transport", "auto") in ("auto", "ipc"))
        pool_handle = getattr(self.engine.runner, "pool_ipc_handle", None)
        if same_host_gpu and pool_handle:
            # the pool itself is mapped by the producer: every page lands where it will live
            if not bm.can_allocate(len(missing)):
                raise MemoryError("decode replica has no free KV pages for the transfer")
            pages = bm.allocate(len(missing))
            import time

            now = time.time()
            for h, p in zip(missing, pages):
                self.reserved[h] = p
                self._reserved_at[h] = now
            return {"pages": pages, "missing": missing, "more": False, "ipc_handle": pool_handle,
                    "ipc_offset": 0, "ipc_slots": pages, "direct": True, "host_id": host_identity(), "gpu": True}
        by_ipc = bool(same_host_gpu and self._landing_window(page_bytes))
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
                "ipc_slots": slots, "host_id": host_identity(), "gpu": on_dev}

    def rccl_init(self, key: str) -> dict:
        """Join the two-rank communicator a prefill replica is creating (blocks until it joined)."""
        pool = self.engine.runner.pool
        if not _on_device(pool):
            raise ValueError("RCCL transport needs a GPU replica")
        if key not in self._rccl:
            uid = bytes.fromhex(key)
            c = comm().comm_rccl_init(ctypes.create_string_buffer(uid, len(uid)), 2, 1, _device_index(pool))
            if not c:
                raise RuntimeError("ncclCommInitRank failed")
            self._rccl[key] = c
        return {"ok": True}

    def rccl_recv(self, key: str, pages: list[int], page_bytes: int) -> dict:
        """Post ncclRecv for ``pages`` on the receive stream; ``commit`` waits (bounded, off the
        engine thread) for their completion event."""
        pool = self.engine.runner.pool
        if page_bytes != pool.stride(0) * pool.element_size():
            raise ValueError("page size mismatch")
        self._require_reserved(pages)
        c = self._rccl[key]
        # a stream per communicator: a receive that never completes (dead sender) blocks only the
        # transfers of that one prefill replica
        stream = self._rccl_streams.get(key)
        if stream is None and pool.is_cuda:
            stream = self._rccl_streams[key] = _new_stream(pool)
        n = len(pages)
        dp = (ctypes.c_int64 * n)(*pages)
        rc = comm().comm_rccl_recv_pages(c_void_p(c), c_void_p(pool.data_ptr()), page_bytes, dp, n, 0,
                                         _stream_ptr(stream))
        if rc != 0:
            raise RuntimeError(f"ncclRecv failed ({rc})")
        self._rccl_events[tuple(pages)] = (_record(stream), key)
        return {"ok": True}

    def push(self, pages: list[int], raw: bytes) -> dict:
        """TCP data plane: page bytes arrive in the request body."""
        pool = self.engine.runner.pool
        n = len(pages)
        page_bytes = pool.stride(0) * pool.element_size()
        self._require_reserved(pages)
        if len(raw) != n * page_bytes:
            raise ValueError(f"payload of {len(raw)} bytes for {n} pages of {page_bytes}")
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).view(pool.dtype).view(n, *pool.shape[1:])
        pool[torch.tensor(pages, device=pool.device)] = t.to(pool.device, non_blocking=False)
        return {"ok": True}

    # commit = begin (engine thread: validate, enqueue the window -> pool copy on a side stream)
    #        -> wait for the returned event (a worker thread, bounded: ``wait_event``)
    #        -> finish (engine thread: publish) or abort (engine thread: drop the reservation)
    def commit_begin(self, hashes: list[int], pages: list[int], landed: bool = False, rccl: bool = False):
        """Returns the event that says the bytes are in the pool pages (None: they already are)."""
        pool = self.engine.runner.pool
        if len(hashes) != len(pages) or any(self.reserved.get(h) != p for h, p in zip(hashes, pages)):
            raise ValueError("commit of pages that are not reserved (the reservation expired?)")
        if landed:
            slots = [self._slot_of.get(h) for h in hashes]
            if any(s is None for s in slots):
                raise ValueError("commit of landed pages without landing slots")
            n = len(pages)
            sp = (ctypes.c_int64 * n)(*slots)
            dp = (ctypes.c_int64 * n)(*pages)
            if self._copy_stream is None and pool.is_cuda:
                self._copy_stream = _new_stream(pool)
            # the producer synchronised its copy stream before it sent the commit: the window
            # holds the bytes; the pool pages are reserved, nothing else touches them
            rc = comm().comm_push_pages(c_void_p(self._landing), c_void_p(pool.data_ptr()),
                                        pool.stride(0) * pool.element_size(), sp, dp, n,
                                        _stream_ptr(self._copy_stream))
            if rc != 0:
                raise RuntimeError(f"landing window copy failed ({rc})")
            return _record(self._copy_stream)
        if rccl:
            ent = self._rccl_events.get(tuple(pages))
            if ent is None:
                raise ValueError("commit of an RCCL transfer whose receive was never posted")
            return ent[0]
        return None

    def commit_finish(self, hashes: list[int], pages: list[int]) -> dict:
        bm = self.engine.bm
        if any(self.reserved.get(h) != p for h, p in zip(hashes, pages)):
            raise ValueError("commit of pages that are not reserved (the reservation expired?)")
        for h, p in zip(hashes, pages):
            slot = self._slot_of.pop(h, None)
            if slot is not None:
                self._free_slots.append(slot)
            self.reserved.pop(h, None)
            self._reserved_at.pop(h, None)
            bm.register(p, h)
        self._rccl_events.pop(tuple(pages), None)
        bm.free(pages)  # published pages wait in the prefix cache (evictable, LRU)
        return {"ok": True, "published": len(pages)}

    def _sweep_quarantine(self) -> None:
        keep = []
        for ev, pages in self._quarantine:
            if ev.query():
                self.engine.bm.free(pages)
            else:
                keep.append((ev, pages))
        self._quarantine = keep

    def commit_abort(self, hashes: list[int]) -> None:
        """The bytes never arrived (dead sender): give pages and slots back unpublished.  Pages
        with an RCCL receive still posted on them are NOT handed back: the receive may yet
        complete and write them.  They are quarantined until its event fires, and that
        communicator (and its stream, stuck behind the receive) is dropped, so the prefill
        replica's next transfer sets up a new one or takes another transport."""
        pages = [self.reserved.get(h) for h in hashes]
        stuck = None
        for key_pages, (ev, ckey) in list(self._rccl_events.items()):
            if set(key_pages) & set(p for p in pages if p is not None) and not ev.query():
                stuck = (ev, ckey)
                self._rccl_events.pop(key_pages, None)
        held: list[int] = []
        for h in hashes:
            self._reserved_at.pop(h, None)
            page = self.reserved.pop(h, None)
            if page is not None:
                if stuck is not None:
                    held.append(page)
                else:
                    self.engine.bm.free([page])
            slot = self._slot_of.pop(h, None)
            if slot is not None:
                self._free_slots.append(slot)
        if stuck is not None:
            self._quarantine.append((stuck[0], held))
            self._rccl.pop(stuck[1], None)  # not destroyed: ncclCommDestroy would wait for the stuck receive
            self._rccl_streams.pop(stuck[1], None)
            log.error("RCCL receive of %d pages never completed: pages quarantined, communicator dropped", len(held))

    def commit(self, hashes: list[int], pages: list[int], landed: bool = False, rccl: bool = False) -> dict:
        """All three stages on the calling thread (same-process transfers and tests; the HTTP
        handler runs the wait on a worker thread instead)."""
        ev = self.commit_begin(hashes, pages, landed, rccl)
        if ev is not None and not wait_event(ev, 20.0):
            self.commit_abort(hashes)
            raise TimeoutError("transferred KV pages did not arrive within 20 s")
        return self.commit_finish(hashes, pages)


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
            if _on_device(spool):
                n = len(src)
                sp = (ctypes.c_int64 * n)(*src)
                dp = (ctypes.c_int64 * n)(*(res["ipc_slots"] if landed else dst))
                base = receiver._landing if landed else dpool.data_ptr()
                stream = _new_stream(spool)
                if stream is not None:
                    stream.wait_stream(torch.cuda.current_stream(spool.device))
                rc = comm().comm_push_pages(c_void_p(spool.data_ptr()), c_void_p(base),
                                            page_bytes, sp, dp, n, _stream_ptr(stream))
                if rc != 0:
                    raise RuntimeError(f"comm_push_pages failed ({rc})")
                if stream is not None:
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
