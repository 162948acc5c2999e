"""KV connector of an engine replica: everything that happens to KV pages outside HBM.

Implements the contract the reference configures through ``--kv-transfer-config`` and the
LMCACHE_* environment (/root/reference/helm/templates/deployment-vllm-multi.yaml:182-195,
256-344) with MI355X-native machinery:

  * write-through offload of every newly filled page to pinned host DRAM on a side stream
    (``LMCACHE_LOCAL_CPU`` / ``LMCACHE_MAX_LOCAL_CPU_SIZE``), spill to local disk
    (``LMCACHE_MAX_LOCAL_DISK_SIZE``) and to the shared remote cache server
    (``LMCACHE_REMOTE_URL`` + ``LMCACHE_REMOTE_SERDE`` naive | cachegen);
  * prefix lookup across those tiers when HBM misses, with asynchronous reload fenced by a
    GPU-side event (the engine thread never blocks on a copy);
  * KV controller reporting (``LMCACHE_ENABLE_CONTROLLER`` ...): admitted / evicted page
    hashes, so the router's kvaware logic can find the replica holding a prompt;
  * disaggregated prefill (``kv_role`` kv_producer / kv_consumer): a finished prompt's pages
    are pushed into the decode replica's pool and published in its prefix cache, so the
    decode request simply finds its prompt resident.
"""

from __future__ import annotations

import json
import logging
import queue
import socket
import threading
import time
import urllib.request

import torch

from .. import ops
from ..engine.block_manager import PAGE
from .remote import RemoteKVClient
from .serde import PageSerde
from .tiers import DiskKVStore, HostKVPool

log = logging.getLogger("vpsa.kv")


class KVConnector:
    def __init__(self, engine, off, kvt):
        self.engine = engine
        self.off = off
        self.kvt = kvt
        self.role = kvt.kv_role if kvt else "kv_both"
        r = engine.runner
        mc = engine.model_cfg
        self.pool = r.pool
        self.device = r.device
        self.serde = PageSerde(mc.num_layers, r.n_kv_local, r.kv_head_dim, r.dtype, off.remote_serde)
        # Pages of the lower tiers are keyed by the token chain hash, which says nothing about the
        # weights that produced them.  Everything that leaves this process (remote server keys,
        # disk files) is therefore namespaced by the model identity and the shard of it this rank
        # holds: two models with the same tokenizer and KV geometry sharing one cache server (a
        # base model and a fine-tune, a multi-model stack) can never load each other's pages.
        # (The reference's LMCache keys carry model name and world size for the same reason.)
        self.namespace = self._model_namespace(engine, r, mc)
        self.disk = None
        if off.max_local_disk_gb > 0:
            import os

            self.disk = DiskKVStore(os.path.join(off.local_disk_path, self.namespace),
                                    int(off.max_local_disk_gb * (1 << 30)),
                                    PageSerde(mc.num_layers, r.n_kv_local, r.kv_head_dim, r.dtype, "naive"))
        self.host = None
        if off.local_cpu or off.max_local_cpu_gb > 0:
            cap = int((off.max_local_cpu_gb or 5.0) * (1 << 30))
            self.host = HostKVPool(self.pool, cap, on_spill=self._spill)
            log.info("host KV tier: %d slots of %.1f MiB (%.1f GiB%s)", self.host.n_slots,
                     self.host.page_bytes / (1 << 20), self.host.n_slots * self.host.page_bytes / (1 << 30),
                     " pinned" if self.pool.is_cuda else "")
        if self.disk is not None:
            log.info("disk KV tier: %.1f GiB under %s", off.max_local_disk_gb, self.disk.root
                     if hasattr(self.disk, "root") else off.local_disk_path)
        self.remote = RemoteKVClient(off.remote_url) if off.remote_url else None
        if self.remote is not None:
            log.info("remote KV tier: %s (keys under %s/)", off.remote_url, self.namespace)
        self.cfg_remote_url = off.remote_url
        self.remote_known = _BoundedSet(1_000_000)  # what the cache server is believed to hold
        self.remote_dropped = 0  # fresh pages given up on because the server stopped draining the ring
        self.loaded_from = {"host": 0, "disk": 0, "remote": 0}  # pages brought back into HBM, by tier
        # Raw (naive) pages of a GPU replica travel through a ring of pinned staging slots:
        # device -> slot on a side stream, slot -> socket on the background thread (and the
        # other way round on load), so the step loop never waits for a host copy or a send.
        self._stage = None
        self._pool_exec = None  # worker threads of the codec path, created on first use
        self._inflight: set[int] = set()  # hashes whose upload has been queued but not finished
        if (self.remote is not None and self.pool.is_cuda and self.serde.kind == 0
                and ops.page_numel(self.pool) == self.serde.page_elems):
            n_stage = 128  # 1 GiB of pinned memory at 8 MiB per page
            self._stage = torch.empty((n_stage, self.serde.page_elems), dtype=self.pool.dtype, pin_memory=True)
            self._stage_free: queue.Queue = queue.Queue()
            for i in range(n_stage):
                self._stage_free.put(i)
            self._remote_stream = torch.cuda.Stream(device=self.device)
        self._pending: list[tuple[object, list[int]]] = []  # (event, page ids holding a ref)
        self._pusher = None  # P->D push worker, created with the first transfer
        self._push_futures: dict = {}  # request id -> future of its transfer
        self._push_born: dict = {}  # request id -> when that future was created (unclaimed ones are swept)
        self._bg: queue.Queue = queue.Queue()
        self._bg_thread = threading.Thread(target=self._bg_loop, name="kv-bg", daemon=True)
        self._bg_thread.start()
        self._closed = False
        # controller reporting
        self.instance_id = off.instance_id or f"{socket.gethostname()}:{engine.cfg.port}"
        self.controller_url = None
        if off.enable_controller and off.controller_url:
            u = off.controller_url
            self.controller_url = u if u.startswith("http") else f"http://{u}"
            self._report({"op": "register", "instance_id": self.instance_id, "url": self._self_url(),
                          "ip": self._self_ip(), "model": engine.cfg.model_name})
        m = engine.metrics
        self.metrics = m

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _model_namespace(engine, runner, mc) -> str:
        import hashlib

        from ..parallel.state import get_pp, get_tp

        tp, pp = get_tp(), get_pp()
        ident = "|".join(str(x) for x in (
            engine.cfg.model, engine.cfg.model_name, runner.dtype, mc.num_layers, mc.hidden_size,
            runner.kv_head_dim if hasattr(runner, "kv_head_dim") else mc.head_dim, f"tp{tp.rank}of{tp.size}", f"pp{pp.rank}of{pp.size}",
            getattr(engine.cfg, "kv_namespace", "") or "",
            "page-layout-2"))  # in-page K/V order (blocks of 32 channels / 32 keys): pages written under
        #                        another order must never be read back
        return hashlib.sha1(ident.encode()).hexdigest()[:12]

    def _key(self, h: int) -> bytes:
        return RemoteKVClient.key(h, self.namespace + "/")

    def _self_ip(self) -> str:
        import os

        return os.environ.get("POD_IP") or "127.0.0.1"

    def _self_url(self) -> str:
        host = self._self_ip() if self.engine.cfg.host in ("0.0.0.0", "::") else self.engine.cfg.host
        return f"http://{host}:{self.engine.cfg.port}"

    def _bg_loop(self) -> None:
        while True:
            job = self._bg.get()
            if job is None:
                return
            try:
                job()
            except Exception as e:
                log.warning("kv background job failed: %s", e)

    def _report(self, payload: dict) -> None:
        if not self.controller_url:
            return
        data = json.dumps(payload).encode()

        def send():
            req = urllib.request.Request(self.controller_url + "/report", data=data,
                                         headers={"Content-Type": "application/json"})
            urllib.request.urlopen(req, timeout=5).read()

        self._bg.put(send)

    def _spill(self, h: int, page_cpu: torch.Tensor) -> None:
        """Host-tier eviction: keep the page on disk if that tier exists."""
        if self.disk is not None:
            self.disk.put(h, page_cpu.clone())
        elif self.controller_url and not (self.remote and h in self.remote_known):
            self._report({"instance_id": self.instance_id, "evict": [h]})

    # ------------------------------------------------------------------ scheduler hooks
    def lookup(self, hashes: list[int]) -> int:
        """Consecutive leading pages of ``hashes`` available below HBM."""
        n = 0
        unknown: list[int] = []
        for h in hashes:
            if (self.host is not None and self.host.contains(h)) or \
                    (self.disk is not None and self.disk.contains(h)) or h in self.remote_known:
                n += 1
                continue
            unknown = hashes[n:]
            break
        if unknown and self.remote is not None:
            try:
                flags = self.remote.exists_many([self._key(h) for h in unknown])
            except Exception as e:
                log.warning("remote KV lookup failed: %s", e)
                flags = []
            for h, ok in zip(unknown, flags):
                if not ok:
                    break
                self.remote_known.add(h)
                n += 1
        self.metrics.kv_req_tokens.inc(len(hashes) * PAGE)
        self.metrics.kv_hit_tokens.inc(n * PAGE)
        return n

    def tier_stats(self) -> dict:
        """What the lower tiers hold and have served (for ``GET /debug/stats``)."""
        out: dict = {"pages_loaded_from": dict(self.loaded_from)}
        if self.host is not None:
            out["host"] = {"pages": len(self.host.index), "slots": self.host.n_slots, "stores": self.host.stores,
                           "loads": self.host.loads, "evictions": self.host.evictions}
        if self.disk is not None:
            out["disk"] = {"pages": len(self.disk.index), "bytes": self.disk.bytes}
        if self.remote is not None:
            out["remote"] = {"url": self.cfg_remote_url, "pages_known": len(self.remote_known),
                             "uploads_in_flight": len(self._inflight), "pages_dropped": self.remote_dropped,
                             "serde": "cachegen" if self.serde.kind else "naive"}
        return out

    def set_pool(self, pool) -> None:
        """The runner replaced its pool tensor (sleep: an empty one; wake-up: a new allocation)."""
        self.pool = pool
        if self.host is not None:
            self.host.pool = pool

    STAGE_WAIT_S = 2.0  # how long a download waits for a staging slot before it goes page by page

    def load_pages(self, hashes: list[int], pages: list[int]):
        """Start loading ``hashes`` into HBM ``pages``.  Returns ``(fence, n_loaded)``: an
        event-like object the runner fences the compute stream on (None when the data is already
        in place) and the number of LEADING pages that were actually found.  A page that has left
        the lower tiers since ``lookup`` saw it (the cache server is a bounded LRU and may have
        restarted; a disk file may be gone), a server that does not answer and a blob that does
        not decode are all cache misses: the caller recomputes from the first missing page on,
        this method never raises."""
        t0 = time.time()
        items = list(zip(hashes, pages))
        idx_of = {h: i for i, h in enumerate(hashes)}
        first_bad = len(items)

        def bad(h: int, why) -> None:
            nonlocal first_bad
            self.remote_known.discard(h)
            first_bad = min(first_bad, idx_of[h])
            log.warning("KV page %016x is gone from the lower tiers (%s): recomputing from there", h, why)

        from_host, slow = [], []
        for h, p in items:
            if self.host is not None and self.host.contains(h):
                from_host.append((h, p))
            else:
                slow.append((h, p))
        bulk_done: list[tuple[int, int]] = []  # fetched in bulk; copied to the host tier once first_bad is known
        try:
            if self._stage is not None and slow:
                on_disk = [(h, p) for h, p in slow if self.disk is not None and self.disk.contains(h)]
                fetched = [(h, p) for h, p in slow if (h, p) not in set(on_disk)]
                missing = self._download_staged(fetched)
                self.loaded_from["remote"] += len(fetched) - len(missing)
                bulk_done = [x for x in fetched if x not in set(missing)]
                slow = on_disk + missing
            elif self.remote is not None and self.serde.kind != 0 and slow:
                on_disk = [(h, p) for h, p in slow if self.disk is not None and self.disk.contains(h)]
                fetched = [x for x in slow if x not in set(on_disk)]
                missing = self._download_codec(fetched)
                self.loaded_from["remote"] += len(fetched) - len(missing)
                bulk_done = [x for x in fetched if x not in set(missing)]
                slow = on_disk + missing
        except Exception as e:  # connection lost half way: what was not confirmed is looked at one by one
            log.warning("bulk KV download failed: %s", e)
        for h, p in sorted(slow, key=lambda x: idx_of[x[0]]):
            if idx_of[h] >= first_bad:
                break  # nothing behind the first missing page can be used
            # disk / remote: decode on this thread, then a plain H2D copy
            t = None
            try:
                t = self.disk.get(h, self.device) if self.disk is not None and self.disk.contains(h) else None
                src = "disk"
                if t is None and self.remote is not None:
                    src = "remote"
                    blob = self.remote.get(self._key(h))
                    for _ in range(100):  # our own upload of this page may still be on its way
                        if blob is not None or h not in self._inflight:
                            break
                        time.sleep(0.05)
                        blob = self.remote.get(self._key(h))
                    if blob is not None:
                        t = self.serde.decode(blob, self.device)
            except Exception as e:
                bad(h, e)
                continue
            if t is None:
                bad(h, "not found")
                continue
            self.loaded_from[src] += 1
            ops.page_assign(self.pool, p, t.view(-1))
            bulk_done.append((h, p))
        # Only pages in front of the first missing one stay with the request: the scheduler frees
        # the others at once and may hand them to somebody else while a copy to the host tier is
        # still queued, which would file wrong bytes under a valid hash.  The kept pages are
        # pinned until their copy has landed, like freshly computed ones.
        self._host_store_pinned([(h, p) for h, p in bulk_done if idx_of[h] < first_bad])
        from_host = [(h, p) for h, p in from_host if idx_of[h] < first_bad]
        ev = self.host.load(from_host) if from_host else None
        self.loaded_from["host"] += len(from_host)
        dt = max(time.time() - t0, 1e-6)
        self.metrics.kv_retrieve_speed.observe(first_bad * PAGE / dt)
        if ev is None:
            return None, first_bad
        return _ComputeFence(self.host, ev), first_bad

    def _host_store_pinned(self, pairs: list[tuple[int, int]]) -> None:
        """Copy (hash, page) pairs to the host tier; the HBM pages keep a reference until the
        (asynchronous) copy has read them."""
        if self.host is None or not pairs:
            return
        ev = self.host.store(pairs)
        if ev is not None:
            pages = [p for _, p in pairs]
            self.engine.bm.pin(pages)
            self._pending.append((ev, pages))

    def store_pages(self, fresh: list[tuple[int, int]]) -> None:
        """Write-through of newly filled pages (hash, page)."""
        t0 = time.time()
        if self.host is not None:
            self._host_store_pinned(fresh)
            self.metrics.kv_local_usage.set(self.host.used_bytes())
            self.metrics.cpu_cache_usage.set(len(self.host.index) / self.host.n_slots)
        if self.remote is not None and self._stage is not None:
            self._upload_staged([(h, p) for h, p in fresh if h not in self.remote_known])
        elif self.remote is not None:
            for h, p in fresh:
                if h in self.remote_known:
                    continue
                self.remote_known.add(h)
                page = ops.page_tensor(self.pool, p)
                if self.serde.uses_codec(page):
                    # quantise on the GPU here; the range coder (tens of ms per page) and the
                    # upload run on the worker pool
                    sym, sc = self.serde.quantize(page.reshape(-1))
                    self._inflight.add(h)
                    self._workers().submit(self._pack_and_put, h, sym, sc)
                else:
                    blob = self.serde.encode(page)
                    self._bg.put(lambda k=self._key(h), b=blob: self.remote.put(k, b))
        elif self.disk is not None and self.host is None:
            for h, p in fresh:
                self.disk.put(h, ops.page_tensor(self.pool, p).cpu())
        if self.controller_url:
            self._report({"instance_id": self.instance_id, "admit": [h for h, _ in fresh],
                          "url": self._self_url(), "ip": self._self_ip(),
                          "model": self.engine.cfg.model_name})
        dt = max(time.time() - t0, 1e-6)
        self.metrics.kv_store_speed.observe(len(fresh) * PAGE / dt)
        self.poll()

    def _workers(self):
        if self._pool_exec is None:
            import os
            from concurrent.futures import ThreadPoolExecutor

            n = max(2, min(16, (os.cpu_count() or 4) // 2))
            self._pool_exec = ThreadPoolExecutor(max_workers=n, thread_name_prefix="kv-codec")
        return self._pool_exec

    def _pack_and_put(self, h: int, sym, sc) -> None:
        try:
            if not self.remote.put(self._key(h), self.serde.pack(sym, sc)):
                self.remote_known.discard(h)
        except Exception as e:
            self.remote_known.discard(h)
            log.warning("remote KV upload of %016x failed: %s", h, e)
        finally:
            self._inflight.discard(h)

    def _fetch_unpacked(self, h: int):
        blob = self.remote.get(self._key(h))
        return None if blob is None else self.serde.unpack(blob)

    def _download_codec(self, items: list[tuple[int, int]]) -> list[tuple[int, int]]:
        """Remote pages in codec form: fetch + entropy-decode on the worker pool (in page
        order, so the GPU half of page i overlaps the host half of the following pages)."""
        left: list[tuple[int, int]] = []
        futs = [self._workers().submit(self._fetch_unpacked, h) for h, _ in items]
        for (h, p), f in zip(items, futs):
            got = f.result()
            if got is None:
                left.append((h, p))
            elif got[0] == "raw":
                t = torch.from_numpy(got[1]).view(self.pool.dtype)
                ops.page_assign(self.pool, p, t.to(self.device))
            else:
                ops.page_assign(self.pool, p, self.serde.restore(got, self.device).view(-1))
        return left

    def _upload_staged(self, todo: list[tuple[int, int]]) -> None:
        """Raw pages to the remote server without blocking the step loop."""
        taken: list[tuple[int, int, int]] = []
        for i, (h, p) in enumerate(todo):
            try:
                # a full ring means the uploads are behind the prefill: wait for a slot (a
                # dropped page would cut the hash chain and orphan every later page)
                taken.append((h, p, self._stage_free.get(timeout=10.0)))
            except queue.Empty:
                self.remote_dropped += len(todo) - i
                break
        if not taken:
            return
        st = self._remote_stream
        st.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(st):
            for _h, p, slot in taken:
                ops.page_read_into(self.pool, p, self._stage[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        pages = [p for _, p, _ in taken]
        self.engine.bm.pin(pages)  # the HBM pages must stay put until the side-stream copy has read them
        self._pending.append((_TorchEvent(ev), pages))
        for h, _, _ in taken:
            self.remote_known.add(h)
            self._inflight.add(h)
        header = self.serde._header(self.serde.page_elems * self._stage.element_size())

        def send(taken=taken, ev=ev):
            ev.synchronize()
            for h, _p, slot in taken:
                try:
                    view = memoryview(self._stage[slot].view(torch.uint8).numpy())
                    if not self.remote.put_parts(self._key(h), header, view):
                        self.remote_known.discard(h)
                except Exception:
                    self.remote_known.discard(h)
                    raise
                finally:
                    self._inflight.discard(h)
                    self._stage_free.put(slot)

        self._bg.put(send)

    def _download_staged(self, items: list[tuple[int, int]]) -> list[tuple[int, int]]:
        """Raw remote pages -> pinned slot -> HBM page (async H2D); returns the items that were
        not served this way (miss or a blob of another format)."""
        left: list[tuple[int, int]] = []
        hdr_len = len(self.serde._header(0))
        want = self.serde._header(self.serde.page_elems * self._stage.element_size())
        stream = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else _NoStream()
        used: list[int] = []
        slot = None
        try:
            for i, (h, p) in enumerate(items):
                try:
                    slot = self._stage_free.get_nowait()
                except queue.Empty:
                    stream.synchronize()  # the copies out of the slots taken so far have landed
                    for sl in used:
                        self._stage_free.put(sl)
                    used = []
                    try:  # uploads hold the rest of the ring: wait a little, then take the per-page path
                        slot = self._stage_free.get(timeout=self.STAGE_WAIT_S)
                    except queue.Empty:
                        left.extend(items[i:])
                        break
                view = memoryview(self._stage[slot].view(torch.uint8).numpy())
                hdr = self.remote.get_page_into(self._key(h), hdr_len, view)
                if hdr != want:
                    self._stage_free.put(slot)
                    slot = None
                    left.append((h, p))
                    continue
                ops.page_assign(self.pool, p, self._stage[slot], non_blocking=True)
                used.append(slot)
                slot = None
        finally:
            # also on a connection lost half way: every slot goes back to the ring, once the copies
            # that read them have landed
            if used:
                stream.synchronize()
                for sl in used:
                    self._stage_free.put(sl)
            if slot is not None:
                self._stage_free.put(slot)
        return left

    def poll(self) -> None:
        """Release page references whose offload copies or P->D pushes have completed."""
        if self._pusher is not None:
            self._pusher.drain_released()
            if len(self._push_born) > 64:
                # a handler that was cancelled (client gone) never claims its future: forget finished
                # ones after a while instead of keeping them for the life of the process
                now = time.time()
                for rid in [r for r, t in self._push_born.items() if now - t > 120 and self._push_futures[r].done()]:
                    self._push_born.pop(rid, None)
                    self._push_futures.pop(rid, None)
        still = []
        for ev, pages in self._pending:
            if ev.query():
                self.engine.bm.free(pages)
            else:
                still.append((ev, pages))
        self._pending = still

    def on_evict(self, h: int, page: int) -> None:
        """HBM eviction of a cached page: lower tiers may still hold it (write-through)."""
        held = (self.host is not None and self.host.contains(h)) or \
               (self.disk is not None and self.disk.contains(h)) or h in self.remote_known
        if not held and self.controller_url:
            self._report({"instance_id": self.instance_id, "evict": [h]})

    # ------------------------------------------------------------------ disaggregated prefill
    @property
    def pd_stats(self) -> dict:
        if not hasattr(self, "_pd_stats"):
            self._pd_stats = {"pushes": 0, "pages": 0, "seconds": 0.0, "by_mode": {}}
        return self._pd_stats

    def _push_target(self, req):
        """Where this request's pages go (a prefill request of a disaggregated pair), or None."""
        kvp = req.kv_transfer_params or {}
        if not kvp.get("do_remote_decode") or self.role == "kv_consumer":
            return None
        target = kvp.get("decode_target")
        if not target:
            return None
        from ..parallel.kv_transfer import target_allowed

        if not target_allowed(target):
            if not getattr(req, "_push_refused", False):
                req._push_refused = True
                log.error("KV push target %r refused (not http(s) or not in VPSA_PD_ALLOWED_TARGETS)", target)
            return None
        return target

    def _submit_push(self, req, target: str, final: bool) -> None:
        from ..parallel.kv_transfer import PagePusher

        n_full = min(req.num_computed // PAGE, len(req.hashes), len(req.pages))
        done = getattr(req, "_pushed_pages", 0)
        if n_full <= done and not (final and done > 0):
            return  # nothing new; a request that never submitted anything needs no closing job
        if self._pusher is None:
            self._pusher = PagePusher(self.engine)
        first = req.request_id not in self._push_futures and done == 0
        fut = self._pusher.submit(req.request_id, target, req.hashes[done:n_full], req.pages[done:n_full], final)
        req._pushed_pages = n_full
        if first:
            import time

            self._push_futures[req.request_id] = fut
            self._push_born[req.request_id] = time.time()
            rid = req.request_id

            def finished(f, rid=rid, target=target):
                st = self.pd_stats
                if f.exception() is not None:
                    st["failed"] = st.get("failed", 0) + 1
                    log.error("KV push of %s to %s failed: %s", rid, target, f.exception())
                    return
                r = f.result()
                st["pushes"] += 1
                st["pages"] += r["pages"]
                st["jobs"] = st.get("jobs", 0) + r["jobs"]
                st["seconds"] += r["seconds"]
                st["tail_seconds"] = st.get("tail_seconds", 0.0) + r["tail_seconds"]
                st["by_mode"][r["mode"]] = st["by_mode"].get(r["mode"], 0) + 1
                log.info("pushed %d KV pages of %s to %s via %s (%d jobs over %.1f ms; %.1f ms after the request's end)",
                         r["pages"], rid, target, r["mode"], r["jobs"], r["seconds"] * 1e3, r["tail_seconds"] * 1e3)

            fut.add_done_callback(finished)

    def on_chunk_computed(self, req) -> None:
        """Engine thread, after a prefill chunk of ``req`` ran: the pages it filled (and, the first
        time, the pages it found cached) leave for the decode replica now, while the following
        chunks are computed.  A reference per page, one stream event, one queue put."""
        target = self._push_target(req)
        if target is not None:
            self._submit_push(req, target, final=False)

    def on_request_finished(self, req) -> None:
        """Engine thread, while the request still owns its pages (also for aborted requests).  A
        prefill request of a disaggregated pair hands the pages of its last chunk to the push
        worker here and closes its transfer.  The transfers themselves (control-plane round
        trips, the copies, back-pressure of a busy decode replica) run on that worker; the HTTP
        handler of the request awaits ``push_future`` before it answers, so the router sends the
        decode request only once the pages are published there."""
        target = self._push_target(req)
        if target is not None:
            self._submit_push(req, target, final=True)

    def push_future(self, request_id: str):
        """The transfer started for ``request_id`` when it finished (None: there was none).  Taken
        once: the caller owns it."""
        self._push_born.pop(request_id, None)
        return self._push_futures.pop(request_id, None)

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        self._bg.put(None)
        if self._pusher is not None:
            self._pusher.close()
        if self._pool_exec is not None:
            self._pool_exec.shutdown(wait=True)
        for ev, pages in self._pending:
            ev.wait()
            self.engine.bm.free(pages)
        self._pending = []
        if self.host is not None:
            self.host.close()
        if self.remote is not None:
            self.remote.close()


class _NoStream:
    """Stands in for the compute stream where copies are synchronous (host-only runs)."""

    def synchronize(self) -> None:
        pass


class _TorchEvent:
    """Gives a torch event the query/wait pair the pending-copy list expects."""

    def __init__(self, ev):
        self.ev = ev

    def query(self) -> bool:
        return self.ev.query()

    def wait(self) -> None:
        self.ev.synchronize()


class _BoundedSet:
    """Insertion-ordered set that forgets its older half when it reaches ``limit`` entries.  The
    belief about what a remote cache server holds may be stale either way (the server evicts on its
    own): forgetting a page costs one redundant upload, remembering every page ever uploaded costs
    memory without end on a replica that runs for weeks."""

    def __init__(self, limit: int):
        self.limit = limit
        self._d: dict[int, None] = {}

    def add(self, h: int) -> None:
        self._d[h] = None
        if len(self._d) >= self.limit:
            for k in list(self._d)[: self.limit // 2]:
                del self._d[k]

    def discard(self, h: int) -> None:
        self._d.pop(h, None)

    def __contains__(self, h) -> bool:
        return h in self._d

    def __len__(self) -> int:
        return len(self._d)


class _ComputeFence:
    """Adapter giving a copy-engine event the ``wait`` the model runner expects: the wait is a
    GPU-side stream dependency, not a host block."""

    def __init__(self, host: HostKVPool, ev):
        self.host, self.ev = host, ev

    def wait(self) -> None:
        self.host.make_compute_wait(self.ev)
