"""Router data plane on more than one core: ``--router-workers N``.

The reference router is one uvicorn process (/root/reference/src/vllm_router/app.py:284-300);
every SSE chunk of every stream crosses its one event loop
(/root/reference/src/vllm_router/services/request_service/request.py:109-119).  At eight engine
replicas with tens of users each that relay loop, not a GPU, becomes the first thing to saturate.

Here N identical router processes listen on the same port (``SO_REUSEPORT``: the kernel spreads
new connections over them) and each relays the streams of the connections it accepted.  Every
worker is a complete router (own discovery, own engine-stats scraper, own request statistics);
what must not be per worker is what an observer or a conversation sees as ONE router:

  * ``/metrics``: a scrape lands on one worker chosen by the kernel, and has to report the whole
    router.  Every worker publishes a snapshot of its metric registry to the supervisor once a
    second and right before it answers a scrape; the scraped worker merges all snapshots (request
    rates, in-flight counts and counters are summed, averages are weighted by each worker's
    request rate, what every worker observes alike (engine gauges, pod health) is reported once):
    ``router/metrics.py::merge_snapshots``.  The supervisor's unix socket carries both this and
    the prefix trie;

  * the prefix trie of ``prefixaware`` routing lives once, in the supervisor process, behind a
    unix socket (``SharedTrieServer`` / ``SharedTrieClient``): a follow-up turn finds the prefix
    its first turn inserted whichever worker accepted either connection;
  * the round-robin position is one counter in shared memory;
  * ``session`` routing is a pure function of the session key and the endpoint set (consistent
    hashing): nothing to share;
  * ``kvaware`` routing embeds the KV controller (one listening port, one instance registry) and
    stays single-process.

Load-based tie-breaks (least in-flight requests, lowest QPS) use each worker's own share of the
traffic, which the kernel's connection spreading makes a uniform sample of the whole.
"""

from __future__ import annotations

import asyncio
import logging
import multiprocessing
import os
import signal
import struct
import tempfile
import threading
import time

import msgpack

from .native import HashTrie

logger = logging.getLogger("vpsa.router.workers")

# set in a worker process before its app is built (see ``worker_main``)
_shared_state_path: str | None = None  # the supervisor's socket (metric snapshots; the trie when there is one)
_shared_trie_path: str | None = None
_shared_rr = None  # multiprocessing.Value('Q') or None
_worker_index = 0
_worker_count = 1


def worker_index() -> int:
    return _worker_index


def worker_count() -> int:
    return _worker_count


def shared_rr_counter():
    return _shared_rr


def shared_trie_path() -> str | None:
    return _shared_trie_path


def shared_state_path() -> str | None:
    return _shared_state_path


# ------------------------------------------------------------------ shared prefix trie
class SharedTrieServer:
    """The supervisor's shared state behind a unix socket: the prefix trie (match / insert /
    forget) and the workers' metric snapshots (stats_put / stats_all; a snapshot that has not been
    refreshed for ``SNAPSHOT_TTL`` seconds belongs to a worker that died and is dropped).  Frames
    are a 4-byte little-endian length followed by a msgpack array."""

    SNAPSHOT_TTL = 15.0

    def __init__(self, path: str, max_nodes: int):
        self.path = path
        self.trie = HashTrie(chunk_size=128, max_nodes=max_nodes)
        self.snapshots: dict[int, tuple[float, bytes]] = {}  # worker index -> (when, msgpack payload)
        self._loop: asyncio.AbstractEventLoop | None = None
        self._ready = threading.Event()
        self._thread = threading.Thread(target=self._run, name="shared-trie", daemon=True)

    def start(self) -> None:
        self._thread.start()
        if not self._ready.wait(10):
            raise RuntimeError("shared prefix-trie service did not start")

    def _run(self) -> None:
        loop = asyncio.new_event_loop()
        asyncio.set_event_loop(loop)
        self._loop = loop

        async def boot():
            await asyncio.start_unix_server(self._serve, path=self.path)
            self._ready.set()

        loop.run_until_complete(boot())
        loop.run_forever()

    async def _serve(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter) -> None:
        try:
            while True:
                head = await reader.readexactly(4)
                (n,) = struct.unpack("<I", head)
                msg = msgpack.unpackb(await reader.readexactly(n), raw=False)
                op = msg[0]
                if op == "match":
                    length, urls = self.trie.longest_prefix_match(msg[1], set(msg[2]))
                    out = [int(length), sorted(urls)]
                elif op == "insert":
                    self.trie.insert(msg[1], msg[2])
                    out = [0]
                elif op == "forget":
                    self.trie.remove_endpoint(msg[1])
                    out = [0]
                elif op == "stats_put":
                    self.snapshots[int(msg[1])] = (time.time(), msg[2])
                    out = [0]
                elif op == "stats_all":
                    lim = time.time() - self.SNAPSHOT_TTL
                    for w in [w for w, (t, _) in self.snapshots.items() if t < lim]:
                        del self.snapshots[w]
                    out = [0, [[w, blob] for w, (_, blob) in sorted(self.snapshots.items())]]
                else:
                    out = [-1]
                body = msgpack.packb(out)
                writer.write(struct.pack("<I", len(body)) + body)
                await writer.drain()
        except (asyncio.IncompleteReadError, ConnectionError):
            pass
        finally:
            writer.close()

    def stop(self) -> None:
        if self._loop is not None:
            self._loop.call_soon_threadsafe(self._loop.stop)
        try:
            os.unlink(self.path)
        except OSError:
            pass


class SharedTrieClient:
    """A worker's handle on the supervisor's trie: the ``HashTrie`` calls, awaited."""

    def __init__(self, path: str):
        self.path = path
        self._reader = None
        self._writer = None
        self._lock = asyncio.Lock()

    async def _call(self, msg: list):
        async with self._lock:
            for attempt in (0, 1):
                try:
                    if self._writer is None:
                        self._reader, self._writer = await asyncio.open_unix_connection(self.path)
                    body = msgpack.packb(msg)
                    self._writer.write(struct.pack("<I", len(body)) + body)
                    await self._writer.drain()
                    (n,) = struct.unpack("<I", await self._reader.readexactly(4))
                    return msgpack.unpackb(await self._reader.readexactly(n), raw=False)
                except (OSError, asyncio.IncompleteReadError):
                    if self._writer is not None:
                        self._writer.close()
                    self._reader = self._writer = None
                    if attempt:
                        raise

    async def longest_prefix_match(self, text: str, available: set[str]) -> tuple[int, set[str]]:
        if not available:
            return 0, set()
        length, urls = await self._call(["match", text, sorted(available)])
        return int(length), set(urls)

    async def insert(self, text: str, endpoint: str) -> None:
        await self._call(["insert", text, endpoint])

    async def remove_endpoint(self, endpoint: str) -> None:
        await self._call(["forget", endpoint])

    async def put_stats(self, worker: int, snapshot: bytes) -> None:
        await self._call(["stats_put", worker, snapshot])

    async def all_stats(self) -> list[tuple[int, bytes]]:
        out = await self._call(["stats_all"])
        return [(int(w), blob) for w, blob in out[1]]

    async def close(self) -> None:
        if self._writer is not None:
            self._writer.close()
            self._writer = None


# ------------------------------------------------------------------ processes
def effective_workers(args) -> int:
    n = max(1, int(getattr(args, "router_workers", 1) or 1))
    if n > 1 and args.routing_logic == "kvaware":
        logger.warning("kvaware routing embeds the KV controller and runs in one process: "
                       "--router-workers %d ignored", n)
        return 1
    return n


def worker_main(args, index: int, count: int, trie_path: str | None, rr, state_path: str | None = None) -> None:
    global _shared_trie_path, _shared_state_path, _shared_rr, _worker_index, _worker_count
    _shared_trie_path, _shared_rr, _worker_index, _worker_count = trie_path, rr, index, count
    _shared_state_path = state_path or trie_path
    signal.signal(signal.SIGINT, signal.SIG_IGN)  # the supervisor forwards the stop as SIGTERM
    from aiohttp import web

    from . import app as router_app

    application = router_app.create_app(args)
    web.run_app(application, host=args.host, port=args.port, access_log=None, backlog=8192,
                reuse_port=True,
                print=lambda *a: logger.info("router worker %d/%d listening on %s:%d", index + 1,
                                             count, args.host, args.port))


def run_workers(args, n: int) -> None:
    """Supervisor: start the shared state, fork the workers, restart a worker that dies, stop
    them all on SIGTERM / SIGINT."""
    ctx = multiprocessing.get_context("fork")
    rr = ctx.Value("Q", 0)
    # one socket for everything the workers share: metric snapshots always, the prefix trie when
    # the routing logic has one
    state_path = os.path.join(tempfile.mkdtemp(prefix="vpsa-router-"), "state.sock")
    trie_srv = SharedTrieServer(state_path, args.prefix_trie_max_nodes)
    trie_srv.start()
    trie_path = state_path if args.routing_logic == "prefixaware" else None
    procs: dict[int, multiprocessing.Process] = {}
    stopping = threading.Event()

    def spawn(i: int) -> None:
        p = ctx.Process(target=worker_main, args=(args, i, n, trie_path, rr, state_path),
                        name=f"router-worker-{i}")
        p.start()
        procs[i] = p

    def stop(*_):
        stopping.set()

    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)
    for i in range(n):
        spawn(i)
    logger.info("router supervisor: %d workers on %s:%d", n, args.host, args.port)
    try:
        while not stopping.wait(0.5):
            for i, p in list(procs.items()):
                if not p.is_alive():
                    logger.error("router worker %d exited with code %s: restarting", i, p.exitcode)
                    spawn(i)
    finally:
        for p in procs.values():
            if p.is_alive():
                p.terminate()
        for p in procs.values():
            p.join(10)
            if p.is_alive():
                p.kill()
        if trie_srv is not None:
            trie_srv.stop()
