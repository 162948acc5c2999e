"""Shared remote KV cache server (``lm://host:port``).

Role of the ``lmcache_experimental_server 0.0.0.0 <port>`` process the reference's chart and
operator launch (/root/reference/helm/templates/deployment-cache-server.yaml:34-43,
/root/reference/operator/internal/controller/cacheserver_controller.go:185-188): engines
PUT serialised KV pages under their chain hash and GET them from any replica.

Protocol (binary, little endian, one TCP connection carries any number of requests):
    request   op u8 | key_len u16 | key | value_len u64 | value
    response  status u8 | value_len u64 | value
    ops       1 PUT  2 GET  3 EXISTS  4 EXISTS_MANY (value = keys joined by \\n; response =
              one byte 0/1 per key)  5 DELETE  6 STATS (JSON)
The store is an in-memory LRU bounded by ``--max-size-gb``.
"""

from __future__ import annotations

import argparse
import asyncio
import json
import struct
import sys
from collections import OrderedDict

OP_PUT, OP_GET, OP_EXISTS, OP_EXISTS_MANY, OP_DELETE, OP_STATS = 1, 2, 3, 4, 5, 6
ST_OK, ST_MISS, ST_ERR = 0, 1, 2


class LRUStore:
    def __init__(self, max_bytes: int):
        self.max_bytes = max_bytes
        self.data: OrderedDict[bytes, bytes] = OrderedDict()
        self.bytes = 0
        self.hits = self.misses = self.puts = self.evictions = 0
        self._spare: dict[int, list] = {}

    def put(self, k: bytes, v: bytes) -> None:
        old = self.data.pop(k, None)
        if old is not None:
            self.bytes -= len(old)
        self.data[k] = v
        self.bytes += len(v)
        self.puts += 1
        while self.bytes > self.max_bytes and len(self.data) > 1:
            _, ev = self.data.popitem(last=False)
            self.bytes -= len(ev)
            self.evictions += 1
            # reuse only a buffer nobody else holds (a queued response would keep a reference)
            if isinstance(ev, bytearray) and len(ev) >= (1 << 20) and sys.getrefcount(ev) == 2:
                spare = self._spare.setdefault(len(ev), [])
                if len(spare) < 16:
                    spare.append(ev)

    def buffer(self, n: int):
        """A writable buffer of ``n`` bytes for an incoming value.  Buffers of evicted values of
        the same size are handed out again: pages all have one size, and a recycled buffer is
        already mapped (a fresh 8 MiB allocation costs a page fault per 4 KiB)."""
        pool = self._spare.get(n)
        if pool:
            return pool.pop()
        return bytearray(n)

    def get(self, k: bytes) -> bytes | None:
        v = self.data.get(k)
        if v is None:
            self.misses += 1
            return None
        self.data.move_to_end(k)
        self.hits += 1
        return v

    def stats(self) -> dict:
        return {"keys": len(self.data), "bytes": self.bytes, "max_bytes": self.max_bytes,
                "hits": self.hits, "misses": self.misses, "puts": self.puts,
                "evictions": self.evictions}


class _Conn(asyncio.BufferedProtocol):
    """One client connection.  Requests are parsed with a four-state machine (op+key length,
    key, value length, value) and the value bytes are received straight into the buffer that
    the store will keep: no intermediate stream buffer, no joins.  Responses are written as
    header and value separately for the same reason."""

    def __init__(self, server: "CacheServer"):
        self.server = server
        self.transport: asyncio.Transport | None = None
        self._small = bytearray(65536 + 16)
        self._expect(0, 3)

    def _expect(self, state: int, n: int, buf: bytearray | None = None) -> None:
        self.state, self.need, self.got = state, n, 0
        self.buf = buf if buf is not None else self._small

    def connection_made(self, transport) -> None:
        self.transport = transport
        self.server._conns.add(self)

    def connection_lost(self, exc) -> None:
        self.server._conns.discard(self)

    def get_buffer(self, sizehint: int) -> memoryview:
        return memoryview(self.buf)[self.got:self.need]

    def buffer_updated(self, nbytes: int) -> None:
        self.got += nbytes
        while self.got == self.need:
            if self.state == 0:
                self.op, klen = struct.unpack_from("<BH", self.buf, 0)
                if klen:
                    self._expect(1, klen)
                    return
                self.key = b""
                self._expect(2, 8)
                return
            if self.state == 1:
                self.key = bytes(self.buf[: self.need])
                self._expect(2, 8)
                return
            if self.state == 2:
                (vlen,) = struct.unpack_from("<Q", self.buf, 0)
                if vlen > self.server.store.max_bytes + (1 << 20):
                    self.transport.close()
                    return
                if vlen:
                    self._expect(3, vlen, self.server.store.buffer(vlen))
                    return
                self._dispatch(b"")
            else:
                self._dispatch(self.buf)
            self._expect(0, 3)
            return

    def _dispatch(self, val) -> None:
        store, op, key = self.server.store, self.op, self.key
        status, out = ST_OK, b""
        if op == OP_PUT:
            store.put(key, val)
        elif op == OP_GET:
            v = store.get(key)
            status, out = (ST_OK, v) if v is not None else (ST_MISS, b"")
        elif op == OP_EXISTS:
            status = ST_OK if key in store.data else ST_MISS
        elif op == OP_EXISTS_MANY:
            keys = bytes(val).split(b"\n") if val else []
            out = bytes(1 if k in store.data else 0 for k in keys)
        elif op == OP_DELETE:
            v = store.data.pop(key, None)
            if v is not None:
                store.bytes -= len(v)
        elif op == OP_STATS:
            out = json.dumps(store.stats()).encode()
        else:
            status = ST_ERR
        self.transport.write(struct.pack("<BQ", status, len(out)))
        if out:
            self.transport.write(out)


class CacheServer:
    def __init__(self, host: str = "0.0.0.0", port: int = 8000, max_bytes: int = 64 << 30):
        self.host, self.port = host, port
        self.store = LRUStore(max_bytes)
        self._server: asyncio.AbstractServer | None = None
        self._conns: set[_Conn] = set()

    async def start(self) -> "CacheServer":
        loop = asyncio.get_running_loop()
        self._server = await loop.create_server(lambda: _Conn(self), self.host, self.port,
                                                reuse_address=True, backlog=256)
        self.port = self._server.sockets[0].getsockname()[1]
        return self

    async def stop(self) -> None:
        if self._server:
            self._server.close()
            await self._server.wait_closed()
        for c in list(self._conns):  # wait_closed() does not cover connections already open
            if c.transport is not None:
                c.transport.close()
        self._conns.clear()


def main(argv: list[str] | None = None) -> None:
    ap = argparse.ArgumentParser(prog="vpsa-cache-server")
    ap.add_argument("host", nargs="?", default="0.0.0.0")
    ap.add_argument("port", nargs="?", type=int, default=8000)
    ap.add_argument("--max-size-gb", type=float, default=64.0)
    ap.add_argument("--max-bytes", type=int, default=None, help="capacity in bytes (overrides --max-size-gb)")
    a = ap.parse_args(argv)
    if a.max_bytes:
        a.max_size_gb = a.max_bytes / (1 << 30)

    async def run():
        srv = await CacheServer(a.host, a.port, int(a.max_size_gb * (1 << 30))).start()
        print(f"KV cache server listening on {a.host}:{srv.port}", flush=True)
        await asyncio.Event().wait()

    asyncio.run(run())
