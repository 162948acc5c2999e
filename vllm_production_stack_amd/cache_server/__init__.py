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
from collections import OrderedDict

OP_PUT, OP_GET, OP_EXISTS, OP_EXISTS_MANY, OP_DELETE, OP_STATS = 1, 2, 3, 4, 5, 6
ST_OK, ST_MISS, ST_ERR = 0, 1, 2


class LRUStore:
    def __init__(self, max_bytes: int):
        self.max_bytes = max_bytes
        self.data: OrderedDict[bytes, bytes] = OrderedDict()
        self.bytes = 0
        self.hits = self.misses = self.puts = self.evictions = 0

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


class CacheServer:
    def __init__(self, host: str = "0.0.0.0", port: int = 8000, max_bytes: int = 64 << 30):
        self.host, self.port = host, port
        self.store = LRUStore(max_bytes)
        self._server: asyncio.AbstractServer | None = None
        self._conns: set[asyncio.Task] = set()

    async def _handle(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter) -> None:
        task = asyncio.current_task()
        self._conns.add(task)
        try:
            while True:
                try:
                    hdr = await reader.readexactly(3)
                except asyncio.IncompleteReadError:
                    break
                op, klen = struct.unpack("<BH", hdr)
                key = await reader.readexactly(klen)
                (vlen,) = struct.unpack("<Q", await reader.readexactly(8))
                val = await reader.readexactly(vlen) if vlen else b""
                status, out = ST_OK, b""
                if op == OP_PUT:
                    self.store.put(key, val)
                elif op == OP_GET:
                    v = self.store.get(key)
                    status, out = (ST_OK, v) if v is not None else (ST_MISS, b"")
                elif op == OP_EXISTS:
                    status = ST_OK if key in self.store.data else ST_MISS
                elif op == OP_EXISTS_MANY:
                    keys = val.split(b"\n") if val else []
                    out = bytes(1 if k in self.store.data else 0 for k in keys)
                elif op == OP_DELETE:
                    v = self.store.data.pop(key, None)
                    if v is not None:
                        self.store.bytes -= len(v)
                elif op == OP_STATS:
                    out = json.dumps(self.store.stats()).encode()
                else:
                    status = ST_ERR
                writer.write(struct.pack("<BQ", status, len(out)) + out)
                await writer.drain()
        except (ConnectionResetError, BrokenPipeError, asyncio.CancelledError):
            pass
        finally:
            self._conns.discard(task)
            writer.close()

    async def start(self) -> "CacheServer":
        self._server = await asyncio.start_server(self._handle, self.host, self.port,
                                                  limit=1 << 26, reuse_address=True)
        self.port = self._server.sockets[0].getsockname()[1]
        return self

    async def stop(self) -> None:
        if self._server:
            self._server.close()
            await self._server.wait_closed()
        # wait_closed() does not cover connections that are already open: end their handlers
        conns = list(self._conns)
        for t in conns:
            t.cancel()
        if conns:
            await asyncio.gather(*conns, return_exceptions=True)


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
