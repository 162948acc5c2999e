"""Blocking client of the shared remote KV cache server (``lm://host:port``); used from the
connector's worker threads, one connection per thread."""

from __future__ import annotations

import json
import socket
import struct
import threading

from ..cache_server import (OP_EXISTS, OP_EXISTS_MANY, OP_GET, OP_PUT, OP_STATS, ST_OK)


def parse_lm_url(url: str) -> tuple[str, int]:
    rest = url.split("://", 1)[-1].rstrip("/")
    host, _, port = rest.rpartition(":")
    return host or "localhost", int(port)


class RemoteKVClient:
    def __init__(self, url: str, timeout: float = 10.0):
        self.host, self.port = parse_lm_url(url)
        self.timeout = timeout
        self._local = threading.local()

    def _sock(self) -> socket.socket:
        s = getattr(self._local, "sock", None)
        if s is None:
            s = socket.create_connection((self.host, self.port), timeout=self.timeout)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._local.sock = s
        return s

    def _drop(self) -> None:
        s = getattr(self._local, "sock", None)
        if s is not None:
            try:
                s.close()
            except OSError:
                pass
            self._local.sock = None

    @staticmethod
    def _recv(s: socket.socket, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            chunk = s.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("cache server closed the connection")
            buf += chunk
        return bytes(buf)

    def _call(self, op: int, key: bytes, value: bytes = b"") -> tuple[int, bytes]:
        for attempt in (0, 1):
            try:
                s = self._sock()
                s.sendall(struct.pack("<BH", op, len(key)) + key + struct.pack("<Q", len(value)))
                if value:
                    s.sendall(value)
                status, vlen = struct.unpack("<BQ", self._recv(s, 9))
                return status, self._recv(s, vlen) if vlen else b""
            except (OSError, ConnectionError):
                self._drop()
                if attempt:
                    raise
        raise ConnectionError("unreachable")

    @staticmethod
    def key(h: int, prefix: str = "") -> bytes:
        return f"{prefix}{h:016x}".encode()

    def put(self, key: bytes, value: bytes) -> bool:
        return self._call(OP_PUT, key, value)[0] == ST_OK

    def get(self, key: bytes) -> bytes | None:
        st, v = self._call(OP_GET, key)
        return v if st == ST_OK else None

    def exists(self, key: bytes) -> bool:
        return self._call(OP_EXISTS, key)[0] == ST_OK

    def exists_many(self, keys: list[bytes]) -> list[bool]:
        if not keys:
            return []
        _, v = self._call(OP_EXISTS_MANY, b"", b"\n".join(keys))
        return [b == 1 for b in v]

    def stats(self) -> dict:
        return json.loads(self._call(OP_STATS, b"")[1] or b"{}")

    def close(self) -> None:
        self._drop()
