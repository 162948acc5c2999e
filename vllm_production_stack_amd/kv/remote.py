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
    def _recv_into(s: socket.socket, view: memoryview) -> None:
        got, n = 0, len(view)
        while got < n:
            k = s.recv_into(view[got:], n - got)
            if k == 0:
                raise ConnectionError("cache server closed the connection")
            got += k

    def put_parts(self, key: bytes, *parts) -> bool:
        """PUT whose value is the concatenation of ``parts`` (bytes or memoryviews), sent
        without joining them: a page goes from its pinned staging slot straight to the socket."""
        total = sum(len(p) for p in parts)
        for attempt in (0, 1):
            try:
                s = self._sock()
                s.sendall(struct.pack("<BH", OP_PUT, len(key)) + key + struct.pack("<Q", total))
                for part in parts:
                    s.sendall(part)
                status, vlen = struct.unpack("<BQ", self._recv(s, 9))
                if vlen:
                    self._recv(s, vlen)
                return status == ST_OK
            except (OSError, ConnectionError):
                self._drop()
                if attempt:
                    raise
        return False

    def get_page_into(self, key: bytes, header_len: int, payload: memoryview) -> bytes | None:
        """GET of a value laid out as ``header | payload``: the payload is received directly
        into ``payload`` (e.g. pinned memory).  Returns the header, or None on a miss or when
        the stored value does not have that size (the caller then takes the generic path)."""
        for attempt in (0, 1):
            try:
                s = self._sock()
                s.sendall(struct.pack("<BH", OP_GET, len(key)) + key + struct.pack("<Q", 0))
                status, vlen = struct.unpack("<BQ", self._recv(s, 9))
                if status != ST_OK:
                    if vlen:
                        self._recv(s, vlen)
                    return None
                if vlen != header_len + len(payload):
                    self._recv(s, vlen)
                    return None
                hdr = self._recv(s, header_len)
                self._recv_into(s, payload)
                return hdr
            except (OSError, ConnectionError):
                self._drop()
                if attempt:
                    raise
        return None

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
