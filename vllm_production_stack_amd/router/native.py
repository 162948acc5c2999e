"""ctypes bindings to the native routing core (``libvpsa_routercore.so``): prefix trie,
consistent-hash ring and round-robin counter, with pure-Python twins that implement the
identical algorithms (same xxh64 chunk hashes, same ring placement) for hosts where the
library has not been built."""

from __future__ import annotations

import bisect
import ctypes
from ctypes import c_char_p, c_int, c_size_t, c_uint32, c_uint64, c_void_p

import xxhash

from .. import _native

_lib = None


def _load():
    global _lib
    if _lib is None and _native.have("vpsa_routercore"):
        lib = _native.load("vpsa_routercore")
        lib.vpsa_trie_new.restype = c_void_p
        lib.vpsa_trie_new.argtypes = [c_size_t, c_size_t]
        lib.vpsa_trie_free.argtypes = [c_void_p]
        lib.vpsa_trie_insert.argtypes = [c_void_p, c_char_p, c_size_t, c_uint32]
        lib.vpsa_trie_nodes.restype = c_size_t
        lib.vpsa_trie_nodes.argtypes = [c_void_p]
        lib.vpsa_trie_remove_endpoint.argtypes = [c_void_p, c_uint32]
        lib.vpsa_trie_match.restype = c_size_t
        lib.vpsa_trie_match.argtypes = [c_void_p, c_char_p, c_size_t, ctypes.POINTER(c_uint32),
                                        c_size_t, ctypes.POINTER(c_uint32), ctypes.POINTER(c_size_t)]
        lib.vpsa_ring_new.restype = c_void_p
        lib.vpsa_ring_new.argtypes = [c_int]
        lib.vpsa_ring_free.argtypes = [c_void_p]
        lib.vpsa_ring_add.argtypes = [c_void_p, c_char_p]
        lib.vpsa_ring_remove.argtypes = [c_void_p, c_char_p]
        lib.vpsa_ring_get.restype = c_int
        lib.vpsa_ring_get.argtypes = [c_void_p, c_char_p, c_size_t, c_char_p, c_size_t]
        lib.vpsa_rr_new.restype = c_void_p
        lib.vpsa_rr_free.argtypes = [c_void_p]
        lib.vpsa_rr_next.restype = c_uint64
        lib.vpsa_rr_next.argtypes = [c_void_p, c_uint64]
        _lib = lib
    return _lib


def native_available() -> bool:
    return _load() is not None


# ------------------------------------------------------------------ prefix trie
class _PyNode:
    __slots__ = ("children", "endpoints")

    def __init__(self):
        self.children: dict[int, _PyNode] = {}
        self.endpoints: set[int] = set()


class HashTrie:
    """Prompt-prefix index: the prompt is cut into ``chunk_size``-character chunks, each
    hashed with xxh64; a node holds the endpoints that have served the prefix ending there."""

    def __init__(self, chunk_size: int = 128, max_nodes: int = 0, force_python: bool = False):
        self.chunk_size = chunk_size
        self._ids: dict[str, int] = {}
        self._names: list[str] = []
        lib = None if force_python else _load()
        self._lib = lib
        if lib is not None:
            self._h = lib.vpsa_trie_new(chunk_size, max_nodes)
        else:
            self._root = _PyNode()
            self._count = 1

    def __del__(self):
        if getattr(self, "_lib", None) is not None and getattr(self, "_h", None):
            self._lib.vpsa_trie_free(self._h)
            self._h = None

    def _id(self, endpoint: str) -> int:
        i = self._ids.get(endpoint)
        if i is None:
            i = self._ids[endpoint] = len(self._names)
            self._names.append(endpoint)
        return i

    def _hashes(self, text: str):
        for i in range(0, len(text), self.chunk_size):
            yield xxhash.xxh64_intdigest(text[i:i + self.chunk_size].encode())

    def insert(self, text: str, endpoint: str) -> None:
        eid = self._id(endpoint)
        if self._lib is not None:
            b = text.encode()
            self._lib.vpsa_trie_insert(self._h, b, len(b), eid)
            return
        node = self._root
        node.endpoints.add(eid)
        for h in self._hashes(text):
            nxt = node.children.get(h)
            if nxt is None:
                nxt = node.children[h] = _PyNode()
                self._count += 1
            node = nxt
            node.endpoints.add(eid)

    def longest_prefix_match(self, text: str, available: set[str]) -> tuple[int, set[str]]:
        if not available:
            return 0, set()
        if self._lib is not None:
            ids = [self._id(e) for e in available]
            n = len(ids)
            arr = (c_uint32 * n)(*ids)
            out = (c_uint32 * n)()
            n_out = c_size_t(0)
            b = text.encode()
            m = self._lib.vpsa_trie_match(self._h, b, len(b), arr, n, out, ctypes.byref(n_out))
            return int(m), {self._names[out[i]] for i in range(n_out.value)}
        node = self._root
        sel = {self._id(e) for e in available}
        matched = 0
        for h in self._hashes(text):
            node = node.children.get(h)
            if node is None:
                break
            inter = node.endpoints & sel
            if not inter:
                break
            matched += self.chunk_size
            sel = inter
        return matched, {self._names[i] for i in sel}

    def remove_endpoint(self, endpoint: str) -> None:
        eid = self._ids.get(endpoint)
        if eid is None:
            return
        if self._lib is not None:
            self._lib.vpsa_trie_remove_endpoint(self._h, eid)
            return

        def rec(n: _PyNode):
            n.endpoints.discard(eid)
            for k in list(n.children):
                rec(n.children[k])
                if not n.children[k].endpoints:
                    del n.children[k]

        rec(self._root)

    def node_count(self) -> int:
        if self._lib is not None:
            return int(self._lib.vpsa_trie_nodes(self._h))
        return self._count


# ------------------------------------------------------------------ consistent hash ring
class HashRing:
    """Virtual-node ring; a key maps to the first virtual node clockwise of xxh64(key).
    Adding or removing a node only remaps the keys that fall on its arcs."""

    def __init__(self, vnodes: int = 160, force_python: bool = False):
        self.vnodes = vnodes
        self._members: set[str] = set()
        lib = None if force_python else _load()
        self._lib = lib
        if lib is not None:
            self._h = lib.vpsa_ring_new(vnodes)
        else:
            self._keys: list[int] = []
            self._owner: dict[int, str] = {}

    def __del__(self):
        if getattr(self, "_lib", None) is not None and getattr(self, "_h", None):
            self._lib.vpsa_ring_free(self._h)
            self._h = None

    def get_nodes(self) -> set[str]:
        return set(self._members)

    def add_node(self, name: str) -> None:
        if name in self._members:
            return
        self._members.add(name)
        if self._lib is not None:
            self._lib.vpsa_ring_add(self._h, name.encode())
            return
        for i in range(self.vnodes):
            h = xxhash.xxh64_intdigest(f"{name}#{i}".encode())
            cur = self._owner.get(h)
            if cur is None:
                bisect.insort(self._keys, h)
                self._owner[h] = name
            elif name < cur:
                self._owner[h] = name

    def remove_node(self, name: str) -> None:
        if name not in self._members:
            return
        self._members.discard(name)
        if self._lib is not None:
            self._lib.vpsa_ring_remove(self._h, name.encode())
            return
        dead = [h for h, o in self._owner.items() if o == name]
        for h in dead:
            del self._owner[h]
        self._keys = sorted(self._owner)

    def get_node(self, key: str) -> str | None:
        if not self._members:
            return None
        if self._lib is not None:
            buf = ctypes.create_string_buffer(1024)
            k = key.encode()
            n = self._lib.vpsa_ring_get(self._h, k, len(k), buf, 1024)
            return buf.value.decode() if n > 0 else None
        h = xxhash.xxh64_intdigest(key.encode())
        i = bisect.bisect_left(self._keys, h)
        if i == len(self._keys):
            i = 0
        return self._owner[self._keys[i]]
