"""Paged KV-cache bookkeeping: 64-token pages, reference counts, hash-chained prefix cache.

A full page is identified by the hash chain of all tokens up to and including it, so two
requests that share a prefix share the pages (and a later turn of the same conversation
finds its history resident).  Pages with no reference stay in the cache on an LRU list
and are reclaimed only under memory pressure; reclaim is reported through ``on_evict`` so
a lower tier (pinned host DRAM) can keep the content.
"""

from __future__ import annotations

import struct
import sys
from array import array
from collections import OrderedDict, deque
from typing import Callable, Iterable

import xxhash

PAGE = 64


def chain_hashes(token_ids: list[int], salt: bytes = b"", prev: int = 0, start_page: int = 0,
                 page: int = PAGE) -> list[int]:
    """Hash chain over the *full* pages of ``token_ids`` beginning at ``start_page``."""
    out = []
    h = prev
    n_full = len(token_ids) // page
    if n_full <= start_page:
        return out
    # one packing pass for all pages (same bytes as struct.pack("<{page}i") per page)
    a = array("i", token_ids[start_page * page:n_full * page])
    if sys.byteorder != "little":  # pragma: no cover - hashes are defined on little-endian words
        a.byteswap()
    mv = memoryview(a.tobytes())
    step = 4 * page
    pack_q = struct.Struct("<Q").pack
    for i in range(n_full - start_page):
        x = xxhash.xxh64(pack_q(h))
        if salt:
            x.update(salt)
        x.update(mv[i * step:(i + 1) * step])
        h = x.intdigest()
        out.append(h)
    return out


class BlockManager:
    def __init__(self, num_pages: int, enable_prefix_caching: bool = True,
                 on_evict: Callable[[int, int], None] | None = None, reuse_in_hbm: bool = True):
        """``enable_prefix_caching``: full pages are identified by their chain hash (needed by
        the offload tiers and by transfers even when the engine's own prefix cache is off).
        ``reuse_in_hbm`` False is ``--no-enable-prefix-caching`` with a KV connector configured
        (the reference's benchmark setup, /root/reference/tutorials/07-benchmark-multi-round-qa-single-gpu.md:31-67):
        pages are hashed and offloaded, but a finished request's pages go straight back to the
        free list and a new request never finds its prefix in HBM - it comes from the tiers."""
        self.num_pages = num_pages
        self.enable_prefix_caching = enable_prefix_caching
        self.reuse_in_hbm = reuse_in_hbm
        self.on_evict = on_evict
        self.ref = [0] * num_pages
        self.page_hash: list[int | None] = [None] * num_pages
        self.free_ids: deque[int] = deque(range(num_pages))
        self.cached: dict[int, int] = {}  # hash -> page
        self.evictable: OrderedDict[int, None] = OrderedDict()  # page ids, LRU first
        # prefix-cache counters, in tokens (exported as vllm:gpu_prefix_cache_*)
        self.queries = 0
        self.hits = 0

    # ------------------------------------------------------------------ capacity
    def num_free(self) -> int:
        return len(self.free_ids) + len(self.evictable)

    def usage(self) -> float:
        if self.num_pages == 0:
            return 0.0
        return 1.0 - self.num_free() / self.num_pages

    # ------------------------------------------------------------------ prefix cache
    def match_prefix(self, hashes: Iterable[int], count_stats: bool = True) -> list[int]:
        """Longest cached run from the start of ``hashes``; takes a reference on each page."""
        pages: list[int] = []
        hashes = list(hashes)
        if self.enable_prefix_caching and self.reuse_in_hbm:
            for h in hashes:
                p = self.cached.get(h)
                if p is None:
                    break
                pages.append(p)
            for p in pages:
                self._take(p)
        if count_stats:
            self.queries += len(hashes) * PAGE
            self.hits += len(pages) * PAGE
        return pages

    def peek_prefix(self, hashes: Iterable[int]) -> int:
        n = 0
        if not self.reuse_in_hbm:
            return 0
        for h in hashes:
            if h not in self.cached:
                break
            n += 1
        return n

    def contains(self, h: int) -> bool:
        return h in self.cached

    def _take(self, p: int) -> None:
        if self.ref[p] == 0:
            self.evictable.pop(p, None)
        self.ref[p] += 1

    def pin(self, pages: Iterable[int]) -> None:
        """One more reference on each page for an asynchronous reader (an offload copy, a push to
        a decode replica); ``free`` gives it back.  Only pages somebody owns, or cached pages
        waiting for eviction, can be pinned: a page on the free list may be handed out at any
        moment, and a reference on it would be lost (``allocate`` starts a page at one)."""
        for p in pages:
            if self.ref[p] == 0 and p not in self.evictable:
                raise AssertionError(f"page {p} is free: nothing to pin")
            self._take(p)

    def register(self, page: int, h: int) -> None:
        """Mark ``page`` (now full) as holding the content identified by ``h``."""
        if not self.enable_prefix_caching or self.page_hash[page] is not None:
            return
        if h in self.cached:
            return  # an identical page is already indexed; this one stays private
        self.page_hash[page] = h
        self.cached[h] = page

    # ------------------------------------------------------------------ alloc / free
    def can_allocate(self, n: int) -> bool:
        return self.num_free() >= n

    def allocate(self, n: int) -> list[int]:
        if n > self.num_free():
            raise MemoryError(f"KV pool exhausted: need {n}, free {self.num_free()}")
        out = []
        for _ in range(n):
            if self.free_ids:
                p = self.free_ids.popleft()
            else:
                p, _ = self.evictable.popitem(last=False)
                h = self.page_hash[p]
                if h is not None:
                    if self.on_evict is not None:
                        self.on_evict(h, p)
                    del self.cached[h]
                    self.page_hash[p] = None
            self.ref[p] = 1
            out.append(p)
        return out

    def free(self, pages: Iterable[int]) -> None:
        # release in reverse so that the tail of a sequence is evicted before its head
        for p in reversed(list(pages)):
            self.ref[p] -= 1
            assert self.ref[p] >= 0
            if self.ref[p] == 0:
                h = self.page_hash[p]
                if h is not None and self.reuse_in_hbm:
                    self.evictable[p] = None
                else:
                    if h is not None:  # hashed for the tiers only: nothing stays cached in HBM
                        if self.on_evict is not None:
                            self.on_evict(h, p)
                        del self.cached[h]
                        self.page_hash[p] = None
                    self.free_ids.append(p)

    def reset_prefix_cache(self) -> int:
        """Drop every unreferenced cached page (used by /sleep and tests)."""
        n = 0
        for p in list(self.evictable):
            h = self.page_hash[p]
            if h is not None:
                del self.cached[h]
                self.page_hash[p] = None
            del self.evictable[p]
            self.free_ids.append(p)
            n += 1
        return n
