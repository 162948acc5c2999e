"""Semantic cache for chat completions (alpha, behind the ``SemanticCache`` feature gate).

A chat request is embedded as the concatenation of its ``role: content`` lines; the cache
is an exact inner-product index over L2-normalised vectors (top-1); a hit needs
similarity >= threshold and the same model.  Per-request knobs: ``skip_cache`` and
``cache_similarity_threshold``.  Reference:
/root/reference/src/vllm_router/experimental/semantic_cache/semantic_cache.py:16-313,
db_adapters/faiss_adapter.py:14-134, semantic_cache_integration.py:26-306.

Embedding back ends: ``sentence-transformers`` when importable, otherwise a built-in
hashed character-n-gram embedder (deterministic, no model files).  The index is a numpy
matrix (flat IP search, what the reference's FAISS IndexFlatIP computes) persisted as
``.npz`` + JSON, never as pickle.
"""

from __future__ import annotations

import json
import os
import time
import uuid

import numpy as np
import xxhash
from prometheus_client import Gauge

from .log import init_logger
from .metrics import REGISTRY

logger = init_logger(__name__)

hits_g = Gauge("vllm:semantic_cache_hits", "Semantic cache hits", ["server"], registry=REGISTRY)
misses_g = Gauge("vllm:semantic_cache_misses", "Semantic cache misses", ["server"], registry=REGISTRY)
ratio_g = Gauge("vllm:semantic_cache_hit_ratio", "Semantic cache hit ratio", ["server"], registry=REGISTRY)
size_g = Gauge("vllm:semantic_cache_size", "Semantic cache entries", ["server"], registry=REGISTRY)
latency_g = Gauge("vllm:semantic_cache_latency", "Semantic cache lookup latency", ["server"], registry=REGISTRY)


class HashedNgramEmbedder:
    """Character 3..5-gram feature hashing into ``dim`` buckets with signed counts."""

    def __init__(self, dim: int = 384):
        self.dim = dim

    def encode(self, text: str) -> np.ndarray:
        v = np.zeros(self.dim, dtype=np.float32)
        t = " " + " ".join(text.lower().split()) + " "
        for n in (3, 4, 5):
            for i in range(max(0, len(t) - n + 1)):
                h = xxhash.xxh64_intdigest(t[i:i + n].encode())
                v[h % self.dim] += 1.0 if (h >> 63) & 1 else -1.0
        nrm = float(np.linalg.norm(v))
        return v / nrm if nrm > 0 else v


def _load_embedder(model_name: str):
    try:
        from sentence_transformers import SentenceTransformer  # type: ignore

        m = SentenceTransformer(model_name)

        class _ST:
            dim = m.get_sentence_embedding_dimension()

            @staticmethod
            def encode(text: str) -> np.ndarray:
                v = np.asarray(m.encode([text])[0], dtype=np.float32)
                n = float(np.linalg.norm(v))
                return v / n if n > 0 else v

        return _ST()
    except Exception:
        logger.info("sentence-transformers unavailable: using the built-in hashed n-gram embedder")
        return HashedNgramEmbedder()


class FlatIPIndex:
    def __init__(self, dim: int, cache_dir: str | None = None):
        self.dim = dim
        self.vectors = np.zeros((0, dim), dtype=np.float32)
        self.meta: list[dict] = []
        self.cache_dir = cache_dir
        if cache_dir:
            os.makedirs(cache_dir, exist_ok=True)
            self._load()

    def _paths(self):
        return (os.path.join(self.cache_dir, "index.npz"), os.path.join(self.cache_dir, "metadata.json"))

    def _load(self) -> None:
        ip, mp = self._paths()
        if os.path.exists(ip) and os.path.exists(mp):
            try:
                v = np.load(ip, allow_pickle=False)["vectors"]
                with open(mp) as f:
                    meta = json.load(f)
                if v.shape[1] == self.dim and len(meta) == v.shape[0]:
                    self.vectors, self.meta = v.astype(np.float32), meta
            except Exception as e:
                logger.warning("could not load semantic cache: %s", e)

    def save(self) -> None:
        if not self.cache_dir:
            return
        ip, mp = self._paths()
        np.savez(ip, vectors=self.vectors)
        with open(mp, "w") as f:
            json.dump(self.meta, f)

    def add(self, vec: np.ndarray, meta: dict) -> None:
        self.vectors = np.vstack([self.vectors, vec[None, :]])
        self.meta.append(meta)

    def search(self, vec: np.ndarray, k: int = 1):
        if not len(self.meta):
            return []
        sims = self.vectors @ vec
        idx = np.argsort(-sims)[:k]
        return [(float(sims[i]), self.meta[i]) for i in idx]

    def __len__(self) -> int:
        return len(self.meta)


class SemanticCache:
    def __init__(self, embedding_model: str = "all-MiniLM-L6-v2", cache_dir: str | None = None,
                 default_similarity_threshold: float = 0.95, embedder=None):
        self.embedder = embedder or _load_embedder(embedding_model)
        self.index = FlatIPIndex(self.embedder.dim, cache_dir)
        self.threshold = default_similarity_threshold
        self.hits = 0
        self.misses = 0

    @staticmethod
    def _text(messages: list[dict]) -> str:
        parts = []
        for m in messages:
            c = m.get("content", "")
            if isinstance(c, list):
                c = " ".join(p.get("text", "") for p in c if isinstance(p, dict))
            parts.append(f"{m.get('role', '')}: {c}")
        return "\n".join(parts)

    def search(self, messages: list[dict], model: str, similarity_threshold: float | None = None):
        thr = self.threshold if similarity_threshold is None else similarity_threshold
        res = self.index.search(self.embedder.encode(self._text(messages)), 1)
        if res and res[0][0] >= thr and res[0][1].get("model") == model:
            self.hits += 1
            return {**res[0][1], "similarity_score": res[0][0]}
        self.misses += 1
        return None

    def store(self, request_messages: list[dict], response_messages: list[dict], model: str,
              usage: dict | None = None) -> None:
        self.index.add(self.embedder.encode(self._text(request_messages)), {
            "model": model, "request_messages": request_messages,
            "response_messages": response_messages, "usage": usage or {}, "timestamp": time.time()})
        self.index.save()


_cache: SemanticCache | None = None


def initialize_semantic_cache(embedding_model, cache_dir, default_similarity_threshold, embedder=None):
    global _cache
    _cache = SemanticCache(embedding_model, cache_dir, default_similarity_threshold, embedder)
    return _cache


def get_semantic_cache() -> SemanticCache | None:
    return _cache


def _update_gauges() -> None:
    c = _cache
    if c is None:
        return
    hits_g.labels(server="router").set(c.hits)
    misses_g.labels(server="router").set(c.misses)
    tot = c.hits + c.misses
    ratio_g.labels(server="router").set(c.hits / tot if tot else 0.0)
    size_g.labels(server="router").set(len(c.index))


async def check_semantic_cache(request_json: dict):
    """-> OpenAI chat.completion dict on a hit, else None."""
    c = _cache
    if c is None or request_json.get("skip_cache") or request_json.get("stream"):
        return None
    messages = request_json.get("messages")
    if not messages:
        return None
    t0 = time.time()
    hit = c.search(messages, request_json.get("model", ""), request_json.get("cache_similarity_threshold"))
    latency_g.labels(server="router").set(time.time() - t0)
    _update_gauges()
    if hit is None:
        return None
    usage = hit.get("usage") or {"prompt_tokens": 0, "completion_tokens": 0, "total_tokens": 0}
    return {
        "id": "chatcmpl-" + uuid.uuid4().hex, "object": "chat.completion", "created": int(time.time()),
        "model": hit["model"],
        "choices": [{"index": i, "message": m, "finish_reason": "stop"}
                    for i, m in enumerate(hit["response_messages"])],
        "usage": usage, "cached": True, "similarity_score": hit["similarity_score"]}


async def store_in_semantic_cache(request_json: dict, response_body: bytes) -> None:
    c = _cache
    if c is None or request_json.get("skip_cache") or request_json.get("stream"):
        return
    try:
        d = json.loads(response_body)
        msgs = [ch["message"] for ch in d.get("choices", []) if "message" in ch]
        if msgs:
            c.store(request_json.get("messages", []), msgs, request_json.get("model", ""), d.get("usage"))
            _update_gauges()
    except Exception as e:
        logger.debug("semantic cache store skipped: %s", e)
