"""Routing strategies.

Semantics reproduce /root/reference/src/vllm_router/routers/routing_logic.py:50-526:
  roundrobin             sorted-by-URL rotation with a global counter
  session                consistent-hash ring on a header value; lowest-QPS without it
  prefixaware            longest chunk-hash prefix match in a trie, random tie-break, insert
  kvaware                KV-controller lookup of the prompt's token chain; session fallback
  disaggregated_prefill  prefill pool / decode pool selection by model label
The ring, trie and counter run in the native routing core when it is built.

Interface (structural - tests drive it with duck-typed fakes):
    route_request(endpoints, engine_stats, request_stats, request, request_json=None) -> url
where ``endpoints`` have ``.url``, ``request`` has ``.headers``.
"""

from __future__ import annotations

import abc
import enum
import asyncio
import inspect
import random

import aiohttp

from .log import init_logger
from .native import HashRing, HashTrie

logger = init_logger(__name__)


class RoutingLogic(str, enum.Enum):
    ROUND_ROBIN = "roundrobin"
    SESSION_BASED = "session"
    KVAWARE = "kvaware"
    PREFIXAWARE = "prefixaware"
    DISAGGREGATED_PREFILL = "disaggregated_prefill"


class RoutingInterface(abc.ABC):
    def _qps_routing(self, endpoints, request_stats) -> str:
        lowest, ret = float("inf"), None
        for info in endpoints:
            st = request_stats.get(info.url)
            if st is None:
                return info.url  # an engine with no traffic yet wins immediately
            if st.qps < lowest:
                lowest, ret = st.qps, info.url
        return ret

    def _update_hash_ring(self, endpoints) -> None:
        new = {e.url for e in endpoints}
        cur = self.hash_ring.get_nodes()
        for n in cur - new:
            self.hash_ring.remove_node(n)
        for n in new - cur:
            self.hash_ring.add_node(n)

    @abc.abstractmethod
    def route_request(self, endpoints, engine_stats, request_stats, request, request_json=None):
        raise NotImplementedError

    async def close(self) -> None:
        return None


class RoundRobinRouter(RoutingInterface):
    def __init__(self):
        from . import workers

        self.req_id = 0
        self._key = None
        self._sorted: list[str] = []
        self._shared = workers.shared_rr_counter()  # one position for all router workers

    def route_request(self, endpoints, engine_stats, request_stats, request, request_json=None):
        key = tuple(e.url for e in endpoints)
        if key != self._key:
            self._sorted = sorted(key)
            self._key = key
        if self._shared is not None:
            with self._shared.get_lock():
                n = self._shared.value
                self._shared.value = n + 1
        else:
            n = self.req_id
            self.req_id += 1
        return self._sorted[n % len(self._sorted)]


class SessionRouter(RoutingInterface):
    def __init__(self, session_key: str | None = None):
        if session_key is None:
            raise ValueError("SessionRouter must be initialized with a session_key")
        self.session_key = session_key
        self.hash_ring = HashRing()

    def route_request(self, endpoints, engine_stats, request_stats, request, request_json=None):
        session_id = request.headers.get(self.session_key, None)
        self._update_hash_ring(endpoints)
        if session_id is None:
            return self._qps_routing(endpoints, request_stats)
        return self.hash_ring.get_node(str(session_id))


def extract_prompt_text(request_json: dict) -> str:
    """Chat: every message's text joined by newlines (multimodal lists contribute their text
    parts); otherwise the ``prompt`` field."""
    if "messages" in request_json:
        parts = []
        for m in request_json["messages"] or []:
            c = m.get("content", "")
            if isinstance(c, list):
                parts.append(" ".join(p.get("text", "") for p in c
                                      if isinstance(p, dict) and p.get("type") == "text"))
            elif c is not None:
                parts.append(str(c))
        return "\n".join(parts)
    p = request_json.get("prompt", "")
    if isinstance(p, list):
        p = p[0] if p and isinstance(p[0], str) else ""
    return p or ""


class PrefixAwareRouter(RoutingInterface):
    """Longest-prefix affinity with load-aware tie-breaking.

    A match is honoured when it covers at least ``min_match_ratio`` of the prompt (a mere
    shared system prompt must not pin every conversation to one engine); among the
    candidate endpoints the one with the fewest router-tracked in-flight requests wins.  Equals
    are taken in rotation, from a position shared by all router workers: a burst of new
    conversations (every user of a benchmark round at once, spread over several workers that
    each see only their own in-flight counts) lands evenly on the replicas instead of
    binomially (320 users on 8 replicas: 40 each instead of 40 +- 6, and the slowest replica
    sets the round time).  ``min_match_ratio=0`` and no request statistics give the reference
    behaviour: any match wins among the matched endpoints."""

    def __init__(self, max_nodes: int = 4_000_000, min_match_ratio: float = 0.5):
        from . import workers

        path = workers.shared_trie_path()
        # with several router workers the trie lives in the supervisor process: every worker sees
        # the prefixes any of them inserted
        self.shared = workers.SharedTrieClient(path) if path else None
        self.hashtrie = None if self.shared else HashTrie(chunk_size=128, max_nodes=max_nodes)
        self.min_match_ratio = min_match_ratio
        self._turn = workers.shared_rr_counter()  # one rotation for all router workers
        self._local_turn = random.randrange(1 << 16)

    def _next_turn(self) -> int:
        if self._turn is not None:
            with self._turn.get_lock():
                n = self._turn.value
                self._turn.value = n + 1
            return n
        self._local_turn += 1
        return self._local_turn

    def _least_loaded(self, urls: list[str], request_stats, turn: int | None = None) -> str:
        def load(u):
            st = request_stats.get(u) if request_stats else None
            return 0 if st is None else st.in_prefill_requests + st.in_decoding_requests

        best = min(load(u) for u in urls)
        equals = [u for u in urls if load(u) == best]
        if len(equals) == 1:
            return equals[0]
        # the first of the equals at or after the rotating position in the (sorted) candidate list
        start = (self._next_turn() if turn is None else turn) % len(urls)
        for k in range(len(urls)):
            u = urls[(start + k) % len(urls)]
            if u in equals:
                return u
        return equals[0]

    def _choose(self, prompt, available, matched_len, matched, request_stats, turn: int | None = None) -> str:
        strong = matched and matched_len >= self.min_match_ratio * len(prompt)
        candidates = sorted(matched) if strong else sorted(available)
        return self._least_loaded(candidates, request_stats, turn)

    def route_request(self, endpoints, engine_stats, request_stats, request, request_json=None):
        prompt = extract_prompt_text(request_json or {})
        available = {e.url for e in endpoints}
        if self.shared is not None:
            return self._route_shared(prompt, available, request_stats)
        matched_len, matched = self.hashtrie.longest_prefix_match(prompt, available)
        selected = self._choose(prompt, available, matched_len, matched, request_stats)
        self.hashtrie.insert(prompt, selected)
        return selected

    async def _route_shared(self, prompt, available, request_stats, turn: int | None = None) -> str:
        matched_len, matched = await self.shared.longest_prefix_match(prompt, available)
        selected = self._choose(prompt, available, matched_len, matched, request_stats, turn)
        await self.shared.insert(prompt, selected)
        return selected

    async def pick(self, prompt: str, urls: set[str], request_stats, turn: int | None = None) -> str:
        """The endpoint among ``urls`` for ``prompt`` (and remember the choice): what
        ``route_request`` does, for callers that keep several pools in one trie.  ``turn``: the
        rotation position to break ties with (default: the next one)."""
        if self.shared is not None:
            return await self._route_shared(prompt, urls, request_stats, turn)
        matched_len, matched = self.hashtrie.longest_prefix_match(prompt, urls)
        selected = self._choose(prompt, urls, matched_len, matched, request_stats, turn)
        self.hashtrie.insert(prompt, selected)
        return selected

    def forget_endpoint(self, url: str) -> None:
        if self.shared is not None:
            asyncio.ensure_future(self.shared.remove_endpoint(url))
        else:
            self.hashtrie.remove_endpoint(url)

    async def close(self) -> None:
        if self.shared is not None:
            await self.shared.close()


class KvawareRouter(RoutingInterface):
    """Route to the instance that already holds the longest KV prefix of the prompt."""

    def __init__(self, controller=None, session_key: str | None = None, kv_aware_threshold: int = 2000,
                 controller_port: int = 9000, tokenize=None, fallback: str = "session"):
        from . import workers
        from ..controller import KVController

        # what to do when no replica holds enough of the prompt: "session" (the reference,
        # /root/reference/src/vllm_router/routers/routing_logic.py:318-322: hash ring on the session
        # header) or "load" (fewest in-flight requests, equals in rotation: new conversations are
        # dealt out evenly instead of binomially)
        self.fallback = fallback
        self._turn = workers.shared_rr_counter()
        self._local_turn = random.randrange(1 << 16)

        self.controller = controller or KVController()
        self.controller_port = controller_port
        self.session_key = session_key
        self.threshold = kv_aware_threshold
        self.hash_ring = HashRing()
        self._tokenize = tokenize
        self._runner = None
        self._http: aiohttp.ClientSession | None = None
        self._tokenizers: dict[str, object] = {}

    async def start(self) -> None:
        from ..controller import start_controller

        if self.controller_port:
            try:
                self._runner = await start_controller(self.controller, "0.0.0.0", self.controller_port,
                                                      self._tokenize_text)
                logger.info("KV controller listening on :%d", self.controller_port)
            except OSError as e:
                logger.error("could not start KV controller on port %d: %s", self.controller_port, e)

    async def close(self) -> None:
        if self._runner is not None:
            await self._runner.cleanup()
        if self._http is not None:
            await self._http.close()

    async def _tokenize_text(self, model: str, prompt: str) -> list[int]:
        return await self._tokens_for({"model": model, "prompt": prompt}, None)

    async def _tokens_for(self, request_json: dict, endpoints) -> list[int]:
        if self._tokenize is not None:
            r = self._tokenize(request_json)
            return await r if inspect.isawaitable(r) else r
        model = request_json.get("model", "")
        tok = self._tokenizers.get(model)
        if tok is None:
            tok = self._local_tokenizer(model)
            self._tokenizers[model] = tok
        if tok:
            if "messages" in request_json:
                return tok.encode_chat(request_json["messages"], True, request_json.get("tools"))
            return tok.encode(str(request_json.get("prompt", "")))
        if not endpoints:
            return []
        if self._http is None:
            self._http = aiohttp.ClientSession()
        body = {"model": model}
        if "messages" in request_json:
            body["messages"] = request_json["messages"]
        else:
            body["prompt"] = request_json.get("prompt", "")
        async with self._http.post(endpoints[0].url + "/tokenize", json=body,
                                   timeout=aiohttp.ClientTimeout(total=10)) as r:
            r.raise_for_status()
            return (await r.json()).get("tokens", [])

    @staticmethod
    def _local_tokenizer(model: str):
        """Tokenizer constructed without any network: checkpoint dir or known architecture."""
        try:
            from ..engine.config import resolve_model
            from ..engine.tokenizer import load_tokenizer

            cfg, ckpt = resolve_model(model)
            return load_tokenizer(ckpt, cfg.vocab_size)
        except Exception:
            return False

    async def route_request(self, endpoints, engine_stats, request_stats, request, request_json=None):
        from ..engine.block_manager import chain_hashes

        request_json = request_json or {}
        try:
            tokens = await self._tokens_for(request_json, endpoints)
        except Exception as e:
            logger.warning("kvaware: tokenization failed (%s); falling back", e)
            tokens = []
        url = None
        if tokens:
            matches = self.controller.lookup(chain_hashes(tokens), None)
            by_url = {}
            urls = {e.url for e in endpoints}
            for inst, n in matches.items():
                u = self.controller.url_of(inst)
                if u in urls:
                    by_url[u] = max(by_url.get(u, 0), n)
            if by_url:
                best_url, matched = max(by_url.items(), key=lambda kv: kv[1])
                if matched >= max(len(tokens) - self.threshold, 0):
                    url = best_url
        if url is None and self.fallback == "load":
            url = self._least_loaded(sorted(e.url for e in endpoints), request_stats)
        elif url is None:
            session_id = request.headers.get(self.session_key, None) if self.session_key else None
            self._update_hash_ring(endpoints)
            if session_id is None:
                url = self._qps_routing(endpoints, request_stats)
            else:
                url = self.hash_ring.get_node(str(session_id))
        return url

    _next_turn = PrefixAwareRouter._next_turn
    _least_loaded = PrefixAwareRouter._least_loaded


class DisaggregatedPrefillRouter(RoutingInterface):
    """Two pools selected by model label.  A conversation stays where its KV is: the prefill
    replica that computed the earlier turns has them in its prefix cache, and the decode replica
    that answered them holds their pages, so a follow-up turn recomputes and ships only what is
    new.  Both choices therefore go through one prefix trie (``route_pair``); a prompt that matches
    nothing long enough (a new conversation: at most the shared system prompt) takes the
    least-loaded endpoint of the pool (fewest router-tracked in-flight requests, equals in a
    rotation shared by all router workers), so N prefill x M decode deployments stay balanced.
    The reference runs one replica of each (/root/reference/src/vllm_router/service_discovery.py:321-341),
    where neither question arises."""

    def __init__(self, prefill_model_labels: list[str], decode_model_labels: list[str],
                 prefix_trie_max_nodes: int = 4_000_000):
        self.prefill_model_labels = prefill_model_labels or []
        self.decode_model_labels = decode_model_labels or []
        self._rr = 0
        self._affinity = PrefixAwareRouter(prefix_trie_max_nodes)

    def _pick(self, pool, request_stats) -> str:
        def load(e):
            st = request_stats.get(e.url)
            if st is None:
                return (0, 0.0)
            return (st.in_prefill_requests + st.in_decoding_requests, st.qps)

        best = min(load(e) for e in pool)
        ties = sorted(e.url for e in pool if load(e) == best)
        self._rr += 1
        return ties[self._rr % len(ties)]

    def _pool(self, endpoints, prefill: bool):
        labels = self.prefill_model_labels if prefill else self.decode_model_labels
        pool = [e for e in endpoints if e.model_label in labels]
        if not pool:
            raise ValueError(f"no {'prefill' if prefill else 'decode'} endpoint with labels {labels}")
        return pool

    def route_request(self, endpoints, engine_stats, request_stats, request, request_json=None):
        """Load-only choice within one pool (``max_tokens == 1`` marks the prefill hop)."""
        is_prefill = bool(request_json) and request_json.get("max_tokens", 0) == 1
        return self._pick(self._pool(endpoints, is_prefill), request_stats)

    async def route_pair(self, endpoints, request_stats, request_json) -> tuple[str, str]:
        """(prefill url, decode url) for a request, with conversation affinity in both pools."""
        prompt = extract_prompt_text(request_json or {})
        pre = {e.url for e in self._pool(endpoints, True)}
        dec = {e.url for e in self._pool(endpoints, False)}
        # one rotation position per request for both pools: a burst of new conversations decides its
        # decode replicas before any of them shows up in the decode pool's in-flight counts, so the
        # rotation alone spreads them, and it must step by one per request in either pool
        turn = self._affinity._next_turn()
        return (await self._affinity.pick(prompt, pre, request_stats, turn),
                await self._affinity.pick(prompt, dec, request_stats, turn))

    def forget_endpoint(self, url: str) -> None:
        self._affinity.forget_endpoint(url)

    async def close(self) -> None:
        await self._affinity.close()


def initialize_routing_logic(routing_logic, *, session_key=None, lmcache_controller_port=9000,
                             prefill_model_labels=None, decode_model_labels=None,
                             kv_aware_threshold=2000, prefix_trie_max_nodes=4_000_000,
                             controller=None, kv_aware_fallback="session") -> RoutingInterface:
    rl = RoutingLogic(routing_logic) if not isinstance(routing_logic, RoutingLogic) else routing_logic
    if rl is RoutingLogic.ROUND_ROBIN:
        logger.info("Initializing round-robin routing logic")
        return RoundRobinRouter()
    if rl is RoutingLogic.SESSION_BASED:
        logger.info("Initializing session-based routing logic with session key %s", session_key)
        return SessionRouter(session_key)
    if rl is RoutingLogic.KVAWARE:
        logger.info("Initializing kvaware routing logic")
        return KvawareRouter(controller, session_key, kv_aware_threshold, lmcache_controller_port,
                             fallback=kv_aware_fallback)
    if rl is RoutingLogic.PREFIXAWARE:
        logger.info("Initializing prefix-aware routing logic")
        return PrefixAwareRouter(prefix_trie_max_nodes)
    if rl is RoutingLogic.DISAGGREGATED_PREFILL:
        logger.info("Initializing disaggregated prefill routing logic")
        return DisaggregatedPrefillRouter(prefill_model_labels, decode_model_labels, prefix_trie_max_nodes)
    raise ValueError(f"Invalid routing logic {routing_logic}")
