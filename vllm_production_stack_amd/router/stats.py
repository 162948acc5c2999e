"""Router-side statistics: engine /metrics scraper and per-engine request statistics.

Reference behaviour: /root/reference/src/vllm_router/stats/engine_stats.py:42-209 (six
gauges per engine, stale engines dropped), /root/reference/src/vllm_router/stats/request_stats.py:34-306
(sliding-window QPS / TTFT / latency, stage counters) and stats/log_stats.py:37-115.
Differences by design: the scraper is an asyncio task issuing all scrapes concurrently;
sliding windows keep running sums (O(1) per query instead of re-summing); inter-token
latency and decoding length are actually measured; per-request bookkeeping is dropped when
the request completes (the reference never prunes it).
"""

from __future__ import annotations

import asyncio
import time
from collections import deque
from dataclasses import dataclass

import aiohttp
from prometheus_client.parser import text_string_to_metric_families

from .log import init_logger

logger = init_logger(__name__)


# ------------------------------------------------------------------ engine stats
@dataclass
class EngineStats:
    num_running_requests: int = 0
    num_queuing_requests: int = 0
    gpu_prefix_cache_hit_rate: float = 0.0
    gpu_prefix_cache_hits_total: int = 0
    gpu_prefix_cache_queries_total: int = 0
    gpu_cache_usage_perc: float = 0.0

    @classmethod
    def from_vllm_scrape(cls, text: str) -> "EngineStats":
        s = cls()
        for fam in text_string_to_metric_families(text):
            for sample in fam.samples:
                n, v = sample.name, sample.value
                if n == "vllm:num_requests_running":
                    s.num_running_requests = int(v)
                elif n == "vllm:num_requests_waiting":
                    s.num_queuing_requests = int(v)
                elif n == "vllm:gpu_prefix_cache_hit_rate":
                    s.gpu_prefix_cache_hit_rate = v
                elif n == "vllm:gpu_prefix_cache_hits_total":
                    s.gpu_prefix_cache_hits_total = int(v)
                elif n == "vllm:gpu_prefix_cache_queries_total":
                    s.gpu_prefix_cache_queries_total = int(v)
                elif n == "vllm:gpu_cache_usage_perc":
                    s.gpu_cache_usage_perc = v
        if s.gpu_prefix_cache_queries_total and not s.gpu_prefix_cache_hit_rate:
            s.gpu_prefix_cache_hit_rate = s.gpu_prefix_cache_hits_total / s.gpu_prefix_cache_queries_total
        return s


class EngineStatsScraper:
    def __init__(self, scrape_interval: float, discovery_getter, api_key: str | None = None):
        if scrape_interval is None:
            raise ValueError("EngineStatsScraper must be initialized with scrape_interval")
        self.interval = scrape_interval
        self._get_discovery = discovery_getter
        self.api_key = api_key
        self.engine_stats: dict[str, EngineStats] = {}
        self._task: asyncio.Task | None = None
        self._known_urls: set[str] = set()
        # called with the URLs that left service discovery since the previous scrape, so that
        # per-engine router state (request statistics, prefix-trie ownership) does not outlive them
        self.on_engines_gone = None

    async def start(self) -> None:
        self._task = asyncio.create_task(self._loop())

    async def close(self) -> None:
        if self._task:
            self._task.cancel()

    def get_health(self) -> bool:
        return self._task is not None and not self._task.done()

    def get_engine_stats(self) -> dict[str, EngineStats]:
        return self.engine_stats

    async def _scrape_one(self, session: aiohttp.ClientSession, url: str) -> EngineStats | None:
        headers = {"Authorization": f"Bearer {self.api_key}"} if self.api_key else {}
        try:
            async with session.get(url + "/metrics", headers=headers,
                                   timeout=aiohttp.ClientTimeout(total=self.interval)) as r:
                r.raise_for_status()
                return EngineStats.from_vllm_scrape(await r.text())
        except asyncio.TimeoutError:
            # a busy engine answers late: its previous numbers stay in use, nothing failed
            logger.warning("metrics scrape of %s took longer than %.0f s; keeping the previous sample", url,
                           self.interval)
            return self.engine_stats.get(url)
        except Exception as e:
            logger.error("Failed to scrape metrics from %s: %s", url, e)
            return None

    async def scrape_once(self, session: aiohttp.ClientSession) -> None:
        sd = self._get_discovery()
        urls = [e.url for e in sd.get_endpoint_info()] if sd else []
        res = await asyncio.gather(*[self._scrape_one(session, u) for u in urls])
        # replace wholesale: engines that disappeared from discovery are dropped
        self.engine_stats = {u: s for u, s in zip(urls, res) if s is not None}
        gone = self._known_urls - set(urls)
        self._known_urls = set(urls)
        if gone and self.on_engines_gone is not None:
            self.on_engines_gone(sorted(gone))

    async def _loop(self) -> None:
        async with aiohttp.ClientSession() as session:
            while True:
                try:
                    await self.scrape_once(session)
                except asyncio.CancelledError:
                    raise
                except Exception as e:
                    logger.error("engine stats scrape failed: %s", e)
                await asyncio.sleep(self.interval)


# ------------------------------------------------------------------ request stats
@dataclass
class RequestStats:
    qps: float
    ttft: float
    in_prefill_requests: int
    in_decoding_requests: int
    finished_requests: int
    uptime: float
    avg_decoding_length: float
    avg_latency: float
    avg_itl: float
    num_swapped_requests: int


class MovingAverageMonitor:
    """Values inside a sliding time window with O(1) sum / average."""

    def __init__(self, sliding_window_size: float):
        self.sliding_window_size = sliding_window_size
        self.items: deque[tuple[float, float]] = deque()
        self.total = 0.0

    def _expire(self, now: float) -> None:
        lim = now - self.sliding_window_size
        items = self.items
        while items and items[0][0] < lim:
            self.total -= items.popleft()[1]
        if not items:
            self.total = 0.0

    def update(self, timestamp: float, value: float) -> None:
        self.items.append((timestamp, value))
        self.total += value
        self._expire(timestamp)

    def update_no_value(self, timestamp: float) -> None:
        self._expire(timestamp)

    def get_average(self) -> float:
        return self.total / len(self.items) if self.items else -1

    def get_sum(self) -> float:
        return self.total


class _EngineCounters:
    __slots__ = ("qps", "ttft", "latency", "decoding", "itl", "in_prefill", "in_decoding",
                 "finished", "swapped")

    def __init__(self, w: float):
        self.qps = MovingAverageMonitor(w)
        self.ttft = MovingAverageMonitor(w)
        self.latency = MovingAverageMonitor(w)
        self.decoding = MovingAverageMonitor(w)
        self.itl = MovingAverageMonitor(w)
        self.in_prefill = 0
        self.in_decoding = 0
        self.finished = 0
        self.swapped = 0


class RequestStatsMonitor:
    def __init__(self, sliding_window_size: float):
        if sliding_window_size is None:
            raise ValueError("RequestStatsMonitor must be initialized with sliding_window_size")
        self.sliding_window_size = sliding_window_size
        self.engines: dict[str, _EngineCounters] = {}
        # (engine_url, request_id) -> [start, first_token, last_chunk, n_chunks]
        self.inflight: dict[tuple[str, str], list] = {}
        self.first_query_time: float | None = None

    def _eng(self, url: str) -> _EngineCounters:
        e = self.engines.get(url)
        if e is None:
            e = self.engines[url] = _EngineCounters(self.sliding_window_size)
        return e

    def on_new_request(self, engine_url: str, request_id: str, timestamp: float) -> None:
        self.inflight[(engine_url, request_id)] = [timestamp, None, None, 0]
        e = self._eng(engine_url)
        e.in_prefill += 1
        e.qps.update(timestamp, 1)
        if self.first_query_time is None:
            self.first_query_time = timestamp

    def on_request_response(self, engine_url: str, request_id: str, timestamp: float) -> None:
        """First chunk of the response (time to first token)."""
        st = self.inflight.get((engine_url, request_id))
        if st is None or st[1] is not None:
            return
        st[1] = st[2] = timestamp
        e = self._eng(engine_url)
        e.in_prefill = max(0, e.in_prefill - 1)
        e.in_decoding += 1
        e.ttft.update(timestamp, timestamp - st[0])

    def on_request_chunk(self, engine_url: str, request_id: str, timestamp: float) -> None:
        """A later chunk: feeds the inter-token latency window."""
        st = self.inflight.get((engine_url, request_id))
        if st is None or st[2] is None:
            return
        self._eng(engine_url).itl.update(timestamp, timestamp - st[2])
        st[2] = timestamp
        st[3] += 1

    def on_request_complete(self, engine_url: str, request_id: str, timestamp: float) -> None:
        st = self.inflight.pop((engine_url, request_id), None)
        e = self._eng(engine_url)
        if st is not None and st[1] is None:
            e.in_prefill = max(0, e.in_prefill - 1)
        else:
            e.in_decoding = max(0, e.in_decoding - 1)
        e.finished += 1
        if st is not None:
            e.latency.update(timestamp, timestamp - st[0])
            if st[1] is not None:
                e.decoding.update(timestamp, timestamp - st[1])

    def on_request_swapped(self, engine_url: str, request_id: str, timestamp: float) -> None:
        self._eng(engine_url).swapped += 1

    def on_request_failed(self, engine_url: str, request_id: str, timestamp: float) -> None:
        st = self.inflight.pop((engine_url, request_id), None)
        e = self._eng(engine_url)
        if st is not None and st[1] is None:
            e.in_prefill = max(0, e.in_prefill - 1)
        elif st is not None:
            e.in_decoding = max(0, e.in_decoding - 1)

    def forget_engine(self, engine_url: str) -> None:
        self.engines.pop(engine_url, None)

    def get_request_stats(self, current_time: float) -> dict[str, RequestStats]:
        out = {}
        up = current_time - self.first_query_time if self.first_query_time else 0
        for url, e in self.engines.items():
            for m in (e.qps, e.ttft, e.latency, e.decoding, e.itl):
                m.update_no_value(current_time)
            out[url] = RequestStats(
                qps=e.qps.get_sum() / self.sliding_window_size, ttft=e.ttft.get_average(),
                in_prefill_requests=e.in_prefill, in_decoding_requests=e.in_decoding,
                finished_requests=e.finished, uptime=up,
                avg_decoding_length=e.decoding.get_average(), avg_latency=e.latency.get_average(),
                avg_itl=e.itl.get_average(), num_swapped_requests=e.swapped)
        return out


# ------------------------------------------------------------------ periodic stats log
def format_stats(endpoints, engine_stats: dict, request_stats: dict) -> str:
    lines = ["", "=" * 50]
    for ep in endpoints:
        lines.append(f"Server: {ep.url}")
        lines.append(f"Models: {', '.join(ep.model_names)}")
        es = engine_stats.get(ep.url)
        if es:
            lines.append(" Engine Stats: Running Requests: %d, Queued Requests: %d, GPU Cache Hit Rate: %.2f"
                         % (es.num_running_requests, es.num_queuing_requests, es.gpu_prefix_cache_hit_rate))
        else:
            lines.append(" Engine Stats: No stats available")
        rs = request_stats.get(ep.url)
        if rs:
            lines.append(" Request Stats: QPS: %.2f, Avg Latency: %s, Avg ITL: %s, Prefill Requests: %d, "
                         "Decoding Requests: %d, Swapped Requests: %d, Finished Requests: %d, Uptime: %.2f sec"
                         % (rs.qps, rs.avg_latency, rs.avg_itl, rs.in_prefill_requests,
                            rs.in_decoding_requests, rs.num_swapped_requests, rs.finished_requests,
                            rs.uptime))
        else:
            lines.append(" Request Stats: No stats available")
        lines.append("-" * 50)
    lines.append("=" * 50)
    return "\n".join(lines)
