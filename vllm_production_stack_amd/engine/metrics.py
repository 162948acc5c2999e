"""Prometheus series exported by an engine replica.

Names follow what the router scraper, the autoscaler rule and the Grafana dashboards of
the reference read (/root/reference/src/vllm_router/stats/engine_stats.py:63-76,
/root/reference/observability/vllm-dashboard.json, /root/reference/observability/prom-adapter.yaml,
/root/reference/observability/lmcache-dashboard-cm.yaml).
"""

from __future__ import annotations

from prometheus_client import CollectorRegistry, Counter, Gauge, Histogram, generate_latest

_LAT = (0.001, 0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.1, 0.25, 0.5, 0.75, 1.0, 2.5, 5.0, 7.5, 10.0,
        20.0, 40.0, 80.0)
_TPOT = (0.001, 0.0025, 0.005, 0.0075, 0.01, 0.02, 0.03, 0.04, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3,
         0.4, 0.5, 0.75, 1.0, 2.5)


class EngineMetrics:
    def __init__(self, model_name: str):
        self.registry = CollectorRegistry()
        lab = ["model_name"]
        self._lv = {"model_name": model_name}

        def gauge(name, doc):
            return Gauge(name, doc, lab, registry=self.registry).labels(**self._lv)

        def counter(name, doc):
            return Counter(name, doc, lab, registry=self.registry).labels(**self._lv)

        def hist(name, doc, buckets):
            return Histogram(name, doc, lab, buckets=buckets, registry=self.registry).labels(**self._lv)

        self.running = gauge("vllm:num_requests_running", "Requests currently on the GPU")
        self.waiting = gauge("vllm:num_requests_waiting", "Requests waiting to be scheduled")
        self.swapped = gauge("vllm:num_requests_swapped", "Requests swapped to CPU")
        self.cache_usage = gauge("vllm:gpu_cache_usage_perc", "KV pool usage (1 = full)")
        self.hit_rate = gauge("vllm:gpu_prefix_cache_hit_rate", "Prefix cache hit rate")
        self.cpu_cache_usage = gauge("vllm:cpu_cache_usage_perc", "Host KV tier usage")
        self.hits = counter("vllm:gpu_prefix_cache_hits", "Prefix cache hit tokens")
        self.queries = counter("vllm:gpu_prefix_cache_queries", "Prefix cache queried tokens")
        self.prompt_tokens = counter("vllm:prompt_tokens", "Prompt tokens processed")
        self.generation_tokens = counter("vllm:generation_tokens", "Tokens generated")
        self.preemptions = counter("vllm:num_preemptions", "Sequence preemptions")
        self.success = counter("vllm:request_success", "Finished requests")
        self.ttft = hist("vllm:time_to_first_token_seconds", "Time to first token", _LAT)
        self.tpot = hist("vllm:time_per_output_token_seconds", "Time per output token", _TPOT)
        self.e2e = hist("vllm:e2e_request_latency_seconds", "End-to-end request latency", _LAT)
        self.queue_time = hist("vllm:request_queue_time_seconds", "Time in the waiting queue", _LAT)
        # KV tiering (LMCache-dashboard compatible names)
        self.kv_hit_tokens = counter("lmcache:num_hit_tokens", "Tokens served from lower KV tiers")
        self.kv_req_tokens = counter("lmcache:num_requested_tokens", "Tokens looked up in lower KV tiers")
        self.kv_local_usage = gauge("lmcache:local_cache_usage", "Bytes held in the host KV tier")
        self.kv_retrieve_speed = hist("lmcache:retrieve_speed", "KV retrieve speed (tokens/s)",
                                      (1e3, 1e4, 5e4, 1e5, 2e5, 5e5, 1e6, 2e6, 5e6, 1e7))
        self.kv_store_speed = hist("lmcache:store_speed", "KV store speed (tokens/s)",
                                   (1e3, 1e4, 5e4, 1e5, 2e5, 5e5, 1e6, 2e6, 5e6, 1e7))
        self._last_hits = 0
        self._last_queries = 0
        self._last_preempt = 0

    def update_scheduler(self, scheduler, block_manager) -> None:
        self.running.set(len(scheduler.running))
        self.waiting.set(len(scheduler.waiting))
        self.cache_usage.set(block_manager.usage())
        dq = block_manager.queries - self._last_queries
        dh = block_manager.hits - self._last_hits
        if dq > 0:
            self.queries.inc(dq)
        if dh > 0:
            self.hits.inc(dh)
        self._last_queries, self._last_hits = block_manager.queries, block_manager.hits
        if block_manager.queries:
            self.hit_rate.set(block_manager.hits / block_manager.queries)
        dp = scheduler.num_preemptions - self._last_preempt
        if dp > 0:
            self.preemptions.inc(dp)
        self._last_preempt = scheduler.num_preemptions

    def render(self) -> bytes:
        return generate_latest(self.registry)
