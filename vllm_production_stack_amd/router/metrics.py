"""Router Prometheus series (names and the ``server`` label are consumed by the Grafana
dashboards and the autoscaler).  Reference:
/root/reference/src/vllm_router/services/metrics_service/__init__.py:5-47,
/root/reference/src/vllm_router/routers/metrics_router.py:42-123."""

from __future__ import annotations

import time

import psutil
from prometheus_client import CollectorRegistry, Gauge, generate_latest

REGISTRY = CollectorRegistry()


def _g(name: str, doc: str, labels=("server",)) -> Gauge:
    return Gauge(name, doc, list(labels), registry=REGISTRY)


num_requests_running = _g("vllm:num_requests_running", "Number of running requests")
num_requests_waiting = _g("vllm:num_requests_waiting", "Number of waiting requests")
gpu_prefix_cache_hit_rate = _g("vllm:gpu_prefix_cache_hit_rate", "GPU prefix cache hit rate")
gpu_prefix_cache_hits_total = _g("vllm:gpu_prefix_cache_hits_total", "GPU prefix cache hits")
gpu_prefix_cache_queries_total = _g("vllm:gpu_prefix_cache_queries_total", "GPU prefix cache queries")
current_qps = _g("vllm:current_qps", "Current queries per second")
avg_decoding_length = _g("vllm:avg_decoding_length", "Average decoding length")
num_prefill_requests = _g("vllm:num_prefill_requests", "Number of prefill requests")
num_decoding_requests = _g("vllm:num_decoding_requests", "Number of decoding requests")
healthy_pods_total = _g("vllm:healthy_pods_total", "Number of healthy serving engines")
avg_latency = _g("vllm:avg_latency", "Average end-to-end request latency")
avg_itl = _g("vllm:avg_itl", "Average inter-token latency")
num_requests_swapped = _g("vllm:num_requests_swapped", "Number of swapped requests")
router_cpu = Gauge("router_cpu_usage_percent", "CPU usage percent", registry=REGISTRY)
router_mem = Gauge("router_memory_usage_percent", "Memory usage percent", registry=REGISTRY)
router_disk = Gauge("router_disk_usage_percent", "Disk usage percent", registry=REGISTRY)
routing_latency = Gauge("vllm:router_decision_seconds", "Last routing decision latency", registry=REGISTRY)


def refresh(state) -> None:
    """Recompute every gauge from the live state objects."""
    now = time.time()
    rstats = state.request_stats_monitor.get_request_stats(now)
    estats = state.engine_stats_scraper.get_engine_stats()
    endpoints = state.service_discovery.get_endpoint_info()
    for url, s in rstats.items():
        current_qps.labels(server=url).set(s.qps)
        avg_decoding_length.labels(server=url).set(s.avg_decoding_length)
        num_prefill_requests.labels(server=url).set(s.in_prefill_requests)
        num_decoding_requests.labels(server=url).set(s.in_decoding_requests)
        avg_latency.labels(server=url).set(s.avg_latency)
        avg_itl.labels(server=url).set(s.avg_itl)
        num_requests_swapped.labels(server=url).set(s.num_swapped_requests)
    for url, e in estats.items():
        num_requests_running.labels(server=url).set(e.num_running_requests)
        num_requests_waiting.labels(server=url).set(e.num_queuing_requests)
        gpu_prefix_cache_hit_rate.labels(server=url).set(e.gpu_prefix_cache_hit_rate)
        gpu_prefix_cache_hits_total.labels(server=url).set(e.gpu_prefix_cache_hits_total)
        gpu_prefix_cache_queries_total.labels(server=url).set(e.gpu_prefix_cache_queries_total)
    # an endpoint is healthy when discovery lists it and it is awake
    live = {ep.url for ep in endpoints}
    for ep in endpoints:
        healthy_pods_total.labels(server=ep.url).set(0 if ep.sleep else 1)
    for g in (healthy_pods_total,):
        for (url,) in list(g._metrics):
            if url not in live:
                g.remove(url)
    router_cpu.set(psutil.cpu_percent())
    router_mem.set(psutil.virtual_memory().percent)
    router_disk.set(psutil.disk_usage("/").percent)


def render(state) -> bytes:
    refresh(state)
    return generate_latest(REGISTRY)
