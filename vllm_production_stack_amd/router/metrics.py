"""Router Prometheus series (names and the ``server`` label are consumed by the Grafana
dashboards and the autoscaler).  Reference:
/root/reference/src/vllm_router/services/metrics_service/__init__.py:5-47,
/root/reference/src/vllm_router/routers/metrics_router.py:42-123."""

from __future__ import annotations

import time

import psutil
from prometheus_client import CollectorRegistry, Counter, Gauge, generate_latest

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
# demand that found no replica (none deployed, all asleep, scaled to zero): the signal a
# scale-from-zero trigger needs, since engine gauges do not exist while there is no engine
unroutable_requests = Counter("vllm:router_unroutable_requests", "Requests for a model that no awake engine serves",
                              ["model"], registry=REGISTRY)


def count_unroutable(model: str, max_series: int = 64) -> None:
    """The model label is client input: bounded in length, and in how many distinct values get a
    series of their own (the rest are counted under ``other``)."""
    model = model[:128]
    known = unroutable_requests._metrics
    if (model,) not in known and len(known) >= max_series:
        model = "other"
    unroutable_requests.labels(model=model).inc()


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


# ------------------------------------------------------------------ one truth under --router-workers
# Reference: one process, so one registry is the whole router
# (/root/reference/src/vllm_router/routers/metrics_router.py:57-123).  With N worker processes a
# scrape reaches one of them; it answers for all.
#
# How a family of one name is merged over the workers' snapshots:
#   SUMMED    per-worker shares of one quantity: request rates, in-flight counts, cache counters
#   AVERAGED  per-worker averages of the same population, weighted by the worker's request rate for
#             that server (a worker that saw no request reports -1 and has no weight)
#   counters and histograms (type counter / histogram): summed
#   everything else: what all workers observe alike (engine gauges, pod health, host usage) is
#             taken from the worker that answers the scrape
SUMMED = {"vllm:current_qps", "vllm:num_prefill_requests", "vllm:num_decoding_requests",
          "vllm:num_requests_swapped", "vllm:semantic_cache_hits", "vllm:semantic_cache_misses",
          "vllm:semantic_cache_size"}
AVERAGED = {"vllm:avg_latency", "vllm:avg_itl", "vllm:avg_decoding_length", "vllm:semantic_cache_latency"}
WEIGHT = "vllm:current_qps"


def snapshot(state) -> bytes:
    """This worker's registry as a msgpack blob: [[family, type, doc, [[sample, labels, value], ...]], ...]."""
    import msgpack

    refresh(state)
    fams = []
    for fam in REGISTRY.collect():
        fams.append([fam.name, fam.type, fam.documentation,
                     [[s.name, dict(s.labels), float(s.value)] for s in fam.samples]])
    return msgpack.packb(fams)


def merge_snapshots(own: int, blobs: list[tuple[int, bytes]]) -> bytes:
    """Prometheus text for the whole router from the workers' snapshots; ``own`` is the index of
    the worker that answers (its copy is used for what is not summed or averaged)."""
    import msgpack
    from prometheus_client.core import Metric

    per_worker = {w: msgpack.unpackb(b, raw=False) for w, b in blobs}
    if own not in per_worker:
        own = next(iter(per_worker))
    # request-rate weights: (worker, server) -> qps
    weight: dict[tuple[int, str], float] = {}
    for w, fams in per_worker.items():
        for name, _typ, _doc, samples in fams:
            if name == WEIGHT:
                for _sn, labels, v in samples:
                    weight[(w, labels.get("server", ""))] = max(v, 0.0)
    order: list[str] = []
    meta: dict[str, tuple[str, str]] = {}
    acc: dict[str, dict[tuple, list]] = {}  # family -> (sample name, labels) -> [sum, weight sum]
    for w, fams in per_worker.items():
        for name, typ, doc, samples in fams:
            if name not in meta:
                meta[name] = (typ, doc)
                order.append(name)
                acc[name] = {}
            summed = name in SUMMED or typ in ("counter", "histogram", "summary")
            for sn, labels, v in samples:
                key = (sn, tuple(sorted(labels.items())))
                if summed:
                    if sn.endswith("_created"):  # a counter's creation time: the earliest
                        cur = acc[name].get(key)
                        acc[name][key] = [v if cur is None else min(cur[0], v), 1.0]
                    else:
                        acc[name].setdefault(key, [0.0, 1.0])[0] += v
                elif name in AVERAGED:
                    wt = weight.get((w, labels.get("server", "")), 0.0)
                    slot = acc[name].setdefault(key, [0.0, 0.0])
                    if v >= 0 and wt > 0:
                        slot[0] += v * wt
                        slot[1] += wt
                elif w == own:
                    acc[name][key] = [v, 1.0]
    # the cache's hit ratio follows from the summed hits and misses
    if "vllm:semantic_cache_hit_ratio" in acc:
        for key in list(acc["vllm:semantic_cache_hit_ratio"]):
            lab = key[1]
            h = acc.get("vllm:semantic_cache_hits", {}).get(("vllm:semantic_cache_hits", lab), [0.0])[0]
            m = acc.get("vllm:semantic_cache_misses", {}).get(("vllm:semantic_cache_misses", lab), [0.0])[0]
            acc["vllm:semantic_cache_hit_ratio"][key] = [h / (h + m) if h + m > 0 else 0.0, 1.0]

    class _Merged:
        def collect(self_inner):
            for name in order:
                typ, doc = meta[name]
                m = Metric(name, doc, typ)
                for (sn, labels), (num, den) in acc[name].items():
                    if name in AVERAGED:
                        val = num / den if den > 0 else -1.0
                    else:
                        val = num
                    m.add_sample(sn, dict(labels), val)
                yield m

    reg = CollectorRegistry()
    reg.register(_Merged())
    return generate_latest(reg)


async def render_all_workers(state, client, own: int) -> bytes:
    """``/metrics`` of one worker of several: publish this worker's fresh snapshot, then merge all."""
    mine = snapshot(state)
    try:
        await client.put_stats(own, mine)
        blobs = await client.all_stats()
    except Exception:  # the supervisor is gone or restarting: this worker's own view is better than a 500
        blobs = [(own, mine)]
    return merge_snapshots(own, blobs)
