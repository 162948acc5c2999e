#!/usr/bin/env python3
"""Generates the Grafana dashboards of the stack (serving dashboard JSON and the KV-tier
dashboard ConfigMap) from a compact panel list, so panels and the metric names they chart stay
reviewable.  Run from this directory:  python gen_dashboards.py

Metric names are the contract between engines/router and every consumer
(/root/reference/observability/vllm-dashboard.json:93-1286 charts the same series;
/root/reference/observability/lmcache-dashboard-cm.yaml:117-535 the KV-tier ones).
"""
import json

import yaml

DS = {"type": "prometheus", "uid": "prometheus"}


def panel(pid, title, exprs, x, y, w=8, h=8, kind="timeseries", unit="short", legend="{{instance}}"):
    targets = []
    for i, e in enumerate(exprs if isinstance(exprs, list) else [exprs]):
        expr, leg = (e if isinstance(e, tuple) else (e, legend))
        targets.append({"datasource": DS, "expr": expr, "legendFormat": leg, "refId": chr(65 + i)})
    return {"id": pid, "title": title, "type": kind, "datasource": DS,
            "gridPos": {"x": x, "y": y, "w": w, "h": h},
            "fieldConfig": {"defaults": {"unit": unit}, "overrides": []},
            "options": {"legend": {"displayMode": "list", "placement": "bottom"}} if kind == "timeseries" else {},
            "targets": targets}


def row(pid, title, y):
    return {"id": pid, "title": title, "type": "row", "collapsed": False,
            "gridPos": {"x": 0, "y": y, "w": 24, "h": 1}, "panels": []}


def quantiles(metric, by="instance"):
    return [(f"histogram_quantile({q}, sum by (le, {by}) (rate({metric}_bucket[1m])))", f"p{int(q * 100)} {{{{{by}}}}}")
            for q in (0.5, 0.9, 0.99)]


def serving_dashboard():
    p = []
    p.append(row(1, "Fleet", 0))
    p.append(panel(2, "Healthy engine replicas", "sum(vllm:healthy_pods_total)", 0, 1, 6, 6, "stat"))
    p.append(panel(3, "Requests running", "sum(vllm:num_requests_running)", 6, 1, 6, 6, "stat"))
    p.append(panel(4, "Requests waiting (autoscaling signal)", "sum(vllm:num_requests_waiting)", 12, 1, 6, 6, "stat"))
    p.append(panel(5, "Router QPS", "sum(vllm:current_qps)", 18, 1, 6, 6, "stat", "reqps"))
    p.append(row(10, "Latency", 7))
    p.append(panel(11, "Time to first token", quantiles("vllm:time_to_first_token_seconds"), 0, 8, 8, 8, unit="s"))
    p.append(panel(12, "Time per output token", quantiles("vllm:time_per_output_token_seconds"), 8, 8, 8, 8, unit="s"))
    p.append(panel(13, "End-to-end request latency", quantiles("vllm:e2e_request_latency_seconds"), 16, 8, 8, 8, unit="s"))
    p.append(panel(14, "Router-observed average latency", "vllm:avg_latency", 0, 16, 8, 8, unit="s", legend="{{server}}"))
    p.append(panel(15, "Router-observed inter-token latency", "vllm:avg_itl", 8, 16, 8, 8, unit="s", legend="{{server}}"))
    p.append(panel(16, "Average decoding length", "vllm:avg_decoding_length", 16, 16, 8, 8, legend="{{server}}"))
    p.append(row(20, "Engine load", 24))
    p.append(panel(21, "Running requests per engine", "vllm:num_requests_running", 0, 25))
    p.append(panel(22, "Waiting requests per engine", "vllm:num_requests_waiting", 8, 25))
    p.append(panel(23, "Swapped / preempted requests", "vllm:num_requests_swapped", 16, 25))
    p.append(panel(24, "Prefill vs decode in flight (router view)",
                   [("vllm:num_prefill_requests", "prefill {{server}}"), ("vllm:num_decoding_requests", "decode {{server}}")], 0, 33, 12))
    p.append(panel(25, "Per-engine QPS (router view)", "vllm:current_qps", 12, 33, 12, unit="reqps", legend="{{server}}"))
    p.append(row(30, "KV cache (HBM)", 41))
    p.append(panel(31, "HBM KV pool usage", "vllm:gpu_cache_usage_perc", 0, 42, unit="percentunit"))
    p.append(panel(32, "Prefix cache hit rate", "vllm:gpu_prefix_cache_hit_rate", 8, 42, unit="percentunit"))
    p.append(panel(33, "Prefix cache hits / queries (tokens/s)",
                   [("rate(vllm:gpu_prefix_cache_hits_total[1m])", "hits {{instance}}"),
                    ("rate(vllm:gpu_prefix_cache_queries_total[1m])", "queries {{instance}}")], 16, 42))
    p.append(row(40, "Throughput", 50))
    p.append(panel(41, "Prompt tokens/s", "rate(vllm:prompt_tokens_total[1m])", 0, 51, 12))
    p.append(panel(42, "Generated tokens/s", "rate(vllm:generation_tokens_total[1m])", 12, 51, 12))
    p.append(row(50, "Router process", 59))
    p.append(panel(51, "Router CPU %", "router_cpu_usage_percent", 0, 60, unit="percent"))
    p.append(panel(52, "Router memory %", "router_memory_usage_percent", 8, 60, unit="percent"))
    p.append(panel(53, "Router disk %", "router_disk_usage_percent", 16, 60, unit="percent"))
    return {"title": "MI355X serving stack", "uid": "vpsa-serving", "schemaVersion": 39, "version": 1,
            "refresh": "10s", "time": {"from": "now-30m", "to": "now"}, "timezone": "browser",
            "tags": ["vllm", "router", "mi355x"], "panels": p, "templating": {"list": []},
            "annotations": {"list": []}, "editable": True}


def kv_dashboard():
    p = []
    p.append(row(1, "KV tiers (host DRAM / disk / remote)", 0))
    p.append(panel(2, "Retrieve speed (tokens/s)", quantiles("lmcache:retrieve_speed"), 0, 1, 12, 8))
    p.append(panel(3, "Tier hit ratio",
                   "rate(lmcache:num_hit_tokens_total[1m]) / clamp_min(rate(lmcache:num_requested_tokens_total[1m]), 1)",
                   12, 1, 12, 8, unit="percentunit"))
    p.append(panel(4, "Hit tokens/s", "rate(lmcache:num_hit_tokens_total[1m])", 0, 9, 8))
    p.append(panel(5, "Requested tokens/s", "rate(lmcache:num_requested_tokens_total[1m])", 8, 9, 8))
    p.append(panel(6, "Host-tier usage (bytes)", "lmcache:local_cache_usage", 16, 9, 8, unit="bytes"))
    return {"title": "MI355X KV tiers", "uid": "vpsa-kv-tiers", "schemaVersion": 39, "version": 1,
            "refresh": "10s", "time": {"from": "now-30m", "to": "now"}, "tags": ["kv-cache"], "panels": p,
            "templating": {"list": []}, "annotations": {"list": []}, "editable": True}


if __name__ == "__main__":
    with open("serving-dashboard.json", "w") as f:
        json.dump(serving_dashboard(), f, indent=1)
        f.write("\n")
    cm = {"apiVersion": "v1", "kind": "ConfigMap",
          "metadata": {"name": "vpsa-kv-tier-dashboard", "namespace": "monitoring",
                       "labels": {"grafana_dashboard": "1"}},
          "data": {"kv-tier-dashboard.json": json.dumps(kv_dashboard(), indent=1)}}
    with open("kv-tier-dashboard-cm.yaml", "w") as f:
        f.write("# Grafana sidecar picks up ConfigMaps labelled grafana_dashboard=1.\n")
        yaml.safe_dump(cm, f, sort_keys=False)
